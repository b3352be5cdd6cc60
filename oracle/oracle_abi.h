/*
 * oracle_abi.h -- TEST INFRASTRUCTURE, not product code.
 *
 * One flat C ABI shared by the two CPU checkers:
 *   - oracle/mifc_oracle.cc   : from-scratch CPU restatement      (prefix mifcorc_)
 *   - oracle/ref_shim.cc      : the real reference, compiled from
 *                               /root/reference where it lies      (prefix mifcref_)
 * Both libraries export the same entry points so tests can run the same
 * inputs through either and compare bit for bit.
 *
 * Conventions follow the reference (FieldCalculations.h:102-107): nx, ny,
 * input fields, scalars, "compute" last, output, fDefined (in/out), undef.
 * `int* fdefined` carries miutil::ValuesDefined (0 ALL, 1 NONE, 2 SOME;
 * FieldDefined.h:41). Return value 1/0 mirrors the reference's bool.
 * `unit` is a NUL-terminated string ("celsius", "kelvin", "1", ...).
 */
#ifndef MIFC_ORACLE_ABI_H
#define MIFC_ORACLE_ABI_H

#ifndef MIFC_ORACLE_PREFIX
#error "define MIFC_ORACLE_PREFIX (mifcorc_ or mifcref_) before including oracle_abi.h"
#endif

#define MIFC_OCAT2(a, b) a##b
#define MIFC_OCAT(a, b) MIFC_OCAT2(a, b)
#define ORC(name) MIFC_OCAT(MIFC_ORACLE_PREFIX, name)

#ifdef __cplusplus
extern "C" {
#endif

/* FieldCalculations.cc:1819 */
int ORC(vectorabs)(int nx, int ny, const float* u, const float* v, float* ff, int* fdefined, float undef);
/* FieldCalculations.cc:1843, :1875, :1910 */
int ORC(relvort)(int nx, int ny, const float* u, const float* v, const float* xmapr, const float* ymapr, float* out, int* fdefined, float undef);
int ORC(absvort)(int nx, int ny, const float* u, const float* v, const float* xmapr, const float* ymapr, const float* fcoriolis, float* out, int* fdefined,
                 float undef);
int ORC(divergence)(int nx, int ny, const float* u, const float* v, const float* xmapr, const float* ymapr, float* out, int* fdefined, float undef);
/* FieldCalculations.cc:1985 */
int ORC(gradient)(int nx, int ny, const float* field, const float* xmapr, const float* ymapr, int compute, float* out, int* fdefined, float undef);
/* FieldCalculations.cc:638, :674, :708 */
int ORC(plevelgwind_xcomp)(int nx, int ny, const float* z, const float* xmapr, const float* ymapr, const float* fcoriolis, float* ug, int* fdefined,
                           float undef);
int ORC(plevelgwind_ycomp)(int nx, int ny, const float* z, const float* xmapr, const float* ymapr, const float* fcoriolis, float* vg, int* fdefined,
                           float undef);
int ORC(plevelgvort)(int nx, int ny, const float* z, const float* xmapr, const float* ymapr, const float* fcoriolis, float* gvort, int* fdefined,
                     float undef);
/* FieldCalculations.cc:1511 */
int ORC(ilevelgwind)(int nx, int ny, const float* mpot, const float* xmapr, const float* ymapr, const float* fcoriolis, float* ug, float* vg,
                     int* fdefined, float undef);
/* FieldCalculations.cc:328, :1046, :1310 */
int ORC(pleveltemp)(int nx, int ny, const float* tinp, float p, const char* unit, int compute, float* tout, int* fdefined, float undef);
int ORC(hleveltemp)(int nx, int ny, const float* tinp, const float* ps, float alevel, float blevel, const char* unit, int compute, float* tout,
                    int* fdefined, float undef);
int ORC(aleveltemp)(int nx, int ny, const float* tinp, const float* p, const char* unit, int compute, float* tout, int* fdefined, float undef);
/* FieldCalculations.cc:400, :1145, :1394, :1738 */
int ORC(plevelhum)(int nx, int ny, const float* t, const float* huminp, float p, const char* unit, int compute, float* humout, int* fdefined,
                   float undef);
int ORC(hlevelhum)(int nx, int ny, const float* t, const float* huminp, const float* ps, float alevel, float blevel, const char* unit, int compute,
                   float* humout, int* fdefined, float undef);
int ORC(alevelhum)(int nx, int ny, const float* t, const float* huminp, const float* p, const char* unit, int compute, float* humout, int* fdefined,
                   float undef);
int ORC(cvhum)(int nx, int ny, const float* t, const float* huminp, const char* unit, int compute, float* humout, int* fdefined, float undef);

/* FieldCalculations.cc:1942 advection, :2424 jacobian, :2351/:2387 momentum coordinates, :2266 thermalFrontParameter */
int ORC(advection)(int nx, int ny, const float* f, const float* u, const float* v, const float* xmapr, const float* ymapr, float hours, float* advec,
                   int* fdefined, float undef);
int ORC(jacobian)(int nx, int ny, const float* field1, const float* field2, const float* xmapr, const float* ymapr, float* fjacobian, int* fdefined,
                  float undef);
int ORC(momentumXcoordinate)(int nx, int ny, const float* v, const float* xmapr, const float* fcoriolis, float fcoriolisMin, float* mxy, int* fdefined,
                             float undef);
int ORC(momentumYcoordinate)(int nx, int ny, const float* u, const float* ymapr, const float* fcoriolis, float fcoriolisMin, float* nxy, int* fdefined,
                             float undef);
int ORC(thermalFrontParameter)(int nx, int ny, const float* tx, const float* xmapr, const float* ymapr, float* tfp, int* fdefined, float undef);
/* FieldCalculations.cc:505 plevelqvector (three passes: geostrophic wind x, y, then the Q-vector component) */
int ORC(plevelqvector)(int nx, int ny, const float* z, const float* t, const float* xmapr, const float* ymapr, const float* fcoriolis, float p,
                       int compute, float* qcomp, int* fdefined, float undef);

/* identification string: "restatement" or "reference <version>" */
const char* ORC(kind)(void);

#ifdef __cplusplus
}
#endif

#endif /* MIFC_ORACLE_ABI_H */
