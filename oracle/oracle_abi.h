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

/* ---- the rest of the pointwise catalogue (FieldCalculations.cc line of each in the comment) ---- */
/* :369, :1100, :1355 */
int ORC(plevelthe)(int nx, int ny, const float* t, const float* rh, float p, int compute, float* the, int* fdefined, float undef);
int ORC(hlevelthe)(int nx, int ny, const float* t, const float* q, const float* ps, float alevel, float blevel, int compute, float* the, int* fdefined,
                   float undef);
int ORC(alevelthe)(int nx, int ny, const float* t, const float* q, const float* p, int compute, float* the, int* fdefined, float undef);
/* :597, :1219, :1460, :1276, :466 */
int ORC(plevelducting)(int nx, int ny, const float* t, const float* h, float p, int compute, float* duct, int* fdefined, float undef);
int ORC(hlevelducting)(int nx, int ny, const float* t, const float* h, const float* ps, float alevel, float blevel, int compute, float* duct,
                       int* fdefined, float undef);
int ORC(alevelducting)(int nx, int ny, const float* t, const float* h, const float* p, int compute, float* duct, int* fdefined, float undef);
int ORC(hlevelpressure)(int nx, int ny, const float* ps, float alevel, float blevel, float* p, int* fdefined, float undef);
int ORC(pleveldz2tmean)(int nx, int ny, const float* z1, const float* z2, float p1, float p2, int compute, float* tmean, int* fdefined, float undef);
/* :745, :816, :872, :973, :1016 */
int ORC(kIndex)(int nx, int ny, const float* t500, const float* t700, const float* rh700, const float* t850, const float* rh850, float p500, float p700,
                float p850, int compute, float* kfield, int* fdefined, float undef);
int ORC(ductingIndex)(int nx, int ny, const float* t850, const float* rh850, float p850, int compute, float* duct, int* fdefined, float undef);
int ORC(showalterIndex)(int nx, int ny, const float* t500, const float* t850, const float* rh850, float p500, float p850, int compute, float* sfield,
                        int* fdefined, float undef);
int ORC(boydenIndex)(int nx, int ny, const float* t700, const float* z700, const float* z1000, float p700, float p1000, int compute, float* bfield,
                     int* fdefined, float undef);
int ORC(sweatIndex)(int nx, int ny, const float* t850, const float* t500, const float* td850, const float* td500, const float* u850, const float* v850,
                    const float* u500, const float* v500, float* sindex, int* fdefined, float undef);
/* :1555, :1608, :1676, :2181, :2231, :2311, :3063, :2462 */
int ORC(seaSoundSpeed)(int nx, int ny, const float* t, const float* s, float z, int compute, float* soundspeed, int* fdefined, float undef);
int ORC(cvtemp)(int nx, int ny, const float* tinp, int compute, float* tout, int* fdefined, float undef);
int ORC(abshum)(int nx, int ny, const float* t, const float* rhum, float* abshumout, int* fdefined, float undef);
int ORC(windCooling)(int nx, int ny, const float* t, const float* u, const float* v, int compute, float* dtcool, int* fdefined, float undef);
int ORC(underCooledRain)(int nx, int ny, const float* precip, const float* snow, const float* tk, float precipMin, float snowRateMax, float tcMax,
                         float* undercooled, int* fdefined, float undef);
int ORC(pressure2FlightLevel)(int nx, int ny, const float* pressure, float* flightlevel, int* fdefined, float undef);
int ORC(snow_in_cm)(int nx, int ny, const float* snow_water, const float* tk2m, const float* td2m, float* snow_cm, int* fdefined, float undef);
/* the reference takes std::vector<float> values; here pointer + length */
int ORC(values2classes)(int nx, int ny, const float* fvalue, float* fclass, const float* values, int nvalues, int* fdefined, float undef);
/* FieldCalculationsVesselIcing.cc:77 and :114 */
int ORC(vesselIcingOverland)(int nx, int ny, const float* airtemp, const float* seatemp, const float* u, const float* v, const float* sal,
                             const float* aice, float* icing, int* fdefined, float undef);
int ORC(vesselIcingMertins)(int nx, int ny, const float* airtemp, const float* seatemp, const float* u, const float* v, const float* sal,
                            const float* aice, float* icing, int* fdefined, float undef);
/* FieldCalculations.cc:2076 (the reference's `field` is non-const only because it may alias fsmooth) */
int ORC(shapiro2_filter)(int nx, int ny, const float* field, float* fsmooth, int* fdefined, float undef);
/* field algebra :2501-2669 (the reference's void functions return 1 here) */
int ORC(minvalueFields)(int nx, int ny, const float* field1, const float* field2, float* fres, int* fdefined, float undef);
int ORC(maxvalueFields)(int nx, int ny, const float* field1, const float* field2, float* fres, int* fdefined, float undef);
int ORC(minvalueFieldConst)(int nx, int ny, const float* field1, float value, float* fres, int* fdefined, float undef);
int ORC(maxvalueFieldConst)(int nx, int ny, const float* field1, float value, float* fres, int* fdefined, float undef);
int ORC(absvalueField)(int nx, int ny, const float* field, float* fres, int* fdefined, float undef);
int ORC(log10Field)(int nx, int ny, const float* field, float* fres, int* fdefined, float undef);
int ORC(pow10Field)(int nx, int ny, const float* field, float* fres, int* fdefined, float undef);
int ORC(logField)(int nx, int ny, const float* field, float* fres, int* fdefined, float undef);
int ORC(expField)(int nx, int ny, const float* field, float* fres, int* fdefined, float undef);
int ORC(powerField)(int nx, int ny, const float* field, float value, float* fres, int* fdefined, float undef);
int ORC(replaceUndefined)(int nx, int ny, const float* field, float value, float* fres, int* fdefined, float undef);
int ORC(replaceDefined)(int nx, int ny, const float* field, float value, float* fres, int* fdefined, float undef);
int ORC(fieldOPERfield)(int compute, int nx, int ny, const float* field1, const float* field2, float* fres, int* fdefined, float undef);
int ORC(fieldOPERconstant)(int compute, int nx, int ny, const float* field, float value, float* fres, int* fdefined, float undef);
int ORC(constantOPERfield)(int compute, int nx, int ny, float value, const float* field, float* fres, int* fdefined, float undef);

/* ---- ensemble reductions :2671-2860 (std::vector<float*> -> pointer table + length,
 *      std::vector<ValuesDefined> -> int array of the same length) ---- */
int ORC(sumFields)(int nx, int ny, const float* const* fields, int nfields, float* fres, int* fdefined, float undef);
int ORC(meanValue)(int nx, int ny, const float* const* fields, const int* fdefined_in, int nfields, float* fres, int* fdefined_out, float undef);
int ORC(stddevValue)(int nx, int ny, const float* const* fields, const int* fdefined_in, int nfields, float* fres, int* fdefined_out, float undef);
int ORC(extremeValue)(int compute, int nx, int ny, const float* const* fields, int nfields, float* fres, int* fdefined, float undef);
int ORC(probability)(int compute, int nx, int ny, const float* const* fields, const int* fdefined_in, int nfields, const float* limits, int nlimits,
                     float* fres, int* fdefined_out, float undef);

/* identification string: "restatement" or "reference <version>" */
const char* ORC(kind)(void);

#ifdef __cplusplus
}
#endif

#endif /* MIFC_ORACLE_ABI_H */
