// Meteorological constants and the saturation-vapour-pressure table used by
// the humidity / theta operators.  Names, types and values are those of the
// reference's src/mi_fieldcalc/MetConstants.h:39-84 so that callers keep
// compiling; the GPU kernels hold their own copy (csrc/mifc_device.h).
#ifndef MI_FIELDCALC_METCONSTANTS_H
#define MI_FIELDCALC_METCONSTANTS_H

#include <string>

namespace miutil {
namespace constants {

const float r = 287., cp = 1004., p0 = 1000., t0 = 273.15;
const float eps = 0.622;
const float xlh = 2.501e+6;
const float rcp = r / cp, cplr = xlh / rcp, exl = eps * xlh;
const float p0inv = 1. / p0;
const float kappa = r / cp;
const float g = 9.8;
const float ginv = 1. / g;
const float rhmin = 0.02, rhmax = 1.00;

const double ft_per_m = 3.2808399;
const double ms2knots = 3600.0 / 1852.0, knots2ms = 1 / ms2knots;

// e_sat over water at -100, -95, ... +100 degrees Celsius
const int N_EWT = 41;
extern const float ewt[N_EWT];

// Linear interpolation in ewt[] and its inverse (host-side helper; the
// operators evaluate the same arithmetic on the GPU).
class ewt_calculator
{
public:
  ewt_calculator(float t_celsius);
  bool defined() const { return l >= 0 && l < N_EWT - 1; }
  bool defined(bool& allDefined, float undef, float& out) const
  {
    if (defined())
      return true;
    allDefined = false;
    out = undef;
    return false;
  }
  float value() const { return ewt[l] + (ewt[l + 1] - ewt[l]) * (x - l); }
  float inverse(float et) const;

private:
  float x;
  int l;
};

// standard pressure levels (hPa) and the flight levels (100 ft) drawn for them
// (reference MetConstants.h:86-92; pressure2FlightLevel interpolates in these)
const int nLevelTable = 16;
const float pLevelTable[nLevelTable] = {1000, 925, 850, 800, 700, 500, 400, 300, 250, 200, 150, 100, 70, 50, 30, 10};
const float fLevelTable[nLevelTable] = {5, 25, 50, 65, 100, 185, 235, 300, 340, 385, 445, 530, 605, 675, 780, 1020};
const float fLevelTable_old[nLevelTable] = {0, 25, 50, 70, 100, 180, 240, 300, 340, 390, 450, 530, 600, 700, 800, 999};

// ICAO standard atmosphere (doc 7488), pressure in hPa, geopotential altitude in m
double ICAO_geo_altitude_from_pressure(double pressure);
double ICAO_pressure_from_geo_altitude(double altitude);
// altitude (m) -> flight level rounded to 500 ft, and back (no rounding)
int FL_from_geo_altitude(double a);
double geo_altitude_from_FL(double fl);

const std::string VerticalName[7] = {"none", "pressure", "hybrid", "atmospheric", "isentropic", "oceandepth", "other"};

} // namespace constants

inline float ms2knots(float ff) { return ff * miutil::constants::ms2knots; }
inline float knots2ms(float ff) { return ff * miutil::constants::knots2ms; }

} // namespace miutil

#endif // MI_FIELDCALC_METCONSTANTS_H
