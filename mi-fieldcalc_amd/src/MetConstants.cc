// Host-side helpers of mi_fieldcalc/MetConstants.h that are not operators: the
// ICAO standard atmosphere (doc 7488) between pressure and geopotential altitude,
// and flight-level rounding.  Written from the published layer model; the layer
// constants are those the reference uses (MetConstants.cc:48-80), including its
// sea-level pressure of 1013.15 hPa, so that both libraries return the same numbers.
#include "mi_fieldcalc/MetConstants.h"

#include <cmath>

namespace miutil {
namespace constants {

namespace {

const double G0 = 9.80665;    // m / s^2
const double RAIR = 287.05287; // J / (kg K)
const int NLAYER = 7;
// base geopotential height (m) and temperature gradient (K/m) of each layer
const double BASE_H[NLAYER + 1] = {0, 11000, 20000, 32000, 47000, 51000, 71000, 84852};
const double LAPSE[NLAYER] = {-6.5e-3, 0, 1.0e-3, 2.8e-3, 0, -2.8e-3, -2.0e-3};

struct Atmosphere
{
  double t[NLAYER + 1]; // temperature at the layer bases (K)
  double p[NLAYER + 1]; // pressure at the layer bases (hPa)
  Atmosphere()
  {
    t[0] = 288.15;
    p[0] = 1013.15;
    for (int k = 0; k < NLAYER; ++k) {
      const double dh = BASE_H[k + 1] - BASE_H[k];
      t[k + 1] = t[k] + dh * LAPSE[k];
      p[k + 1] = p[k] * factor(k, dh);
    }
  }
  // p(base + dh) / p(base) inside layer k
  double factor(int k, double dh) const
  {
    if (LAPSE[k] != 0)
      return std::pow(1 + dh * LAPSE[k] / t[k], -G0 / (LAPSE[k] * RAIR));
    return std::exp(-dh * G0 / (RAIR * t[k]));
  }
};

const Atmosphere& atmosphere()
{
  static const Atmosphere a;
  return a;
}

} // namespace

double ICAO_geo_altitude_from_pressure(double pressure)
{
  const Atmosphere& a = atmosphere();
  int k = 0;
  while (k < NLAYER && pressure < a.p[k + 1])
    ++k;
  if (k >= NLAYER)
    return BASE_H[NLAYER] + 1000; // above the model: one kilometre beyond its top
  const double ratio = pressure / a.p[k];
  if (LAPSE[k] != 0)
    return BASE_H[k] + (a.t[k] / LAPSE[k]) * (std::pow(ratio, -(LAPSE[k] * RAIR) / G0) - 1);
  return BASE_H[k] - std::log(ratio) * (RAIR * a.t[k]) / G0;
}

double ICAO_pressure_from_geo_altitude(double altitude)
{
  const Atmosphere& a = atmosphere();
  int k = 0;
  while (k < NLAYER && altitude > BASE_H[k + 1])
    ++k;
  if (k >= NLAYER)
    return a.p[NLAYER] - 1; // above the model
  return a.p[k] * a.factor(k, altitude - BASE_H[k]);
}

int FL_from_geo_altitude(double a)
{
  return 5 * (int)round(a * ft_per_m / 500);
}

double geo_altitude_from_FL(double fl)
{
  return fl * 100 / ft_per_m;
}

} // namespace constants
} // namespace miutil
