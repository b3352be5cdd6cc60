"""Drop-in for the reference's Python module ``mi_fieldcalc``
(python/py_mi_fieldcalc.cc:179-208): same function names, positional argument
order and behaviour -- every array is force-cast to C-contiguous float32
(:40), all arrays of a call must be 2-D with equal shapes or the call returns
``None`` (:82-83), the operator runs with ``fDefined = SOME_DEFINED`` (:89) and
a failing operator yields ``None`` (:92-93); the result is a new float32 array
of the input's shape.

The work is done by the HIP library through the C ABI (include/mifc.h) with
host pointers; there is no CPU path: without a gfx950 device the first call
raises.  One GPU context per calling thread (the reference releases the GIL,
:75, so concurrent callers are real).

Not built yet: vesselIcingModStall, vesselIcingMincog (the two iterative
models of FieldCalculationsVesselIcing.cc:182 and :629) -- they raise
NotImplementedError rather than compute elsewhere.
"""
import enum
import threading

import numpy as np

import mi_fieldcalc_amd as _fc


class ValuesDefined(enum.IntEnum):  # py_mi_fieldcalc.cc:182-185
    ALL_DEFINED = 0
    NONE_DEFINED = 1
    SOME_DEFINED = 2


_tls = threading.local()


def _ctx():
    c = getattr(_tls, "ctx", None)
    if c is None:
        c = _tls.ctx = _fc.Context(0)
    return c


def _wrap_2d(method, arrays, scalars, undef, lead=()):
    """py_wrap_2d (:79-96): arrays first, then the scalars, in the operator's order."""
    arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in arrays]
    if arrs[0].ndim != 2 or any(a.shape != arrs[0].shape for a in arrs[1:]):
        return None
    res = getattr(_ctx(), method)(*lead, *arrs, *scalars, fdefined=int(ValuesDefined.SOME_DEFINED), undef=float(undef))
    if res is None:
        return None
    return res[0]


def kIndex(t500, t700, rh700, t850, rh850, p500, p700, p850, compute, undef):
    return _wrap_2d("kIndex", [t500, t700, rh700, t850, rh850], [p500, p700, p850, compute], undef)


def ductingIndex(t850, rh850, p850, compute, undef):
    return _wrap_2d("ductingIndex", [t850, rh850], [p850, compute], undef)


def showalterIndex(t500, t850, rh850, p500, p850, compute, undef):
    return _wrap_2d("showalterIndex", [t500, t850, rh850], [p500, p850, compute], undef)


def boydenIndex(t700, z700, z1000, p700, p1000, compute, undef):
    return _wrap_2d("boydenIndex", [t700, z700, z1000], [p700, p1000, compute], undef)


def sweatIndex(t850, t500, td850, td500, u850, v850, u500, v500, undef):
    return _wrap_2d("sweatIndex", [t850, t500, td850, td500, u850, v850, u500, v500], [], undef)


def seaSoundSpeed(t, s, z, compute, undef):
    return _wrap_2d("seaSoundSpeed", [t, s], [z, compute], undef)


def cvtemp(tinp, compute, undef):
    return _wrap_2d("cvtemp", [tinp], [compute], undef)


def cvhum(t, huminp, unit, compute, undef):
    return _wrap_2d("cvhum", [t, huminp], [unit, compute], undef)


def abshum(t, rhum, undef):
    return _wrap_2d("abshum", [t, rhum], [], undef)


def windCooling(t, u, v, compute, undef):
    return _wrap_2d("windCooling", [t, u, v], [compute], undef)


def underCooledRain(precip, snow, tk, precipMin, snowRateMax, tcMax, undef):
    return _wrap_2d("underCooledRain", [precip, snow, tk], [precipMin, snowRateMax, tcMax], undef)


def vesselIcingOverland(airtemp, seatemp, u, v, sal, aice, undef):
    return _wrap_2d("vesselIcingOverland", [airtemp, seatemp, u, v, sal, aice], [], undef)


def vesselIcingMertins(airtemp, seatemp, u, v, sal, aice, undef):
    return _wrap_2d("vesselIcingMertins", [airtemp, seatemp, u, v, sal, aice], [], undef)


def vesselIcingModStall(sal, wave, x_wind, y_wind, airtemp, rh, sst, p, Pw, aice, depth, vs, alpha, zmin, zmax, undef):
    raise NotImplementedError("vesselIcingModStall is not built on the GPU yet (and there is no CPU path)")


def vesselIcingMincog(sal, wave, x_wind, y_wind, airtemp, rh, sst, p, Pw, aice, depth, vs, alpha, zmin, zmax, alt, undef):
    raise NotImplementedError("vesselIcingMincog is not built on the GPU yet (and there is no CPU path)")
