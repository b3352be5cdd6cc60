#!/usr/bin/env python3
"""Generates (and verifies) the lookup tables of csrc/mifc_device.h::pow_kappa.

x^kappa for float x > 0, evaluated in double:
  x = m * 2^e with m in [sqrt(1/2), sqrt(2))  (integer ops on the float bits)
  i = top 4 bits of (bits(m) - bits(sqrt(1/2)))          -> 16 sub-intervals
  r = m * invc[i] - 1, |r| <= 1/32;  log2 m = logc[i] + log2(1 + r)
  log2(1+r) = r/ln2 * (1 - r/2 + r^2/3 - r^3/4 + r^4/5 - r^5/6)
  t = kappa * (e + log2 m);  k = rint(32 t);  g = (t - k/32) ln 2, |g| <= 0.0109
  2^t = 2^(k>>5) * T[k&31] * (1 + g + g^2/2 + g^3/6 + g^4/24),  T[j] = 2^(j/32)
Prints the tables as C hex-float initialisers and checks the float result
against the correctly rounded power on random arguments.
"""
import math

import numpy as np

OFF = 0x3F3504F3  # bits of sqrt(1/2) as float
kappa32 = np.float32(287.0) / np.float32(1004.0)
K = np.float64(kappa32)
LN2 = 0.6931471805599453
INVLN2 = 1.4426950408889634


def tables():
    invc, logc = [], []
    for i in range(16):
        lo = OFF + (i << 19)
        mid = lo + (1 << 18)
        c = float(np.array([mid], dtype=np.int32).view(np.float32)[0])
        ic = 1.0 / c
        invc.append(ic)
        logc.append(-math.log2(ic))
    T = [2.0 ** (j / 32.0) for j in range(32)]
    return np.array(invc), np.array(logc), np.array(T)


def pow_kappa(x, invc, logc, T):
    x = np.asarray(x, dtype=np.float32)
    ix = x.view(np.int32)
    e = (ix - np.int32(OFF)) >> 23
    im = ix - (e << 23)
    i = (im - np.int32(OFF)) >> 19
    m = im.astype(np.int32).view(np.float32).astype(np.float64)
    r = m * invc[i] - 1.0
    p = -1.0 / 6
    for c in (1.0 / 5, -1.0 / 4, 1.0 / 3, -0.5, 1.0):
        p = p * r + c
    log2m = logc[i] + r * INVLN2 * p
    t = K * (e.astype(np.float64) + log2m)
    k = np.rint(t * 32.0)
    g = (t - k * (1.0 / 32.0)) * LN2
    ki = k.astype(np.int64)
    q = 1.0 / 24
    for c in (1.0 / 6, 0.5, 1.0, 1.0):
        q = q * g + c
    return np.ldexp(T[ki & 31] * q, ki >> 5).astype(np.float32)


if __name__ == "__main__":
    invc, logc, T = tables()
    print("// log2 table: {1/c_i, log2(c_i)}")
    for a, b in zip(invc, logc):
        print("  {%s, %s}," % (float(a).hex(), float(b).hex()))
    print("// exp2 table: 2^(j/32)")
    for j in range(0, 32, 4):
        print("  " + ", ".join(float(v).hex() for v in T[j:j + 4]) + ",")
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(1e-3, 1.2, 2_000_000), rng.uniform(0.1, 1.1, 2_000_000),
                        np.exp(rng.uniform(np.log(1e-25), np.log(1e25), 500_000))]).astype(np.float32)
    got = pow_kappa(x, invc, logc, T)
    ref = np.power(x.astype(np.float64), K).astype(np.float32)
    d = np.abs(got.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))
    print("// check: mismatches vs correctly rounded: %d of %d, max ulp %d" % (np.count_nonzero(d), len(x), d.max()))
