#!/usr/bin/env python3
"""Bit-level A/B of two builds of the library on the humidity / index operators: run once per build
(MIFC_LIB_PATH selects it) with `dump FILE`, then `compare FILE_A FILE_B`.

    MIFC_LIB_PATH=tools/ab/libmifc_base.so python tools/ab_bits.py dump /tmp/a.npz
    python tools/ab_bits.py dump /tmp/b.npz && python tools/ab_bits.py compare /tmp/a.npz /tmp/b.npz

Inputs: seeded 1440 x 720 fields over wide ranges (temperatures 150 .. 400 K so that the table's ends and the
not-covered cells are in, humidities 0 .. 120 %, a sprinkle of undefined values, NaN, zeros and infinities)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def inputs():
    rng = np.random.default_rng(20260303)
    nx, ny = 1440, 720
    t = rng.uniform(150.0, 400.0, (ny, nx)).astype(np.float32)
    t2 = (t + rng.uniform(-40.0, 10.0, (ny, nx))).astype(np.float32)
    t3 = (t + rng.uniform(-20.0, 20.0, (ny, nx))).astype(np.float32)
    rh = rng.uniform(-5.0, 120.0, (ny, nx)).astype(np.float32)
    rh2 = rng.uniform(0.0, 100.0, (ny, nx)).astype(np.float32)
    q = rng.uniform(0.0, 0.03, (ny, nx)).astype(np.float32)
    ps = rng.uniform(500.0, 1080.0, (ny, nx)).astype(np.float32)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e35, 1e-40, -1e-40, 3.4e38, 1e-30], np.float32)
    for a in (t, t2, rh, q):
        idx = rng.integers(0, a.size, 4000)
        a.reshape(-1)[idx] = special[rng.integers(0, len(special), 4000)]
    return nx, ny, t, t2, t3, rh, rh2, q, ps


def dump(path):
    import mi_fieldcalc_amd as fc

    ctx = fc.Context(0)
    nx, ny, t, t2, t3, rh, rh2, q, ps = inputs()
    out = {}
    for flag, tag in ((fc.ALL_DEFINED, "all"), (fc.SOME_DEFINED, "some")):
        for compute in (1, 2):
            r = ctx.showalterIndex(t2, t, rh, 500.0, 850.0, compute, fdefined=flag)
            out["showalter_%d_%s" % (compute, tag)] = r[0]
            r = ctx.kIndex(t2, t3, rh2, t, rh, 500.0, 700.0, 850.0, compute, fdefined=flag)
            out["kindex_%d_%s" % (compute, tag)] = r[0]
            r = ctx.ductingIndex(t, rh, 850.0, compute, fdefined=flag)
            out["ducting_%d_%s" % (compute, tag)] = r[0]
        for compute in range(1, 13):
            r = ctx.hlevelhum(t, q if compute in (1, 3, 5, 7, 9, 11) else rh2, ps, 10.0, 0.9, "1", compute, fdefined=flag)
            if r is not None:
                out["hlevelhum_%d_%s" % (compute, tag)] = r[0]
        for compute in range(1, 13):
            for pres in (850.0, 1e-30, 0.0):
                r = ctx.plevelhum(t, q if compute in (1, 3, 5, 7, 9, 11) else rh2, pres, "1", compute, fdefined=flag)
                if r is not None:
                    out["plevelhum_%d_%g_%s" % (compute, pres, tag)] = r[0]
        for compute in range(1, 6):
            r = ctx.hleveltemp(t, ps, 10.0, 0.9, "1", compute, fdefined=flag)
            if r is not None:
                out["hleveltemp_%d_%s" % (compute, tag)] = r[0]
    np.savez(path, **{k: np.asarray(v) for k, v in out.items()})
    print("wrote %d arrays to %s (library: %s)" % (len(out), path, os.environ.get("MIFC_LIB_PATH", "default")))


def compare(a, b):
    A, B = np.load(a), np.load(b)
    worst = 0
    for k in A.files:
        x, y = A[k].view(np.uint32), B[k].view(np.uint32)
        nanboth = np.isnan(A[k]) & np.isnan(B[k])
        diff = int(np.count_nonzero((x != y) & ~nanboth))
        worst = max(worst, diff)
        print("%-24s %s" % (k, "identical" if diff == 0 else "%d cells differ" % diff))
    sys.exit(1 if worst else 0)


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2])
    else:
        compare(sys.argv[2], sys.argv[3])
