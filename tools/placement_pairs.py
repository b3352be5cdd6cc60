#!/usr/bin/env python3
"""Which PAIRS of arrays stream well together?  M arrays of the headline batch's size; (1) the time of filling
two of them at once (linear one-shot fill of both, the write side of the operator) for every pair, (2) of reading
two at once, (3) the operator with a fixed input pair and every output pair of a subset.  Virtual addresses are
printed: the pattern, if any, is in the physical pages.
Usage (GPU box): python tools/placement_pairs.py [M]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIFC_LIB_PATH", os.path.join(ROOT, "mi-fieldcalc_amd", "libmifc_measure.so"))  # mifc_timing_* / yardsticks: measurement build

import torch  # noqa: E402

import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402

NX, NY, NLEV = 1440, 720, 137


def timed(fn, reps=3, inner=4):
    fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(inner):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms.append(s.elapsed_time(e) / inner)
    return float(np.median(ms))


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    dev = torch.device("cuda", 0)
    ctx = fc.Context(0)
    xm, ym, _ = synth.grid_maps(NX, NY)
    dxm, dym = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
    flags = np.full(NLEV, fc.ALL_DEFINED, np.int32)
    su, sv = synth.device_wind(NX, NY, NLEV, 1234, dev)
    arrays = [ctx.batch_empty(NLEV, NY, NX) for _ in range(m)]
    for _ in range(30):
        ctx.bench_stream2(5, 0, arrays[0], arrays[1], arrays[0], arrays[1])
    torch.cuda.synchronize()
    lo = min(a.data_ptr() for a in arrays)
    print("virtual offsets (MiB): " + " ".join("%.1f" % ((a.data_ptr() - lo) / 2**20) for a in arrays))
    w = np.zeros((m, m))
    for i in range(m):
        for j in range(i + 1, m):
            w[i, j] = w[j, i] = timed(lambda: ctx.bench_stream2(5, 0, arrays[i], arrays[j], arrays[i], arrays[j]))
    print("fill two arrays at once, ms (568 MB each):")
    for i in range(m):
        print("  " + " ".join("%.4f" % w[i, j] if i != j else "  --  " for j in range(m)))
    off = w[~np.eye(m, dtype=bool)]
    print("  min %.4f median %.4f max %.4f" % (off.min(), float(np.median(off)), off.max()))
    # the operator: inputs = arrays 0, 1; every output pair among the others
    arrays[0].copy_(su)
    arrays[1].copy_(sv)
    print("operator with inputs (0, 1) and outputs (i, j): kernel ms next to the pair's fill time")
    rows = []
    for i in range(2, m):
        for j in range(i + 1, m):
            k = timed(lambda: ctx.vortdiv_levels_enqueue(arrays[0], arrays[1], dxm, dym, arrays[i], arrays[j], fdefined=flags))
            rows.append((k, w[i, j], i, j))
    for k, f, i, j in sorted(rows):
        print("  out (%2d,%2d) kernel %.4f fill %.4f" % (i, j, k, f))
    ks, fs = np.array([r[0] for r in rows]), np.array([r[1] for r in rows])
    print("correlation(kernel, fill of its output pair) = %.3f" % float(np.corrcoef(ks, fs)[0, 1]))


if __name__ == "__main__":
    main()
