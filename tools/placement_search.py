#!/usr/bin/env python3
"""Search over COMBINATIONS of arrays: a pool of M arrays of the headline batch's size, the operator timed on
(u, v, rvort, diverg) drawn from the pool.  (1) the M/4 disjoint batches in allocation order (what
choose_placement sees), (2) random combinations, (3) coordinate descent from the best one: replace one of the
four arrays at a time by every other array of the pool, keep what is faster.
Usage (GPU box): python tools/placement_search.py [M] [random combinations]
"""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402

NX, NY, NLEV = 1440, 720, 137


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    nrand = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    dev = torch.device("cuda", 0)
    ctx = fc.Context(0)
    xm, ym, _ = synth.grid_maps(NX, NY)
    dxm, dym = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
    flags = np.full(NLEV, fc.ALL_DEFINED, np.int32)
    pool = [ctx.batch_empty(NLEV, NY, NX) for _ in range(m)]
    nprobe = [0]

    def probe(c):  # c = (u, v, rv, dg) indices; values do not matter for the time
        a, b, r, d = (pool[i] for i in c)
        ms = []
        for k in range(4):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(4):
                ctx.vortdiv_levels_enqueue(a, b, dxm, dym, r, d, fdefined=flags)
            e.record()
            torch.cuda.synchronize()
            if k:
                ms.append(s.elapsed_time(e) / 4)
        nprobe[0] += 1
        return float(np.median(ms))

    for _ in range(5):
        probe((0, 1, 2, 3))
    batches = [(4 * i, 4 * i + 1, 4 * i + 2, 4 * i + 3) for i in range(m // 4)]
    tb = [probe(c) for c in batches]
    print("disjoint batches in allocation order: " + " ".join("%.4f" % t for t in tb) + "  -> best %.4f" % min(tb))
    rng = random.Random(5)
    combos = [tuple(rng.sample(range(m), 4)) for _ in range(nrand)]
    tr = [probe(c) for c in combos]
    print("%d random combinations: min %.4f median %.4f max %.4f" % (nrand, min(tr), float(np.median(tr)), max(tr)))
    allc = batches + combos
    allt = tb + tr
    best = list(allc[int(np.argmin(allt))])
    tbest = min(allt)
    for sweep in range(2):
        for pos in (2, 3, 0, 1):
            cand = [i for i in range(m) if i not in best]
            ts = []
            for i in cand:
                c = list(best)
                c[pos] = i
                ts.append(probe(tuple(c)))
            j = int(np.argmin(ts))
            print("  sweep %d position %d: alternatives min %.4f median %.4f max %.4f (current %.4f)" % (sweep, pos, min(ts), float(np.median(ts)), max(ts), tbest))
            if ts[j] < tbest:
                best[pos] = cand[j]
                tbest = ts[j]
    print("coordinate descent: %s %.4f after %d probes; re-probed %.4f %.4f" % (best, tbest, nprobe[0], probe(tuple(best)), probe(tuple(best))))


if __name__ == "__main__":
    main()
