#!/usr/bin/env python3
"""Pool of M separately allocated arrays (allocation order = index); the operator on (u, v, rvort, diverg) =
pool[b], pool[b + s], pool[b + 2 s], pool[b + 3 s] for spacings s = 1 .. and a few bases b: are arrays that were
allocated next to each other bad company, and is any fixed spacing reliably good?
Usage (GPU box): python tools/placement_spacing.py [M]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402

NX, NY, NLEV = 1440, 720, 137


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda", 0)
    ctx = fc.Context(0)
    xm, ym, _ = synth.grid_maps(NX, NY)
    dxm, dym = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
    flags = np.full(NLEV, fc.ALL_DEFINED, np.int32)
    pool = [ctx.batch_empty(NLEV, NY, NX) for _ in range(m)]
    lo = min(a.data_ptr() for a in pool)
    print("virtual offsets (MiB): " + " ".join("%d" % ((a.data_ptr() - lo) >> 20) for a in pool))

    def probe(c):
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
        return float(np.median(ms))

    for _ in range(5):
        probe((0, 1, 2, 3))
    print("spacing s: kernel ms for bases b = 0, 1, 2, ... (as many as fit, at most 8)")
    for s in range(1, (m - 1) // 3 + 1):
        bases = [b for b in range(0, m - 3 * s)][:8]
        ts = [probe((b, b + s, b + 2 * s, b + 3 * s)) for b in bases]
        print("  s=%2d  %s   median %.4f" % (s, " ".join("%.4f" % t for t in ts), float(np.median(ts))))
    # only ONE pair adjacent: which adjacency hurts?
    far = (0, 10, 20, 30) if m > 30 else (0, m // 4, m // 2, 3 * m // 4)
    print("far apart %s: %.4f" % (far, probe(far)))
    for name, c in (("u,v adjacent", (0, 1, 20, 30)), ("rv,dg adjacent", (0, 10, 20, 21)), ("u,rv adjacent", (0, 10, 1, 30)), ("v,dg adjacent", (0, 10, 20, 11)),
                    ("u,v and rv,dg adjacent", (0, 1, 20, 21)), ("u,rv and v,dg adjacent", (0, 10, 1, 11))):
        if max(c) < m:
            print("  %-26s %s: %.4f" % (name, c, probe(c)))


if __name__ == "__main__":
    main()
