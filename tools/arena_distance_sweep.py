#!/usr/bin/env python3
"""The four arrays of the headline batch carved out of ONE big allocation at a chosen DISTANCE from each other:
u at 0, v at D, rvort at 2 D, diverg at 3 D, D = array size rounded to 2 MiB + gap, gap = 0 ... in steps of
STEP MiB.  Separately allocated arrays sit 544 MiB apart (the size rounded up) -- the worst case of the
placement experiments; is there a distance that is reliably good?
Usage (GPU box): python tools/arena_distance_sweep.py [step MiB] [count] [arena GiB]
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
    step = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    dev = torch.device("cuda", 0)
    ctx = fc.Context(0)
    xm, ym, _ = synth.grid_maps(NX, NY)
    dxm, dym = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
    flags = np.full(NLEV, fc.ALL_DEFINED, np.int32)
    n = NX * NY * NLEV
    span = ((n * 4 + (2 << 20) - 1) >> 21) << 21
    dmax = span + step * (count - 1) * (1 << 20)
    total = 3 * dmax + span
    arena = torch.empty(total // 4, dtype=torch.float32, device=dev)
    print("arena %.2f GiB at %#x; array %.1f MiB (span %d MiB)" % (total / 2**30, arena.data_ptr(), n * 4 / 2**20, span >> 20))

    def views(d_bytes):
        st = d_bytes // 4
        return [arena[k * st:k * st + n].view(NLEV, NY, NX) for k in range(4)]

    def probe(v):
        ms = []
        for k in range(4):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(4):
                ctx.vortdiv_levels_enqueue(v[0], v[1], dxm, dym, v[2], v[3], fdefined=flags)
            e.record()
            torch.cuda.synchronize()
            if k:
                ms.append(s.elapsed_time(e) / 4)
        return float(np.median(ms))

    for _ in range(5):
        probe(views(span))
    print("gap MiB -> kernel ms (two passes)")
    res = []
    for rep in range(2):
        row = []
        for i in range(count):
            row.append(probe(views(span + i * step * (1 << 20))))
        res.append(row)
    for i in range(count):
        print("%6d  %.4f  %.4f" % (i * step, res[0][i], res[1][i]))
    a = np.array(res[1])
    print("min %.4f at gap %d MiB; median %.4f; max %.4f at gap %d MiB" % (a.min(), int(np.argmin(a)) * step, float(np.median(a)), a.max(), int(np.argmax(a)) * step))


if __name__ == "__main__":
    main()
