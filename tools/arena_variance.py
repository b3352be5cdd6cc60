#!/usr/bin/env python3
"""Follow-up of tools/alloc_variance.py: the four arrays of the headline batch carved out of ONE allocation.
If the allocation is physically contiguous (or at least mapped in large fragments), the RELATIVE physical
placement of the arrays is then under our control: vary the gap between them and re-allocate a few times.
Usage (GPU box): python tools/arena_variance.py [reallocations]
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
GAPS_KIB = [0, 4, 64, 260, 1028, 2052, 33 * 1024 + 4, 256 * 1024 + 260]


def timed(fn, rounds=5, inner=5):
    fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(inner):
            fn()
        e.record()
        torch.cuda.synchronize()
        out.append(s.elapsed_time(e) / inner)
    return float(np.median(out))


def main():
    reallocs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda", 0)
    ctx = fc.Context(0)
    xm, ym, _ = synth.grid_maps(NX, NY)
    dxm, dym = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
    flags = np.full(NLEV, fc.ALL_DEFINED, np.int32)
    n = NX * NY * NLEV
    su, sv = synth.device_wind(NX, NY, NLEV, 1234, dev)
    hu, hv = su.cpu(), sv.cpu()
    del su, sv
    print("%7s " % "realloc" + " ".join("%9s" % ("gap %dK" % g) for g in GAPS_KIB) + "   (levelwalk kernel ms; arrays at k*(array bytes rounded to 2 MiB + gap))")
    for r in range(reallocs):
        torch.cuda.empty_cache()
        span = ((n * 4 + (2 << 20) - 1) >> 21 << 21)
        total = 4 * (span + max(GAPS_KIB) * 1024)
        arena = torch.empty(total // 4, dtype=torch.float32, device=dev)
        row = []
        for g in GAPS_KIB:
            step = (span + g * 1024) // 4
            views = [arena[k * step:k * step + n].view(NLEV, NY, NX) for k in range(4)]
            views[0].copy_(hu)
            views[1].copy_(hv)
            du, dv, rv, dg = views
            row.append(timed(lambda: ctx.vortdiv_levels_enqueue(du, dv, dxm, dym, rv, dg, fdefined=flags)))
        print("%7d " % r + " ".join("%9.4f" % t for t in row), flush=True)
        del arena, views, du, dv, rv, dg


if __name__ == "__main__":
    main()
