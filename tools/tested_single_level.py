#!/usr/bin/env python3
"""One 4000 x 4000 level (BASELINE.json config 4's field), fused vorticity + divergence: what do the undefined tests and the
undefined COUNTS cost on a single huge level?  ALL_DEFINED, SOME_DEFINED on clean data (tests, no counts to add), SOME_DEFINED
with 1 % / 10 % of the cells undefined (every workgroup has a count to hand over).  Three rotating buffer sets (cold),
ms per launch from bursts of 10.
    python tools/tested_single_level.py [nx,ny]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402


def main():
    nx, ny = (int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "4000,4000").split(","))
    nlev = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dev = torch.device("cuda", 0)
    ctx = fc.Context(0)
    ctx.use_torch_stream()
    xm, ym, fcor = synth.grid_maps(nx, ny, h=2500.0)
    dxm, dym = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
    sets = []
    for k in range(3):
        u, v = synth.device_wind(nx, ny, nlev, 70 + k, dev)
        sets.append([u, v, torch.empty_like(u), torch.empty_like(u)])
    cnt = torch.zeros(nlev, dtype=torch.int64, device=dev)
    state = {"k": 0}

    def run(flag):
        u, v, rv, dg = sets[state["k"] % 3]
        state["k"] += 1
        flags = np.full(nlev, flag, np.int32)
        assert ctx.vortdiv_levels_enqueue(u, v, dxm, dym, rv, dg, fdefined=flags, n_undefined=cnt if flag != fc.ALL_DEFINED else None)

    def timed(flag):
        for _ in range(6):
            run(flag)
        torch.cuda.synchronize()
        ms = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run(flag)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1) / 10)
        return float(np.median(ms))

    alg = nx * ny * (16.0 * nlev + 8.0)
    print("%dx%d, %d level(s), fused vorticity + divergence, ms per launch (cold, bursts of 10); form: see last column" % (nx, ny, nlev))
    rows = [("ALL_DEFINED", fc.ALL_DEFINED, 0.0), ("SOME_DEFINED, clean data", fc.SOME_DEFINED, 0.0), ("SOME_DEFINED, 1 % undefined", fc.SOME_DEFINED, 0.01),
            ("SOME_DEFINED, 10 % undefined", fc.SOME_DEFINED, 0.10)]
    for name, flag, frac in rows:
        if frac > 0:
            g = torch.Generator(device=dev)
            g.manual_seed(5)
            for s in sets:
                mask = torch.rand(s[0].shape, generator=g, device=dev) < frac
                s[0][mask] = float(fc.UNDEF)
        t = timed(flag)
        print("%-32s %8.4f ms  %5.1f %% of 8 TB/s   %s" % (name, t, alg / t / 1e6 / 80.0, ctx.last_stencil_form()))


if __name__ == "__main__":
    main()
