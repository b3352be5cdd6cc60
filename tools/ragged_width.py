#!/usr/bin/env python3
"""Widths that are not a multiple of 4 take the one-lane-per-cell stencil kernel: what does that cost?  Fused vorticity +
divergence and |grad| on a 65-level batch of 949 x 1069 (a 2.5-km regional grid) next to 948 and 952 columns.
Usage (GPU box): python tools/ragged_width.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402


def timed(fn, rounds=7, inner=5):
    fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(inner):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms.append(s.elapsed_time(e) / inner)
    return float(np.median(ms))


def main():
    dev = torch.device("cuda", 0)
    ctx = fc.Context(0)
    ctx.use_torch_stream()
    ny, nlev = 1069, 65
    print("%-10s %-34s %10s %10s" % ("nx", "operator", "ms", "% of 8 TB/s"))
    for nx in (948, 949, 950, 952):
        xm, ym, fcor = synth.grid_maps(nx, ny)
        dxm, dym = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
        du, dv = synth.device_wind(nx, ny, nlev, 3, dev)
        rv, dg = torch.empty_like(du), torch.empty_like(du)
        cnt = torch.zeros(nlev, dtype=torch.int64, device=dev)
        n = nx * ny * nlev
        for name, flag in (("vorticity + divergence, ALL", fc.ALL_DEFINED), ("vorticity + divergence, tested", fc.SOME_DEFINED)):
            flags = np.full(nlev, flag, np.int32)
            t = timed(lambda: ctx.vortdiv_levels_enqueue(du, dv, dxm, dym, rv, dg, fdefined=flags, n_undefined=cnt if flag != fc.ALL_DEFINED else None))
            print("%-10d %-34s %10.4f %10.1f" % (nx, name, t, n * 16 / t / 1e6 / 80.0))
        flags = np.full(nlev, fc.ALL_DEFINED, np.int32)
        t = timed(lambda: ctx.stencil_levels("gradient3", du, None, dxm, dym, fdefined=flags, out0=rv))
        print("%-10d %-34s %10.4f %10.1f" % (nx, "|grad f| (stencil_levels), ALL", t, n * 8 / t / 1e6 / 80.0))
        if os.environ.get("RAGGED_F1"):  # the rest of the stencil family (still on the per-cell / level-by-level paths for these widths)
            t = timed(lambda: ctx.stencil_levels("jacobian", du, dv, dxm, dym, fdefined=flags, out0=rv))
            print("%-10d %-34s %10.4f %10.1f" % (nx, "jacobian (stencil_levels), ALL", t, n * 12 / t / 1e6 / 80.0))
            t = timed(lambda: ctx.stencil_levels_ex("advection", du, du, dv, xmapr=dxm, ymapr=dym, scalar=1.0, fdefined=flags, out0=rv))
            print("%-10d %-34s %10.4f %10.1f" % (nx, "advection (stencil_levels_ex), ALL", t, n * 16 / t / 1e6 / 80.0))
            t = timed(lambda: ctx.stencil_levels_ex("thermalFrontParameter", du, xmapr=dxm, ymapr=dym, fdefined=flags, out0=rv))
            print("%-10d %-34s %10.4f %10.1f" % (nx, "TFP (stencil_levels_ex), ALL", t, n * 8 / t / 1e6 / 80.0))
            dfc = torch.from_numpy(fcor).to(dev)
            plev = np.linspace(1000.0, 100.0, nlev).astype(np.float32)
            t = timed(lambda: ctx.stencil_levels_ex("plevelqvector", du, dv, xmapr=dxm, ymapr=dym, fcoriolis=dfc, level_scalars=plev, compute=2,
                                                    fdefined=flags, out0=rv))
            print("%-10d %-34s %10.4f %10.1f" % (nx, "Q-vector x (stencil_levels_ex), ALL", t, n * 12 / t / 1e6 / 80.0))
            t = timed(lambda: ctx.stencil_levels_ex("shapiro2_filter", du, fdefined=flags, out0=rv))
            print("%-10d %-34s %10.4f %10.1f" % (nx, "shapiro2 (stencil_levels_ex), ALL", t, n * 8 / t / 1e6 / 80.0))


if __name__ == "__main__":
    main()
