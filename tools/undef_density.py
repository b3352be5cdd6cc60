#!/usr/bin/env python3
"""How much do the per-wave atomics of the undefined counts cost when undefined values are EVERYWHERE (a masked field:
every wave has something to count) rather than nowhere or in one corner?  Fused vorticity + divergence (tested variant,
137 levels) and two single-field elementwise operators, kernel ms by HIP events, one process.
Usage (GPU box): python tools/undef_density.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402

NX, NY, NLEV = 1440, 720, 137


def timed(fn, rounds=7, inner=3):
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
    xm, ym, _ = synth.grid_maps(NX, NY)
    dxm, dym = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
    du, dv = synth.device_wind(NX, NY, NLEV, 99, dev)
    rv, dg = torch.empty_like(du), torch.empty_like(du)
    cnt = torch.zeros(NLEV, dtype=torch.int64, device=dev)
    flags = np.full(NLEV, fc.SOME_DEFINED, np.int32)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    print("%-44s %10s" % ("case", "kernel ms"))
    for name, frac in (("no undefined value", 0.0), ("one corner of every level", -1.0), ("0.1 % of the cells, everywhere", 0.001), ("2 % of the cells, everywhere", 0.02),
                       ("30 % of the cells, everywhere", 0.3)):
        u = du.clone()
        if frac < 0:
            u[:, 100:110, 200:260] = float(fc.UNDEF)
        elif frac > 0:
            m = torch.rand(u.shape, generator=g, device=dev) < frac
            u[m] = float(fc.UNDEF)
            del m
        t = timed(lambda: ctx.vortdiv_levels_enqueue(u, dv, dxm, dym, rv, dg, fdefined=flags, n_undefined=cnt))
        print("%-44s %10.4f   (undefined cells counted in level 0: %d)" % ("vortdiv tested, " + name, t, int(cnt[0].item())))
        for gop in ("gradient1", "gradient3"):  # the level-walking and the row-walking scalar kernels
            tg = timed(lambda: ctx.stencil_levels(gop, u, None, dxm, dym, None, fdefined=flags, out0=rv))
            print("%-44s %10.4f" % (gop + " tested, " + name, tg))
        # single-field elementwise operator on the tall field: vectorabs with the tested flag
        tall_u, tall_v = u.view(NLEV * NY, NX), dv.view(NLEV * NY, NX)
        out = rv.view(NLEV * NY, NX)
        t2 = timed(lambda: ctx.vectorabs(tall_u, tall_v, fdefined=fc.SOME_DEFINED, out=out))
        print("%-44s %10.4f" % ("vectorabs tested (sync call), " + name, t2))
        # one level (the reference's per-field call pattern), enqueue form: kernel + memset, no host round trip
        u1, v1 = u[:1], dv[:1]
        f1 = np.full(1, fc.SOME_DEFINED, np.int32)
        t3 = timed(lambda: ctx.vortdiv_levels_enqueue(u1, v1, dxm, dym, rv[:1], dg[:1], fdefined=f1, n_undefined=cnt[:1]), inner=20)
        print("%-44s %10.4f" % ("vortdiv tested, ONE level, " + name, t3))
        del u


if __name__ == "__main__":
    main()
