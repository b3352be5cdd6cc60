#!/usr/bin/env python3
"""The f1 operators over a 137-level batch with SHARED map factors (mifc_stencil_levels_ex) against the
round-1 way of batching them (one tall 1440 x 98640 field with the map factors tiled per level):
kernel time by HIP events, % of 8 TB/s on the algorithmic bytes (maps counted once per batch)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIFC_LIB_PATH", os.path.join(ROOT, "mi-fieldcalc_amd", "libmifc_measure.so"))  # mifc_timing_* / yardsticks: measurement build

import torch  # noqa: E402

import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402

NX, NY, NLEV = 1440, 720, int(sys.argv[1]) if len(sys.argv) > 1 else 137


def main():
    dev = torch.device("cuda", 0)
    ctx = fc.Context(0)
    xm, ym, fcor = synth.grid_maps(NX, NY)
    dxm, dym, dfc = (torch.from_numpy(a).to(dev) for a in (xm, ym, fcor))
    u, v = synth.device_wind(NX, NY, NLEV, 5, dev)
    z = (5500.0 + 6.0 * u + 0.5 * v).contiguous()
    t = (250.0 + 0.05 * (z - 5500.0)).contiguous()
    out = torch.empty_like(z)
    pres = np.linspace(1000.0, 100.0, NLEV).astype(np.float32)
    n = NX * NY * NLEV
    tall = lambda a: a.repeat(NLEV, 1).contiguous()
    txm, tym, tfc = tall(dxm), tall(dym), tall(dfc)
    zt, tt_, ut, vt, ot = (a.view(NLEV * NY, NX) for a in (z, t, u, v, out))
    print("%dx%dx%d, kernel ms (HIP events around the launches, median of 7); %% of 8 TB/s on bytes/cell + maps once" % (NX, NY, NLEV))
    print("%-44s %10s %8s %8s" % ("operator", "flag", "ms", "frac"))
    for flag in (fc.ALL_DEFINED, fc.SOME_DEFINED):
        flags = np.full(NLEV, flag, np.int32)
        rows = [
            ("advection, level batch", 16, 2, lambda: ctx.stencil_levels_ex("advection", z, u, v, dxm, dym, scalar=1.0, fdefined=flags, out0=out)),
            ("advection, tall field (maps per cell)", 24, 0, lambda: ctx.advection(zt, ut, vt, txm, tym, 1.0, fdefined=flag, out=ot)),
            ("thermalFrontParameter, level batch", 8, 2, lambda: ctx.stencil_levels_ex("thermalFrontParameter", z, xmapr=dxm, ymapr=dym, fdefined=flags, out0=out)),
            ("thermalFrontParameter, tall field", 16, 0, lambda: ctx.thermalFrontParameter(zt, txm, tym, fdefined=flag, out=ot)),
            ("plevelqvector c=1, level batch", 12, 3, lambda: ctx.stencil_levels_ex("plevelqvector", z, t, None, dxm, dym, dfc, level_scalars=pres, compute=1, fdefined=flags, out0=out)),
            ("plevelqvector c=1, tall field", 24, 0, lambda: ctx.plevelqvector(zt, tt_, txm, tym, tfc, 700.0, 1, fdefined=flag, out=ot)),
            ("shapiro2_filter, level batch", 8, 0, lambda: ctx.stencil_levels_ex("shapiro2_filter", z, fdefined=flags, out0=out)),
            ("shapiro2_filter, tall field", 8, 0, lambda: ctx.shapiro2_filter(zt, fdefined=flag, out=ot)),
        ]
        for name, bpc, nmaps, call in rows:
            ms = []
            for r in range(9):
                ctx.timing_begin()
                assert call() is not None
                k = ctx.timing_end_ms()
                if r >= 2:
                    ms.append(k)
            med = float(np.median(ms))
            alg = n * bpc + nmaps * NX * NY * 4
            print("%-44s %10s %8.4f %8.4f" % (name, "ALL" if flag == fc.ALL_DEFINED else "SOME", med, alg / med / 1e6 / 8000.0))


if __name__ == "__main__":
    main()
