#!/usr/bin/env python3
"""A/B of the level-walking form against the row-walking form for the one-input stencil operators over a
device-resident level batch (MIFC_VORTDIV_LEVELWALK=0 / 1, interleaved rounds in one process).
Usage (GPU box): python tools/ab_levelwalk_ops.py [nx,ny,nlev]
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

OPS = [("gradient1", False, 8), ("gradient2", False, 8), ("gradient3", False, 8), ("gradient4", False, 8),
       ("plevelgwind_xcomp", True, 8), ("plevelgwind_ycomp", True, 8), ("plevelgvort", True, 8), ("ilevelgwind", True, 12)]


def main():
    nx, ny, nlev = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1440,720,137").split(","))
    dev = torch.device("cuda", 0)
    ctx = fc.Context(0)
    xm, ym, fcor = synth.grid_maps(nx, ny)
    dxm, dym, dfc = (torch.from_numpy(a).to(dev) for a in (xm, ym, fcor))
    z, _ = synth.device_wind(nx, ny, nlev, 99, dev)
    out0, out1 = torch.empty_like(z), torch.empty_like(z)
    flags = np.full(nlev, fc.ALL_DEFINED, np.int32)
    print("%dx%dx%d, kernel ms by HIP events around the launches (median of 7), ALL_DEFINED" % (nx, ny, nlev))
    print("%-20s %9s %9s %7s %12s" % ("operator", "rows ms", "walk ms", "ratio", "walk % 8TB/s"))
    for op, use_fc, bpc in OPS:
        def run(mode):
            os.environ["MIFC_VORTDIV_LEVELWALK"] = mode
            ctx.reload_env()
            r = ctx.stencil_levels(op, z, None, dxm, dym, dfc if use_fc else None, fdefined=flags, out0=out0, out1=out1 if op == "ilevelgwind" else None)
            assert r is not None
        res = {"0": [], "1": []}
        for m in res:
            run(m)
        torch.cuda.synchronize()
        for _ in range(7):
            for m in res:
                ctx.timing_begin()
                run(m)
                torch.cuda.synchronize()
                res[m].append(ctx.timing_end_ms())
        a, b = float(np.median(res["0"])), float(np.median(res["1"]))
        alg = nx * ny * nlev * bpc
        print("%-20s %9.4f %9.4f %7.3f %12.1f" % (op, a, b, b / a, alg / b / 1e6 / 8000 * 100), flush=True)
    os.environ.pop("MIFC_VORTDIV_LEVELWALK", None)


if __name__ == "__main__":
    main()
