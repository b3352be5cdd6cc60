#!/usr/bin/env python3
"""When should the level-walking form of the wind kernel take over from the row-walking one?
Times the library's own choice with MIFC_VORTDIV_LEVELWALK=1 and =0 (interleaved rounds, one process)
over a list of shapes, both for the fused pair and for a single output (relative vorticity).
Usage (GPU box): python tools/levelwalk_threshold.py [nx,ny,nlev ...]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402

SHAPES = ["1440,720,3", "1440,720,4", "1440,720,8", "1440,720,16", "1440,720,32", "1440,720,64", "1440,720,137",
          "720,360,137", "360,180,137", "4000,4000,3", "4000,4000,8", "2880,1440,20"]
ROUNDS, INNER = 7, 5


def main():
    shapes = sys.argv[1:] or SHAPES
    dev = torch.device("cuda", 0)
    ctx = fc.Context(0)
    print("%-16s %-8s %11s %11s %8s" % ("shape", "outputs", "rows ms", "levelwalk ms", "ratio"))
    for shp in shapes:
        nx, ny, nlev = (int(x) for x in shp.split(","))
        xm, ym, _ = synth.grid_maps(nx, ny)
        dxm, dym = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
        du, dv = synth.device_wind(nx, ny, nlev, 1234, dev)
        rv, dg = torch.empty_like(du), torch.empty_like(du)
        flags = np.full(nlev, fc.ALL_DEFINED, np.int32)

        def run(mode, both):
            os.environ["MIFC_VORTDIV_LEVELWALK"] = mode
            ctx.reload_env()
            assert ctx.vortdiv_levels_enqueue(du, dv, dxm, dym, rv, dg if both else None, fdefined=flags)

        for both in (True, False):
            res = {"0": [], "1": []}
            for m in res:
                run(m, both)
            torch.cuda.synchronize()
            for _ in range(ROUNDS):
                for m in res:
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for _ in range(INNER):
                        run(m, both)
                    e.record()
                    torch.cuda.synchronize()
                    res[m].append(s.elapsed_time(e) / INNER)
            a, b = float(np.median(res["0"])), float(np.median(res["1"]))
            print("%-16s %-8s %11.4f %11.4f %8.3f" % (shp, "rv+div" if both else "rv", a, b, b / a), flush=True)
        del du, dv, rv, dg
    os.environ.pop("MIFC_VORTDIV_LEVELWALK", None)


if __name__ == "__main__":
    main()
