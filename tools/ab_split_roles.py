#!/usr/bin/env python3
"""Level-walking kernel (the library's own choice) against the split-role form (MIFC_VORTDIV_TUNE="K=4,...") over a list
of shapes, interleaved rounds in one process on the same arrays; ALL_DEFINED and the tested variant.
Usage (GPU box): python tools/ab_split_roles.py [nx,ny,nlev ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402

SHAPES = ["1440,720,137", "1440,720,32", "1440,720,12", "1440,720,6", "720,360,137", "360,180,137", "1000,1000,10", "2880,1440,20", "4000,4000,8", "4000,4000,3"]
ROUNDS, INNER = 7, 5


def main():
    shapes = sys.argv[1:] or SHAPES
    dev = torch.device("cuda", 0)
    ctx = fc.Context(0)
    print("%-16s %-6s %12s %12s %12s %12s" % ("shape", "flags", "default ms", "K=4 RB=14", "K=4 RB=12", "K=4 RB=10"))
    for shp in shapes:
        nx, ny, nlev = (int(x) for x in shp.split(","))
        xm, ym, _ = synth.grid_maps(nx, ny)
        dxm, dym = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
        du, dv = synth.device_wind(nx, ny, nlev, 1234, dev)
        rv, dg = torch.empty_like(du), torch.empty_like(du)
        cnt = torch.zeros(nlev, dtype=torch.int64, device=dev)
        target = 6 if nlev >= 48 else 8
        nchunks = -(-nlev // target)
        lg = -(-nlev // nchunks)
        tunes = ["", "K=4,RB=14,D=1,LG=%d" % lg, "K=4,RB=12,D=1,LG=%d" % lg, "K=4,RB=10,D=1,LG=%d" % lg]
        for tested in (False, True):
            flags = np.full(nlev, fc.SOME_DEFINED if tested else fc.ALL_DEFINED, np.int32)

            def run(t):
                if t:
                    os.environ["MIFC_VORTDIV_TUNE"] = t
                else:
                    os.environ.pop("MIFC_VORTDIV_TUNE", None)
                ctx.reload_env()
                assert ctx.vortdiv_levels_enqueue(du, dv, dxm, dym, rv, dg, fdefined=flags, n_undefined=cnt if tested else None)

            res = {t: [] for t in tunes}
            for t in tunes:
                run(t)
            torch.cuda.synchronize()
            for _ in range(ROUNDS):
                for t in tunes:
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for _ in range(INNER):
                        run(t)
                    e.record()
                    torch.cuda.synchronize()
                    res[t].append(s.elapsed_time(e) / INNER)
            print("%-16s %-6s " % (shp, "SOME" if tested else "ALL") + " ".join("%12.4f" % float(np.median(res[t])) for t in tunes), flush=True)
        del du, dv, rv, dg
    os.environ.pop("MIFC_VORTDIV_TUNE", None)


if __name__ == "__main__":
    main()
