#!/usr/bin/env python3
"""Where do the RAGGED split-role kernels differ from the oracle?  (GPU box; prints mismatch positions per operator.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MIFC_LEVELWALK_MIN_UNITS"] = "1"

import torch  # noqa: E402

import cases  # noqa: E402
import cpulib  # noqa: E402
import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402


def main():
    nx, ny, nlev = (int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "949,23,4").split(","))
    oracle = cpulib.CpuLib("oracle")
    ctx = fc.Context(0)
    xm, ym, fcor = synth.grid_maps(nx, ny)
    z = np.stack([synth.scalar_field(nx, ny, 4300 + l) for l in range(nlev)])
    flags = np.full(nlev, fc.SOME_DEFINED, np.int32)
    flags[0] = fc.ALL_DEFINED
    for l in range(1, nlev):
        if l % 2:
            z[l] = synth.sprinkle_undef(z[l], 20 + l, 0.03)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    for name in ("plevelgwind_xcomp", "plevelgwind_ycomp", "plevelgvort", "ilevelgwind", "gradient3"):
        args_tail = [xm, ym] + ([fcor] if name != "gradient3" else []) + ([3] if name == "gradient3" else [])
        (r0, r1), fo = ctx.stencil_levels(name, dev(z), None, dev(xm), dev(ym), dev(fcor) if name != "gradient3" else None, fdefined=flags)
        r0 = r0.cpu().numpy()
        for l in range(nlev):
            ok, e, f_e = oracle.call("gradient" if name == "gradient3" else name, nx, ny, z[l], *args_tail, fdefined=int(flags[l]))
            e0 = e[0] if isinstance(e, tuple) else e
            diff = np.argwhere(r0[l].view(np.int32) != e0.view(np.int32))
            print(name, "level", l, "flag", fo[l], f_e, "mismatches", len(diff), diff[:8].tolist(), [(float(r0[l][tuple(d)]), float(e0[tuple(d)])) for d in diff[:4]])


if __name__ == "__main__":
    main()
