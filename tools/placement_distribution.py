#!/usr/bin/env python3
"""Distribution of the headline kernel's time over candidate placements of its batch in ONE process:
N candidates allocated one after the other (all alive, spacers in between as in placement.choose_placement),
each probed three times in rotation so that drift over time and placement can be told apart.
Usage (GPU box): python tools/placement_distribution.py [candidates]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402
from mi_fieldcalc_amd.placement import SPACERS_MIB  # noqa: E402

NX, NY, NLEV = 1440, 720, 137


def main():
    ncand = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = torch.device("cuda", 0)
    ctx = fc.Context(0)
    xm, ym, _ = synth.grid_maps(NX, NY)
    dxm, dym = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
    flags = np.full(NLEV, fc.ALL_DEFINED, np.int32)
    su, sv = synth.device_wind(NX, NY, NLEV, 1234, dev)
    cands, spacers = [], []
    for i in range(ncand):
        mib = SPACERS_MIB[i % len(SPACERS_MIB)]
        if mib:
            spacers.append(torch.empty(mib << 20, dtype=torch.uint8, device=dev))
        arrays = tuple(ctx.batch_empty(NLEV, NY, NX) for _ in range(4))
        arrays[0].copy_(su)
        arrays[1].copy_(sv)
        cands.append(arrays)

    def probe(arrays):
        a, b, c, d = arrays
        ms = []
        for k in range(6):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(4):
                ctx.vortdiv_levels_enqueue(a, b, dxm, dym, c, d, fdefined=flags)
            e.record()
            torch.cuda.synchronize()
            if k:
                ms.append(s.elapsed_time(e) / 4)
        return float(np.median(ms))

    for _ in range(3):
        probe(cands[0])  # settle the clocks
    rounds = [[probe(c) for c in cands] for _ in range(3)]
    print("candidate  round1  round2  round3   (ms per launch, 1440x720x137 level-walking kernel)")
    for i in range(ncand):
        print("%9d  %.4f  %.4f  %.4f" % (i, rounds[0][i], rounds[1][i], rounds[2][i]))
    best = [min(r) for r in rounds]
    print("best of the first 1/2/4/6/8/12/all candidates (round 3): " + " ".join("%.4f" % min(rounds[2][:k]) for k in (1, 2, 4, 6, 8, 12, ncand) if k <= ncand))
    print("min / median / max over candidates, round 3: %.4f / %.4f / %.4f" % (min(rounds[2]), float(np.median(rounds[2])), max(rounds[2])))


if __name__ == "__main__":
    main()
