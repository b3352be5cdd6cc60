#!/usr/bin/env python3
"""Follow-up of placement_distribution.py: is the LAST candidate special, and does freeing the others change
the winner's time?  12 candidates probed in two rounds; losers freed; winner probed again; then 3 fresh batches.
Usage (GPU box): python tools/placement_followup.py
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
    dev = torch.device("cuda", 0)
    ctx = fc.Context(0)
    xm, ym, _ = synth.grid_maps(NX, NY)
    dxm, dym = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
    flags = np.full(NLEV, fc.ALL_DEFINED, np.int32)
    su, sv = synth.device_wind(NX, NY, NLEV, 1234, dev)

    def alloc():
        arrays = tuple(ctx.batch_empty(NLEV, NY, NX) for _ in range(4))
        arrays[0].copy_(su)
        arrays[1].copy_(sv)
        return arrays

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

    cands, spacers = [], []
    first = []
    for i in range(12):
        mib = SPACERS_MIB[i % len(SPACERS_MIB)]
        if mib:
            spacers.append(torch.empty(mib << 20, dtype=torch.uint8, device=dev))
        cands.append(alloc())
        if i == 0:
            for _ in range(3):
                probe(cands[0])
        first.append(probe(cands[-1]))
    print("probed right after its allocation: " + " ".join("%.4f" % t for t in first))
    for r in range(2):
        print("all alive, round %d:               " % (r + 1) + " ".join("%.4f" % probe(c) for c in cands))
    best = int(np.argmin(first))
    keep = cands[best]
    del cands, spacers
    torch.cuda.empty_cache()
    print("winner %d after the others were freed: %.4f %.4f" % (best, probe(keep), probe(keep)))
    fresh = [alloc() for _ in range(3)]
    print("three fresh batches allocated now:     " + " ".join("%.4f" % probe(c) for c in fresh))
    print("winner again:                          %.4f" % probe(keep))


if __name__ == "__main__":
    main()
