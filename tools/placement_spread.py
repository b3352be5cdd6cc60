#!/usr/bin/env python3
"""The four arrays of the headline batch allocated with a temporary SPACER allocation between them (freed
afterwards: the arrays stay where they are, the spacers' memory returns to the allocator): kernel time for
spacers of 0 ... GiB, several batches each, in one process.
Usage (GPU box): python tools/placement_spread.py [batches per spacer size]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402

NX, NY, NLEV = 1440, 720, 137


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda", 0)
    ctx = fc.Context(0)
    xm, ym, _ = synth.grid_maps(NX, NY)
    dxm, dym = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
    flags = np.full(NLEV, fc.ALL_DEFINED, np.int32)

    def probe(v):
        ms = []
        for k in range(4):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(4):
                ctx.vortdiv_levels_enqueue(v[0], v[1], dxm, dym, v[2], v[3], fdefined=flags)
            e.record()
            torch.cuda.synchronize()
            if k:
                ms.append(s.elapsed_time(e) / 4)
        return float(np.median(ms))

    warm = [ctx.batch_empty(NLEV, NY, NX) for _ in range(4)]
    for _ in range(5):
        probe(warm)
    print("spacer GiB: kernel ms of %d batches allocated one after the other (earlier batches stay alive)" % nb)
    keep = [warm]
    for gib in (0, 1, 2, 3, 4, 6, 8, 12, 16):
        ts = []
        for b in range(nb):
            arrays, spacers = [], []
            for k in range(4):
                arrays.append(ctx.batch_empty(NLEV, NY, NX))
                if gib and k < 3:
                    spacers.append(torch.empty(gib << 30, dtype=torch.uint8, device=dev))
            del spacers
            torch.cuda.empty_cache()
            ts.append(probe(arrays))
            keep.append(arrays)
        print("  %2d  %s   median %.4f" % (gib, " ".join("%.4f" % t for t in ts), float(np.median(ts))), flush=True)


if __name__ == "__main__":
    main()
