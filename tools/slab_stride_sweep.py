#!/usr/bin/env python3
"""The four arrays of a level batch carved out of ONE device allocation, `stride` MiB apart: time of the fused
vorticity+divergence launch as a function of the stride.  (profiles/r03/experiments/vmm_placement_*.txt: with physical
memory the program maps itself the time is a smooth, reproducible function of the distance between the arrays; round 2 only
ever tried distances within 4 MiB of the array size, which is the slow end.)  No virtual-memory API here: one plain
allocation (torch.empty -> hipMalloc), repeated `--slabs` times with other allocations in between.

    python tools/slab_stride_sweep.py [--nlev 137] [--strides 544,560,...] [--slabs 3]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402

NX, NY = 1440, 720
MIB = 1 << 20


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nlev", type=int, default=137)
    ap.add_argument("--strides", default="")
    ap.add_argument("--slabs", type=int, default=3)
    ap.add_argument("--order", default="uvrd", help="which of the four slots u, v, rvort, diverg take (a permutation of uvrd)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    ctx = fc.Context(0)
    ctx.use_torch_stream()
    nlev = args.nlev
    n = NX * NY * nlev
    arr_mib = -(-n * 4 // MIB)
    strides = [int(s) for s in args.strides.split(",") if s] or [arr_mib + 2 + 16 * k for k in range(0, 20)]
    xm, ym, _ = synth.grid_maps(NX, NY)
    dxm, dym = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
    su, sv = synth.device_wind(NX, NY, nlev, 7, dev)
    flags = np.full(nlev, fc.ALL_DEFINED, np.int32)
    alg = 16 * n + 8 * NX * NY
    slot = {c: i for i, c in enumerate(args.order)}
    print("1440x720x%d: one array = %d MiB; arrays carved from ONE allocation, slots %s; ms per launch (8 back to back, median of 5), %% of 8 TB/s" % (nlev, arr_mib, args.order))
    print("stride MiB " + " ".join("%16s" % ("slab %d" % k) for k in range(args.slabs)))
    ballast = []
    table = {s: [] for s in strides}
    for k in range(args.slabs):
        slab = torch.empty(4 * max(strides) * MIB // 4, dtype=torch.float32, device=dev)
        for s in strides:
            views = [slab[i * s * MIB // 4: i * s * MIB // 4 + n].view(nlev, NY, NX) for i in range(4)]
            u, v, rv, dg = (views[slot[c]] for c in "uvrd")
            u.copy_(su)
            v.copy_(sv)
            for _ in range(6):
                ctx.vortdiv_levels_enqueue(u, v, dxm, dym, rv, dg, fdefined=flags)
            ms = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(8):
                    ctx.vortdiv_levels_enqueue(u, v, dxm, dym, rv, dg, fdefined=flags)
                e1.record()
                torch.cuda.synchronize()
                ms.append(e0.elapsed_time(e1) / 8)
            table[s].append(float(np.median(ms)))
        del slab, views, u, v, rv, dg
        ballast.append(torch.empty((300 + 211 * k) * MIB, dtype=torch.uint8, device=dev))  # the next slab lands somewhere else
        torch.cuda.empty_cache()
    for s in strides:
        print("%10d " % s + " ".join("%8.4f %6.1f%%" % (t, alg / t / 1e6 / 8000 * 100) for t in table[s]), flush=True)


if __name__ == "__main__":
    main()
