#!/usr/bin/env python3
"""Where the four arrays of the headline kernel lie, and how far apart the levels of a
batch are, against kernel time -- all in ONE process, interleaved rounds.

The fused vorticity+divergence kernel reads u, v and writes rvort, diverg
(1440x720x137 float32 each).  Round 1 saw the same binary swing 62.7-70.4 % of 8 TB/s
between processes with where four separate allocations landed.  Here everything is
carved out of one slab so that the relative offsets are ours to choose:

  sweep A  level stride (floats) with the four arrays packed back to back
  sweep B  gap between consecutive arrays (bytes) at a fixed level stride

Usage (GPU box):  python tools/sweep_placement.py [A] [B]   (default: both)
Prints one line per configuration: median / min ms, % of 8 TB/s."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402

NX, NY, NLEV = (int(x) for x in os.environ.get("SWEEP_SHAPE", "1440,720,137").split(","))
ROUNDS = int(os.environ.get("SWEEP_ROUNDS", "5"))
INNER = int(os.environ.get("SWEEP_INNER", "5"))
N = NX * NY
ALG = N * NLEV * 16 + 2 * N * 4


def main():
    which = [a.upper() for a in sys.argv[1:]] or ["A", "B"]
    dev = torch.device("cuda", 0)
    ctx = fc.Context(0)
    ctx.use_torch_stream()
    xm, ym, _ = synth.grid_maps(NX, NY)
    dxm, dym = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
    su, sv = synth.device_wind(NX, NY, NLEV, 1234, dev)  # sources, copied into every layout
    flags = np.full(NLEV, fc.ALL_DEFINED, np.int32)

    max_ls = N + 64 * NX
    max_gap = 8 << 20
    slab = torch.empty(4 * (NLEV * max_ls + max_gap // 4) + 1024, dtype=torch.float32, device=dev)
    base_off = (-slab.data_ptr() // 4) % (1 << 19)  # start the layouts on a 2 MiB boundary
    print("slab base %#x (+%d floats to the next 2 MiB boundary)" % (slab.data_ptr(), base_off))

    def layout(ls, gaps_bytes):
        """four (NLEV, NY, NX) views: array k starts gap_k bytes after the end of array k-1"""
        views = []
        off = base_off
        for k in range(4):
            off += gaps_bytes[k] // 4
            views.append(torch.as_strided(slab, (NLEV, NY, NX), (ls, NX, 1), storage_offset=off))
            off += NLEV * ls
        return views

    configs = []
    if "A" in which:
        for pad in (0, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, NX, 2 * NX, 3 * NX, 4 * NX, 5 * NX, 6 * NX, 7 * NX, 8 * NX, 11 * NX, 16 * NX, 23 * NX,
                    32 * NX, (1 << 20) - N % (1 << 20) if N % (1 << 20) else 0, (1 << 20) - N % (1 << 20) + 256, (1 << 20) - N % (1 << 20) + 4096):
            if pad % 4 == 0 and N + pad <= max_ls:
                configs.append(("A level stride n+%d floats (%d B)" % (pad, 4 * (N + pad)), N + pad, (0, 0, 0, 0)))
    if "B" in which:
        for gap in (0, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1 << 20, 3 << 19, 2 << 20, 4 << 20):
            configs.append(("B gap %d B between arrays, stride n" % gap, N, (0, gap, gap, gap)))
        for gap in (256, 4096, 65536, 1 << 20):
            configs.append(("B gap %d B only between inputs and outputs" % gap, N, (0, 0, gap, 0)))
    # the reference point: four separate allocations from torch's caching allocator
    sep = [su, sv, torch.empty_like(su), torch.empty_like(su)]

    times = {name: [] for name, _, _ in configs}
    times["separate torch allocations"] = []

    def run(arrs):
        assert ctx.vortdiv_levels_enqueue(arrs[0], arrs[1], dxm, dym, arrs[2], arrs[3], fdefined=flags)

    def timed(arrs):
        run(arrs)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(INNER):
            run(arrs)
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / INNER

    for r in range(ROUNDS):
        times["separate torch allocations"].append(timed(sep))
        for name, ls, gaps in configs:
            arrs = layout(ls, gaps)
            arrs[0].copy_(su)
            arrs[1].copy_(sv)
            times[name].append(timed(arrs))
        print("round %d done" % r, flush=True)
    print("shape %dx%dx%d, algorithmic bytes %.3f GB, %d rounds x %d launches" % (NX, NY, NLEV, ALG / 1e9, ROUNDS, INNER))
    print("%-62s %8s %8s %7s" % ("layout", "med ms", "min ms", "frac"))
    for name, ts in times.items():
        med, mn = float(np.median(ts)), float(np.min(ts))
        print("%-62s %8.4f %8.4f %7.4f" % (name, med, mn, ALG / med / 1e6 / 8000.0))


if __name__ == "__main__":
    main()
