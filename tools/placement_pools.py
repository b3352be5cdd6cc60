#!/usr/bin/env python3
"""Is "this process found no fast placement" a property of the POOL of arrays or of the process?  Three pools of 24 arrays
allocated one after the other (all alive), 40 random index sets of each probed with the headline kernel; then the same
again after everything was freed.  Usage (GPU box): python tools/placement_pools.py"""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402

NX, NY, NLEV = 1440, 720, 137


def main():
    dev = torch.device("cuda", 0)
    ctx = fc.Context(0)
    ctx.use_torch_stream()
    xm, ym, _ = synth.grid_maps(NX, NY)
    dxm, dym = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
    flags = np.full(NLEV, fc.ALL_DEFINED, np.int32)

    def probe(arrs):
        a, b, c, d = arrs
        ms = []
        for k in range(4):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(4):
                ctx.vortdiv_levels_enqueue(a, b, dxm, dym, c, d, fdefined=flags, n_undefined=None)
            e.record()
            torch.cuda.synchronize()
            if k:
                ms.append(s.elapsed_time(e) / 4)
        return float(np.median(ms))

    rng = random.Random(3)
    for epoch in range(2):
        pools = []
        for p in range(3):
            pool = [ctx.batch_empty(NLEV, NY, NX) for _ in range(24)]
            pools.append(pool)
            if epoch == 0 and p == 0:
                for _ in range(15):
                    probe(pool[:4])
            ts = [probe([pool[i] for i in rng.sample(range(24), 4)]) for _ in range(40)]
            lo = min(int(x.data_ptr()) for x in pool) >> 20
            print("epoch %d pool %d (lowest address %d MiB): adjacent %.4f  random sets min %.4f median %.4f max %.4f" %
                  (epoch, p, lo, probe(pool[:4]), min(ts), float(np.median(ts)), max(ts)), flush=True)
        # mixed: inputs from one pool, outputs from another
        ts = [probe([pools[0][rng.randrange(24)], pools[0][rng.randrange(24)], pools[2][rng.randrange(24)], pools[2][rng.randrange(24)]]) for _ in range(20)]
        print("epoch %d inputs from pool 0, outputs from pool 2: min %.4f median %.4f" % (epoch, min(ts), float(np.median(ts))), flush=True)
        del pools, pool
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
