#!/usr/bin/env python3
"""Does the time of the headline kernel depend on WHERE its four arrays were allocated?
One process; each trial frees everything (empty_cache), optionally keeps a dummy allocation of a random
size alive (so the next arrays land on other physical pages), allocates u, v, rvort, diverg anew and times
the library's default kernel, the row-walking kernel and the plain 2-in/2-out stream on them.
Usage (GPU box): python tools/alloc_variance.py [trials]
"""
import os
import sys
import random

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIFC_LIB_PATH", os.path.join(ROOT, "mi-fieldcalc_amd", "libmifc_measure.so"))  # mifc_timing_* / yardsticks: measurement build

import torch  # noqa: E402

import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402

NX, NY, NLEV = 1440, 720, 137


def timed(fn, rounds=5, inner=5):
    fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(inner):
            fn()
        e.record()
        torch.cuda.synchronize()
        out.append(s.elapsed_time(e) / inner)
    return float(np.median(out))


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda", 0)
    ctx = fc.Context(0)
    xm, ym, _ = synth.grid_maps(NX, NY)
    dxm, dym = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
    flags = np.full(NLEV, fc.ALL_DEFINED, np.int32)
    rng = random.Random(7)
    print("%5s %10s %12s %10s %10s   %s" % ("trial", "dummy MiB", "levelwalk ms", "rows ms", "stream ms", "addresses of u, v, rvort, diverg (GiB offsets from the lowest)"))
    for trial in range(trials):
        torch.cuda.empty_cache()
        dummy_mib = 0 if trial == 0 else rng.choice([0, 3, 64, 515, 1031, 2050, 4099])
        dummy = torch.empty(dummy_mib << 20, dtype=torch.uint8, device=dev) if dummy_mib else None
        du, dv = synth.device_wind(NX, NY, NLEV, 1234, dev)
        rv, dg = torch.empty_like(du), torch.empty_like(du)

        def run(mode):
            os.environ["MIFC_VORTDIV_LEVELWALK"] = mode
            ctx.reload_env()
            assert ctx.vortdiv_levels_enqueue(du, dv, dxm, dym, rv, dg, fdefined=flags)

        t_lw = timed(lambda: run("1"))
        t_rows = timed(lambda: run("0"))
        t_st = timed(lambda: ctx.bench_stream2(0, 0, rv, dg, du, dv))
        ptrs = [t.data_ptr() for t in (du, dv, rv, dg)]
        lo = min(ptrs)
        print("%5d %10d %12.4f %10.4f %10.4f   %s" % (trial, dummy_mib, t_lw, t_rows, t_st, " ".join("%.4f" % ((p - lo) / 2**30) for p in ptrs)), flush=True)
        del du, dv, rv, dg, dummy
    os.environ.pop("MIFC_VORTDIV_LEVELWALK", None)


if __name__ == "__main__":
    main()
