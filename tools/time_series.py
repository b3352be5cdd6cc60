#!/usr/bin/env python3
"""Time series of the headline kernel on ONE set of buffers: does the kernel time drift with time (clocks,
power, neighbours on the node) while nothing about the allocation changes?  Then the same after re-allocating.
Usage (GPU box): python tools/time_series.py [samples] [reallocations]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402

NX, NY, NLEV = 1440, 720, 137


def main():
    samples = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    reallocs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dev = torch.device("cuda", 0)
    ctx = fc.Context(0)
    xm, ym, _ = synth.grid_maps(NX, NY)
    dxm, dym = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
    flags = np.full(NLEV, fc.ALL_DEFINED, np.int32)
    t0 = time.time()
    for r in range(reallocs):
        torch.cuda.empty_cache()
        du, dv = synth.device_wind(NX, NY, NLEV, 1234, dev)
        rv, dg = torch.empty_like(du), torch.empty_like(du)
        ctx.vortdiv_levels_enqueue(du, dv, dxm, dym, rv, dg, fdefined=flags)
        torch.cuda.synchronize()
        series = []
        for i in range(samples):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                ctx.vortdiv_levels_enqueue(du, dv, dxm, dym, rv, dg, fdefined=flags)
            e.record()
            torch.cuda.synchronize()
            series.append(s.elapsed_time(e) / 10)
            if i % 10 == 9:
                time.sleep(0.5)  # an idle gap: clocks may drop
        print("allocation %d at t=%.1fs u@%#x: min %.4f median %.4f max %.4f" % (r, time.time() - t0, du.data_ptr(), min(series), float(np.median(series)), max(series)))
        print("   " + " ".join("%.3f" % x for x in series), flush=True)
        del du, dv, rv, dg


if __name__ == "__main__":
    main()
