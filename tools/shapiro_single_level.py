#!/usr/bin/env python3
"""measurement only (tools/): single-field shapiro2_filter calls on device-resident fields, us per call (20 back to back)."""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import torch
import mi_fieldcalc_amd as fc
import mi_fieldcalc_amd.synth as synth
dev = torch.device("cuda", 0)
ctx = fc.Context(0); ctx.use_torch_stream()
for nx, ny in ((1440, 720), (4000, 4000), (256, 256), (2880, 1440)):
    z = torch.from_numpy(synth.scalar_field(nx, ny, 5)).to(dev)
    out = torch.empty_like(z)
    for flag, name in ((fc.ALL_DEFINED, "ALL"), (fc.SOME_DEFINED, "SOME")):
        for _ in range(20):
            ctx.shapiro2_filter(z, fdefined=flag, out=out)
        torch.cuda.synchronize()
        ms = []
        for r in range(9):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                ctx.shapiro2_filter(z, fdefined=flag, out=out)
            e.record(); torch.cuda.synchronize()
            ms.append(s.elapsed_time(e) / 20)
        print("%dx%d one level, %s: %.1f us per call" % (nx, ny, name, 1e3 * float(np.median(ms))))
