"""Latency of the single-field calls the reference API offers (one 1440x720 level, device-resident):
thermalFrontParameter / plevelqvector through the fused launch (band heights) and the multi-pass path."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import mi_fieldcalc_amd as fc
import mi_fieldcalc_amd.synth as synth

NX, NY = (int(x) for x in os.environ.get("SHAPE", "1440,720").split(","))
dev = torch.device("cuda", 0)
ctx = fc.Context(0)
xm, ym, fcor = synth.grid_maps(NX, NY)
dxm, dym, dfc = (torch.from_numpy(a).to(dev) for a in (xm, ym, fcor))
z = torch.from_numpy(synth.scalar_field(NX, NY, 5)).to(dev)
t = (250.0 + 0.05 * (z - 5500.0)).contiguous()
out = torch.empty_like(z)


def timed(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def kernel_us(fn, n=50):
    tot = 0.0
    for _ in range(n):
        ctx.timing_begin()
        fn()
        tot += ctx.timing_end_ms()
    return tot / n * 1e3


configs = [("multi-pass", {"MIFC_FUSED2": "0"})] + [("fused, band %s" % b, {"MIFC_FUSED2": "1", "MIFC_FUSED2_BAND": b}) for b in ("1", "2", "4", "8", "16")] + [("fused, default band", {"MIFC_FUSED2": "1"})]
print("%dx%d, one level, device-resident; synchronous call / kernels only (us)" % (NX, NY))
for name, env in configs:
    for k in ("MIFC_FUSED2", "MIFC_FUSED2_BAND"):
        os.environ.pop(k, None)
    os.environ.update(env)
    tfp = lambda: ctx.thermalFrontParameter(z, dxm, dym, fdefined=fc.ALL_DEFINED, out=out)
    qv = lambda: ctx.plevelqvector(z, t, dxm, dym, dfc, 500.0, 1, fdefined=fc.ALL_DEFINED, out=out)
    print("%-22s  TFP %7.1f / %6.1f   Q-vector %7.1f / %6.1f" % (name, timed(tfp), kernel_us(tfp), timed(qv), kernel_us(qv)))
