"""Latency of the single-field calls the reference API offers (one 1440x720 level, device-resident):
thermalFrontParameter / plevelqvector through the fused launch (band heights) and the multi-pass path."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIFC_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mi-fieldcalc_amd", "libmifc_measure.so"))  # mifc_timing_* / yardsticks: measurement build
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
    ctx.reload_env()
    tfp = lambda: ctx.thermalFrontParameter(z, dxm, dym, fdefined=fc.ALL_DEFINED, out=out)
    qv = lambda: ctx.plevelqvector(z, t, dxm, dym, dfc, 500.0, 1, fdefined=fc.ALL_DEFINED, out=out)
    print("%-22s  TFP %7.1f / %6.1f   Q-vector %7.1f / %6.1f" % (name, timed(tfp), kernel_us(tfp), timed(qv), kernel_us(qv)))

# ---- the other hot-path operators on one level (default settings)
for k in ("MIFC_FUSED2", "MIFC_FUSED2_BAND"):
    os.environ.pop(k, None)
ctx.reload_env()
u_np, v_np = synth.wind(NX, NY, 11)
u, v = torch.from_numpy(u_np).to(dev), torch.from_numpy(v_np).to(dev)
q = torch.full_like(z, 0.004)
tk = torch.full_like(z, 280.0) + 0.001 * (z - 5500.0)
ps = torch.full_like(z, 1000.0)
out2 = torch.empty_like(z)
print("\n%-34s %10s %10s" % ("operator, one level", "call us", "kernel us"))
for flag_name, flag in (("ALL_DEFINED", fc.ALL_DEFINED), ("SOME_DEFINED", fc.SOME_DEFINED)):
    ops = [
        ("relvort", lambda: ctx.relvort(u, v, dxm, dym, fdefined=flag, out=out)),
        ("divergence", lambda: ctx.divergence(u, v, dxm, dym, fdefined=flag, out=out)),
        ("gradient c=3", lambda: ctx.gradient(z, dxm, dym, 3, fdefined=flag, out=out)),
        ("plevelgvort", lambda: ctx.plevelgvort(z, dxm, dym, dfc, fdefined=flag, out=out)),
        ("vectorabs", lambda: ctx.vectorabs(u, v, fdefined=flag, out=out)),
        ("pleveltemp c=3", lambda: ctx.pleveltemp(tk, 850.0, "kelvin", 3, fdefined=flag, out=out)),
        ("hlevelhum c=1", lambda: ctx.hlevelhum(tk, q, ps, 12.5, 0.73, "kelvin", 1, fdefined=flag, out=out)),
        ("jacobian", lambda: ctx.jacobian(z, u, dxm, dym, fdefined=flag, out=out)),
        ("advection", lambda: ctx.advection(z, u, v, dxm, dym, 1.0, fdefined=flag, out=out)),
        ("shapiro2_filter", lambda: ctx.shapiro2_filter(z, fdefined=flag, out=out)),
        ("showalterIndex", lambda: ctx.showalterIndex(tk - 20.0, tk, q * 0 + 70.0, 500.0, 850.0, 1, fdefined=flag, out=out)),
    ]
    for name, fn in ops:
        print("%-34s %10.1f %10.1f" % (name + " (" + flag_name + ")", timed(fn), kernel_us(fn)))
