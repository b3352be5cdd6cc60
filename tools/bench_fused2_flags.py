"""Kernel time of the fused thermalFrontParameter / plevelqvector launch on the 1440x98640 tall field with an
ALL_DEFINED and a SOME_DEFINED input flag (the latter runs the tested variant and the edge-count workgroups)."""
import os, sys; sys.path.insert(0, '.')
os.environ.setdefault("MIFC_LIB_PATH", os.path.join(os.getcwd(), "mi-fieldcalc_amd", "libmifc_measure.so"))  # measurement build
import numpy as np, torch
import mi_fieldcalc_amd as fc, mi_fieldcalc_amd.synth as synth
NX, NY, NLEV = 1440, 720, 137
dev = torch.device("cuda", 0)
ctx = fc.Context(0)
xm, ym, fcor = synth.grid_maps(NX, NY)
xm_t, ym_t, fc_t = (torch.from_numpy(a).to(dev).repeat(NLEV, 1).contiguous() for a in (xm, ym, fcor))
z = torch.from_numpy(synth.scalar_field(NX, NY, 5)).to(dev).repeat(NLEV, 1).contiguous()
t = (250.0 + 0.05 * (z - 5500.0)).contiguous()
out = torch.empty_like(z)
for name, flag in (("ALL", fc.ALL_DEFINED), ("SOME", fc.SOME_DEFINED)):
    for op, fn in (("TFP", lambda: ctx.thermalFrontParameter(z, xm_t, ym_t, fdefined=flag, out=out)),
                   ("QVEC", lambda: ctx.plevelqvector(z, t, xm_t, ym_t, fc_t, 500.0, 1, fdefined=flag, out=out))):
        for _ in range(3): fn()
        tot = 0.0
        for _ in range(10):
            ctx.timing_begin(); fn(); tot += ctx.timing_end_ms()
        print("%s %s tall field: kernels %.4f ms" % (op, name, tot / 10))
