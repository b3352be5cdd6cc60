"""One 4000x4000 level, read from HBM (three rotating buffer sets): the one-input stencil operators in the form the
launcher picks for one or two levels (one-shot) against the row-walking form (MIFC_SCALAR_ROWS_R=8)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIFC_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mi-fieldcalc_amd", "libmifc_measure.so"))  # mifc_timing_* / yardsticks: measurement build
import numpy as np
import torch

import mi_fieldcalc_amd as fc
import mi_fieldcalc_amd.synth as synth

nx = ny = 4000
dev = torch.device("cuda", 0)
ctx = fc.Context(0)
xm, ym, fcor = synth.grid_maps(nx, ny, h=2500.0)
sets = []
for k in range(3):
    z = torch.from_numpy(synth.scalar_field(nx, ny, 40 + k)).to(dev)
    sets.append((z, torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev), torch.from_numpy(fcor).to(dev), torch.empty_like(z)))
state = {"k": 0}


def timed(fn, n=21):
    for _ in range(3):
        fn()
    tot = 0.0
    for _ in range(n):
        ctx.timing_begin()
        fn()
        tot += ctx.timing_end_ms()
    return tot / n


for env in ({}, {"MIFC_SCALAR_ROWS_R": "8"}):
    os.environ.pop("MIFC_SCALAR_ROWS_R", None)
    os.environ.update(env)
    ctx.reload_env()
    for name, nbytes, call in (
        ("gradient c=3", 16, lambda s: ctx.gradient(s[0], s[1], s[2], 3, fdefined=fc.ALL_DEFINED, out=s[4])),
        ("plevelgvort", 20, lambda s: ctx.plevelgvort(s[0], s[1], s[2], s[3], fdefined=fc.ALL_DEFINED, out=s[4])),
    ):
        def run():
            state["k"] += 1
            call(sets[state["k"] % 3])
        ms = timed(run)
        print("%-14s %-24s %.4f ms  %.1f %% of 8 TB/s (%d B/cell)" % (name, "row-walking, 8-row bands" if env else "default (one-shot)", ms,
                                                                   100 * nx * ny * nbytes / ms / 1e6 / 8000.0, nbytes))
