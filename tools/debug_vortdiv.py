#!/usr/bin/env python3
"""Debug helper (GPU box): where does the fused kernel differ from the oracle?
   python tools/debug_vortdiv.py NX NY NLEV ["tune"]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402
from cpulib import CpuLib  # noqa: E402

nx, ny, nlev = (int(a) for a in sys.argv[1:4])
if len(sys.argv) > 4:
    os.environ["MIFC_VORTDIV_TUNE"] = sys.argv[4]
oracle = CpuLib("oracle")
xm, ym, _ = synth.grid_maps(nx, ny)
u, v = synth.wind(nx, ny, 77, nlev=nlev)
flags = np.full(nlev, fc.ALL_DEFINED, np.int32)
ctx = fc.Context(0)
(rv, dg), fo = ctx.vortdiv_levels(u, v, xm, ym, fdefined=flags)
for l in range(nlev):
    for name, got, op in (("rvort", rv[l], "relvort"), ("diverg", dg[l], "divergence")):
        ok, e, _ = oracle.call(op, nx, ny, u[l], v[l], xm, ym, fdefined=fc.ALL_DEFINED)
        bad = np.argwhere(got.view(np.uint32) != e.view(np.uint32))
        if len(bad):
            cols = sorted(set(int(b[1]) for b in bad))
            rows = sorted(set(int(b[0]) for b in bad))
            print("level %d %s: %d cells differ; cols %s%s rows %s%s" % (
                l, name, len(bad), cols[:24], "..." if len(cols) > 24 else "", rows[:12], "..." if len(rows) > 12 else ""))
            j, i = bad[0]
            print("   first (%d,%d): got %r expected %r" % (j, i, got[j, i], e[j, i]))
        else:
            print("level %d %s: identical" % (l, name))
