#!/usr/bin/env python3
"""tested_variants.py -- measurement only (tools/): every batched stencil operator on clean data, once with the levels'
flags ALL_DEFINED (no per-cell tests) and once SOME_DEFINED (per-cell undefined tests + per-level counts): what the tests cost.
1440 x 720 x 137 by default.  Usage (GPU box): python3 tools/tested_variants.py [nlev]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402

NX, NY = 1440, 720
ROUNDS, INNER = 7, 5


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(ROUNDS):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(INNER):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms.append(s.elapsed_time(e) / INNER)
    return float(np.median(ms))


def main():
    nlev = int(sys.argv[1]) if len(sys.argv) > 1 else 137
    dev = torch.device("cuda", 0)
    ctx = fc.Context(0)
    ctx.use_torch_stream()
    xm, ym, fcor = synth.grid_maps(NX, NY)
    dxm, dym, dfc = (torch.from_numpy(a).to(dev) for a in (xm, ym, fcor))
    u, v = synth.device_wind(NX, NY, nlev, 77, dev)
    z = u * 3.0 + 5500.0
    out, out2 = torch.empty_like(u), torch.empty_like(u)
    cnt = torch.zeros(nlev, dtype=torch.int64, device=dev)
    plev = np.linspace(1000.0, 100.0, nlev).astype(np.float32)
    n = NX * NY * nlev

    def ops(flags, counts):
        st = lambda op, f0, f1, fcc, two=False: (lambda: ctx.stencil_levels(op, f0, f1, dxm, dym, fcc, fdefined=flags, out0=out, out1=out2 if two else None))  # noqa: E731
        ex = lambda op, **kw: (lambda: ctx.stencil_levels_ex(op, fdefined=flags, out0=out, **kw))  # noqa: E731
        return [
            ("relvort+divergence (fused)", 16, lambda: ctx.vortdiv_levels_enqueue(u, v, dxm, dym, out, out2, fdefined=flags, n_undefined=counts)),
            ("relvort", 12, st("relvort", u, v, None)),
            ("divergence", 12, st("divergence", u, v, None)),
            ("absvort", 12, st("absvort", u, v, dfc)),
            ("gradient compute=1", 8, st("gradient1", z, None, None)),
            ("gradient compute=2", 8, st("gradient2", z, None, None)),
            ("gradient compute=3", 8, st("gradient3", z, None, None)),
            ("gradient compute=4", 8, st("gradient4", z, None, None)),
            ("plevelgwind_xcomp", 8, st("plevelgwind_xcomp", z, None, dfc)),
            ("plevelgvort", 8, st("plevelgvort", z, None, dfc)),
            ("ilevelgwind", 12, st("ilevelgwind", z, None, dfc, True)),
            ("jacobian", 12, st("jacobian", z, u, None)),
            ("advection", 16, ex("advection", f0=z, f1=u, f2=v, xmapr=dxm, ymapr=dym, scalar=1.0)),
            ("thermalFrontParameter", 8, ex("thermalFrontParameter", f0=z, xmapr=dxm, ymapr=dym)),
            ("plevelqvector c=2", 12, ex("plevelqvector", f0=z, f1=u, xmapr=dxm, ymapr=dym, fcoriolis=dfc, level_scalars=plev, compute=2)),
            ("shapiro2_filter", 8, ex("shapiro2_filter", f0=z)),
        ]

    all_flags = np.full(nlev, fc.ALL_DEFINED, np.int32)
    some_flags = np.full(nlev, fc.SOME_DEFINED, np.int32)
    print("%dx%dx%d, clean data; ms per call (median of %d x %d), %% of 8 TB/s on the algorithmic bytes" % (NX, NY, nlev, ROUNDS, INNER))
    print("%-30s %10s %8s %10s %8s %8s" % ("operator", "ALL ms", "%", "tested ms", "%", "ratio"))
    a_ops, s_ops = ops(all_flags, None), ops(some_flags, cnt)
    only = os.environ.get("BENCH_ONLY")  # substring of the operator name
    for (name, bpc, fa), (_, _, fs) in zip(a_ops, s_ops):
        if only and only not in name:
            continue
        try:
            ta = timed(lambda: fa())
            ts = timed(lambda: fs())
        except Exception as e:  # an operator name this build does not know: say so and go on
            print("%-30s failed: %s" % (name, e))
            continue
        print("%-30s %10.4f %8.1f %10.4f %8.1f %8.3f" % (name, ta, n * bpc / ta / 1e6 / 80.0, ts, n * bpc / ts / 1e6 / 80.0, ts / ta), flush=True)


if __name__ == "__main__":
    main()
