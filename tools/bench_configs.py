#!/usr/bin/env python3
"""Secondary measurements for the BASELINE.json configurations other than the
headline one (bench.py owns that).  One JSON line per configuration; kernel
times from HIP events on the launch stream, median of ROUNDS.

    python tools/bench_configs.py [c1] [c2] [c4] [c5]      (default: all)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402

PEAK = 8000.0
ROUNDS = int(os.environ.get("BENCH_ROUNDS", "7"))
DEV = torch.device("cuda", 0)


def timed(fn, inner):
    ms = []
    for _ in range(ROUNDS):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(inner):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms.append(s.elapsed_time(e) / inner)
    return float(np.median(ms)), float(np.min(ms))


def line(name, workload, cells, alg_bytes, med, mn, extra=None):
    out = {
        "config": name, "workload": workload, "cells_per_launch": cells, "ms_per_launch_median": round(med, 5), "ms_min": round(mn, 5),
        "Mcells_per_s": round(cells / med / 1e3, 1),
        "roofline": {"bound": "hbm", "algorithmic_bytes": alg_bytes, "achieved": round(alg_bytes / med / 1e6, 1), "peak": PEAK,
                     "unit": "GB/s", "frac": round(alg_bytes / med / 1e6 / PEAK, 4)},
    }
    if extra:
        out.update(extra)
    print(json.dumps(out), flush=True)


def c1(ctx):
    """256x256 vectorabs through the UNCHANGED host-pointer signature (plumbing, PCIe + sync inclusive)."""
    nx = ny = 256
    u, v = synth.wind(nx, ny, 1)
    out = np.empty_like(u)
    for _ in range(5):
        ctx.vectorabs(u, v, fdefined=fc.ALL_DEFINED, out=out)
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        ctx.vectorabs(u, v, fdefined=fc.ALL_DEFINED, out=out)
    dt = (time.perf_counter() - t0) / n * 1e3
    line("c1", "256x256 vectorabs via legacy host pointers (H2D + kernel + D2H + sync per call)", nx * ny, nx * ny * 12, dt, dt,
         {"note": "PCIe/launch-latency bound by construction; never the headline"})


def c2(ctx):
    """1440x720: fused (ff, rh, theta).  One level is launch-bound (33 MB): timed as back-to-back launches; plus the 137-level batch."""
    nx, ny = 1440, 720
    for nlev in (1, 137):
        u, v = synth.device_wind(nx, ny, nlev, 5, DEV)
        t, q, ps = synth.device_thermo(nx, ny, nlev, 6, DEV)
        a, b = synth.hybrid_levels(max(nlev, 2))
        a, b = a[:nlev], b[:nlev]
        ff, rh, th = (torch.empty_like(u) for _ in range(3))
        cnt = torch.zeros(3 * nlev, dtype=torch.int64, device=DEV)
        flags = np.full(nlev, fc.ALL_DEFINED, np.int32)

        def run():
            assert ctx.hlevel_derived_levels_enqueue(u, v, t, q, ps, a, b, ff, rh, th, cnt, fdef_wind=flags, fdef_thermo=flags)

        for _ in range(3):
            run()
        torch.cuda.synchronize()
        med, mn = timed(run, 50 if nlev == 1 else 5)
        cells = nx * ny * nlev
        alg = cells * 28 + nx * ny * 4  # u,v,t,q in + ff,rh,theta out per cell, ps once
        extra = {"note": "single level: launch/flag-upload bound" if nlev == 1 else "levels batched"}
        if nlev == 1:
            # the 33 MB of one level fit the 256 MB Infinity Cache: also time it rotating over 20 buffer sets (660 MB)
            sets = [tuple(x.clone() for x in (u, v, t, q, ff, rh, th)) for _ in range(20)]
            state = {"k": 0}

            def run_cold():
                uu, vv, tt, qq, f1, r1, t1 = sets[state["k"] % len(sets)]
                state["k"] += 1
                assert ctx.hlevel_derived_levels_enqueue(uu, vv, tt, qq, ps, a, b, f1, r1, t1, cnt, fdef_wind=flags, fdef_thermo=flags)

            for _ in range(20):
                run_cold()
            torch.cuda.synchronize()
            cmed, cmn = timed(run_cold, 60)
            extra.update({"cold_ms_median": round(cmed, 5), "cold_frac": round(alg / cmed / 1e6 / PEAK, 4),
                          "cold_note": "rotating over 20 buffer sets (660 MB) so that the Infinity Cache cannot hold the working set"})
            del sets
        line("c2" if nlev == 1 else "c2x137", "1440x720x%d fused ff + RH(%%) + theta (hybrid levels), ALL_DEFINED, device resident" % nlev, cells, alg, med, mn,
             extra)


def c4(ctx):
    """4000x4000: whole field on one GPU, and one of 8 row slabs (500 rows + halo rows) as each rank of an 8-GPU run would execute it."""
    nx = ny = 4000
    xm, ym, _ = synth.grid_maps(nx, ny, h=2500.0)
    dxm, dym = torch.from_numpy(xm).to(DEV), torch.from_numpy(ym).to(DEV)
    u, v = synth.device_wind(nx, ny, 1, 9, DEV)
    rv, dg = torch.empty_like(u), torch.empty_like(u)
    flags = np.full(1, fc.ALL_DEFINED, np.int32)

    def whole():
        assert ctx.vortdiv_levels_enqueue(u, v, dxm, dym, rv, dg, fdefined=flags)

    for _ in range(3):
        whole()
    torch.cuda.synchronize()
    med, mn = timed(whole, 20)
    cells = nx * ny
    # 384 MB per launch already exceed the 256 MB Infinity Cache; three rotating buffer sets (1.15 GB) make sure
    sets = [(u.clone(), v.clone(), dxm.clone(), dym.clone(), torch.empty_like(u), torch.empty_like(u)) for _ in range(3)]
    state = {"k": 0}

    def whole_cold():
        uu, vv, xx, yy, r1, d1 = sets[state["k"] % 3]
        state["k"] += 1
        assert ctx.vortdiv_levels_enqueue(uu, vv, xx, yy, r1, d1, fdefined=flags)

    for _ in range(3):
        whole_cold()
    torch.cuda.synchronize()
    cmed, cmn = timed(whole_cold, 21)
    del sets
    line("c4-whole", "4000x4000 single level fused relvort+divergence on ONE GPU", cells, cells * 16 + 2 * cells * 4, med, mn,
         {"cold_ms_median": round(cmed, 5), "cold_frac": round((cells * 24) / cmed / 1e6 / PEAK, 4), "cold_note": "rotating over 3 buffer sets (1.15 GB)"})
    from mi_fieldcalc_amd.sharding import slab_rows

    j0, nloc = slab_rows(ny, 8, 3)
    uh = torch.empty((nloc + 2, nx), dtype=torch.float32, device=DEV).copy_(u[0, j0 - 1:j0 + nloc + 1])
    vh = torch.empty((nloc + 2, nx), dtype=torch.float32, device=DEV).copy_(v[0, j0 - 1:j0 + nloc + 1])
    sx, sy = dxm[j0:j0 + nloc].contiguous(), dym[j0:j0 + nloc].contiguous()
    orv = torch.empty((nloc, nx), dtype=torch.float32, device=DEV)
    odg = torch.empty_like(orv)

    def slab():
        assert ctx.vortdiv_slab_enqueue(nx, ny, j0, nloc, uh, vh, sx, sy, orv, odg, fdefined_in=fc.ALL_DEFINED)

    for _ in range(3):
        slab()
    torch.cuda.synchronize()
    med, mn = timed(slab, 50)
    cells = nx * nloc
    line("c4-slab", "one of 8 row slabs (500+2 rows x 4000) of the 4000x4000 field, kernel only (halo exchange not included)", cells,
         cells * 16 + 2 * cells * 4, med, mn, {"note": "8 MB per slab: launch-latency bound; halo = 2 rows x 16 kB per field"})


def c5(ctx, members):
    """Ensemble pipeline on one GPU: per member (137 levels) derived (ff, rh, theta) + fused vorticity/divergence."""
    nx, ny, nlev = 1440, 720, 137
    xm, ym, _ = synth.grid_maps(nx, ny)
    dxm, dym = torch.from_numpy(xm).to(DEV), torch.from_numpy(ym).to(DEV)
    a, b = synth.hybrid_levels(nlev)
    flags = np.full(nlev, fc.ALL_DEFINED, np.int32)
    data = []
    for m in range(members):
        u, v = synth.device_wind(nx, ny, nlev, 100 + m, DEV)
        t, q, ps = synth.device_thermo(nx, ny, nlev, 200 + m, DEV)
        data.append((u, v, t, q, ps, [torch.empty_like(u) for _ in range(5)], torch.zeros(3 * nlev, dtype=torch.int64, device=DEV)))

    def run():
        for (u, v, t, q, ps, o, cnt) in data:
            assert ctx.hlevel_derived_levels_enqueue(u, v, t, q, ps, a, b, o[0], o[1], o[2], cnt, fdef_wind=flags, fdef_thermo=flags)
            assert ctx.vortdiv_levels_enqueue(u, v, dxm, dym, o[3], o[4], fdefined=flags)

    run()
    torch.cuda.synchronize()
    med, mn = timed(run, 2)
    cells = nx * ny * nlev * members
    alg = members * (nx * ny * nlev * (28 + 16) + 3 * nx * ny * 4)
    line("c5-1gpu", "%d members x 137 levels x 1440x720: (ff, rh, theta) + (relvort, divergence) per member, one GPU" % members, cells, alg, med, mn,
         {"members": members, "note": "members are independent: N GPUs take members/N each, no collective"})


def main():
    which = [a for a in sys.argv[1:] if not a.startswith("-")] or ["c1", "c2", "c4", "c5"]
    ctx = fc.Context(0)
    if "c1" in which:
        c1(ctx)
    if "c2" in which:
        c2(ctx)
    if "c4" in which:
        c4(ctx)
    if "c5" in which:
        c5(ctx, int(os.environ.get("BENCH_MEMBERS", "4")))


if __name__ == "__main__":
    main()
