"""Legacy host-pointer path, end to end including the host <-> HBM
copies: what an existing caller of the unchanged signatures gets.

  python tools/bench_hostpath.py [--nlev 137] [--reps 3]

Prints one JSON line per case.  Values are PCIe-inclusive and never the headline
`value` of bench.py (which starts with the inputs resident in HBM)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402


def timed(fn, reps):
    fn()  # warm-up: scratch allocation, page faults of the outputs
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nlev", type=int, default=137)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--only-batched", action="store_true", help="just the pipelined batch (knob sweeps)")
    a = ap.parse_args()
    nx, ny, nlev = 1440, 720, a.nlev
    ctx = fc.Context(0)
    xm, ym, fcor = synth.grid_maps(nx, ny)
    u, v = synth.wind(nx, ny, 0x5EED0000, nlev=nlev)
    rv = np.empty_like(u)
    dv = np.empty_like(u)
    cells = nx * ny * nlev

    def batched():
        res = ctx.vortdiv_levels(u, v, xm, ym, fdefined=[fc.ALL_DEFINED] * nlev, rvort=rv, diverg=dv)
        assert res is not None

    for pipe in ("1",) if a.only_batched else ("1", "0"):
        os.environ["MIFC_HOST_PIPELINE"] = pipe
        ctx.reload_env()
        dt = timed(batched, a.reps)
        print(json.dumps({"case": "vortdiv_levels host pointers, " + ("chunked full-duplex pipeline" if pipe == "1" else "whole batch staged"),
                          "nlev": nlev, "ms": 1e3 * dt, "mcells_per_s": cells / dt / 1e6, "link_GBps_each_way": cells * 8 / dt / 1e9}), flush=True)
    os.environ["MIFC_HOST_PIPELINE"] = "1"
    ctx.reload_env()
    if a.only_batched:
        return

    # BASELINE.json config 2 x nlev from host memory: fused ff + RH + theta
    t, q, ps = synth.thermo(nx, ny, 0x5EED0000 + 1, nlev=nlev)
    al, bl = synth.hybrid_levels(nlev)
    outs = {"ff": np.empty_like(u), "rh": np.empty_like(u), "theta": np.empty_like(u)}
    allf = [fc.ALL_DEFINED] * nlev

    def derived():
        assert ctx.hlevel_derived_levels(u, v, t, q, ps, al, bl, fdef_wind=allf, fdef_thermo=allf, out=outs) is not None

    for pipe in ("1", "0"):
        os.environ["MIFC_HOST_PIPELINE"] = pipe
        ctx.reload_env()
        dt = timed(derived, a.reps)
        print(json.dumps({"case": "hlevel_derived_levels (ff, RH, theta) host pointers, " + ("chunked full-duplex pipeline" if pipe == "1" else "whole batch staged"),
                          "nlev": nlev, "ms": 1e3 * dt, "mcells_per_s": cells / dt / 1e6, "in_GBps": cells * 16 / dt / 1e9, "out_GBps": cells * 12 / dt / 1e9}),
              flush=True)
    os.environ["MIFC_HOST_PIPELINE"] = "1"
    ctx.reload_env()
    del t, q, outs

    nl = min(nlev, 16)

    def per_field():
        for l in range(nl):
            r1 = ctx.relvort(u[l], v[l], xm, ym, fdefined=fc.ALL_DEFINED, out=rv[l])
            r2 = ctx.divergence(u[l], v[l], xm, ym, fdefined=fc.ALL_DEFINED, out=dv[l])
            assert r1 is not None and r2 is not None

    dt = timed(per_field, a.reps)
    print(json.dumps({"case": "relvort + divergence per level, host pointers (legacy call pattern)", "nlev": nl, "ms": 1e3 * dt,
                      "mcells_per_s": nx * ny * nl / dt / 1e6}), flush=True)

    ctx.hold_field(xm)
    ctx.hold_field(ym)
    dt = timed(per_field, a.reps)
    print(json.dumps({"case": "relvort + divergence per level, host pointers, map ratios held on the device", "nlev": nl, "ms": 1e3 * dt,
                      "mcells_per_s": nx * ny * nl / dt / 1e6}), flush=True)
    ctx.release_field(xm)
    ctx.release_field(ym)

    u1, v1 = synth.wind(256, 256, 7)
    ff = np.empty_like(u1)

    def c1():
        assert ctx.vectorabs(u1, v1, fdefined=fc.ALL_DEFINED, out=ff) is not None

    dt = timed(c1, 20)
    print(json.dumps({"case": "C1 vectorabs 256x256 host pointers", "us": 1e6 * dt, "mcells_per_s": 65536 / dt / 1e6}), flush=True)


if __name__ == "__main__":
    main()
