#!/usr/bin/env python3
"""A/B of the split-role level-walking forms (round 3) against the forms whose waves load AND store, for every stencil
operator of SURVEY.md 8a over a device-resident level batch: interleaved rounds in ONE process on ONE set of arrays,
kernel ms from HIP events around the launches.  Optional shape sweep of the one-input split-role kernel.

    python tools/ab_split_ops.py [nx,ny,nlev] [--shapes "TR=12,NL=2,PF=2;TR=14,NL=2,PF=2"] [--tested]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402

# (operator, needs fcoriolis, two inputs, two outputs, algorithmic bytes per cell)
OPS = [("vortdiv", False, True, True, 16), ("relvort", False, True, False, 12), ("divergence", False, True, False, 12), ("absvort", True, True, False, 12),
       ("gradient1", False, False, False, 8), ("gradient2", False, False, False, 8), ("gradient3", False, False, False, 8), ("gradient4", False, False, False, 8),
       ("plevelgwind_xcomp", True, False, False, 8), ("plevelgwind_ycomp", True, False, False, 8), ("plevelgvort", True, False, False, 8),
       ("ilevelgwind", True, False, True, 12)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape", nargs="?", default="1440,720,137")
    ap.add_argument("--shapes", default="")
    ap.add_argument("--tested", action="store_true", help="input flags SOME_DEFINED (clean data): the variants with tests and counts")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--ops", default="")
    ap.add_argument("--placed", action="store_true", help="u, v and the outputs chosen from a pool by the placement search bench.py uses")
    ap.add_argument("--burst", type=int, default=10, help="launches back to back between one pair of events")
    args = ap.parse_args()
    nx, ny, nlev = (int(x) for x in args.shape.split(","))
    dev = torch.device("cuda", 0)
    ctx = fc.Context(0)
    xm, ym, fcor = synth.grid_maps(nx, ny)
    dxm, dym, dfc = (torch.from_numpy(a).to(dev) for a in (xm, ym, fcor))
    u, v = synth.device_wind(nx, ny, nlev, 99, dev)
    out0, out1 = torch.empty_like(u), torch.empty_like(u)
    if args.placed:
        # the four arrays chosen like bench.py chooses the headline batch (mi-fieldcalc_amd/placement.py), the fused kernel as the probe
        from mi_fieldcalc_amd.placement import choose_search_rounds

        pflags = np.full(nlev, fc.ALL_DEFINED, np.int32)
        ctx.use_torch_stream()

        def probe(arrays):
            a, b, c, d = arrays
            ms = []
            for k in range(4):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(4):
                    ctx.vortdiv_levels_enqueue(a, b, dxm, dym, c, d, fdefined=pflags, n_undefined=None)
                e.record()
                torch.cuda.synchronize()
                if k:
                    ms.append(s.elapsed_time(e) / 4)
            return float(np.median(ms))

        su, sv = u, v
        del out0, out1
        (u, v, out0, out1), rep = choose_search_rounds(lambda: torch.empty((nlev, ny, nx), dtype=torch.float32, device=dev), 4, probe, rounds=2, pool_size=48,
                                                       random_sets=48, max_probes=200, device=dev)
        u.copy_(su)
        v.copy_(sv)
        del su, sv
        torch.cuda.empty_cache()
        print("placed arrays: fused pair %.4f ms on the chosen set, %.4f as allocated" % (rep["chosen_reprobed_ms"], rep["allocated_in_one_go_ms"]))
    flags = np.full(nlev, fc.SOME_DEFINED if args.tested else fc.ALL_DEFINED, np.int32)
    counts = torch.zeros(nlev, dtype=torch.int64, device=dev)
    ctx.use_torch_stream()
    # "pf1": the wind operators' split-role kernel with the loaders ONE level ahead (two level buffers: two workgroups per CU)
    # "k4": the wind operators' split-role kernel two levels ahead, forced also where the default keeps the first level-walking form (single outputs)
    modes = [("old", {"MIFC_VORTDIV_SPLIT": "0"}), ("split", {}), ("pf1", {"MIFC_VORTDIV_TUNE": "K=4,RB=12,D=0,WPB=2,LG=6"}),
             ("k4", {"MIFC_VORTDIV_TUNE": "K=4,RB=12,D=1,WPB=2,LG=6"})]
    for sh in [s for s in args.shapes.split(";") if s]:
        modes.append((sh, {"MIFC_SCALAR_SPLIT_TUNE": sh}))
    keys = sorted({k for _, e in modes for k in e})
    print("%dx%dx%d, %s; ms per launch, %d launches back to back between one pair of HIP events (median of %d interleaved rounds); %% of 8 TB/s on the algorithmic bytes"
          % (nx, ny, nlev, "tested (SOME_DEFINED, clean data)" if args.tested else "ALL_DEFINED", args.burst, args.rounds))
    print("%-20s" % "operator" + "".join(" %22s" % m for m, _ in modes))
    for op, use_fc, two_in, two_out, bpc in OPS:
        if args.ops and op not in args.ops.split(","):
            continue
        scalar = not two_in

        def select(envs):
            for k in keys:
                os.environ.pop(k, None)
            os.environ.update(envs)
            ctx.reload_env()

        def run():
            if not ctx.stencil_levels_enqueue(op, u, v if two_in else None, dxm, dym, dfc if use_fc else None, out0, out1 if two_out else None,
                                              fdefined=flags, n_undefined=counts if args.tested else None):
                raise RuntimeError(ctx.last_error())

        res = {m: [] for m, _ in modes}
        use = [(m, e) for m, e in modes if (scalar and m not in ("pf1", "k4")) or (not scalar and m in ("old", "split", "pf1", "k4"))]
        for m, e in use:
            select(e)
            for _ in range(args.burst):
                run()
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for m, e in use:
                select(e)
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(3):
                    run()  # the clocks are up and the previous mode's tail is gone when the timed burst starts
                ev0.record()
                for _ in range(args.burst):
                    run()
                ev1.record()
                torch.cuda.synchronize()
                res[m].append(ev0.elapsed_time(ev1) / args.burst)
        alg = nx * ny * (nlev * bpc + (12 if use_fc else 8))
        line = "%-20s" % op
        for m, _ in modes:
            if res[m]:
                t = float(np.median(res[m]))
                line += " %12.4f %8.1f%%" % (t, alg / t / 1e6 / 8000 * 100)
            else:
                line += " %22s" % "-"
        print(line, flush=True)
    for k in keys:
        os.environ.pop(k, None)
    ctx.reload_env()


if __name__ == "__main__":
    main()
