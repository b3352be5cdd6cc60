#!/usr/bin/env python3
"""N-rank drivers of the two multi-GPU configurations of BASELINE.json, one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P tools/bench_multigpu.py --config 4|5 [--steps K --warmup W] [--check] [--dump DIR]

config 4  one 4000x4000 field split into N row slabs (STRONG scaling: the field is fixed).
          A step = RCCL halo exchange of u and v (one row per neighbour and field, over xGMI)
          overlapped with the interior rows of the slab kernel, then the two boundary strips,
          then -- in the tested variant -- an 8-byte all-reduce of the undefined counts.
          --check: rank 0 also computes the whole field on its own GPU and every slab is
          compared with it bit for bit (values and the global undefined count).
config 5  51 ensemble members x 137 levels x 1440x720, members sharded over the ranks
          (STRONG scaling: 51 members in total, no data-path collective).  Per member: the
          fused derived batch (ff, RH, theta; mifc_hlevel_derived_levels) and the fused
          vorticity+divergence batch.  Every rank rotates over two resident member-sized
          buffer sets (10 GB) so that no pass is served from the 256 MB Infinity Cache.
          --check: rank r compares sampled levels of its first member between the
          batched kernels and the single-field entry points (independent kernels).

Rank 0 prints ONE JSON line (value = whole-job Mcells/s, max over ranks of the
barrier-bracketed wall time).  MIFC_BENCH_BACKEND=gloo rehearses the control flow on a
one-GPU box (ranks share cuda:0, halo rows staged through the host)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def setup():
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("MIFC_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return world, rank, dev_index, dev, backend


def max_over_ranks(x, dev, backend):
    import torch
    import torch.distributed as dist

    t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_true(ok, dev, backend):
    import torch
    import torch.distributed as dist

    t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def timed_steps(step, steps, warmup):
    import torch
    import torch.distributed as dist

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    return time.perf_counter() - t0


# ------------------------------------------------------------------ config 4
def config4(args):
    import numpy as np
    import torch
    import torch.distributed as dist

    import mi_fieldcalc_amd as fc
    import mi_fieldcalc_amd.synth as synth
    from mi_fieldcalc_amd.sharding import begin_halo_exchange, global_undefined_count, slab_rows, vortdiv_slab_overlapped

    world, rank, dev_index, dev, backend = setup()
    nx = ny = args.size
    nlev = args.levels
    j0, nloc = slab_rows(ny, world, rank)
    # the whole field is generated on every rank from the seed (closed-form waves + seeded noise) and cut:
    # what a rank keeps resident is its slab (+ halo rows, filled by the exchange)
    xm, ym, _ = synth.grid_maps(nx, ny, h=2500.0)
    tested = not args.all_defined
    flag = fc.SOME_DEFINED if tested else fc.ALL_DEFINED
    fields = []
    for l in range(nlev):
        u, v = synth.wind(nx, ny, 0x5EED0000 + 4000 + l)
        if tested:
            u = synth.sprinkle_undef(u, 41 + l, 0.001)
        fields.append((u, v))

    def slab_with_halo(k):
        t = torch.zeros((nlev, nloc + 2, nx), dtype=torch.float32, device=dev)
        for l in range(nlev):
            t[l, 1:-1] = torch.from_numpy(fields[l][k][j0:j0 + nloc]).to(dev)
        return t

    uh, vh = slab_with_halo(0), slab_with_halo(1)
    dxm, dym = torch.from_numpy(np.ascontiguousarray(xm[j0:j0 + nloc])).to(dev), torch.from_numpy(np.ascontiguousarray(ym[j0:j0 + nloc])).to(dev)
    rv = torch.empty((nlev, nloc, nx), dtype=torch.float32, device=dev)
    dg = torch.empty_like(rv)
    cnt = torch.zeros(nlev, dtype=torch.int64, device=dev)
    ctx = fc.Context(dev_index)
    ctx.use_torch_stream()

    # The step is ONE call of the C ABI (mifc_slab_plan_step: RCCL exchange from C++, interior rows meanwhile, strips,
    # count all-reduce; a HIP graph replayed per step).  --legacy-step keeps round 2's Python-orchestrated sequence
    # (single level only) for A/B; the gloo rehearsal brackets a host relay of the rows with begin() / finish().
    plan = None
    if args.legacy_step:
        if nlev != 1:
            raise SystemExit("--legacy-step is the single-level sequence of round 2")

        def step():
            if not vortdiv_slab_overlapped(ctx, nx, ny, j0, nloc, uh[0], vh[0], dxm, dym, rv[0], dg[0], rank, world, fdefined_in=flag,
                                           n_undefined=cnt if tested else None):
                raise RuntimeError(ctx.last_error())
            if tested:
                if backend == "nccl":
                    global_undefined_count(cnt)
                else:
                    c = cnt.cpu()
                    global_undefined_count(c)
                    cnt.copy_(c)
    else:
        plan = ctx.slab_plan(nx, ny, j0, nloc, uh, vh, dxm, dym, rv, dg, fdefined_in=flag, n_undefined=cnt if tested else None)
        if backend == "nccl":
            if world > 1:
                ctx.comm_init_from_torch()
            step = plan.step
        else:
            def step():
                plan.begin()
                begin_halo_exchange([uh[l] for l in range(nlev)] + [vh[l] for l in range(nlev)], rank, world).wait()
                plan.finish()
                if tested:
                    c = cnt.cpu()
                    global_undefined_count(c)
                    cnt.copy_(c)

    my_wall = timed_steps(step, args.steps, args.warmup)
    wall = max_over_ranks(my_wall, dev, backend)
    # this rank's share of the algorithmic traffic per step: 16 B per owned cell and level + its rows of the map factors once
    my_bytes = nlev * nloc * nx * 16 + 2 * nloc * nx * 4
    my_frac = my_bytes / (my_wall / args.steps) / 8e12
    fr = torch.tensor([my_frac, -my_frac], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(fr, op=dist.ReduceOp.MAX)
    frac_max, frac_min = float(fr[0]), -float(fr[1])

    out = {
        "metric": "Mcells/s fused vorticity+divergence, %d level(s) of one %dx%d field in %d row slabs (BASELINE.json configs[3])" % (nlev, nx, ny, world),
        "value": round(nlev * nx * ny * args.steps / wall / 1e6, 1), "unit": "Mcells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "dtype": "f32 in/out, f64 combine",
        "data": "synthetic",
        "roofline": {"bound": "hbm", "achieved": round((nlev * nx * ny * 16 + 2 * nx * ny * 4) / (wall / args.steps) / 1e9, 1), "peak": 8000.0 * world,
                     "unit": "GB/s", "frac": round((nlev * nx * ny * 16 + 2 * nx * ny * 4) / (wall / args.steps) / 8e12 / world, 4),
                     "per_rank_frac_min": round(frac_min, 4), "per_rank_frac_max": round(frac_max, 4), "traffic": None,
                     "note": "whole step on the wall clock (exchange, launches and reduce included), not kernel time"},
        "config": {"workload": "%d level(s) of %dx%d float32, row slabs of %d..%d rows, 1-row halo of u and v per level and neighbour exchanged each step "
                               "(RCCL send/recv, overlapped with the interior rows), %s" % (
                                   nlev, nx, ny, ny // world, -(-ny // world), "per-cell undefined tests + count all-reduce" if tested else "ALL_DEFINED inputs"),
                   "step": "round-2 Python sequence" if args.legacy_step else ("mifc_slab_plan_step, HIP graph replay" if plan.uses_graph else
                                                                               ("mifc_slab_plan_step, direct" if backend == "nccl" else "mifc_slab_plan_begin / host relay / finish")),
                   "backend": backend},
    }
    if args.check:
        # one more decomposed pass, then every slab against rank 0's whole-field result, bit for bit
        step()
        torch.cuda.synchronize()
        total = cnt.clone()  # every form of the step leaves the whole field's counts on every rank
        ok = True
        whole = None
        if rank == 0:
            du = torch.from_numpy(np.stack([f[0] for f in fields])).to(dev)
            dv_ = torch.from_numpy(np.stack([f[1] for f in fields])).to(dev)
            fx, fy = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
            (wrv, wdg), wflag = ctx.vortdiv_levels(du, dv_, fx, fy, fdefined=[flag] * nlev)
            whole = (wrv, wdg, [int(f) for f in wflag])
            ctx.use_torch_stream()
        # gather the slabs on rank 0 (through the host: this is the checker, not the timed path)
        parts = [None] * world if rank == 0 else None
        dist.gather_object((j0, nloc, rv.cpu().numpy(), dg.cpu().numpy()), parts, dst=0)
        if rank == 0:
            got_rv = np.empty((nlev, ny, nx), np.float32)
            got_dg = np.empty((nlev, ny, nx), np.float32)
            for pj0, pn, prv, pdg in parts:
                got_rv[:, pj0:pj0 + pn], got_dg[:, pj0:pj0 + pn] = prv, pdg
            wrv_h, wdg_h = whole[0].cpu().numpy(), whole[1].cpu().numpy()
            same = lambda a, b: bool(np.array_equal(a.view(np.uint32)[~np.isnan(a)], b.view(np.uint32)[~np.isnan(b)]) and np.array_equal(np.isnan(a), np.isnan(b)))
            counts = [int(x) for x in total.cpu().numpy()]
            got_flags = [fc.ALL_DEFINED if not tested else fc.classify(c, nx * ny - 2 * nx) for c in counts]
            ok = same(got_rv, wrv_h) and same(got_dg, wdg_h) and got_flags == whole[2]
            out["verified"] = ok
            out["check"] = "every slab == rank 0's whole-field result bit for bit; global undefined counts %s -> flags %s" % (counts[:4], got_flags[:4])
            if args.dump:
                os.makedirs(args.dump, exist_ok=True)
                np.save(os.path.join(args.dump, "config4_rvort.npy"), got_rv[0])
                np.save(os.path.join(args.dump, "config4_diverg.npy"), got_dg[0])
                with open(os.path.join(args.dump, "config4_meta.json"), "w") as f:
                    json.dump({"nx": nx, "ny": ny, "seed": 0x5EED0000 + 4000, "tested": tested, "flag": got_flags[0], "count": counts[0]}, f)
        ok = all_true(ok, dev, backend)
        if rank == 0 and not ok:
            out["verified"] = False
    if plan is not None:
        plan.close()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)
        if args.check and not out.get("verified", False):
            sys.exit(1)


# ------------------------------------------------------------------ config 5
def config5(args):
    import numpy as np
    import torch
    import torch.distributed as dist

    import mi_fieldcalc_amd as fc
    import mi_fieldcalc_amd.synth as synth
    from mi_fieldcalc_amd.sharding import shard_range

    world, rank, dev_index, dev, backend = setup()
    nx, ny, nlev, nmem = 1440, 720, args.nlev, args.members
    m0, m1 = shard_range(nmem, world, rank)
    ctx = fc.Context(dev_index)
    ctx.use_torch_stream()
    xm, ym, _ = synth.grid_maps(nx, ny)
    dxm, dym = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
    al, bl = synth.hybrid_levels(nlev)
    flags = np.full(nlev, fc.ALL_DEFINED, np.int32)
    nsets = 2 if backend == "nccl" or world == 1 else 1  # gloo rehearsal: the ranks share one GPU's memory
    sets = []
    for s in range(nsets):
        u, v = synth.device_wind(nx, ny, nlev, 0x5EED0000 + 5000 + 97 * rank + s, dev)
        t, q, ps = synth.device_thermo(nx, ny, nlev, 0x5EED0000 + 5500 + 97 * rank + s, dev)
        outs = [torch.empty_like(u) for _ in range(5)]
        sets.append(dict(u=u, v=v, t=t, q=q, ps=ps, ff=outs[0], rh=outs[1], th=outs[2], rv=outs[3], dg=outs[4],
                         cnt=torch.zeros(3 * nlev, dtype=torch.int64, device=dev)))

    def member_pass(k):
        s = sets[k % nsets]
        if args.unfused_ff:  # round 2: ff in the derived launch (reads u, v), the stencil pair in a second one (reads u, v again): 44 B/cell
            ok = ctx.hlevel_derived_levels_enqueue(s["u"], s["v"], s["t"], s["q"], s["ps"], al, bl, s["ff"], s["rh"], s["th"], s["cnt"],
                                                   fdef_wind=flags, fdef_thermo=flags)
            ok = ok and ctx.vortdiv_levels_enqueue(s["u"], s["v"], dxm, dym, s["rv"], s["dg"], fdefined=flags)
        else:  # ff rides on the fused vorticity + divergence kernel (u, v read once: 36 B/cell); RH and theta from t, q, ps
            ok = ctx.hlevel_derived_levels_enqueue(s["u"], s["v"], s["t"], s["q"], s["ps"], al, bl, None, s["rh"], s["th"], s["cnt"],
                                                   fdef_wind=flags, fdef_thermo=flags)
            ok = ok and ctx.vortdiv_ff_levels_enqueue(s["u"], s["v"], dxm, dym, s["rv"], s["dg"], s["ff"], fdefined=flags)
        if not ok:
            raise RuntimeError(ctx.last_error())

    def step():  # one pass over this rank's share of the ensemble
        for k in range(m1 - m0):
            member_pass(k)

    wall = max_over_ranks(timed_steps(step, args.steps, args.warmup), dev, backend)
    cells = nx * ny * nlev * nmem
    out = {
        "metric": "Mcells/s derived (ff, RH, theta) + vorticity/divergence pipeline, %d members x 1440x720x%d (BASELINE.json configs[4])" % (nmem, nlev),
        "value": round(cells * args.steps / wall / 1e6, 1), "unit": "Mcells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "dtype": "f32 (f64 combine / x^kappa)",
        "data": "synthetic",
        "config": {"workload": ("%d members sharded over %d ranks (%d on the busiest), per member 2 launches over %d levels: " % (nmem, world, -(-nmem // world), nlev))
                   + ("fused ff+RH+theta (28 B/cell) and fused relvort+divergence (16 B/cell)" if args.unfused_ff else
                      "fused RH+theta (16 B/cell) and fused relvort+divergence+ff (20 B/cell: u and v are read once per member)")
                   + "; inputs ALL_DEFINED, resident in HBM, two rotating member-sized buffer sets per rank",
                   "algorithmic_bytes_per_member": nx * ny * nlev * (44 if args.unfused_ff else 36) + 3 * nx * ny * 4, "backend": backend},
    }
    bytes_member = out["config"]["algorithmic_bytes_per_member"]
    out["roofline"] = {"bound": "hbm", "achieved": round(bytes_member * nmem * args.steps / wall / 1e9, 1), "peak": 8000.0 * world, "unit": "GB/s",
                       "frac": round(bytes_member * nmem * args.steps / wall / 8e12 / world, 4), "traffic": None,
                       "note": "whole pipeline on the wall clock, max over ranks"}
    if args.check:
        # sampled levels of this rank's first member against the CPU reference path (oracle/_ref when built, else the
        # restatement) -- the checker, outside every timed region: the stencil outputs and ff bit for bit, RH and theta
        # within 1e-5 relative (BASELINE.json; a libm power per cell)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import cpulib

        which = "ref" if cpulib.available("ref") else "oracle"
        cpu = cpulib.CpuLib(which)
        s = sets[0]
        member_pass(0)
        torch.cuda.synchronize()
        ok = True
        h = lambda x: x.cpu().numpy()  # noqa: E731
        bits = lambda a, b: bool(np.array_equal(a.view(np.uint32), b.view(np.uint32)))  # noqa: E731
        close = lambda a, b: bool(np.all(np.abs(a.astype(np.float64) - b) <= 1e-5 * np.abs(b.astype(np.float64)) + 1e-30))  # noqa: E731
        ps_h = h(s["ps"])
        worst = 0.0
        for l in sorted({0, nlev // 2, nlev - 1}):
            ul, vl, tl, ql = h(s["u"][l]), h(s["v"][l]), h(s["t"][l]), h(s["q"][l])
            _, e, _ = cpu.call("relvort", nx, ny, ul, vl, xm, ym, fdefined=fc.ALL_DEFINED)
            ok = ok and bits(h(s["rv"][l]), e)
            _, e, _ = cpu.call("divergence", nx, ny, ul, vl, xm, ym, fdefined=fc.ALL_DEFINED)
            ok = ok and bits(h(s["dg"][l]), e)
            _, e, _ = cpu.call("vectorabs", nx, ny, ul, vl, fdefined=fc.ALL_DEFINED)
            ok = ok and bits(h(s["ff"][l]), e)
            _, e, _ = cpu.call("hlevelhum", nx, ny, tl, ql, ps_h, float(al[l]), float(bl[l]), "", 1, fdefined=fc.ALL_DEFINED)
            ok = ok and close(h(s["rh"][l]), e)
            worst = max(worst, float(np.max(np.abs(h(s["rh"][l]).astype(np.float64) - e) / np.maximum(np.abs(e), 1e-30))))
            _, e, _ = cpu.call("hleveltemp", nx, ny, tl, ps_h, float(al[l]), float(bl[l]), "", 3, fdefined=fc.ALL_DEFINED)
            ok = ok and close(h(s["th"][l]), e)
            worst = max(worst, float(np.max(np.abs(h(s["th"][l]).astype(np.float64) - e) / np.maximum(np.abs(e), 1e-30))))
        ctx.use_torch_stream()
        ok = all_true(ok, dev, backend)
        out["verified"] = ok
        out["check"] = ("levels 0 / mid / last of every rank's first member against the %s CPU path: relvort, divergence, ff bit for bit; RH, theta within "
                        "1e-5 relative (rank 0 worst %.2e)" % ("reference" if which == "ref" else "restated", worst))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)
        if args.check and not out.get("verified", False):
            sys.exit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, choices=(4, 5), required=True)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--dump", default=None, help="config 4 + --check: directory for the assembled result (rank 0)")
    ap.add_argument("--size", type=int, default=4000, help="config 4: the field is size x size")
    ap.add_argument("--all-defined", action="store_true", help="config 4: ALL_DEFINED inputs (no tests, no count all-reduce)")
    ap.add_argument("--levels", type=int, default=1, help="config 4: levels in the slab batch (one exchange of that many rows per neighbour and field)")
    ap.add_argument("--legacy-step", action="store_true", help="config 4: round 2's Python-orchestrated step (A/B)")
    ap.add_argument("--unfused-ff", action="store_true", help="config 5: round 2's pipeline, ff in the derived launch (A/B: 44 instead of 36 B/cell)")
    ap.add_argument("--members", type=int, default=51, help="config 5")
    ap.add_argument("--nlev", type=int, default=137, help="config 5")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 200 if args.config == 4 else 3
    (config4 if args.config == 4 else config5)(args)


if __name__ == "__main__":
    main()
