#!/usr/bin/env python3
"""Headline benchmark: Mcells/s of fused relative vorticity + divergence on a
1440x720x137 float32 grid (BASELINE.json), one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the fused kernel over one batch of 137 levels (one
ensemble member) that is already resident in HBM.  With N GPUs every rank owns
one such member (members are independent: no data-path collective, weak
scaling); `value` = cells processed by all ranks / max-over-ranks wall time.

The JSON line also carries
  roofline     : algorithmic bytes (16 B/cell + map factors once per batch)
                 over the kernel's average duration measured with HIP events
                 on the launch stream, against the 8 TB/s HBM3E peak;
  cpu_baseline : the reference CPU path (oracle/_ref, compiled from the
                 reference's own sources; or the bit-exact restatement if that
                 library is not present) calling relvort then divergence per
                 level on this host, on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NX, NY, NLEV = 1440, 720, 137
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
SEED = 0x5EED0000 + 3000


def algorithmic_bytes(nx, ny, nlev):
    # read u,v + write rvort,diverg = 16 B/cell, xmapr+ymapr once per batch
    return nx * ny * nlev * 16 + 2 * nx * ny * 4


def cpu_baseline(seconds_target=10.0):
    """Reference CPU path timed on this host.  The oracle / reference build is used
    here ONLY as the reported baseline, never on the product path."""
    import concurrent.futures as cf

    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpulib
    import mi_fieldcalc_amd.synth as synth

    which = "ref" if cpulib.available("ref") else "oracle"
    lib = cpulib.CpuLib(which)
    kind = "reference" if which == "ref" else "port"
    nlev = NLEV
    xm, ym, _ = synth.grid_maps(NX, NY)
    u, v = synth.wind(NX, NY, SEED, nlev=nlev)
    n = NX * NY
    relvort, diverg = lib.raw("relvort"), lib.raw("divergence")
    import ctypes

    def one_level(l, out):
        fd = ctypes.c_int(0)  # ALL_DEFINED, like the GPU run
        relvort(NX, NY, u[l].ctypes.data, v[l].ctypes.data, xm.ctypes.data, ym.ctypes.data, out.ctypes.data, ctypes.addressof(fd), 1e35)
        fd = ctypes.c_int(0)
        diverg(NX, NY, u[l].ctypes.data, v[l].ctypes.data, xm.ctypes.data, ym.ctypes.data, out.ctypes.data, ctypes.addressof(fd), 1e35)

    out = np.empty((NY, NX), np.float32)
    for l in range(3):  # warm-up
        one_level(l, out)
    t0 = time.perf_counter()
    passes = 0
    while True:
        for l in range(nlev):
            one_level(l, out)
        passes += 1
        if time.perf_counter() - t0 >= seconds_target:
            break
    dt = time.perf_counter() - t0
    res = {
        "value": round(n * nlev * passes / dt / 1e6, 1),
        "unit": "Mcells/s",
        "cores": 1,
        "kind": kind,
        "sample": "%d passes over %d levels of 1440x720 (relvort then divergence per level, ALL_DEFINED, default build: OpenMP off), %.1f s" % (passes, nlev, dt),
    }
    extras = {"host_cpus": os.cpu_count()}
    try:
        with open("/proc/cpuinfo") as f:
            extras["cpu_model"] = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        pass
    # whole host: outer thread-per-level loop over serial reference calls (ctypes releases the GIL)
    workers = os.cpu_count() or 1
    outs = [np.empty((NY, NX), np.float32) for _ in range(workers)]

    def chunk(w):
        for l in range(w, nlev, workers):
            one_level(l, outs[w])

    with cf.ThreadPoolExecutor(workers) as ex:
        list(ex.map(chunk, range(workers)))  # warm-up
        t0 = time.perf_counter()
        passes = 0
        while time.perf_counter() - t0 < seconds_target / 2:
            list(ex.map(chunk, range(workers)))
            passes += 1
        dt = time.perf_counter() - t0
    extras["all_cores"] = {"value": round(n * nlev * passes / dt / 1e6, 1), "unit": "Mcells/s", "cores": workers, "kind": kind,
                           "sample": "%d passes, thread per level" % passes}
    # the library's own OpenMP path (ENABLE_OPENMP=ON), capped at 8 threads by the reference (openmp_tools.cc:54,65)
    if which == "ref" and cpulib.available("ref_omp"):
        os.environ.setdefault("OMP_NUM_THREADS", "8")
        omp = cpulib.CpuLib("ref_omp")
        r2, d2 = omp.raw("relvort"), omp.raw("divergence")

        def one_level_omp(l):
            fd = ctypes.c_int(0)
            r2(NX, NY, u[l].ctypes.data, v[l].ctypes.data, xm.ctypes.data, ym.ctypes.data, out.ctypes.data, ctypes.addressof(fd), 1e35)
            fd = ctypes.c_int(0)
            d2(NX, NY, u[l].ctypes.data, v[l].ctypes.data, xm.ctypes.data, ym.ctypes.data, out.ctypes.data, ctypes.addressof(fd), 1e35)

        for l in range(3):
            one_level_omp(l)
        t0 = time.perf_counter()
        passes = 0
        while time.perf_counter() - t0 < seconds_target / 2:
            for l in range(nlev):
                one_level_omp(l)
            passes += 1
        dt = time.perf_counter() - t0
        extras["openmp8"] = {"value": round(n * nlev * passes / dt / 1e6, 1), "unit": "Mcells/s", "cores": min(8, workers), "kind": kind,
                             "sample": "%d passes, ENABLE_OPENMP=ON, OMP_NUM_THREADS=8" % passes}
    return res, extras


def pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/),
    corrected as MI355X_MICROARCH.md prescribes; None if not collected."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(p):
        try:
            with open(p) as f:
                return json.load(f).get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", action="store_true", help="also time the SOME_DEFINED (per-cell test + count) variant")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import mi_fieldcalc_amd as fc
    import mi_fieldcalc_amd.synth as synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    # MIFC_BENCH_BACKEND=gloo is a self-test aid only (rehearsing the N>1 control flow on a
    # one-GPU box, ranks sharing cuda:0); the driver's runs use nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("MIFC_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # ---- this rank's batch: one ensemble member = 137 levels, generated in HBM
    xm, ym, _ = synth.grid_maps(NX, NY)
    dxm, dym = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
    du, dv = synth.device_wind(NX, NY, NLEV, SEED + 17 * rank, dev)
    rv = torch.empty_like(du)
    dg = torch.empty_like(du)
    flags = np.full(NLEV, fc.ALL_DEFINED, np.int32)
    counts = torch.zeros(NLEV, dtype=torch.int64, device=dev)

    ctx = fc.Context(dev_index)
    ctx.use_torch_stream()

    def step(check=False):
        if check:
            ok = ctx.vortdiv_levels_enqueue(du, dv, dxm, dym, rv, dg, fdefined=None, n_undefined=counts)
        else:
            ok = ctx.vortdiv_levels_enqueue(du, dv, dxm, dym, rv, dg, fdefined=flags, n_undefined=None)
        if not ok:
            raise RuntimeError(ctx.last_error())

    def barrier():
        if world > 1:
            dist.barrier()

    def timed(nsteps, check=False):
        starts = [torch.cuda.Event(enable_timing=True) for _ in range(nsteps)]
        ends = [torch.cuda.Event(enable_timing=True) for _ in range(nsteps)]
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(nsteps):
            starts[k].record()
            step(check)
            ends[k].record()
        torch.cuda.synchronize()
        barrier()
        wall = time.perf_counter() - t0
        kern_ms = [s.elapsed_time(e) for s, e in zip(starts, ends)]
        return wall, kern_ms

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    wall, kern_ms = timed(args.steps)
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    cells_per_step = NX * NY * NLEV
    value = cells_per_step * world * args.steps / wall / 1e6
    avg_kernel_s = float(np.mean(kern_ms)) / 1e3
    alg = algorithmic_bytes(NX, NY, NLEV)
    achieved = alg / avg_kernel_s / 1e9
    out = {
        "metric": "Mcells/s fused vorticity+divergence, 1440x720x137 grid; % HBM roofline",
        "value": round(value, 1),
        "unit": "Mcells/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32 in/out, f64 combine",
        "data": "synthetic",
        "config": {
            "workload": "1440x720x137 float32 fused relvort+divergence (BASELINE.json configs[2]), one 137-level ensemble member per GPU, inputs ALL_DEFINED and resident in HBM",
            "nx": NX, "ny": NY, "nlev": NLEV, "members_per_gpu": 1, "sharding": "members across GPUs, no collective",
            "tuning": os.environ.get("MIFC_VORTDIV_TUNE", "default"),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": pmc_traffic(),
            "algorithmic_bytes_per_launch": alg,
            "kernel_ms_avg": round(avg_kernel_s * 1e3, 4),
            "kernel_ms_min": round(float(np.min(kern_ms)), 4),
        },
    }
    if args.check and rank == 0:
        du[:, 100:110, 200:260] = float(fc.UNDEF)  # some undefined cells so that the count path does real work
        for _ in range(2):
            step(True)
        torch.cuda.synchronize()
        _, kms = timed(max(5, args.steps // 2), check=True)
        out["check_variant"] = {"kernel_ms_avg": round(float(np.mean(kms)), 4),
                                "Mcells_per_s": round(cells_per_step / (float(np.mean(kms)) / 1e3) / 1e6, 1),
                                "note": "SOME_DEFINED inputs: per-cell undefined tests + per-level counts (memset + kernel)"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base, extras = cpu_baseline()
        out["cpu_baseline"] = base
        out["cpu_baseline_extra"] = extras
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
