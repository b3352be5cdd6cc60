#!/usr/bin/env python3
"""Per-operator table: every function on the hot path on a
1440x720xNLEV device-resident batch -- GPU time of the synchronous batched call
(kernel + flag read-back), algorithmic bytes, fraction of the HBM roofline --
next to the reference CPU path (oracle/_ref when present, else the restatement)
on one core for the same operator.

    python tools/bench_ops.py [NLEV]      -> one JSON line per operator + a text table
"""
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIFC_LIB_PATH", os.path.join(ROOT, "mi-fieldcalc_amd", "libmifc_measure.so"))  # mifc_timing_* / yardsticks: measurement build
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import cpulib  # noqa: E402  (CPU baseline only)
import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402

NX, NY = 1440, 720
NLEV = int(sys.argv[1]) if len(sys.argv) > 1 else 137
PEAK = 8000.0
DEV = torch.device("cuda", 0)
ROUNDS = 5


CTX = None


def gpu_time(fn):
    """-> (wall ms of the synchronous call, kernel ms from HIP events around the launches); medians."""
    fn()
    torch.cuda.synchronize()
    ts, ks = [], []
    for _ in range(ROUNDS):
        CTX.timing_begin()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        ks.append(CTX.timing_end_ms())
    return float(np.median(ts)), float(np.median(ks))


def cpu_rate(lib, op, args, nlev_cpu=4, min_s=1.5):
    """Mcells/s of the CPU checker on one core; args are per-level numpy fields / scalars."""
    t0 = time.perf_counter()
    n = 0
    while True:
        for _ in range(nlev_cpu):
            lib.call(op, NX, NY, *args, fdefined=fc.ALL_DEFINED)
        n += nlev_cpu
        if time.perf_counter() - t0 > min_s:
            break
    return NX * NY * n / (time.perf_counter() - t0) / 1e6


def main():
    which = "ref" if cpulib.available("ref") else "oracle"
    cpu = cpulib.CpuLib(which)
    global CTX
    ctx = CTX = fc.Context(0)
    n = NX * NY
    cells = n * NLEV
    xm, ym, fcor = synth.grid_maps(NX, NY)
    dxm, dym, dfc = (torch.from_numpy(a).to(DEV) for a in (xm, ym, fcor))
    u, v = synth.device_wind(NX, NY, NLEV, 3, DEV)
    placement = None
    if os.environ.get("BENCH_OPS_PLACED", "1") != "0":
        # the four arrays the stencil operators stream (u, v and the two outputs) are chosen like bench.py chooses the headline
        # batch: from pools of arrays, by timing the fused kernel on index sets (mi-fieldcalc_amd/placement.py)
        from mi_fieldcalc_amd.placement import choose_search_rounds

        pflags = np.full(NLEV, fc.ALL_DEFINED, np.int32)

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

        ctx.use_torch_stream()
        (pu, pv, po, po2), placement = choose_search_rounds(lambda: torch.empty((NLEV, NY, NX), dtype=torch.float32, device=DEV), 4, probe, rounds=2,
                                                            pool_size=48, random_sets=48, max_probes=200, device=DEV)
        pu.copy_(u)
        pv.copy_(v)
        u, v = pu, pv
        print(json.dumps({"placement": placement}), flush=True)
    z = (5500.0 + 10.0 * u).contiguous()
    t, q, ps = synth.device_thermo(NX, NY, NLEV, 4, DEV)
    # "tall field" view for the single-field elementwise operators: one field of NLEV*NY rows
    tall = lambda x: x.reshape(NLEV * NY, NX)  # noqa: E731
    ps_tall = ps.repeat(NLEV, 1).contiguous()
    p_tall = (ps_tall * 0.7 + 10.0).contiguous()
    rh = (q * 4000.0 + 5.0).contiguous()
    out = po if placement is not None else torch.empty_like(u)
    out2 = po2 if placement is not None else torch.empty_like(u)
    flags = np.full(NLEV, fc.ALL_DEFINED, np.int32)
    # host copies of ONE level for the CPU baseline
    h = {k: x[0].cpu().numpy() for k, x in dict(u=u, v=v, z=z, t=t, q=q, rh=rh).items()}
    h["ps"] = ps.cpu().numpy()
    h["p"] = (h["ps"] * 0.7 + 10.0).astype(np.float32)

    rows = []

    only = os.environ.get("BENCH_OPS_ONLY")  # regular expression on the operator name

    tested_flags = np.full(NLEV, fc.SOME_DEFINED, np.int32)
    counts = torch.zeros(NLEV, dtype=torch.int64, device=DEV)
    counts5 = torch.zeros(5 * NLEV, dtype=torch.int64, device=DEV)

    def burst_ms(enq, k=10, reps=5):
        """ms per launch of `k` asynchronous launches back to back between ONE pair of events -- how bench.py times the headline"""
        for _ in range(k):
            enq()
        torch.cuda.synchronize()
        ms = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                enq()
            e0.record()
            for _ in range(k):
                enq()
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1) / k)
        return float(np.median(ms))

    def add(name, bytes_per_cell, once_bytes, gpu_fn, cpu_op, cpu_args, enq=None, enq_tested=None):
        if only and not re.search(only, name):
            return
        ms, kms = gpu_time(gpu_fn)
        alg = cells * bytes_per_cell + once_bytes
        cr = 0.0 if os.environ.get("BENCH_OPS_NO_CPU") else cpu_rate(cpu, cpu_op, cpu_args)  # A/B runs of two library builds skip the CPU column
        rec = {"op": name, "ms": round(ms, 4), "kernel_ms": round(kms, 4), "Mcells_per_s": round(cells / ms / 1e3, 1), "algorithmic_bytes": alg,
               "GBps": round(alg / ms / 1e6, 1), "frac_of_8TBps": round(alg / ms / 1e6 / PEAK, 4),
               "kernel_frac_of_8TBps": round(alg / kms / 1e6 / PEAK, 4) if kms > 0 else None,
               "cpu_Mcells_per_s_1core": round(cr, 1), "cpu_kind": "reference" if which == "ref" else "port", "nlev": NLEV}
        if enq is not None:  # steady state: launches back to back, the way the headline is timed
            b = burst_ms(enq)
            rec["burst_ms"], rec["burst_frac_of_8TBps"] = round(b, 4), round(alg / b / 1e6 / PEAK, 4)
        if enq_tested is not None:
            b = burst_ms(enq_tested)
            rec["burst_tested_ms"], rec["burst_tested_frac_of_8TBps"] = round(b, 4), round(alg / b / 1e6 / PEAK, 4)
        rows.append(rec)
        print(json.dumps(rec), flush=True)

    st = lambda op, f0, f1, fcc, two=False: (lambda: ctx.stencil_levels(op, f0, f1, dxm, dym, fcc, fdefined=flags, out0=out, out1=out2 if two else None))  # noqa: E731
    en = lambda op, f0, f1, fcc, two=False, tested=False: (lambda: ctx.stencil_levels_enqueue(  # noqa: E731
        op, f0, f1, dxm, dym, fcc, out, out2 if two else None, fdefined=tested_flags if tested else flags, n_undefined=counts if tested else None))

    def add_st(name, bpc, once, op, f0, f1, fcc, cpu_op, cpu_args, two=False):
        add(name, bpc, once, st(op, f0, f1, fcc, two), cpu_op, cpu_args, enq=en(op, f0, f1, fcc, two), enq_tested=en(op, f0, f1, fcc, two, True))

    add_st("relvort+divergence (fused)", 16, 8 * n, "vortdiv", u, v, None, "relvort", [h["u"], h["v"], xm, ym], two=True)
    add_st("relvort", 12, 8 * n, "relvort", u, v, None, "relvort", [h["u"], h["v"], xm, ym])
    add_st("divergence", 12, 8 * n, "divergence", u, v, None, "divergence", [h["u"], h["v"], xm, ym])
    add_st("absvort", 12, 12 * n, "absvort", u, v, dfc, "absvort", [h["u"], h["v"], xm, ym, fcor])
    for c in (1, 2, 3, 4):
        add_st("gradient compute=%d" % c, 8, 8 * n, "gradient%d" % c, z, None, None, "gradient", [h["z"], xm, ym, c])
    add_st("plevelgwind_xcomp", 8, 8 * n, "plevelgwind_xcomp", z, None, dfc, "plevelgwind_xcomp", [h["z"], xm, ym, fcor])
    add_st("plevelgwind_ycomp", 8, 8 * n, "plevelgwind_ycomp", z, None, dfc, "plevelgwind_ycomp", [h["z"], xm, ym, fcor])
    add_st("plevelgvort", 8, 12 * n, "plevelgvort", z, None, dfc, "plevelgvort", [h["z"], xm, ym, fcor])
    add_st("ilevelgwind", 12, 12 * n, "ilevelgwind", z, None, dfc, "ilevelgwind", [h["z"], xm, ym, fcor], two=True)
    o_t = tall(out)
    add("vectorabs", 12, 0, lambda: ctx.vectorabs(tall(u), tall(v), fdefined=fc.ALL_DEFINED, out=o_t), "vectorabs", [h["u"], h["v"]])
    add("pleveltemp c=3 (T->theta)", 8, 0, lambda: ctx.pleveltemp(tall(t), 850.0, "kelvin", 3, fdefined=fc.ALL_DEFINED, out=o_t), "pleveltemp", [h["t"], 850.0, "kelvin", 3])
    add("hleveltemp c=3 (T->theta)", 12, 0, lambda: ctx.hleveltemp(tall(t), ps_tall, 12.5, 0.73, "kelvin", 3, fdefined=fc.ALL_DEFINED, out=o_t), "hleveltemp",
        [h["t"], h["ps"], 12.5, 0.73, "kelvin", 3])
    add("aleveltemp c=3 (T->theta)", 12, 0, lambda: ctx.aleveltemp(tall(t), p_tall, "kelvin", 3, fdefined=fc.ALL_DEFINED, out=o_t), "aleveltemp", [h["t"], h["p"], "kelvin", 3])
    add("plevelhum c=1 (T,q->RH)", 12, 0, lambda: ctx.plevelhum(tall(t), tall(q), 850.0, "kelvin", 1, fdefined=fc.ALL_DEFINED, out=o_t), "plevelhum", [h["t"], h["q"], 850.0, "kelvin", 1])
    add("hlevelhum c=1 (T,q->RH)", 16, 0, lambda: ctx.hlevelhum(tall(t), tall(q), ps_tall, 12.5, 0.73, "kelvin", 1, fdefined=fc.ALL_DEFINED, out=o_t), "hlevelhum",
        [h["t"], h["q"], h["ps"], 12.5, 0.73, "kelvin", 1])
    add("hlevelhum c=9 (T,q->Td K)", 16, 0, lambda: ctx.hlevelhum(tall(t), tall(q), ps_tall, 12.5, 0.73, "kelvin", 9, fdefined=fc.ALL_DEFINED, out=o_t), "hlevelhum",
        [h["t"], h["q"], h["ps"], 12.5, 0.73, "kelvin", 9])
    add("alevelhum c=2 (theta,q->RH)", 16, 0, lambda: ctx.alevelhum(tall(t), tall(q), p_tall, "kelvin", 2, fdefined=fc.ALL_DEFINED, out=o_t), "alevelhum", [h["t"], h["q"], h["p"], "kelvin", 2])
    add("cvhum c=1 (T,RH->Td K)", 12, 0, lambda: ctx.cvhum(tall(t), tall(rh), "kelvin", 1, fdefined=fc.ALL_DEFINED, out=o_t), "cvhum", [h["t"], h["rh"], "kelvin", 1])
    a, b = synth.hybrid_levels(NLEV)
    add("fused ff+RH+theta (hybrid)", 28, 4 * n, lambda: ctx.hlevel_derived_levels(u, v, t, q, ps, a, b, fdef_wind=flags, fdef_thermo=flags,
                                                                                    out={"ff": out, "rh": out2, "theta": o_t.reshape(NLEV, NY, NX)}),
        "hlevelhum", [h["t"], h["q"], h["ps"], 12.5, 0.73, "kelvin", 1],
        enq=lambda: ctx.hlevel_derived_batch(u, v, t, q, ps, a, b, temp=("", 3), hum=("", 1), fdef_wind=flags, fdef_thermo=flags,
                                             out={"ff": out, "hum": out2, "temp": o_t.reshape(NLEV, NY, NX)}, enqueue_counts=counts5))
    o3 = torch.empty_like(u)
    add("fused ff+RH+theta+Td (hybrid, one launch)", 32, 4 * n,
        lambda: ctx.hlevel_derived_batch(u, v, t, q, ps, a, b, temp=("", 3), hum=("", 1), hum2=("", 9), fdef_wind=flags, fdef_thermo=flags,
                                         out={"ff": out, "hum": out2, "temp": o_t.reshape(NLEV, NY, NX), "hum2": o3}),
        "hlevelhum", [h["t"], h["q"], h["ps"], 12.5, 0.73, "kelvin", 9],
        enq=lambda: ctx.hlevel_derived_batch(u, v, t, q, ps, a, b, temp=("", 3), hum=("", 1), hum2=("", 9), fdef_wind=flags, fdef_thermo=flags,
                                             out={"ff": out, "hum": out2, "temp": o_t.reshape(NLEV, NY, NX), "hum2": o3}, enqueue_counts=counts5))
    del o3
    # ---- the rest of the stencil family, on the batch seen as one tall field
    xm_t, ym_t, fc_t = (x.repeat(NLEV, 1).contiguous() for x in (dxm, dym, dfc))
    ALLD = fc.ALL_DEFINED
    add("advection", 24, 0, lambda: ctx.advection(tall(z), tall(u), tall(v), xm_t, ym_t, 1.0, fdefined=ALLD, out=o_t), "advection",
        [h["z"], h["u"], h["v"], xm, ym, 1.0])
    add("jacobian (one tall field)", 20, 0, lambda: ctx.jacobian(tall(z), tall(u), xm_t, ym_t, fdefined=ALLD, out=o_t), "jacobian", [h["z"], h["u"], xm, ym])
    add("jacobian (level batch)", 12, 2 * 4 * n, lambda: ctx.stencil_levels("jacobian", z, u, dxm, dym, fdefined=flags, out0=out), "jacobian",
        [h["z"], h["u"], xm, ym])
    add("shapiro2_filter", 8, 0, lambda: ctx.shapiro2_filter(tall(z), fdefined=ALLD, out=o_t), "shapiro2_filter", [h["z"]])
    add("thermalFrontParameter", 16, 0, lambda: ctx.thermalFrontParameter(tall(t), xm_t, ym_t, fdefined=ALLD, out=o_t),
        "thermalFrontParameter", [h["t"], xm, ym])
    add("plevelqvector c=1", 24, 0, lambda: ctx.plevelqvector(tall(z), tall(t), xm_t, ym_t, fc_t, 500.0, 1, fdefined=ALLD, out=o_t),
        "plevelqvector", [h["z"], h["t"], xm, ym, fcor, 500.0, 1])
    # ---- 8f-3: pointwise catalogue
    t2 = (t - 20.0).contiguous()
    t3, t4, rh2, u2, v2 = (t - 8.0).contiguous(), (t - 28.0).contiguous(), (rh * 0.9).contiguous(), (u * 0.5).contiguous(), (v * 0.5).contiguous()
    add("plevelthe c=1", 12, 0, lambda: ctx.plevelthe(tall(t), tall(rh), 850.0, 1, fdefined=ALLD, out=o_t), "plevelthe", [h["t"], h["rh"], 850.0, 1])
    add("hlevelthe c=1", 16, 0, lambda: ctx.hlevelthe(tall(t), tall(q), ps_tall, 12.5, 0.73, 1, fdefined=ALLD, out=o_t), "hlevelthe",
        [h["t"], h["q"], h["ps"], 12.5, 0.73, 1])
    add("kIndex", 24, 0, lambda: ctx.kIndex(tall(t2), tall(t3), tall(rh2), tall(t), tall(rh), 500.0, 700.0, 850.0, 1, fdefined=ALLD, out=o_t), "kIndex",
        [h["t"] - 20, h["t"], h["rh"], h["t"], h["rh"], 500.0, 700.0, 850.0, 1])
    add("showalterIndex", 16, 0, lambda: ctx.showalterIndex(tall(t2), tall(t), tall(rh), 500.0, 850.0, 1, fdefined=ALLD, out=o_t), "showalterIndex",
        [h["t"] - 20, h["t"], h["rh"], 500.0, 850.0, 1])
    add("sweatIndex", 36, 0, lambda: ctx.sweatIndex(tall(t), tall(t2), tall(t3), tall(t4), tall(u), tall(v), tall(u2), tall(v2), fdefined=ALLD, out=o_t),
        "sweatIndex", [h["t"], h["t"] - 20, h["t"], h["t"] - 20, h["u"], h["v"], h["v"], h["u"]])
    add("abshum", 12, 0, lambda: ctx.abshum(tall(t), tall(rh), fdefined=ALLD, out=o_t), "abshum", [h["t"], h["rh"]])
    add("windCooling", 16, 0, lambda: ctx.windCooling(tall(t), tall(u), tall(v), 1, fdefined=ALLD, out=o_t), "windCooling", [h["t"], h["u"], h["v"], 1])
    add("cvtemp c=1", 8, 0, lambda: ctx.cvtemp(tall(t), 1, fdefined=ALLD, out=o_t), "cvtemp", [h["t"], 1])
    add("fieldOPERfield +", 12, 0, lambda: ctx.fieldOPERfield(1, tall(t), tall(q), fdefined=ALLD, out=o_t), "fieldOPERfield", [1, h["t"], h["q"]])
    add("log10Field", 8, 0, lambda: ctx.log10Field(tall(t), fdefined=ALLD, out=o_t), "log10Field", [h["t"]])
    add("expField", 8, 0, lambda: ctx.expField(tall(q), fdefined=ALLD, out=o_t), "expField", [h["q"]])
    add("powerField ^0.37", 8, 0, lambda: ctx.powerField(tall(t), 0.37, fdefined=ALLD, out=o_t), "powerField", [h["t"], 0.37])
    add("snow_in_cm", 16, 0, lambda: ctx.snow_in_cm(tall(q), tall(t), tall(t2), fdefined=ALLD, out=o_t), "snow_in_cm", [h["q"], h["t"], h["t"] - 20])
    # ---- the f1 operators over the level batch with shared map factors (mifc_stencil_levels_ex)
    pres = np.linspace(1000.0, 100.0, NLEV).astype(np.float32)
    add("advection (level batch)", 16, 8 * n, lambda: ctx.stencil_levels_ex("advection", z, u, v, dxm, dym, scalar=1.0, fdefined=flags, out0=out), "advection",
        [h["z"], h["u"], h["v"], xm, ym, 1.0])
    add("thermalFrontParameter (level batch)", 8, 8 * n, lambda: ctx.stencil_levels_ex("thermalFrontParameter", t, xmapr=dxm, ymapr=dym, fdefined=flags, out0=out),
        "thermalFrontParameter", [h["t"], xm, ym])
    add("plevelqvector c=1 (level batch)", 12, 12 * n,
        lambda: ctx.stencil_levels_ex("plevelqvector", z, t, None, dxm, dym, dfc, level_scalars=pres, compute=1, fdefined=flags, out0=out), "plevelqvector",
        [h["z"], h["t"], xm, ym, fcor, 500.0, 1])
    add("shapiro2_filter (level batch)", 8, 0, lambda: ctx.stencil_levels_ex("shapiro2_filter", z, fdefined=flags, out0=out), "shapiro2_filter", [h["z"]])
    # ---- 8f-4: reductions over the batch's levels taken as ensemble members of one 1440x720 field
    nm = min(NLEV, 51)
    # 51 members are 211 MB -- they would sit in the 256 MB Infinity Cache between calls: every call takes the
    # next of four member sets (t, q-, u-, v-derived fields of the same size), 0.85 GB in rotation
    msets = [[src[k] for k in range(nm)] for src in (t, (t + 1.0).contiguous(), (t - 2.0).contiguous(), (t + 3.0).contiguous())] if NLEV >= nm else [[t[k] for k in range(nm)]]
    state = {"k": 0}

    def next_members():
        state["k"] += 1
        return msets[state["k"] % len(msets)]

    hm = [t[k].cpu().numpy() for k in range(min(nm, 8))]
    o1 = out[0]

    def add_ens(name, gpu_fn, cpu_op, cpu_args):
        if only and not re.search(only, name):
            return
        ms, kms = gpu_time(gpu_fn)
        alg = n * 4 * (nm + 1)
        t0 = time.perf_counter()
        cpu.call(cpu_op, NX, NY, *cpu_args, fdefined=fc.ALL_DEFINED)
        cr = n * len(hm) / (time.perf_counter() - t0) / 1e6
        rec = {"op": name, "ms": round(ms, 4), "kernel_ms": round(kms, 4), "Mcells_per_s": round(n * nm / ms / 1e3, 1), "algorithmic_bytes": alg,
               "GBps": round(alg / ms / 1e6, 1), "frac_of_8TBps": round(alg / ms / 1e6 / PEAK, 4),
               "kernel_frac_of_8TBps": round(alg / kms / 1e6 / PEAK, 4) if kms > 0 else None, "cpu_Mcells_per_s_1core": round(cr, 1),
               "cpu_kind": "reference" if which == "ref" else "port", "nlev": nm}
        rows.append(rec)
        print(json.dumps(rec), flush=True)

    mflags = [fc.ALL_DEFINED] * nm
    add_ens("meanValue (%d members)" % nm, lambda: ctx.meanValue(next_members(), mflags, out=o1), "meanValue", [hm, mflags[:len(hm)]])
    add_ens("stddevValue (%d members)" % nm, lambda: ctx.stddevValue(next_members(), mflags, out=o1), "stddevValue", [hm, mflags[:len(hm)]])
    add_ens("probability>280 (%d members)" % nm, lambda: ctx.probability(1, next_members(), mflags, [280.0], out=o1), "probability",
            [1, hm, mflags[:len(hm)], [280.0]])
    print()
    print("call ms: the synchronous call (kernel + flag read-back); kernel ms: HIP events around that call's launches; burst: ms per launch of 10 asynchronous")
    print("launches back to back between one pair of events (how bench.py times the headline), ALL_DEFINED and tested (SOME_DEFINED flags, clean data)")
    print("%-32s %9s %9s %11s %7s %9s %9s %7s %9s %7s %13s" % ("operator (1440x720x%d)" % NLEV, "call ms", "kernel ms", "Mcells/s", "frac", "k.frac", "burst ms", "frac",
                                                                "tested ms", "frac", "CPU Mcells/s"))
    for r in rows:
        b = ("%9.4f %7.3f" % (r["burst_ms"], r["burst_frac_of_8TBps"])) if "burst_ms" in r else "%9s %7s" % ("-", "-")
        bt = ("%9.4f %7.3f" % (r["burst_tested_ms"], r["burst_tested_frac_of_8TBps"])) if "burst_tested_ms" in r else "%9s %7s" % ("-", "-")
        print("%-32s %9.4f %9.4f %11.0f %7.3f %9.3f %s %s %13.0f" % (r["op"], r["ms"], r["kernel_ms"], r["Mcells_per_s"], r["frac_of_8TBps"],
                                                                    r["kernel_frac_of_8TBps"] or 0.0, b, bt, r["cpu_Mcells_per_s_1core"]))


if __name__ == "__main__":
    main()
