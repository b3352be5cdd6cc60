#!/usr/bin/env python3
"""Fused derived batch on hybrid levels (BASELINE.json config 2 x 137 levels): kernel time by HIP events
for the trio ff + RH + theta (28 B/cell + ps) and the quartet with the dew point (32 B/cell + ps),
device resident, rotating over two buffer sets (11 GB: nothing is served from the Infinity Cache),
ALL_DEFINED and SOME_DEFINED, for a few persistent-grid sizes (MIFC_DERIVED_BLOCKS).
Usage: python tools/bench_derived.py [nlev]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402

NX, NY = 1440, 720
NLEV = int(sys.argv[1]) if len(sys.argv) > 1 else 137


def main():
    dev = torch.device("cuda", 0)
    ctx = fc.Context(0)
    ctx.use_torch_stream()
    al, bl = synth.hybrid_levels(max(NLEV, 3))
    al, bl = al[:NLEV], bl[:NLEV]
    # SPREAD_GIB=<g>: every array of the two buffer sets is allocated behind a spacer of g GiB, so that the arrays a launch
    # streams do not lie next to each other in physical memory (arrays allocated one after the other are the slow
    # placement: mi-fieldcalc_amd/placement.py); unset = allocations as they come
    spread = float(os.environ.get("SPREAD_GIB", "0"))
    spacers = []

    def fresh():
        if spread > 0:
            spacers.append(torch.empty(int(spread * 2**30), dtype=torch.uint8, device=dev))
        return torch.empty((NLEV, NY, NX), dtype=torch.float32, device=dev)

    sets = []
    for s in range(2):
        dst = {k: fresh() for k in ("u", "ff", "t", "temp", "v", "hum", "q", "hum2", "dd")}
        u, v = synth.device_wind(NX, NY, NLEV, 100 + s, dev)
        t, q, ps = synth.device_thermo(NX, NY, NLEV, 200 + s, dev)
        for k, src in (("u", u), ("v", v), ("t", t), ("q", q)):
            dst[k].copy_(src)
        del u, v, t, q
        torch.cuda.empty_cache()
        sets.append(dict(u=dst["u"], v=dst["v"], t=dst["t"], q=dst["q"], ps=ps, out={k: dst[k] for k in ("ff", "temp", "hum", "hum2", "dd")}))
    if spread > 0:
        print("arrays allocated behind spacers of %.1f GiB" % spread)
    cnt = torch.zeros(5 * NLEV, dtype=torch.int64, device=dev)
    n = NX * NY * NLEV
    if os.environ.get("DERIVED_PLACEMENT"):
        # the seven big arrays of the trio chosen from pools like bench.py's four (placement.choose_search_rounds): what a
        # long-lived batch can get on this box, next to "as allocated" below.  Timed hot on the one chosen set.
        from mi_fieldcalc_amd.placement import choose_search_rounds

        flags_all = np.full(NLEV, fc.ALL_DEFINED, np.int32)
        ps0 = sets[0]["ps"]
        spare = {k: sets[0]["out"][k] for k in ("hum2", "dd")}

        def probe(arr):
            u_, v_, t_, q_, ff_, te_, hu_ = arr
            # the inputs must be REAL data: this kernel's time depends on the values (table lookups: lanes with equal
            # indices are served by one LDS access; a first version probed uninitialised arrays and "found" 0.68 ms)
            for dstt, src in ((u_, sets[0]["u"]), (v_, sets[0]["v"]), (t_, sets[0]["t"]), (q_, sets[0]["q"])):
                if dstt.data_ptr() != src.data_ptr():
                    dstt.copy_(src)
            out = dict(ff=ff_, temp=te_, hum=hu_, **spare)
            ms = []
            for k in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    assert ctx.hlevel_derived_batch(u_, v_, t_, q_, ps0, al, bl, fdef_wind=flags_all, fdef_thermo=flags_all, out=out, enqueue_counts=cnt,
                                                    temp=("", 3), hum=("", 1)) is not None
                e1.record()
                torch.cuda.synchronize()
                if k:
                    ms.append(e0.elapsed_time(e1) / 3)
            return float(np.median(ms))

        chosen, rep = choose_search_rounds(lambda: torch.empty((NLEV, NY, NX), dtype=torch.float32, device=dev), 7, probe, rounds=2, pool_size=40,
                                           random_sets=40, max_probes=200, device=dev)
        def busy(arr):  # freeing the pools idles the GPU for a while; this kernel is sensitive to the clocks it then starts with
            for _ in range(6):
                probe(arr)

        busy(chosen)
        t_best = probe(chosen)
        alg = n * 28 + NX * NY * 4
        print("ff + RH + theta, seven arrays chosen by the placement search (hot, one set): %.4f ms = %.1f %% of 8 TB/s; as allocated in one go %.4f; "
              "probes min / median / max %s; per pool %s" % (t_best, alg / t_best / 1e6 / 80.0, rep["allocated_in_one_go_ms"], rep["probe_ms_min_median_max"],
                                                            rep["rounds_chosen_ms"]))
        first = (sets[0]["u"], sets[0]["v"], sets[0]["t"], sets[0]["q"], sets[0]["out"]["ff"], sets[0]["out"]["temp"], sets[0]["out"]["hum"])
        busy(first)
        t_hot = probe(first)
        print("ff + RH + theta, the tool's first buffer set as allocated (hot, one set): %.4f ms = %.1f %%" % (t_hot, alg / t_hot / 1e6 / 80.0))
        del chosen
    print("%dx%dx%d, kernel ms by HIP events (median of 9), two rotating buffer sets, %s" % (NX, NY, NLEV, "8 launches back to back per event pair" if os.environ.get("SUSTAINED", "1") != "0" else "one launch per event pair, synchronize after each"))
    print("%-34s %-14s %8s %9s %7s" % ("outputs", "flags / blocks", "ms", "GB/s", "frac"))
    for blocks in os.environ.get("DERIVED_SWEEP", ",4096,8192,16384,32768,65536,140000").split(","):
        pipe = None
        if blocks.startswith("pipe"):  # "pipe0" / "pipe1": the default grid with / without the two-trip software pipeline
            pipe, blocks = blocks[4:], ""
        if blocks:
            os.environ["MIFC_DERIVED_BLOCKS"] = blocks
        else:
            os.environ.pop("MIFC_DERIVED_BLOCKS", None)
        if pipe is None:
            os.environ.pop("MIFC_DERIVED_PIPE", None)
        else:
            os.environ["MIFC_DERIVED_PIPE"] = pipe
            blocks = "pipe" + pipe
        ctx.reload_env()
        if os.environ.get("DERIVED_ONLY_TRIO"):
            pass
        for name, kw, bytes_per_cell in (("ff + RH + theta", dict(temp=("", 3), hum=("", 1)), 28), ("ff + RH + theta + Td", dict(temp=("", 3), hum=("", 1), hum2=("", 9)), 32),
                                         ("RH + theta (no wind)", dict(temp=("", 3), hum=("", 1), ff=False), 16),
                                         ("ff + dd + RH + theta + Td", dict(temp=("", 3), hum=("", 1), hum2=("", 9), dd=True), 36)):
            for flag in (fc.ALL_DEFINED, fc.SOME_DEFINED):
                flags = np.full(NLEV, flag, np.int32)
                # SUSTAINED=1 (default): 8 launches back to back, alternating the two buffer sets (cold caches, the clocks of a
                # GPU that is kept busy -- what a member pipeline sees); SUSTAINED=0: one launch per event pair with a
                # synchronize after each, as the first half of the round measured (the GPU idles in between and this
                # VALU-heavy kernel then runs ~10 % slower: profiles/r02/experiments/bench_derived_end_of_round.txt)
                sustained = os.environ.get("SUSTAINED", "1") != "0"
                ms = []
                for r in range(11):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for k in range(8 if sustained else 1):
                        s = sets[(r + k) % 2]
                        assert ctx.hlevel_derived_batch(s["u"], s["v"], s["t"], s["q"], s["ps"], al, bl, fdef_wind=flags, fdef_thermo=flags, out=s["out"],
                                                        enqueue_counts=cnt, **kw) is not None
                    e1.record()
                    torch.cuda.synchronize()
                    if r >= 2:
                        ms.append(e0.elapsed_time(e1) / (8 if sustained else 1))
                med = float(np.median(ms))
                alg = n * bytes_per_cell + NX * NY * 4
                print("%-34s %-14s %8.4f %9.1f %7.4f" % (name, ("ALL" if flag == fc.ALL_DEFINED else "SOME") + " / " + (blocks or "default"), med, alg / med / 1e6, alg / med / 1e6 / 8000.0))


if __name__ == "__main__":
    main()
