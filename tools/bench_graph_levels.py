#!/usr/bin/env python3
"""Launch-bound work (BASELINE.json config 2: one 1440x720 level of the fused elementwise batch = 33 MB, 4 us at peak): a
caller's loop over levels, one asynchronous call per level, (a) call by call through Python, (b) recorded into ONE HIP graph
(mifc_graph_begin / _end) and replayed, next to (c) the level batch as one launch.  Cold: the 137 levels are 4.4 GB, far
beyond the 256 MB Infinity Cache.

    python tools/bench_graph_levels.py [nlev]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402

NX, NY = 1440, 720


def main():
    nlev = int(sys.argv[1]) if len(sys.argv) > 1 else 137
    dev = torch.device("cuda", 0)
    ctx = fc.Context(0)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx.use_torch_stream()
        u, v = synth.device_wind(NX, NY, nlev, 3, dev)
        t, q, ps = synth.device_thermo(NX, NY, nlev, 4, dev)
        al, bl = synth.hybrid_levels(nlev)
        ff, rh, th = torch.empty_like(u), torch.empty_like(u), torch.empty_like(u)
        cnt = torch.zeros(5 * nlev, dtype=torch.int64, device=dev)
        cnt1 = [torch.zeros(5, dtype=torch.int64, device=dev) for _ in range(nlev)]
        flags = np.full(nlev, fc.ALL_DEFINED, np.int32)
        one = np.full(1, fc.ALL_DEFINED, np.int32)
        alg = NX * NY * (nlev * 28 + 4)

        def batch():
            ctx.hlevel_derived_batch(u, v, t, q, ps, al, bl, temp=("", 3), hum=("", 1), fdef_wind=flags, fdef_thermo=flags,
                                     out={"ff": ff, "temp": th, "hum": rh}, enqueue_counts=cnt)

        def per_level():
            for l in range(nlev):
                ctx.hlevel_derived_batch(u[l:l + 1], v[l:l + 1], t[l:l + 1], q[l:l + 1], ps, al[l:l + 1], bl[l:l + 1], temp=("", 3), hum=("", 1),
                                         fdef_wind=one, fdef_thermo=one, out={"ff": ff[l:l + 1], "temp": th[l:l + 1], "hum": rh[l:l + 1]},
                                         enqueue_counts=cnt1[l])

        def timed(fn, reps=10):
            fn()
            torch.cuda.synchronize()
            ms, wall = [], []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                wall.append((time.perf_counter() - t0) * 1e3)
                ms.append(e0.elapsed_time(e1))
            return float(np.median(ms)), float(np.median(wall))

        batch()
        torch.cuda.synchronize()
        want = (ff.clone(), rh.clone(), th.clone())
        rows = [("one launch over the level batch", ) + timed(batch)]
        for x in (ff, rh, th):
            x.zero_()
        rows.append(("one call per level, through Python", ) + timed(per_level))
        torch.cuda.synchronize()
        ok_loop = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip((ff, rh, th), want))
        prepared = ctx.prepare(per_level)
        for x in (ff, rh, th):
            x.zero_()
        rows.append(("one call per level, prepared (Context.prepare)", ) + timed(prepared.launch))
        torch.cuda.synchronize()
        ok_loop = ok_loop and all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip((ff, rh, th), want))
        t0 = time.perf_counter()
        for _ in range(20):
            prepared.launch()
        host_us = (time.perf_counter() - t0) / (20 * nlev) * 1e6
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            per_level()
        host_us_wrapped = (time.perf_counter() - t0) / (5 * nlev) * 1e6
        torch.cuda.synchronize()
        with ctx.graph_capture(max_levels_per_call=8) as g:
            per_level()
        for x in (ff, rh, th):
            x.zero_()
        rows.append(("the same calls recorded into ONE HIP graph", ) + timed(g.launch))
        torch.cuda.synchronize()
        ok_graph = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip((ff, rh, th), want))
        for lanes in (2, 4, 8):
            with ctx.graph_capture(max_levels_per_call=8, lanes=lanes) as gl:
                for l in range(nlev):
                    gl.lane(l % lanes)
                    ctx.hlevel_derived_batch(u[l:l + 1], v[l:l + 1], t[l:l + 1], q[l:l + 1], ps, al[l:l + 1], bl[l:l + 1], temp=("", 3), hum=("", 1),
                                             fdef_wind=one, fdef_thermo=one, out={"ff": ff[l:l + 1], "temp": th[l:l + 1], "hum": rh[l:l + 1]},
                                             enqueue_counts=cnt1[l])
            for x in (ff, rh, th):
                x.zero_()
            rows.append(("... recorded in %d lanes (levels are independent)" % lanes, ) + timed(gl.launch))
            torch.cuda.synchronize()
            ok_graph = ok_graph and all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip((ff, rh, th), want))
            gl.close()
        # ... and with ONE fill of all counters at the head of the graph instead of one per call
        call = torch.zeros(5 * nlev, dtype=torch.int64, device=dev)
        for lanes in (1, 4, 8, 16):
            with ctx.graph_capture(max_levels_per_call=8, lanes=lanes) as gl:
                ctx.zero_counts_enqueue(call)
                ctx.counts_accumulate(True)
                for l in range(nlev):
                    if lanes > 1:
                        gl.lane(l % lanes)
                    ctx.hlevel_derived_batch(u[l:l + 1], v[l:l + 1], t[l:l + 1], q[l:l + 1], ps, al[l:l + 1], bl[l:l + 1], temp=("", 3), hum=("", 1),
                                             fdef_wind=one, fdef_thermo=one, out={"ff": ff[l:l + 1], "temp": th[l:l + 1], "hum": rh[l:l + 1]},
                                             enqueue_counts=call[5 * l:5 * l + 5])
                ctx.counts_accumulate(False)
            for x in (ff, rh, th):
                x.zero_()
            rows.append(("... one fill of all counters, %d lane(s)" % lanes, ) + timed(gl.launch))
            torch.cuda.synchronize()
            ok_graph = ok_graph and all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip((ff, rh, th), want))
            gl.close()
        print("fused ff + RH + theta, 1440x720x%d, ALL_DEFINED, cold (4.4 GB of fields); GPU ms between events / host wall ms; %% of 8 TB/s on 28 B/cell" % nlev)
        for name, ms, wall in rows:
            print("%-46s %8.3f ms %8.3f ms wall %6.1f %%   %6.2f us per level" % (name, ms, wall, alg / ms / 1e6 / 8000 * 100, ms / nlev * 1e3))
        print("host time per single-level enqueue: %.1f us through the wrapper, %.1f us prepared" % (host_us_wrapped, host_us))
        print("per-level loop == batch: %s; graph replay == batch: %s" % (ok_loop, ok_graph))
        g.close()
        if not (ok_loop and ok_graph):
            sys.exit(1)


if __name__ == "__main__":
    main()
