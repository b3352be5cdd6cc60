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
    print("%dx%dx%d, kernel ms by HIP events (median of 9), two rotating buffer sets" % (NX, NY, NLEV))
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
                ms = []
                for r in range(11):
                    s = sets[r % 2]
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    assert ctx.hlevel_derived_batch(s["u"], s["v"], s["t"], s["q"], s["ps"], al, bl, fdef_wind=flags, fdef_thermo=flags, out=s["out"],
                                                    enqueue_counts=cnt, **kw) is not None
                    e1.record()
                    torch.cuda.synchronize()
                    if r >= 2:
                        ms.append(e0.elapsed_time(e1))
                med = float(np.median(ms))
                alg = n * bytes_per_cell + NX * NY * 4
                print("%-34s %-14s %8.4f %9.1f %7.4f" % (name, ("ALL" if flag == fc.ALL_DEFINED else "SOME") + " / " + (blocks or "default"), med, alg / med / 1e6, alg / med / 1e6 / 8000.0))


if __name__ == "__main__":
    main()
