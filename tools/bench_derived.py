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
    sets = []
    for s in range(2):
        u, v = synth.device_wind(NX, NY, NLEV, 100 + s, dev)
        t, q, ps = synth.device_thermo(NX, NY, NLEV, 200 + s, dev)
        sets.append(dict(u=u, v=v, t=t, q=q, ps=ps, out={k: torch.empty_like(u) for k in ("ff", "temp", "hum", "hum2", "dd")}))
    cnt = torch.zeros(5 * NLEV, dtype=torch.int64, device=dev)
    n = NX * NY * NLEV
    print("%dx%dx%d, kernel ms by HIP events (median of 9), two rotating buffer sets" % (NX, NY, NLEV))
    print("%-34s %-14s %8s %9s %7s" % ("outputs", "flags / blocks", "ms", "GB/s", "frac"))
    for blocks in os.environ.get("DERIVED_SWEEP", ",4096,8192,16384,32768,65536,140000").split(","):
        if blocks:
            os.environ["MIFC_DERIVED_BLOCKS"] = blocks
        else:
            os.environ.pop("MIFC_DERIVED_BLOCKS", None)
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
