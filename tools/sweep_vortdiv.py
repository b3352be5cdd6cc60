#!/usr/bin/env python3
"""Tuning sweep of the fused vorticity+divergence kernel on the headline
configuration (1440x720x137): interleaved rounds in ONE process (methodology
rule 24 of the CDNA guide), median / min kernel time per tuning string, with a
plain device-to-device copy of the same byte volume as the achievable-bandwidth
yardstick.  Usage (GPU box):  python tools/sweep_vortdiv.py "R=32,D=2" "R=16,D=2" ...
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the measurement build of the library: the only one that contains the XH / XS / XL / PADROWS knobs
os.environ.setdefault("MIFC_LIB_PATH", os.path.join(ROOT, "mi-fieldcalc_amd", "libmifc_measure.so"))

import torch  # noqa: E402

import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402

NX, NY, NLEV = (int(x) for x in os.environ.get("SWEEP_SHAPE", "1440,720,137").split(","))
ROUNDS = int(os.environ.get("SWEEP_ROUNDS", "7"))
INNER = int(os.environ.get("SWEEP_INNER", "5"))
# yardstick kernels: name -> (variant, blocks); blocks 0 = one lane per 16 B
YARD = {
    "stream2 plain, 1 lane/16B": (0, 0),
    "stream2 plain, 2048 blocks": (0, 2048),
    "stream2 plain, 8192 blocks": (0, 8192),
    "stream2 nt-store, 1 lane/16B": (1, 0),
    "stream2 nt-ld+st, 1 lane/16B": (2, 0),
    "stream2 nt-ld+st, 4096 blk": (2, 4096),
    "split-role loop copy, 1024 blk": (3, 1024),
    "split-role loop copy, 2048 blk": (3, 2048),
    "split-role loop copy, 4096 blk": (3, 4096),
    "split-role copy, 1 tile pair/blk": (3, 0),
    # every workgroup streams one contiguous chunk (row-wide workgroups walking down a band of rows)
    "band copy, 64-row chunks": (4, 1542),
    "band copy, 32-row chunks": (4, 3083),
    "band copy, 16-row chunks": (4, 6165),
    "band copy, 8-row chunks": (4, 12330),
    # write-only yardsticks (half the bytes of the copies: the two outputs only)
    "fill both outputs, linear one-shot": (5, 0),
    "fill both outputs, 4x256-col tiles one-shot": (6, 0),
    "fill both outputs, waves loop over 8 rows": (7, 0),
    "fill both outputs, 6-wave workgroups write full rows, 8 rows": (8, 0),
}


def main():
    tunes = sys.argv[1:] or ["R=8"]
    dev = torch.device("cuda", 0)
    xm, ym, _ = synth.grid_maps(NX, NY)
    dxm, dym = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
    du, dv = synth.device_wind(NX, NY, NLEV, 1234, dev)
    rv, dg = torch.empty_like(du), torch.empty_like(du)
    flags = np.full(NLEV, fc.ALL_DEFINED, np.int32)
    ctx = fc.Context(0)
    alg = NX * NY * NLEV * 16 + 2 * NX * NY * 4
    results = {t: [] for t in tunes}
    copy_ms = []
    yard_ms = {k: [] for k in YARD}

    def run(tune):
        os.environ["MIFC_VORTDIV_TUNE"] = tune
        ctx.reload_env()
        assert ctx.vortdiv_levels_enqueue(du, dv, dxm, dym, rv, dg, fdefined=flags)

    for t in tunes:  # warm-up + first-use
        run(t)
    torch.cuda.synchronize()
    for r in range(ROUNDS):
        for t in tunes:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(INNER):
                run(t)
            e.record()
            torch.cuda.synchronize()
            results[t].append(s.elapsed_time(e) / INNER)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(INNER):
            rv.copy_(du)
            dg.copy_(dv)
        e.record()
        torch.cuda.synchronize()
        copy_ms.append(s.elapsed_time(e) / INNER)
        for key in (YARD if not os.environ.get("SWEEP_NO_YARD") else {"stream2 plain, 1 lane/16B": (0, 0), "stream2 nt-ld+st, 1 lane/16B": (2, 0)}):
            variant, blocks = YARD[key]
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(INNER):
                ctx.bench_stream2(variant, blocks, rv, dg, du, dv)
            e.record()
            torch.cuda.synchronize()
            yard_ms[key].append(s.elapsed_time(e) / INNER)
    print("shape %dx%dx%d, algorithmic bytes %.3f GB, %d rounds x %d launches" % (NX, NY, NLEV, alg / 1e9, ROUNDS, INNER))
    print("%-28s %9s %9s %9s %7s" % ("tuning", "med ms", "min ms", "GB/s(med)", "frac"))
    for t in tunes:
        med, mn = float(np.median(results[t])), float(np.min(results[t]))
        print("%-28s %9.4f %9.4f %9.1f %7.4f" % (t, med, mn, alg / med / 1e6, alg / med / 1e6 / 8000.0))
    cb = NX * NY * NLEV * 16
    med = float(np.median(copy_ms))
    print("%-28s %9.4f %9.4f %9.1f %7.4f   (2x torch copy_, same u+v -> 2 outputs bytes)" % ("d2d copy", med, float(np.min(copy_ms)), cb / med / 1e6, cb / med / 1e6 / 8000.0))
    for key in YARD:
        if not yard_ms[key]:
            continue
        med = float(np.median(yard_ms[key]))
        print("%-28s %9.4f %9.4f %9.1f %7.4f   (2-in/2-out float4 stream, no arithmetic)" % (key, med, float(np.min(yard_ms[key])), cb / med / 1e6, cb / med / 1e6 / 8000.0))


if __name__ == "__main__":
    main()
