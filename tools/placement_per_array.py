#!/usr/bin/env python3
"""Is the placement effect a property of each ARRAY (then the four best arrays of many can be combined) or of
the combination?  16 x 4 arrays; each array's own read time (sum) and write time (fill); the kernel on every
candidate batch; then the kernel on batches composed of the individually fastest / slowest arrays.
Usage (GPU box): python tools/placement_per_array.py [candidates]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import mi_fieldcalc_amd as fc  # noqa: E402
import mi_fieldcalc_amd.synth as synth  # noqa: E402
from mi_fieldcalc_amd.placement import SPACERS_MIB  # noqa: E402

NX, NY, NLEV = 1440, 720, 137


def timed(fn, reps=5, inner=4):
    fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(inner):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms.append(s.elapsed_time(e) / inner)
    return float(np.median(ms))


def main():
    ncand = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    dev = torch.device("cuda", 0)
    ctx = fc.Context(0)
    xm, ym, _ = synth.grid_maps(NX, NY)
    dxm, dym = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
    flags = np.full(NLEV, fc.ALL_DEFINED, np.int32)
    su, sv = synth.device_wind(NX, NY, NLEV, 1234, dev)
    arrays, spacers = [], []
    for i in range(ncand):
        mib = SPACERS_MIB[i % len(SPACERS_MIB)]
        if mib:
            spacers.append(torch.empty(mib << 20, dtype=torch.uint8, device=dev))
        for _ in range(4):
            arrays.append(ctx.batch_empty(NLEV, NY, NX))
    for _ in range(20):
        arrays[0].fill_(1.0)
    torch.cuda.synchronize()
    rd = [timed(lambda a=a: torch.sum(a)) for a in arrays]
    wr = [timed(lambda a=a: a.fill_(0.5)) for a in arrays]

    def kernel(iu, iv, ir, idg):
        arrays[iu].copy_(su)
        arrays[iv].copy_(sv)
        return timed(lambda: ctx.vortdiv_levels_enqueue(arrays[iu], arrays[iv], dxm, dym, arrays[ir], arrays[idg], fdefined=flags), reps=5, inner=4)

    print("per array: read (torch.sum) ms, write (fill_) ms; per candidate batch: kernel ms")
    kb = []
    for c in range(ncand):
        k = kernel(4 * c, 4 * c + 1, 4 * c + 2, 4 * c + 3)
        kb.append(k)
        print("cand %2d  read %s  write %s  kernel %.4f  (sum of its reads+writes: %.4f)" % (
            c, " ".join("%.4f" % rd[4 * c + j] for j in range(4)), " ".join("%.4f" % wr[4 * c + j] for j in range(4)), k,
            rd[4 * c] + rd[4 * c + 1] + wr[4 * c + 2] + wr[4 * c + 3]))
    pred = [rd[4 * c] + rd[4 * c + 1] + wr[4 * c + 2] + wr[4 * c + 3] for c in range(ncand)]
    print("correlation(kernel, sum of own array times) = %.3f" % float(np.corrcoef(kb, pred)[0, 1]))
    order_r = sorted(range(len(arrays)), key=lambda i: rd[i])
    order_w = sorted(range(len(arrays)), key=lambda i: wr[i])
    best_r = order_r[:2]
    best_w = [i for i in order_w if i not in best_r][:2]
    worst_r = order_r[-2:]
    worst_w = [i for i in order_w[::-1] if i not in worst_r][:2]
    print("composed of the 2 fastest readers %s + 2 fastest writers %s: kernel %.4f" % (best_r, best_w, kernel(best_r[0], best_r[1], best_w[0], best_w[1])))
    print("composed of the 2 slowest readers %s + 2 slowest writers %s: kernel %.4f" % (worst_r, worst_w, kernel(worst_r[0], worst_r[1], worst_w[0], worst_w[1])))
    b = int(np.argmin(kb))
    print("best candidate batch %d: %.4f; its arrays with inputs and outputs swapped: %.4f" % (b, kb[b], kernel(4 * b + 2, 4 * b + 3, 4 * b, 4 * b + 1)))
    w = int(np.argmax(kb))
    print("mixed: inputs of the best batch + outputs of the worst (%d): %.4f; the other way round: %.4f" % (
        w, kernel(4 * b, 4 * b + 1, 4 * w + 2, 4 * w + 3), kernel(4 * w, 4 * w + 1, 4 * b + 2, 4 * b + 3)))


if __name__ == "__main__":
    main()
