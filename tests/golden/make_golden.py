#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref/libmifc_ref.so,
compiled from /root/reference by oracle/Makefile).  Run in the build container:

    make -C oracle && python tests/golden/make_golden.py

Each fixture stores, per seeded case of tests/cases.py (subset of small grids):
the reference's return value, output flag, output field(s) and a SHA-256 of
the inputs (so a drift of the input generator is noticed instead of silently
comparing different inputs).  One full 1440x720 level of the headline
configuration is stored as SHA-256 digests of the outputs only.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cases  # noqa: E402
from cpulib import CpuLib  # noqa: E402

GOLDEN_STENCIL_GRIDS = [(3, 3), (5, 4), (17, 9), (64, 48), (260, 11)]
GOLDEN_EWISE_GRIDS = [(1, 1), (5, 4), (17, 9)]
GOLDEN_CATALOGUE_GRIDS = [(5, 4), (17, 9)]
GOLDEN_ENSEMBLE_GRIDS = [(5, 4), (17, 9)]


def input_digest(case):
    h = hashlib.sha256()
    def feed(a):
        if isinstance(a, np.ndarray):
            h.update(np.ascontiguousarray(a, dtype=np.float32).tobytes())
        elif isinstance(a, (list, tuple)):  # table of member fields, per-member flags, limits
            h.update(b"[%d]" % len(a))
            for x in a:
                feed(x)
        else:
            h.update(repr(a).encode())

    for a in case["args"]:
        feed(a)
    h.update(repr((case["nx"], case["ny"], case["fdefined"], float(case["undef"]))).encode())
    return h.hexdigest()


def build(ref, cs):
    store = {}
    labels = []
    for case in cs:
        if case["op"] == "plevelgwind_ycomp" and (case["nx"] < 3 or case["ny"] < 3):
            continue
        ok, out, flag = cases.run_cpu(ref, case)
        lab = case["label"]
        assert lab not in labels, lab
        labels.append(lab)
        outs = out if isinstance(out, tuple) else (out,)
        store[lab + "/meta"] = np.array([int(ok), int(flag), len(outs)], dtype=np.int32)
        store[lab + "/digest"] = np.frombuffer(bytes.fromhex(input_digest(case)), dtype=np.uint8)
        if ok:
            for k, o in enumerate(outs):
                store["%s/out%d" % (lab, k)] = o
    store["labels"] = np.array(labels)
    return store


def main():
    ref = CpuLib("ref")
    print("generating from:", ref.kind)
    st = build(ref, cases.stencil_cases(grids=GOLDEN_STENCIL_GRIDS))
    np.savez_compressed(os.path.join(HERE, "stencil_golden.npz"), **st)
    ew = build(ref, cases.ewise_cases(grids=GOLDEN_EWISE_GRIDS))
    np.savez_compressed(os.path.join(HERE, "ewise_golden.npz"), **ew)
    cat = build(ref, cases.catalogue_cases(grids=GOLDEN_CATALOGUE_GRIDS))
    np.savez_compressed(os.path.join(HERE, "catalogue_golden.npz"), **cat)
    ens = build(ref, cases.ensemble_cases(grids=GOLDEN_ENSEMBLE_GRIDS))
    np.savez_compressed(os.path.join(HERE, "ensemble_golden.npz"), **ens)

    # headline level, digests only
    import mi_fieldcalc_amd.synth as synth

    nx, ny = 1440, 720
    xm, ym, fc = synth.grid_maps(nx, ny)
    u, v = synth.wind(nx, ny, 0x5EED0000 + 3000)
    big = {}
    for mode in ("all", "some"):
        (u_, v_), flag = cases._apply_mode([u, v], mode, 99, 0.01)
        for op in ("relvort", "divergence"):
            case = dict(op=op, nx=nx, ny=ny, args=[u_, v_, xm, ym], fdefined=flag, undef=cases.UNDEF, label="%s-%s" % (op, mode))
            ok, out, oflag = cases.run_cpu(ref, case)
            big["%s-%s/meta" % (op, mode)] = np.array([int(ok), int(oflag)], dtype=np.int32)
            big["%s-%s/in" % (op, mode)] = np.frombuffer(bytes.fromhex(input_digest(case)), dtype=np.uint8)
            big["%s-%s/out" % (op, mode)] = np.frombuffer(hashlib.sha256(out.tobytes()).digest(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "headline_level_digests.npz"), **big)
    for f in ("stencil_golden.npz", "ewise_golden.npz", "catalogue_golden.npz", "ensemble_golden.npz", "headline_level_digests.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
