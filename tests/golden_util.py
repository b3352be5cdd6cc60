"""Loads the committed golden vectors (tests/golden/*.npz, produced from the real
reference by tests/golden/make_golden.py) and pairs them with the seeded cases."""
import hashlib
import os

import numpy as np

import cases

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

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


class Golden:
    def __init__(self, fname):
        self.z = np.load(os.path.join(GOLDEN, fname), allow_pickle=False)
        self.labels = set(str(x) for x in self.z["labels"])

    def expect(self, case):
        """-> (ok, flag, [outputs]) recorded from the reference for this case."""
        lab = case["label"]
        ok, flag, n_out = (int(x) for x in self.z[lab + "/meta"])
        digest = bytes(self.z[lab + "/digest"]).hex()
        assert digest == input_digest(case), "input generator drifted for %s: regenerate tests/golden" % lab
        outs = [self.z["%s/out%d" % (lab, k)] for k in range(n_out)] if ok else []
        return bool(ok), flag, outs


def stencil_golden_cases():
    g = Golden("stencil_golden.npz")
    return g, [c for c in cases.stencil_cases(grids=GOLDEN_STENCIL_GRIDS) if c["label"] in g.labels]


def ewise_golden_cases():
    g = Golden("ewise_golden.npz")
    return g, [c for c in cases.ewise_cases(grids=GOLDEN_EWISE_GRIDS) if c["label"] in g.labels]


def catalogue_golden_cases():
    g = Golden("catalogue_golden.npz")
    return g, [c for c in cases.catalogue_cases(grids=GOLDEN_CATALOGUE_GRIDS) if c["label"] in g.labels]


def ensemble_golden_cases():
    g = Golden("ensemble_golden.npz")
    return g, [c for c in cases.ensemble_cases(grids=GOLDEN_ENSEMBLE_GRIDS) if c["label"] in g.labels]
