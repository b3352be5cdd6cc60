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


def input_digest(case):
    h = hashlib.sha256()
    for a in case["args"]:
        if isinstance(a, np.ndarray):
            h.update(np.ascontiguousarray(a, dtype=np.float32).tobytes())
        else:
            h.update(repr(a).encode())
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
