"""Pins the CPU restatement (oracle/mifc_oracle.cc) against the REAL reference,
compiled from /root/reference by oracle/Makefile into oracle/_ref/ -- bit for
bit, on the seeded sweep of tests/cases.py (every hot-path operator x grid
shape x undefined-value mode x compute/unit variant, plus the argument
validation cases).  Runs on the CPU; skipped where the reference build is not
available (the GPU box only ships the prebuilt library, which still works)."""
import numpy as np
import pytest

import cases


def _check(oracle, ref, case):
    ok_r, out_r, flag_r = cases.run_cpu(ref, case)
    ok_o, out_o, flag_o = cases.run_cpu(oracle, case)
    assert ok_o == ok_r, case["label"]
    if not ok_r:
        return
    assert flag_o == flag_r, case["label"]
    outs_r = out_r if isinstance(out_r, tuple) else (out_r,)
    outs_o = out_o if isinstance(out_o, tuple) else (out_o,)
    for a, b in zip(outs_o, outs_r):
        if not cases.same_bits(a, b, nan_payload=False):  # NaN sign/payload of an arithmetic NaN depends on instruction selection
            bad = np.nonzero(a.view(np.uint32) != b.view(np.uint32))
            raise AssertionError("%s: %d cells differ, first at %s: oracle %r ref %r" % (
                case["label"], len(bad[0]), (bad[0][0], bad[1][0]), a[bad][0], b[bad][0]))


def test_kinds(oracle, ref):
    assert oracle.kind == "restatement"
    assert ref.kind.startswith("reference 0.1.")


def test_stencils_bit_exact(oracle, ref):
    cs = cases.stencil_cases()
    assert len(cs) > 300
    for case in cs:
        if case["op"] == "plevelgwind_ycomp" and (case["nx"] < 3 or case["ny"] < 3):
            continue  # the reference has no guard there (UB); the restatement returns false
        _check(oracle, ref, case)


def test_two_pass_stencils_bit_exact(oracle, ref):
    """thermalFrontParameter / plevelqvector where the flag handed from pass to pass matters
    (clean first pass, zero-gradient plateaus, a NaN gradient from defined inputs)."""
    cs = cases.fused2_cases()
    assert len(cs) > 250
    for case in cs:
        _check(oracle, ref, case)


def test_elementwise_bit_exact(oracle, ref):
    cs = cases.ewise_cases()
    assert len(cs) > 1500
    for case in cs:
        _check(oracle, ref, case)


def test_catalogue_bit_exact(oracle, ref):
    """theta-e, ducting, indices, conversions, field algebra."""
    cs = cases.catalogue_cases()
    assert len(cs) > 1500
    for case in cs:
        _check(oracle, ref, case)


def test_ensemble_reductions_bit_exact(oracle, ref):
    """SURVEY.md 8f-4."""
    cs = cases.ensemble_cases()
    assert len(cs) > 400
    for case in cs:
        _check(oracle, ref, case)


def test_headline_level_bit_exact(oracle, ref):
    """One full 1440x720 level of the headline configuration, all-defined and with undefined cells."""
    import mi_fieldcalc_amd.synth as synth

    nx, ny = 1440, 720
    xm, ym, fc = synth.grid_maps(nx, ny)
    u, v = synth.wind(nx, ny, 0x5EED0000 + 3000)
    for mode in ("all", "some"):
        (u_, v_), flag = cases._apply_mode([u, v], mode, 99, 0.01)
        for op in ("relvort", "divergence"):
            _check(oracle, ref, dict(op=op, nx=nx, ny=ny, args=[u_, v_, xm, ym], fdefined=flag, undef=cases.UNDEF, label="%s-1440x720-%s" % (op, mode)))
