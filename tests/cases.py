"""Seeded parity cases shared by the CPU tests (restatement vs real reference),
the GPU tests (HIP vs restatement) and tests/golden/make_golden.py.

A case is a dict: op, nx, ny, args (reference argument order between (nx, ny)
and the outputs), fdefined (input flag), undef, label.
"""
import itertools

import numpy as np

import mi_fieldcalc_amd.synth as synth

ALL_DEFINED, NONE_DEFINED, SOME_DEFINED = 0, 1, 2
UNDEF = np.float32(1.0e35)

# (nx, ny): tiny, odd, ragged (nx % 4 != 0), vector-friendly multi-wave-column
GRIDS_STENCIL = [(3, 3), (5, 4), (8, 3), (17, 9), (64, 48), (260, 11), (516, 37)]
GRIDS_EWISE = [(1, 1), (5, 4), (17, 9), (64, 48), (129, 3)]

MODES = ("all", "some", "none", "lie")


def _apply_mode(fields, mode, seed, frac):
    """Returns (fields, input flag).  'lie' = flag says ALL_DEFINED but undef/NaN are present."""
    if mode == "all":
        return fields, ALL_DEFINED
    if mode == "none":
        return [np.full_like(f, UNDEF) for f in fields], SOME_DEFINED
    out = [synth.sprinkle_undef(f, seed + 31 * k, frac=frac) for k, f in enumerate(fields)]
    return out, (SOME_DEFINED if mode == "some" else ALL_DEFINED)


def _frac(nx, ny):
    return 0.15 if nx * ny < 200 else 0.02


def stencil_cases(grids=GRIDS_STENCIL, modes=MODES):
    cases = []
    for (nx, ny), mode in itertools.product(grids, modes):
        seed = 1000 * nx + ny
        xm, ym, fc = synth.grid_maps(nx, ny)
        u, v = synth.wind(nx, ny, seed)
        z = synth.scalar_field(nx, ny, seed + 5)
        (u_, v_, z_), flag = _apply_mode([u, v, z], mode, seed, _frac(nx, ny))
        base = dict(nx=nx, ny=ny, fdefined=flag, undef=UNDEF)
        lab = "%dx%d-%s" % (nx, ny, mode)
        cases.append(dict(base, op="relvort", args=[u_, v_, xm, ym], label="relvort-" + lab))
        cases.append(dict(base, op="divergence", args=[u_, v_, xm, ym], label="divergence-" + lab))
        cases.append(dict(base, op="absvort", args=[u_, v_, xm, ym, fc], label="absvort-" + lab))
        for c in (1, 2, 3, 4):
            cases.append(dict(base, op="gradient", args=[z_, xm, ym, c], label="gradient%d-%s" % (c, lab)))
        cases.append(dict(base, op="plevelgwind_xcomp", args=[z_, xm, ym, fc], label="gwindx-" + lab))
        cases.append(dict(base, op="plevelgwind_ycomp", args=[z_, xm, ym, fc], label="gwindy-" + lab))
        cases.append(dict(base, op="plevelgvort", args=[z_, xm, ym, fc], label="gvort-" + lab))
        cases.append(dict(base, op="ilevelgwind", args=[z_, xm, ym, fc], label="igwind-" + lab))
        # SURVEY.md 8f-1
        cases.append(dict(base, op="advection", args=[z_, u_, v_, xm, ym, 1.0 / 3600.0], label="advection-" + lab))
        cases.append(dict(base, op="jacobian", args=[z_, u_, xm, ym], label="jacobian-" + lab))
        cases.append(dict(base, op="thermalFrontParameter", args=[z_, xm, ym], label="tfp-" + lab))
        cases.append(dict(base, op="momentumXcoordinate", args=[v_, xm, fc, 2.0e-5], label="momx-" + lab))
        cases.append(dict(base, op="momentumYcoordinate", args=[u_, ym, fc, -3.0e-5], label="momy-" + lab))
        tq = (250.0 + 0.05 * (z_ - 5500.0)).astype(np.float32) if mode in ("all",) else np.where((z_ == UNDEF) | np.isnan(z_), z_, 250.0 + 0.05 * (z_ - 5500.0)).astype(np.float32)
        for c in (0, 1, 2, 3, 4, 5):
            cases.append(dict(base, op="plevelqvector", args=[z_, tq, xm, ym, fc, 500.0, c], label="qvector%d-%s" % (c, lab)))
    # invalid sizes / compute -> the operator returns false
    xm, ym, fc = synth.grid_maps(4, 2)
    u, v = synth.wind(4, 2, 1)
    cases.append(dict(op="relvort", nx=4, ny=2, args=[u, v, xm, ym], fdefined=SOME_DEFINED, undef=UNDEF, label="relvort-too-small"))
    xm, ym, fc = synth.grid_maps(5, 4)
    z = synth.scalar_field(5, 4, 3)
    cases.append(dict(op="gradient", nx=5, ny=4, args=[z, xm, ym, 5], fdefined=SOME_DEFINED, undef=UNDEF, label="gradient-bad-compute"))
    cases.append(dict(op="plevelqvector", nx=5, ny=4, args=[z, z, xm, ym, fc, 0.0, 1], fdefined=SOME_DEFINED, undef=UNDEF, label="qvector-bad-p"))
    return cases


def ewise_cases(grids=GRIDS_EWISE, modes=MODES):
    cases = []
    for (nx, ny), mode in itertools.product(grids, modes):
        seed = 77 * nx + ny
        u, v = synth.wind(nx, ny, seed)
        t, q, ps = synth.thermo(nx, ny, seed)
        # a few temperatures outside the ewt table (-100..+100 C) and on bin edges
        t = t.copy()
        flat = t.reshape(-1)
        if flat.size >= 8:
            flat[1] = 273.15 + 100.0
            flat[2] = 273.15 - 100.0
            flat[3] = 150.0
            flat[4] = 400.0
            flat[5] = 273.15 + 25.0
        rh = synth.uniform((ny, nx), seed + 9, 0.5, 110.0).astype(np.float32)  # % (beyond 100 exercises clamp_rh)
        td = (t - synth.uniform((ny, nx), seed + 10, 0.0, 25.0)).astype(np.float32)
        theta = (t * 1.05).astype(np.float32)
        p3 = synth.uniform((ny, nx), seed + 11, 150.0, 1040.0).astype(np.float32)
        (u_, v_, t_, q_, ps_, rh_, td_, th_, p3_), flag = _apply_mode([u, v, t, q, ps, rh, td, theta, p3], mode, seed, _frac(nx, ny))
        base = dict(nx=nx, ny=ny, fdefined=flag, undef=UNDEF)
        lab = "%dx%d-%s" % (nx, ny, mode)
        cases.append(dict(base, op="vectorabs", args=[u_, v_], label="vectorabs-" + lab))
        units = ("celsius", "kelvin", "")
        for c, unit in itertools.product(range(0, 7), units):
            tin = th_ if c in (1, 2, 5) else t_
            cases.append(dict(base, op="pleveltemp", args=[tin, 850.0, unit, c], label="pleveltemp%d%s-%s" % (c, unit, lab)))
            cases.append(dict(base, op="hleveltemp", args=[tin, ps_, 12.5, 0.73, unit, c], label="hleveltemp%d%s-%s" % (c, unit, lab)))
            cases.append(dict(base, op="aleveltemp", args=[tin, p3_, unit, c], label="aleveltemp%d%s-%s" % (c, unit, lab)))
        for c, unit in itertools.product(range(0, 14), units):
            # plevelhum numbering: 1,2 q->RH; 3,4 RH->q; 5,6,9,10 RH->Td; 7,8,11,12 q->Td
            tin = th_ if c % 2 == 0 else t_
            hp = q_ if c in (1, 2, 7, 8, 11, 12) else rh_
            cases.append(dict(base, op="plevelhum", args=[tin, hp, 700.0, unit, c], label="plevelhum%d%s-%s" % (c, unit, lab)))
            # a/hlevelhum numbering: 1,2 q->RH; 3,4 RH->q; 5,6,9,10 q->Td; 7,8,11,12 RH->Td
            ha = q_ if c in (1, 2, 5, 6, 9, 10) else rh_
            cases.append(dict(base, op="hlevelhum", args=[tin, ha, ps_, 12.5, 0.73, unit, c], label="hlevelhum%d%s-%s" % (c, unit, lab)))
            cases.append(dict(base, op="alevelhum", args=[tin, ha, p3_, unit, c], label="alevelhum%d%s-%s" % (c, unit, lab)))
        for c, unit in itertools.product(range(0, 7), ("celsius", "kelvin", "1", "")):
            if c in (4, 5):
                tin = t_ if c == 4 else (t_ - np.float32(273.15)).astype(np.float32)
                h = td_ if c == 4 else (td_ - np.float32(273.15)).astype(np.float32)
            else:
                tin = (t_ - np.float32(273.15)).astype(np.float32) if c == 3 else t_
                h = rh_
            if mode in ("some", "lie", "none"):  # keep the sprinkled undef intact after the offset
                tin = np.where((t_ == UNDEF) | np.isnan(t_), t_, tin).astype(np.float32)
                h = np.where((td_ == UNDEF) | np.isnan(td_), td_, h).astype(np.float32) if c in (4, 5) else h
            cases.append(dict(base, op="cvhum", args=[tin, h, unit, c], label="cvhum%d%s-%s" % (c, unit, lab)))
    # argument validation
    t, q, ps = synth.thermo(5, 4, 1)
    base = dict(nx=5, ny=4, fdefined=SOME_DEFINED, undef=UNDEF)
    cases.append(dict(base, op="pleveltemp", args=[t, -1.0, "kelvin", 3], label="pleveltemp-badp"))
    cases.append(dict(base, op="plevelhum", args=[t, q, 0.0, "kelvin", 1], label="plevelhum-badp"))
    cases.append(dict(base, op="plevelhum", args=[t, q, float(UNDEF), "kelvin", 1], label="plevelhum-undefp"))
    cases.append(dict(base, op="plevelhum", args=[t, q, float(UNDEF), "kelvin", 5], label="plevelhum-undefp-rhtd"))
    for a, b in ((-1.0, 0.5), (1.0, -0.5), (0.0, 0.0), (1.0, 1.5)):
        cases.append(dict(base, op="hleveltemp", args=[t, ps, a, b, "kelvin", 3], label="hleveltemp-badlevel%g%g" % (a, b)))
        cases.append(dict(base, op="hlevelhum", args=[t, q, ps, a, b, "kelvin", 1], label="hlevelhum-badlevel%g%g" % (a, b)))
    return cases


N_OUT = {"ilevelgwind": 2}


def run_cpu(lib, case, prefill=None):
    """Runs one case through a CpuLib; outputs start from a recognisable fill so
    that 'cells left unwritten' are compared too."""
    n_out = N_OUT.get(case["op"], 1)
    fill = np.float32(-7777.0) if prefill is None else prefill
    outs = [np.full((case["ny"], case["nx"]), fill, dtype=np.float32) for _ in range(n_out)]
    return lib.call(case["op"], case["nx"], case["ny"], *case["args"], fdefined=case["fdefined"], undef=case["undef"], outs=outs)


def same_bits(a, b, nan_payload=True):
    """Bit-for-bit equality.  nan_payload=False: a NaN matches any NaN (sign and
    payload of an arithmetic NaN are an ISA property, x86 SSE gives 0xFFC00000,
    gfx950 0x7FC00000; NaNs only arise when a caller passes ALL_DEFINED for a
    field that holds NaN) -- used for CPU-vs-GPU comparisons only."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    ai, bi = a.view(np.uint32), b.view(np.uint32)
    if nan_payload:
        return np.array_equal(ai, bi)
    an, bn = np.isnan(a), np.isnan(b)
    return np.array_equal(an, bn) and np.array_equal(ai[~an], bi[~bn])
