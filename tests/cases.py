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
        cases.append(dict(base, op="shapiro2_filter", args=[z_], label="shapiro2-" + lab))
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


def fused2_cases():
    """thermalFrontParameter / plevelqvector on grids the single-launch kernel takes (nx % 4 == 0):
    one column group wide, one band, many bands, the headline width; the usual flag modes plus
    the ones that matter to the pass-to-pass flag logic of the reference (:2286, :664)."""
    out = []
    # ... and widths around the 240-column tiles of the kernel: one partial tile, exactly one, one column
    # group into the second, two and a bit, many
    for nx, ny in [(4, 3), (4, 9), (8, 3), (12, 20), (64, 48), (128, 301), (236, 7), (240, 9), (244, 12), (484, 5), (1440, 37), (4096, 6),
                   (5000, 4)]:
        xm, ym, fc = synth.grid_maps(nx, ny)
        seed = 31 * nx + ny
        z = synth.scalar_field(nx, ny, seed)
        variants = []
        for mode in MODES:
            (z_,), flag = _apply_mode([z], mode, seed, _frac(nx, ny))
            variants.append((mode, z_, flag, xm))
        # every value defined, but the caller does not promise it: the first pass of TFP
        # finds nothing and hands ALL_DEFINED to the second
        variants.append(("clean-some", z, SOME_DEFINED, xm))
        # a plateau: |grad| == 0 cells are rejected whatever the flag says (:2292)
        zp = z.copy()
        zp[ny // 3:, : max(3, nx // 2)] = np.float32(5432.0)
        variants.append(("plateau-all", zp, ALL_DEFINED, xm))
        variants.append(("plateau-some", zp, SOME_DEFINED, xm))
        if nx >= 8 and ny >= 9:
            # defined inputs whose gradient is NaN (0 * inf): with a clean first pass the
            # reference computes with it, a tested second pass would not
            zn, xn = z.copy(), xm.copy()
            j, i = ny // 2, nx // 2
            zn[j, i - 1], zn[j, i + 1] = np.float32(-3.0e38), np.float32(3.0e38)
            xn[j, i] = np.float32(0.0)
            variants.append(("nan-gradient-some", zn, SOME_DEFINED, xn))
            variants.append(("nan-gradient-all", zn, ALL_DEFINED, xn))
        if ny >= 7:
            # map factors below 2^-125 in two rows (subnormal, odd mantissas: 0.5 * m is not exact) next to ordinary rows,
            # with a field large enough there that the products are ordinary numbers again: the one-launch kernels halve
            # a row of map factors once when they can and must take the generic products for these rows and for the
            # iterations that mix them with halved rows
            zt, xt, yt = z.copy(), xm.copy(), ym.copy()
            j = ny // 2
            tiny = np.float32(2.0) ** np.float32(-110)
            xt[j:j + 2] = (xt[j:j + 2] * tiny).astype(np.float32)
            yt[j + 1:j + 2] = (yt[j + 1:j + 2] * tiny).astype(np.float32)
            zt[j - 1:j + 3] = (zt[j - 1:j + 3] * np.float32(2.0) ** np.float32(60)).astype(np.float32)
            variants.append(("tiny-maps-all", zt, ALL_DEFINED, xt, yt))
            variants.append(("tiny-maps-some", zt, SOME_DEFINED, xt, yt))
        for name, z_, flag, xm_, *rest in variants:
            ym_ = rest[0] if rest else ym
            base = dict(nx=nx, ny=ny, fdefined=flag, undef=UNDEF)
            lab = "%dx%d-%s" % (nx, ny, name)
            out.append(dict(base, op="thermalFrontParameter", args=[z_, xm_, ym_], label="tfp-" + lab))
            bad = (z_ == UNDEF) | np.isnan(z_)
            with np.errstate(all="ignore"):
                tq = np.where(bad, z_, np.float32(250.0) + np.float32(0.05) * (z_ - np.float32(5500.0))).astype(np.float32)
            for c in (1, 2, 3, 4):
                out.append(dict(base, op="plevelqvector", args=[z_, tq, xm_, ym_, fc, 700.0, c], label="qvector%d-%s" % (c, lab)))
    return out


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


def catalogue_cases(grids=GRIDS_EWISE, modes=MODES):
    """the rest of the pointwise catalogue."""
    cases = []
    for (nx, ny), mode in itertools.product(grids, modes):
        seed = 313 * nx + ny
        u, v = synth.wind(nx, ny, seed)
        u5, v5 = synth.wind(nx, ny, seed + 1)
        t, q, ps = synth.thermo(nx, ny, seed)
        t = t.copy()
        flat = t.reshape(-1)
        if flat.size >= 8:  # outside / on the edge of the ewt table
            flat[1] = 273.15 + 100.0
            flat[2] = 273.15 - 100.0
            flat[3] = 150.0
            flat[4] = 400.0
        shape = (ny, nx)
        rh = synth.uniform(shape, seed + 9, 0.5, 110.0).astype(np.float32)
        rh7 = synth.uniform(shape, seed + 12, 0.5, 110.0).astype(np.float32)
        theta = (t * 1.05).astype(np.float32)
        p3 = synth.uniform(shape, seed + 11, 150.0, 1040.0).astype(np.float32)
        t5 = (t - 30.0).astype(np.float32)
        t7 = (t - 12.0).astype(np.float32)
        td = (t - synth.uniform(shape, seed + 10, 0.0, 25.0)).astype(np.float32)
        td5 = (t5 - synth.uniform(shape, seed + 13, 0.0, 25.0)).astype(np.float32)
        z7 = synth.uniform(shape, seed + 14, 2700.0, 3200.0).astype(np.float32)
        z10 = synth.uniform(shape, seed + 15, -100.0, 250.0).astype(np.float32)
        sal = synth.uniform(shape, seed + 16, 5.0, 38.0).astype(np.float32)
        tsea = synth.uniform(shape, seed + 17, -1.5, 25.0).astype(np.float32)
        precip = synth.uniform(shape, seed + 18, 0.0, 6.0).astype(np.float32)
        snow = (precip * synth.uniform(shape, seed + 19, 0.0, 1.0)).astype(np.float32)
        snoww = synth.uniform(shape, seed + 20, -1.0, 30.0).astype(np.float32)
        anyf = synth.uniform(shape, seed + 21, -50.0, 50.0).astype(np.float32)
        posf = synth.uniform(shape, seed + 22, 0.001, 500.0).astype(np.float32)
        smallf = synth.uniform(shape, seed + 23, -8.0, 8.0).astype(np.float32)
        zerof = np.where(synth.uniform(shape, seed + 24, 0.0, 1.0) < 0.2, 0.0, anyf).astype(np.float32)
        fields = [u, v, u5, v5, t, q, ps, rh, rh7, theta, p3, t5, t7, td, td5, z7, z10, sal, tsea, precip, snow, snoww, anyf, posf, smallf, zerof]
        (u_, v_, u5_, v5_, t_, q_, ps_, rh_, rh7_, th_, p3_, t5_, t7_, td_, td5_, z7_, z10_, sal_, tsea_, precip_, snow_, snoww_, anyf_, posf_,
         smallf_, zerof_), flag = _apply_mode(fields, mode, seed, _frac(nx, ny))
        base = dict(nx=nx, ny=ny, fdefined=flag, undef=UNDEF)
        lab = "%dx%d-%s" % (nx, ny, mode)

        def add(op, args, tag=""):
            cases.append(dict(base, op=op, args=args, label="%s%s-%s" % (op, tag, lab)))

        for c in (0, 1, 2, 3):
            add("plevelthe", [th_ if c == 2 else t_, rh_, 850.0, c], str(c))
            add("hlevelthe", [th_ if c == 2 else t_, q_, ps_, 12.5, 0.73, c], str(c))
            add("alevelthe", [th_ if c == 2 else t_, q_, p3_, c], str(c))
            add("kIndex", [t5_, t7_, rh7_, t_, rh_, 500.0, 700.0, 850.0, c], str(c))
            add("ductingIndex", [th_ if c == 2 else t_, rh_, 850.0, c], str(c))
            add("showalterIndex", [t5_, t_, rh_, 500.0, 850.0, c], str(c))
            add("boydenIndex", [t7_, z7_, z10_, 700.0, 1000.0, c], str(c))
            add("seaSoundSpeed", [tsea_ if c == 1 else (t_ if mode != "all" else (tsea + np.float32(273.15)).astype(np.float32)), sal_, -75.0, c], str(c))
            add("windCooling", [t_ if c == 1 else tsea_, u_, v_, c], str(c))
        for c in (0, 1, 2, 3, 4, 5):
            hum = q_ if c in (1, 2) else rh_
            add("plevelducting", [th_ if c % 2 == 0 else t_, hum, 925.0, c], str(c))
            add("hlevelducting", [th_ if c % 2 == 0 else t_, hum, ps_, 12.5, 0.73, c], str(c))
            add("alevelducting", [th_ if c % 2 == 0 else t_, hum, p3_, c], str(c))
            add("cvtemp", [tsea_ if c in (2, 4) else t_, c], str(c))
            add("fieldOPERfield", [c, anyf_, zerof_], str(c))
            add("fieldOPERconstant", [c, anyf_, 2.5], str(c))
            add("fieldOPERconstant", [c, anyf_, 0.0], "%dzero" % c)
            add("constantOPERfield", [c, 2.5, zerof_], str(c))
        add("cvtemp", [t_, 4], "4-looks-like-kelvin")  # mean > t0/2: input copied, flag untouched
        add("cvtemp", [tsea_, 3], "3-looks-like-celsius")
        add("fieldOPERconstant", [1, anyf_, float(UNDEF)], "-undef-constant")
        add("constantOPERfield", [9, float(UNDEF), anyf_], "-undef-constant-bad-compute")
        add("hlevelpressure", [ps_, 12.5, 0.73])
        for c in (0, 1, 2, 3, 4):
            add("pleveldz2tmean", [z7_, z10_, 700.0, 1000.0, c], str(c))
        add("sweatIndex", [t_, t5_, td_, td5_, u_, v_, u5_, v5_])
        add("abshum", [t_, rh_])
        add("underCooledRain", [precip_, snow_, t_, 0.5, 0.3, 1.0])
        add("pressure2FlightLevel", [p3_])
        add("snow_in_cm", [snoww_, t_, td_])
        # vessel icing: air / sea temperature in Celsius, storm-force winds so that every Mertins class occurs
        tair = synth.uniform(shape, seed + 30, -25.0, 3.0).astype(np.float32)
        tsst = synth.uniform(shape, seed + 31, -3.0, 8.0).astype(np.float32)
        ice = synth.uniform(shape, seed + 32, 0.0, 0.8).astype(np.float32)
        (tair_, tsst_, ice_), _ = _apply_mode([tair, tsst, ice], mode, seed + 33, _frac(nx, ny))
        add("vesselIcingOverland", [tair_, tsst_, u_, v_, sal_, ice_])
        add("vesselIcingMertins", [tair_, tsst_, (u_ * np.float32(1.6)).astype(np.float32) if mode == "all" else u_, v_, sal_, ice_])
        add("values2classes", [anyf_, [-40.0, -10.0, 0.0, 5.0, 20.0, 45.0]])
        add("values2classes", [anyf_, [-40.0, 45.0]], "-two")
        add("values2classes", [anyf_, [1.0]], "-too-few")
        add("minvalueFields", [anyf_, smallf_])
        add("maxvalueFields", [anyf_, smallf_])
        for val, tag in ((3.5, ""), (float(UNDEF), "-undef")):
            add("minvalueFieldConst", [anyf_, val], tag)
            add("maxvalueFieldConst", [anyf_, val], tag)
            add("powerField", [posf_, 0.37 if tag == "" else val], tag)
            add("replaceUndefined", [anyf_, val], tag)
            add("replaceDefined", [anyf_, val], tag)
        add("absvalueField", [anyf_])
        add("log10Field", [posf_])
        add("logField", [posf_])
        add("pow10Field", [smallf_])
        add("expField", [smallf_])
        # the flag is an input of replaceUndefined / replaceDefined: all three values
        for f in (ALL_DEFINED, NONE_DEFINED, SOME_DEFINED):
            cases.append(dict(base, fdefined=f, op="replaceUndefined", args=[anyf_, -1.0], label="replaceUndefined-flag%d-%s" % (f, lab)))
            cases.append(dict(base, fdefined=f, op="replaceDefined", args=[anyf_, -1.0], label="replaceDefined-flag%d-%s" % (f, lab)))
    # argument validation
    t, q, ps = synth.thermo(5, 4, 1)
    base = dict(nx=5, ny=4, fdefined=SOME_DEFINED, undef=UNDEF)
    cases.append(dict(base, op="plevelthe", args=[t, q, 0.0, 1], label="plevelthe-badp"))
    cases.append(dict(base, op="plevelducting", args=[t, q, -5.0, 1], label="plevelducting-badp"))
    cases.append(dict(base, op="kIndex", args=[t, t, q, t, q, 700.0, 700.0, 850.0, 1], label="kIndex-bad-levels"))
    cases.append(dict(base, op="showalterIndex", args=[t, t, q, 850.0, 500.0, 1], label="showalter-bad-levels"))
    cases.append(dict(base, op="boydenIndex", args=[t, t, t, 1000.0, 700.0, 1], label="boyden-bad-levels"))
    cases.append(dict(base, op="ductingIndex", args=[t, q, 0.0, 1], label="ductingIndex-badp"))
    cases.append(dict(base, op="pleveldz2tmean", args=[t, t, 500.0, 500.0, 1], label="dz2tmean-same-p"))
    for a, b in ((-1.0, 0.5), (0.0, 0.0), (1.0, 1.5)):
        cases.append(dict(base, op="hlevelthe", args=[t, q, ps, a, b, 1], label="hlevelthe-badlevel%g%g" % (a, b)))
        cases.append(dict(base, op="hlevelducting", args=[t, q, ps, a, b, 1], label="hlevelducting-badlevel%g%g" % (a, b)))
        cases.append(dict(base, op="hlevelpressure", args=[ps, a, b], label="hlevelpressure-badlevel%g%g" % (a, b)))
    return cases


def ensemble_cases(grids=((5, 4), (17, 9), (64, 48)), modes=MODES):
    """reductions over ensemble members."""
    cases = []
    for (nx, ny), mode, nmem in itertools.product(grids, modes, (1, 3, 8)):
        seed = 911 * nx + ny + nmem
        members = [synth.uniform((ny, nx), seed + 7 * k, -5.0, 30.0).astype(np.float32) for k in range(nmem)]
        members, flag = _apply_mode(members, mode, seed, 0.3 if mode != "none" else 1.0)
        flags_in = [flag] * nmem
        if nmem >= 3:
            flags_in[1] = NONE_DEFINED  # probability() skips such members
            if mode == "all":
                flags_in[2] = SOME_DEFINED
        base = dict(nx=nx, ny=ny, fdefined=flag, undef=UNDEF)
        lab = "%dx%d-%s-m%d" % (nx, ny, mode, nmem)
        cases.append(dict(base, op="sumFields", args=[members], label="sumFields-" + lab))
        cases.append(dict(base, op="meanValue", args=[members, flags_in], label="meanValue-" + lab))
        cases.append(dict(base, op="stddevValue", args=[members, flags_in], label="stddevValue-" + lab))
        for c in (0, 1, 2, 3, 4):
            cases.append(dict(base, op="extremeValue", args=[c, members], label="extremeValue%d-%s" % (c, lab)))
        for c in (1, 2, 3, 4, 5, 6):
            cases.append(dict(base, op="probability", args=[c, members, flags_in, [5.0, 20.0]], label="probability%d-%s" % (c, lab)))
        cases.append(dict(base, op="probability", args=[3, members, flags_in, [5.0]], label="probability-between-one-limit-" + lab))
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
