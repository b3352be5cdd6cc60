"""GPU parity tests: the HIP kernels, called through the C ABI (include/mifc.h),
against the CPU restatement (oracle/) on identical seeded inputs and against
the committed golden vectors recorded from the real reference.

Bar: bit-exact for the stencil operators, vectorabs, cvhum, plevel* (pure
IEEE float/double arithmetic, table walks, powf hoisted to the host as in the
reference); <= 1e-5 relative (BASELINE.json) for the operators that evaluate
powf per cell on the device (hlevel/alevel theta, humidity from theta)."""
import os

import numpy as np
import pytest

import cases
import golden_util
import gpu_util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

ALL, NONE, SOME = cases.ALL_DEFINED, cases.NONE_DEFINED, cases.SOME_DEFINED


def _check_case(ctx, oracle, case, device=False, expected=None):
    if expected is None:
        ok_e, out_e, flag_e = cases.run_cpu(oracle, case)
        outs_e = list(out_e) if isinstance(out_e, tuple) else [out_e]
    else:
        ok_e, flag_e, outs_e = expected
    ok, out, flag = gpu_util.run_gpu(ctx, case, device=device)
    assert ok == ok_e, case["label"]
    if not ok_e:
        return
    exact = not gpu_util.uses_device_powf(case)
    outs = list(out) if isinstance(out, tuple) else [out]
    for a, b in zip(outs, outs_e):
        gpu_util.compare(case, np.asarray(a), np.asarray(b), exact)
    if exact:
        assert flag == flag_e, case["label"]
    else:
        assert flag == flag_e, "%s: flag %d vs %d" % (case["label"], flag, flag_e)


def test_native_library_is_loaded(gpu_ctx):
    import mi_fieldcalc_amd._capi as capi

    assert os.path.exists(capi.LIB_PATH)
    with open("/proc/self/maps") as f:
        assert "libmifc.so" in f.read()


def test_stencils_host_pointers(gpu_ctx, oracle):
    for case in cases.stencil_cases():
        _check_case(gpu_ctx, oracle, case, device=False)


def test_stencils_device_resident(gpu_ctx, oracle):
    cs = [c for c in cases.stencil_cases(grids=[(5, 4), (64, 48), (516, 37)])]
    for case in cs:
        _check_case(gpu_ctx, oracle, case, device=True)


@pytest.mark.parametrize("rows", ["1", "3", "8"])
def test_scalar_row_kernels_band_heights(gpu_ctx, oracle, rows, mifc_env):
    """The one-input row-walking kernels pick their band height from the launch size; every height
    gives the reference result (the seeded cases are small, so the default runs 2-row bands)."""
    mifc_env("MIFC_SCALAR_ROWS_R", rows)
    ops = ("gradient", "plevelgwind_xcomp", "plevelgwind_ycomp", "plevelgvort", "ilevelgwind")
    for case in cases.stencil_cases(grids=[(8, 3), (64, 48), (516, 37)]):
        if case["op"] in ops:
            _check_case(gpu_ctx, oracle, case, device=True)


@pytest.mark.parametrize("fused", ["1", "0", "lds", "band5"])
def test_shapiro_filter_one_launch_and_four(gpu_ctx, oracle, fused, mifc_env):
    """The four sweeps in one launch (tiles of 240 columns, bands of rows; rows in registers, or -- "lds" -- in LDS rings)
    == the sweep-by-sweep path == the reference.  "band5": the register kernel with 5-row bands, so that every band
    boundary phase of its three-times-unrolled row loop occurs."""
    import mi_fieldcalc_amd.synth as synth

    mifc_env("MIFC_SHAPIRO_FUSED", "0" if fused == "0" else "1")
    mifc_env("MIFC_SHAPIRO_REGS", "0" if fused == "lds" else "1")
    mifc_env("MIFC_FUSED2_BAND", "5" if fused == "band5" else None)
    for nx, ny in [(4, 3), (8, 5), (236, 7), (240, 9), (244, 12), (484, 5), (128, 301), (1440, 37), (5000, 4)]:
        z = synth.scalar_field(nx, ny, 3 * nx + ny)
        for mode in cases.MODES:
            (z_,), flag = cases._apply_mode([z], mode, nx + ny, cases._frac(nx, ny))
            case = dict(nx=nx, ny=ny, fdefined=flag, undef=cases.UNDEF, op="shapiro2_filter", args=[z_], label="shapiro2-%dx%d-%s" % (nx, ny, mode))
            _check_case(gpu_ctx, oracle, case, device=(nx % 8 == 0))


@pytest.mark.parametrize("tune", ["K=1", "K=2", "R=8", "R=2,WPB=1"])
def test_single_field_wind_operators_in_every_kernel_form(gpu_ctx, oracle, tune, mifc_env):
    """relvort / divergence / absvort / jacobian through the C ABI's single-field calls: the launcher picks the
    one-shot, the one-shot-tile or the row-walking form by launch size; forced here, all give the reference."""
    mifc_env("MIFC_VORTDIV_TUNE", tune)
    ops = ("relvort", "divergence", "absvort", "jacobian")
    for case in cases.stencil_cases(grids=[(8, 3), (64, 48), (260, 11), (516, 37)]):
        if case["op"] in ops:
            _check_case(gpu_ctx, oracle, case, device=True)


def test_elementwise_host_pointers(gpu_ctx, oracle):
    for case in cases.ewise_cases():
        _check_case(gpu_ctx, oracle, case, device=False)


def test_elementwise_device_resident(gpu_ctx, oracle):
    for case in cases.ewise_cases(grids=[(17, 9), (64, 48)], modes=("all", "some")):
        _check_case(gpu_ctx, oracle, case, device=True)


def test_catalogue_host_pointers(gpu_ctx, oracle):
    """theta-e, ducting, indices, conversions, field algebra."""
    cs = cases.catalogue_cases()
    assert len(cs) > 1500
    for case in cs:
        _check_case(gpu_ctx, oracle, case, device=False)


def test_catalogue_device_resident(gpu_ctx, oracle):
    for case in cases.catalogue_cases(grids=[(17, 9), (64, 48)], modes=("all", "some", "lie")):
        _check_case(gpu_ctx, oracle, case, device=True)


def test_python_surface_matches_reference_module(gpu_ctx, oracle):
    """mi_fieldcalc.py mirrors python/py_mi_fieldcalc.cc: the reference's own Python
    test (python/test_mi_fieldcalc.py:36-41: abshum(293.16 K, 0.8) = 13.83 +- 0.02),
    None on shape mismatch / non-2-D input / failing operator, float32 result."""
    import mi_fieldcalc as pyfc

    ah = pyfc.abshum(np.array([[293.16]]), np.array([[0.8]]), -1)
    assert ah is not None and ah.dtype == np.float32 and ah.shape == (1, 1)
    assert abs(float(ah[0, 0]) - 13.83) <= 0.02
    assert int(pyfc.ValuesDefined.SOME_DEFINED) == 2
    t = np.full((6, 5), 280.0)
    assert pyfc.cvtemp(t, 1, 1e35).shape == (6, 5)
    assert pyfc.cvtemp(t[0], 1, 1e35) is None  # not 2-D
    assert pyfc.abshum(t, t[:3], 1e35) is None  # shapes differ
    assert pyfc.cvtemp(t, 9, 1e35) is None  # operator returns false
    rh = np.full((6, 5), 55.0)
    td = pyfc.cvhum(t, rh, "kelvin", 1, 1e35)
    ok, expect, _ = oracle.call("cvhum", 5, 6, t.astype(np.float32), rh.astype(np.float32), "kelvin", 1, fdefined=SOME)
    assert ok and cases.same_bits(td, expect)
    for name in ("kIndex", "ductingIndex", "showalterIndex", "boydenIndex", "sweatIndex", "seaSoundSpeed", "cvtemp", "cvhum", "abshum",
                 "windCooling", "underCooledRain", "vesselIcingOverland", "vesselIcingMertins", "vesselIcingModStall", "vesselIcingMincog"):
        assert callable(getattr(pyfc, name))  # the 15 functions of py_mi_fieldcalc.cc:189-207
    with pytest.raises(NotImplementedError):
        pyfc.vesselIcingMincog(*([t] * 11), 5.0, 0.5, 1.0, 4.0, 1, 1e35)


@pytest.mark.parametrize("nx,ny", [(129, 40), (128, 40), (484, 71), (1440, 9)])
@pytest.mark.parametrize("flag", [ALL, SOME])
def test_shapiro_filter_in_place(gpu_ctx, oracle, flag, nx, ny):
    """shapiro2_filter allows field == fsmooth (FieldCalculations.cc:2088, :2099): device tensor smoothed in place
    (widths the one-launch kernel takes, and one it does not)."""
    import torch

    import mi_fieldcalc_amd.synth as synth

    z = synth.scalar_field(nx, ny, 77)
    if flag == SOME:
        z = synth.sprinkle_undef(z, 5, 0.05)
    expect = z.copy()
    ok, out_e, flag_e = oracle.call("shapiro2_filter", nx, ny, expect, fdefined=flag, outs=[expect])  # the oracle in place, too
    assert ok and flag_e == ALL
    dz = torch.from_numpy(z.copy()).cuda()
    res, flag_g = gpu_ctx.shapiro2_filter(dz, fdefined=flag, out=dz)
    assert flag_g == ALL and res.data_ptr() == dz.data_ptr()
    assert cases.same_bits(dz.cpu().numpy(), expect, nan_payload=False)


def test_cxx_api_runs_on_the_gpu(gpu_ctx, tmp_path):
    """The source-compatible C++ API (unchanged caller code, host pointers, std::vector
    signatures) with a device present: operators return true and the values are right."""
    import test_capi_and_host as host_tests

    host_tests.test_cxx_header_is_source_compatible(host_tests.LIB, tmp_path)


def test_concurrent_callers_with_their_own_contexts(oracle):
    """The reference is re-entrant and its Python binding releases the GIL
    (python/py_mi_fieldcalc.cc:75): several threads call at once.  One context per
    thread; every thread checks its own results against the oracle."""
    import threading

    import mi_fieldcalc_amd as fc
    import mi_fieldcalc_amd.synth as synth

    nx, ny = 516, 70
    xm, ym, fcor = synth.grid_maps(nx, ny)
    errors = []

    def worker(k):
        try:
            ctx = fc.Context(0)
            for it in range(6):
                u, v = synth.wind(nx, ny, 100 * k + it)
                u = synth.sprinkle_undef(u, k + it, 0.02)
                for op, args in (("relvort", [u, v, xm, ym]), ("absvort", [u, v, xm, ym, fcor]), ("vectorabs", [u, v])):
                    ok, expect, flag_e = oracle.call(op, nx, ny, *args, fdefined=SOME)
                    out, flag = getattr(ctx, op)(*args, fdefined=SOME)
                    if not (ok and flag == flag_e and cases.same_bits(out, expect, nan_payload=False)):
                        errors.append((k, it, op))
            ctx.close()
        except Exception as e:  # noqa: BLE001 - reported below
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_ensemble_reductions(gpu_ctx, oracle):
    """sum / mean / stddev / extreme / probability over members, bit-exact
    (the members are reduced in index order, like the reference's inner loop)."""
    cs = cases.ensemble_cases()
    assert len(cs) > 400
    for case in cs:
        _check_case(gpu_ctx, oracle, case, device=False)
    for case in cases.ensemble_cases(grids=((17, 9), (64, 48)), modes=("all", "some")):
        _check_case(gpu_ctx, oracle, case, device=True)


def test_ensemble_51_members_full_field(gpu_ctx, oracle):
    """BASELINE.json config 5 has 51 members: one 1440x720 level of them, device resident."""
    import torch

    import mi_fieldcalc_amd.synth as synth

    nx, ny, nmem = 1440, 720, 51
    members = [synth.uniform((ny, nx), 4000 + k, -5.0, 30.0).astype(np.float32) for k in range(nmem)]
    members[7] = synth.sprinkle_undef(members[7], 1, 0.01)
    flags = [ALL] * nmem
    flags[7] = SOME
    dm = [torch.from_numpy(m).cuda() for m in members]
    for op, args_cpu, call in (
        ("meanValue", [members, flags], lambda: gpu_ctx.meanValue(dm, flags)),
        ("stddevValue", [members, flags], lambda: gpu_ctx.stddevValue(dm, flags)),
        ("probability", [1, members, flags, [20.0]], lambda: gpu_ctx.probability(1, dm, flags, [20.0])),
        ("extremeValue", [3, members], lambda: gpu_ctx.extremeValue(3, dm, fdefined=SOME)),
    ):
        ok, expect, flag_e = oracle.call(op, nx, ny, *args_cpu, fdefined=SOME)
        out, flag = call()
        assert ok and flag == flag_e and cases.same_bits(out.cpu().numpy(), expect, nan_payload=False), op


@pytest.mark.parametrize("device", [False, True])
def test_ensemble_more_members_than_ride_in_the_kernel_arguments(gpu_ctx, oracle, device):
    """Up to 64 member pointers travel in the kernel arguments; a larger ensemble uses the device table."""
    import torch

    import mi_fieldcalc_amd.synth as synth

    nx, ny, nmem = 68, 21, 70
    members = [synth.uniform((ny, nx), 9000 + k, -5.0, 30.0).astype(np.float32) for k in range(nmem)]
    members[66] = synth.sprinkle_undef(members[66], 3, 0.05)
    members[2] = np.full_like(members[2], cases.UNDEF)
    flags = [ALL] * nmem
    flags[66] = SOME
    flags[2] = NONE
    m = [torch.from_numpy(x).cuda() for x in members] if device else members
    for op, args_cpu, call in (
        ("meanValue", [members, flags], lambda: gpu_ctx.meanValue(m, flags)),
        ("stddevValue", [members, flags], lambda: gpu_ctx.stddevValue(m, flags)),
        ("probability", [4, members, flags, [20.0]], lambda: gpu_ctx.probability(4, m, flags, [20.0])),
        ("extremeValue", [4, members], lambda: gpu_ctx.extremeValue(4, m, fdefined=SOME)),
        ("sumFields", [members], lambda: gpu_ctx.sumFields(m, fdefined=SOME)),
    ):
        ok, expect, flag_e = oracle.call(op, nx, ny, *args_cpu, fdefined=SOME)
        out, flag = call()
        out = out.cpu().numpy() if device else out
        assert ok and flag == flag_e and cases.same_bits(out, expect, nan_payload=False), op


def test_against_golden_vectors(gpu_ctx):
    g, cs = golden_util.ensemble_golden_cases()
    for case in cs:
        _check_case(gpu_ctx, None, case, expected=g.expect(case))
    g, cs = golden_util.stencil_golden_cases()
    for case in cs:
        _check_case(gpu_ctx, None, case, expected=g.expect(case))
    g, cs = golden_util.ewise_golden_cases()
    for case in cs:
        _check_case(gpu_ctx, None, case, expected=g.expect(case))
    g, cs = golden_util.catalogue_golden_cases()
    for case in cs:
        _check_case(gpu_ctx, None, case, expected=g.expect(case))


def test_reference_known_answers(gpu_ctx):
    """test/FieldCalculationsTest.cc:70-143 through the HIP path."""
    from test_oracle_golden import KA_UNDEF, xlevelhum_known_answers

    def call(op, args, fdef):
        res = getattr(gpu_ctx, op)(*args, fdefined=fdef, undef=KA_UNDEF)
        if res is None:
            return False, float("nan"), -1
        out, flag = res
        return True, float(out[0, 0]), flag

    xlevelhum_known_answers(call)


def test_aleveltemp_theta_within_4ulp(gpu_ctx):
    """test/FieldCalculationsTest.cc:145-170 through the HIP path (device powf)."""
    from test_oracle_golden import aleveltemp_performance_inputs, ulp_diff

    n, tk, p = aleveltemp_performance_inputs()
    res = gpu_ctx.aleveltemp(tk.reshape(n, 1), p.reshape(n, 1), "kelvin", 3, fdefined=ALL, undef=np.float32(1e30))
    assert res is not None
    th, flag = res
    p0inv = np.float32(1.0 / 1000.0)
    kappa = np.float32(287.0) / np.float32(1004.0)
    ex = tk / np.power(p * p0inv, kappa, dtype=np.float32)
    assert ulp_diff(th.reshape(-1), ex).max() <= 4


# ------------------------------------------------------------------ batched levels
def _levels_inputs(nx, ny, nlev, seed, mixed=True):
    import mi_fieldcalc_amd.synth as synth

    xm, ym, fc = synth.grid_maps(nx, ny)
    u, v = synth.wind(nx, ny, seed, nlev=nlev)
    flags = np.full(nlev, SOME, np.int32)
    if mixed:
        for l in range(nlev):
            kind = l % 4
            if kind == 0:
                flags[l] = ALL
            elif kind == 1:
                u[l] = synth.sprinkle_undef(u[l], seed + l, 0.02)
                v[l] = synth.sprinkle_undef(v[l], seed + 100 + l, 0.02)
            elif kind == 2:
                u[l] = cases.UNDEF
                v[l] = cases.UNDEF
            else:  # flag lies: ALL_DEFINED but undefined values present
                flags[l] = ALL
                u[l] = synth.sprinkle_undef(u[l], seed + l, 0.02)
    else:
        flags[:] = ALL
    return u, v, xm, ym, flags


def _expect_levels(oracle, u, v, xm, ym, flags):
    nlev, ny, nx = u.shape
    rv = np.empty_like(u)
    dv = np.empty_like(u)
    fo = np.empty(nlev, np.int32)
    for l in range(nlev):
        ok, o, f1 = oracle.call("relvort", nx, ny, u[l], v[l], xm, ym, fdefined=int(flags[l]))
        assert ok
        rv[l] = o
        ok, o, f2 = oracle.call("divergence", nx, ny, u[l], v[l], xm, ym, fdefined=int(flags[l]))
        dv[l] = o
        assert f1 == f2
        fo[l] = f1
    return rv, dv, fo


@pytest.mark.parametrize("nx,ny,nlev", [(64, 48, 9), (516, 70, 6), (260, 11, 5), (17, 9, 7), (1440, 75, 5)])
@pytest.mark.parametrize("tune", ["", "R=7,D=0", "R=5,D=1,NT=1", "R=64,D=1,WPB=8", "K=1", "K=1,XCD=0,NT=0", "K=2", "K=2,XCD=0", "R=1", "R=2,WPB=2", "R=8",
                                  "K=2,LG=8", "K=2,LG=4,XCD=0", "K=1,LG=3", "K=2,RB=14", "K=2,RB=14,LG=16",
                                  "K=3", "K=3,RB=8,LG=2", "K=3,RB=16,LG=4,XCD=0", "K=3,RB=12,LG=1", "K=3,RB=16,LG=3,D=0", "K=3,RB=8,D=0,LG=4",
                                  "K=3,RB=16,LG=2,D=0,ZZ=1", "K=3,RB=12,D=1,ZZ=1", "K=3,RB=8,LG=5,D=0,ZZ=1,XCD=0",
                                  "K=4", "K=4,D=0,LG=1", "K=4,D=1,LG=3", "K=4,D=2,LG=2,XCD=0", "K=4,RB=6,D=2,LG=4", "K=4,RB=8,D=1,LG=5", "K=4,RB=12,D=1,LG=2",
                                  "K=4,RB=6,D=0"])
def test_vortdiv_levels_matches_per_level_reference_calls(gpu_ctx, oracle, nx, ny, nlev, tune, mifc_env):
    import torch

    if tune:
        mifc_env("MIFC_VORTDIV_TUNE", tune)
    u, v, xm, ym, flags = _levels_inputs(nx, ny, nlev, 4242 + nx)
    rv_e, dv_e, fo_e = _expect_levels(oracle, u, v, xm, ym, flags)
    # host pointers
    (rv, dv), fo = gpu_ctx.vortdiv_levels(u, v, xm, ym, fdefined=flags)
    assert cases.same_bits(rv, rv_e, nan_payload=False) and cases.same_bits(dv, dv_e, nan_payload=False)
    assert np.array_equal(fo, fo_e)
    # device resident, single outputs
    du, dvv, dxm, dym = (torch.from_numpy(a).cuda() for a in (u, v, xm, ym))
    (rv2, none), fo2 = gpu_ctx.vortdiv_levels(du, dvv, dxm, dym, fdefined=flags, want=("rvort",))
    assert none is None and cases.same_bits(rv2.cpu().numpy(), rv_e, nan_payload=False) and np.array_equal(fo2, fo_e)
    (none, dv2), fo3 = gpu_ctx.vortdiv_levels(du, dvv, dxm, dym, fdefined=flags, want=("diverg",))
    assert none is None and cases.same_bits(dv2.cpu().numpy(), dv_e, nan_payload=False) and np.array_equal(fo3, fo_e)


@pytest.mark.parametrize("nx,ny,nlev", [(949, 23, 4), (1001, 13, 3), (258, 9, 2), (6, 300, 2), (4, 3, 1), (5, 4, 3), (1443, 7, 2)])
@pytest.mark.parametrize("force_cell", ["0", "1"])
def test_wind_operators_on_ragged_widths(gpu_ctx, oracle, nx, ny, nlev, force_cell, mifc_env):
    """Widths that are not a multiple of 4 take the flat four-cells-per-lane kernel (dword-aligned 16-byte accesses, rows
    above / below as the same loads nx cells away, first / last columns and rows through the per-cell path); with
    MIFC_FORCE_CELL_KERNEL=1 the one-lane-per-cell kernel.  Fused pair, single outputs and absvort, mixed flags per level."""
    import torch

    import mi_fieldcalc_amd.synth as synth

    mifc_env("MIFC_FORCE_CELL_KERNEL", force_cell)
    u, v, xm, ym, flags = _levels_inputs(nx, ny, nlev, 777 + nx)
    rv_e, dv_e, fo_e = _expect_levels(oracle, u, v, xm, ym, flags)
    du, dvv, dxm, dym = (torch.from_numpy(a).cuda() for a in (u, v, xm, ym))
    (rv, dv), fo = gpu_ctx.vortdiv_levels(du, dvv, dxm, dym, fdefined=flags)
    assert cases.same_bits(rv.cpu().numpy(), rv_e, nan_payload=False) and cases.same_bits(dv.cpu().numpy(), dv_e, nan_payload=False)
    assert np.array_equal(fo, fo_e)
    (rv2, none), fo2 = gpu_ctx.vortdiv_levels(du, dvv, dxm, dym, fdefined=flags, want=("rvort",))
    assert none is None and cases.same_bits(rv2.cpu().numpy(), rv_e, nan_payload=False) and np.array_equal(fo2, fo_e)
    (none, dv2), fo3 = gpu_ctx.vortdiv_levels(du, dvv, dxm, dym, fdefined=flags, want=("diverg",))
    assert none is None and cases.same_bits(dv2.cpu().numpy(), dv_e, nan_payload=False) and np.array_equal(fo3, fo_e)
    _, _, fcor = synth.grid_maps(nx, ny)
    for l in range(nlev):
        ok, av_e, f = oracle.call("absvort", nx, ny, u[l], v[l], xm, ym, fcor, fdefined=int(flags[l]))
        res = gpu_ctx.absvort(du[l], dvv[l], dxm, dym, torch.from_numpy(fcor).cuda(), fdefined=int(flags[l]))
        assert ok and res is not None and res[1] == f and cases.same_bits(res[0].cpu().numpy(), av_e, nan_payload=False)


def test_vortdiv_levels_all_defined_fast_path(gpu_ctx, oracle):
    u, v, xm, ym, flags = _levels_inputs(256, 40, 8, 777, mixed=False)
    rv_e, dv_e, fo_e = _expect_levels(oracle, u, v, xm, ym, flags)
    (rv, dv), fo = gpu_ctx.vortdiv_levels(u, v, xm, ym, fdefined=flags)
    assert cases.same_bits(rv, rv_e, nan_payload=False) and cases.same_bits(dv, dv_e, nan_payload=False) and np.array_equal(fo, fo_e)


@pytest.mark.parametrize("nlev,want", [(21, ("rvort", "diverg")), (18, ("diverg",))])
def test_vortdiv_levels_host_pipeline(gpu_ctx, oracle, nlev, want, mifc_env):
    """A host-resident batch above 64 MiB per field is streamed through the
    device in chunks (ragged last chunk here); results and flags must equal the
    whole-batch path and, on sampled levels, the reference's per-level calls."""
    nx, ny = 1440, 720
    u, v, xm, ym, flags = _levels_inputs(nx, ny, nlev, 31337)
    kw = dict(fdefined=flags, want=want)
    (rv, dv), fo = gpu_ctx.vortdiv_levels(u, v, xm, ym, **kw)
    mifc_env("MIFC_HOST_PIPELINE", "0")
    (rv0, dv0), fo0 = gpu_ctx.vortdiv_levels(u, v, xm, ym, **kw)
    assert np.array_equal(fo, fo0)
    for a, b in ((rv, rv0), (dv, dv0)):
        assert (a is None) == (b is None)
        if a is not None:
            assert cases.same_bits(a, b, nan_payload=False)
    for l in (0, 1, 5, nlev - 2, nlev - 1):
        ok, o, f = oracle.call("divergence", nx, ny, u[l], v[l], xm, ym, fdefined=int(flags[l]))
        assert ok and f == fo[l] and cases.same_bits(dv[l], o, nan_payload=False)


def test_held_constant_fields(gpu_ctx, oracle):
    """Map ratios declared constant are uploaded once; results do not change."""
    import mi_fieldcalc_amd.synth as synth

    nx, ny = 260, 37
    xm, ym, fcor = synth.grid_maps(nx, ny)
    u, v = synth.wind(nx, ny, 5)
    ok, expect, flag_e = oracle.call("absvort", nx, ny, u, v, xm, ym, fcor, fdefined=SOME)
    for a in (xm, ym, fcor):
        gpu_ctx.hold_field(a)
    try:
        out, flag = gpu_ctx.absvort(u, v, xm, ym, fcor, fdefined=SOME)
        assert flag == flag_e and cases.same_bits(out, expect, nan_payload=False)
        # a changed array must be held again to take effect
        xm2 = xm.copy()
        xm[...] = xm * np.float32(2)
        out_stale, _ = gpu_ctx.absvort(u, v, xm, ym, fcor, fdefined=SOME)
        assert cases.same_bits(out_stale, expect, nan_payload=False)
        gpu_ctx.hold_field(xm)
        ok, expect2, _ = oracle.call("absvort", nx, ny, u, v, xm, ym, fcor, fdefined=SOME)
        out2, _ = gpu_ctx.absvort(u, v, xm, ym, fcor, fdefined=SOME)
        assert cases.same_bits(out2, expect2, nan_payload=False)
        xm[...] = xm2
    finally:
        for a in (xm, ym, fcor):
            gpu_ctx.release_field(a)
    out3, _ = gpu_ctx.absvort(u, v, xm, ym, fcor, fdefined=SOME)
    assert cases.same_bits(out3, expect, nan_payload=False)


def test_vortdiv_enqueue_counts(gpu_ctx, oracle):
    import torch

    nx, ny, nlev = 128, 33, 6
    u, v, xm, ym, flags = _levels_inputs(nx, ny, nlev, 99)
    rv_e, dv_e, fo_e = _expect_levels(oracle, u, v, xm, ym, flags)
    du, dvv, dxm, dym = (torch.from_numpy(a).cuda() for a in (u, v, xm, ym))
    rv = torch.empty_like(du)
    dv = torch.empty_like(du)
    cnt = torch.full((nlev,), -1, dtype=torch.int64, device="cuda")
    gpu_ctx.use_torch_stream()
    try:
        assert gpu_ctx.vortdiv_levels_enqueue(du, dvv, dxm, dym, rv, dv, fdefined=flags, n_undefined=cnt)
        torch.cuda.synchronize()
    finally:
        gpu_ctx.set_stream(None)
    import mi_fieldcalc_amd as fc

    fo = np.array([fc.classify(int(c), nx * ny - 2 * nx) for c in cnt.cpu().numpy()])
    fo = np.where(flags == ALL, ALL, fo)  # ALL_DEFINED levels run without tests: count stays 0
    assert np.array_equal(fo, fo_e)
    assert cases.same_bits(rv.cpu().numpy(), rv_e, nan_payload=False) and cases.same_bits(dv.cpu().numpy(), dv_e, nan_payload=False)


# ------------------------------------------------------------------ row slabs
@pytest.mark.parametrize("nx,ny,nslab", [(64, 40, 4), (260, 23, 3), (512, 64, 8), (33, 17, 2), (949, 41, 3), (258, 20, 4)])
@pytest.mark.parametrize("mode", ["all", "some"])
@pytest.mark.parametrize("tune", [None, "K=2", "R=8", "K=4,D=1", "K=4,RB=6,D=2"])
def test_vortdiv_row_slabs_equal_whole_field(gpu_ctx, oracle, nx, ny, nslab, mode, tune, mifc_env):
    """Config 4 decomposition exercised on one GPU with a loop-back halo 'exchange' (the launcher picks the
    kernel form by launch size: the default, the one-shot tiles and the row-walking form all have to agree)."""
    import torch

    if tune is not None:
        mifc_env("MIFC_VORTDIV_TUNE", tune)

    import mi_fieldcalc_amd as fc
    import mi_fieldcalc_amd.synth as synth
    from mi_fieldcalc_amd.sharding import slab_rows

    xm, ym, _ = synth.grid_maps(nx, ny)
    u, v = synth.wind(nx, ny, 31337 + nx)
    (u, v), flag = cases._apply_mode([u, v], mode, 5, 0.03)
    ok, rv_e, f1 = oracle.call("relvort", nx, ny, u, v, xm, ym, fdefined=flag)
    ok, dv_e, f2 = oracle.call("divergence", nx, ny, u, v, xm, ym, fdefined=flag)
    rv = np.empty_like(u)
    dv = np.empty_like(u)
    total = 0
    gpu_ctx.use_torch_stream()
    try:
        for r in range(nslab):
            j0, nloc = slab_rows(ny, nslab, r)
            uh = np.zeros((nloc + 2, nx), np.float32)
            vh = np.zeros((nloc + 2, nx), np.float32)
            uh[1:-1] = u[j0:j0 + nloc]
            vh[1:-1] = v[j0:j0 + nloc]
            if j0 > 0:
                uh[0], vh[0] = u[j0 - 1], v[j0 - 1]
            if j0 + nloc < ny:
                uh[-1], vh[-1] = u[j0 + nloc], v[j0 + nloc]
            t = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (uh, vh, xm[j0:j0 + nloc], ym[j0:j0 + nloc])]
            o_rv = torch.empty((nloc, nx), dtype=torch.float32, device="cuda")
            o_dv = torch.empty_like(o_rv)
            cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
            assert gpu_ctx.vortdiv_slab_enqueue(nx, ny, j0, nloc, t[0], t[1], t[2], t[3], o_rv, o_dv, fdefined_in=flag, n_undefined=cnt)
            torch.cuda.synchronize()
            rv[j0:j0 + nloc] = o_rv.cpu().numpy()
            dv[j0:j0 + nloc] = o_dv.cpu().numpy()
            total += int(cnt.item())
    finally:
        gpu_ctx.set_stream(None)
    assert cases.same_bits(rv, rv_e, nan_payload=False) and cases.same_bits(dv, dv_e, nan_payload=False)
    got_flag = ALL if flag == ALL else fc.classify(total, nx * ny - 2 * nx)
    assert got_flag == f1 == f2


# ------------------------------------------------------------------ fused derived variables
@pytest.mark.parametrize("mode", ["all", "some"])
def test_hlevel_derived_levels(gpu_ctx, oracle, mode):
    import mi_fieldcalc_amd.synth as synth

    nx, ny, nlev = 64, 36, 5
    u, v = synth.wind(nx, ny, 11, nlev=nlev)
    t, q, ps = synth.thermo(nx, ny, 12, nlev=nlev)
    a, b = synth.hybrid_levels(nlev)
    fw = np.full(nlev, ALL if mode == "all" else SOME, np.int32)
    ft = fw.copy()
    if mode == "some":
        for l in range(nlev):
            u[l] = synth.sprinkle_undef(u[l], 50 + l, 0.02)
            t[l] = synth.sprinkle_undef(t[l], 60 + l, 0.02)
            q[l] = synth.sprinkle_undef(q[l], 70 + l, 0.02)
        ps = synth.sprinkle_undef(ps, 80, 0.02, nan_every=0)
        t[0, 0, :8] = 400.0  # outside the ewt table: rh undefined, theta still defined
    res, flags = gpu_ctx.hlevel_derived_levels(u, v, t, q, ps, a, b, fdef_wind=fw, fdef_thermo=ft)
    for l in range(nlev):
        ok, ff_e, f_ff = oracle.call("vectorabs", nx, ny, u[l], v[l], fdefined=int(fw[l]))
        ok, rh_e, f_rh = oracle.call("hlevelhum", nx, ny, t[l], q[l], ps, float(a[l]), float(b[l]), "", 1, fdefined=int(ft[l]))
        ok, th_e, f_th = oracle.call("hleveltemp", nx, ny, t[l], ps, float(a[l]), float(b[l]), "", 3, fdefined=int(ft[l]))
        case = dict(label="derived-l%d" % l, undef=cases.UNDEF)
        gpu_util.compare(case, res["ff"][l], ff_e, True)
        gpu_util.compare(case, res["rh"][l], rh_e, True)   # T,q -> RH: no powf involved
        gpu_util.compare(case, res["theta"][l], th_e, False)  # theta: device powf, 1e-5 relative
        assert (flags["ff"][l], flags["rh"][l], flags["theta"][l]) == (f_ff, f_rh, f_th)
    # bad hybrid level -> false
    assert gpu_ctx.hlevel_derived_levels(u, v, t, q, ps, -a, b, fdef_wind=fw, fdef_thermo=ft) is None


@pytest.mark.parametrize("want", [("ff", "rh", "theta"), ("theta",)])
def test_hlevel_derived_levels_host_pipeline(gpu_ctx, want, mifc_env):
    """The fused ff / RH / theta batch from host memory streams through the device in
    chunks (ragged last chunk); values and flags must equal the whole-batch path bit for bit."""
    import mi_fieldcalc_amd.synth as synth

    nx, ny, nlev = 1440, 720, 18
    u, v = synth.wind(nx, ny, 21, nlev=nlev)
    t, q, ps = synth.thermo(nx, ny, 22, nlev=nlev)
    a, b = synth.hybrid_levels(nlev)
    fw = np.full(nlev, SOME, np.int32)
    ft = fw.copy()
    fw[::3] = ALL
    for l in (1, 7, nlev - 1):
        u[l] = synth.sprinkle_undef(u[l], 50 + l, 0.02)
        t[l] = synth.sprinkle_undef(t[l], 60 + l, 0.02)
    kw = dict(fdef_wind=fw, fdef_thermo=ft, want=want)
    res, flags = gpu_ctx.hlevel_derived_levels(u, v, t, q, ps, a, b, **kw)
    mifc_env("MIFC_HOST_PIPELINE", "0")
    res0, flags0 = gpu_ctx.hlevel_derived_levels(u, v, t, q, ps, a, b, **kw)
    for k in want:
        assert cases.same_bits(res[k], res0[k], nan_payload=False), k
        assert np.array_equal(flags[k], flags0[k]), k


# ------------------------------------------------------------------ headline size
def test_headline_1440x720x137_properties(gpu_ctx, oracle, mifc_env):
    """Full BASELINE.json configuration on the device: sampled levels against the
    oracle bit for bit, every level through a size-independent property (the
    fused row-sliding kernel and the one-lane-per-cell kernel are independent
    implementations and must agree on every bit of every level)."""
    import torch

    import mi_fieldcalc_amd.synth as synth

    nx, ny, nlev = 1440, 720, 137
    xm, ym, _ = synth.grid_maps(nx, ny)
    dxm, dym = torch.from_numpy(xm).cuda(), torch.from_numpy(ym).cuda()
    du, dv = synth.device_wind(nx, ny, nlev, 0x5EED0000 + 3000, "cuda")
    flags = np.full(nlev, ALL, np.int32)
    (rv, dg), fo = gpu_ctx.vortdiv_levels(du, dv, dxm, dym, fdefined=flags)
    assert np.all(fo == ALL)
    for l in (0, 1, 68, 136):
        ul, vl = du[l].cpu().numpy(), dv[l].cpu().numpy()
        ok, e, _ = oracle.call("relvort", nx, ny, ul, vl, xm, ym, fdefined=ALL)
        assert cases.same_bits(rv[l].cpu().numpy(), e, nan_payload=False)
        ok, e, _ = oracle.call("divergence", nx, ny, ul, vl, xm, ym, fdefined=ALL)
        assert cases.same_bits(dg[l].cpu().numpy(), e, nan_payload=False)
    mifc_env("MIFC_FORCE_CELL_KERNEL", "1")
    (rv2, dg2), _ = gpu_ctx.vortdiv_levels(du, dv, dxm, dym, fdefined=flags)
    mifc_env("MIFC_FORCE_CELL_KERNEL", None)
    assert torch.equal(rv.view(torch.int32), rv2.view(torch.int32))
    assert torch.equal(dg.view(torch.int32), dg2.view(torch.int32))
    # edge rule of fillEdges on every level: rows 0 / ny-1 and columns 0 / nx-1 are copies
    assert torch.equal(rv[:, 0, :], rv[:, 1, :]) and torch.equal(rv[:, -1, :], rv[:, -2, :])
    assert torch.equal(dg[:, :, 0], dg[:, :, 1]) and torch.equal(dg[:, :, -1], dg[:, :, -2])
    # with undefined cells sprinkled in: counts per level agree between the two kernels
    du[:, 100:110, 200:260] = float(cases.UNDEF)
    flags[:] = SOME
    (rv3, dg3), fo3 = gpu_ctx.vortdiv_levels(du, dv, dxm, dym, fdefined=flags)
    mifc_env("MIFC_FORCE_CELL_KERNEL", "1")
    (rv4, dg4), fo4 = gpu_ctx.vortdiv_levels(du, dv, dxm, dym, fdefined=flags)
    mifc_env("MIFC_FORCE_CELL_KERNEL", None)
    assert np.array_equal(fo3, fo4) and np.all(fo3 == SOME)
    assert torch.equal(rv3.view(torch.int32), rv4.view(torch.int32)) and torch.equal(dg3.view(torch.int32), dg4.view(torch.int32))
    l = 5
    ok, e, f = oracle.call("relvort", nx, ny, du[l].cpu().numpy(), dv[l].cpu().numpy(), xm, ym, fdefined=SOME)
    assert cases.same_bits(rv3[l].cpu().numpy(), e, nan_payload=False) and f == fo3[l]


# ------------------------------------------------------------------ generic batched stencils
@pytest.mark.parametrize("nx,ny,nlev", [(64, 48, 5), (260, 21, 4), (520, 25, 7), (1440, 13, 9), (8, 3, 3)])
@pytest.mark.parametrize("walk", [None, "1", "1/nosplit", "1/TR=12,NL=4,PF=2", "1/TR=14,NL=2,PF=2,LG=2", "1/TR=8,NL=2,PF=2,LG=3", "1/TR=12,NL=2,PF=3,LG=4",
                                  "1/TR=12,NL=2,PF=1,LG=1", "1/TR=10,NL=2,PF=2", "1/TR=13,NL=3,PF=2,LG=5", "1/TR=12,NL=2,PF=2"])
def test_stencil_levels_every_operator(gpu_ctx, oracle, nx, ny, nlev, walk, mifc_env):
    """mifc_stencil_levels: each operator over a batch == the per-level reference call, flags included.
    walk="1": the level-walking forms (tiles that stay put and walk the levels) are chosen whatever the launch
    size -- ragged tiles, widths below one segment, chunks of unequal length; device-resident batches then.  By default
    those are the split-role kernels (loader waves / compute waves) for every operator but the Jacobian; "1/nosplit" keeps
    the forms whose waves load and store, "1/TR=..." other shapes of the one-input split-role kernel."""
    import torch

    import mi_fieldcalc_amd.synth as synth

    nosplit = False
    if walk is not None and "/" in walk:
        walk, how = walk.split("/")
        if how == "nosplit":
            nosplit = True
            mifc_env("MIFC_VORTDIV_SPLIT", "0")
        else:
            mifc_env("MIFC_SCALAR_SPLIT_TUNE", how)
    mifc_env("MIFC_LEVELWALK_MIN_UNITS", walk)
    on_device = walk is not None
    xm, ym, fcor = synth.grid_maps(nx, ny)
    u, v = synth.wind(nx, ny, 606 + nx, nlev=nlev)
    z = np.stack([synth.scalar_field(nx, ny, 700 + l) for l in range(nlev)])
    flags = np.full(nlev, SOME, np.int32)
    flags[0] = ALL
    for l in range(1, nlev):
        if l % 2:
            u[l] = synth.sprinkle_undef(u[l], 10 + l, 0.03)
            z[l] = synth.sprinkle_undef(z[l], 20 + l, 0.03)
    table = [
        ("relvort", "relvort", u, v, False, None, []), ("divergence", "divergence", u, v, False, None, []),
        ("absvort", "absvort", u, v, True, None, []), ("vortdiv", None, u, v, False, None, []),
        ("gradient1", "gradient", z, None, False, 1, []), ("gradient2", "gradient", z, None, False, 2, []),
        ("gradient3", "gradient", z, None, False, 3, []), ("gradient4", "gradient", z, None, False, 4, []),
        ("plevelgwind_xcomp", "plevelgwind_xcomp", z, None, True, None, []), ("plevelgwind_ycomp", "plevelgwind_ycomp", z, None, True, None, []),
        ("plevelgvort", "plevelgvort", z, None, True, None, []), ("ilevelgwind", "ilevelgwind", z, None, True, None, []),
        ("jacobian", "jacobian", z, u, False, None, []),
    ]
    dev = (lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()) if on_device else (lambda a: a)
    host = (lambda a: None if a is None else a.cpu().numpy()) if on_device else (lambda a: a)
    dxm, dym, dfc = dev(xm), dev(ym), dev(fcor)
    for name, cpu_op, f0, f1, use_fc, compute, _ in table:
        res = gpu_ctx.stencil_levels(name, dev(f0), dev(f1), dxm, dym, dfc if use_fc else None, fdefined=flags)
        assert res is not None, name
        if on_device:  # the case reaches the kernel it is meant for
            if name == "jacobian":
                want_form = "wind_rows" if nosplit else "wind_split"
            elif f1 is not None:
                split = not nosplit and name in ("vortdiv", "absvort")
                want_form = "wind_split" if split else ("wind_rows" if name == "absvort" else "wind_levelwalk")
            else:
                want_form = "scalar_levelwalk" if nosplit else "scalar_split"
            gpu_util.check_form(gpu_ctx, want_form, what=name)
        (o0, o1), fo = res
        o0, o1 = host(o0), host(o1)
        for l in range(nlev):
            if name == "vortdiv":
                ok, e0, f_e = oracle.call("relvort", nx, ny, u[l], v[l], xm, ym, fdefined=int(flags[l]))
                ok, e1, _ = oracle.call("divergence", nx, ny, u[l], v[l], xm, ym, fdefined=int(flags[l]))
            else:
                args = [f0[l]] + ([f1[l]] if f1 is not None else []) + [xm, ym] + ([fcor] if use_fc else []) + ([compute] if compute else [])
                ok, e, f_e = oracle.call(cpu_op, nx, ny, *args, fdefined=int(flags[l]))
                e0, e1 = (e if isinstance(e, tuple) else (e, None))
            assert cases.same_bits(o0[l], e0, nan_payload=False), (name, l)
            if e1 is not None:
                assert cases.same_bits(o1[l], e1, nan_payload=False), (name, l)
            assert fo[l] == f_e, (name, l, fo[l], f_e)


@pytest.mark.parametrize("nx,ny,nlev,shift", [(949, 23, 4, 0), (1001, 27, 3, 0), (258, 15, 5, 0), (1443, 16, 3, 0), (515, 14, 7, 0), (1442, 26, 3, 0), (257, 9, 3, 0),
                                               (516, 25, 4, 1), (1440, 15, 3, 3), (260, 38, 3, 2)])
@pytest.mark.parametrize("ragged_split", ["1", "0"])
def test_split_role_kernels_on_rows_at_any_alignment(gpu_ctx, oracle, nx, ny, nlev, shift, ragged_split, mifc_env):
    """The RAGGED variants of the split-role level-walking kernels (mifc_vortdiv.hip, mifc_stencil_split.hip): widths that are
    not a multiple of 4 (the last column group of a row is partial: remainders 1, 2 and 3, the fill column in the lane below
    and in the own lane), and widths that are but with every field `shift` floats off the 16-byte grid (views into larger
    arrays).  Each operator over a batch == the per-level reference call, flags included; the batch ends where its
    allocation ends, so the last group of the last level is the one loaded cell by cell.  257 (one column in the last segment)
    is declined and takes the flat kernels, as everything does under MIFC_RAGGED_SPLIT=0."""
    import torch

    import mi_fieldcalc_amd.synth as synth

    mifc_env("MIFC_LEVELWALK_MIN_UNITS", "1")
    mifc_env("MIFC_RAGGED_SPLIT", ragged_split)
    xm, ym, fcor = synth.grid_maps(nx, ny)
    u, v = synth.wind(nx, ny, 4242 + nx, nlev=nlev)
    z = np.stack([synth.scalar_field(nx, ny, 4300 + l) for l in range(nlev)])
    flags = np.full(nlev, SOME, np.int32)
    flags[0] = ALL
    for l in range(1, nlev):
        if l % 2:
            u[l] = synth.sprinkle_undef(u[l], 10 + l, 0.03)
            z[l] = synth.sprinkle_undef(z[l], 20 + l, 0.03)
    # undefined values where the partial group and the fill columns are: last columns of rows 1 and ny-2, first column
    u[nlev - 1, 1, nx - 2] = cases.UNDEF
    z[nlev - 1, ny - 2, nx - 1] = cases.UNDEF
    z[nlev - 1, ny - 1, nx - 1] = cases.UNDEF
    z[nlev - 1, 2, 0] = cases.UNDEF

    def dev(a):
        if a is None:
            return None
        a = np.ascontiguousarray(a)
        big = torch.empty(a.size + shift, dtype=torch.float32, device="cuda")
        t = big[shift:].view(a.shape)
        t.copy_(torch.from_numpy(a))
        assert t.data_ptr() % 16 == 4 * shift
        return t

    dxm, dym, dfc = dev(xm), dev(ym), dev(fcor)
    def table_of(u, v, z):
        return [("vortdiv", None, u, v, False, None), ("relvort", "relvort", u, v, False, None), ("divergence", "divergence", u, v, False, None),
                ("absvort", "absvort", u, v, True, None), ("gradient1", "gradient", z, None, False, 1), ("gradient2", "gradient", z, None, False, 2),
                ("gradient3", "gradient", z, None, False, 3), ("gradient4", "gradient", z, None, False, 4),
                ("plevelgwind_xcomp", "plevelgwind_xcomp", z, None, True, None), ("plevelgwind_ycomp", "plevelgwind_ycomp", z, None, True, None),
                ("plevelgvort", "plevelgvort", z, None, True, None), ("ilevelgwind", "ilevelgwind", z, None, True, None),
                ("jacobian", "jacobian", z, u, False, None)]

    for all_defined in (False, True):
        fl = flags
        if all_defined:  # the variants without tests: clean data, every level ALL_DEFINED
            fl = np.full(nlev, ALL, np.int32)
            u, v = synth.wind(nx, ny, 5252 + nx, nlev=nlev)
            z = np.stack([synth.scalar_field(nx, ny, 5300 + l) for l in range(nlev)])
        table = table_of(u, v, z)
        for name, cpu_op, f0, f1, use_fc, compute in table:
            o0 = dev(np.zeros_like(f0))
            o1 = dev(np.zeros_like(f0)) if name in ("vortdiv", "ilevelgwind") else None
            res = gpu_ctx.stencil_levels(name, dev(f0), dev(f1), dxm, dym, dfc if use_fc else None, fdefined=fl, out0=o0, out1=o1)
            assert res is not None, name
            if ragged_split == "1" and nx % 256 != 1:
                gpu_util.check_form(gpu_ctx, "wind_split_ragged" if f1 is not None else "scalar_split_ragged", what=name)
            else:
                gpu_util.check_form(gpu_ctx, "cell" if name == "gradient1" else "flat4", what=name)
            (r0, r1), fo = res
            r0 = r0.cpu().numpy()
            r1 = None if r1 is None else r1.cpu().numpy()
            for l in range(nlev):
                if name == "vortdiv":
                    ok, e0, f_e = oracle.call("relvort", nx, ny, f0[l], f1[l], xm, ym, fdefined=int(fl[l]))
                    ok, e1, _ = oracle.call("divergence", nx, ny, f0[l], f1[l], xm, ym, fdefined=int(fl[l]))
                else:
                    args = [f0[l]] + ([f1[l]] if f1 is not None else []) + [xm, ym] + ([fcor] if use_fc else []) + ([compute] if compute else [])
                    ok, e, f_e = oracle.call(cpu_op, nx, ny, *args, fdefined=int(fl[l]))
                    e0, e1 = (e if isinstance(e, tuple) else (e, None))
                assert cases.same_bits(r0[l], e0, nan_payload=False), (name, l, all_defined)
                if e1 is not None:
                    assert cases.same_bits(r1[l], e1, nan_payload=False), (name, l, all_defined)
                assert fo[l] == f_e, (name, l, fo[l], f_e)


@pytest.mark.parametrize("nx,ny,nlev,walk", [(516, 70, 6, "1"), (1440, 40, 9, "1"), (260, 11, 5, "1"), (64, 48, 2, None), (949, 23, 4, "1"), (1440, 75, 5, None)])
def test_vortdiv_ff_levels_three_outputs_in_one_pass(gpu_ctx, oracle, nx, ny, nlev, walk, mifc_env):
    """mifc_vortdiv_ff_levels_enqueue: relvort, divergence and vectorabs of every level == the three reference calls bit for
    bit, flags from the two counter arrays (count domains nx*ny - 2*nx and nx*ny).  walk="1" forces the split-role
    three-output kernel whatever the launch size (rows 0 / ny-1 of ff come from the halo slots there); the other shapes
    (two levels, a ragged width, a launch below the threshold) take the pair + batched vectorabs route."""
    import torch

    import mi_fieldcalc_amd as fc

    mifc_env("MIFC_LEVELWALK_MIN_UNITS", walk)
    u, v, xm, ym, flags = _levels_inputs(nx, ny, nlev, 5150 + nx)
    v[0, 0, 3] = cases.UNDEF  # an undefined value in row 0: counts for ff, not for the stencil outputs of that column's interior
    u[nlev - 1, ny - 1, nx - 1] = np.nan
    rv_e, dv_e, fo_e = _expect_levels(oracle, u, v, xm, ym, flags)
    du, dvv, dxm, dym = (torch.from_numpy(a).cuda() for a in (u, v, xm, ym))
    rv, dg, ff = torch.empty_like(du), torch.empty_like(du), torch.empty_like(du)
    cnt = torch.full((nlev,), 99, dtype=torch.int64, device="cuda")
    cnt_ff = torch.full((nlev,), 99, dtype=torch.int64, device="cuda")
    assert gpu_ctx.vortdiv_ff_levels_enqueue(du, dvv, dxm, dym, rv, dg, ff, fdefined=flags, n_undefined=cnt, n_undefined_ff=cnt_ff)
    if walk == "1" and nx % 4 == 0:
        gpu_util.check_form(gpu_ctx, "wind_split_ff")
    else:
        gpu_util.check_form(gpu_ctx, differs_from="wind_split_ff")
    torch.cuda.synchronize()
    assert cases.same_bits(rv.cpu().numpy(), rv_e, nan_payload=False) and cases.same_bits(dg.cpu().numpy(), dv_e, nan_payload=False)
    c, cf = cnt.cpu().numpy(), cnt_ff.cpu().numpy()
    ffh = ff.cpu().numpy()
    for l in range(nlev):
        ok, e, f_e = oracle.call("vectorabs", nx, ny, u[l], v[l], fdefined=int(flags[l]))
        assert ok and cases.same_bits(ffh[l], e, nan_payload=False), l
        assert (ALL if flags[l] == ALL else fc.classify(int(cf[l]), nx * ny)) == f_e, (l, cf[l], f_e)
        assert (ALL if flags[l] == ALL else fc.classify(int(c[l]), nx * ny - 2 * nx)) == fo_e[l], (l, c[l], fo_e[l])


@pytest.mark.parametrize("lanes", [1, 3])
def test_graph_of_per_level_calls_replays_on_new_input(oracle, lanes):
    """mifc_graph_begin / _end / _launch: a caller's loop of one asynchronous call per level -- the stencil pair and the fused
    derived batch, a level each -- recorded once and replayed on CHANGED inputs: every replay equals the per-level reference
    calls.  With three lanes the levels are recorded side by side (counters zeroed once at the head of the graph,
    mifc_counts_accumulate).  A synchronous call inside a capture invalidates it: mifc_graph_end says so, the context stays
    usable.  Context.prepare repeats the same C calls without the wrapper."""
    import torch

    import mi_fieldcalc_amd as fc
    import mi_fieldcalc_amd.synth as synth

    nx, ny, nlev = 516, 40, 6
    xm, ym, _ = synth.grid_maps(nx, ny)
    al, bl = synth.hybrid_levels(nlev)
    with fc.Context(0) as ctx:
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            dxm, dym = torch.from_numpy(xm).cuda(), torch.from_numpy(ym).cuda()
            du, dv = torch.zeros((nlev, ny, nx), device="cuda"), torch.zeros((nlev, ny, nx), device="cuda")
            dt, dq, dps = torch.zeros_like(du), torch.zeros_like(du), torch.zeros((ny, nx), device="cuda")
            rv, dg, ff, th, rh = (torch.empty_like(du) for _ in range(5))
            cnt = torch.zeros(nlev, dtype=torch.int64, device="cuda")
            cnt5 = torch.zeros(5 * nlev, dtype=torch.int64, device="cuda")
            some = np.full(1, SOME, np.int32)

            def per_level(g=None):
                for l in range(nlev):
                    if g is not None and lanes > 1:
                        g.lane(l % lanes)
                    assert ctx.stencil_levels_enqueue("vortdiv", du[l:l + 1], dv[l:l + 1], dxm, dym, None, rv[l:l + 1], dg[l:l + 1], fdefined=some,
                                                      n_undefined=cnt[l:l + 1])
                    ctx.hlevel_derived_batch(du[l:l + 1], dv[l:l + 1], dt[l:l + 1], dq[l:l + 1], dps, al[l:l + 1], bl[l:l + 1], temp=("", 3), hum=("", 1),
                                             fdef_wind=some, fdef_thermo=some, out={"ff": ff[l:l + 1], "temp": th[l:l + 1], "hum": rh[l:l + 1]},
                                             enqueue_counts=cnt5[5 * l:5 * l + 5])

            with ctx.graph_capture(max_levels_per_call=8, lanes=lanes) as g:
                if lanes > 1:
                    ctx.zero_counts_enqueue(cnt)
                    ctx.zero_counts_enqueue(cnt5)
                    ctx.counts_accumulate(True)
                per_level(g)
                ctx.counts_accumulate(False)
            prepared = ctx.prepare(per_level)
            assert len(prepared) == 2 * nlev
            for it, runner in enumerate((g.launch, g.launch, prepared.launch)):
                u, v = synth.wind(nx, ny, 700 + it, nlev=nlev)
                u[it] = synth.sprinkle_undef(u[it], it, 0.02)
                du.copy_(torch.from_numpy(u))
                dv.copy_(torch.from_numpy(v))
                t, q, ps = (x.cpu().numpy() for x in synth.device_thermo(nx, ny, nlev, 800 + it, "cuda"))
                dt.copy_(torch.from_numpy(t))
                dq.copy_(torch.from_numpy(q))
                dps.copy_(torch.from_numpy(ps))
                runner()
                torch.cuda.synchronize()
                c = cnt.cpu().numpy()
                for l in range(nlev):
                    ok, e, f = oracle.call("relvort", nx, ny, u[l], v[l], xm, ym, fdefined=SOME)
                    assert cases.same_bits(rv[l].cpu().numpy(), e, nan_payload=False) and fc.classify(int(c[l]), nx * ny - 2 * nx) == f, (it, l)
                    ok, e, f = oracle.call("divergence", nx, ny, u[l], v[l], xm, ym, fdefined=SOME)
                    assert cases.same_bits(dg[l].cpu().numpy(), e, nan_payload=False), (it, l)
                    ok, e, f = oracle.call("vectorabs", nx, ny, u[l], v[l], fdefined=SOME)
                    assert cases.same_bits(ff[l].cpu().numpy(), e, nan_payload=False) and fc.classify(int(cnt5[5 * l]), nx * ny) == f, (it, l)
                    ok, e, f = oracle.call("hleveltemp", nx, ny, t[l], ps, float(al[l]), float(bl[l]), "", 3, fdefined=SOME)
                    got = th[l].cpu().numpy()
                    assert np.all(np.abs(got.astype(np.float64) - e) <= 1e-5 * np.abs(e)), (it, l)
            g.close()
            # a synchronous entry point cannot be recorded
            with pytest.raises(RuntimeError):
                with ctx.graph_capture() as bad:
                    try:
                        ctx.relvort(du[0], dv[0], dxm, dym, fdefined=SOME)
                    except RuntimeError:
                        pass
            res = ctx.relvort(du[0], dv[0], dxm, dym, fdefined=SOME)  # ... and the context is still usable
            assert res is not None and torch.equal(res[0].view(torch.int32), rv[0].view(torch.int32))


def test_stencil_levels_enqueue_equals_the_synchronous_call(gpu_ctx, oracle):
    """mifc_stencil_levels_enqueue (nothing read back; counts stay on the device) against mifc_stencil_levels: same fields bit
    for bit, and classify(count, mifc_stencil_count_domain) gives the flags -- plevelgwind_xcomp excepted, which is
    NONE_DEFINED whatever the count (FieldCalculations.cc:664).  Mixed flags per level; one call with every level
    ALL_DEFINED and no counter array."""
    import torch

    import mi_fieldcalc_amd as fc
    import mi_fieldcalc_amd.synth as synth

    nx, ny, nlev = 516, 40, 6
    xm, ym, fcor = synth.grid_maps(nx, ny)
    u, v = synth.wind(nx, ny, 31, nlev=nlev)
    z = np.stack([synth.scalar_field(nx, ny, 900 + l) for l in range(nlev)])
    flags = np.array([ALL, SOME, SOME, ALL, SOME, SOME], np.int32)
    for l in (1, 4):
        u[l] = synth.sprinkle_undef(u[l], 3 + l, 0.03)
        z[l] = synth.sprinkle_undef(z[l], 13 + l, 0.03)
    z[2] = cases.UNDEF
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    du, dv, dz, dxm, dym, dfc = (dev(a) for a in (u, v, z, xm, ym, fcor))
    counts = torch.full((nlev,), 12345, dtype=torch.int64, device="cuda")
    for name, f0, f1, use_fc in [("vortdiv", du, dv, False), ("relvort", du, dv, False), ("absvort", du, dv, True), ("gradient1", dz, None, False),
                                 ("gradient3", dz, None, False), ("plevelgwind_xcomp", dz, None, True), ("plevelgvort", dz, None, True),
                                 ("ilevelgwind", dz, None, True), ("jacobian", dz, du, False)]:
        two = name in ("vortdiv", "ilevelgwind")
        (e0, e1), fe = gpu_ctx.stencil_levels(name, f0, f1, dxm, dym, dfc if use_fc else None, fdefined=flags)
        o0, o1 = torch.empty_like(du), (torch.empty_like(du) if two else None)
        assert gpu_ctx.stencil_levels_enqueue(name, f0, f1, dxm, dym, dfc if use_fc else None, o0, o1, fdefined=flags, n_undefined=counts), name
        torch.cuda.synchronize()
        assert torch.equal(o0.view(torch.int32), e0.view(torch.int32)), name
        if two:
            assert torch.equal(o1.view(torch.int32), e1.view(torch.int32)), name
        dom = gpu_ctx.stencil_count_domain(name, nx, ny)
        got = [fc.NONE_DEFINED if name == "plevelgwind_xcomp" else fc.classify(int(c), dom) for c in counts.cpu().numpy()]
        assert got == list(fe), (name, got, list(fe))
    # every level ALL_DEFINED: no tests, no counters
    clean_u, clean_v = synth.wind(nx, ny, 32, nlev=nlev)
    o0 = torch.empty_like(du)
    all_flags = np.full(nlev, ALL, np.int32)
    assert gpu_ctx.stencil_levels_enqueue("divergence", dev(clean_u), dev(clean_v), dxm, dym, None, o0, fdefined=all_flags, n_undefined=None)
    torch.cuda.synchronize()
    for l in (0, nlev - 1):
        ok, e, _ = oracle.call("divergence", nx, ny, clean_u[l], clean_v[l], xm, ym, fdefined=ALL)
        assert ok and cases.same_bits(o0[l].cpu().numpy(), e, nan_payload=False)
    # a tested call without a counter array is refused
    with pytest.raises(RuntimeError, match="n_undefined_dev"):
        gpu_ctx.stencil_levels_enqueue("divergence", du, dv, dxm, dym, None, o0, fdefined=flags, n_undefined=None)


@pytest.mark.parametrize("tiny", ["some", "all", "none"])
def test_gradients_with_map_factors_whose_halves_are_inexact(gpu_ctx, oracle, tiny, mifc_env):
    """(float)(0.5 * m * d) of the one-sided gradients (FieldCalculations.cc:2015, :2027, :2040): the level-walking kernels
    multiply by a pre-halved map factor where 0.5f * m is exact for every lane of a wave, and take the general route
    (half_prod: halve the larger factor) otherwise.  Map factors below 2^-125 -- subnormal halves -- next to fields of
    1e30 .. 1e38, so that the products are ordinary numbers and a wrongly halved factor shows; zeros and infinities too."""
    import mi_fieldcalc_amd.synth as synth

    mifc_env("MIFC_LEVELWALK_MIN_UNITS", "1")
    nx, ny, nlev = 516, 31, 5
    rng = np.random.default_rng(77)
    xm, ym, _ = synth.grid_maps(nx, ny)
    small = np.array([1e-39, 2.0 ** -126, 3.0 * 2.0 ** -127, 2.0 ** -149, 1.1754942e-38, 2.0 ** -125, 0.0, -(2.0 ** -126), np.inf], np.float32)
    if tiny != "none":
        pick = rng.random((ny, nx)) < (0.05 if tiny == "some" else 1.0)
        xm = np.where(pick, rng.choice(small, (ny, nx)), xm).astype(np.float32)
        ym = np.where(pick.T.reshape(-1)[: nx * ny].reshape(ny, nx), rng.choice(small, (ny, nx)), ym).astype(np.float32)
    z = (rng.standard_normal((nlev, ny, nx)) * 10.0 ** rng.uniform(30, 38, (nlev, ny, nx))).astype(np.float32)
    flags = np.array([ALL, SOME, ALL, SOME, ALL], np.int32)
    z[1] = synth.sprinkle_undef(z[1], 5, 0.02)
    with np.errstate(all="ignore"):
        for compute in (1, 2, 3):
            res = gpu_ctx.stencil_levels("gradient%d" % compute, z, None, xm, ym, None, fdefined=flags)
            assert res is not None
            (o0, _), fo = res
            for l in range(nlev):
                ok, e, f_e = oracle.call("gradient", nx, ny, z[l], xm, ym, compute, fdefined=int(flags[l]))
                assert ok and cases.same_bits(o0[l], e, nan_payload=False), (compute, l)
                assert fo[l] == f_e, (compute, l)


@pytest.mark.parametrize("where", ["row 11", "rows 3 and 20", "scattered", "everywhere"])
@pytest.mark.parametrize("nx", [516, 949])
def test_two_stage_kernels_with_map_factors_whose_halves_are_inexact(gpu_ctx, oracle, where, nx):
    """thermalFrontParameter and plevelqvector: the float-rounded partials (float)(0.5 * m * d) of both stages (half_prod) with map
    factors below 2^-125 -- where 0.5f * m is not exact -- in one row, in two rows, scattered, everywhere, next to fields of
    1e30 .. 1e37, so that the products are ordinary numbers and a wrongly halved factor shows.  (A form of the kernel that kept
    pre-halved map rows while they were exact passed this and was 5 % faster on one of the two operators: not kept.)"""
    ny, nlev = 31, 3
    rng = np.random.default_rng(78)
    import mi_fieldcalc_amd.synth as synth

    xm, ym, fcor = synth.grid_maps(nx, ny)
    small = np.array([1e-39, 2.0 ** -126, 3.0 * 2.0 ** -127, 2.0 ** -149, 1.1754942e-38, 2.0 ** -125, -(2.0 ** -126)], np.float32)
    pick = np.zeros((ny, nx), bool)
    if where == "row 11":
        pick[11, nx // 3] = True
    elif where == "rows 3 and 20":
        pick[3, 5] = pick[20, nx - 2] = True
    elif where == "scattered":
        pick = rng.random((ny, nx)) < 0.02
    else:
        pick[:] = True
    xm = np.where(pick, rng.choice(small, (ny, nx)), xm).astype(np.float32)
    ym = np.where(pick[::-1], rng.choice(small, (ny, nx)), ym).astype(np.float32)
    z = (rng.standard_normal((nlev, ny, nx)) * 10.0 ** rng.uniform(30, 37, (nlev, ny, nx))).astype(np.float32)
    t = (250.0 + rng.standard_normal((nlev, ny, nx)) * 10.0 ** rng.uniform(28, 33, (nlev, ny, nx))).astype(np.float32)
    flags = np.array([ALL, SOME, SOME], np.int32)
    z[1, 7:9, 10:20] = cases.UNDEF
    pres = np.array([850.0, 700.0, 500.0], np.float32)
    with np.errstate(all="ignore"):
        out, fo = gpu_ctx.stencil_levels_ex("thermalFrontParameter", z, xmapr=xm, ymapr=ym, fdefined=flags)
        for l in range(nlev):
            ok, e, f = oracle.call("thermalFrontParameter", nx, ny, z[l], xm, ym, fdefined=int(flags[l]))
            assert ok and cases.same_bits(out[l], e, nan_payload=False) and fo[l] == f, ("tfp", l)
        for c in (1, 2, 3, 4):
            out, fo = gpu_ctx.stencil_levels_ex("plevelqvector", z, t, None, xm, ym, fcor, level_scalars=pres, compute=c, fdefined=flags)
            for l in range(nlev):
                ok, e, f = oracle.call("plevelqvector", nx, ny, z[l], t[l], xm, ym, fcor, float(pres[l]), c, fdefined=int(flags[l]))
                assert ok and cases.same_bits(out[l], e, nan_payload=False) and fo[l] == f, ("qvector", c, l)


@pytest.mark.parametrize("nx,ny", [(64, 24), (1440, 37), (260, 9)])
def test_gradient_x_counts_the_outer_rows(gpu_ctx, oracle, nx, ny):
    """gradient compute=1 tests and counts over the flat cells 1 .. nx*ny-2 (FieldCalculations.cc:2013-2021): rows 0 and ny-1
    too, whose values fillEdges overwrites.  Levels whose ONLY undefined inputs touch those rows must still come back
    SOME_DEFINED (the row kernels do not walk them: a small count kernel behind the launch does), a clean level ALL_DEFINED."""
    import mi_fieldcalc_amd.synth as synth

    xm, ym, _ = synth.grid_maps(nx, ny)
    spots = [None, (0, 5), (ny - 1, 0), (1, 0), (0, 0), (ny - 1, nx - 1), (0, nx - 1), (ny - 1, nx - 2)]
    z = np.stack([synth.scalar_field(nx, ny, 40 + l) for l in range(len(spots))])
    for l, sp in enumerate(spots):
        if sp is not None:
            z[l][sp] = cases.UNDEF
    flags = np.full(len(spots), SOME, np.int32)
    res = gpu_ctx.stencil_levels("gradient1", z, None, xm, ym, None, fdefined=flags)
    assert res is not None
    (o0, _), fo = res
    for l in range(len(spots)):
        ok, e, f_e = oracle.call("gradient", nx, ny, z[l], xm, ym, 1, fdefined=SOME)
        assert ok and cases.same_bits(o0[l], e, nan_payload=False), l
        assert fo[l] == f_e, (l, spots[l], fo[l], f_e)
        # and the single-field call
        got = gpu_ctx.gradient(z[l], xm, ym, 1, fdefined=SOME)
        assert got is not None and cases.same_bits(got[0], e, nan_payload=False) and got[1] == f_e, (l, spots[l], got[1], f_e)
    assert fo[0] == ALL and all(f == SOME for f in fo[1:]), fo


# ------------------------------------------------------------------ fused stencil-of-a-stencil kernels (mifc_fused2.hip)
@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("device", [False, True])
def test_fused_tfp_and_qvector(gpu_ctx, oracle, fused, device, mifc_env):
    """One launch with the intermediate fields in LDS == the reference's pass-by-pass result, flags
    included; MIFC_FUSED2=0 runs the multi-pass path on the same cases."""
    mifc_env("MIFC_FUSED2", fused)
    for case in cases.fused2_cases():
        _check_case(gpu_ctx, oracle, case, device=device)


def test_shared_reciprocal_division_is_the_plain_division():
    """The fused kernels divide two numerators by one denominator through one refined reciprocal
    (csrc/mifc_device.h); for float-born operands that must be the f64 division, bit for bit --
    including zeros, infinities, NaNs, denormals and the ends of the float range."""
    import torch

    n = 1 << 22
    gen = torch.Generator().manual_seed(2024)
    bits = lambda: torch.randint(-(2 ** 31), 2 ** 31 - 1, (n,), generator=gen, dtype=torch.int64).to(torch.int32).view(torch.float32)
    special = torch.tensor([0.0, -0.0, float("inf"), -float("inf"), float("nan"), 1e-45, -1e-45, 1.17549435e-38, 3.4028235e38, -3.4028235e38,
                            1.0, -1.0, 3.0, 1e35, 9.8, 1e-5, 2.5e-5, 1.4e-4], dtype=torch.float32)
    ops = []
    for k in range(3):
        x = bits()  # every bit pattern is a float: all exponents, denormals, NaNs
        ordinary = (torch.rand(n, generator=gen) * 200 - 100) * 10 ** torch.randint(-6, 6, (n,), generator=gen).float()
        x = torch.where(torch.rand(n, generator=gen) < 0.5, x, ordinary)
        x[: special.numel() ** 2] = special.repeat_interleave(special.numel()) if k == 0 else special.repeat(special.numel())
        ops.append(x.cuda())
    ops[1][: special.numel() ** 2] = 1.0
    shared = torch.empty(n, device="cuda")
    plain = torch.empty(n, device="cuda")
    # mifc_diag_division lives in the measurement build (include/mifc_measure.h): the same device functions, compiled from the
    # same header, in libmifc_measure.so -- loaded here next to the product library, with a context of its own
    import ctypes

    measure = ctypes.CDLL(os.path.join(ROOT, "mi-fieldcalc_amd", "libmifc_measure.so"))
    measure.mifc_create.restype = ctypes.c_void_p
    mctx = ctypes.c_void_p(measure.mifc_create(0))
    assert mctx.value
    torch.cuda.synchronize()
    ok = measure.mifc_diag_division(mctx, *[ctypes.c_void_p(t.data_ptr()) for t in (ops[0], ops[1], ops[2], shared, plain)], ctypes.c_size_t(n))
    measure.mifc_synchronize(mctx)
    measure.mifc_destroy(mctx)
    assert ok
    torch.cuda.synchronize()
    s, p = shared.cpu().numpy(), plain.cpu().numpy()
    assert cases.same_bits(s, p, nan_payload=False)
    assert np.isfinite(p).sum() > n // 4  # the sweep is not all overflow


def test_fused_tfp_and_qvector_random_shapes(gpu_ctx, oracle):
    """Seeded random widths (multiples of 4, so the fused kernel runs), heights around the band sizes
    the launcher picks, and flag modes."""
    import mi_fieldcalc_amd.synth as synth

    rng = np.random.default_rng(77)
    for k in range(48):
        nx = 4 * int(rng.integers(1, 200)) if k % 4 else 4 * int(rng.integers(200, 1025))
        ny = int(rng.choice([3, 4, 5, 9, 10, 11, 17, 18, 19, 26, 27, 64, 67, 131]))
        if nx * ny > 400000:
            ny = max(3, 400000 // nx)
        mode = cases.MODES[k % len(cases.MODES)]
        seed = 5000 + k
        xm, ym, fc = synth.grid_maps(nx, ny)
        z = synth.scalar_field(nx, ny, seed)
        (z_,), flag = cases._apply_mode([z], mode, seed, cases._frac(nx, ny))
        base = dict(nx=nx, ny=ny, fdefined=flag, undef=cases.UNDEF)
        lab = "%dx%d-%s" % (nx, ny, mode)
        _check_case(gpu_ctx, oracle, dict(base, op="thermalFrontParameter", args=[z_, xm, ym], label="tfp-" + lab), device=bool(k & 1))
        bad = (z_ == cases.UNDEF) | np.isnan(z_)
        with np.errstate(all="ignore"):
            tq = np.where(bad, z_, np.float32(250.0) + np.float32(0.05) * (z_ - np.float32(5500.0))).astype(np.float32)
        c = 1 + k % 4
        _check_case(gpu_ctx, oracle, dict(base, op="plevelqvector", args=[z_, tq, xm, ym, fc, 850.0, c], label="qvector%d-%s" % (c, lab)),
                    device=bool(k & 1))


@pytest.mark.parametrize("mode", ["all", "some"])
def test_one_launch_kernels_equal_their_multi_pass_paths_on_a_large_field(gpu_ctx, mode, mifc_env):
    """1440 x 11520 (16 levels seen as one tall field, far beyond what the CPU oracle checks in seconds): the
    one-launch forms of thermalFrontParameter, plevelqvector and shapiro2_filter give bit for bit what the
    pass-by-pass kernels give (which the seeded cases pin to the reference), flags included."""
    import torch

    import mi_fieldcalc_amd.synth as synth

    nx, ny, nlev = 1440, 720, 16
    xm, ym, fcor = synth.grid_maps(nx, ny)
    tall = lambda a: torch.from_numpy(np.ascontiguousarray(np.tile(a, (nlev, 1)))).cuda()
    dxm, dym, dfc = tall(xm), tall(ym), tall(fcor)
    z = np.concatenate([synth.scalar_field(nx, ny, 900 + l) for l in range(nlev)], axis=0)
    flag = ALL
    if mode == "some":
        z = synth.sprinkle_undef(z, 17, 0.001)
        z[5000:5003, 700:720] = np.float32(5432.0)  # a small plateau: |grad| == 0
        flag = SOME
    dz = torch.from_numpy(z).cuda()
    dt = torch.where((dz == float(cases.UNDEF)) | torch.isnan(dz), dz, 250.0 + 0.05 * (dz - 5500.0)).contiguous()

    def run(env):
        for k, v in env.items():
            mifc_env(k, v)
        out = {}
        out["tfp"] = gpu_ctx.thermalFrontParameter(dz, dxm, dym, fdefined=flag)
        for c in (1, 4):
            out["q%d" % c] = gpu_ctx.plevelqvector(dz, dt, dxm, dym, dfc, 700.0, c, fdefined=flag)
        out["shapiro"] = gpu_ctx.shapiro2_filter(dz, fdefined=flag)
        return {k: (v[0].cpu().numpy(), v[1]) for k, v in out.items()}

    one = run({"MIFC_FUSED2": "1", "MIFC_SHAPIRO_FUSED": "1"})
    many = run({"MIFC_FUSED2": "0", "MIFC_SHAPIRO_FUSED": "0"})
    for k in one:
        assert one[k][1] == many[k][1], (k, one[k][1], many[k][1])
        assert cases.same_bits(one[k][0], many[k][0], nan_payload=False), k


# ------------------------------------------------------------------ the f1 operators over level batches (shared map factors)
@pytest.mark.parametrize("nx,ny,nlev", [(64, 48, 5), (129, 21, 3), (1440, 37, 6), (240, 9, 4), (949, 12, 5)])
@pytest.mark.parametrize("device", [False, True])
def test_stencil_levels_ex_f1_operators(gpu_ctx, oracle, nx, ny, nlev, device):
    """mifc_stencil_levels_ex: advection, thermalFrontParameter, plevelqvector (pressure per level) and
    shapiro2_filter over a batch == the per-level reference call, flags included.  Mixed input flags in
    one batch (the two-stage kernels run the ALL_DEFINED levels and the tested ones as separate launches),
    a level whose flag lies, and a width the one-launch kernels do not take (129: level-by-level path)."""
    import torch

    import mi_fieldcalc_amd.synth as synth

    xm, ym, fcor = synth.grid_maps(nx, ny)
    u, v = synth.wind(nx, ny, 808 + nx, nlev=nlev)
    z = np.stack([synth.scalar_field(nx, ny, 900 + l) for l in range(nlev)])
    flags = np.full(nlev, SOME, np.int32)
    flags[0] = ALL
    for l in range(1, nlev):
        if l % 2:
            z[l] = synth.sprinkle_undef(z[l], 20 + l, 0.03)
            u[l] = synth.sprinkle_undef(u[l], 30 + l, 0.03)
    if nlev > 2:
        flags[nlev - 1] = ALL  # clean or not, the caller promises ALL_DEFINED: no tests
        z[2, ny // 3:, : max(3, nx // 2)] = np.float32(5432.0)  # a plateau: |grad| == 0 is rejected whatever the flag says
    with np.errstate(all="ignore"):
        bad = (z == cases.UNDEF) | np.isnan(z)
        t = np.where(bad, z, np.float32(250.0) + np.float32(0.05) * (z - np.float32(5500.0))).astype(np.float32)
    pres = np.array([1000.0, 850.0, 700.0, 500.0, 300.0, 250.0][:nlev], np.float32)
    dev = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()) if device else (lambda a: a)
    host = (lambda a: a.cpu().numpy()) if device else (lambda a: a)
    dxm, dym, dfc = dev(xm), dev(ym), dev(fcor)

    def check(name, res, per_level):
        assert res is not None, name
        out, fo = res
        out = host(out)
        for l in range(nlev):
            ok, e, f = per_level(l)
            assert ok and cases.same_bits(out[l], e, nan_payload=False), (name, l)
            assert fo[l] == f, (name, l, fo[l], f)

    check("advection", gpu_ctx.stencil_levels_ex("advection", dev(z), dev(u), dev(v), dxm, dym, scalar=1.0 / 3600.0, fdefined=flags),
          lambda l: oracle.call("advection", nx, ny, z[l], u[l], v[l], xm, ym, 1.0 / 3600.0, fdefined=int(flags[l])))
    check("tfp", gpu_ctx.stencil_levels_ex("thermalFrontParameter", dev(z), xmapr=dxm, ymapr=dym, fdefined=flags),
          lambda l: oracle.call("thermalFrontParameter", nx, ny, z[l], xm, ym, fdefined=int(flags[l])))
    for c in (1, 2, 3, 4):
        check("qvector%d" % c, gpu_ctx.stencil_levels_ex("plevelqvector", dev(z), dev(t), None, dxm, dym, dfc, level_scalars=pres, compute=c, fdefined=flags),
              lambda l: oracle.call("plevelqvector", nx, ny, z[l], t[l], xm, ym, fcor, float(pres[l]), c, fdefined=int(flags[l])))
    check("shapiro", gpu_ctx.stencil_levels_ex("shapiro2_filter", dev(z), fdefined=flags),
          lambda l: oracle.call("shapiro2_filter", nx, ny, z[l], fdefined=int(flags[l])))
    # invalid arguments -> false, like the per-level calls
    assert gpu_ctx.stencil_levels_ex("plevelqvector", dev(z), dev(t), None, dxm, dym, dfc, level_scalars=-pres, compute=1, fdefined=flags) is None
    assert gpu_ctx.stencil_levels_ex("plevelqvector", dev(z), dev(t), None, dxm, dym, dfc, level_scalars=pres, compute=7, fdefined=flags) is None


@pytest.mark.parametrize("nx,ny,nlev", [(516, 40, 5), (1440, 27, 7), (64, 15, 3), (260, 14, 9), (949, 29, 4), (515, 13, 3), (1442, 14, 5)])
def test_advection_level_batch_on_the_split_role_kernel(gpu_ctx, oracle, nx, ny, nlev, mifc_env):
    """advection over a deep batch: loader waves bring f (with halo rows) and the tile's rows of u and v into LDS, compute waves
    read LDS and store (advection_split_kernel); per-level reference call bit for bit, mixed flags, undefined values in f, u
    and v (each one's own test), ragged last tile, a batch that ends where its allocation ends."""
    import torch

    import mi_fieldcalc_amd.synth as synth

    mifc_env("MIFC_LEVELWALK_MIN_UNITS", "1")
    xm, ym, _ = synth.grid_maps(nx, ny)
    u, v = synth.wind(nx, ny, 77 + nx, nlev=nlev)
    z = np.stack([synth.scalar_field(nx, ny, 800 + l) for l in range(nlev)])
    flags = np.full(nlev, SOME, np.int32)
    flags[0] = ALL
    for l in range(1, nlev):
        f = (z, u, v)[l % 3]
        f[l] = synth.sprinkle_undef(f[l], 40 + l, 0.03)
    v[nlev - 1, ny - 2, nx - 1] = cases.UNDEF
    z[nlev - 1, ny - 1, nx - 2] = cases.UNDEF
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    for fl in (flags, np.full(nlev, ALL, np.int32)):
        if fl is not flags:
            u, v = synth.wind(nx, ny, 78 + nx, nlev=nlev)
            z = np.stack([synth.scalar_field(nx, ny, 850 + l) for l in range(nlev)])
        res = gpu_ctx.stencil_levels_ex("advection", dev(z), dev(u), dev(v), dev(xm), dev(ym), scalar=1.0 / 3600.0, fdefined=fl)
        assert res is not None
        gpu_util.check_form(gpu_ctx, "advection_split" if nx % 4 == 0 else "advection_split_ragged")
        out, fo = res
        out = out.cpu().numpy()
        for l in range(nlev):
            ok, e, f = oracle.call("advection", nx, ny, z[l], u[l], v[l], xm, ym, 1.0 / 3600.0, fdefined=int(fl[l]))
            assert ok and cases.same_bits(out[l], e, nan_payload=False), l
            assert fo[l] == f, (l, fo[l], f)
    mifc_env("MIFC_VORTDIV_SPLIT", "0")
    assert gpu_ctx.stencil_levels_ex("advection", dev(z), dev(u), dev(v), dev(xm), dev(ym), scalar=1.0 / 3600.0, fdefined=fl) is not None
    gpu_util.check_form(gpu_ctx, "advection_oneshot" if nx % 4 == 0 else "cell")


def test_shapiro_levels_in_place(gpu_ctx, oracle):
    import torch

    import mi_fieldcalc_amd.synth as synth

    nx, ny, nlev = 484, 30, 3
    z = np.stack([synth.scalar_field(nx, ny, 70 + l) for l in range(nlev)])
    z[1] = synth.sprinkle_undef(z[1], 3, 0.02)
    flags = np.array([ALL, SOME, SOME], np.int32)
    dz = torch.from_numpy(z.copy()).cuda()
    out, fo = gpu_ctx.stencil_levels_ex("shapiro2_filter", dz, fdefined=flags, out0=dz)
    assert out.data_ptr() == dz.data_ptr() and np.all(fo == ALL)
    for l in range(nlev):
        ok, e, _ = oracle.call("shapiro2_filter", nx, ny, z[l], fdefined=int(flags[l]))
        assert cases.same_bits(dz[l].cpu().numpy(), e, nan_payload=False), l


def test_winddir_extension(gpu_ctx, oracle):
    """EXTENSION, no reference function: BASELINE.json names "wind direction from u/v", the reference has
    none.  Pinned against its DEFINITION restated on the CPU in float64 and rounded once (oracle/mifc_oracle.cc:
    mifcorc_winddir -- the meteorological direction the wind blows from, 270 - atan2(v, u) * 180 / pi in [0, 360), calm -> 0):
    1e-5 relative (BASELINE.json) on every cell, directions next to north included, a handful of ulps at most; the calm and
    axis cases exactly; undefined placement and flag as vectorabs."""
    import torch

    import mi_fieldcalc_amd.synth as synth
    from test_oracle_golden import ulp_diff

    nx, ny = 1440, 90
    u, v = synth.wind(nx, ny, 4711)
    u[0, :8] = [0.0, 0.0, 5.0, -5.0, 3.0, -3.0, 0.0, 1e-20]
    v[0, :8] = [0.0, 5.0, 0.0, 0.0, 3.0, -3.0, -5.0, -1e-20]
    # a fan of directions around north (dd -> 0 and dd -> 360), where the subtraction form loses everything
    k = np.arange(200)
    u[1, :200] = np.float32(-10.0) * np.sin(np.float32(10.0) ** (-k / 25.0)).astype(np.float32) * np.where(k % 2, 1, -1).astype(np.float32)
    v[1, :200] = np.float32(-10.0)
    ok, expect, flag_e = oracle.call("winddir", nx, ny, u, v, fdefined=ALL)
    assert ok and flag_e == ALL
    worst_ulp = 0
    for device in (False, True):
        a = [torch.from_numpy(x).cuda() for x in (u, v)] if device else [u, v]
        dd, flag = gpu_ctx.winddir(*a, fdefined=ALL)
        dd = dd.cpu().numpy() if device else dd
        assert flag == ALL and dd.min() >= 0.0 and dd.max() < 360.0
        err = np.abs(dd.astype(np.float64) - expect.astype(np.float64))
        wrap = err > 359.0  # 359.99998 against 0: one float spacing of 360 apart on the circle
        assert np.all((360.0 - err[wrap]) <= 3.1e-5), (360.0 - err[wrap]).max()
        rel = err[~wrap] / np.maximum(np.abs(expect[~wrap].astype(np.float64)), 1e-30)
        rel[expect[~wrap] == 0] = np.where(dd[~wrap][expect[~wrap] == 0] == 0, 0.0, np.inf)
        assert rel.max() <= 1e-5, rel.max()
        worst_ulp = max(worst_ulp, int(ulp_diff(dd[~wrap].reshape(-1), expect[~wrap].reshape(-1)).max()))
        assert list(dd[0, :7]) == [0.0, 180.0, 270.0, 90.0, 225.0, 45.0, 0.0]  # calm, from S, from W, from E, from SW, from NE, from N
    assert worst_ulp <= 8, worst_ulp
    uu = synth.sprinkle_undef(u, 5, 0.01)
    dd, flag = gpu_ctx.winddir(uu, v, fdefined=SOME)
    ok, e, f_e = oracle.call("winddir", nx, ny, uu, v, fdefined=SOME)
    bad = (uu == cases.UNDEF) | np.isnan(uu)
    assert flag == f_e == SOME and np.array_equal(dd == cases.UNDEF, e == cases.UNDEF) and np.all(dd[bad] == cases.UNDEF) and np.all(dd[~bad] != cases.UNDEF)


@pytest.mark.parametrize("force_cell", ["0", "1"])
def test_one_input_stencils_on_ragged_widths(gpu_ctx, oracle, force_cell, mifc_env):
    """gradient (all four), geostrophic wind / vorticity, ilevelgwind on widths that are not a multiple of 4, large enough for
    whole waves of the flat four-cells-per-lane kernel (and the one-lane-per-cell kernel under MIFC_FORCE_CELL_KERNEL=1)."""
    mifc_env("MIFC_FORCE_CELL_KERNEL", force_cell)
    ops = ("gradient", "plevelgwind_xcomp", "plevelgwind_ycomp", "plevelgvort", "ilevelgwind", "relvort", "divergence", "absvort", "jacobian")
    n = 0
    for case in cases.stencil_cases(grids=[(949, 23), (1001, 7), (258, 9), (6, 40)]):
        if case["op"] in ops:
            _check_case(gpu_ctx, oracle, case, device=True)
            n += 1
    assert n >= 4 * 4 * 12
