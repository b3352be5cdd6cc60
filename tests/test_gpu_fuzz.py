"""Seeded random cases for the batched stencil operators: widths around the kernels' segment and alignment boundaries, short
and ragged heights, shallow and deeper batches, mixed per-level flags, fields off the 16-byte grid, the path-selecting
switches drawn at random -- every operator of mifc_stencil_levels against the per-level reference call, bit for bit, flags
included.  The fixed cases elsewhere name what they test; this file looks where nobody thought to look (the tested
plevelgwind_xcomp split-role kernel's missing x-neighbours, round 3, would have shown in any of these cases).

    python tests/test_gpu_fuzz.py 500      -> the same generator over 500 cases (GPU box), first mismatch reported"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cases  # noqa: E402

pytestmark = pytest.mark.gpu

ALL, SOME = cases.ALL_DEFINED, cases.SOME_DEFINED
WIDTHS = [8, 9, 12, 17, 33, 64, 100, 241, 252, 255, 256, 257, 258, 260, 263, 511, 512, 513, 516, 770, 949, 1000, 1023, 1024, 1025, 1029, 1440, 1443]
SWITCHES = [("MIFC_LEVELWALK_MIN_UNITS", [None, "1", "1", "1"]), ("MIFC_RAGGED_SPLIT", [None, None, "0"]), ("MIFC_VORTDIV_SPLIT", [None, None, None, "0"]),
            ("MIFC_VORTDIV_LEVELWALK", [None, None, None, "0"]), ("MIFC_FORCE_CELL_KERNEL", [None, None, None, None, "1"]),
            ("MIFC_SCALAR_SPLIT_TUNE", [None, None, "TR=14,NL=2,PF=2", "TR=12,NL=2,PF=3,LG=2", "TR=8,NL=2,PF=2,LG=3"])]
OPS = [("vortdiv", None, 2, False, None), ("relvort", "relvort", 2, False, None), ("divergence", "divergence", 2, False, None), ("absvort", "absvort", 2, True, None),
       ("gradient1", "gradient", 1, False, 1), ("gradient2", "gradient", 1, False, 2), ("gradient3", "gradient", 1, False, 3), ("gradient4", "gradient", 1, False, 4),
       ("plevelgwind_xcomp", "plevelgwind_xcomp", 1, True, None), ("plevelgwind_ycomp", "plevelgwind_ycomp", 1, True, None),
       ("plevelgvort", "plevelgvort", 1, True, None), ("ilevelgwind", "ilevelgwind", 1, True, None), ("jacobian", "jacobian", 2, False, None)]


def make_case(index):
    rng = np.random.default_rng(0xF0220000 + index)
    nx = int(rng.choice(WIDTHS))
    ny = int(rng.choice([3, 4, 5, 7, 9, 13, 14, 15, 16, 17, 25, 29, 40, 61]))
    nlev = int(rng.choice([1, 2, 3, 3, 4, 5, 7, 9]))
    if nx * ny * nlev > 400000:
        nlev = max(1, 400000 // (nx * ny))
    env = {name: str(v) if v is not None else None for name, vals in SWITCHES for v in [vals[int(rng.integers(len(vals)))]]}
    return dict(index=index, nx=nx, ny=ny, nlev=nlev, env=env, shift=int(rng.choice([0, 0, 0, 1, 2, 3])), density=float(rng.choice([0.0, 0.002, 0.03, 0.3])),
                all_levels=[bool(rng.integers(3) == 0) for _ in range(nlev)], seed=int(rng.integers(1 << 30)), nan_undef=bool(rng.integers(12) == 0))


def run_case(ctx, oracle, case, set_env):
    import torch

    import mi_fieldcalc_amd.synth as synth

    nx, ny, nlev = case["nx"], case["ny"], case["nlev"]
    for name, value in case["env"].items():
        set_env(name, value)
    undef = float("nan") if case["nan_undef"] else cases.UNDEF
    xm, ym, fcor = synth.grid_maps(nx, ny)
    u, v = synth.wind(nx, ny, case["seed"], nlev=nlev)
    z = np.stack([synth.scalar_field(nx, ny, case["seed"] + 17 + l) for l in range(nlev)])
    if nlev == 1:
        u, v = u.reshape(1, ny, nx), v.reshape(1, ny, nx)
    flags = np.full(nlev, SOME, np.int32)
    rng = np.random.default_rng(case["seed"])
    for l in range(nlev):
        if case["all_levels"][l]:
            flags[l] = ALL  # clean data under ALL_DEFINED
        elif case["density"] > 0:
            for f in (u, v, z):
                mask = rng.random((ny, nx)) < case["density"]
                f[l][mask] = undef
                f[l][mask & (rng.random((ny, nx)) < 0.2)] = np.nan
    shift = case["shift"]

    def dev(a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        big = torch.empty(a.size + shift, dtype=torch.float32, device="cuda")
        t = big[shift:].view(a.shape)
        t.copy_(torch.from_numpy(a))
        return t

    du, dv, dz, dxm, dym, dfc = dev(u), dev(v), dev(z), dev(xm), dev(ym), dev(fcor)
    for name, cpu_op, nin, use_fc, compute in OPS:
        f0 = u if nin == 2 else z
        f1 = v if nin == 2 else None
        if name == "jacobian":
            f0, f1 = z, u
        d0 = dz if f0 is z else du
        d1 = None if f1 is None else (dv if f1 is v else du)
        res = ctx.stencil_levels(name, d0, d1, dxm, dym, dfc if use_fc else None, fdefined=flags, undef=undef)
        assert res is not None, (case, name, ctx.last_error())
        (o0, o1), fo = res
        o0 = o0.cpu().numpy()
        o1 = None if o1 is None else o1.cpu().numpy()
        form = ctx.last_stencil_form()
        for l in range(nlev):
            if name == "vortdiv":
                ok, e0, f_e = oracle.call("relvort", nx, ny, f0[l], f1[l], xm, ym, fdefined=int(flags[l]), undef=undef)
                ok, e1, _ = oracle.call("divergence", nx, ny, f0[l], f1[l], xm, ym, fdefined=int(flags[l]), undef=undef)
            else:
                args = [f0[l]] + ([f1[l]] if f1 is not None else []) + [xm, ym] + ([fcor] if use_fc else []) + ([compute] if compute else [])
                ok, e, f_e = oracle.call(cpu_op, nx, ny, *args, fdefined=int(flags[l]), undef=undef)
                e0, e1 = (e if isinstance(e, tuple) else (e, None))
            where = (case, name, l, form)
            assert cases.same_bits(o0[l], e0, nan_payload=False), where
            if e1 is not None:
                assert cases.same_bits(o1[l], e1, nan_payload=False), where
            assert fo[l] == f_e, where + (int(fo[l]), int(f_e))


F1_SWITCHES = [("MIFC_LEVELWALK_MIN_UNITS", [None, "1", "1"]), ("MIFC_VORTDIV_SPLIT", [None, None, None, "0"]), ("MIFC_FUSED2", [None, None, None, "0"]), ("MIFC_SHAPIRO_FUSED", [None, None, None, "0"]), ("MIFC_SHAPIRO_REGS", [None, None, "0"]),
               ("MIFC_FUSED2_BAND", [None, None, "3", "8", "17"]), ("MIFC_FORCE_CELL_KERNEL", [None, None, None, None, "1"])]


def make_f1_case(index):
    rng = np.random.default_rng(0xF1F10000 + index)
    nx = int(rng.choice(WIDTHS))
    ny = int(rng.choice([3, 4, 5, 6, 7, 9, 13, 16, 17, 25, 29, 40, 61]))
    nlev = int(rng.choice([1, 2, 3, 4, 6]))
    if nx * ny * nlev > 300000:
        nlev = max(1, 300000 // (nx * ny))
    env = {name: vals[int(rng.integers(len(vals)))] for name, vals in F1_SWITCHES}
    return dict(index=index, nx=nx, ny=ny, nlev=nlev, env=env, density=float(rng.choice([0.0, 0.002, 0.03, 0.3])),
                all_levels=[bool(rng.integers(3) == 0) for _ in range(nlev)], seed=int(rng.integers(1 << 30)), plateau=bool(rng.integers(3) == 0),
                device=bool(rng.integers(4) != 0))


def run_f1_case(ctx, oracle, case, set_env):
    """advection, thermalFrontParameter, plevelqvector (compute 1..4) and shapiro2_filter over a batch (mifc_stencil_levels_ex)."""
    import torch

    import mi_fieldcalc_amd.synth as synth

    nx, ny, nlev = case["nx"], case["ny"], case["nlev"]
    for name, value in case["env"].items():
        set_env(name, value)
    xm, ym, fcor = synth.grid_maps(nx, ny)
    u, v = synth.wind(nx, ny, case["seed"], nlev=nlev)
    u, v = u.reshape(nlev, ny, nx), v.reshape(nlev, ny, nx)
    z = np.stack([synth.scalar_field(nx, ny, case["seed"] + 17 + l) for l in range(nlev)])
    flags = np.full(nlev, SOME, np.int32)
    rng = np.random.default_rng(case["seed"])
    if case["plateau"]:
        z[nlev // 2, ny // 3:, : max(3, nx // 2)] = np.float32(5432.0)  # |grad| == 0 is rejected whatever the flag says
    for l in range(nlev):
        if case["all_levels"][l]:
            flags[l] = ALL
        elif case["density"] > 0:
            for f in (u, z):
                mask = rng.random((ny, nx)) < case["density"]
                f[l][mask] = cases.UNDEF
                f[l][mask & (rng.random((ny, nx)) < 0.2)] = np.nan
    with np.errstate(all="ignore"):
        bad = (z == cases.UNDEF) | np.isnan(z)
        t = np.where(bad, z, np.float32(250.0) + np.float32(0.05) * (z - np.float32(5500.0))).astype(np.float32)
    pres = np.array([1000.0, 850.0, 700.0, 500.0, 300.0, 250.0][:nlev], np.float32)
    dev = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()) if case["device"] else (lambda a: a)
    host = (lambda a: a.cpu().numpy()) if case["device"] else (lambda a: a)
    dxm, dym, dfc = dev(xm), dev(ym), dev(fcor)

    def check(name, res, per_level):
        assert res is not None, (case, name, ctx.last_error())
        out, fo = res
        out = host(out)
        for l in range(nlev):
            ok, e, f = per_level(l)
            assert ok and cases.same_bits(out[l], e, nan_payload=False), (case, name, l)
            assert fo[l] == f, (case, name, l, int(fo[l]), int(f))

    check("advection", ctx.stencil_levels_ex("advection", dev(z), dev(u), dev(v), dxm, dym, scalar=1.0 / 3600.0, fdefined=flags),
          lambda l: oracle.call("advection", nx, ny, z[l], u[l], v[l], xm, ym, 1.0 / 3600.0, fdefined=int(flags[l])))
    check("tfp", ctx.stencil_levels_ex("thermalFrontParameter", dev(z), xmapr=dxm, ymapr=dym, fdefined=flags),
          lambda l: oracle.call("thermalFrontParameter", nx, ny, z[l], xm, ym, fdefined=int(flags[l])))
    for c in (1, 2, 3, 4):
        check("qvector%d" % c, ctx.stencil_levels_ex("plevelqvector", dev(z), dev(t), None, dxm, dym, dfc, level_scalars=pres, compute=c, fdefined=flags),
              lambda l: oracle.call("plevelqvector", nx, ny, z[l], t[l], xm, ym, fcor, float(pres[l]), c, fdefined=int(flags[l])))
    check("shapiro", ctx.stencil_levels_ex("shapiro2_filter", dev(z), fdefined=flags), lambda l: oracle.call("shapiro2_filter", nx, ny, z[l], fdefined=int(flags[l])))


def check_seeded_case(ctx, oracle, case, device):
    """One case of tests/cases.py (any operator family) against the CPU checker, as test_gpu_parity.py judges it."""
    import gpu_util

    ok_e, out_e, flag_e = cases.run_cpu(oracle, case)
    ok, out, flag = gpu_util.run_gpu(ctx, case, device=device)
    assert ok == ok_e, case["label"]
    if not ok_e:
        return
    exact = not gpu_util.uses_device_powf(case)
    outs_e = list(out_e) if isinstance(out_e, tuple) else [out_e]
    outs = list(out) if isinstance(out, tuple) else [out]
    for a, b in zip(outs, outs_e):
        gpu_util.compare(case, np.asarray(a), np.asarray(b), exact)
    assert flag == flag_e, "%s: flag %d vs %d" % (case["label"], flag, flag_e)


def run_grid_case(ctx, oracle, index):
    """The seeded cases of the elementwise operators, the pointwise catalogue and the ensemble reductions on a random grid:
    cell counts that are not multiples of 4 (the scalar tail launch), one row, one column, host or device memory."""
    rng = np.random.default_rng(0xE7150000 + index)
    nx = int(rng.choice([1, 2, 3, 5, 8, 17, 63, 64, 65, 127, 250, 257, 511]))
    ny = int(rng.choice([1, 2, 3, 4, 7, 16, 33, 61]))
    device = bool(rng.integers(2))
    kind = int(rng.integers(3))
    gen = (cases.ewise_cases, cases.catalogue_cases, cases.ensemble_cases)[kind]
    for case in gen(grids=((nx, ny),)):
        check_seeded_case(ctx, oracle, case, device)


@pytest.mark.parametrize("index", range(24))
def test_seeded_operator_cases_on_random_grids(gpu_ctx, oracle, index):
    run_grid_case(gpu_ctx, oracle, index)


@pytest.mark.parametrize("index", range(64))
def test_random_stencil_batches_equal_the_reference(gpu_ctx, oracle, index, mifc_env):
    run_case(gpu_ctx, oracle, make_case(index), mifc_env)


@pytest.mark.parametrize("index", range(48))
def test_random_f1_batches_equal_the_reference(gpu_ctx, oracle, index, mifc_env):
    run_f1_case(gpu_ctx, oracle, make_f1_case(index), mifc_env)


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import cpulib
    import mi_fieldcalc_amd as fc

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    ctx, orc = fc.Context(0), cpulib.CpuLib("oracle")
    touched = set()

    def set_env(name, value):
        touched.add(name)
        if value is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = value
        ctx.reload_env()

    for i in range(first, first + n):
        for name in touched:
            os.environ.pop(name, None)
        run_case(ctx, orc, make_case(i), set_env)
        for name in touched:
            os.environ.pop(name, None)
        run_f1_case(ctx, orc, make_f1_case(i), set_env)
        if i % 10 == 0:
            for name in touched:
                os.environ.pop(name, None)
            ctx.reload_env()
            run_grid_case(ctx, orc, i)
        if (i - first) % 250 == 249:
            print("%d cases of each kind passed" % (i - first + 1), flush=True)
    print("all %d + %d cases passed" % (n, n))
