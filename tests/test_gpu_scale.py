"""GPU parity at the sizes of BASELINE.json's configurations (the seeded
cases of test_gpu_parity.py stop at a few thousand cells; here every kernel family meets
the oracle on >= 1 M cells, the 4000x4000 field whole and in 8 row slabs, one 137-level
member through the derived + stencil pipeline, and the N-rank drivers).

Everything goes through the C ABI; the oracle (tests/cpulib.py) is the checker."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import cases
import gpu_util

pytestmark = pytest.mark.gpu

ALL, NONE, SOME = cases.ALL_DEFINED, cases.NONE_DEFINED, cases.SOME_DEFINED
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bits_equal(a, b):
    return cases.same_bits(np.asarray(a), np.asarray(b), nan_payload=False)


# ------------------------------------------------------------------ config 4: 4000 x 4000
@pytest.fixture(scope="module")
def field4000(oracle):
    import mi_fieldcalc_amd.synth as synth

    nx = ny = 4000
    xm, ym, _ = synth.grid_maps(nx, ny, h=2500.0)
    u, v = synth.wind(nx, ny, 0x5EED0000 + 4000)
    data = {"nx": nx, "ny": ny, "xm": xm, "ym": ym}
    for mode in ("all", "some"):
        if mode == "some":
            uu, vv = synth.sprinkle_undef(u, 41, 0.001), synth.sprinkle_undef(v, 43, 0.001)
            # undefined values in the first / last columns and around slab boundaries: the wrapped
            # neighbours of the flat loop and the halo rows must count them exactly once
            uu[499:502, :3] = cases.UNDEF
            vv[1000, -2:] = cases.UNDEF
            uu[2499:2501, 1700:1710] = np.nan
            flag = SOME
        else:
            uu, vv, flag = u, v, ALL
        ok, rv_e, f1 = oracle.call("relvort", nx, ny, uu, vv, xm, ym, fdefined=flag)
        ok2, dv_e, f2 = oracle.call("divergence", nx, ny, uu, vv, xm, ym, fdefined=flag)
        assert ok and ok2 and f1 == f2
        data[mode] = dict(u=uu, v=vv, flag=flag, rv=rv_e, dv=dv_e, flag_out=f1)
    return data


@pytest.mark.parametrize("mode", ["all", "some"])
@pytest.mark.parametrize("tune", [None, "R=8", "K=1"])
def test_config4_whole_field_4000x4000(gpu_ctx, field4000, mode, tune, mifc_env):
    """One 4000 x 4000 level (8 wave-columns of 512, the last one partial) through the batched entry:
    the form the launcher picks for one level (one-shot tiles), the row-walking form and the plain one-shot form."""
    import torch

    if tune:
        mifc_env("MIFC_VORTDIV_TUNE", tune)
    d, m = field4000, field4000[mode]
    du, dv, dxm, dym = (torch.from_numpy(a).cuda() for a in (m["u"], m["v"], d["xm"], d["ym"]))
    (rv, dg), fo = gpu_ctx.vortdiv_levels(du[None], dv[None], dxm, dym, fdefined=[m["flag"]])
    assert _bits_equal(rv[0].cpu().numpy(), m["rv"]) and _bits_equal(dg[0].cpu().numpy(), m["dv"])
    assert fo[0] == m["flag_out"]


@pytest.mark.parametrize("nlev", [1, 2, 5])
def test_big_tested_levels_count_by_partials(gpu_ctx, oracle, mifc_env, nlev):
    """Tested levels of 4096 x 2100: thousands of workgroups per level leave their undefined counts in a partials buffer
    ([level][unit]) that a small launch adds up (StencilParams::partials) instead of one atomic each on the level's counter --
    the one-shot kernels (one or two levels) and the split-role level-walking kernels (five).  Counts and values against the
    row-walking forms (one atomic per wave), for the wind operators and the one-input operators, through the asynchronous
    entry and the synchronous one; one level also against the oracle and through a one-slab plan (its own buffer, graph)."""
    import torch

    import mi_fieldcalc_amd as fc
    import mi_fieldcalc_amd.synth as synth

    nx, ny = 4096, 2100
    xm, ym, fcor = synth.grid_maps(nx, ny, h=2500.0)
    u, v = synth.wind(nx, ny, 991)
    z = synth.scalar_field(nx, ny, 992)
    u, z = synth.sprinkle_undef(u, 5, 0.01), synth.sprinkle_undef(z, 6, 0.01)
    u[700:703, :2] = cases.UNDEF
    z[1500, -3:] = cases.UNDEF
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    du, dv, dz = (dev(a[None]).repeat(nlev, 1, 1).contiguous() for a in (u, v, z))
    for l in range(1, nlev):  # other undefined cells on the other levels: the counts differ per level
        du[l, 5 * l:9 * l, 100:200] = float(cases.UNDEF)
        dz[l, 40:44, 7 * l:300 * l] = float(cases.UNDEF)
    dxm, dym, dfc = dev(xm), dev(ym), dev(fcor)
    flags = np.full(nlev, SOME, np.int32)
    wind_form, scalar_form = ("wind_oneshot_tiles", "scalar_oneshot") if nlev <= 2 else ("wind_split", "scalar_split")
    table = [("vortdiv", du, dv, False, wind_form), ("relvort", du, dv, False, wind_form), ("absvort", du, dv, True, wind_form),
             ("gradient1", dz, None, False, scalar_form), ("gradient3", dz, None, False, scalar_form),
             ("plevelgwind_ycomp", dz, None, True, scalar_form), ("plevelgvort", dz, None, True, scalar_form),
             ("ilevelgwind", dz, None, True, scalar_form)]
    if nlev > 2:
        table += [("jacobian", dz, du, False, "wind_split")]
    for name, f0, f1, use_fc, form in table:
        two = name in ("vortdiv", "ilevelgwind")
        got = {}
        for how in ("partials", "rows"):
            if how == "rows":
                mifc_env("MIFC_VORTDIV_TUNE", "R=8")
                mifc_env("MIFC_SCALAR_ROWS_R", "8")
            else:
                mifc_env("MIFC_VORTDIV_TUNE", None)
                mifc_env("MIFC_SCALAR_ROWS_R", None)
            o0, o1 = torch.empty_like(du), (torch.empty_like(du) if two else None)
            cnt = torch.full((nlev,), 777, dtype=torch.int64, device="cuda")
            assert gpu_ctx.stencil_levels_enqueue(name, f0, f1, dxm, dym, dfc if use_fc else None, o0, o1, fdefined=flags, n_undefined=cnt), name
            if how == "partials":
                gpu_util.check_form(gpu_ctx, form, what=name)
            else:
                gpu_util.check_form(gpu_ctx, differs_from=form, what=name)
            torch.cuda.synchronize()
            got[how] = (o0, o1, cnt.cpu().numpy().copy())
        mifc_env("MIFC_VORTDIV_TUNE", None)
        mifc_env("MIFC_SCALAR_ROWS_R", None)
        a, b = got["partials"], got["rows"]
        assert np.array_equal(a[2], b[2]) and a[2].min() > 1000 and (nlev == 1 or len(set(a[2].tolist())) > 1), (name, a[2], b[2])
        assert torch.equal(a[0].view(torch.int32), b[0].view(torch.int32)) and (a[1] is None or torch.equal(a[1].view(torch.int32), b[1].view(torch.int32))), name
        # the synchronous entry: flags from the same counts
        (s0, s1), fo = gpu_ctx.stencil_levels(name, f0, f1, dxm, dym, dfc if use_fc else None, fdefined=flags)
        dom = gpu_ctx.stencil_count_domain(name, nx, ny)
        assert torch.equal(s0.view(torch.int32), a[0].view(torch.int32)) and [int(x) for x in fo] == [fc.classify(int(c), dom) for c in a[2]], name
        del got, a, b, s0, s1, o0, o1
    o0, o1 = torch.empty_like(du), torch.empty_like(du)
    cnt = torch.zeros(nlev, dtype=torch.int64, device="cuda")
    assert gpu_ctx.stencil_levels_enqueue("vortdiv", du, dv, dxm, dym, None, o0, o1, fdefined=flags, n_undefined=cnt)
    torch.cuda.synchronize()
    if nlev == 1:
        ok, e, f = oracle.call("relvort", nx, ny, u, v, xm, ym, fdefined=SOME)
        ok2, e2, _ = oracle.call("divergence", nx, ny, u, v, xm, ym, fdefined=SOME)
        assert ok and ok2 and _bits_equal(o0[0].cpu().numpy(), e) and _bits_equal(o1[0].cpu().numpy(), e2)
        assert fc.classify(int(cnt.item()), nx * ny - 2 * nx) == f
    whole = cnt.cpu().numpy().copy()
    # the whole field as ONE slab of a plan: the plan's own partials buffer, inside its graph
    uh, vh = torch.zeros((nlev, ny + 2, nx), device="cuda"), torch.zeros((nlev, ny + 2, nx), device="cuda")
    uh[:, 1:-1], vh[:, 1:-1] = du, dv
    rv, dg = torch.empty_like(du), torch.empty_like(du)
    pc = torch.zeros(nlev, dtype=torch.int64, device="cuda")
    plan = gpu_ctx.slab_plan(nx, ny, 0, ny, uh, vh, dxm, dym, rv, dg, fdefined_in=SOME, n_undefined=pc)
    for _ in range(2):  # the second step replays the graph
        plan.step()
        torch.cuda.synchronize()
        assert np.array_equal(pc.cpu().numpy(), whole)
        assert torch.equal(rv.view(torch.int32), o0.view(torch.int32)) and torch.equal(dg.view(torch.int32), o1.view(torch.int32))
    plan.close()


@pytest.mark.parametrize("mode", ["all", "some"])
@pytest.mark.parametrize("overlap", [False, True])
def test_config4_eight_row_slabs_4000x4000(gpu_ctx, field4000, mode, overlap):
    """BASELINE.json config 4 on one GPU: 8 slabs of 500 rows, halo rows filled by a loop-back 'exchange';
    overlap=True runs each slab as interior rows + two boundary strips (mifc_vortdiv_slab_rows_enqueue),
    the launch sequence the N-rank driver uses around the RCCL exchange."""
    import torch

    import mi_fieldcalc_amd as fc
    from mi_fieldcalc_amd.sharding import slab_rows

    d, m = field4000, field4000[mode]
    nx, ny, nslab = d["nx"], d["ny"], 8
    u, v = m["u"], m["v"]
    rv = np.empty((ny, nx), np.float32)
    dv = np.empty((ny, nx), np.float32)
    total = 0
    gpu_ctx.use_torch_stream()
    try:
        for r in range(nslab):
            j0, nloc = slab_rows(ny, nslab, r)
            uh = np.zeros((nloc + 2, nx), np.float32)
            vh = np.zeros((nloc + 2, nx), np.float32)
            uh[1:-1], vh[1:-1] = u[j0:j0 + nloc], v[j0:j0 + nloc]
            if j0 > 0:
                uh[0], vh[0] = u[j0 - 1], v[j0 - 1]
            if j0 + nloc < ny:
                uh[-1], vh[-1] = u[j0 + nloc], v[j0 + nloc]
            t = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (uh, vh, d["xm"][j0:j0 + nloc], d["ym"][j0:j0 + nloc])]
            o_rv = torch.full((nloc, nx), -7777.0, dtype=torch.float32, device="cuda")
            o_dv = torch.full_like(o_rv, -7777.0)
            cnt = torch.full((1,), 12345, dtype=torch.int64, device="cuda")
            kw = dict(fdefined_in=m["flag"], n_undefined=cnt)
            if overlap:
                assert gpu_ctx.vortdiv_slab_enqueue(nx, ny, j0, nloc, *t, o_rv, o_dv, rows=(2, nloc - 2), **kw)
                assert gpu_ctx.vortdiv_slab_enqueue(nx, ny, j0, nloc, *t, o_rv, o_dv, rows=(0, 2), accumulate=True, **kw)
                assert gpu_ctx.vortdiv_slab_enqueue(nx, ny, j0, nloc, *t, o_rv, o_dv, rows=(nloc - 2, nloc), accumulate=True, **kw)
            else:
                assert gpu_ctx.vortdiv_slab_enqueue(nx, ny, j0, nloc, *t, o_rv, o_dv, **kw)
            torch.cuda.synchronize()
            rv[j0:j0 + nloc], dv[j0:j0 + nloc] = o_rv.cpu().numpy(), o_dv.cpu().numpy()
            total += int(cnt.item())
    finally:
        gpu_ctx.set_stream(None)
    assert _bits_equal(rv, m["rv"]) and _bits_equal(dv, m["dv"])
    got_flag = ALL if m["flag"] == ALL else fc.classify(total, nx * ny - 2 * nx)
    assert got_flag == m["flag_out"]


def test_slab_row_ranges_are_validated(gpu_ctx):
    """A row range may not separate a global edge row from the row it is filled from."""
    import torch

    nx, ny = 64, 40
    z = lambda r: torch.zeros((r, nx), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    gpu_ctx.use_torch_stream()
    try:
        for j0, nloc, rows in ((0, 20, (1, 20)), (0, 20, (0, 1)), (20, 20, (0, 19)), (20, 20, (19, 20))):
            a = (z(nloc + 2), z(nloc + 2), z(nloc), z(nloc), z(nloc), z(nloc))
            with pytest.raises(RuntimeError):
                gpu_ctx.vortdiv_slab_enqueue(nx, ny, j0, nloc, *a, n_undefined=cnt, rows=rows)
        a = (z(22), z(22), z(20), z(20), z(20), z(20))
        assert gpu_ctx.vortdiv_slab_enqueue(nx, ny, 0, 20, *a, n_undefined=cnt, rows=(0, 2))
        assert not gpu_ctx.vortdiv_slab_enqueue(nx, ny, 0, 20, *a, n_undefined=cnt, rows=(5, 5))  # empty range: false, no error
        torch.cuda.synchronize()
    finally:
        gpu_ctx.set_stream(None)


def test_halo_copy_between_contexts(gpu_ctx):
    """mifc_halo_copy_enqueue: the transport of a process that drives several GPUs itself, exercised with two
    contexts on this box's one device (the copy is ordered after the producer queued on the source context)."""
    import torch

    import mi_fieldcalc_amd as fc

    other = fc.Context(0)
    try:
        src = torch.arange(4000, dtype=torch.float32, device="cuda")
        dst = torch.zeros(4000, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        assert gpu_ctx.halo_copy_enqueue(dst, other, src)
        gpu_ctx.synchronize()
        assert torch.equal(dst, src)
    finally:
        other.close()


# ------------------------------------------------------------------ level-padded batches
@pytest.mark.parametrize("pad", [0, 4, 1440 * 5])
def test_vortdiv_levels_with_padded_level_stride(gpu_ctx, oracle, pad):
    """mifc_vortdiv_levels_strided_enqueue: levels 'pad' floats apart; the padding is never written."""
    import torch

    import mi_fieldcalc_amd as fc
    import mi_fieldcalc_amd.synth as synth

    nx, ny, nlev = 1440, 40, 11
    xm, ym, _ = synth.grid_maps(nx, ny)
    u, v = synth.wind(nx, ny, 77, nlev=nlev)
    u[3] = synth.sprinkle_undef(u[3], 9, 0.01)
    flags = np.full(nlev, ALL, np.int32)
    flags[3] = SOME
    ls = nx * ny + pad
    bu, bv, brv, bdg = (gpu_ctx.batch_empty(nlev, ny, nx, level_stride=ls) for _ in range(4))
    for b in (brv, bdg):  # recognisable padding
        torch.as_strided(b, (nlev * ls,), (1,)).fill_(-7777.0)
    bu.copy_(torch.from_numpy(u))
    bv.copy_(torch.from_numpy(v))
    cnt = torch.zeros(nlev, dtype=torch.int64, device="cuda")
    gpu_ctx.use_torch_stream()
    try:
        assert gpu_ctx.vortdiv_levels_enqueue(bu, bv, torch.from_numpy(xm).cuda(), torch.from_numpy(ym).cuda(), brv, bdg, fdefined=flags, n_undefined=cnt)
        torch.cuda.synchronize()
    finally:
        gpu_ctx.set_stream(None)
    for l in range(nlev):
        ok, e, f = oracle.call("relvort", nx, ny, u[l], v[l], xm, ym, fdefined=int(flags[l]))
        assert _bits_equal(brv[l].cpu().numpy(), e)
        ok, e, f2 = oracle.call("divergence", nx, ny, u[l], v[l], xm, ym, fdefined=int(flags[l]))
        assert _bits_equal(bdg[l].cpu().numpy(), e)
        assert (ALL if flags[l] == ALL else fc.classify(int(cnt[l].item()), nx * ny - 2 * nx)) == f == f2
    if pad:
        flat = torch.as_strided(brv, (nlev, ls), (ls, 1))
        assert bool((flat[:, nx * ny:] == -7777.0).all())
    assert gpu_ctx.batch_level_stride(nx, ny) % 4 == 0 and gpu_ctx.batch_level_stride(nx, ny) >= nx * ny


# ------------------------------------------------------------------ config 2: fused ff / RH / theta at 1440 x 720
@pytest.mark.parametrize("nlev", [1, 9])
@pytest.mark.parametrize("mode", ["all", "some"])
def test_config2_derived_batch_1440x720(gpu_ctx, oracle, nlev, mode):
    """BASELINE.json config 2 at its own size: one level (per-level scalars in the kernel arguments) and
    nine (device tables), against the oracle's per-level vectorabs / hlevelhum / hleveltemp."""
    import torch

    import mi_fieldcalc_amd.synth as synth

    nx, ny = 1440, 720
    u, v = synth.wind(nx, ny, 211, nlev=nlev)
    t, q, ps = synth.thermo(nx, ny, 212, nlev=nlev)
    a, b = synth.hybrid_levels(max(nlev, 3))
    a, b = a[:nlev], b[:nlev]
    fw = np.full(nlev, ALL if mode == "all" else SOME, np.int32)
    ft = fw.copy()
    if mode == "some":
        for l in range(nlev):
            u[l] = synth.sprinkle_undef(u[l], 50 + l, 0.01)
            t[l] = synth.sprinkle_undef(t[l], 60 + l, 0.01)
            q[l] = synth.sprinkle_undef(q[l], 70 + l, 0.01)
        ps = synth.sprinkle_undef(ps, 80, 0.01, nan_every=0)
        t[0, 5, :64] = 400.0  # outside the ewt table: rh undefined, theta defined
    dev = [torch.from_numpy(x).cuda() for x in (u, v, t, q, ps)]
    for args in (dev, (u, v, t, q, ps)) if nlev == 1 else (dev,):
        res, flags = gpu_ctx.hlevel_derived_levels(*args, a, b, fdef_wind=fw, fdef_thermo=ft)
        get = (lambda x: x.cpu().numpy()) if args is dev else (lambda x: x)
        for l in range(nlev):
            ok, ff_e, f_ff = oracle.call("vectorabs", nx, ny, u[l], v[l], fdefined=int(fw[l]))
            ok, rh_e, f_rh = oracle.call("hlevelhum", nx, ny, t[l], q[l], ps, float(a[l]), float(b[l]), "", 1, fdefined=int(ft[l]))
            ok, th_e, f_th = oracle.call("hleveltemp", nx, ny, t[l], ps, float(a[l]), float(b[l]), "", 3, fdefined=int(ft[l]))
            case = dict(label="derived-1440x720-l%d" % l, undef=cases.UNDEF, op="derived")
            gpu_util.compare(case, get(res["ff"])[l], ff_e, True)
            gpu_util.compare(case, get(res["rh"])[l], rh_e, True)
            gpu_util.compare(case, get(res["theta"])[l], th_e, False)
            assert (flags["ff"][l], flags["rh"][l], flags["theta"][l]) == (f_ff, f_rh, f_th)


def _check_derived_batch(gpu_ctx, oracle, u, v, t, h, ps, a, b, fw, ft, temp, hum, hum2, ff, device):
    """One mifc_hlevel_derived_batch call against the per-level reference calls it stands for."""
    import torch

    nlev, ny, nx = t.shape
    args = [u, v, t, h, ps]
    if device:
        args = [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in args]
    tag = "temp=%s hum=%s hum2=%s" % (temp, hum, hum2)
    call = lambda: gpu_ctx.hlevel_derived_batch(*args, a, b, temp=temp, hum=hum, hum2=hum2, ff=ff, fdef_wind=fw, fdef_thermo=ft)
    if temp is not None:
        c = temp[1]
        if c < 3:
            c = 1 if temp[0] == "celsius" else (2 if temp[0] == "kelvin" else c)
        if not 1 <= c <= 5:  # the reference would leave cells unwritten (:1080-1090): the batched entry refuses, with a message
            with pytest.raises(RuntimeError):
                call()
            return
    if any(w is not None and not 1 <= w[1] <= 12 for w in (hum, hum2)):  # hlevelhum returns false (:1168)
        assert call() is None, tag
        return
    res = call()
    assert res is not None, tag
    out, flags = res
    get = (lambda x: x.cpu().numpy()) if device else (lambda x: x)
    for l in range(nlev):
        if ff:
            ok, e, f = oracle.call("vectorabs", nx, ny, u[l], v[l], fdefined=int(fw[l]))
            gpu_util.compare(dict(label="ff " + tag, undef=cases.UNDEF, op="vectorabs"), get(out["ff"])[l], e, True)
            assert flags["ff"][l] == f, tag
        if temp is not None:
            cargs = [t[l], ps, float(a[l]), float(b[l]), temp[0], temp[1]]
            ok, e, f = oracle.call("hleveltemp", nx, ny, *cargs, fdefined=int(ft[l]))
            assert ok
            case = dict(label="temp l%d %s" % (l, tag), undef=cases.UNDEF, op="hleveltemp", args=cargs)
            gpu_util.compare(case, get(out["temp"])[l], e, False)
            assert flags["temp"][l] == f, tag
        for name, w in (("hum", hum), ("hum2", hum2)):
            if w is None:
                continue
            cargs = [t[l], h[l], ps, float(a[l]), float(b[l]), w[0], w[1]]
            ok, e, f = oracle.call("hlevelhum", nx, ny, *cargs, fdefined=int(ft[l]))
            assert ok
            case = dict(label="%s l%d %s" % (name, l, tag), undef=cases.UNDEF, op="hlevelhum", args=cargs)
            gpu_util.compare(case, get(out[name])[l], e, not gpu_util.uses_device_powf(case))
            assert flags[name][l] == f, (tag, name, l)


@pytest.mark.parametrize("nlev", [3, 9])
def test_derived_batch_every_variant(gpu_ctx, oracle, nlev):
    """Every hleveltemp / hlevelhum variant the batched entry offers (compile-time combinations and the
    generic instantiation; per-level scalars in the kernel arguments for nlev <= 8, device tables above),
    each against the per-level reference call, flags included; invalid computes return None."""
    import mi_fieldcalc_amd.synth as synth

    nx, ny = 64, 36
    u, v = synth.wind(nx, ny, 311, nlev=nlev)
    t, q, ps = synth.thermo(nx, ny, 312, nlev=nlev)
    a, b = synth.hybrid_levels(max(nlev, 3))
    a, b = a[:nlev], b[:nlev]
    rh = synth.uniform((nlev, ny, nx), 313, 0.5, 110.0).astype(np.float32)
    fw = np.full(nlev, SOME, np.int32)
    ft = fw.copy()
    fw[0] = ft[0] = ALL  # level 0: the no-test path (its fields stay clean)
    for l in range(1, nlev):
        u[l] = synth.sprinkle_undef(u[l], 50 + l, 0.03)
        t[l] = synth.sprinkle_undef(t[l], 60 + l, 0.03)
        q[l] = synth.sprinkle_undef(q[l], 70 + l, 0.03)
        rh[l] = synth.sprinkle_undef(rh[l], 75 + l, 0.03)
    ps = synth.sprinkle_undef(ps, 80, 0.02, nan_every=3)
    ps[0, :4] = 1000.0
    t[0, 0, :8] = 400.0  # outside the saturation table
    k = 0
    for c in range(0, 7):
        for unit in ("celsius", "kelvin", ""):
            _check_derived_batch(gpu_ctx, oracle, u, v, t, q, ps, a, b, fw, ft, (unit, c), None, None, ff=bool(k % 2), device=bool(k % 3 == 0))
            k += 1
    for c in range(0, 14):
        for unit in ("celsius", "kelvin"):
            h = q if c in (1, 2, 5, 6, 9, 10) else rh
            _check_derived_batch(gpu_ctx, oracle, u, v, t, h, ps, a, b, fw, ft, None, (unit, c), None, ff=False, device=bool(k % 2))
            k += 1
    # the compile-time combinations and mixed ones
    for temp, hum, hum2, ff in ((("", 3), ("", 1), None, True), (("", 3), ("", 1), ("", 9), True), (("", 3), ("", 1), None, False),
                                (("", 3), ("", 1), ("", 9), False), (None, None, None, True), (("kelvin", 1), ("", 9), ("celsius", 9), True),
                                (("", 4), ("", 5), ("", 1), False), (("", 5), ("", 2), ("", 10), True), (None, ("", 1), ("", 7), True)):
        hh = q
        _check_derived_batch(gpu_ctx, oracle, u, v, t, hh, ps, a, b, fw, ft, temp, hum, hum2, ff, device=True)
    # a bad hybrid level -> false, like the reference (:298)
    assert gpu_ctx.hlevel_derived_batch(u, v, t, q, ps, -a, b, temp=("", 3), fdef_wind=fw, fdef_thermo=ft) is None
    # the wind direction (extension) as a fifth output: what mifc_winddir gives per level, flags included
    res, flags = gpu_ctx.hlevel_derived_batch(u, v, t, q, ps, a, b, temp=("", 3), hum=("", 1), hum2=("", 9), dd=True, fdef_wind=fw, fdef_thermo=ft)
    only_dd, flags2 = gpu_ctx.hlevel_derived_batch(u, v, None, None, None, None, None, ff=False, dd=True, fdef_wind=fw)
    for l in range(nlev):
        e, f = gpu_ctx.winddir(u[l], v[l], fdefined=int(fw[l]))
        assert _bits_equal(res["dd"][l], e) and _bits_equal(only_dd["dd"][l], e) and flags["dd"][l] == f == flags2["dd"][l]
        ok, e, f = oracle.call("vectorabs", nx, ny, u[l], v[l], fdefined=int(fw[l]))
        assert _bits_equal(res["ff"][l], e) and flags["ff"][l] == f


@pytest.mark.parametrize("mode", ["all", "some"])
def test_config2_quartet_with_dew_point_1440x720(gpu_ctx, oracle, mode):
    """ff + theta + RH + dew point (K) in one launch at BASELINE.json config 2's size, 9 levels."""
    import mi_fieldcalc_amd.synth as synth

    nx, ny, nlev = 1440, 720, 9
    u, v = synth.wind(nx, ny, 411, nlev=nlev)
    t, q, ps = synth.thermo(nx, ny, 412, nlev=nlev)
    a, b = synth.hybrid_levels(nlev)
    fw = np.full(nlev, ALL if mode == "all" else SOME, np.int32)
    ft = fw.copy()
    if mode == "some":
        for l in range(nlev):
            u[l] = synth.sprinkle_undef(u[l], 50 + l, 0.01)
            t[l] = synth.sprinkle_undef(t[l], 60 + l, 0.01)
            q[l] = synth.sprinkle_undef(q[l], 70 + l, 0.01)
        ps = synth.sprinkle_undef(ps, 80, 0.01, nan_every=0)
    _check_derived_batch(gpu_ctx, oracle, u, v, t, q, ps, a, b, fw, ft, ("", 3), ("", 1), ("", 9), True, device=True)


# ------------------------------------------------------------------ config 5: one member through the pipeline
def test_config5_one_member_pipeline(gpu_ctx, oracle):
    """One ensemble member of BASELINE.json config 5 (1440 x 720 x 137, device resident) through the
    derived batch and the stencil batch as the N-rank driver runs them; sampled levels against the
    oracle (one of them with undefined values), every level through the flags."""
    import torch

    import mi_fieldcalc_amd as fc
    import mi_fieldcalc_amd.synth as synth

    nx, ny, nlev = 1440, 720, 137
    xm, ym, _ = synth.grid_maps(nx, ny)
    dxm, dym = torch.from_numpy(xm).cuda(), torch.from_numpy(ym).cuda()
    du, dv = synth.device_wind(nx, ny, nlev, 0x5EED0000 + 5000, "cuda")
    dt, dq, dps = synth.device_thermo(nx, ny, nlev, 0x5EED0000 + 5500, "cuda")
    a, b = synth.hybrid_levels(nlev)
    fw = np.full(nlev, ALL, np.int32)
    ft = np.full(nlev, ALL, np.int32)
    bad = 77  # one level carries undefined values
    fw[bad] = ft[bad] = SOME
    du[bad, 100:104, 10:700] = float(cases.UNDEF)
    dt[bad, 300:302, :] = float("nan")
    dq[bad, 600, 5:9] = float(cases.UNDEF)
    ff, rh, th, rv, dg = (torch.empty_like(du) for _ in range(5))
    cnt_d = torch.zeros(3 * nlev, dtype=torch.int64, device="cuda")
    cnt_s = torch.zeros(nlev, dtype=torch.int64, device="cuda")
    gpu_ctx.use_torch_stream()
    try:
        assert gpu_ctx.hlevel_derived_levels_enqueue(du, dv, dt, dq, dps, a, b, ff, rh, th, cnt_d, fdef_wind=fw, fdef_thermo=ft)
        assert gpu_ctx.vortdiv_levels_enqueue(du, dv, dxm, dym, rv, dg, fdefined=fw, n_undefined=cnt_s)
        torch.cuda.synchronize()
    finally:
        gpu_ctx.set_stream(None)
    n = nx * ny
    cd, cs = cnt_d.cpu().numpy().reshape(3, nlev), cnt_s.cpu().numpy()
    ps_h = dps.cpu().numpy()
    for l in (0, 1, 68, bad, 136):
        ul, vl, tl, ql = (x[l].cpu().numpy() for x in (du, dv, dt, dq))
        ok, e, f = oracle.call("relvort", nx, ny, ul, vl, xm, ym, fdefined=int(fw[l]))
        assert _bits_equal(rv[l].cpu().numpy(), e) and (ALL if fw[l] == ALL else fc.classify(int(cs[l]), n - 2 * nx)) == f
        ok, e, f = oracle.call("divergence", nx, ny, ul, vl, xm, ym, fdefined=int(fw[l]))
        assert _bits_equal(dg[l].cpu().numpy(), e)
        case = dict(label="member-l%d" % l, undef=cases.UNDEF, op="derived")
        ok, e, f = oracle.call("vectorabs", nx, ny, ul, vl, fdefined=int(fw[l]))
        gpu_util.compare(case, ff[l].cpu().numpy(), e, True)
        assert fc.classify(int(cd[0, l]), n) == f
        ok, e, f = oracle.call("hlevelhum", nx, ny, tl, ql, ps_h, float(a[l]), float(b[l]), "", 1, fdefined=int(ft[l]))
        gpu_util.compare(case, rh[l].cpu().numpy(), e, True)
        assert fc.classify(int(cd[1, l]), n) == f
        ok, e, f = oracle.call("hleveltemp", nx, ny, tl, ps_h, float(a[l]), float(b[l]), "", 3, fdefined=int(ft[l]))
        gpu_util.compare(case, th[l].cpu().numpy(), e, False)
        assert fc.classify(int(cd[2, l]), n) == f
    # every other level is clean
    clean = np.arange(nlev) != bad
    assert np.all(cd[:, clean] == 0) and np.all(cs[clean] == 0) and cs[bad] > 0 and np.all(cd[:, bad] > 0)
    # fillEdges on every level of the stencil outputs
    assert torch.equal(rv[:, 0, :], rv[:, 1, :]) and torch.equal(dg[:, :, -1], dg[:, :, -2])


# ------------------------------------------------------------------ every elementwise / pointwise family at >= 1 M cells
BIG = (1440, 720)  # 1 036 800 cells: 1013 workgroups of 1024 cells -- the grid-stride kernels take several trips


def _pick(all_cases, wanted):
    """wanted: {op: set of compute values or None}; compute = the last int in args (or the first for fieldOPER*)."""
    out = []
    for c in all_cases:
        sel = wanted.get(c["op"], False)
        if sel is False:
            continue
        if sel is None:
            out.append(c)
            continue
        ints = [a for a in c["args"] if isinstance(a, (int, np.integer)) and not isinstance(a, bool)]
        comp = ints[0] if c["op"].startswith(("fieldOPER", "constantOPER")) else (ints[-1] if ints else None)
        if comp in sel:
            out.append(c)
    return out


def _run_big(gpu_ctx, oracle, cs, device):
    from test_gpu_parity import _check_case

    assert cs
    for case in cs:
        _check_case(gpu_ctx, oracle, case, device=device)


@pytest.mark.parametrize("mode", ["all", "some"])
def test_elementwise_families_at_a_million_cells(gpu_ctx, oracle, mode):
    """vectorabs, theta (scalar / hybrid / field pressure), the humidity variants incl. both dew-point
    paths and humidity from potential temperature, cvhum both directions -- 1440 x 720 each."""
    wanted = {"vectorabs": None, "pleveltemp": {1, 3, 4}, "hleveltemp": {1, 3, 5}, "aleveltemp": {2, 3, 4},
              "plevelhum": {1, 4, 5, 8}, "hlevelhum": {1, 2, 3, 6, 9, 12}, "alevelhum": {1, 2, 5, 8, 11}, "cvhum": {1, 3, 4, 5}}
    cs = [c for c in _pick(cases.ewise_cases(grids=[BIG], modes=(mode,)), wanted) if c["op"] == "vectorabs" or c["args"][-2] in ("kelvin", "celsius", "1")]
    # one unit string per (op, compute) is enough at this size
    seen, keep = set(), []
    for c in cs:
        key = (c["op"], c["args"][-1] if c["op"] != "vectorabs" else 0)
        if key not in seen:
            seen.add(key)
            keep.append(c)
    _run_big(gpu_ctx, oracle, keep, device=True)


@pytest.mark.parametrize("mode", ["all", "some"])
def test_pointwise_catalogue_families_at_a_million_cells(gpu_ctx, oracle, mode):
    """theta-e, ducting, the indices, conversions, libm-class functions and field algebra -- 1440 x 720 each."""
    wanted = {"plevelthe": {1}, "hlevelthe": {1, 2}, "alevelthe": {1}, "kIndex": {1}, "ductingIndex": {1}, "showalterIndex": {1}, "boydenIndex": {1},
              "sweatIndex": None, "seaSoundSpeed": {1}, "windCooling": {1, 2}, "plevelducting": {1}, "hlevelducting": {2, 3}, "alevelducting": {1},
              "cvtemp": {1, 3}, "fieldOPERfield": {1, 4}, "fieldOPERconstant": {3, 4}, "constantOPERfield": {4}, "hlevelpressure": None,
              "pleveldz2tmean": {1}, "abshum": None, "underCooledRain": None, "pressure2FlightLevel": None, "snow_in_cm": None,
              "vesselIcingOverland": None, "vesselIcingMertins": None, "minvalueFields": None, "maxvalueFieldConst": None, "absvalueField": None,
              "log10Field": None, "logField": None, "pow10Field": None, "expField": None, "powerField": None, "replaceUndefined": None}
    cs = _pick(cases.catalogue_cases(grids=[BIG], modes=(mode,)), wanted)
    seen, keep = set(), []
    for c in cs:  # one case per operator variant
        ints = tuple(a for a in c["args"] if isinstance(a, (int, np.integer)))
        key = (c["op"], ints[:1])
        if key not in seen and "values2classes" not in c["label"]:
            seen.add(key)
            keep.append(c)
    _run_big(gpu_ctx, oracle, keep, device=True)


def test_single_field_stencils_at_a_million_cells(gpu_ctx, oracle):
    """The one-input stencil operators and the f1 family on one 1440 x 720 level (the seeded cases stop at 516 x 37)."""
    ops = {"gradient", "plevelgwind_xcomp", "plevelgwind_ycomp", "plevelgvort", "ilevelgwind", "absvort", "advection", "jacobian",
           "thermalFrontParameter", "shapiro2_filter", "momentumXcoordinate"}
    cs = [c for c in cases.stencil_cases(grids=[BIG], modes=("some",)) if c["op"] in ops]
    cs += [c for c in cases.stencil_cases(grids=[BIG], modes=("some",)) if c["op"] == "plevelqvector" and c["args"][-1] in (1, 4)]
    _run_big(gpu_ctx, oracle, cs, device=True)


# ------------------------------------------------------------------ ADVICE round 1
def test_stream_switch_is_ordered_against_enqueued_work(gpu_ctx, oracle):
    """An *_enqueue call leaves a kernel on stream A that reads the context's per-level flag scratch; a call on
    stream B right behind it re-uploads that scratch.  The switch must make B wait (mifc_capi.hip:440)."""
    import torch

    import mi_fieldcalc_amd as fc
    import mi_fieldcalc_amd.synth as synth

    nx, ny, nlev = 1440, 720, 24
    xm, ym, _ = synth.grid_maps(nx, ny)
    dxm, dym = torch.from_numpy(xm).cuda(), torch.from_numpy(ym).cuda()
    u, v = synth.wind(nx, ny, 5150, nlev=nlev)
    # flags lie on the odd levels: ALL_DEFINED although undefined values are present -> the result depends on the flag
    for l in range(nlev):
        u[l] = synth.sprinkle_undef(u[l], 900 + l, 0.01)
    flags_a = np.array([ALL if l % 2 else SOME for l in range(nlev)], np.int32)
    flags_b = np.array([SOME if l % 2 else ALL for l in range(nlev)], np.int32)
    du, dv = torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda()
    rv_a, dg_a, rv_b, dg_b = (torch.empty_like(du) for _ in range(4))
    cnt_a = torch.zeros(nlev, dtype=torch.int64, device="cuda")
    cnt_b = torch.zeros(nlev, dtype=torch.int64, device="cuda")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    try:
        # something long in front of A so that its kernel is still queued when B's upload is issued
        with torch.cuda.stream(sa):
            for _ in range(20):
                dg_a.copy_(du)
            assert gpu_ctx.vortdiv_levels_enqueue(du, dv, dxm, dym, rv_a, dg_a, fdefined=flags_a, n_undefined=cnt_a)
        with torch.cuda.stream(sb):
            assert gpu_ctx.vortdiv_levels_enqueue(du, dv, dxm, dym, rv_b, dg_b, fdefined=flags_b, n_undefined=cnt_b)
        torch.cuda.synchronize()
    finally:
        gpu_ctx.set_stream(None)
    for flags, rv in ((flags_a, rv_a), (flags_b, rv_b)):
        for l in (0, 1, 10, 23):
            ok, e, _ = oracle.call("relvort", nx, ny, u[l], v[l], xm, ym, fdefined=int(flags[l]))
            assert _bits_equal(rv[l].cpu().numpy(), e), l


def test_wrapper_rejects_mismatched_fields(gpu_ctx):
    """a field of another shape must never reach the kernels (out-of-bounds reads).  Like the
    reference's binding (py_mi_fieldcalc.cc:82-83) the single-field calls return None; the batched ones raise."""
    import torch

    u = np.ones((20, 30), np.float32)
    small = np.ones((19, 30), np.float32)
    assert gpu_ctx.vectorabs(u, small) is None
    assert gpu_ctx.relvort(u, u, small, u) is None
    assert gpu_ctx.vectorabs(u, u, out=np.empty((30, 20), np.float32)) is None
    assert gpu_ctx.meanValue([u, small], [ALL, ALL]) is None
    with pytest.raises(ValueError):
        gpu_ctx.vectorabs(u, u, out=np.empty((20, 30), np.float64))  # would be converted: the caller's array never written
    lev = torch.ones((3, 20, 30), device="cuda")
    maps = torch.ones((20, 30), device="cuda")
    with pytest.raises(ValueError):
        gpu_ctx.vortdiv_levels(lev, lev, maps.t().contiguous(), maps)
    with pytest.raises(ValueError):
        gpu_ctx.vortdiv_levels_enqueue(lev, lev[:2], maps, maps, torch.empty_like(lev), None, fdefined=[ALL] * 3)
    with pytest.raises(ValueError):  # host arrays are not device pointers
        gpu_ctx.vortdiv_levels_enqueue(np.ones((3, 20, 30), np.float32), lev, maps, maps, torch.empty_like(lev), None, fdefined=[ALL] * 3)
    with pytest.raises(ValueError):
        gpu_ctx.stencil_levels("gradient3", lev, None, maps[:10], maps)


# ------------------------------------------------------------------ placed batches through the C ABI
class _PlacementReport(__import__("ctypes").Structure):
    import ctypes as _c

    _fields_ = [("strategy", _c.c_int), ("pool_size", _c.c_int), ("probes", _c.c_int), ("as_allocated_ms", _c.c_float), ("chosen_ms", _c.c_float),
                ("probe_ms_min", _c.c_float), ("probe_ms_median", _c.c_float), ("probe_ms_max", _c.c_float), ("array_distance_bytes", _c.c_size_t)]


def _placed_batch_roundtrip(gpu_ctx, oracle, strategy, probe=None, budget=0):
    """mifc_batch_alloc_placed -> the four arrays of a small level batch; fused vorticity+divergence on them through raw
    device pointers == the oracle; mifc_batch_free_placed."""
    import ctypes

    import mi_fieldcalc_amd.synth as synth

    lib, ctx = gpu_ctx._lib, gpu_ctx._ctx
    nx, ny, nlev = 1440, 60, 6
    n = nx * ny * nlev
    arrays = (ctypes.c_void_p * 4)()
    rep = _PlacementReport()
    cb = ctypes.cast(probe, ctypes.c_void_p) if probe is not None else None
    ok = lib.mifc_batch_alloc_placed(ctx, 4, n * 4, strategy, budget, nx, ny, nlev, cb, None, ctypes.cast(arrays, ctypes.c_void_p),
                                     ctypes.cast(ctypes.pointer(rep), ctypes.c_void_p))
    assert ok, gpu_ctx.last_error()
    assert len({int(a) for a in arrays}) == 4 and all(int(a) % 256 == 0 for a in arrays)
    xm, ym, _ = synth.grid_maps(nx, ny)
    u, v = synth.wind(nx, ny, 4711, nlev=nlev)
    dxm, dym = lib.mifc_device_alloc(ctx, nx * ny * 4), lib.mifc_device_alloc(ctx, nx * ny * 4)
    for dst, src in ((arrays[0], u), (arrays[1], v), (dxm, xm), (dym, ym)):
        assert lib.mifc_copy_to_device(ctx, dst, src.ctypes.data, src.nbytes)
    flags = np.full(nlev, ALL, np.int32)
    assert lib.mifc_vortdiv_levels(ctx, nx, ny, nlev, arrays[0], arrays[1], dxm, dym, arrays[2], arrays[3], flags.ctypes.data, 1e35, 1), gpu_ctx.last_error()
    rv, dg = np.empty_like(u), np.empty_like(u)
    assert lib.mifc_copy_to_host(ctx, rv.ctypes.data, arrays[2], rv.nbytes) and lib.mifc_copy_to_host(ctx, dg.ctypes.data, arrays[3], dg.nbytes)
    for l in (0, nlev - 1):
        ok, e, _ = oracle.call("relvort", nx, ny, u[l], v[l], xm, ym, fdefined=ALL)
        assert ok and _bits_equal(rv[l], e)
        ok, e, _ = oracle.call("divergence", nx, ny, u[l], v[l], xm, ym, fdefined=ALL)
        assert ok and _bits_equal(dg[l], e)
    assert lib.mifc_batch_free_placed(ctx, ctypes.cast(arrays, ctypes.c_void_p), 4) and all(a is None for a in arrays)
    lib.mifc_device_free(ctx, dxm)
    lib.mifc_device_free(ctx, dym)
    return rep


def test_batch_alloc_placed_search_with_the_librarys_probe(gpu_ctx, oracle):
    rep = _placed_batch_roundtrip(gpu_ctx, oracle, 1)
    assert rep.strategy == 1 and rep.pool_size == 48 and 4 < rep.probes <= 160
    assert 0.0 < rep.probe_ms_min <= rep.chosen_ms <= rep.as_allocated_ms <= rep.probe_ms_max and rep.chosen_ms == rep.probe_ms_min
    # a budget of ten arrays: a pool of ten
    rep = _placed_batch_roundtrip(gpu_ctx, oracle, 1, budget=10 * 1440 * 60 * 6 * 4)
    assert rep.pool_size == 10 and rep.probes > 1
    rep = _placed_batch_roundtrip(gpu_ctx, oracle, 0)
    assert rep.strategy == 0 and rep.pool_size == 4 and rep.probes == 0


def test_batch_alloc_placed_search_with_a_callers_probe(gpu_ctx, oracle):
    """The search keeps the set the probe likes best: a synthetic probe that prefers arrays at low addresses ends up with the
    four lowest of the pool in ascending order of cost."""
    import ctypes

    seen = []

    @ctypes.CFUNCTYPE(ctypes.c_float, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p))
    def probe(user, arrays):
        a = [int(arrays[k]) for k in range(4)]
        seen.append(tuple(a))
        return 1.0 + 1e-12 * sum((k + 1) * x for k, x in enumerate(a)) % 1.0

    rep = _placed_batch_roundtrip(gpu_ctx, oracle, 1, probe=probe)
    assert rep.probes == len(set(seen)) and rep.chosen_ms == rep.probe_ms_min


@pytest.mark.skipif(os.environ.get("MIFC_TEST_VMM") != "1", reason="opt-in: MIFC_PLACE_VMM maps memory with HIP's virtual-memory API (MIFC_TEST_VMM=1)")
def test_batch_alloc_placed_virtual_memory_strategy(gpu_ctx, oracle):
    rep = _placed_batch_roundtrip(gpu_ctx, oracle, 2)
    assert rep.strategy == 2 and 96 <= (rep.array_distance_bytes >> 20) % 256 <= 128 and rep.chosen_ms > 0


# ------------------------------------------------------------------ N-rank drivers
def _torchrun(nproc, script_args, env_extra, timeout=600):
    env = dict(os.environ, **env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + (os.getpid() % 300)), os.path.join(ROOT, "tools", "bench_multigpu.py")] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def _last_json(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert lines, stdout
    return json.loads(lines[-1])


def test_config4_driver_two_ranks_on_this_gpu(oracle, tmp_path):
    """tools/bench_multigpu.py --config 4 with two ranks sharing this box's GPU (gloo carries the halo rows through the
    host): slabs + overlapped exchange + count all-reduce == rank 0's whole-field result, and == the oracle."""
    import mi_fieldcalc_amd.synth as synth

    p = _torchrun(2, ["--config", "4", "--size", "1000", "--steps", "3", "--warmup", "1", "--check", "--dump", str(tmp_path)],
                  {"MIFC_BENCH_BACKEND": "gloo"})
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    res = _last_json(p.stdout)
    assert res["verified"] is True and res["n_gpus"] == 2 and res["scaling"] == "strong"
    meta = json.load(open(tmp_path / "config4_meta.json"))
    nx = ny = 1000
    xm, ym, _ = synth.grid_maps(nx, ny, h=2500.0)
    u, v = synth.wind(nx, ny, meta["seed"])
    u = synth.sprinkle_undef(u, 41, 0.001)
    ok, e, f = oracle.call("relvort", nx, ny, u, v, xm, ym, fdefined=SOME)
    assert _bits_equal(np.load(tmp_path / "config4_rvort.npy"), e) and f == meta["flag"]
    ok, e, f = oracle.call("divergence", nx, ny, u, v, xm, ym, fdefined=SOME)
    assert _bits_equal(np.load(tmp_path / "config4_diverg.npy"), e)


def test_config4_driver_two_ranks_three_levels(tmp_path):
    """... and a slab that is a batch of three levels (one exchange of three rows per neighbour and field)."""
    p = _torchrun(2, ["--config", "4", "--size", "1000", "--levels", "3", "--steps", "2", "--warmup", "1", "--check"], {"MIFC_BENCH_BACKEND": "gloo"})
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    res = _last_json(p.stdout)
    assert res["verified"] is True and res["n_gpus"] == 2 and "begin / host relay / finish" in res["config"]["step"]


@pytest.mark.parametrize("graph", ["1", "0"])
def test_config4_driver_one_rank_over_rccl(tmp_path, graph):
    """One rank, backend nccl: mifc_slab_plan_step (HIP-graph replay, or the direct sequence under MIFC_SLAB_GRAPH=0) on the
    whole field as ONE slab, checked against the whole-field call; the roofline block is there."""
    p = _torchrun(1, ["--config", "4", "--size", "1000", "--levels", "2", "--steps", "4", "--warmup", "1", "--check"], {"MIFC_SLAB_GRAPH": graph})
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    res = _last_json(p.stdout)
    assert res["verified"] is True and res["n_gpus"] == 1
    assert res["config"]["step"] == ("mifc_slab_plan_step, HIP graph replay" if graph == "1" else "mifc_slab_plan_step, direct")
    assert 0.0 < res["roofline"]["frac"] <= 1.0 and res["roofline"]["per_rank_frac_min"] <= res["roofline"]["per_rank_frac_max"]


@pytest.mark.parametrize("nx,ny,nslab,nlev", [(1000, 403, 4, 1), (1440, 360, 3, 5), (516, 64, 8, 2)])
@pytest.mark.parametrize("mode", ["all", "some"])
def test_slab_plans_in_loopback_equal_the_whole_field(gpu_ctx, oracle, nx, ny, nslab, nlev, mode):
    """mifc_slab_plan_begin / finish on every slab of a field held on ONE GPU, the halo rows copied between the slabs in
    between (what RCCL does between ranks): the assembled levels equal the oracle's whole-field relvort / divergence bit
    for bit, the summed counts give its flags.  Slabs of unequal height, level batches deep enough for the level-walking
    kernels, thin slabs that are not split."""
    import torch

    import mi_fieldcalc_amd as mi_fc
    import mi_fieldcalc_amd.synth as synth
    from mi_fieldcalc_amd.sharding import slab_rows

    xm, ym, _ = synth.grid_maps(nx, ny)
    u, v = synth.wind(nx, ny, 515 + nx, nlev=nlev)
    flag = ALL if mode == "all" else SOME
    if mode == "some":
        for l in range(nlev):
            u[l] = synth.sprinkle_undef(u[l], 60 + l, 0.01)
        v[0, ny // nslab - 1] = cases.UNDEF  # undefined values in a row next to a slab boundary
    slabs = []
    for r in range(nslab):
        j0, n = slab_rows(ny, nslab, r)
        uh = torch.zeros((nlev, n + 2, nx), device="cuda")
        vh = torch.zeros_like(uh)
        uh[:, 1:-1] = torch.from_numpy(u[:, j0:j0 + n]).cuda()
        vh[:, 1:-1] = torch.from_numpy(v[:, j0:j0 + n]).cuda()
        rv, dg = torch.empty((nlev, n, nx), device="cuda"), torch.empty((nlev, n, nx), device="cuda")
        cnt = torch.full((nlev,), 777, dtype=torch.int64, device="cuda")
        plan = gpu_ctx.slab_plan(nx, ny, j0, n, uh, vh, torch.from_numpy(np.ascontiguousarray(xm[j0:j0 + n])).cuda(),
                                 torch.from_numpy(np.ascontiguousarray(ym[j0:j0 + n])).cuda(), rv, dg, fdefined_in=flag,
                                 n_undefined=cnt if flag != ALL else None)
        slabs.append((j0, n, uh, vh, rv, dg, cnt, plan))
    for s in slabs:
        s[7].begin()
    for r in range(nslab):  # the exchange: my first / last owned rows into the neighbours' halo rows
        _, n, uh, vh = slabs[r][:4]
        if r > 0:
            slabs[r - 1][2][:, -1], slabs[r - 1][3][:, -1] = uh[:, 1], vh[:, 1]
        if r < nslab - 1:
            slabs[r + 1][2][:, 0], slabs[r + 1][3][:, 0] = uh[:, n], vh[:, n]
    for s in slabs:
        s[7].finish()
    torch.cuda.synchronize()
    got_rv = np.concatenate([s[4].cpu().numpy() for s in slabs], axis=1)
    got_dg = np.concatenate([s[5].cpu().numpy() for s in slabs], axis=1)
    counts = sum(s[6].cpu().numpy() for s in slabs)
    for l in range(nlev):
        ok, e, f = oracle.call("relvort", nx, ny, u[l], v[l], xm, ym, fdefined=flag)
        assert ok and _bits_equal(got_rv[l], e), l
        ok, e, f2 = oracle.call("divergence", nx, ny, u[l], v[l], xm, ym, fdefined=flag)
        assert ok and _bits_equal(got_dg[l], e), l
        got = ALL if flag == ALL else mi_fc.classify(int(counts[l]), nx * ny - 2 * nx)
        assert got == f == f2, (l, got, f)
    for s in slabs:
        s[7].close()


def test_slab_plan_one_rank_communicator_and_graph_replay(gpu_ctx, oracle):
    """mifc_comm_unique_id / mifc_comm_init for ONE rank (RCCL loaded lazily, ncclCommInitRank) and mifc_slab_plan_step on the
    whole field as one slab: the second and third step replay the captured graph on NEW input (the graph reads the plan's
    buffers, not a snapshot)."""
    import torch

    import mi_fieldcalc_amd as fc
    import mi_fieldcalc_amd.synth as synth

    nx, ny, nlev = 516, 120, 4
    xm, ym, _ = synth.grid_maps(nx, ny)
    with fc.Context(0) as ctx:
        ctx.comm_init(ctx.comm_unique_id(), 0, 1)
        assert ctx.comm_info() == (True, 0, 1)
        uh = torch.zeros((nlev, ny + 2, nx), device="cuda")
        vh = torch.zeros_like(uh)
        rv, dg = torch.empty((nlev, ny, nx), device="cuda"), torch.empty((nlev, ny, nx), device="cuda")
        cnt = torch.zeros(nlev, dtype=torch.int64, device="cuda")
        plan = ctx.slab_plan(nx, ny, 0, ny, uh, vh, torch.from_numpy(xm).cuda(), torch.from_numpy(ym).cuda(), rv, dg, fdefined_in=SOME, n_undefined=cnt)
        for it in range(3):
            u, v = synth.wind(nx, ny, 90 + it, nlev=nlev)
            u[it] = synth.sprinkle_undef(u[it], it, 0.02)
            uh[:, 1:-1], vh[:, 1:-1] = torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda()
            plan.step()
            torch.cuda.synchronize()
            assert plan.uses_graph
            c = cnt.cpu().numpy()
            for l in range(nlev):
                ok, e, f = oracle.call("relvort", nx, ny, u[l], v[l], xm, ym, fdefined=SOME)
                assert _bits_equal(rv[l].cpu().numpy(), e) and fc.classify(int(c[l]), nx * ny - 2 * nx) == f, (it, l)
                ok, e, f = oracle.call("divergence", nx, ny, u[l], v[l], xm, ym, fdefined=SOME)
                assert _bits_equal(dg[l].cpu().numpy(), e), (it, l)
        plan.close()
        assert ctx.comm_release() and ctx.comm_info()[0] is False


def test_config5_driver_two_ranks_on_this_gpu():
    p = _torchrun(2, ["--config", "5", "--members", "3", "--nlev", "16", "--steps", "1", "--warmup", "1", "--check"], {"MIFC_BENCH_BACKEND": "gloo"})
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    res = _last_json(p.stdout)
    assert res["verified"] is True and res["n_gpus"] == 2


def test_config4_driver_rccl_two_gpus(oracle, tmp_path):
    """The same over RCCL on device tensors, one rank per GPU -- needs two GPUs (the builder's box has one;
    the driver's 8-GPU node runs it)."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    n = min(8, torch.cuda.device_count())
    p = _torchrun(n, ["--config", "4", "--steps", "5", "--warmup", "2", "--check", "--dump", str(tmp_path)], {})
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    res = _last_json(p.stdout)
    assert res["verified"] is True and res["n_gpus"] == n
    meta = json.load(open(tmp_path / "config4_meta.json"))
    assert meta["nx"] == 4000
    p = _torchrun(min(2, n), ["--config", "5", "--members", "4", "--nlev", "16", "--steps", "1", "--warmup", "1", "--check"], {})
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert _last_json(p.stdout)["verified"] is True


# ------------------------------------------------------------------ strict 1e-5 report
def test_zz_report_cells_beyond_strict_relative_tolerance():
    """The 1e-5 bound of the Celsius / wind-chill outputs is taken against the Kelvin-sized terms they are a
    difference of (gpu_util.compare).  This writes, per operator, how many compared cells exceed a STRICT
    1e-5 * |expected| and by how much -- reported, not asserted."""
    rows = gpu_util.strict_report()
    text = "\n".join(rows)
    print(text)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "strict_tolerance_report.txt"), "w") as f:
            f.write(text + "\n")
    assert rows


# ------------------------------------------------------------------ config 1: 256 x 256 through the unchanged C++ signature
CONFIG1_CALLER = r"""
// BASELINE.json config 1: wind speed from u/v on one 256x256 float32 field through the reference's own C++
// signature (host pointers, ValuesDefined&), exactly as an existing caller is written.
#include <mi_fieldcalc/FieldCalculations.h>
#include <cstdio>
#include <vector>
int main(int argc, char** argv)
{
  const int nx = 256, ny = 256;
  std::vector<float> u(nx * ny), v(nx * ny), ff(nx * ny, -1.f);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(u.data(), 4, u.size(), f) != u.size() || std::fread(v.data(), 4, v.size(), f) != v.size())
    return 2;
  std::fclose(f);
  miutil::ValuesDefined fDefined = argv[3][0] == 'A' ? miutil::ALL_DEFINED : miutil::SOME_DEFINED;
  if (!miutil::fieldcalc::vectorabs(nx, ny, u.data(), v.data(), ff.data(), fDefined, miutil::UNDEF))
    return 3;
  f = std::fopen(argv[2], "wb");
  std::fwrite(ff.data(), 4, ff.size(), f);
  std::fclose(f);
  std::printf("%d\n", (int)fDefined);
  return 0;
}
"""


@pytest.mark.parametrize("mode", ["all", "some"])
def test_config1_vectorabs_256x256_through_the_cxx_signature(oracle, tmp_path, mode):
    import mi_fieldcalc_amd.synth as synth

    nx = ny = 256
    u, v = synth.wind(nx, ny, 0x5EED0000 + 1000)
    flag = ALL
    if mode == "some":
        u, flag = synth.sprinkle_undef(u, 3, 0.01), SOME
    src, exe, fin, fout = tmp_path / "c1.cc", tmp_path / "c1", tmp_path / "in.bin", tmp_path / "out.bin"
    src.write_text(CONFIG1_CALLER)
    inc, libdir = os.path.join(ROOT, "mi-fieldcalc_amd", "include"), os.path.join(ROOT, "mi-fieldcalc_amd")
    subprocess.run(["g++", "-std=c++11", "-Wall", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lmi-fieldcalc", "-lmifc", "-Wl,-rpath," + libdir],
                   check=True)
    with open(fin, "wb") as f:
        f.write(u.tobytes())
        f.write(v.tobytes())
    res = subprocess.run([str(exe), str(fin), str(fout), "A" if flag == ALL else "S"], capture_output=True, text=True, check=True)
    got = np.fromfile(fout, np.float32).reshape(ny, nx)
    ok, e, f_e = oracle.call("vectorabs", nx, ny, u, v, fdefined=flag)
    assert ok and _bits_equal(got, e) and int(res.stdout.split()[0]) == f_e


@pytest.mark.parametrize("mode", ["all", "some"])
def test_counts_of_launches_above_the_partial_count_threshold(gpu_ctx, oracle, mode):
    """From 2 048 workgroups (2.1 M cells) on, the one-shot elementwise / pointwise kernels leave their undefined counts in
    per-workgroup slots that a small kernel adds up (no queue of same-address atomics): values, counts and flags of a
    2048 x 1100 field with undefined values everywhere equal the reference's."""
    grid = (2048, 1100)
    ew = [c for c in cases.ewise_cases(grids=[grid], modes=(mode,)) if c["op"] in ("vectorabs", "hlevelhum", "cvhum")]
    seen, keep = set(), []
    for c in ew:
        if c["op"] not in seen:
            seen.add(c["op"])
            keep.append(c)
    pw = _pick(cases.catalogue_cases(grids=[grid], modes=(mode,)), {"cvtemp": {1}, "fieldOPERfield": {1}, "kIndex": {1}, "showalterIndex": {1}})
    seen = set()
    for c in pw:
        if c["op"] not in seen:
            seen.add(c["op"])
            keep.append(c)
    assert len(keep) >= 6
    _run_big(gpu_ctx, oracle, keep, device=True)


def test_masked_level_batch_counts(gpu_ctx, oracle, mifc_env):
    """A level batch whose undefined values are EVERYWHERE (every wave of every workgroup has something to count): the
    split-role kernel adds a level's counts up in LDS and hands one total per workgroup to the level's counter; the first
    level-walking form counts per wave.  Counts (flags) and values of sampled levels equal the reference's, both forms."""
    import torch

    import mi_fieldcalc_amd as fc
    import mi_fieldcalc_amd.synth as synth

    nx, ny, nlev = 1440, 720, 13
    xm, ym, _ = synth.grid_maps(nx, ny)
    u, v = synth.wind(nx, ny, 2025, nlev=nlev)
    rng = np.random.default_rng(7)
    for l in range(nlev):  # level 3 stays clean, level 5 is undefined altogether, the others carry 0.1 .. 30 %
        frac = 0.0 if l == 3 else (1.0 if l == 5 else (0.001, 0.02, 0.3)[l % 3])
        m = rng.random((ny, nx)) < frac
        u[l][m] = cases.UNDEF
        v[l][rng.random((ny, nx)) < frac / 2] = np.nan
    du, dv, dxm, dym = (torch.from_numpy(a).cuda() for a in (u, v, xm, ym))
    flags = np.full(nlev, fc.SOME_DEFINED, np.int32)
    for split in ("1", "0"):
        mifc_env("MIFC_VORTDIV_SPLIT", split)
        (rv, dg), fo = gpu_ctx.vortdiv_levels(du, dv, dxm, dym, fdefined=flags)
        for l in range(nlev):
            ok, rv_e, f1 = oracle.call("relvort", nx, ny, u[l], v[l], xm, ym, fdefined=fc.SOME_DEFINED)
            ok2, dv_e, f2 = oracle.call("divergence", nx, ny, u[l], v[l], xm, ym, fdefined=fc.SOME_DEFINED)
            assert ok and ok2 and f1 == f2 == fo[l], (split, l, f1, fo[l])
            if l in (0, 3, 5, nlev - 1):
                assert _bits_equal(rv[l].cpu().numpy(), rv_e) and _bits_equal(dg[l].cpu().numpy(), dv_e)


@pytest.mark.parametrize("undef", [float("nan"), -32767.0])
def test_deep_batch_with_an_unusual_undef_value(gpu_ctx, oracle, undef):
    """The split-role kernel tests a value with ONE compare ("ordered and != undef"), which is is_defined() only for an
    undef that is not NaN: a caller whose undef IS NaN gets the level-walking kernel with the generic test.  Deep batch,
    tested flags, NaN and undef values sprinkled, against the reference with the same undef."""
    import torch

    import mi_fieldcalc_amd as fc
    import mi_fieldcalc_amd.synth as synth

    nx, ny, nlev = 1440, 360, 16
    xm, ym, _ = synth.grid_maps(nx, ny)
    u, v = synth.wind(nx, ny, 77, nlev=nlev)
    rng = np.random.default_rng(3)
    for l in range(nlev):
        if l % 4 == 1:
            continue  # clean levels
        u[l][rng.random((ny, nx)) < 0.01] = np.float32(undef)
        v[l][rng.random((ny, nx)) < 0.01] = np.nan
        if not np.isnan(undef):
            v[l][rng.random((ny, nx)) < 0.005] = np.float32(undef)
    du, dv, dxm, dym = (torch.from_numpy(a).cuda() for a in (u, v, xm, ym))
    flags = np.full(nlev, fc.SOME_DEFINED, np.int32)
    (rv, dg), fo = gpu_ctx.vortdiv_levels(du, dv, dxm, dym, fdefined=flags, undef=undef)
    for l in range(nlev):
        ok, rv_e, f1 = oracle.call("relvort", nx, ny, u[l], v[l], xm, ym, fdefined=fc.SOME_DEFINED, undef=undef)
        ok2, dv_e, f2 = oracle.call("divergence", nx, ny, u[l], v[l], xm, ym, fdefined=fc.SOME_DEFINED, undef=undef)
        assert ok and ok2 and f1 == f2 == fo[l], (l, f1, f2, fo[l])
        if l in (0, 1, 7, nlev - 1):
            assert _bits_equal(rv[l].cpu().numpy(), rv_e) and _bits_equal(dg[l].cpu().numpy(), dv_e), l


def test_masked_gradient_level_batch_counts(gpu_ctx, oracle):
    """Deep batches of gradient compute 1 / 2 run on the scalar level-walking kernel, which adds a level's undefined counts up
    in LDS and hands one total per workgroup and level to the counter (a masked field used to queue 4 320 same-address
    atomics per level).  Flags of every level (clean, masked, without any defined value) and values of sampled levels
    equal the reference's."""
    import torch

    import mi_fieldcalc_amd as fc
    import mi_fieldcalc_amd.synth as synth

    nx, ny, nlev = 1440, 720, 13
    xm, ym, _ = synth.grid_maps(nx, ny)
    z = np.stack([synth.scalar_field(nx, ny, 300 + l) for l in range(nlev)])
    rng = np.random.default_rng(11)
    for l in range(nlev):  # level 3 stays clean, levels 5 and 12 (the last of a chunk) are undefined altogether
        frac = 0.0 if l == 3 else (1.0 if l in (5, nlev - 1) else (0.001, 0.02, 0.3)[l % 3])
        z[l][rng.random((ny, nx)) < frac] = cases.UNDEF
    dz, dxm, dym = (torch.from_numpy(a).cuda() for a in (z, xm, ym))
    flags = np.full(nlev, fc.SOME_DEFINED, np.int32)
    for name, compute in (("gradient1", 1), ("gradient2", 2)):
        (o0, _), fo = gpu_ctx.stencil_levels(name, dz, None, dxm, dym, None, fdefined=flags)
        for l in range(nlev):
            ok, e, f_e = oracle.call("gradient", nx, ny, z[l], xm, ym, compute, fdefined=fc.SOME_DEFINED)
            assert ok and f_e == fo[l], (name, l, f_e, fo[l])
            if l in (0, 3, 5, 7, nlev - 1):
                assert _bits_equal(o0[l].cpu().numpy(), e), (name, l)
        # a level without any defined value: SOME_DEFINED for compute 1 (its count never covers the whole field), NONE_DEFINED
        # for compute 2 -- which pins the kernel's count of such a level exactly
        empty = fc.SOME_DEFINED if compute == 1 else fc.NONE_DEFINED
        assert fo[3] == fc.ALL_DEFINED and fo[5] == empty and fo[nlev - 1] == empty, (name, fo)


@pytest.mark.parametrize("tune", [None, "K=1", "K=2", "K=2,RB=14", "R=8"])
def test_masked_single_level_counts(gpu_ctx, oracle, tune, mifc_env):
    """One and two levels with undefined values everywhere through the forms a shallow launch takes (one-shot, one-shot
    tiles, row-walking): every workgroup hands ONE total to the level's counter; counts, flags and values equal the reference's.
    Then a deep batch with a single output (the first level-walking form, per-level counts added up in LDS)."""
    import torch

    import mi_fieldcalc_amd as fc
    import mi_fieldcalc_amd.synth as synth

    if tune:
        mifc_env("MIFC_VORTDIV_TUNE", tune)
    nx, ny = 1440, 725  # a last row block that is not full
    xm, ym, _ = synth.grid_maps(nx, ny)
    u, v = synth.wind(nx, ny, 4711, nlev=2)
    rng = np.random.default_rng(11)
    u[0][rng.random((ny, nx)) < 0.02] = cases.UNDEF
    v[1][rng.random((ny, nx)) < 0.3] = np.nan
    du, dv, dxm, dym = (torch.from_numpy(a).cuda() for a in (u, v, xm, ym))
    expect = []
    for l in range(2):
        ok, rv_e, f1 = oracle.call("relvort", nx, ny, u[l], v[l], xm, ym, fdefined=fc.SOME_DEFINED)
        ok2, dv_e, f2 = oracle.call("divergence", nx, ny, u[l], v[l], xm, ym, fdefined=fc.SOME_DEFINED)
        assert ok and ok2 and f1 == f2
        expect.append((rv_e, dv_e, f1))
    for nl in (1, 2):
        (rv, dg), fo = gpu_ctx.vortdiv_levels(du[:nl], dv[:nl], dxm, dym, fdefined=[fc.SOME_DEFINED] * nl)
        for l in range(nl):
            assert fo[l] == expect[l][2]
            assert _bits_equal(rv[l].cpu().numpy(), expect[l][0]) and _bits_equal(dg[l].cpu().numpy(), expect[l][1])
    if tune is None:
        nlev = 9
        uu = np.repeat(u[:1], nlev, axis=0).copy()
        vv = np.repeat(v[1:2], nlev, axis=0).copy()
        duu, dvv = torch.from_numpy(uu).cuda(), torch.from_numpy(vv).cuda()
        ok, rv_e, f1 = oracle.call("relvort", nx, ny, uu[0], vv[0], xm, ym, fdefined=fc.SOME_DEFINED)
        (rv, none), fo = gpu_ctx.vortdiv_levels(duu, dvv, dxm, dym, fdefined=[fc.SOME_DEFINED] * nlev, want=("rvort",))
        assert none is None and all(f == f1 for f in fo)
        assert _bits_equal(rv[0].cpu().numpy(), rv_e) and _bits_equal(rv[nlev - 1].cpu().numpy(), rv_e)
