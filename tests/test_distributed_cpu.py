"""N>1 paths on the CPU with the gloo backend, world_size 2 (and 3): the level
sharding used by bench.py and the row-slab halo exchange of config 4.  The HIP
kernels cannot run here; the slab arithmetic is checked with the oracle on the
exchanged buffers (rows that depend on the halo), the kernels themselves are
checked against the same decomposition on the GPU (test_gpu_parity.py)."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nx, ny, result_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist

    import mi_fieldcalc_amd.synth as synth
    from cpulib import CpuLib
    from mi_fieldcalc_amd.sharding import exchange_halo_rows, global_undefined_count, shard_range, slab_rows

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        oracle = CpuLib("oracle")
        xm, ym, _ = synth.grid_maps(nx, ny)
        u, v = synth.wind(nx, ny, 4711)
        u = synth.sprinkle_undef(u, 1, 0.03)
        v = synth.sprinkle_undef(v, 2, 0.03)
        ok, rv_g, flag_g = oracle.call("relvort", nx, ny, u, v, xm, ym, fdefined=2)
        j0, nloc = slab_rows(ny, world, rank)
        # every rank starts with ONLY its own rows; halos are poisoned
        uh = torch.full((nloc + 2, nx), float("nan"))
        vh = torch.full((nloc + 2, nx), float("nan"))
        uh[1:-1] = torch.from_numpy(u[j0:j0 + nloc])
        vh[1:-1] = torch.from_numpy(v[j0:j0 + nloc])
        exchange_halo_rows([uh, vh], rank, world)
        if rank > 0:
            assert np.array_equal(uh[0].numpy(), u[j0 - 1], equal_nan=True) and np.array_equal(vh[0].numpy(), v[j0 - 1], equal_nan=True)
        if rank < world - 1:
            assert np.array_equal(uh[-1].numpy(), u[j0 + nloc], equal_nan=True) and np.array_equal(vh[-1].numpy(), v[j0 + nloc], equal_nan=True)
        # owned rows that are interior to the GLOBAL field, computed from the haloed buffer as a mini field
        lo = 0 if rank > 0 else 1            # buffer row index (0 = north halo)
        hi = nloc + 2 if rank < world - 1 else nloc + 1
        mini_u = np.ascontiguousarray(uh.numpy()[lo:hi])
        mini_v = np.ascontiguousarray(vh.numpy()[lo:hi])
        g0 = j0 - 1 + lo                     # global row of mini row 0
        mny = hi - lo
        ok, rv_m, _ = oracle.call("relvort", nx, mny, mini_u, mini_v, xm[g0:g0 + mny], ym[g0:g0 + mny], fdefined=2)
        # mini rows 1..mny-2 are true stencil results (columns 1..nx-2 untouched by the mini field's edge fill)
        a = rv_m[1:-1, 1:-1]
        b = rv_g[g0 + 1:g0 + mny - 1, 1:-1]
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        # undefined count: per-slab shares add up to the global count through one all-reduce
        rows = [j for j in range(j0, j0 + nloc) if 1 <= j <= ny - 2]
        # raw count (before the edge fill) over this slab's rows, from the definition of the test
        def bad(x):
            return np.isnan(x) | (x == synth.UNDEF)
        fu, fv = u.reshape(-1), v.reshape(-1)
        local = 0
        for j in rows:
            i = np.arange(j * nx, (j + 1) * nx)
            local += int(np.count_nonzero(bad(fv[i - 1]) | bad(fv[i + 1]) | bad(fu[i - nx]) | bad(fu[i + nx])))
        t = torch.tensor([local], dtype=torch.int64)
        global_undefined_count(t)
        total = int(t.item())
        ii = np.arange(nx, nx * ny - nx)
        expect = int(np.count_nonzero(bad(fv[ii - 1]) | bad(fv[ii + 1]) | bad(fu[ii - nx]) | bad(fu[ii + nx])))
        assert total == expect
        assert (0 if total == 0 else (1 if total == nx * ny - 2 * nx else 2)) == flag_g
        # level sharding: ranks cover all levels exactly once
        a0, a1 = shard_range(137, world, rank)
        cover = torch.zeros(137, dtype=torch.int64)
        cover[a0:a1] = 1
        dist.all_reduce(cover)
        assert bool((cover == 1).all())
        open(os.path.join(result_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_halo_exchange_and_sharding_gloo(world, tmp_path):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(world, port, 48, 23, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(world))


def _ensemble_worker(rank, world, port, nx, ny, nmem, result_dir):
    """Members sharded over ranks -> every rank collects all members of its row slab
    (sharding.reshard_members_to_rows) and reduces locally, in member order.  The
    reduction itself is a HIP kernel and cannot run here; the oracle reduces the
    resharded slab and the result must equal the whole-field reference bit for bit."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist

    import mi_fieldcalc_amd.synth as synth
    from cpulib import CpuLib
    from mi_fieldcalc_amd.sharding import combine_slab_flags, gather_member_flags, reshard_members_to_rows, shard_range, slab_rows

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        oracle = CpuLib("oracle")
        members = [synth.uniform((ny, nx), 900 + k, -5.0, 30.0).astype(np.float32) for k in range(nmem)]
        members[1] = synth.sprinkle_undef(members[1], 3, 0.2)
        flags = [2 if k == 1 else 0 for k in range(nmem)]
        ok, mean_g, flag_g = oracle.call("meanValue", nx, ny, members, flags, fdefined=2)
        ok, std_g, _ = oracle.call("stddevValue", nx, ny, members, flags, fdefined=2)
        m0, m1 = shard_range(nmem, world, rank)
        local = torch.from_numpy(np.stack(members[m0:m1])) if m1 > m0 else torch.empty((0, ny, nx))
        slab = reshard_members_to_rows(local, rank, world)
        j0, rows = slab_rows(ny, world, rank)
        assert tuple(slab.shape) == (nmem, rows, nx)
        for k in range(nmem):
            assert np.array_equal(slab[k].numpy().view(np.uint32), members[k][j0:j0 + rows].view(np.uint32))
        all_flags = gather_member_flags(flags[m0:m1], rank, world)
        assert all_flags == flags
        slab_members = [np.ascontiguousarray(slab[k].numpy()) for k in range(nmem)]
        ok, mean_l, flag_l = oracle.call("meanValue", nx, rows, slab_members, all_flags, fdefined=2)
        ok, std_l, _ = oracle.call("stddevValue", nx, rows, slab_members, all_flags, fdefined=2)
        assert np.array_equal(mean_l.view(np.uint32), mean_g[j0:j0 + rows].view(np.uint32))
        assert np.array_equal(std_l.view(np.uint32), std_g[j0:j0 + rows].view(np.uint32))
        assert combine_slab_flags(flag_l) == flag_g
        open(os.path.join(result_dir, "ens%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nmem", [(2, 5), (3, 7), (3, 2)])
def test_member_sharded_ensemble_reshard_gloo(world, nmem, tmp_path):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_ensemble_worker, args=(world, port, 40, 17, nmem, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ens%d" % r)) for r in range(world))
