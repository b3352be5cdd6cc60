"""Multi-GPU decomposition of the field operators.

Two cases, one process per GPU (torch.distributed; backend "nccl" is RCCL on
ROCm, "gloo" in the CPU tests):

* levels / ensemble members are independent 2-D fields -> ``shard_range``
  partitions the flattened (member, level) index range contiguously; there is
  NO collective on the data path.  The only cross-rank datum is the per-field
  undefined count, which each rank keeps for its own fields.

* one large field split along y into row slabs (BASELINE.json config 4) ->
  ``exchange_halo_rows`` swaps ONE row of every differenced input with the two
  neighbours (grouped send/recv, nearest neighbour only, nx*4 bytes per field
  and direction over xGMI), then every rank runs the slab kernel
  (mifc_vortdiv_slab_enqueue); the undefined counts are summed with one
  8-byte all-reduce to classify the whole field.

* reductions over ensemble members when the MEMBERS are what
  is sharded -> ``reshard_members_to_rows`` moves the ensemble once so that
  every rank holds all members of its own row slab (grouped send/recv), the
  per-cell reduction then runs locally in member order (bit-identical to the
  unsharded result); ``gather_member_flags`` / ``combine_slab_flags`` carry the
  ValuesDefined flags across.
"""


def shard_range(n_items, world_size, rank):
    """Contiguous block of ``n_items`` for ``rank``: sizes differ by at most one,
    the first (n_items % world_size) ranks get the extra item."""
    base, extra = divmod(int(n_items), int(world_size))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def slab_rows(ny, world_size, rank):
    """(first global row, owned rows) of a rank's row slab."""
    j0, j1 = shard_range(ny, world_size, rank)
    return j0, j1 - j0


class HaloExchange:
    """One halo exchange in flight (begin_halo_exchange); wait() completes it.

    With RCCL the transfers run on the communicator's own stream: wait() makes the
    CURRENT stream wait for them (no host block), so kernels enqueued between begin and
    wait overlap the exchange.  With gloo (CPU rehearsal; device tensors are staged
    through host rows there) wait() blocks the host and copies the rows back."""

    def __init__(self, reqs, staged):
        self._reqs, self._staged = reqs, staged

    def wait(self):
        for req in self._reqs:
            req.wait()
        for dst, src in self._staged:
            dst.copy_(src)
        self._reqs, self._staged = [], []


def begin_halo_exchange(fields_with_halo, rank, world_size, group=None):
    """fields_with_halo: list of tensors of shape (ny_local + 2, nx) whose rows
    1..ny_local are owned.  Starts filling row 0 from the rank above (its last owned
    row) and row ny_local+1 from the rank below (its first owned row); the outermost
    halo rows of the first / last rank are left as they are (never read).  Nearest
    neighbour only: 2 messages of nx*4 bytes per field and neighbour (16 kB at nx=4000),
    grouped into one RCCL group call."""
    import torch.distributed as dist

    via_host = str(dist.get_backend(group)).lower() != "nccl"
    ops, staged = [], []

    def send(row, peer):
        ops.append(dist.P2POp(dist.isend, row.cpu() if (via_host and row.is_cuda) else row, peer, group))

    def recv(row, peer):
        if via_host and row.is_cuda:
            import torch

            buf = torch.empty(row.shape, dtype=row.dtype)
            staged.append((row, buf))
            ops.append(dist.P2POp(dist.irecv, buf, peer, group))
        else:
            ops.append(dist.P2POp(dist.irecv, row, peer, group))

    for f in fields_with_halo:
        if rank > 0:
            send(f[1], rank - 1)
            recv(f[0], rank - 1)
        if rank < world_size - 1:
            send(f[-2], rank + 1)
            recv(f[-1], rank + 1)
    reqs = dist.batch_isend_irecv(ops) if ops else []
    return HaloExchange(reqs, staged)


def exchange_halo_rows(fields_with_halo, rank, world_size, group=None):
    """Blocking form of begin_halo_exchange()."""
    begin_halo_exchange(fields_with_halo, rank, world_size, group).wait()


def vortdiv_slab_overlapped(ctx, nx, ny_global, j0, ny_local, u_halo, v_halo, xmapr, ymapr, rvort, diverg, rank, world_size,
                            fdefined_in=2, undef=1.0e35, n_undefined=None, group=None):
    """One decomposed step of BASELINE.json config 4 on this rank's row slab: start the halo
    exchange of u and v, compute the owned rows that read no halo row while it is in flight,
    then the two boundary strips (mifc_vortdiv_slab_rows_enqueue).  Asynchronous on the
    current stream; n_undefined (int64[1], device) receives this slab's undefined count."""
    ex = begin_halo_exchange([u_halo, v_halo], rank, world_size, group)
    b = 2  # strip height: keeps rows 0/1 and ny-2/ny-1 of the whole field together
    if ny_local >= 3 * b + 1:
        ok = ctx.vortdiv_slab_enqueue(nx, ny_global, j0, ny_local, u_halo, v_halo, xmapr, ymapr, rvort, diverg, fdefined_in=fdefined_in, undef=undef,
                                      n_undefined=n_undefined, rows=(b, ny_local - b))
        ex.wait()
        for rows in ((0, b), (ny_local - b, ny_local)):
            ok = ok and ctx.vortdiv_slab_enqueue(nx, ny_global, j0, ny_local, u_halo, v_halo, xmapr, ymapr, rvort, diverg, fdefined_in=fdefined_in,
                                                 undef=undef, n_undefined=n_undefined, rows=rows, accumulate=True)
        return ok
    ex.wait()  # a thin slab: nothing worth overlapping
    return ctx.vortdiv_slab_enqueue(nx, ny_global, j0, ny_local, u_halo, v_halo, xmapr, ymapr, rvort, diverg, fdefined_in=fdefined_in, undef=undef,
                                    n_undefined=n_undefined)


def _collective_device(group=None):
    """Small bookkeeping tensors live where the backend wants them (RCCL: on the GPU)."""
    import torch
    import torch.distributed as dist

    if str(dist.get_backend(group)).lower() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def reshard_members_to_rows(local_members, rank, world_size, group=None):
    """Ensemble reductions (sumFields, meanValue, stddevValue, extremeValue,
    probability) reduce over the members of every cell IN MEMBER ORDER; that order is
    what makes the float results identical to the reference's.  When the members
    are sharded over ranks (``shard_range`` blocks), each rank therefore first
    collects, for its own row slab, the rows of ALL members -- one all-to-all of
    the ensemble, done here with grouped send/recv (RCCL over xGMI; gloo in the CPU
    tests) -- and then reduces locally without any further exchange.

    local_members: tensor [m_local, ny, nx], this rank's contiguous block of members.
    Returns a tensor [m_total, rows_local, nx] in global member order, where
    rows_local = slab_rows(ny, world_size, rank)[1]."""
    import torch
    import torch.distributed as dist

    m_local, ny, nx = local_members.shape
    counts = torch.zeros(world_size, dtype=torch.int64, device=local_members.device)
    counts[rank] = m_local
    dist.all_reduce(counts, group=group)
    counts = [int(c) for c in counts.tolist()]
    j0, rows = slab_rows(ny, world_size, rank)
    parts = [None] * world_size
    send_keep = []
    ops = []
    for peer in range(world_size):
        if peer == rank:
            parts[peer] = local_members[:, j0:j0 + rows, :]
            continue
        p0, prow = slab_rows(ny, world_size, peer)
        if m_local > 0 and prow > 0:
            out = local_members[:, p0:p0 + prow, :].contiguous()
            send_keep.append(out)
            ops.append(dist.P2POp(dist.isend, out, peer, group))
        buf = torch.empty((counts[peer], rows, nx), dtype=local_members.dtype, device=local_members.device)
        parts[peer] = buf
        if counts[peer] > 0 and rows > 0:
            ops.append(dist.P2POp(dist.irecv, buf, peer, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return torch.cat(parts, dim=0).contiguous()


def gather_member_flags(local_flags, rank, world_size, group=None):
    """Per-member ValuesDefined flags of all ranks, in global member order."""
    import torch
    import torch.distributed as dist

    dev = _collective_device(group)
    n = torch.zeros(world_size, dtype=torch.int64, device=dev)
    n[rank] = len(local_flags)
    dist.all_reduce(n, group=group)
    width = int(n.max().item())
    mine = torch.full((width,), -1, dtype=torch.int64, device=dev)
    mine[:len(local_flags)] = torch.tensor([int(f) for f in local_flags], dtype=torch.int64, device=dev)
    gathered = [torch.empty(width, dtype=torch.int64, device=dev) for _ in range(world_size)]
    dist.all_gather(gathered, mine, group=group)
    return [int(x) for r, g in enumerate(gathered) for x in g[:int(n[r].item())].tolist()]


def combine_slab_flags(local_flag, group=None):
    """ValuesDefined of the whole field from the flags of its row slabs (every
    slab non-empty): ALL if all are ALL, NONE if all are NONE, else SOME."""
    import torch
    import torch.distributed as dist

    t = torch.tensor([1 if local_flag == 0 else 0, 1 if local_flag == 1 else 0], dtype=torch.int64, device=_collective_device(group))
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return 0 if int(t[0]) == 1 else (1 if int(t[1]) == 1 else 2)


def global_undefined_count(local_count_tensor, group=None):
    """Sum of the per-slab undefined counts (int64 tensor of one element)."""
    import torch.distributed as dist

    dist.all_reduce(local_count_tensor, op=dist.ReduceOp.SUM, group=group)
    return local_count_tensor
