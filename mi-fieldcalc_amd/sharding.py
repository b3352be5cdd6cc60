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


def exchange_halo_rows(fields_with_halo, rank, world_size, group=None):
    """fields_with_halo: list of tensors of shape (ny_local + 2, nx) whose rows
    1..ny_local are owned.  Fills row 0 from the rank above (its last owned row)
    and row ny_local+1 from the rank below (its first owned row).  The outermost
    halo rows of the first / last rank are left as they are (never read)."""
    import torch.distributed as dist

    ops = []
    for f in fields_with_halo:
        if rank > 0:
            ops.append(dist.P2POp(dist.isend, f[1], rank - 1, group))
            ops.append(dist.P2POp(dist.irecv, f[0], rank - 1, group))
        if rank < world_size - 1:
            ops.append(dist.P2POp(dist.isend, f[-2], rank + 1, group))
            ops.append(dist.P2POp(dist.irecv, f[-1], rank + 1, group))
    if not ops:
        return
    for req in dist.batch_isend_irecv(ops):
        req.wait()


def global_undefined_count(local_count_tensor, group=None):
    """Sum of the per-slab undefined counts (int64 tensor of one element)."""
    import torch.distributed as dist

    dist.all_reduce(local_count_tensor, op=dist.ReduceOp.SUM, group=group)
    return local_count_tensor
