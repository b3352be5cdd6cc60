#!/usr/bin/env python3
"""rccl_selfcheck.py -- measurement / rehearsal only (tools/).

The builder's boxes have ONE GPU, so the N-rank drivers (bench.py --gpus N, tools/bench_multigpu.py) have only run
with gloo there.  This script takes the RCCL ("nccl") backend itself through everything those drivers and
mi_fieldcalc_amd.sharding ask of it, in a group of ONE rank on cuda:0: process-group creation bound to the
device, barrier, the all-reduces on device tensors (float64 MAX for the timing, int64 SUM / MIN for counts and
flags), all_gather, a grouped send/recv of a device row (to the rank itself -- the only peer there is) and the
sharding helpers on device tensors.  It cannot show a second GPU's behaviour; it shows that the calls, dtypes and
tensor placements are ones RCCL on this image accepts.

Run: python3 tools/rccl_selfcheck.py        (prints one line per step; exit code 0 = all passed)
"""
import os
import socket
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(free_port()))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist

    import mi_fieldcalc_amd as fc
    import mi_fieldcalc_amd.sharding as sh

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    print("process group: backend %s, world %d" % (dist.get_backend(), dist.get_world_size()), flush=True)
    try:
        print("RCCL version", torch.cuda.nccl.version(), flush=True)
    except Exception as e:  # informational only
        print("RCCL version unavailable:", e, flush=True)

    dist.barrier()
    torch.cuda.synchronize()
    print("barrier ok", flush=True)

    t = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t.item()) == 1.25
    print("all_reduce MAX float64 on the device ok", flush=True)

    c = torch.tensor([123456789012], dtype=torch.int64, device=dev)
    assert int(sh.global_undefined_count(c).item()) == 123456789012
    print("global_undefined_count (int64 SUM) ok", flush=True)

    assert sh.combine_slab_flags(fc.ALL_DEFINED) == fc.ALL_DEFINED
    assert sh.combine_slab_flags(fc.SOME_DEFINED) == fc.SOME_DEFINED
    print("combine_slab_flags (int64 MIN) ok", flush=True)

    flags = [fc.ALL_DEFINED, fc.SOME_DEFINED, fc.NONE_DEFINED]
    assert sh.gather_member_flags(flags, 0, 1) == [int(f) for f in flags]
    print("gather_member_flags (all_gather int64) ok", flush=True)

    members = torch.arange(3 * 8 * 16, dtype=torch.float32, device=dev).reshape(3, 8, 16)
    rows = sh.reshard_members_to_rows(members, 0, 1)
    assert torch.equal(rows, members)
    print("reshard_members_to_rows ok", flush=True)

    # halo rows: with one rank there is no neighbour, so the call must be a no-op that returns
    f = torch.arange(6 * 32, dtype=torch.float32, device=dev).reshape(6, 32)
    g = f.clone()
    sh.exchange_halo_rows([g], 0, 1)
    assert torch.equal(f, g)
    print("exchange_halo_rows (no neighbour) ok", flush=True)

    # the grouped device-to-device send/recv the exchange is made of, to the only peer there is
    src = torch.arange(4000, dtype=torch.float32, device=dev)
    dst = torch.zeros(4000, dtype=torch.float32, device=dev)
    ops = [dist.P2POp(dist.isend, src, 0), dist.P2POp(dist.irecv, dst, 0)]
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    torch.cuda.synchronize()
    assert torch.equal(src, dst)
    print("batch_isend_irecv of a 16-kB device row (self) ok", flush=True)

    dist.barrier()
    dist.destroy_process_group()
    print("all steps passed", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
