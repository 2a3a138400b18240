"""world_size-2 gloo test of the multi-GPU path's host logic: contiguous sharding by signature,
no data-path collective, one sum all-reduce for the aggregate verdict, optional status gather.
The per-shard verifier here is the CPU oracle (tests may use it); on GPUs bench.py runs the same
sharding helpers with the HIP engine and RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import Oracle
    from schnorr_sig_amd.sharding import (aggregate_fail_count, batch_verdict, broadcast_rows, gather_status,
                                          scatter_rows, shard_range)
    orc = Oracle()
    rng = np.random.default_rng(77)     # every rank derives the same global batch
    sks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sks[:, 31] &= 0x3F; sks[:, 0] |= 1
    nonces = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); nonces[:, 31] &= 0x3F; nonces[:, 0] |= 1
    msgs = rng.integers(0, 256, size=(n, 80), dtype=np.uint8)
    lo, hi = shard_range(n, rank, world)
    # rank-0-resident messages reach the other ranks by scatter; must equal the local slice
    full = torch.from_numpy(msgs) if rank == 0 else None
    got = scatter_rows(full, n, 80, rank, world, dist)
    assert (got.numpy() == msgs[lo:hi]).all()
    # ... or by broadcast of the whole array (north_star's variant; bench.py times both)
    got_b = broadcast_rows(torch.from_numpy(msgs.copy()) if rank == 0 else None, n, 80, rank, world, dist)
    assert (got_b.numpy() == msgs[lo:hi]).all()
    # equal shards are sent straight out of the resident array (views)
    even = scatter_rows(torch.from_numpy(msgs[:36].copy()) if rank == 0 else None, 36, 80, rank, world, dist)
    assert (even.numpy() == msgs[:36][18 * rank:18 * (rank + 1)]).all()
    pks, sigs = orc.keygen_sign_many(sks[lo:hi], nonces[lo:hi], msgs[lo:hi], threads=2)
    bad_global = [3, n // 2, n - 1]
    for b in bad_global:
        if lo <= b < hi:
            sigs[b - lo, 50] ^= 1
    st = orc.verify_many(sigs, pks, msgs[lo:hi], check_torsion=False, threads=2)
    cnt = torch.tensor([int((st != 0).sum())], dtype=torch.int64)
    aggregate_fail_count(cnt, dist)
    full = gather_status(torch.from_numpy(st), n, rank, world, dist)
    q.put((rank, int(cnt.item()), batch_verdict(cnt.item()), full.numpy().tolist(), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_batch_verdict_world2():
    world, n = 2, 37            # ragged: 19 + 18
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert [r[4] for r in res] == [(0, 19), (19, 37)]
    for _, cnt, verdict, full, _ in res:
        assert cnt == 3 and verdict == 2
        assert len(full) == n
        assert [i for i, s in enumerate(full) if s != 0] == [3, n // 2, n - 1]
        assert all(full[i] == 2 for i in (3, n // 2, n - 1))
