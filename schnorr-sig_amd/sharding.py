"""Batch-level sharding of a verification batch across the GPUs of one node.

The path shards by signature (each verification reads only its own (sig, pk, msg):
reference src/signature.rs:181-205), so there is no data-path collective.  The only
exchange is the aggregate verdict: one sum all-reduce of the per-rank rejection counts
(RCCL over xGMI on GPUs; gloo in the CPU tests), and an optional gather of status bytes.
"""


def shard_range(n, rank, world):
    """Contiguous range [lo, hi) of rank `rank`: sizes differ by at most one, earlier ranks larger."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def aggregate_fail_count(local_count_tensor, dist=None):
    """In-place sum over ranks of a 1-element integer tensor; returns it. dist=None: single rank."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(local_count_tensor, op=dist.ReduceOp.SUM)
    return local_count_tensor


def batch_verdict(total_fail):
    """verify_batch's Result for the whole sharded batch (src/batch.rs:125-129): 0 Ok, 2 InvalidSignature."""
    return 0 if int(total_fail) == 0 else 2


def gather_status(local_status_tensor, n, rank, world, dist):
    """All-gather of per-shard status bytes into the full n-vector (ragged shards padded to the max)."""
    import torch
    if dist is None:
        return local_status_tensor
    width = (n + world - 1) // world
    pad = torch.full((width,), 255, dtype=local_status_tensor.dtype, device=local_status_tensor.device)
    pad[: local_status_tensor.numel()] = local_status_tensor
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = []
    for r in range(world):
        lo, hi = shard_range(n, r, world)
        out.append(parts[r][: hi - lo])
    return torch.cat(out)


def scatter_rows(full, n, row_bytes, rank, world, dist, device=None, src=0):
    """Input distribution for a batch that is resident on rank `src`: every rank receives its own
    contiguous shard of the (n, row_bytes) uint8 array, each shard crossing one link once (direct
    scatter; a broadcast of the whole array would move `world` times the bytes and is ring/per-link
    bound on xGMI).  `full` is only read on rank `src`.  Equal shards are sent straight out of `full`
    (views, no staging copy); ragged shards are padded to the largest."""
    import torch
    lo, hi = shard_range(n, rank, world)
    if dist is None:        # no process group; with one the collective runs even at world size 1 (RCCL rehearsal)
        return full[lo:hi]
    width = (n + world - 1) // world
    dev = device if device is not None else (full.device if full is not None else "cpu")
    out = torch.empty((width, row_bytes), dtype=torch.uint8, device=dev)
    chunks = None
    if rank == src:
        if n % world == 0:
            chunks = [full[r * width:(r + 1) * width] for r in range(world)]
        else:
            chunks = []
            for r in range(world):
                a, b = shard_range(n, r, world)
                c = torch.zeros((width, row_bytes), dtype=torch.uint8, device=dev)
                c[: b - a] = full[a:b]
                chunks.append(c)
    dist.scatter(out, chunks, src=src)
    return out[: hi - lo]


def broadcast_rows(full, n, row_bytes, rank, world, dist, device=None, src=0):
    """The simpler distribution BASELINE.json's north_star names: broadcast the WHOLE (n, row_bytes) array
    from rank `src`, every rank then slices its own contiguous shard.  Moves `world` times the bytes of
    scatter_rows; kept as an option and timed beside it."""
    import torch
    lo, hi = shard_range(n, rank, world)
    if dist is None:
        return full[lo:hi]
    dev = device if device is not None else (full.device if full is not None else "cpu")
    buf = full if rank == src else torch.empty((n, row_bytes), dtype=torch.uint8, device=dev)
    dist.broadcast(buf, src=src)
    return buf[lo:hi]


MSM_PARTIAL_WORDS = 24   # SSA_MSM_PARTIAL_WORDS of include/schnorr_sig_amd.h


def gather_msm_partials(local_record, world, dist):
    """All-gather of the ranks' MSM-form shard records (int64[24] each, include/schnorr_sig_amd.h:
    ssa_verify_batch_msm_partial_device): 24 words per rank are the only traffic of the MSM-form verify_batch across
    processes (reference src/batch.rs:98-129: every shard contributes one point and one scalar).  Returns the
    (world, 24) tensor on every rank, in rank order; dist=None: the single record as a (1, 24) tensor."""
    import torch
    rec = local_record.reshape(MSM_PARTIAL_WORDS)
    if dist is None:
        return rec.reshape(1, MSM_PARTIAL_WORDS).clone()
    out = torch.empty(world * MSM_PARTIAL_WORDS, dtype=rec.dtype, device=rec.device)   # flat: gloo insists
    dist.all_gather_into_tensor(out, rec.contiguous())
    return out.reshape(world, MSM_PARTIAL_WORDS)


def msm_verdict(local_record, world, dist, combine, engine=None):
    """verify_batch's single verdict for a batch sharded over `world` ranks in the MSM form: gather the shard records,
    then `combine(records)` -- ssa_msm_combine[_device] on this rank's context: one point addition per shard, [sum]G,
    x-only compare -- on EVERY rank (each holds all the records, so no broadcast of the verdict is needed and the ranks
    cannot disagree: the combination is deterministic).  Returns (verdict, records).

    ORDERING.  ssa_verify_batch_msm_partial_device only ENQUEUES the record on the engine's stream, and the gather
    runs on torch's current stream: pass `engine` (the Engine that produced `local_record`) and the two are ordered
    here -- torch's stream waits for the engine before the gather (ssa_ctx_stream_release), the engine's stream waits
    for the gather before `combine` (ssa_ctx_stream_acquire), no host synchronisation.  With engine=None the caller
    vouches that the record is complete (Engine.sync(), or one shared stream through Engine.set_stream).  A record the
    kernel has not written yet is NOT mistaken for an empty shard: every record carries a magic word and the
    combination returns MALFORMED without it (include/schnorr_sig_amd.h)."""
    on_device = engine is not None and getattr(local_record, "is_cuda", False)
    if on_device:
        import torch
        engine.stream_release(torch.cuda.current_stream(local_record.device).cuda_stream)
    elif engine is not None:
        engine.sync()
    records = gather_msm_partials(local_record, world, dist)
    if on_device:
        engine.stream_acquire(torch.cuda.current_stream(local_record.device).cuda_stream)
    verdict = combine(records)
    if on_device:
        # torch's stream waits for the combination too: `records` (and the verdict buffer) may be released to torch's
        # caching allocator -- which reuses a block for later work on ITS stream -- while the engine's kernel still reads
        engine.stream_release(torch.cuda.current_stream(local_record.device).cuda_stream)
    return verdict, records
