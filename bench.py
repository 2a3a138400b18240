#!/usr/bin/env python3
"""bench.py -- Schnorr verifications/sec on MI355X (BASELINE.json metric).

A step = one pass of the hot path (hash_message -> [h]P + [e]G -> x-compare -> wave-ballot
aggregate) over one batch of synthetic signatures per GPU, inputs resident in HBM.
Workload: SURVEY.md §8(d) config 3 (random keypairs, one signature each, distinct 80-byte
messages), generated on the GPU by the engine's own keygen/sign kernel.  Semantics of the
headline number: verify_batch (src/batch.rs: no torsion check, R decompressed with the flag byte of
sig.x -- flags = SSA_FLAG_SIG_FLAG_BYTE, exactly what ssa_verify_batch sets); the Signature::verify
number (flags = SSA_FLAG_CHECK_TORSION: the [q]P subgroup check, src/signature.rs:182, flag byte
ignored) is reported beside it.  The CPU leg runs the headline's flags.

Multi-GPU: one process per GPU over RCCL.  The batch shards by signature, no data-path
collective; the only exchange is one 8-byte all-reduce of the rejection counts per step.
  * under torchrun (WORLD_SIZE set, as the driver launches it) this process is one rank;
  * `python bench.py --gpus N` with N > 1 and no WORLD_SIZE: this process only LAUNCHES -- it starts
    `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child before any GPU call
    and exits with the child's status (it never re-execs and never touches the GPU itself).
Modes:
  default          weak scaling: 2^20 signatures per GPU (the metric's batch size on every GPU)
  --total T        strong scaling (config 4: T = 4194304): ONE batch of T signatures generated on
                   rank 0, distributed by direct scatter (timed: scatter_ms) and -- for comparison,
                   SURVEY.md §8(e) "report both" -- by broadcast of the whole arrays (broadcast_ms),
                   each rank verifying its contiguous shard
With N > 1 the weak run also carries a short config-4 leg (`config4_strong`) so that one driver
invocation records both.
  --force-dist     (or SSA_BENCH_FORCE_DIST=1) the process-group path at ANY world size, N = 1 included: a one-rank
                   `nccl` group is initialised and every collective of the N > 1 run (scatter, broadcast, all-reduce,
                   all-gather of the MSM records, max-over-ranks, the config-4 leg) executes through RCCL on device
                   tensors -- the rehearsal of the multi-GPU code on a one-GPU box (tests/test_gpu_round3.py)
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# ---- algorithmic work per verification (DESIGN.md "Work formula"), in Fp-mul = one 64x64->128
# product with its share of a Goldilocks reduction ----
F6_MUL, F6_SQR = 36, 21
W_DBL = 1 * F6_MUL + 8 * F6_SQR                 # dbl-2007-bl, a = 1
W_MADD = 7 * F6_MUL + 4 * F6_SQR                # mixed addition
W_ADD = 11 * F6_MUL + 5 * F6_SQR                # general Jacobian addition
W_LADDER = 252 * W_DBL + 63 * W_MADD            # 64 signed 4-bit windows (top one is a table pick)
W_INV = 4 * F6_MUL + 6 + 72 + 6                 # Fp6 inverse through the norm + one Fp inverse
W_TABLE = 4 * W_DBL + 3 * W_MADD + 18 * F6_MUL + W_INV + 7 * (3 * F6_MUL + F6_SQR)  # 2P..8P, normalised
W_BASE = 16 * W_MADD                            # comb, 16-bit windows (round 4 runs 11 additions over 24-bit windows)
W_FINAL = F6_MUL + F6_SQR + 2 * F6_SQR + F6_MUL  # x*Z^2 compare + on-curve check
W_VERIFY_KERNEL = W_TABLE + W_LADDER + W_BASE + W_FINAL
W_VERIFY_KEYED = W_LADDER + W_BASE + F6_MUL + F6_SQR   # keyed context: table and key checks are cached
W_VERIFY_KEYED_COMB = (32 + 16) * W_MADD + F6_MUL + F6_SQR   # per-key comb: no doublings at all (textbook count of round 2:
                                                             # 8-bit key windows + 16-bit comb; round 4 executes 16 + 11 additions)
W_TORSION = W_LADDER
# what the kernel EXECUTES ("The ladder window as one statement"): the re-arranged doubling runs 234 products
# (M = ZZ^2 + 3 X^2 as one 42-product accumulation), the mixed addition 330, the per-lane table is built in affine
# coordinates with shared inversions; the comb additions and the final compare are unchanged
W_DBL_EXEC, W_MADD_EXEC = 234, 330
# round 4: 16-entry table (1P..16P: four shared inversions, 65 M + 23 S) and signed 5-bit windows: 250 doublings + 51 additions
W_TABLE_EXEC = 4 * W_INV + 65 * F6_MUL + 23 * F6_SQR
W_VERIFY_EXECUTED = W_TABLE_EXEC + 250 * W_DBL_EXEC + 51 * W_MADD_EXEC + 11 * W_MADD_EXEC + W_FINAL   # 24-bit comb: 11 additions
W_HASH = 4 * 7 * (12 * 4 + 12 * 72 + 2 * 144)   # 4 permutations x 7 rounds (80-byte message)
BYTES_PER_VERIFY = 81 + 96 + 80 + 1             # algorithmic HBM bytes
# Peak of the multiplier, measured in round 2 (tools/isa_probe, DESIGN.md "instruction cost table"): v_mad_u64_u32
# issues once per ~4 cycles per wave64 (4.1-4.7 measured; the class of v_fma_f64 -- 16 lanes per clock), NOT the
# 8 cycles ("quarter rate") round 1 assumed.  Peak = 256 CU x 4 SIMD x 2.4 GHz / 4 cycles x 64 lanes / 4 multiplies
# per 64x64 product, and nothing but multiplies.
PEAK_FPMUL = 256 * 4 * 2.4e9 / 4 * 64 / 4       # 9.83e12 Fp-mul/s
PEAK_FPMUL_R01 = 256 * 4 * 32 * 2.4e9 / 16      # round 1's figure (8 cycles per multiply), kept for continuity
PEAK_HBM_GBPS = 8000.0
LIB = os.environ.get("SSA_LIB") or os.path.join(ROOT, "schnorr-sig_amd", "csrc", "libschnorr_sig_amd.so")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", dest="n", type=int, default=1 << 20, help="signatures per GPU per step (weak mode)")
    ap.add_argument("--total", type=int, default=0,
                    help="strong scaling: ONE batch of this many signatures sharded over the GPUs "
                         "(config 4: 4194304); overrides --batch")
    ap.add_argument("--corrupt", type=float, default=0.0, help="fraction of corrupted signatures (config 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=65536)
    ap.add_argument("--skip-torsion-leg", action="store_true",
                    help="profiling runs: only the timed steps (no torsion / MSM / host-path / keyed legs)")
    ap.add_argument("--no-strong-leg", action="store_true", help="N>1 weak run: skip the config-4 leg")
    ap.add_argument("--strong-total", type=int, default=1 << 22, help="batch size of the config-4 leg")
    ap.add_argument("--force-dist", action="store_true",
                    default=os.environ.get("SSA_BENCH_FORCE_DIST", "") not in ("", "0"),
                    help="initialise the process group and run every collective even at world size 1")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="rank/launch/collective plumbing with NO GPU work (CPU test of the N>1 launch path; "
                         "prints value 0 and \"plumbing_only\": true -- never a measurement)")
    return ap.parse_args(argv)


def pow2_label(n):
    """2^k for powers of two, the plain number otherwise (the workload label follows the batch actually run)"""
    return "2^%d" % (n.bit_length() - 1) if n > 0 and n & (n - 1) == 0 else str(n)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args):
    """`python bench.py --gpus N` (N > 1, no WORLD_SIZE): start N ranks as a CHILD torchrun and exit with
    its status.  Nothing here initialises the GPU (device_count() does not on this image), and nothing is
    exec'ed: the children are fresh processes."""
    backend = os.environ.get("SSA_BENCH_BACKEND", "gloo" if args.plumbing_only else "nccl")
    if backend == "nccl":
        import torch
        ndev = torch.cuda.device_count()
        if ndev < args.gpus:
            sys.stderr.write("bench.py: --gpus %d needs %d devices, this node has %d (%d ranks, %d device%s): "
                             "refusing to run ranks that would share a GPU\n"
                             % (args.gpus, args.gpus, ndev, args.gpus, ndev, "" if ndev == 1 else "s"))
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["SSA_BENCH_BACKEND"] = backend
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def pipelined_leg(torch, ssa, dev_index, eng, sigs, pks, msgs, n, reps=4):
    """batches of n signatures, each split over two contexts of the same device (streams of their own), verify_batch flags"""
    eng2 = ssa.Engine(dev_index)
    dev = sigs.device
    status = torch.empty(n, dtype=torch.uint8, device=dev)
    nfail = torch.zeros(2, dtype=torch.int64, device=dev)
    h = ((n // 2) // 256) * 256
    parts = ((eng2, 0, h), (eng, h, n - h))

    def batch():
        for e, lo, m in parts:
            e.verify_many_device(sigs[lo:].data_ptr(), pks[lo:].data_ptr(), msgs[lo:].data_ptr(), m, 80, status[lo:].data_ptr(),
                                 nfail[0 if lo == 0 else 1:].data_ptr(), check_torsion=False, sig_flag_byte=True)

    torch.cuda.synchronize()
    batch()
    eng.sync()
    eng2.sync()
    nfail.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        batch()
    eng.sync()
    eng2.sync()
    dt = (time.perf_counter() - t0) / reps
    out = {"workload": "%d signatures per batch, halves on two contexts (two streams) of one device, %d batches back to back"
                       % (n, reps),
           "ms_per_batch": dt * 1e3, "verifications_per_sec": n / dt, "rejected": int(nfail.sum().item()),      # (every call SETS its counter: the two halves of the last batch)
           "note": "not the metric: consecutive batches overlap; the metric's steps run one after the other on one stream"}
    del eng2
    return out


def lib_sha256():
    h = hashlib.sha256()
    with open(LIB, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def pmc_for_loaded_library(n):
    """HBM traffic / VALU counters from the committed PMC passes -- only if they were measured on the very
    library this process has loaded (profiles/*/hbm_traffic.json carries its sha256) and at this batch size."""
    sha = lib_sha256()
    best = None
    prof = os.path.join(ROOT, "profiles")
    for d in sorted(os.listdir(prof)) if os.path.isdir(prof) else []:
        p = os.path.join(prof, d, "hbm_traffic.json")
        if os.path.exists(p):
            try:
                j = json.load(open(p))
            except Exception:
                continue
            if j.get("lib_sha256") == sha and int(j.get("batch", 0)) == n:
                best = (p, j)
    if best is None:
        return None, "no profiles/*/hbm_traffic.json measured on the loaded library (sha256 %s...) at n=%d" % (sha[:12], n)
    return best[1], os.path.relpath(best[0], ROOT)


class Ranks:
    """rank / device / process-group plumbing of one bench process"""

    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.backend = os.environ.get("SSA_BENCH_BACKEND", "gloo" if args.plumbing_only else "nccl")
        self.dist = None
        if self.world != args.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with `python bench.py --gpus N` or "
                             "`torchrun --nproc-per-node N bench.py --gpus N`" % (args.gpus, self.world))
        import torch
        self.torch = torch
        self.dev_index = None
        if not args.plumbing_only:
            ndev = torch.cuda.device_count()
            if self.backend == "nccl":
                if self.local_rank >= ndev:
                    raise SystemExit("bench.py: rank %d has no device of its own (%d ranks, %d devices)"
                                     % (self.rank, self.world, ndev))
                self.dev_index = self.local_rank
            else:       # rehearsal backend on a box with fewer GPUs than ranks: ranks share devices
                self.dev_index = self.local_rank % max(ndev, 1)
        # the device is selected BEFORE the process group exists: RCCL binds its communicator to the current device
        if self.dev_index is not None:
            torch.cuda.set_device(self.dev_index)
            self.dev = torch.device("cuda", self.dev_index)
        else:
            self.dev = torch.device("cpu")
        self.forced = bool(args.force_dist) and self.world == 1
        if self.world > 1 or self.forced:
            import torch.distributed as dist
            if self.forced:      # no launcher set the rendezvous up: a one-rank group on the loopback
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", str(free_port()))
                os.environ.setdefault("RANK", "0")
                os.environ.setdefault("WORLD_SIZE", "1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world,
                                        device_id=torch.device("cuda", self.dev_index))
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)
            assert dist.get_world_size() == args.gpus and dist.get_rank() == self.rank
            self.dist = dist

    @property
    def on_device_collectives(self):
        return self.backend == "nccl"

    def all_reduce_sum(self, t):
        if self.dist is None:
            return t
        if self.on_device_collectives or t.device.type == "cpu":
            self.dist.all_reduce(t)
        else:                     # rehearsal backend: through the host
            h = t.cpu()
            self.dist.all_reduce(h)
            t.copy_(h)
        return t

    def max_over_ranks(self, x):
        if self.dist is None:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev if self.on_device_collectives else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def msm_verdict(self, record, combine, engine):
        """MSM-form verdict over the ranks (schnorr_sig_amd.sharding.msm_verdict): all-gather of the 24-word shard
        records on the device (RCCL), or through the host for the rehearsal backend.  `engine` produced `record`:
        msm_verdict orders its stream against the collective's"""
        from schnorr_sig_amd.sharding import msm_verdict
        if self.on_device_collectives:
            return msm_verdict(record, self.world, self.dist, combine, engine=engine)[0]
        engine.sync()
        _, recs = msm_verdict(record.cpu(), self.world, self.dist, lambda r: None)
        dev_recs = recs.to(record.device)
        engine.stream_acquire(self.torch.cuda.current_stream().cuda_stream)
        return combine(dev_recs)

    def sync_all(self):
        if self.dist is not None:
            self.dist.barrier()
        if self.dev.type == "cuda":
            self.torch.cuda.synchronize()

    def device_report(self):
        """[(rank, device index, device name, pci bus id)] gathered on every rank"""
        torch = self.torch
        if self.dev.type == "cuda":
            p = torch.cuda.get_device_properties(self.dev)
            mine = (self.rank, self.dev_index, p.name, getattr(p, "pci_bus_id", None), getattr(p, "uuid", None) and str(p.uuid))
        else:
            mine = (self.rank, None, "cpu (plumbing only)", None, None)
        if self.dist is None:
            return [mine]
        out = [None] * self.world
        self.dist.all_gather_object(out, mine)
        return out

    def finish(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def plumbing_only(args, rk):
    """N>1 launch path without GPU work (CPU test): same barrier / max-over-ranks / all-reduce sequence."""
    torch = rk.torch
    nfail = torch.zeros(1, dtype=torch.int64)
    rk.sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        nfail.fill_(rk.rank + 1)
        rk.all_reduce_sum(nfail)
    rk.sync_all()
    elapsed = rk.max_over_ranks(time.perf_counter() - t0)
    devices = rk.device_report()
    if rk.rank == 0:
        emit(json.dumps({"metric": "plumbing only (no GPU work, not a measurement)", "value": 0.0,
                          "unit": "verifications/s", "n_gpus": rk.world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "higher_is_better": True,
                          "scaling": "weak", "plumbing_only": True, "backend": rk.backend,
                          "all_reduce_sum": int(nfail.item()), "ranks": [list(d) for d in devices]}))
    rk.finish()


def gen_batch(torch, eng, dev, n, seed, chunk=1 << 20):
    """config-3 inputs on the device: (sigs, pks, msgs) for n random keypairs, 80-byte distinct messages"""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    pks = torch.empty((n, 96), dtype=torch.uint8, device=dev)
    sigs = torch.empty((n, 81), dtype=torch.uint8, device=dev)
    msgs = torch.randint(0, 256, (n, 80), dtype=torch.uint8, device=dev, generator=g)
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        sks = torch.randint(0, 256, (m, 32), dtype=torch.uint8, device=dev, generator=g)
        nonces = torch.randint(0, 256, (m, 32), dtype=torch.uint8, device=dev, generator=g)
        sks[:, 31] &= 0x3F      # < 2^254 < q
        nonces[:, 31] &= 0x3F
        sks[:, 0] |= 1          # never zero
        nonces[:, 0] |= 1
        eng.keygen_sign_many_device(sks.data_ptr(), nonces.data_ptr(), msgs[lo:lo + m].data_ptr(), m, 80,
                                    pks[lo:lo + m].data_ptr(), sigs[lo:lo + m].data_ptr())
        eng.sync()
    return sigs, pks, msgs, g


def distribute(rk, full, total, how):
    """rank-0-resident (sigs, pks, msgs) -> this rank's contiguous shard; returns (shard tensors, milliseconds).
    how = "scatter": every shard crosses one link once; "broadcast": the whole arrays go to every rank."""
    from schnorr_sig_amd.sharding import broadcast_rows, scatter_rows, shard_range
    torch = rk.torch
    lo, hi = shard_range(total, rk.rank, rk.world)
    widths = (81, 96, 80)
    rk.sync_all()
    t0 = time.perf_counter()
    out = []
    for t_, w in zip(full, widths):
        if rk.on_device_collectives:
            fn = scatter_rows if how == "scatter" else broadcast_rows
            out.append(fn(t_, total, w, rk.rank, rk.world, rk.dist, device=rk.dev))
        else:       # rehearsal backend: through the host
            fn = scatter_rows if how == "scatter" else broadcast_rows
            got = fn(t_.cpu() if t_ is not None else None, total, w, rk.rank, rk.world, rk.dist, device="cpu")
            out.append(got.to(rk.dev))
    rk.sync_all()
    ms = rk.max_over_ranks(time.perf_counter() - t0) * 1e3
    assert all(o.shape[0] == hi - lo for o in out)
    return out, ms


_REAL_STDOUT = None


def quiet_stdout():
    """The contract is ONE JSON line on stdout.  Libraries below us write there too (RCCL prints a five-line version
    banner at the first communicator, gloo its rank chatter): from here on file descriptor 1 points at stderr, and only
    emit() writes to the real stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    sys.stdout.flush()
    data = (line + "\n").encode()
    fd = _REAL_STDOUT if _REAL_STDOUT is not None else 1
    while data:
        data = data[os.write(fd, data):]


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))      # (the ranks inherit the real stdout)
    quiet_stdout()
    # read by the HSA runtime when it initialises: must be in the environment before the first torch.cuda call
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    rk = Ranks(args)
    if args.plumbing_only:
        return plumbing_only(args, rk)

    import numpy as np
    torch = rk.torch
    import schnorr_sig_amd as ssa
    from schnorr_sig_amd.sharding import shard_range
    rank, world, dev = rk.rank, rk.world, rk.dev

    eng = ssa.Engine(rk.dev_index)
    # one explicit stream for everything: the engine's kernels, torch's fills and the collectives
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    eng.set_stream(stream.cuda_stream)

    strong = args.total > 0
    scatter_ms = broadcast_ms = None
    if strong:
        total = args.total
        full = (None, None, None)
        if rank == 0:
            s_, p_, m_, g = gen_batch(torch, eng, dev, total, 0x5C4E0224)
            full = (s_, p_, m_)
        if world > 1:
            (sigs_b, pks_b, msgs_b), broadcast_ms = distribute(rk, full, total, "broadcast")
            del sigs_b, pks_b, msgs_b
            (sigs, pks, msgs), scatter_ms = distribute(rk, full, total, "scatter")
            sigs, pks, msgs = sigs.contiguous(), pks.contiguous(), msgs.contiguous()
        else:
            sigs, pks, msgs = full
        del full
        lo, hi = shard_range(total, rank, world)
        n = hi - lo
        g = torch.Generator(device=dev)
        g.manual_seed(0x5C4E0224 + 7 * rank + 1)
        n_all = total
    else:
        n = args.n
        sigs, pks, msgs, g = gen_batch(torch, eng, dev, n, 0x5C4E0222 + rank)
        n_all = n * world

    n_bad_expected = 0
    if args.corrupt > 0:
        n_bad_expected = int(n * args.corrupt)
        idx = torch.randperm(n, device=dev, generator=g)[:n_bad_expected]
        third = n_bad_expected // 3
        sigs[idx[:third], 49] ^= 1                       # e bit flip
        msgs[idx[third:2 * third], 40] ^= 0x10           # message bit flip
        rest = idx[2 * third:]
        sigs[rest, :49] = sigs[(rest + 1) % n, :49]      # someone else's R (canonical, on curve)
    status = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
    nfail = torch.zeros(1, dtype=torch.int64, device=dev)

    # The timed step runs exactly what ssa_verify_batch runs per signature (include/schnorr_sig_amd.h; reference
    # src/batch.rs:56-130): no subgroup check, and R is the point the flag byte of sig.x selects (:104) --
    # flags = SSA_FLAG_SIG_FLAG_BYTE.  The Signature::verify leg below runs flags = SSA_FLAG_CHECK_TORSION
    # (src/signature.rs:181-205: subgroup check, flag byte ignored).
    def step(check_torsion=False):
        eng.verify_many_device(sigs.data_ptr(), pks.data_ptr(), msgs.data_ptr(), n, 80, status.data_ptr(),
                               nfail.data_ptr(), check_torsion=check_torsion, sig_flag_byte=not check_torsion)
        rk.all_reduce_sum(nfail)     # aggregate verdict of the sharded batch (RCCL, 8 bytes)

    for _ in range(args.warmup):
        step()
    rk.sync_all()
    eng.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    rk.sync_all()
    elapsed = time.perf_counter() - t0
    eng.enable_timing(False)
    k_verify_ms, k_cnt = eng.read_timing("ssa_k_verify")
    k_hash_ms, _ = eng.read_timing("ssa_k_hash")
    total_fail = int(nfail.item())
    status_batch_semantics = status[:min(n, args.cpu_sample)].clone()   # what the CPU leg is compared with
    bad_total = torch.tensor([n_bad_expected], dtype=torch.int64, device=dev)
    rk.all_reduce_sum(bad_total)
    ok = total_fail == int(bad_total.item())
    elapsed = rk.max_over_ranks(elapsed)
    devices = rk.device_report()
    legs = not args.skip_torsion_leg

    # ---- Signature::verify semantics (torsion check on), 2 steps ----
    torsion_rate, total_fail_t = None, None
    if legs:
        step(check_torsion=True)
        rk.sync_all()
        t2 = time.perf_counter()
        for _ in range(2):
            step(check_torsion=True)
        rk.sync_all()
        torsion_rate = n_all * 2 / rk.max_over_ranks(time.perf_counter() - t2)
        total_fail_t = int(nfail.item())

    # ---- the reference's own MSM-form verify_batch (one verdict per batch), 3 steps ----
    # one process: ssa_verify_batch_msm_device.  Several ranks: every rank reduces its shard to one 24-word record
    # (ssa_verify_batch_msm_partial_device), the records are all-gathered (24 words per rank: the only traffic) and
    # combined on every rank (ssa_msm_combine_device: one point addition per shard, [sum]G, x-only compare) --
    # SURVEY.md 8(e), reference src/batch.rs:98-129.  A rank with an empty shard still takes part in the collective.
    msm = None
    if legs and (n > 0 or rk.dist is not None):
        coeffs = torch.randint(0, 256, (max(n, 1), 16), dtype=torch.uint8, device=dev, generator=g)
        verdict = torch.full((1,), 255, dtype=torch.int32, device=dev)
        record = torch.zeros(24, dtype=torch.int64, device=dev)

        def combine(records):
            eng.msm_combine_device(records.data_ptr(), records.shape[0], verdict.data_ptr())
            return verdict

        def msm_step():
            if rk.dist is None:
                eng.verify_batch_msm_device(sigs.data_ptr(), pks.data_ptr(), msgs.data_ptr(), n, 80, coeffs.data_ptr(),
                                            16, verdict.data_ptr())
            else:
                eng.verify_batch_msm_partial_device(sigs.data_ptr(), pks.data_ptr(), msgs.data_ptr(), n, 80,
                                                    coeffs.data_ptr(), 16, record.data_ptr())
                rk.msm_verdict(record, combine, eng)
        msm_step()
        rk.sync_all()
        eng.enable_timing(True)
        t3 = time.perf_counter()
        for _ in range(3):
            msm_step()
        rk.sync_all()
        dt = rk.max_over_ranks(time.perf_counter() - t3)
        eng.enable_timing(False)
        stages = {k: eng.read_timing(k)[0] for k in ("ssa_k_hash", "msm_k_prepare", "msm_sort", "msm_k_buckets",
                                                     "msm_reduce", "msm_combine")}
        msm = {"verifications_per_sec": n_all * 3 / dt, "ms_per_batch": dt / 3 * 1e3,
               "verdict": int(verdict.item()), "expected_verdict": 2 if args.corrupt > 0 else 0, "stages_ms": stages,
               "combined_over_ranks": rk.dist is not None,
               "note": ("ONE verdict for the %d signatures of all %d rank(s): per-rank shard records all-gathered "
                        "(24 words each) and combined on every rank" % (n_all, world)) if rk.dist is not None else
                       "one verdict for this process's batch"}

    # ---- host-buffer entry point (what the Rust shim binds): PCIe-inclusive, never `value` ----
    # (the legs after the timed region are reported beside the metric: a failure inside one of them is recorded in its
    # field and must not take the headline line with it)
    def guarded(fn, *a, **kw):
        try:
            return fn(*a, **kw)
        except Exception as exc:          # noqa: BLE001 -- reported, not swallowed
            return {"error": "%s: %s" % (type(exc).__name__, exc)}

    host_path = None
    if legs and rank == 0 and n > 0 and hasattr(eng, "host_path_probe"):
        host_path = guarded(eng.host_path_probe, sigs, pks, msgs, n, reps=3)

    # ---- keyed context: repeated public keys (validator sets), Signature::verify semantics ----
    keyed = None
    if legs and rank == 0 and n > 0 and hasattr(eng, "keyset_create"):
        keyed = guarded(keyed_leg, torch, eng, dev, g, min(n, 1 << 20))

    # ---- the signing side: keygen + sign per second, throughput and constant-time signers ----
    signing = None
    if legs and rank == 0 and n > 0:
        signing = guarded(signing_leg, torch, eng, dev, g, min(n, 1 << 18))

    # ---- what a caller that pipelines batches gets: TWO contexts of the device (own streams, one comb table), each over
    # half of every batch, issued back to back -- the halves' kernels fill each other's tails (the last wave of every SIMD
    # finishes alone: profiles/r04/wave_timeline.txt).  Reported beside the metric, never as `value`. ----
    pipelined = None
    if legs and rank == 0 and n >= (1 << 16):
        pipelined = guarded(pipelined_leg, torch, ssa, rk.dev_index, eng, sigs, pks, msgs, n)

    # ---- config 4 beside the weak run: one 2^22 batch, rank 0 -> shards, verified once per step ----
    config4 = None
    if legs and rk.dist is not None and not strong and not args.no_strong_leg:
        config4 = guarded(strong_leg, rk, eng, args.strong_total)

    if rank == 0:
        value = n_all * args.steps / elapsed
        w_kernel = W_VERIFY_KERNEL
        achieved = w_kernel * n / (k_verify_ms * 1e-3) if k_verify_ms > 0 else 0.0
        pmc, pmc_src = pmc_for_loaded_library(n)
        traffic = pmc["ssa_k_verify"]["hbm_bytes_per_launch"] if pmc else None
        metric = "Schnorr verifications/sec, 2^20-sig batch, 1/2/4/8 MI355X; bit-exact vs CPU"
        try:
            with open(os.path.join(ROOT, "BASELINE.json")) as fh:
                metric = json.load(fh)["metric"]
        except Exception:
            pass
        unpinned = eng.uses_default_params() if hasattr(eng, "uses_default_params") else True
        workload = ("config4: ONE batch of %s random-keypair signatures sharded over %d GPU(s)" % (pow2_label(n_all), world)
                    if strong else "%s: %s random-keypair signatures per GPU%s"
                    % ("config5" if args.corrupt > 0 else "config3", pow2_label(n),
                       ", %g %% corrupted" % (100 * args.corrupt) if args.corrupt > 0 else "")) + \
            ", 80-byte distinct messages, full verify (Rescue hash + [h]P+[e]G + x-compare), verify_batch semantics"
        out = {
            "metric": metric,
            "value": value,
            "unit": "verifications/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": workload, "signatures_per_gpu": n, "signatures_total": n_all, "message_bytes": 80,
                       "flags": {"word": ssa.FLAG_SIG_FLAG_BYTE, "names": ["SSA_FLAG_SIG_FLAG_BYTE"],
                                 "meaning": "what ssa_verify_batch sets: no subgroup check, byte 48 of the signature "
                                            "honoured (src/batch.rs:104); the CPU leg runs the same flags"},
                       "parallelism": "shard%d" % world, "corrupt_fraction": args.corrupt,
                       # what the context holds on the device (ssa_ctx_info, read AFTER the run): the comb for G is a
                       # speed-for-memory choice of the context (ssa_ctx_create_ex), the workspaces are per slice
                       "device_memory": (lambda i: {"gtab_window_bits": i["gtab_bits"], "gtab_windows": i["gtab_windows"],
                                                    "gtab_bytes": i["gtab_bytes"], "workspace_bytes": i["workspace_bytes"],
                                                    "lane_slice": i["lane_slice"], "msm_slice": i["msm_slice"],
                                                    "hbm_budget_bytes": i["hbm_budget_bytes"],
                                                    "slices_on_two_streams": i["two_streams"]})(eng.info())
                       if hasattr(eng, "info") else None,
                       "backend": rk.backend if rk.dist is not None else None,
                       "process_group": ("forced one-rank group (RCCL rehearsal)" if rk.forced else "torchrun ranks")
                       if rk.dist is not None else None},
            "ranks": [list(d) for d in devices],
            "hip_runtime": getattr(ssa, "HIP_RUNTIME_BOUND", None),
            "constants": "builder-default (unpinned)" if unpinned else "caller-supplied blob",
            "parity_unpinned": bool(unpinned),
            "parity_note": "bit-exact against the CPU restatement (oracle/); Rescue constants and generator are not "
                           "upstream's until tools/blob_from_upstream.py + tests/golden/upstream_vectors.json close it",
            "all_verdicts_as_expected": bool(ok),
            "rejected": total_fail,
            "with_torsion_check_verifications_per_sec": torsion_rate,
            "with_torsion_check_rejected": total_fail_t,
            "kernels_ms": {"ssa_k_verify": k_verify_ms, "ssa_k_hash": k_hash_ms, "launches": k_cnt},
            "verify_batch_msm_form": msm,
            "host_path": host_path,
            "keyed_context": keyed,
            "signing": signing,
            "pipelined_two_contexts": pipelined,
            "scatter_ms": scatter_ms,
            "broadcast_ms": broadcast_ms,
            "config4_strong": config4,
            "roofline": {
                "bound": "valu",
                "bound_note": "64-bit integer VALU issue (v_mad_u64_u32), neither hbm nor mfma: SURVEY.md 8(d); peak = "
                              "256 CU x 4 SIMD x 2.4 GHz / 4 cycles per wave64 multiply (measured 4.1-4.7, tools/isa_probe) "
                              "x 64 lanes / 4 multiplies per 64x64 product -- TWICE round 1's peak, which assumed a "
                              "quarter-rate multiplier; frac_r01_basis divides by the old peak for continuity",
                "kernel": "ssa_k_verify",
                "achieved": achieved / 1e9,
                "peak": PEAK_FPMUL / 1e9,
                "unit": "GFp-mul/s",
                "frac": achieved / PEAK_FPMUL,
                "frac_r01_basis": achieved / PEAK_FPMUL_R01,
                "work_per_unit": w_kernel,
                "traffic": traffic,
                "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE)",
                "traffic_source": pmc_src,
                "algorithmic_bytes_per_launch": BYTES_PER_VERIFY * n,
                "work_executed": W_VERIFY_EXECUTED,
                "work_executed_note": "64x64 products the kernel really runs per verification (window statement: 234 per "
                                      "doubling, 330 per mixed addition; 250 doublings + 51 additions over a 16-entry affine "
                                      "table since round 4) -- work_per_unit is the textbook count of round 1's algorithm, "
                                      "kept fixed across rounds, DESIGN.md 'Work formula'",
                "achieved_executed": (W_VERIFY_EXECUTED * n / (k_verify_ms * 1e-3) if k_verify_ms > 0 else 0.0) / 1e9,
                "pmc": {k: pmc["ssa_k_verify"].get(k) for k in ("SQ_INSTS_VALU", "valu_cycles_per_instruction",
                                                                  "valu_active_frac", "valu_instructions_per_product",
                                                                  "duration_ms")} if pmc else None,
            },
            "roofline_hash": {
                "bound": "valu", "kernel": "ssa_k_hash", "work_per_unit": W_HASH,
                "achieved": (W_HASH * n / (k_hash_ms * 1e-3) if k_hash_ms > 0 else 0.0) / 1e9,
                "peak": PEAK_FPMUL / 1e9, "unit": "GFp-mul/s",
                "frac": (W_HASH * n / (k_hash_ms * 1e-3) if k_hash_ms > 0 else 0.0) / PEAK_FPMUL,
            },
            "roofline_hbm": {
                "bound": "hbm",
                "achieved": BYTES_PER_VERIFY * n / (max(k_verify_ms + k_hash_ms, 1e-9) * 1e-3) / 1e9,
                "peak": PEAK_HBM_GBPS,
                "unit": "GB/s",
                "frac": BYTES_PER_VERIFY * n / (max(k_verify_ms + k_hash_ms, 1e-9) * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                "traffic": None,
            },
        }
        try:
            out["measured_fpmul_peak"] = {"fp_mul": eng.bench_fpmul(1), "f6_lazy": eng.bench_fpmul(3),
                                          "unit": "Fp-mul/s"}
        except Exception as exc:  # pragma: no cover
            out["measured_fpmul_peak"] = str(exc)

        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(np, sigs, pks, msgs, status_batch_semantics, min(args.cpu_sample, n))
        elif world > 1:
            out["cpu_baseline"] = None
            out["cpu_baseline_note"] = ("the CPU leg is timed on rank 0 at N = 1 only (the N = 1 line of the same "
                                        "run carries it); an N > 1 line has none by design")
        emit(json.dumps(out))
    rk.finish()


def strong_leg(rk, eng, total):
    """config 4: one batch of `total` signatures lives on rank 0; scatter vs broadcast; 3 timed verification steps"""
    torch = rk.torch
    from schnorr_sig_amd.sharding import shard_range
    full = (None, None, None)
    if rk.rank == 0:
        s_, p_, m_, _ = gen_batch(torch, eng, rk.dev, total, 0x5C4E0224)
        full = (s_, p_, m_)
    (sb, pb, mb), broadcast_ms = distribute(rk, full, total, "broadcast")
    del sb, pb, mb
    (sigs, pks, msgs), scatter_ms = distribute(rk, full, total, "scatter")
    sigs, pks, msgs = sigs.contiguous(), pks.contiguous(), msgs.contiguous()
    del full
    lo, hi = shard_range(total, rk.rank, rk.world)
    n = hi - lo
    status = torch.empty(n, dtype=torch.uint8, device=rk.dev)
    nfail = torch.zeros(1, dtype=torch.int64, device=rk.dev)

    def step():
        eng.verify_many_device(sigs.data_ptr(), pks.data_ptr(), msgs.data_ptr(), n, 80, status.data_ptr(),
                               nfail.data_ptr(), check_torsion=False, sig_flag_byte=True)
        rk.all_reduce_sum(nfail)
    step()
    rk.sync_all()
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    rk.sync_all()
    dt = rk.max_over_ranks(time.perf_counter() - t0) / 3
    return {"workload": "config4: ONE batch of %d signatures on rank 0 -> %d contiguous shards" % (total, rk.world),
            "scaling": "strong", "signatures_total": total, "ms_per_batch": dt * 1e3,
            "verifications_per_sec": total / dt,
            "scatter_ms": scatter_ms, "broadcast_ms": broadcast_ms,
            "verifications_per_sec_including_scatter": total / (dt + scatter_ms * 1e-3),
            "rejected": int(nfail.item())}


def signing_leg(torch, eng, dev, g, n):
    """KeyPair::new + KeyPair::sign for n (sk, nonce, 80-byte message) triples resident in HBM: the throughput signer
    (16-bit comb, variable time) and the constant-time one (SSA_FLAG_SIGN_CT: 4-bit windows, full-table scans, what the
    reference's `&BASEPOINT_TABLE * r` is); the two must emit the same bytes"""
    sks = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    nonces = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    for t in (sks, nonces):
        t[:, 31] &= 0x3F
        t[:, 0] |= 1
    msgs = torch.randint(0, 256, (n, 80), dtype=torch.uint8, device=dev, generator=g)
    out = {"workload": "%d x (keygen + sign), 80-byte messages, inputs resident in HBM" % n}
    res = {}
    for name, ct in (("throughput_signer", False), ("constant_time_signer", True)):
        pks = torch.empty((n, 96), dtype=torch.uint8, device=dev)
        sigs = torch.empty((n, 81), dtype=torch.uint8, device=dev)

        def run():
            eng.keygen_sign_many_device(sks.data_ptr(), nonces.data_ptr(), msgs.data_ptr(), n, 80, pks.data_ptr(),
                                        sigs.data_ptr(), constant_time=ct)
        run()
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(2):
            run()
        eng.sync()
        dt = (time.perf_counter() - t0) / 2
        res[name] = (pks, sigs)
        out[name] = {"signatures_per_sec": n / dt, "ms_per_batch": dt * 1e3}
    out["same_bytes"] = bool((res["throughput_signer"][0] == res["constant_time_signer"][0]).all().item() and
                             (res["throughput_signer"][1] == res["constant_time_signer"][1]).all().item())
    return out


def keyed_leg(torch, eng, dev, g, n, n_keys=64):
    """validator-set workload: n signatures by n_keys signers, through a keyed context (subgroup check and the
    per-key tables done once at ssa_keyset_create); Signature::verify semantics"""
    sks = torch.randint(0, 256, (n_keys, 32), dtype=torch.uint8, device=dev, generator=g)
    sks[:, 31] &= 0x3F
    sks[:, 0] |= 1
    idx = torch.randint(0, n_keys, (n,), dtype=torch.int32, device=dev, generator=g)
    sk_rows = sks[idx.long()].contiguous()
    nonces = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    nonces[:, 31] &= 0x3F
    nonces[:, 0] |= 1
    msgs = torch.randint(0, 256, (n, 80), dtype=torch.uint8, device=dev, generator=g)
    pks = torch.empty((n, 96), dtype=torch.uint8, device=dev)
    sigs = torch.empty((n, 81), dtype=torch.uint8, device=dev)
    eng.keygen_sign_many_device(sk_rows.data_ptr(), nonces.data_ptr(), msgs.data_ptr(), n, 80, pks.data_ptr(),
                                sigs.data_ptr())
    eng.sync()
    first = torch.stack([(idx == k).nonzero()[0, 0] for k in range(n_keys)])
    key_rows = pks[first].contiguous()
    status = torch.empty(n, dtype=torch.uint8, device=dev)
    nfail = torch.zeros(1, dtype=torch.int64, device=dev)
    out = {"workload": "%d signatures by %d signers (keyed context), Signature::verify semantics" % (n, n_keys)}
    for kind, w in (("ladder", W_VERIFY_KEYED), ("comb", W_VERIFY_KEYED_COMB)):
        t0 = time.perf_counter()
        ks = eng.keyset_create_device(key_rows.data_ptr(), n_keys, kind=kind)
        eng.sync()
        create_ms = (time.perf_counter() - t0) * 1e3

        def step():
            eng.verify_many_indexed_device(ks, idx.data_ptr(), sigs.data_ptr(), msgs.data_ptr(), n, 80,
                                           status.data_ptr(), nfail.data_ptr())
        step()
        eng.sync()
        eng.enable_timing(True)
        t0 = time.perf_counter()
        for _ in range(3):
            step()
        eng.sync()
        dt = (time.perf_counter() - t0) / 3
        eng.enable_timing(False)
        k_ms, _ = eng.read_timing("ssa_k_verify_keyed")
        eng.read_timing("ssa_k_hash")
        rej = int(nfail.item())
        eng.keyset_destroy(ks)
        out[kind] = {"verifications_per_sec": n / dt, "ms_per_batch": dt * 1e3, "keyset_create_ms": create_ms,
                     "kernel_ms": k_ms, "rejected": rej, "work_per_unit": w,
                     "roofline_frac": (w * n / (k_ms * 1e-3) / PEAK_FPMUL) if k_ms > 0 else None}
    out["verifications_per_sec"] = max(out["ladder"]["verifications_per_sec"], out["comb"]["verifications_per_sec"])
    return out


def cpu_baseline(np, sigs, pks, msgs, gpu_status_batch_semantics, m):
    """The C restatement (oracle/, test infrastructure) timed on this box's host cores on the first m
    signatures of rank 0's batch -- the same semantics (verify_batch: no torsion check) as the timed GPU steps."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc_mod
    try:
        orc_mod.build(native=True)
        orc = orc_mod.Oracle(native=True)
    except Exception:
        orc = orc_mod.Oracle()
    hs, hp, hm = sigs[:m].cpu().numpy(), pks[:m].cpu().numpy(), msgs[:m].cpu().numpy()
    threads = orc.hw_threads()
    gpu_st = gpu_status_batch_semantics[:m].cpu().numpy()
    # Two CPU forms of the same restatement (oracle/): the PLAIN one is the checker of the test-suite (every product of
    # an Fp6 multiplication reduced on its own, general additions, a naive MSM: a strawman as a baseline); the WINDOWED
    # one (oracle/schnorr_oracle_fast.inc: lazy Fp6 products, width-5 NAF over affine odd multiples, a fixed-base table
    # for G, a bucket MSM) is what a CPU library would run and is the number reported.  Both are timed on the same
    # sample and must agree with each other and with the GPU lane for lane.
    tc = time.perf_counter()
    st_fast = orc.verify_many_fast(hs, hp, hm, check_torsion=False, sig_flag_byte=True, threads=threads)
    dt_fast = time.perf_counter() - tc
    tc = time.perf_counter()
    st = orc.verify_many(hs, hp, hm, check_torsion=False, sig_flag_byte=True, threads=threads)
    dt = time.perf_counter() - tc
    out = {
        "value": m / dt_fast, "unit": "verifications/s", "cores": threads, "kind": "port",
        "algorithms": "windowed: lazy Fp6 products (one reduction per coefficient), width-5 NAF of h over affine odd "
                      "multiples, fixed-base table for G (32 x 255 rows), bucket MSM for the batch form",
        "sample": "first %d signatures of rank 0's batch, C restatement of the reference algorithm "
                  "(oracle/schnorr_oracle.c + schnorr_oracle_fast.inc, -O3 -march=native, OpenMP), verify_batch "
                  "semantics: flags = SSA_FLAG_SIG_FLAG_BYTE on both sides (no subgroup check, flag byte of sig.x "
                  "honoured)" % m,
        "flags": 8,
        "agrees_with_gpu": bool((st_fast == gpu_st).all()),
        "plain_restatement": {"value": m / dt, "unit": "verifications/s", "cores": threads, "kind": "port",
                              "note": "the checker of the test-suite, written for clarity: never quote a ratio against it",
                              "agrees_with_gpu": bool((st == gpu_st).all()),
                              "agrees_with_windowed": bool((st == st_fast).all())},
    }
    # the same restatement the way the reference runs it: one thread (it has no threading), per
    # signature with the subgroup check, and its MSM-form verify_batch
    m1 = min(1024, m)
    co = np.random.default_rng(11).integers(0, 256, size=(m1, 32), dtype=np.uint8)
    co[:, 16:] = 0
    one = {}
    for name, vm, vb in (("windowed", orc.verify_many_fast, orc.verify_batch_msm_fast),
                         ("plain", orc.verify_many, orc.verify_batch_msm)):
        tc = time.perf_counter()
        st1 = vm(hs[:m1], hp[:m1], hm[:m1], check_torsion=True, threads=1)
        t_one = time.perf_counter() - tc
        tc = time.perf_counter()
        verdict_cpu = vb(hs[:m1], hp[:m1], hm[:m1], co, threads=1)
        t_msm = time.perf_counter() - tc
        one[name] = {"signature_verify_per_sec": m1 / t_one, "verify_batch_msm_form_signatures_per_sec": m1 / t_msm,
                     "verify_batch_msm_form_verdict": verdict_cpu, "rejected_with_subgroup_check": int((st1 != 0).sum())}
    out["single_thread"] = dict(one["windowed"], sample="first %d signatures, 1 thread" % m1, plain_restatement=one["plain"])
    # the MSM-form verify_batch on all hardware threads too (windowed: a bucket MSM, windows in
    # parallel; plain: the naive sum of n double-scalar products, parallel over signatures)
    co_all = np.random.default_rng(12).integers(0, 256, size=(m, 32), dtype=np.uint8)
    co_all[:, 16:] = 0
    tc = time.perf_counter()
    verdict_fast = orc.verify_batch_msm_fast(hs, hp, hm, co_all, threads=threads)
    t_msm_fast = time.perf_counter() - tc
    tc = time.perf_counter()
    verdict_all = orc.verify_batch_msm(hs, hp, hm, co_all, threads=threads)
    t_msm_all = time.perf_counter() - tc
    out["all_threads_msm_form"] = {
        "verify_batch_msm_form_signatures_per_sec": m / t_msm_fast, "verify_batch_msm_form_verdict": verdict_fast,
        "cores": threads, "sample": "first %d signatures, %d threads" % (m, threads),
        "plain_restatement": {"verify_batch_msm_form_signatures_per_sec": m / t_msm_all,
                              "verify_batch_msm_form_verdict": verdict_all}}
    return out


if __name__ == "__main__":
    main()
