"""GPU tests added in round 3 (all through the C ABI):
 * the process-group path of bench.py on ONE GPU: a forced one-rank `nccl` group, so that scatter / broadcast /
   all-reduce / all-gather of the MSM records / max-over-ranks run through RCCL on device tensors before the
   driver's multi-GPU node ever sees this code (BASELINE.json configs[3]);
 * the MSM-form verdict across processes through the PUBLIC partial/combine entry points (reference
   src/batch.rs:98-129: one point and one scalar per shard, one addition per shard, one compare);
 * stream ordering and error paths of the pipelined host-buffer upload;
 * a key set that outlives its engine; the debug probes refuse a zero doubling count."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Q = 0x7AF2599B3B3F22D0563FBF0F990A37B5327AA72330157722D443623EAED4ACCF


def make_scalars(rng, n):
    s = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    s[:, 31] &= 0x3F
    s[:, 0] |= 1
    return s


def honest(engine, rng, n, msg_len=80):
    sks, nonces = make_scalars(rng, n), make_scalars(rng, n)
    msgs = rng.integers(0, 256, size=(n, msg_len), dtype=np.uint8)
    pks, sigs = engine.keygen_sign_many(sks, nonces, msgs)
    return sigs, pks, msgs


def coeffs32(rng, n):
    c = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    c[:, 31] &= 0x3F
    return c


# ---------------------------------------------------------------- RCCL under bench.py on one GPU
def _bench(args, timeout=900):
    """bench.py as a FRESH child process (never an exec of this process, which has touched the GPU)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    assert r.stdout.strip() == lines[0], "stdout must hold the JSON line only (RCCL's banner belongs on stderr): " + r.stdout[:300]
    return json.loads(lines[0])


def test_bench_collectives_run_through_rccl_on_one_rank():
    """`--force-dist`: a world-size-1 nccl group; every collective of the N > 1 run executes on device tensors."""
    out = _bench(["--gpus", "1", "--force-dist", "--batch", "65536", "--steps", "2", "--warmup", "1",
                  "--strong-total", "131072", "--cpu-sample", "256", "--corrupt", "0.01"])
    assert out["n_gpus"] == 1 and out["config"]["backend"] == "nccl"
    assert out["config"]["process_group"].startswith("forced one-rank")
    assert out["config"]["workload"].startswith("config5: 2^16 random-keypair signatures per GPU, 1 % corrupted")
    assert out["config"]["signatures_per_gpu"] == 65536
    assert out["all_verdicts_as_expected"] is True and out["rejected"] == 655     # all-reduce of the counts (RCCL)
    c4 = out["config4_strong"]
    assert "error" not in c4, c4
    assert c4["scatter_ms"] is not None and c4["broadcast_ms"] is not None         # dist.scatter / dist.broadcast
    assert c4["signatures_total"] == 131072 and c4["rejected"] == 0
    msm = out["verify_batch_msm_form"]
    assert msm["combined_over_ranks"] is True                                       # all-gather of the 24-word records
    assert msm["verdict"] == msm["expected_verdict"] == 2
    assert msm["stages_ms"]["msm_combine"] > 0
    assert out["cpu_baseline"]["agrees_with_gpu"] is True
    assert out["cpu_baseline"]["all_threads_msm_form"]["verify_batch_msm_form_verdict"] in (0, 2)
    assert out["roofline"]["work_executed"] > out["roofline"]["work_per_unit"] * 0.9


def test_bench_workload_label_follows_the_batch():
    out = _bench(["--gpus", "1", "--batch", "4096", "--steps", "1", "--warmup", "0", "--skip-torsion-leg",
                  "--no-cpu-baseline"])
    assert out["config"]["workload"].startswith("config3: 2^12 random-keypair signatures per GPU,")
    assert out["config"]["backend"] is None and out["all_verdicts_as_expected"] is True


# ---------------------------------------------------------------- MSM verdict across processes: public API
def _shards(n, k):
    from schnorr_sig_amd.sharding import shard_range
    return [shard_range(n, r, k) for r in range(k)]


@pytest.mark.parametrize("n", [2, 3, 64, 1000, 20000])
def test_msm_partial_and_combine_public_api(engine, oracle, n):
    """Three contexts on device 0, each reducing its contiguous shard to one record through
    ssa_verify_batch_msm_partial[_device]; ssa_msm_combine[_device] of the three records must equal
    ssa_verify_batch_msm on the whole batch and the oracle's src/batch.rs restatement -- honest, one corrupted
    signature, an identity key, an undecodable signature."""
    import torch
    import schnorr_sig_amd as ssa
    rng = np.random.default_rng(3000 + n)
    sigs, pks, msgs = honest(engine, rng, n)
    co = coeffs32(rng, n)
    engines = [ssa.Engine(0) for _ in range(3)]
    dev = torch.device("cuda", 0)

    def records_host(sg, inf=None):
        recs = []
        for e, (lo, hi) in zip(engines, _shards(n, 3)):
            recs.append(e.verify_batch_msm_partial(sg[lo:hi], pks[lo:hi], msgs[lo:hi], coeffs=co[lo:hi],
                                                   pk_inf=None if inf is None else inf[lo:hi]))
        return np.stack(recs)

    def records_device(sg):
        recs = torch.zeros((3, 24), dtype=torch.int64, device=dev)
        keep = []
        for r, (e, (lo, hi)) in enumerate(zip(engines, _shards(n, 3))):
            cnt = hi - lo
            ds, dp, dm, dc = (torch.from_numpy(np.ascontiguousarray(a[lo:hi])).to(dev) for a in (sg, pks, msgs, co))
            keep.append((ds, dp, dm, dc))
            e.verify_batch_msm_partial_device(ds.data_ptr() if cnt else 0, dp.data_ptr() if cnt else 0,
                                              dm.data_ptr() if cnt else 0, cnt, 80, dc.data_ptr() if cnt else 0, 32,
                                              recs[r].data_ptr())
            e.sync()
        return recs

    try:
        # honest
        rec = records_host(sigs)
        assert (rec[:, 22] == 0).all() and (rec[:, 23] == np.uint64(ssa.MSM_RECORD_MAGIC)).all()
        assert engines[1].msm_combine(rec) == engine.verify_batch_msm(sigs, pks, msgs, coeffs=co) == 0
        rec_d = records_device(sigs)
        assert (rec_d.cpu().numpy().astype(np.uint64) == rec).all(), "device and host records differ"
        verdict = torch.full((1,), 255, dtype=torch.int32, device=dev)
        engines[0].msm_combine_device(rec_d.data_ptr(), 3, verdict.data_ptr())
        engines[0].sync()
        assert int(verdict.item()) == 0
        # the order of the records does not matter (a sum); honest shards verify on their own (a sub-batch); a record
        # whose scalar or point was tampered with on the way does not
        assert engines[2].msm_combine(rec[::-1].copy()) == 0
        assert engines[2].msm_combine(rec[:2].copy()) == 0
        tam = rec.copy()
        tam[0, 18] ^= np.uint64(1)
        assert engines[2].msm_combine(tam) == 2
        tam = rec.copy()
        tam[0, :18], tam[1, :18] = rec[1, :18], rec[0, :18]                    # points swapped, scalars not: still the same sums
        assert engines[2].msm_combine(tam) == 0
        tam[0, 0] ^= np.uint64(2)                                               # a point off the curve is not a record:
        assert engines[2].msm_combine(tam) == 3                                 # the combination fails closed (round 4)
        for w, v in ((23, 0), (23, ssa.MSM_RECORD_MAGIC ^ 1), (5, 2**64 - 1), (21, 2**63), (22, 2)):
            tam = rec.copy()
            tam[1, w] = np.uint64(v)                                            # no magic / foreign format / limb >= p / scalar >= q / bad flag
            assert engines[2].msm_combine(tam) == 3, (w, v)
        # one corrupted signature in the last shard
        bad = sigs.copy()
        bad[n - 1, 50] ^= 4
        assert engines[0].msm_combine(records_host(bad)) == engine.verify_batch_msm(bad, pks, msgs, coeffs=co) == 2
        # undecodable signature in the middle shard: the reference panics (src/batch.rs:104)
        und = sigs.copy()
        und[n // 2, 48] |= 2
        rec_u = records_host(und)
        assert rec_u[:, 22].sum() == 1
        assert engines[0].msm_combine(rec_u) == engine.verify_batch_msm(und, pks, msgs, coeffs=co) == 3
        # identity key (contributes nothing, src/batch.rs:106): the batch no longer verifies
        inf = np.zeros(n, np.uint8)
        inf[1] = 1
        assert engines[0].msm_combine(records_host(sigs, inf)) == \
            engine.verify_batch_msm(sigs, pks, msgs, coeffs=co, pk_inf=inf) == 2
        if n <= 1000:
            assert oracle.verify_batch_msm(sigs, pks, msgs, co) == 0 and oracle.verify_batch_msm(bad, pks, msgs, co) == 2
            assert oracle.verify_batch_msm(und, pks, msgs, co) == 3
            assert oracle.verify_batch_msm(sigs, pks, msgs, co, pk_inf=inf) == 2
    finally:
        for e in engines:
            e.close()


def ssa_magic():
    import schnorr_sig_amd as ssa
    return np.uint64(ssa.MSM_RECORD_MAGIC)


def test_msm_combine_rejects_bad_arguments(engine):
    with pytest.raises(RuntimeError, match="invalid argument"):
        engine.msm_combine(np.zeros((0, 24), np.uint64))
    # k empty shards: the identity on the left, [0]G on the right -> Ok, like the reference's empty batch
    empty = np.zeros((2, 24), np.uint64)
    empty[:, 23] = ssa_magic()
    assert engine.msm_combine(empty) == 0
    assert (engine.verify_batch_msm_partial(np.zeros((0, 81), np.uint8), np.zeros((0, 96), np.uint8), None) == empty[0]).all()
    # a slot nobody wrote is NOT an empty shard (an unwritten torch.zeros(24) must not read as "accept")
    assert engine.msm_combine(np.zeros((2, 24), np.uint64)) == 3
    empty[1, 23] = 0
    assert engine.msm_combine(empty) == 3


def test_sharding_msm_verdict_single_process(engine):
    """schnorr_sig_amd.sharding.msm_verdict without a process group: the gather is the identity, the combination runs"""
    import torch
    from schnorr_sig_amd.sharding import msm_verdict
    rng = np.random.default_rng(3100)
    sigs, pks, msgs = honest(engine, rng, 50)
    rec = torch.from_numpy(engine.verify_batch_msm_partial(sigs, pks, msgs).astype(np.int64))
    v, recs = msm_verdict(rec, 1, None, lambda r: engine.msm_combine(r.numpy().astype(np.uint64)))
    assert v == 0 and recs.shape == (1, 24)


# ---------------------------------------------------------------- pipelined upload: ordering and error paths
def test_async_device_verify_followed_by_pipelined_host_verify(engine, oracle):
    """An asynchronous *_device call still reads ws_h / the tables while a host-buffer call with n >= 2^17 starts its
    chunked upload on the side streams: the side streams must wait for the context's stream."""
    import torch
    rng = np.random.default_rng(3200)
    n1, n2 = 1 << 18, (1 << 17) + 77
    s1, p1, m1 = honest(engine, rng, n1)
    bad1 = rng.permutation(n1)[:300]
    s1[bad1, 55] ^= 8
    s2, p2, m2 = honest(engine, rng, n2)
    bad2 = rng.permutation(n2)[:200]
    m2[bad2, 3] ^= 1
    dev = torch.device("cuda", 0)
    ds, dp, dm = (torch.from_numpy(a).to(dev) for a in (s1, p1, m1))
    dst = torch.full((n1,), 255, dtype=torch.uint8, device=dev)
    dnf = torch.zeros(1, dtype=torch.int64, device=dev)
    engine.sync()
    for _ in range(3):
        engine.verify_many_device(ds.data_ptr(), dp.data_ptr(), dm.data_ptr(), n1, 80, dst.data_ptr(), dnf.data_ptr(),
                                  mode="lane")
        # no synchronisation here
        st2, nf2 = engine.verify_many(s2, p2, m2, check_torsion=False, mode="lane")
        engine.sync()
        st1 = dst.cpu().numpy()
        assert int(dnf.item()) == 300 and (st1[bad1] == 2).all() and int((st1 != 0).sum()) == 300
        assert nf2 == 200 and (st2[bad2] == 2).all() and int((st2 != 0).sum()) == 200
    samp = np.concatenate([bad1[:50], np.arange(0, n1, 4099)])
    assert (st1[samp] == oracle.verify_many(s1[samp], p1[samp], m1[samp], check_torsion=False)).all()


def test_pipelined_upload_error_paths(engine, oracle):
    """An error after the first enqueue of the chunked upload (injected: ssa_debug_fault_after_chunk) must return an error
    code with every stream drained -- the caller's arrays are unpinned and may be freed at once -- and the next call on
    the context must be correct.  Bad message arguments are refused before anything is pinned or enqueued."""
    rng = np.random.default_rng(3300)
    n = (1 << 17) + 5
    sigs, pks, msgs = honest(engine, rng, n)
    sigs[7, 49] ^= 1
    for chunk in (0, 3, 7):
        try:
            engine.debug_fault_after_chunk(chunk)          # one shot: armed before each call
            with pytest.raises(RuntimeError, match="HIP runtime error"):
                engine.verify_many(sigs, pks, msgs, check_torsion=False, mode="lane")
            engine.debug_fault_after_chunk(chunk)
            with pytest.raises(RuntimeError, match="HIP runtime error"):
                engine.verify_batch_msm(sigs, pks, msgs, coeffs=coeffs32(rng, n))
        finally:
            engine.debug_fault_after_chunk(-1)
        # the arrays the failed call read are released and overwritten right away
        scratch = sigs.copy()
        scratch[:] = 0
        del scratch
        st, nf = engine.verify_many(sigs, pks, msgs, check_torsion=False, mode="lane")
        assert nf == 1 and st[7] == 2 and int((st != 0).sum()) == 1
    assert engine.verify_batch_msm(sigs, pks, msgs) == 2
    # offsets that run backwards: SSA_ERR_ARG from the pipelined path too
    off = np.arange(n + 1, dtype=np.uint64) * 80
    off[5] = off[6] + 1
    with pytest.raises(RuntimeError, match="invalid argument"):
        engine.verify_many(sigs, pks, msgs.reshape(-1), offsets=off, check_torsion=False, mode="lane")
    samp = np.arange(0, 2048)
    assert (st[samp] == oracle.verify_many(sigs[samp], pks[samp], msgs[samp], check_torsion=False)).all()


# ---------------------------------------------------------------- lifetime and probe hygiene
def test_keyset_may_outlive_its_engine():
    import schnorr_sig_amd as ssa
    eng = ssa.Engine(0)
    rng = np.random.default_rng(3400)
    sigs, pks, msgs = honest(eng, rng, 8)
    ks = eng.keyset_create(pks, kind="ladder")
    st, nf = eng.verify_many_indexed(ks, np.arange(8, dtype=np.uint32), sigs, msgs)
    assert nf == 0
    assert ks.engine is eng                      # the wrapper keeps the engine alive
    eng.close()                                  # context destroyed first: the key set is orphaned, not dangling
    with pytest.raises(RuntimeError, match="invalid argument"):
        eng2 = ssa.Engine(0)
        try:
            eng2.keyset_status(ks)
        finally:
            eng2.close()
    ks.close()                                   # frees the host handle only; no use-after-free
    ks.close()


def test_debug_window_probe_refuses_zero_or_mixed_counts(engine):
    a = np.zeros((64, 20), np.uint64)
    b = np.zeros((64, 12), np.uint64)
    a[:, 12] = 1
    a[:, 19] = 0
    for op in (15, 17):
        with pytest.raises(RuntimeError, match="invalid argument"):
            engine.debug_arith(op, a, b, 19)
    a[:, 19] = 2
    a[5, 19] = 3
    with pytest.raises(RuntimeError, match="invalid argument"):
        engine.debug_arith(17, a, b, 19)


def test_abi_version_matches_header():
    import re
    import schnorr_sig_amd as ssa
    hdr = open(os.path.join(ROOT, "include", "schnorr_sig_amd.h")).read()
    assert int(re.search(r"#define SSA_ABI_VERSION (\d+)", hdr).group(1)) == ssa.ABI_VERSION == ssa._lib.ssa_abi_version()


# ---------------------------------------------------------------- the reference's own batch sizes, MSM form
@pytest.mark.parametrize("n", [4, 16, 32, 64, 128, 500, 2047, 2048, 4095, 4096])
def test_verify_batch_msm_at_the_reference_bench_sizes(engine, oracle, n):
    """benches/schnorr.rs:78-96 batches (4..128 signatures) and the sizes around the window-width / small-batch
    switches, through ssa_verify_batch_msm against the oracle's src/batch.rs restatement"""
    rng = np.random.default_rng(3500 + n)
    sigs, pks, msgs = honest(engine, rng, n)
    co = coeffs32(rng, n)
    assert engine.verify_batch_msm(sigs, pks, msgs, coeffs=co) == 0
    assert engine.verify_batch_msm(sigs, pks, msgs) == 0
    bad = sigs.copy()
    bad[n // 3, 60] ^= 1
    assert engine.verify_batch_msm(bad, pks, msgs, coeffs=co) == 2
    und = sigs.copy()
    und[n - 1, 48] = 0xff
    assert engine.verify_batch_msm(und, pks, msgs, coeffs=co) == 3
    dup = sigs.copy()
    dupk = pks.copy()
    dupm = msgs.copy()
    dup[1], dupk[1], dupm[1] = sigs[0], pks[0], msgs[0]                       # equal points meet in one bucket / table
    assert engine.verify_batch_msm(dup, dupk, dupm, coeffs=co) == 0
    if n <= 128:
        assert oracle.verify_batch_msm(sigs, pks, msgs, co) == 0 and oracle.verify_batch_msm(bad, pks, msgs, co) == 2
        assert oracle.verify_batch_msm(und, pks, msgs, co) == 3 and oracle.verify_batch_msm(dup, dupk, dupm, co) == 0


# ---------------------------------------------------------------- Rescue S-boxes: the flagged-lane fallback on hardware
def test_sbox_blocks_flag_and_recompute_the_rare_borrow(engine):
    """The generated S-box blocks (tools/gen_fp_chain_asm.py) leave one event of their reduction to the caller: a
    borrow with probability ~2^-32 per squaring.  x = k 2^48 forces it in the first squaring (x^2 = k^2 2^96: lo = 0,
    hi.lo = 0).  On the GPU: the blocks must flag exactly those lanes (their bit in the wave-wide mask operand), the
    caller's sequence -- asm block, then the compiled exact chain from the inputs for a flagged lane, what sponge_hash does
    for a whole hash -- must return the right values for EVERY lane, neighbours unaffected."""
    P = 2**64 - 2**32 + 1
    e_inv = 10540996611094048183
    rng = np.random.default_rng(3600)
    n = 256
    a = rng.integers(0, 2**64, size=(n, 2), dtype=np.uint64)
    forced = np.zeros(n, bool)
    for i in range(0, n, 7):                     # sprinkled over the waves, either chain
        a[i, i % 2] = np.uint64((int(rng.integers(1, 2**16)) << 48))
        forced[i] = True
    a[3] = (0, 1)
    a[4] = (P - 1, P)
    a[5] = (2**64 - 1, 2**32)
    out = engine.debug_arith(18, a, None, 6)
    for i in range(n):
        x, y = int(a[i, 0]), int(a[i, 1])
        assert [int(v) for v in out[i, :4]] == [pow(x, 7, P), pow(y, 7, P), pow(x, e_inv, P), pow(y, e_inv, P)], i
        if forced[i]:
            assert out[i, 4] == 1 and out[i, 5] == 1, (i, out[i, 4:])        # flagged by both programs
    plain = ~forced
    plain[3:6] = False
    assert (out[plain, 4:] == 0).all()           # a random value does not get there (2^-32 per squaring)


def test_hash_of_inputs_that_force_the_sbox_fallback(engine, oracle):
    """whole-hash check of the same event: felts k 2^48 enter the first forward S-box layer directly"""
    rng = np.random.default_rng(3601)
    felts = rng.integers(0, 2**63, size=(512, 25), dtype=np.uint64)
    felts[::3, :8] = (rng.integers(1, 2**16, size=(felts[::3].shape[0], 8)).astype(np.uint64) << np.uint64(48))
    got = engine.rescue_hash_many(felts)
    want = oracle.hash_field_many(felts)
    assert (got == want).all()


# ---------------------------------------------------------------- Fp6 reductions: the cold path on hardware
def test_reduction_cold_paths_on_the_gpu(engine):
    """The generated Fp6 reductions (tools/gen_f6_asm.py, round 3) settle a negative result -- probability ~2^-28 per
    coefficient on random data -- in a cold path behind one branch per group of three.  Operands k 2^48 in a single
    coefficient make products k k' 2^96 = -k k' (mod p): exactly that case, in the plain and the fused blocks and in
    the ladder's statements (doublings, mixed addition, whole window), on raw limbs, against plain integers.  The CPU
    suite checks with the one-lane interpreter that these operand shapes do take the cold path
    (tests/test_asm_emulation.py)."""
    P = 2**64 - 2**32 + 1
    rng = np.random.default_rng(3700)

    def mulmod(u, v):
        t = [0] * 12
        for i, x in enumerate(u):
            for j, y in enumerate(v):
                t[i + j] += x * y
        return [(t[k] + 7 * t[k + 6]) % P for k in range(6)]

    sub = lambda u, v: [(a - b) % P for a, b in zip(u, v)]
    add = lambda u, v: [(a + b) % P for a, b in zip(u, v)]
    sc = lambda c, u: [c * a % P for a in u]

    def sparse(n, cols):
        out = np.zeros((n, cols), dtype=np.uint64)
        for r in range(n):
            for blk in range(cols // 6):
                out[r, 6 * blk + int(rng.integers(0, 6))] = np.uint64(int(rng.integers(1, 2**15)) << 48)
        return out

    n = 192
    a, b = sparse(n, 12), sparse(n, 12)
    models = {0: lambda u, v, x, y: mulmod(u, x), 1: lambda u, v, x, y: mulmod(u, u),
              8: lambda u, v, x, y: sub(sub(mulmod(u, u), x), y), 12: lambda u, v, x, y: sub(mulmod(u, v), x),
              14: lambda u, v, x, y: add(mulmod(u, v), mulmod(x, y))}
    for op, model in models.items():
        got = engine.debug_arith(op, a, b, 6)
        for i in range(n):
            u, v = [int(t) for t in a[i, :6]], [int(t) for t in a[i, 6:]]
            x, y = [int(t) for t in b[i, :6]], [int(t) for t in b[i, 6:]]
            assert [int(t) for t in got[i]] == model(u, v, x, y), (op, i)

    # the ladder's statements: X, Y, Z and (x2, y2) with one k 2^48 limb each (first coefficient: the exceptional-input
    # tests of the addition look at it and must see a non-zero value)
    def dbl(X, Y, Z):
        XX, YY, ZZ = mulmod(X, X), mulmod(Y, Y), mulmod(Z, Z)
        YYYY = mulmod(YY, YY)
        t = add(X, YY)
        S = sc(2, sub(sub(mulmod(t, t), XX), YYYY))
        M = add(sc(3, XX), mulmod(ZZ, ZZ))
        X3 = sub(mulmod(M, M), sc(2, S))
        return X3, sub(mulmod(M, sub(S, X3)), sc(8, YYYY)), sc(2, mulmod(Y, Z))

    def madd(X, Y, Z, x2, y2):
        ZZ = mulmod(Z, Z)
        H, R = sub(mulmod(x2, ZZ), X), sub(mulmod(mulmod(y2, Z), ZZ), Y)
        HH = mulmod(H, H)
        HHH, V = mulmod(H, HH), mulmod(X, HH)
        X3 = sub(sub(mulmod(R, R), HHH), sc(2, V))
        return (X3, sub(mulmod(R, sub(V, X3)), mulmod(Y, HHH)), mulmod(Z, H)), H

    pa = np.zeros((n, 20), dtype=np.uint64)
    pb = np.zeros((n, 12), dtype=np.uint64)
    for r in range(n):
        for blk in range(3):
            pa[r, 6 * blk] = np.uint64(int(rng.integers(1, 2**15)) << 48)
        for blk in range(2):
            pb[r, 6 * blk] = np.uint64(int(rng.integers(1, 2**15)) << 48)
    pa[:, 18] = 1
    pa[:, 19] = 2
    got15, got16, got17 = (engine.debug_arith(op, pa, pb, 19) for op in (15, 16, 17))
    red = lambda row: [[int(t) % P for t in row[6 * k:6 * k + 6]] for k in range(3)]
    for i in range(n):
        pt = [[int(t) for t in pa[i, 6 * k:6 * k + 6]] for k in range(3)]
        q = [[int(t) for t in pb[i, 6 * k:6 * k + 6]] for k in range(2)]
        d = pt
        for _ in range(2):
            d = [list(v) for v in dbl(*d)]
        assert red(got17[i]) == d, ("dbl_n", i)
        m, H = madd(*d, *q)
        if d[2][0] != 0 and H[0] != 0:
            assert int(got15[i, 18]) == 1 and red(got15[i]) == [list(v) for v in m], ("window", i)
        m, H = madd(*pt, *q)
        if H[0] != 0:
            assert red(got16[i]) == [list(v) for v in m], ("madd", i)
