"""GPU tests added in round 4 (all through the C ABI):
 * bench.py with TWO ranks on real kernels (gloo rendezvous, both ranks on the one GPU of the box): every rank != 0
   branch of the N > 1 run -- the receive side of the scatter / broadcast, the ragged config-4 shard, the all-reduced
   rejection counts, the MSM records of two shards gathered and combined -- before the driver's multi-GPU node sees it
   (BASELINE.json configs[3], SURVEY.md 8(e), reference src/batch.rs:98-129);
 * the headline's flags (SSA_FLAG_SIG_FLAG_BYTE = what ssa_verify_batch runs, src/batch.rs:104) in the bench line."""
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


def _bench(args, env_extra=None, timeout=900):
    """bench.py as a FRESH child process (never an exec of this process, which has touched the GPU)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT",
                                                            "MASTER_ADDR", "GROUP_RANK", "LOCAL_WORLD_SIZE")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


# ---------------------------------------------------------------- N = 2 with real kernels
def test_bench_two_ranks_real_kernels_corrupted_and_ragged():
    """Two ranks (gloo: the box has one GPU, both ranks run their kernels on it), 1 % corrupted shards, a config-4 leg
    whose total does not divide by two (rank 1 gets the short shard), MSM records of both shards combined."""
    out = _bench(["--gpus", "2", "--batch", "65536", "--strong-total", "262145", "--corrupt", "0.01", "--steps", "2",
                  "--warmup", "1"], env_extra={"SSA_BENCH_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["config"]["backend"] == "gloo"
    assert len(out["ranks"]) == 2 and sorted(r[0] for r in out["ranks"]) == [0, 1]
    assert out["config"]["signatures_per_gpu"] == 65536 and out["config"]["signatures_total"] == 131072
    assert out["scaling"] == "weak"
    # the all-reduced rejection count: both shards' 655 corruptions
    assert out["rejected"] == 2 * 655 and out["all_verdicts_as_expected"] is True
    assert out["with_torsion_check_rejected"] == 2 * 655
    assert out["value"] > 0 and abs(out["value"] - 131072 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
    # flags of the timed step = what ssa_verify_batch runs
    assert out["config"]["flags"]["word"] == 8
    # config-4 leg: ONE ragged batch on rank 0 -> scatter and broadcast both timed, honest batch
    c4 = out["config4_strong"]
    assert "error" not in c4, c4
    assert c4["signatures_total"] == 262145 and c4["rejected"] == 0
    assert c4["scatter_ms"] is not None and c4["scatter_ms"] > 0
    assert c4["broadcast_ms"] is not None and c4["broadcast_ms"] > 0
    # MSM form: one verdict over both ranks' records (k = 2)
    msm = out["verify_batch_msm_form"]
    assert msm["combined_over_ranks"] is True and msm["verdict"] == msm["expected_verdict"] == 2
    # the CPU leg belongs to the N = 1 line
    assert out["cpu_baseline"] is None and "N = 1" in out["cpu_baseline_note"]
    assert out["parity_unpinned"] is True


def test_bench_two_ranks_honest_msm_verdict_is_ok():
    out = _bench(["--gpus", "2", "--batch", "16384", "--steps", "1", "--warmup", "1", "--no-strong-leg"],
                 env_extra={"SSA_BENCH_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["rejected"] == 0 and out["all_verdicts_as_expected"] is True
    msm = out["verify_batch_msm_form"]
    assert msm["combined_over_ranks"] is True and msm["verdict"] == msm["expected_verdict"] == 0
    assert out["config4_strong"] is None


def test_bench_strong_mode_two_ranks_ragged_total():
    """`--total` (config 4 as the main mode): rank 0 generates, both distributions run, rank 1's shard is one short"""
    out = _bench(["--gpus", "2", "--total", "100001", "--steps", "1", "--warmup", "1", "--skip-torsion-leg"],
                 env_extra={"SSA_BENCH_BACKEND": "gloo"})
    assert out["scaling"] == "strong" and out["config"]["signatures_total"] == 100001
    assert out["config"]["signatures_per_gpu"] == 50001          # rank 0's shard (earlier ranks take the remainder)
    assert out["scatter_ms"] > 0 and out["broadcast_ms"] > 0
    assert out["rejected"] == 0 and out["all_verdicts_as_expected"] is True


def test_bench_line_names_its_flags_and_cpu_leg_runs_them():
    out = _bench(["--gpus", "1", "--batch", "8192", "--steps", "1", "--warmup", "1", "--skip-torsion-leg",
                  "--cpu-sample", "512", "--corrupt", "0.01"])
    assert out["config"]["flags"] == {"word": 8, "names": ["SSA_FLAG_SIG_FLAG_BYTE"],
                                     "meaning": out["config"]["flags"]["meaning"]}
    assert "verify_batch semantics" in out["config"]["workload"]
    cb = out["cpu_baseline"]
    assert cb["flags"] == 8 and cb["agrees_with_gpu"] is True and "SSA_FLAG_SIG_FLAG_BYTE" in cb["sample"]


def test_bench_pipelined_leg_verifies_what_the_metric_verifies():
    """`pipelined_two_contexts` (two contexts of the device over the halves of every batch: what a caller that pipelines batches
    gets) rejects exactly the corrupted signatures of the batch the timed steps ran on, and is reported beside the metric"""
    out = _bench(["--gpus", "1", "--batch", "65536", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--corrupt", "0.01"])
    leg = out["pipelined_two_contexts"]
    assert "error" not in leg, leg
    assert leg["rejected"] == out["rejected"] == 655
    assert leg["verifications_per_sec"] > 0 and "not the metric" in leg["note"]
    assert out["all_verdicts_as_expected"] is True


# ---------------------------------------------------------------- bounded workspaces: slices of a fixed lane count
def _sliced_engine(lane=65536, msm=65536):
    """an engine whose per-lane kernels / MSM pipeline run over slices of the given size (read at ssa_ctx_create)"""
    import schnorr_sig_amd as ssa
    os.environ["SSA_LANE_SLICE"], os.environ["SSA_MSM_SLICE"] = str(lane), str(msm)
    try:
        return ssa.Engine(0)
    finally:
        del os.environ["SSA_LANE_SLICE"], os.environ["SSA_MSM_SLICE"]


def _spoil(rng, sigs, pks, msgs, count):
    """config-5 style corruptions on `count` distinct lanes; returns (sigs, pks, msgs, pk_inf, bad indices)"""
    n = sigs.shape[0]
    sigs, pks, msgs = sigs.copy(), pks.copy(), msgs.copy()
    bad = rng.permutation(n)[:count]
    q = count // 4
    sigs[bad[:q], 49] ^= 1                                  # e bit flip
    msgs[bad[q:2 * q], 17] ^= 0x20                          # message bit flip
    pks[bad[2 * q:3 * q]] = pks[(bad[2 * q:3 * q] + 1) % n]  # someone else's key
    sigs[bad[3 * q:], :49] = sigs[(bad[3 * q:] + 1) % n, :49]  # someone else's R
    inf = np.zeros(n, np.uint8)
    inf[bad[0]] = 1                                         # an identity key on a lane that is rejected anyway
    return sigs, pks, msgs, inf, bad


def test_lane_kernels_in_slices_equal_the_unsliced_run(engine, oracle):
    """ws_tab is sized for one slice whatever n is (SSA_MAX_BATCH is honest): slice forced to 65 536 lanes at a ragged
    n = 200 001 -- same status vector and ONE rejection count as the unsliced engine, device and host-buffer (pipelined)
    entry points, subgroup check on and off, flag-byte semantics on and off; oracle on a sample and every spoiled lane"""
    import torch
    rng = np.random.default_rng(4100)
    n = 200001
    sigs, pks, msgs = honest(engine, rng, n)
    sigs, pks, msgs, inf, bad = _spoil(rng, sigs, pks, msgs, 400)
    eng2 = _sliced_engine()
    dev = torch.device("cuda", 0)
    try:
        ds, dp, dm, di = (torch.from_numpy(a).to(dev) for a in (sigs, pks, msgs, inf))
        samp = np.unique(np.concatenate([bad, np.arange(0, n, 197), [65535, 65536, 131071, 131072, 196607, 196608, n - 1]]))
        for torsion in (False, True):
            for fb in (False, True):
                ref, nf_ref = engine.verify_many(sigs, pks, msgs, check_torsion=torsion, pk_inf=inf, mode="lane",
                                                 sig_flag_byte=fb)
                got, nf = eng2.verify_many(sigs, pks, msgs, check_torsion=torsion, pk_inf=inf, mode="lane",
                                           sig_flag_byte=fb)                   # n >= 2^17: pipelined upload + slices
                assert nf == nf_ref == int((ref != 0).sum()) and (got == ref).all()
                dst = torch.full((n,), 255, dtype=torch.uint8, device=dev)
                dnf = torch.zeros(1, dtype=torch.int64, device=dev)
                eng2.verify_many_device(ds.data_ptr(), dp.data_ptr(), dm.data_ptr(), n, 80, dst.data_ptr(), dnf.data_ptr(),
                                        d_pk_inf=di.data_ptr(), check_torsion=torsion, mode="lane", sig_flag_byte=fb)
                eng2.sync()
                assert int(dnf.item()) == nf_ref and (dst.cpu().numpy() == ref).all()
                exp = oracle.verify_many(sigs[samp], pks[samp], msgs[samp], check_torsion=torsion, pk_inf=inf[samp],
                                         sig_flag_byte=fb)
                assert (ref[samp] == exp).all()
            assert nf_ref == 400
        # ragged messages through the offset table (the slices move the table, not the bytes)
        lens = rng.integers(0, 30, size=n)
        off = np.zeros(n + 1, np.uint64)
        off[1:] = np.cumsum(lens)
        flat = rng.integers(0, 256, size=int(off[-1]) + 1, dtype=np.uint8)
        m2 = 70001
        sk, nn = make_scalars(rng, m2), make_scalars(rng, m2)
        pk2, sg2 = engine.keygen_sign_many(sk, nn, flat, offsets=off[:m2 + 1])
        sg2[5, 60] ^= 2
        sg2[66000, 49] ^= 1
        ref, nf_ref = engine.verify_many(sg2, pk2, flat, offsets=off[:m2 + 1], check_torsion=False, mode="lane")
        got, nf = eng2.verify_many(sg2, pk2, flat, offsets=off[:m2 + 1], check_torsion=False, mode="lane")
        assert nf == nf_ref == 2 and (got == ref).all() and got[5] == 2 and got[66000] == 2
        # the other kernel family (one block per signature, no per-lane workspace) on the same engine
        sub = slice(0, 70001)
        ref_c, _ = engine.verify_many(sigs[sub], pks[sub], msgs[sub], check_torsion=True, pk_inf=inf[sub], mode="lane")
        got_c, _ = eng2.verify_many(sigs[sub], pks[sub], msgs[sub], check_torsion=True, pk_inf=inf[sub], mode="coop")
        assert (got_c == ref_c).all()
    finally:
        eng2.close()


@pytest.mark.parametrize("n", [200001, 3 * 65536 + 100])
def test_msm_form_in_slices_equals_the_unsliced_verdict(engine, oracle, n):
    """more than msm_slice signatures: every slice is reduced to its record like a shard of a multi-GPU batch and the
    records are combined (src/batch.rs:98-129); forced to 65 536 here -- the last slice is ragged (3 393 signatures) or
    small enough for the cooperative small-batch path (100)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import msm_records as mr
    rng = np.random.default_rng(4200 + n % 1000)
    sigs, pks, msgs = honest(engine, rng, n)
    co = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    co[:, 16:] = 0
    eng2 = _sliced_engine()
    try:
        assert engine.verify_batch_msm(sigs, pks, msgs, coeffs=co) == eng2.verify_batch_msm(sigs, pks, msgs, coeffs=co) == 0
        assert eng2.verify_batch_msm(sigs, pks, msgs) == 0                           # library-drawn coefficients per slice
        # the record of the whole batch: same affine point and scalar whichever way it was cut
        r1 = engine.verify_batch_msm_partial(sigs, pks, msgs, coeffs=co)
        r2 = eng2.verify_batch_msm_partial(sigs, pks, msgs, coeffs=co)
        assert mr.record_point(oracle, r1) == mr.record_point(oracle, r2) and mr.record_lin(r1) == mr.record_lin(r2)
        assert int(r2[23]) == mr.MAGIC and mr.record_is_wellformed(oracle, r2)
        for lane in (7, 65536, n - 1):                                               # one bad signature in the first / a middle / the last slice
            bad = sigs.copy()
            bad[lane, 52] ^= 0x40
            assert eng2.verify_batch_msm(bad, pks, msgs, coeffs=co) == 2
        und = sigs.copy()
        und[n - 2, 48] |= 4                                                         # undecodable flag byte: the reference panics
        assert eng2.verify_batch_msm(und, pks, msgs, coeffs=co) == engine.verify_batch_msm(und, pks, msgs, coeffs=co) == 3
        inf = np.zeros(n, np.uint8)
        inf[70000] = 1
        assert eng2.verify_batch_msm(sigs, pks, msgs, coeffs=co, pk_inf=inf) == 2
    finally:
        eng2.close()


# ---------------------------------------------------------------- stream ordering of the shard records
def test_msm_partial_on_its_own_stream_is_ordered_by_msm_verdict():
    """ssa_verify_batch_msm_partial_device only enqueues on the context's stream -- here the context's OWN non-blocking
    stream, never handed to torch -- while the gather runs on torch's current stream.  sharding.msm_verdict(engine=...)
    orders the two without a host synchronisation; alternating honest and forged batches into the SAME record buffer
    would show a stale or unwritten record as a wrong verdict."""
    import torch
    import schnorr_sig_amd as ssa
    from schnorr_sig_amd.sharding import msm_verdict
    eng = ssa.Engine(0)
    try:
        rng = np.random.default_rng(4300)
        n = 20000
        sigs, pks, msgs = honest(eng, rng, n)
        bad = sigs.copy()
        bad[n // 2, 49] ^= 1
        dev = torch.device("cuda", 0)
        d_good, d_bad, dp, dm = (torch.from_numpy(a).to(dev) for a in (sigs, bad, pks, msgs))
        record = torch.zeros(24, dtype=torch.int64, device=dev)
        verdict = torch.full((1,), 255, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()

        def combine(records):
            eng.msm_combine_device(records.data_ptr(), records.shape[0], verdict.data_ptr())
            return verdict

        side = torch.cuda.Stream(device=dev)
        for it in range(8):
            ds = d_good if it % 2 == 0 else d_bad
            eng.verify_batch_msm_partial_device(ds.data_ptr(), dp.data_ptr(), dm.data_ptr(), n, 80, 0, 16,
                                                record.data_ptr())
            with torch.cuda.stream(side if it >= 4 else torch.cuda.current_stream()):
                _, recs = msm_verdict(record, 1, None, combine, engine=eng)      # no host synchronisation in here
            eng.sync()
            assert int(verdict.cpu().item()) == (0 if it % 2 == 0 else 2), it
            assert int(recs.cpu()[0, 23]) == ssa.MSM_RECORD_MAGIC - (1 << 64 if ssa.MSM_RECORD_MAGIC >= 1 << 63 else 0)
        # a record buffer nobody wrote is not an empty shard
        blank = torch.zeros((1, 24), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        combine(blank)
        eng.sync()
        assert int(verdict.cpu().item()) == 3
    finally:
        eng.close()


# ---------------------------------------------------------------- the signing side: constant-time mode, wire forms
def _scalars_le(values):
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in values), np.uint8).reshape(-1, 32).copy()


EDGE_SCALARS = [1, 2, 15, 16, 17, Q - 1, Q - 2, 1 << 252, (1 << 252) + 1, int("1" * 63, 16), int("6" + "f" * 63, 16),
                int("0f" * 32, 16) % Q, int("f0" * 31 + "70", 16), 16**63, 16**32 - 1, (Q - 1) // 2]


def test_constant_time_signer_bytes_equal_oracle_and_throughput_signer(engine, oracle):
    """SSA_FLAG_SIGN_CT (the reference's constant-time `&BASEPOINT_TABLE * r`, src/signature.rs:67,116,123): the same
    bytes as the throughput signer and as the oracle's signer (src/signature.rs:114-129), random and edge scalars --
    one set digit, all digits 15, q - 1 --, ragged messages incl. empty ones; every signature verifies"""
    rng = np.random.default_rng(4400)
    edge = [v for v in EDGE_SCALARS if 0 < v < Q]
    n = 600
    vals_sk = edge + [int.from_bytes(rng.bytes(64), "little") % Q or 1 for _ in range(n - len(edge))]
    vals_r = edge[::-1] + [int.from_bytes(rng.bytes(64), "little") % Q or 1 for _ in range(n - len(edge))]
    sks, nonces = _scalars_le(vals_sk), _scalars_le(vals_r)
    lens = rng.integers(0, 200, size=n)
    lens[:4] = [0, 7, 14, 1]
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    flat = rng.integers(0, 256, size=int(off[-1]) + 1, dtype=np.uint8)
    pk_v, sig_v = engine.keygen_sign_many(sks, nonces, flat, offsets=off)
    pk_c, sig_c = engine.keygen_sign_many(sks, nonces, flat, offsets=off, constant_time=True)
    assert (pk_c == pk_v).all() and (sig_c == sig_v).all()
    m = 96
    pk_o, sig_o = oracle.keygen_sign_many(sks[:m], nonces[:m], flat, offsets=off[:m + 1])
    assert (pk_c[:m] == pk_o).all() and (sig_c[:m] == sig_o).all()
    st, nf = engine.verify_many(sig_c, pk_c, flat, offsets=off, check_torsion=True)
    assert nf == 0
    # the host form refuses 0 and values >= q in both modes (PrivateKey::new / Scalar::random never yield them)
    for ct in (False, True):
        for bad in (0, Q, Q + 5, (1 << 256) - 1):
            s2 = sks[:3].copy()
            s2[1] = _scalars_le([bad])[0]
            with pytest.raises(RuntimeError, match="invalid argument"):
                engine.keygen_sign_many(s2, nonces[:3], flat, offsets=off[:4], constant_time=ct)
            with pytest.raises(RuntimeError, match="invalid argument"):
                engine.keygen_sign_many(sks[:3], s2, flat, offsets=off[:4], constant_time=ct)


def test_constant_time_signer_device_form_reduces_and_falls_back(engine):
    """the _device form cannot refuse a lane: scalars are reduced mod q; a scalar that reduces to 0 drives the
    constant-time walk onto B - B, the lane is flagged and recomputed by the exact code -- same bytes as the
    throughput signer (identity key / identity R encodings included)"""
    import torch
    dev = torch.device("cuda", 0)
    vals_sk = [Q, 5, Q + 3, 2 * Q, 7, 11]
    vals_r = [9, Q, 2 * Q + 1, 13, (1 << 256) - 1, 1]
    n = len(vals_sk)
    msgs = np.random.default_rng(4500).integers(0, 256, size=(n, 33), dtype=np.uint8)
    d_sk, d_r, d_m = (torch.from_numpy(a).to(dev) for a in (_scalars_le(vals_sk), _scalars_le(vals_r), msgs))
    outs = []
    for ct in (False, True):
        pks = torch.full((n, 96), 0xAA, dtype=torch.uint8, device=dev)
        sigs = torch.full((n, 81), 0xAA, dtype=torch.uint8, device=dev)
        engine.keygen_sign_many_device(d_sk.data_ptr(), d_r.data_ptr(), d_m.data_ptr(), n, 33, pks.data_ptr(),
                                       sigs.data_ptr(), constant_time=ct)
        engine.sync()
        outs.append((pks.cpu().numpy(), sigs.cpu().numpy()))
    assert (outs[0][0] == outs[1][0]).all() and (outs[0][1] == outs[1][1]).all()
    pks, sigs = outs[1]
    assert not pks[0].any() and not pks[3].any()                # sk = 0: the identity key, (0, 0) in affine bytes
    assert not sigs[1, :48].any() and sigs[1, 48] == 0x80       # nonce = 0: R is the identity, [0; 48] || 0x80
    st, nf = engine.verify_many(sigs, pks, msgs, check_torsion=False, pk_inf=(~pks.any(axis=1)).astype(np.uint8))
    assert (st[[2, 4, 5]] == 0).all()                           # ordinary lanes (scalars reduced) verify


def test_keyed_records_and_compression_round_trip(engine, oracle):
    """SSA_FLAG_SIGN_KEYED: 130-byte KeyedSignature records pk(49) || sig(81) straight from the signer
    (sign_and_bind_pkey + KeyedSignature::to_bytes, src/signature.rs:132-156,237-245), both signers; the 49 bytes are
    PublicKey::to_bytes (src/public.rs:49-51) = the oracle's compression; the records verify through
    ssa_verify_keyed_many; ssa_compress_many inverts ssa_decompress_many"""
    rng = np.random.default_rng(4600)
    n = 300
    sks, nonces = make_scalars(rng, n), make_scalars(rng, n)
    msgs = rng.integers(0, 256, size=(n, 40), dtype=np.uint8)
    pks, sigs = engine.keygen_sign_many(sks, nonces, msgs)
    for ct in (False, True):
        pk2, recs = engine.keygen_sign_many(sks, nonces, msgs, constant_time=ct, keyed=True)
        assert recs.shape == (n, 130) and (pk2 == pks).all() and (recs[:, 49:] == sigs).all()
        for i in range(0, n, 7):
            assert recs[i, :49].tobytes() == oracle.compress(pks[i].tobytes())
        st, nf = engine.verify_keyed_many(recs, msgs, check_torsion=True)
        assert nf == 0
        recs[5, 60] ^= 1
        st, nf = engine.verify_keyed_many(recs, msgs, check_torsion=True)
        assert nf == 1 and st[5] == 2
    comp, st = engine.compress_many(pks)
    assert (st == 0).all() and (comp == recs[:, :49]).all()
    back, inf, dst = engine.decompress_many(comp)
    assert (dst == 0).all() and not inf.any() and (back == pks).all()
    # the identity and a non-canonical limb
    pk3 = pks[:4].copy()
    pk3[2, 8:16] = 0xFF                                          # limb 1 of x = 2^64 - 1 >= p
    inf3 = np.array([0, 1, 0, 0], np.uint8)
    comp3, st3 = engine.compress_many(pk3, pk_inf=inf3)
    assert list(st3) == [0, 0, 3, 0]
    assert comp3[1].tobytes() == bytes(48) + b"\x80" == oracle.compress(pk3[1].tobytes(), pk_inf=True)
    assert not comp3[2].any()
    # the object mirror goes through the same entry points
    import schnorr_sig_amd as ssa
    import random
    pyrng = random.Random(7)
    kp = ssa.KeyPair.new(lambda k: bytes(pyrng.getrandbits(8) for _ in range(k)), engine)
    ks = kp.sign_and_bind_pkey(b"keyed message", lambda k: bytes(pyrng.getrandbits(8) for _ in range(k)), engine)
    raw = ks.to_bytes(engine)
    assert len(raw) == 130 and raw[:49] == kp.public_key.to_bytes(engine) == oracle.compress(kp.public_key.affine)
    ks2 = ssa.KeyedSignature.from_bytes(raw, engine)
    assert ks2 is not None and ks2.verify(b"keyed message", engine) is None
    with pytest.raises(ssa.SignatureError):
        ks2.verify(b"another message", engine)


def test_one_hip_runtime_whatever_the_import_order():
    """the package first, torch afterwards (the order that once put two HIP runtimes into the process: "No HIP GPUs are
    available"), and the reverse: both see the device, and the package says which runtime it bound"""
    for order in ("import schnorr_sig_amd as ssa, torch", "import torch, schnorr_sig_amd as ssa"):
        code = order + "\n" + (
            "import numpy as np\n"
            "assert torch.cuda.is_available() and torch.cuda.device_count() >= 1\n"
            "eng = ssa.Engine(0)\n"
            "x = torch.arange(8, device='cuda').sum().item()\n"
            "s = np.ones((2, 32), np.uint8)\n"
            "pks, sigs = eng.keygen_sign_many(s, s, np.zeros((2, 8), np.uint8))\n"
            "print('BOUND', ssa.HIP_RUNTIME_BOUND, x, int(eng.verify_many(sigs, pks, np.zeros((2, 8), np.uint8))[1]))\n")
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        assert "BOUND" in r.stdout and " 28 0" in r.stdout, r.stdout
        bound = r.stdout.split("BOUND", 1)[1]
        assert ("libamdhip64.so" in bound) or ("torch was imported first" in bound), bound


@pytest.mark.parametrize("n", [3500, 4200])
def test_narrow_coefficients_are_the_value_their_signed_digits_represent(engine, oracle, n):
    """MSM form with 16-byte coefficients (what the library draws and what bench.py passes): the bucket method recodes a
    coefficient into signed digits over ITS OWN windows -- no carry window -- so a coefficient that carries out of its
    top window stands for raw - 2^128.  The shard record must be the oracle's sum s_i R_i - s_i h_i P_i / sum s_i e_i with
    exactly that effective coefficient (mod q) -- n = 3500 runs 8-bit windows, n = 4200 16-bit ones --, and forged
    batches are rejected whatever the top bits are."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import msm_records as mr
    rng = np.random.default_rng(4700 + n)
    sigs, pks, msgs = honest(engine, rng, n, msg_len=24)
    co = rng.integers(0, 256, size=(n, 16), dtype=np.uint8)
    co[0] = 0xFF                                   # 2^128 - 1: stands for -1
    co[1] = 0
    co[1, 15] = 0x80                               # 2^127: stands for -2^127
    co[2] = 0x7F                                   # 0x7f7f...7f: the largest value 8-bit windows take as it is
    co[3] = np.frombuffer(bytes.fromhex("ff7f" * 8), np.uint8)   # 0x7fff...7fff (little-endian): the same for 16-bit windows
    c = 16 if n >= 4096 else 8
    top = int.from_bytes(bytes([0xFF] * (c // 8 - 1) + [0x7F]) * (128 // c), "little")   # largest positive value of the digits
    eff = []
    for i in range(n):
        s = int.from_bytes(co[i].tobytes(), "little")
        eff.append(((s if s <= top else s - (1 << 128)) % Q).to_bytes(32, "little"))
    eff = np.frombuffer(b"".join(eff), np.uint8).reshape(n, 32)
    dev = torch.device("cuda", 0)
    ds, dp, dm, dc = (torch.from_numpy(a).to(dev) for a in (sigs, pks, msgs, co))
    rec = torch.zeros(24, dtype=torch.int64, device=dev)
    engine.verify_batch_msm_partial_device(ds.data_ptr(), dp.data_ptr(), dm.data_ptr(), n, 24, dc.data_ptr(), 16,
                                           rec.data_ptr())
    engine.sync()
    rec = rec.cpu().numpy().astype(np.uint64)
    m = 600                                        # the oracle's record of a prefix is enough to pin the rule ...
    rec_m = torch.zeros(24, dtype=torch.int64, device=dev)
    # ... but the prefix must take the same path as the whole: run it inside a batch of the same size by zeroing the
    # other coefficients (a zero coefficient contributes nothing to either side)
    co_m = co.copy()
    co_m[m:] = 0
    dcm = torch.from_numpy(co_m).to(dev)
    engine.verify_batch_msm_partial_device(ds.data_ptr(), dp.data_ptr(), dm.data_ptr(), n, 24, dcm.data_ptr(), 16,
                                           rec_m.data_ptr())
    engine.sync()
    rec_m = rec_m.cpu().numpy().astype(np.uint64)
    pt, lin, mal = mr.cpu_partial(oracle, sigs[:m], pks[:m], msgs[:m], eff[:m])
    assert not mal and int(rec_m[22]) == 0
    assert mr.record_point(oracle, rec_m) == pt and mr.record_lin(rec_m) == lin
    # the whole batch: the left-hand point equals [lin] G (honest batch), i.e. the combination says Ok; one forged
    # signature -- in a lane whose coefficient carries out, and in one whose does not -- says InvalidSignature
    assert engine.msm_combine(rec.reshape(1, 24)) == 0
    verdict = torch.full((1,), 255, dtype=torch.int32, device=dev)
    for lane in (0, 2, n - 1):
        bad = sigs.copy()
        bad[lane, 50] ^= 1
        db = torch.from_numpy(bad).to(dev)
        engine.verify_batch_msm_device(db.data_ptr(), dp.data_ptr(), dm.data_ptr(), n, 24, dc.data_ptr(), 16,
                                       verdict.data_ptr())
        engine.sync()
        assert int(verdict.item()) == 2, lane


# ---------------------------------------------------------------- the reference's own integration tests, name for name
def test_reference_integration_tests_through_the_mirror(engine):
    """tests/schnorr.rs of the reference through the Python mirror of its API (same names, same argument meaning, Option ->
    None, Result -> SignatureError): key_creation_and_conversion (:27-56), signing_and_verification_of_single_signature
    (:59-146) and batch_verification_of_three_signatures (:149-182), serde blocks aside."""
    import os as _os
    import schnorr_sig_amd as ssa
    rng = lambda k: _os.urandom(k)

    # key_creation_and_conversion
    sk = ssa.PrivateKey.new(rng)
    pk = ssa.PublicKey.from_private(sk, engine)
    kp = ssa.KeyPair.from_private(sk, engine)
    assert kp.private_key == sk and kp.public_key == pk
    assert ssa.KeyPair.from_bytes(kp.to_bytes(), engine) == kp
    seed = _os.urandom(64)
    assert ssa.KeyPair.from_seed(seed, engine).private_key == ssa.PrivateKey.from_seed(seed)
    assert ssa.PrivateKey.from_bytes(b"\0" * 32) is None and ssa.PrivateKey.from_bytes(b"\xff" * 32) is None

    # signing_and_verification_of_single_signature, first block: a bare private key signs
    message = b"A random message"
    signature = sk.sign(message, rng, engine)
    assert signature.verify(message, pk, engine) is None
    signature_bytes = signature.to_bytes()
    assert len(signature_bytes) == ssa.SIGNATURE_LENGTH
    assert signature == ssa.Signature.from_bytes(signature_bytes)
    keyed_signature = sk.sign_and_bind_pkey(message, rng, engine)
    assert keyed_signature.verify(message, engine) is None
    signature_bytes = keyed_signature.to_bytes(engine)
    assert len(signature_bytes) == ssa.KEYED_SIGNATURE_LENGTH
    assert keyed_signature == ssa.KeyedSignature.from_bytes(signature_bytes, engine)

    # second block: a key pair signs, every verifier agrees, every codec round-trips
    signer = ssa.KeyPair.new(rng, engine)
    signature = signer.sign(message, rng, engine)
    assert signature.verify(message, signer.public_key, engine) is None
    assert signer.verify_signature(signature, message) is None
    assert signer.public_key.verify_signature(signature, message) is None
    assert signer.sign_and_bind_pkey(message, rng, engine).verify(message, engine) is None
    private_key_bytes, public_key_bytes, keypair_bytes = signer.private_key.to_bytes(), signer.public_key.to_bytes(engine), signer.to_bytes()
    assert len(private_key_bytes) == ssa.PRIVATE_KEY_LENGTH and signer.private_key == ssa.PrivateKey.from_bytes(private_key_bytes)
    assert len(public_key_bytes) == ssa.PUBLIC_KEY_LENGTH and signer.public_key == ssa.PublicKey.from_bytes(public_key_bytes, engine)
    assert len(keypair_bytes) == ssa.KEY_PAIR_LENGTH and signer == ssa.KeyPair.from_bytes(keypair_bytes, engine)

    # batch_verification_of_three_signatures: signer 3 IS signer 1, the messages have three different lengths
    signer_1, signer_2 = ssa.KeyPair.new(rng, engine), ssa.KeyPair.new(rng, engine)
    signer_3 = signer_1
    messages = [b"A random message to sign", b"Another message to sign!", b"And once again another message from the others!!"]
    signers = [signer_1, signer_2, signer_3]
    signatures = [sg.sign(m, rng, engine) for sg, m in zip(signers, messages)]
    public_keys = [sg.public_key for sg in signers]
    for sig, m, k in zip(signatures, messages, public_keys):
        assert sig.verify(m, k, engine) is None
    assert ssa.verify_batch(signatures, public_keys, messages, rng, engine) is None
    assert ssa.verify_batch(signatures, public_keys, messages, rng, engine, msm=True) is None       # the reference's own algorithm

