"""GPU parity tests added in round 2 (all through the C ABI, all against the oracle):
identity public keys in every entry point, verify_batch's flag-byte semantics, the multi-device MSM-form verdict,
keyed contexts, the pipelined host-buffer path, full-coverage oracle comparisons on the BASELINE configs,
generator / scalar validation, and the optional upstream vectors."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 2**64 - 2**32 + 1
Q = 0x7AF2599B3B3F22D0563FBF0F990A37B5327AA72330157722D443623EAED4ACCF


def make_scalars(rng, n):
    s = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    s[:, 31] &= 0x3F          # < 2^254 < q
    s[:, 0] |= 1
    return s


def rand_felts(rng, shape):
    v = rng.integers(0, 2**64, size=shape, dtype=np.uint64)
    return np.where(v >= np.uint64(P), v - np.uint64(P), v)


def honest(engine, rng, n, msg_len=80):
    sks, nonces = make_scalars(rng, n), make_scalars(rng, n)
    msgs = rng.integers(0, 256, size=(n, msg_len), dtype=np.uint8)
    pks, sigs = engine.keygen_sign_many(sks, nonces, msgs)
    return sigs, pks, msgs, sks


def coeffs32(rng, n):
    c = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    c[:, 31] &= 0x3F
    return c


def fixture_small_order_key():
    import pymodel as m
    f = m.FIXTURE_SMALL_ORDER_PK
    return np.frombuffer(m.fp6_to_bytes48(f[0]) + m.fp6_to_bytes48(f[1]), dtype=np.uint8)


# ---------------------------------------------------------------- identity public key, every entry point
def test_identity_key_in_every_entry_point(engine, oracle):
    """The identity is a valid PublicKey (src/public.rs:95-101); verify_batch negates it and feeds it to the MSM
    (src/batch.rs:106) where it contributes nothing."""
    import schnorr_sig_amd as ssa
    rng = np.random.default_rng(501)
    n = 5
    sigs, pks, msgs, _ = honest(engine, rng, n, 24)
    inf = np.zeros(n, np.uint8)
    inf[2] = 1
    pks[2] = 0
    co = coeffs32(rng, n)
    # per-signature vector: the identity key passes the subgroup check, its signature cannot verify
    for torsion in (True, False):
        st, nf = engine.verify_many(sigs, pks, msgs, check_torsion=torsion, pk_inf=inf)
        want = oracle.verify_many(sigs, pks, msgs, check_torsion=torsion, pk_inf=inf)
        assert list(st) == list(want) == [0, 0, 2, 0, 0] and nf == 1
    # AND form and MSM form: InvalidSignature, not Malformed
    assert engine.verify_batch_status(sigs, pks, msgs, pk_inf=inf) == 2
    assert engine.verify_batch_msm(sigs, pks, msgs, coeffs=co, pk_inf=inf) == 2
    assert oracle.verify_batch_msm(sigs, pks, msgs, co, pk_inf=inf) == 2
    # without the marker the all-zero key is not a curve point: Malformed (documented divergence 6)
    assert engine.verify_batch_msm(sigs, pks, msgs, coeffs=co) == 3
    # a signature that DOES verify under the identity key: R = [e]G, any h (P = O contributes nothing)
    e = make_scalars(rng, 1)
    rpk, _ = engine.keygen_sign_many(e, e, np.zeros((1, 1), np.uint8))       # [e]G
    comp = oracle.compress(rpk[0].tobytes())
    sigs[2, :49] = np.frombuffer(comp, np.uint8)
    sigs[2, 49:] = e[0]
    for torsion in (True, False):
        st, nf = engine.verify_many(sigs, pks, msgs, check_torsion=torsion, pk_inf=inf)
        want = oracle.verify_many(sigs, pks, msgs, check_torsion=torsion, pk_inf=inf)
        assert list(st) == list(want) == [0] * 5 and nf == 0
    assert engine.verify_batch_status(sigs, pks, msgs, pk_inf=inf) == 0
    assert engine.verify_batch_msm(sigs, pks, msgs, coeffs=co, pk_inf=inf) == 0
    assert oracle.verify_batch_msm(sigs, pks, msgs, co, pk_inf=inf) == 0
    # the object mirror forwards is_identity (Signature.verify, verify_batch)
    ident = ssa.PublicKey(bytes(96), is_identity=True)
    assert ssa.Signature(sigs[2].tobytes()).verify(msgs[2].tobytes(), ident, engine=engine) is None
    with pytest.raises(ssa.SignatureError) as ei:
        ssa.Signature(sigs[1].tobytes()).verify(msgs[1].tobytes(), ident, engine=engine)
    assert ei.value.kind == "InvalidSignature"
    objs = [ssa.Signature(s.tobytes()) for s in sigs]
    keys = [ident if i == 2 else ssa.PublicKey(pks[i].tobytes()) for i in range(n)]
    assert ssa.verify_batch(objs, keys, [m.tobytes() for m in msgs], engine=engine) is None
    assert ssa.verify_batch(objs, keys, [m.tobytes() for m in msgs], engine=engine, msm=True) is None


# ---------------------------------------------------------------- flag byte of sig.x under verify_batch semantics
@pytest.mark.parametrize("mode", ["lane", "coop"])
def test_sig_flag_byte_semantics(engine, oracle, mode):
    """verify_batch decompresses R with its flag byte (src/batch.rs:104); Signature::verify ignores byte 48
    (src/signature.rs:186).  Both kernel families, against the oracle's restatement of either."""
    rng = np.random.default_rng(502)
    n = 12
    sigs, pks, msgs, _ = honest(engine, rng, n, 33)
    sigs[1, 48] ^= 0x40                      # wrong sort bit: R decodes to -R
    sigs[2, 48] |= 0x01                      # undecodable flag byte
    sigs[3, 48] |= 0x80                      # infinity bit on a non-zero x
    sigs[4, :48] = 0
    sigs[4, 48] = 0x80                       # the identity encoding: decodes, cannot verify
    sigs[5, :48] = 0
    sigs[5, 48] = 0xC0                       # identity with the sort bit: undecodable
    # an x with no curve point (the reference panics at from_compressed)
    while True:
        x = rand_felts(rng, 6)
        cand = np.concatenate([np.frombuffer(x.tobytes(), np.uint8), np.zeros(1, np.uint8)])
        if oracle.decompress(cand.tobytes()) is None:
            break
    sigs[6, :49] = cand
    sigs[7, :49] = sigs[8, :49]              # someone else's R: decodes, InvalidSignature
    st_b, nf_b = engine.verify_many(sigs, pks, msgs, check_torsion=False, mode=mode, sig_flag_byte=True)
    want_b = oracle.verify_many(sigs, pks, msgs, check_torsion=False, sig_flag_byte=True)
    assert list(st_b) == list(want_b)
    assert list(st_b[:9]) == [0, 2, 3, 3, 2, 3, 3, 2, 0] and nf_b == 7
    # Signature::verify semantics: byte 48 is ignored
    st_v, _ = engine.verify_many(sigs, pks, msgs, check_torsion=True, mode=mode)
    want_v = oracle.verify_many(sigs, pks, msgs, check_torsion=True)
    assert list(st_v) == list(want_v)
    assert list(st_v[:9]) == [0, 0, 0, 0, 2, 2, 2, 2, 0]
    # the AND-form batch verdict uses the batch semantics: an undecodable input dominates (the reference panics
    # before it compares anything), then InvalidSignature
    assert engine.verify_batch_status(sigs, pks, msgs) == 3
    ok = [0, 1, 7, 8, 9]
    assert engine.verify_batch_status(sigs[ok], pks[ok], msgs[ok]) == 2
    assert engine.verify_batch_status(sigs[[0, 8, 9]], pks[[0, 8, 9]], msgs[[0, 8, 9]]) == 0
    co = coeffs32(rng, n)
    assert engine.verify_batch_msm(sigs[ok], pks[ok], msgs[ok], coeffs=co[:5]) == \
        oracle.verify_batch_msm(sigs[ok], pks[ok], msgs[ok], co[:5]) == 2
    assert engine.verify_batch_msm(sigs, pks, msgs, coeffs=co) == oracle.verify_batch_msm(sigs, pks, msgs, co) == 3


def test_global_sign_divergence_of_the_msm_form(engine, oracle):
    """verify_batch compares x coordinates of the two SUMS (src/batch.rs:125-129): a batch whose every signature
    has e replaced by q - e satisfies sum s_i R_i - sum s_i h_i P_i = -[sum s_i e_i]G and is accepted by the
    reference's algorithm, while no single signature verifies (documented divergence; both forms reproduce it)."""
    rng = np.random.default_rng(503)
    n = 9
    sigs, pks, msgs, _ = honest(engine, rng, n)
    neg = sigs.copy()
    for i in range(n):
        e = int.from_bytes(sigs[i, 49:].tobytes(), "little")
        neg[i, 49:] = np.frombuffer(((Q - e) % Q).to_bytes(32, "little"), np.uint8)
    co = coeffs32(rng, n)
    assert oracle.verify_batch_msm(neg, pks, msgs, co) == 0
    assert engine.verify_batch_msm(neg, pks, msgs, coeffs=co) == 0
    st, nf = engine.verify_many(neg, pks, msgs, check_torsion=False)
    assert (st == 2).all() and nf == n and (oracle.verify_many(neg, pks, msgs, check_torsion=False) == 2).all()
    assert engine.verify_batch_status(neg, pks, msgs) == 2
    # a mixed batch (some negated, some not) fails in both forms
    mixed = sigs.copy()
    mixed[::2] = neg[::2]
    assert oracle.verify_batch_msm(mixed, pks, msgs, co) == engine.verify_batch_msm(mixed, pks, msgs, coeffs=co) == 2


# ---------------------------------------------------------------- MSM-form verdict across devices
@pytest.mark.parametrize("n", [2, 3, 64, 1000, 20000])
def test_multi_device_msm_verdict_equals_single_context(engine, oracle, n):
    """ssa_multi_verify_batch_msm: per-device partial sums, one point addition per shard and one compare on
    device 0; three contexts on the one device must agree with the single-context verdict."""
    import schnorr_sig_amd as ssa
    rng = np.random.default_rng(600 + n)
    sigs, pks, msgs, _ = honest(engine, rng, n)
    co = coeffs32(rng, n)
    multi = ssa.MultiEngine([0, 0, 0])
    try:
        assert multi.verify_batch_msm(sigs, pks, msgs, coeffs=co) == engine.verify_batch_msm(sigs, pks, msgs, coeffs=co) == 0
        assert multi.verify_batch_msm(sigs, pks, msgs) == 0                    # every device draws its own
        bad = sigs.copy()
        bad[n - 1, 50] ^= 4                                                    # corrupted: last shard
        assert multi.verify_batch_msm(bad, pks, msgs, coeffs=co) == engine.verify_batch_msm(bad, pks, msgs, coeffs=co) == 2
        flip = sigs.copy()
        flip[0, 48] ^= 0x40                                                    # flag flipped: first shard
        assert multi.verify_batch_msm(flip, pks, msgs, coeffs=co) == 2
        und = sigs.copy()
        und[n // 2, 48] |= 2                                                   # undecodable: middle shard
        assert multi.verify_batch_msm(und, pks, msgs, coeffs=co) == engine.verify_batch_msm(und, pks, msgs, coeffs=co) == 3
        if n <= 1000:
            assert oracle.verify_batch_msm(bad, pks, msgs, co) == 2 and oracle.verify_batch_msm(sigs, pks, msgs, co) == 0
        # identity key inside a shard, variable-length messages through offsets
        inf = np.zeros(n, np.uint8)
        inf[1] = 1
        flat = msgs.reshape(-1).copy()
        off = np.arange(n + 1, dtype=np.uint64) * 80
        assert multi.verify_batch_msm(sigs, pks, flat, offsets=off, coeffs=co, pk_inf=inf) == \
            engine.verify_batch_msm(sigs, pks, flat, offsets=off, coeffs=co, pk_inf=inf) == 2
    finally:
        multi.close()


# ---------------------------------------------------------------- keyed context
@pytest.mark.parametrize("kind", ["ladder", "comb"])
def test_keyed_context_64_keys_65536_signatures(engine, oracle, kind):
    """m = 64 keys x 2^16 signatures through ssa_keyset_create + ssa_verify_many_indexed, every lane against the
    oracle's Signature::verify; keys that fail their checks, an out-of-range index.  Both table kinds: the eight
    multiples of the ladder and the per-key comb (no doublings)."""
    rng = np.random.default_rng(700)
    m, n = 64, 1 << 16
    key_sks = make_scalars(rng, m)
    key_pks, _ = engine.keygen_sign_many(key_sks, key_sks, np.zeros((m, 1), np.uint8))
    idx = rng.integers(0, m, size=n).astype(np.uint32)
    nonces = make_scalars(rng, n)
    msgs = rng.integers(0, 256, size=(n, 80), dtype=np.uint8)
    pks, sigs = engine.keygen_sign_many(key_sks[idx], nonces, msgs)
    assert (pks == key_pks[idx]).all()
    # corrupt 1 %: e bit, message bit, someone else's R
    bad = rng.permutation(n)[: n // 100]
    sigs[bad[0::3], 49] ^= 1
    msgs[bad[1::3], 40] ^= 0x10
    sigs[bad[2::3], :49] = sigs[(bad[2::3] + 1) % n, :49]
    # keys 60..63: non-subgroup fixture, off-curve, non-canonical limb, the identity
    keys = key_pks.copy()
    inf = np.zeros(m, np.uint8)
    keys[60] = fixture_small_order_key()
    keys[61, 48] ^= 1
    keys[62, 8:16] = 0xFF
    keys[63] = 0
    inf[63] = 1
    ks = engine.keyset_create(keys, pk_inf=inf, kind=kind)
    try:
        assert list(engine.keyset_status(ks, m)[60:]) == [1, 3, 3, 0]
        assert (engine.keyset_status(ks, m)[:60] == 0).all()
        for torsion in (True, False):
            st, nf = engine.verify_many_indexed(ks, idx, sigs, msgs, check_torsion=torsion)
            want = oracle.verify_many(sigs, keys[idx], msgs, check_torsion=torsion, pk_inf=inf[idx])
            # off-curve keys: the oracle says Malformed too (divergence 6); everything is defined
            assert (st == want).all(), np.nonzero(st != want)[0][:10]
            assert nf == int((want != 0).sum())
        # the same answers as the un-keyed path
        st_plain, _ = engine.verify_many(sigs, keys[idx], msgs, check_torsion=True, pk_inf=inf[idx])
        st_keyed, _ = engine.verify_many_indexed(ks, idx, sigs, msgs, check_torsion=True)
        assert (st_plain == st_keyed).all()
        # index out of range -> Malformed, nothing else disturbed
        idx2 = idx[:1000].copy()
        idx2[7] = m
        idx2[8] = 0xFFFFFFFF
        st, _ = engine.verify_many_indexed(ks, idx2, sigs[:1000], msgs[:1000], check_torsion=True)
        assert st[7] == 3 and st[8] == 3
        keep = np.ones(1000, bool)
        keep[7:9] = False
        assert (st[keep] == st_keyed[:1000][keep]).all()
        # flag-byte semantics through the keyed kernel
        fl = sigs[:64].copy()
        fl[3, 48] ^= 0x40
        st, _ = engine.verify_many_indexed(ks, idx[:64], fl, msgs[:64], check_torsion=False, sig_flag_byte=True)
        want = oracle.verify_many(fl, keys[idx[:64]], msgs[:64], check_torsion=False, pk_inf=inf[idx[:64]],
                                  sig_flag_byte=True)
        assert (st == want).all()
    finally:
        engine.keyset_destroy(ks)


# ---------------------------------------------------------------- BASELINE configs, full oracle coverage
def test_config2_all_65536_digests_vs_oracle(engine, oracle):
    """byte-compare ALL 2^16 x 32 B with the CPU restatement."""
    rng = np.random.default_rng(0x5C4E0221)
    n = 1 << 16
    felts = rand_felts(rng, (n, 25))
    edge = np.array([0, 1, P - 1, 2**32 - 1, 2**32, P - 2**32], dtype=np.uint64)
    for r in range(64):
        felts[r] = np.roll(np.resize(edge, 25), r)
    got = engine.rescue_hash_many(felts)
    want = oracle.hash_field_many(felts)
    assert got.tobytes() == want.tobytes()


def test_config3_and_5_slice_and_every_corrupted_lane_vs_oracle(engine, oracle):
    """2^20 signatures, 1 % corrupted (config 5 = config 3's inputs + corruptions): the oracle recomputes a
    contiguous 2^16 slice and EVERY corrupted lane, both semantics; the rest is expected by construction."""
    rng = np.random.default_rng(0x5C4E0225)
    n = 1 << 20
    sigs, pks, msgs, _ = honest(engine, rng, n)
    nbad = n // 100
    bad = rng.permutation(n)[:nbad]
    f = fixture_small_order_key()
    kinds = np.arange(nbad) % 5
    sigs[bad[kinds == 0], 49] ^= 1
    msgs[bad[kinds == 1], 40] ^= 0x10
    src = (bad[kinds == 2] + 1) % n
    pks[bad[kinds == 2]] = pks[src]
    src = (bad[kinds == 3] + 1) % n
    sigs[bad[kinds == 3], :49] = sigs[src, :49]
    pks[bad[kinds == 4]] = f
    lo = 3 << 16
    sl = np.arange(lo, lo + (1 << 16))
    check = np.union1d(sl, bad)
    for torsion in (False, True):
        st, nf = engine.verify_many(sigs, pks, msgs, check_torsion=torsion)
        want = oracle.verify_many(sigs[check], pks[check], msgs[check], check_torsion=torsion)
        assert (st[check] == want).all()
        mask = np.ones(n, bool)
        mask[bad] = False
        assert (st[mask] == 0).all()                       # untouched lanes verify
        assert (st[bad] != 0).all() and nf == nbad         # every corrupted lane is rejected
        if torsion:
            assert (st[bad[kinds == 4]] == 1).all()
        else:
            assert (st != 1).all()


# ---------------------------------------------------------------- host-buffer path (pinned, chunked uploads)
def test_pipelined_host_path_matches_device_path(engine, oracle):
    import torch
    rng = np.random.default_rng(800)
    n = (1 << 18) + 12345                                   # ragged chunks
    sigs, pks, msgs, _ = honest(engine, rng, n)
    bad = rng.permutation(n)[:500]
    sigs[bad, 60] ^= 2
    inf = np.zeros(n, np.uint8)
    st_h, nf_h = engine.verify_many(sigs, pks, msgs, check_torsion=False, pk_inf=inf)      # pipelined (n >= 2^17)
    dev = torch.device("cuda", 0)
    ds, dp, dm = (torch.from_numpy(a).to(dev) for a in (sigs, pks, msgs))
    dst = torch.empty(n, dtype=torch.uint8, device=dev)
    dnf = torch.zeros(1, dtype=torch.int64, device=dev)
    engine.verify_many_device(ds.data_ptr(), dp.data_ptr(), dm.data_ptr(), n, 80, dst.data_ptr(), dnf.data_ptr())
    engine.sync()
    assert (st_h == dst.cpu().numpy()).all() and nf_h == int(dnf.item()) == 500
    assert (st_h[bad] == 2).all()
    samp = np.concatenate([bad[:200], np.arange(0, n, 997)])
    assert (st_h[samp] == oracle.verify_many(sigs[samp], pks[samp], msgs[samp], check_torsion=False)).all()
    # the MSM form from host buffers shares the upload + hash pipeline
    co = coeffs32(rng, n)
    assert engine.verify_batch_msm(sigs, pks, msgs, coeffs=co) == 2
    good = sigs.copy()
    good[bad, 60] ^= 2
    assert engine.verify_batch_msm(good, pks, msgs, coeffs=co) == 0
    assert engine.verify_batch_msm(good, pks, msgs) == 0                       # library-drawn coefficients
    # variable-length messages through offsets take the same path
    lens = rng.integers(0, 90, size=n)
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    flat = rng.integers(0, 256, size=int(off[-1]) + 1, dtype=np.uint8)
    m = 1 << 17
    sk2, no2 = make_scalars(rng, m), make_scalars(rng, m)
    pk2, sg2 = engine.keygen_sign_many(sk2, no2, flat, offsets=off[: m + 1])
    sg2[5, 49] ^= 1
    st2, nf2 = engine.verify_many(sg2, pk2, flat, offsets=off[: m + 1], check_torsion=True)
    assert nf2 == 1 and st2[5] == 2
    samp = np.arange(0, 4096)
    want = oracle.verify_many(sg2[samp], pk2[samp], flat, offsets=off[: 4097], check_torsion=True)
    assert (st2[samp] == want).all()


# ---------------------------------------------------------------- parameter blob / scalar validation
def test_generator_must_be_in_the_prime_subgroup():
    """ssa_ctx_create validates G on the device: off the curve, or on the curve outside the prime-order subgroup
    (the reference's own non-subgroup fixture, src/signature.rs:387-404) -> SSA_ERR_PARAMS."""
    import pymodel as m
    import schnorr_sig_amd as ssa
    blob = bytearray(ssa.Engine.default_params())
    gx_off = len(blob) - 96
    bad = bytearray(blob)
    bad[gx_off] ^= 1                                       # off the curve
    with pytest.raises(RuntimeError, match="invalid parameter blob"):
        ssa.Engine(0, params=bytes(bad))
    f = m.FIXTURE_SMALL_ORDER_PK
    bad = bytearray(blob)
    bad[gx_off:] = m.fp6_to_bytes48(f[0]) + m.fp6_to_bytes48(f[1])
    with pytest.raises(RuntimeError, match="invalid parameter blob"):
        ssa.Engine(0, params=bytes(bad))
    eng = ssa.Engine(0, params=bytes(blob))                # the default blob passed explicitly: still "default"
    assert eng.uses_default_params() is True
    eng.close()


def test_default_context_says_parity_is_unpinned(engine):
    assert engine.uses_default_params() is True


def test_keygen_sign_rejects_non_canonical_scalars(engine):
    rng = np.random.default_rng(900)
    sks, nonces = make_scalars(rng, 4), make_scalars(rng, 4)
    msgs = np.zeros((4, 8), np.uint8)
    engine.keygen_sign_many(sks, nonces, msgs)
    qb = np.frombuffer(Q.to_bytes(32, "little"), np.uint8)
    for which in ("sk", "nonce"):
        for val in (np.zeros(32, np.uint8), qb, np.full(32, 0xFF, np.uint8)):
            s2, n2 = sks.copy(), nonces.copy()
            (s2 if which == "sk" else n2)[2] = val
            with pytest.raises(RuntimeError, match="invalid argument"):
                engine.keygen_sign_many(s2, n2, msgs)
    # q - 1 is the largest canonical scalar
    s2 = sks.copy()
    s2[1] = np.frombuffer((Q - 1).to_bytes(32, "little"), np.uint8)
    engine.keygen_sign_many(s2, nonces, msgs)


def test_square_probes_run(engine):
    """fp_sqr with three multiplies against fp_mul(x, x): both chains run (rates are recorded by bench tools)."""
    assert engine.bench_fpmul(4) > 1e11 and engine.bench_fpmul(5) > 1e11


# ---------------------------------------------------------------- upstream vectors (closes "parity unpinned")
def test_upstream_vectors_if_present(oracle):
    """tests/golden/upstream_vectors.json does not exist in this repository: nothing in the container pins the
    Rescue constants or the generator.  The day someone drops the file in (format: tools/blob_from_upstream.py
    --help), this test runs upstream's digest, generator and signature through the oracle AND the HIP engine."""
    path = os.path.join(ROOT, "tests", "golden", "upstream_vectors.json")
    if not os.path.exists(path):
        pytest.skip("no upstream vectors in this repository (parity unpinned)")
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import blob_from_upstream as bfu
    import schnorr_sig_amd as ssa
    from oracle import Oracle
    vec = json.load(open(path))
    blob = bfu.build_blob(vec["constants"])
    orc = Oracle(blob=blob)
    eng = ssa.Engine(0, params=blob)
    try:
        assert eng.uses_default_params() is False
        for case in vec.get("hash_field", []):
            felts = np.array([int(x, 0) for x in case["input"]], dtype=np.uint64)
            want = bytes.fromhex(case["digest"])
            assert orc.hash_field(felts).tobytes() == want
            assert eng.rescue_hash_many(felts.reshape(1, -1)).tobytes() == want
        for case in vec.get("signatures", []):
            sig, pk49, msg = bytes.fromhex(case["signature"]), bytes.fromhex(case["public_key"]), bytes.fromhex(case["message"])
            dec = orc.decompress(pk49)
            assert dec is not None
            pk96, is_inf = dec
            pks_g, inf_g, st_g = eng.decompress_many(np.frombuffer(pk49, np.uint8))
            assert st_g[0] == 0 and pks_g[0].tobytes() == pk96
            want = 0 if case.get("valid", True) else 2
            assert orc.verify(sig, pk96, msg, check_torsion=True, pk_inf=is_inf) == want
            assert eng.verify_one(sig, pk96, msg, check_torsion=True, pk_is_identity=is_inf) == want
    finally:
        eng.close()


# ---------------------------------------------------------------- fused product + linear-term blocks
def test_fused_product_blocks(engine):
    """The Fp6 products whose following additions ride in the accumulator (fp6_asm.inc, fused linear terms): every
    block against plain integers, loose (non-canonical) operands and edge values included."""
    rng = np.random.default_rng(1234)
    edge = [0, 1, P - 1, P, P + 1, 2**32 - 1, 2**32, 2**64 - 1, 2**64 - 2**32, 2**63]
    n = 3000
    a = rng.integers(0, 2**64, size=(n, 12), dtype=np.uint64)
    b = rng.integers(0, 2**64, size=(n, 12), dtype=np.uint64)
    for r in range(len(edge) * len(edge)):
        a[r] = np.uint64(edge[r % len(edge)])
        b[r] = np.uint64(edge[r // len(edge)])
    a[200] = 0
    b[200] = np.uint64(2**64 - 1)
    a[201] = np.uint64(2**64 - 1)
    b[201] = 0

    def mulmod(u, v):
        t = [0] * 12
        for i, x in enumerate(u):
            for j, y in enumerate(v):
                t[i + j] += x * y
        return [(t[k] + 7 * t[k + 6]) % P for k in range(6)]

    ops = {8: lambda u, v, x, y: [(m - p_ - q) % P for m, p_, q in zip(mulmod(u, u), x, y)],
           9: lambda u, v, x, y: [(m + 3 * p_) % P for m, p_ in zip(mulmod(u, u), x)],
           10: lambda u, v, x, y: [(m - 4 * p_) % P for m, p_ in zip(mulmod(u, u), x)],
           11: lambda u, v, x, y: [(m - 8 * p_) % P for m, p_ in zip(mulmod(u, v), x)],
           12: lambda u, v, x, y: [(m - p_) % P for m, p_ in zip(mulmod(u, v), x)],
           13: lambda u, v, x, y: [(m - p_ - 2 * q) % P for m, p_, q in zip(mulmod(u, u), x, y)],
           14: lambda u, v, x, y: [(m + w) % P for m, w in zip(mulmod(u, v), mulmod(x, y))]}
    check = list(range(260)) + list(range(260, n, 13))
    for op, model in ops.items():
        got = engine.debug_arith(op, a, b, 6)
        for i in check:
            u, v = [int(t) for t in a[i, :6]], [int(t) for t in a[i, 6:]]
            x, y = [int(t) for t in b[i, :6]], [int(t) for t in b[i, 6:]]
            assert [int(t) for t in got[i]] == model(u, v, x, y), (op, i)


# ---------------------------------------------------------------- generated point operations on raw limbs
def test_generated_point_operations_on_raw_limbs(engine):
    """The asm statements of the ladder (jac_asm.inc: n doublings, mixed addition, one whole window) on the GPU with
    arbitrary loose 64-bit limbs, against the textbook formulas on plain integers.  Rows whose high words are all ones,
    or whose 7x / 3x / 21x multiples land next to 2^32, force every guarded pre-scaling site into its cold path on
    real hardware; rows with Z, x2 or H = 0 (mod p) in the first coefficient take the exceptional exit (window: flag 0,
    doublings done, addition left to the caller; jac_madd_fast: the exact compiled addition)."""
    rng = np.random.default_rng(77)

    def mulmod(u, v):
        t = [0] * 12
        for i, x in enumerate(u):
            for j, y in enumerate(v):
                t[i + j] += x * y
        return [(t[k] + 7 * t[k + 6]) % P for k in range(6)]

    sub = lambda u, v: [(a - b) % P for a, b in zip(u, v)]
    add = lambda u, v: [(a + b) % P for a, b in zip(u, v)]
    sc = lambda c, u: [c * a % P for a in u]

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

    n = 1536
    a = rng.integers(1, 2**64, size=(n, 20), dtype=np.uint64)
    b = rng.integers(1, 2**64, size=(n, 12), dtype=np.uint64)
    hi = np.uint64(0xFFFFFFFF) << np.uint64(32)
    a[256:512, :18] |= hi                      # high words all ones: the doublings' guards
    b[384:640] |= hi
    for r in range(512, 768):                  # c * a_hi within a few units of 2^32: the multiples' guards
        for c_ in range(18):
            c = (3, 7, 21)[(r + c_) % 3]
            h = (2**32 - int(rng.integers(1, 64))) * pow(c, -1, 2**32) % 2**32
            a[r, c_] = np.uint64((h << 32) | int(rng.integers(0, 2**32)))
    a[768:800, :18] = np.uint64(2**64 - 1)
    a[:, 18] = rng.integers(0, 2, size=n) * rng.integers(1, 9, size=n)        # act: zero on about half the rows
    a[:, 19] = 3                                                             # n doublings (uniform per launch)
    # exceptional inputs of the addition (first coefficient = 0 mod p) in a wave of their own: the statement's flag is
    # per wave (one lane with a possible exceptional input sends all its active lanes to the caller's exact addition)
    a[832:848, 12] = 0
    a[848:864, 12] = np.uint64(P)
    b[864:880, 0] = 0
    b[880:896, 0] = np.uint64(P)
    a[832:896, 18] = 1
    check = list(range(0, 256, 5)) + list(range(256, 900, 3)) + list(range(900, n, 11))
    got15 = engine.debug_arith(15, a, b, 19)
    got16 = engine.debug_arith(16, a, b, 19)
    got17 = engine.debug_arith(17, a, b, 19)
    n_cold_like = n_bail = 0
    for i in check:
        pt = [[int(t) for t in a[i, 6 * k:6 * k + 6]] for k in range(3)]
        q = [[int(t) for t in b[i, 6 * k:6 * k + 6]] for k in range(2)]
        act, nd = int(a[i, 18]), int(a[i, 19])
        d = pt
        for _ in range(nd):
            d = [list(v) for v in dbl(*d)]
        red = lambda row: [[int(t) % P for t in row[6 * k:6 * k + 6]] for k in range(3)]
        assert red(got17[i]) == d, ("dbl_n", i)
        want = d
        if act:
            m, H = madd(*d, *q)
            if 832 <= i < 896 or d[2][0] == 0 or q[0][0] % P == 0 or H[0] == 0:
                n_bail += 1        # (rows 864..895 have x2 = 0 in the first coefficient: their whole wave is handed back)
                assert int(got15[i, 18]) == 0 and red(got15[i]) == d, ("window exceptional", i)
                want = None
            else:
                want = [list(v) for v in m]
        if want is not None:
            assert int(got15[i, 18]) == 1 and red(got15[i]) == want, ("window", i)
        # the addition alone, with its exact fallback: generic rows against the formulas
        m, H = madd(*pt, *q)
        if not (pt[2][0] % P == 0 or q[0][0] % P == 0 or H[0] == 0):
            assert red(got16[i]) == [list(v) for v in m], ("madd", i)
        n_cold_like += i >= 256 and i < 800
    assert n_bail >= 4 and n_cold_like > 100
