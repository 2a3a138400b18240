"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Bit-exact everywhere (integer arithmetic)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P = 2**64 - 2**32 + 1
Q = 0x7AF2599B3B3F22D0563FBF0F990A37B5327AA72330157722D443623EAED4ACCF
EDGE = [0, 1, P - 1, 2**32 - 1, 2**32, P - 2**32, 2**63, P - 2, 7, 2**32 + 1]


def rand_felts(rng, shape):
    v = rng.integers(0, 2**64, size=shape, dtype=np.uint64)
    return np.where(v >= np.uint64(P), v - np.uint64(P), v)


def make_scalars(rng, n):
    s = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    s[:, 31] &= 0x3F          # < 2^254 < q, non-zero with overwhelming probability
    s[:, 0] |= 1
    return s


# ---------------------------------------------------------------- arithmetic probes
def test_fp_mul_and_inv(engine):
    rng = np.random.default_rng(1)
    a = np.concatenate([np.array(EDGE, dtype=np.uint64).repeat(len(EDGE)), rand_felts(rng, 4096)])
    b = np.concatenate([np.tile(np.array(EDGE, dtype=np.uint64), len(EDGE)), rand_felts(rng, 4096)])
    got = engine.debug_arith(5, a.reshape(-1, 1), b.reshape(-1, 1), 1)[:, 0]
    want = np.array([int(x) * int(y) % P for x, y in zip(a, b)], dtype=np.uint64)
    assert (got == want).all()
    # loose (non-canonical) inputs are legal inside the kernels
    loose = np.array([2**64 - 1, P, P + 5, 2**64 - 2**31], dtype=np.uint64)
    got = engine.debug_arith(5, loose.reshape(-1, 1), loose[::-1].copy().reshape(-1, 1), 1)[:, 0]
    want = np.array([int(x) * int(y) % P for x, y in zip(loose, loose[::-1])], dtype=np.uint64)
    assert (got == want).all()
    nz = a[a != 0]
    got = engine.debug_arith(6, nz.reshape(-1, 1), None, 1)[:, 0]
    want = np.array([pow(int(x), P - 2, P) for x in nz], dtype=np.uint64)
    assert (got == want).all()


def test_fp6_ops(engine, oracle):
    rng = np.random.default_rng(2)
    n = 512
    a = rand_felts(rng, (n, 6))
    b = rand_felts(rng, (n, 6))
    # edge rows: all-(p-1), sparse, zero
    a[0] = P - 1; b[0] = P - 1
    a[1] = 0; a[2] = [1, 0, 0, 0, 0, 0]; a[3] = [0, 0, 0, 0, 0, P - 1]; b[3] = [0, 0, 0, 0, 0, P - 1]
    got = engine.debug_arith(0, a, b, 6)
    want = np.stack([oracle.fp6_mul(a[i], b[i]) for i in range(n)])
    assert (got == want).all()
    got = engine.debug_arith(1, a, None, 6)
    want = np.stack([oracle.fp6_sqr(a[i]) for i in range(n)])
    assert (got == want).all()
    nz = a[np.any(a != 0, axis=1)][:128]
    got = engine.debug_arith(2, nz, None, 6)
    want = np.stack([oracle.fp6_inv(nz[i]) for i in range(nz.shape[0])])
    assert (got == want).all()


def _aff_rows(pts):
    rows = np.zeros((len(pts), 14), dtype=np.uint64)
    for i, p in enumerate(pts):
        if p is None:
            rows[i, 12] = 1
        else:
            rows[i, :6] = p[0]
            rows[i, 6:12] = p[1]
    return rows


def test_point_add_exceptional_cases(engine, oracle):
    """P+Q generic, P+P (doubling inside add), P+(-P) (identity), identity operands, and the
    order-2-ish fixture point: every branch of jac_madd / jac_add."""
    import pymodel as m
    g = m.default_params().generator()
    p2 = m.pt_mul(2, g)
    p3 = m.pt_mul(3, g)
    f = m.FIXTURE_SMALL_ORDER_PK
    cases = [(g, p2), (g, g), (g, m.pt_neg(g)), (None, g), (g, None), (p3, p2), (f, f), (f, m.pt_neg(f)),
             (f, g), (None, None)]
    for general in (0, 1):
        a = _aff_rows([c[0] for c in cases])
        b = _aff_rows([c[1] for c in cases])
        a[:, 13] = general
        got = engine.debug_arith(3, a, b, 13)
        for i, (x, y) in enumerate(cases):
            want = m.pt_add(x, y)
            if want is None:
                assert got[i, 12] == 1, (general, i)
            else:
                assert got[i, 12] == 0, (general, i)
                assert tuple(int(v) for v in got[i, :6]) == want[0], (general, i)
                assert tuple(int(v) for v in got[i, 6:12]) == want[1], (general, i)


def test_cooperative_point_operations(engine, oracle):
    """the wave-cooperative doubling / mixed addition / general addition (low-latency kernel, MSM tail):
    generic operands and every exceptional branch, operands given as scaled Jacobian points"""
    import pymodel as m
    g = m.default_params().generator()
    p2, p3, p7 = m.pt_mul(2, g), m.pt_mul(3, g), m.pt_mul(7, g)
    f = m.FIXTURE_SMALL_ORDER_PK
    o2 = m.SMALL_ORDER_POINTS[2]
    cases = [(g, p2), (g, g), (g, m.pt_neg(g)), (None, g), (g, None), (p3, p2), (p7, p3), (f, f), (f, m.pt_neg(f)),
             (f, g), (None, None), (o2, o2), (o2, g), (p7, p7)]
    for mode in (0, 1, 2):
        a = _aff_rows([c[0] for c in cases])
        b = _aff_rows([c[1] for c in cases])
        a[:, 13] = mode
        got = engine.debug_arith(7, a, b, 13)
        for i, (x, y) in enumerate(cases):
            want = m.pt_add(x, x) if mode == 2 else m.pt_add(x, y)
            if want is None:
                assert got[i, 12] == 1, (mode, i)
            else:
                assert got[i, 12] == 0, (mode, i)
                assert tuple(int(v) for v in got[i, :6]) == want[0], (mode, i)
                assert tuple(int(v) for v in got[i, 6:12]) == want[1], (mode, i)


def test_scalar_mul_table_path(engine, oracle):
    import pymodel as m
    rng = np.random.default_rng(3)
    g = m.default_params().generator()
    f = m.FIXTURE_SMALL_ORDER_PK
    ks = [0, 1, 2, 7, 8, 9, 15, 16, 17, Q - 1, Q, Q + 1, 2**255 - 1, 0x8888888888888888,
          int("7" + "8" * 63, 16), int("7" * 64, 16), int("f" * 63, 16)]
    ks += [int.from_bytes(rng.bytes(32), "little") >> 1 for _ in range(46)]
    pts = [g] * len(ks) + [f] * 8 + [None] * 2
    ks = ks + [Q, 5, 2 * Q // 5, Q // 5, 2, 3, 1, 0] + [5, 0]
    # points of order 2, 5, 10: multiples inside the 1P..8P table are the identity
    for o in (2, 5, 10):
        sm = list(range(0, 21)) + [Q, Q - 1, 2**255 - 19, int("7" + "8" * 63, 16)]
        pts += [m.SMALL_ORDER_POINTS[o]] * len(sm)
        ks += sm
    a = np.zeros((len(ks), 4), dtype=np.uint64)
    for i, k in enumerate(ks):
        a[i] = np.frombuffer(int(k).to_bytes(32, "little"), dtype=np.uint64)
    b = _aff_rows(pts)[:, :13].copy()
    got = engine.debug_arith(4, a, b, 13)
    for i, (k, p) in enumerate(zip(ks, pts)):
        want = oracle.point_mul(k, p)
        if want is None:
            assert got[i, 12] == 1, i
        else:
            assert got[i, 12] == 0, i
            assert tuple(int(v) for v in got[i, :6]) == want[0], i
            assert tuple(int(v) for v in got[i, 6:12]) == want[1], i


# ---------------------------------------------------------------- Rescue (config 2)
def test_rescue_hash_config2(engine, oracle):
    """2^16 rows of 25 felts incl. an edge-value slab; the oracle
    checks a strided sample of 2048 rows plus the whole edge slab."""
    rng = np.random.default_rng(0x5C4E0221)
    n = 1 << 16
    felts = rand_felts(rng, (n, 25))
    edge = np.array([0, 1, P - 1, 2**32 - 1, 2**32, P - 2**32], dtype=np.uint64)
    for r in range(64):
        felts[r] = np.roll(np.resize(edge, 25), r)
    got = engine.rescue_hash_many(felts)
    idx = np.concatenate([np.arange(64), np.arange(64, n, 32)])
    want = oracle.hash_field_many(felts[idx])
    assert (got[idx] == want).all()
    assert (got < np.uint64(P)).all()


@pytest.mark.parametrize("width", [0, 1, 7, 8, 9, 16, 17, 24])
def test_rescue_hash_ragged_widths(engine, oracle, width):
    rng = np.random.default_rng(width)
    felts = rand_felts(rng, (300, width)) if width else np.zeros((300, 0), dtype=np.uint64)
    got = engine.rescue_hash_many(felts)
    want = oracle.hash_field_many(felts) if width else np.stack([oracle.hash_field(np.zeros(0, np.uint64))] * 300)
    assert (got == want).all()


# ---------------------------------------------------------------- sign / hash_message / verify
@pytest.mark.parametrize("msg_len", [0, 1, 6, 7, 8, 13, 14, 24, 48, 80, 160])
def test_sign_hash_verify_message_lengths(engine, oracle, msg_len):
    rng = np.random.default_rng(100 + msg_len)
    n = 96
    sks, nonces = make_scalars(rng, n), make_scalars(rng, n)
    msgs = rng.integers(0, 256, size=(n, msg_len), dtype=np.uint8)
    pks, sigs = engine.keygen_sign_many(sks, nonces, msgs)
    pks_o, sigs_o = oracle.keygen_sign_many(sks, nonces, msgs)
    assert (pks == pks_o).all()
    assert (sigs == sigs_o).all()
    dig = engine.hash_message_many(sigs, pks, msgs)
    for i in range(0, n, 7):
        assert dig[i].tobytes() == oracle.hash_message(sigs[i, :48].tobytes(), pks[i].tobytes(), msgs[i].tobytes())
    for torsion in (False, True):
        st, nf = engine.verify_many(sigs, pks, msgs, check_torsion=torsion)
        assert nf == 0 and (st == 0).all()


def corrupt(rng, sigs, pks, msgs, frac=0.25):
    """config-5 style corruptions; returns the corrupted copies."""
    import pymodel as m
    sigs, pks, msgs = sigs.copy(), pks.copy(), msgs.copy()
    n = sigs.shape[0]
    idx = rng.permutation(n)[: max(5, int(n * frac))]
    f = m.FIXTURE_SMALL_ORDER_PK
    fbytes = np.frombuffer(m.fp6_to_bytes48(f[0]) + m.fp6_to_bytes48(f[1]), dtype=np.uint8)
    for k, i in enumerate(idx):
        kind = k % 5
        if kind == 0:
            sigs[i, 49] ^= 1                       # flip bit 0 of e (stays < q: top byte untouched)
        elif kind == 1 and msgs.shape[1] > 0:
            msgs[i, msgs.shape[1] // 2] ^= 0x10    # flip one message bit
        elif kind == 2:
            pks[i] = pks[(i + 1) % n]              # someone else's key
        elif kind == 3:
            sigs[i, :49] = sigs[(i + 1) % n, :49]  # someone else's R.x (on curve, canonical)
        else:
            pks[i] = fbytes                        # non-subgroup fixture key
    return sigs, pks, msgs, idx


def test_verify_corrupted_batch_vs_oracle(engine, oracle):
    rng = np.random.default_rng(0x5C4E0225)
    n = 1024
    sks, nonces = make_scalars(rng, n), make_scalars(rng, n)
    msgs = rng.integers(0, 256, size=(n, 80), dtype=np.uint8)
    pks, sigs = engine.keygen_sign_many(sks, nonces, msgs)
    sigs, pks, msgs, idx = corrupt(rng, sigs, pks, msgs)
    for torsion in (True, False):
        st, nf = engine.verify_many(sigs, pks, msgs, check_torsion=torsion)
        want = oracle.verify_many(sigs, pks, msgs, check_torsion=torsion)
        assert (st == want).all()
        assert nf == int((want != 0).sum())
        if torsion:   # every fixture key still in place -> InvalidPublicKey (checked first, :182)
            assert (st[idx[4::5]] == 1).all()
        else:         # batch semantics never return InvalidPublicKey (src/batch.rs)
            assert (st != 1).all()


def test_verify_malformed_and_edge_inputs(engine, oracle):
    rng = np.random.default_rng(9)
    n = 64
    sks, nonces = make_scalars(rng, n), make_scalars(rng, n)
    msgs = rng.integers(0, 256, size=(n, 24), dtype=np.uint8)
    pks, sigs = engine.keygen_sign_many(sks, nonces, msgs)
    sigs[0, 0:8] = 0xFF                 # limb >= p            -> malformed (reference panics)
    sigs[1, 49:81] = 0xFF               # e >= q               -> malformed
    pks[2, 8:16] = 0xFF                 # pk limb >= p         -> malformed
    sigs[3, :48] = 0; sigs[3, 48] = 0x80  # x = identity encoding -> InvalidSignature (src/signature.rs:408-417)
    sigs[4, 49:81] = 0                  # e = 0                -> InvalidSignature (:419-425)
    pks[5, 48] ^= 1                     # pk off the curve     -> malformed (documented divergence)
    st, nf = engine.verify_many(sigs, pks, msgs, check_torsion=True)
    assert list(st[:6]) == [3, 3, 3, 2, 2, 3]
    assert (st[6:] == 0).all() and nf == 6
    # oracle agrees on everything it defines (rows 0-4)
    want = oracle.verify_many(sigs, pks, msgs, check_torsion=True)
    assert (st[:5] == want[:5]).all() and (st[6:] == want[6:]).all()


def test_adversarial_e_hits_table_point(engine, oracle):
    """pk = [k]G known to the attacker, e chosen so that [h]P + [e]G lands exactly on / opposite to
    comb-table points mid-accumulation: exercises P == +-Q inside jac_madd on real verify inputs."""
    import pymodel as m
    prm = m.default_params()
    g = prm.generator()
    msg = b"adversarial"
    cases = []
    for k in (1, 2, 65536, Q - 1):
        pk = m.pt_mul(k, g)
        pk96 = m.fp6_to_bytes48(pk[0]) + m.fp6_to_bytes48(pk[1])
        rx = m.pt_mul(12345, g)[0]
        h = m.scalar_from_digest(m.hash_message(rx, pk, msg, prm))
        for target in (0, 1, 2, 65536, 65537, Q - 1, Q - 65536):
            e = (target - h * k) % Q          # [h]P + [e]G = [target]G
            sig = m.fp6_to_bytes48(rx) + b"\0" + e.to_bytes(32, "little")
            cases.append((sig, pk96))
    sigs = np.frombuffer(b"".join(c[0] for c in cases), dtype=np.uint8).reshape(-1, 81)
    pks = np.frombuffer(b"".join(c[1] for c in cases), dtype=np.uint8).reshape(-1, 96)
    msgs = np.tile(np.frombuffer(msg, dtype=np.uint8), (len(cases), 1))
    st, _ = engine.verify_many(sigs, pks, msgs, check_torsion=True)
    want = oracle.verify_many(sigs, pks, msgs, check_torsion=True)
    assert (st == want).all()


def test_small_order_public_keys(engine, oracle):
    """pk of order 2 / 5 / 10 (on the curve, outside the prime subgroup): InvalidPublicKey with the
    torsion check; without it (batch semantics) the ladder must still be exact although multiples
    in the per-lane table are the identity."""
    import pymodel as m
    rng = np.random.default_rng(21)
    sigs, pks, msgs = [], [], []
    for o in (2, 5, 10):
        p = m.SMALL_ORDER_POINTS[o]
        pk96 = m.fp6_to_bytes48(p[0]) + m.fp6_to_bytes48(p[1])
        for t in range(12):
            rx = m.pt_mul(1000 + t, m.default_params().generator())[0]
            e = int.from_bytes(rng.bytes(32), "little") % Q
            if t == 0:
                e = 0
            sigs.append(m.fp6_to_bytes48(rx) + b"\0" + e.to_bytes(32, "little"))
            pks.append(pk96)
            msgs.append(bytes(rng.integers(0, 256, size=24, dtype=np.uint8)))
    sigs = np.frombuffer(b"".join(sigs), dtype=np.uint8).reshape(-1, 81)
    pks = np.frombuffer(b"".join(pks), dtype=np.uint8).reshape(-1, 96)
    msgs = np.frombuffer(b"".join(msgs), dtype=np.uint8).reshape(-1, 24)
    for torsion in (True, False):
        st, _ = engine.verify_many(sigs, pks, msgs, check_torsion=torsion)
        want = oracle.verify_many(sigs, pks, msgs, check_torsion=torsion)
        assert (st == want).all()
        assert (st == (1 if torsion else 2)).all()
    # and a forged acceptance: with pk of order 2, [h]P is O or P; pick e so that R matches
    g = m.default_params().generator()
    p2 = m.SMALL_ORDER_POINTS[2]
    pk96 = m.fp6_to_bytes48(p2[0]) + m.fp6_to_bytes48(p2[1])
    forged = []
    for e in range(1, 9):
        for addp in (False, True):
            r = m.pt_mul(e, g)
            if addp:
                r = m.pt_add(r, p2)
            forged.append(m.fp6_to_bytes48(r[0]) + b"\0" + e.to_bytes(32, "little"))
    fs = np.frombuffer(b"".join(forged), dtype=np.uint8).reshape(-1, 81)
    fp = np.tile(np.frombuffer(pk96, dtype=np.uint8), (len(forged), 1))
    fm = np.zeros((len(forged), 5), dtype=np.uint8)
    st, _ = engine.verify_many(fs, fp, fm, check_torsion=False)
    want = oracle.verify_many(fs, fp, fm, check_torsion=False)
    assert (st == want).all() and (st == 0).sum() >= len(forged) // 4   # roughly half are accepted


def test_identity_public_key(engine, oracle):
    """sk = 0 is not constructible through PrivateKey::new, but PublicKey(identity) exists
    (src/public.rs:95-101); with pk = O a signature with e = r verifies."""
    rng = np.random.default_rng(11)
    nonces = make_scalars(rng, 4)
    msgs = rng.integers(0, 256, size=(4, 16), dtype=np.uint8)
    # R = [r]G via keygen (pk output of keygen with sk = r)
    rpk, _ = engine.keygen_sign_many(nonces, nonces, msgs)
    pks = np.zeros((4, 96), dtype=np.uint8)
    sigs = np.zeros((4, 81), dtype=np.uint8)
    sigs[:, :48] = rpk[:, :48]
    sigs[:, 49:] = nonces
    inf = np.ones(4, dtype=np.uint8)
    st, nf = engine.verify_many(sigs, pks, msgs, check_torsion=True, pk_inf=inf)
    want = oracle.verify_many(sigs, pks, msgs, check_torsion=True, pk_inf=inf)
    assert (st == want).all() and (st == 0).all() and nf == 0


def test_variable_length_messages_offsets(engine, oracle):
    rng = np.random.default_rng(12)
    lens = [0, 1, 6, 7, 8, 13, 14, 24, 48, 80, 160, 3, 29, 70, 77, 200]
    n = len(lens)
    sks, nonces = make_scalars(rng, n), make_scalars(rng, n)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    flat = rng.integers(0, 256, size=int(off[-1]) + 1, dtype=np.uint8)
    pks, sigs = engine.keygen_sign_many(sks, nonces, flat, offsets=off)
    pks_o, sigs_o = oracle.keygen_sign_many(sks, nonces, flat, offsets=off)
    assert (pks == pks_o).all() and (sigs == sigs_o).all()
    st, nf = engine.verify_many(sigs, pks, flat, offsets=off, check_torsion=True)
    assert nf == 0 and (st == 0).all()
    flat2 = flat.copy()
    flat2[int(off[9])] ^= 1
    st, nf = engine.verify_many(sigs, pks, flat2, offsets=off, check_torsion=True)
    assert nf == 1 and st[9] == 2


def test_empty_batch_and_batch_verdict(engine):
    import schnorr_sig_amd as ssa
    assert engine.verify_batch_status(np.zeros((0, 81), np.uint8), np.zeros((0, 96), np.uint8),
                                      np.zeros((0, 8), np.uint8)) == ssa.OK
    st, nf = engine.verify_many(np.zeros((0, 81), np.uint8), np.zeros((0, 96), np.uint8),
                                np.zeros((0, 8), np.uint8))
    assert st.size == 0 and nf == 0


# ---------------------------------------------------------------- golden fixtures through the HIP path
def _gold():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors.json")))


def test_golden_vectors_through_hip(engine):
    g = _gold()
    a = np.array([r["a"] for r in g["fp6"]], dtype=np.uint64)
    b = np.array([r["b"] for r in g["fp6"]], dtype=np.uint64)
    assert (engine.debug_arith(0, a, b, 6) == np.array([r["mul"] for r in g["fp6"]], dtype=np.uint64)).all()
    assert (engine.debug_arith(1, a, None, 6) == np.array([r["sqr"] for r in g["fp6"]], dtype=np.uint64)).all()
    for rec in g["hash_field"]:
        if rec["in"]:
            got = engine.rescue_hash_many(np.array([rec["in"]], dtype=np.uint64))
            assert [int(v) for v in got[0]] == rec["digest"]
    for rec in g["signatures"]:
        sk = np.frombuffer(bytes.fromhex(rec["sk"]), dtype=np.uint8)
        nonce = np.frombuffer(bytes.fromhex(rec["nonce"]), dtype=np.uint8)
        msg = bytes.fromhex(rec["msg"])
        flat = np.frombuffer(msg + b"\0", dtype=np.uint8)
        off = np.array([0, len(msg)], dtype=np.uint64)
        pks, sigs = engine.keygen_sign_many(sk, nonce, flat, offsets=off)
        assert pks[0].tobytes().hex() == rec["pk"] and sigs[0].tobytes().hex() == rec["sig"]
        assert engine.hash_message_many(sigs, pks, flat, offsets=off)[0].tobytes().hex() == rec["digest"]
        assert engine.verify_one(sigs[0].tobytes(), pks[0].tobytes(), msg, check_torsion=True) == 0
        for neg in rec["negative"]:
            st = engine.verify_one(bytes.fromhex(neg["sig"]), bytes.fromhex(neg["pk"]), bytes.fromhex(neg["msg"]), True)
            assert st == neg["status"], neg["what"]
            if "status_no_torsion" in neg:
                assert engine.verify_one(bytes.fromhex(neg["sig"]), bytes.fromhex(neg["pk"]),
                                         bytes.fromhex(neg["msg"]), False) == neg["status_no_torsion"]


def test_reference_api_mirror_roundtrip(engine):
    """KeyPair::new / sign / verify_signature / verify_batch through the object mirror, in the shape
    of tests/schnorr.rs:59-146,149-182 and src/batch.rs:152-179 of the reference."""
    import os as _os
    import schnorr_sig_amd as ssa
    rng = lambda k: _os.urandom(k)
    kp = ssa.KeyPair.new(rng, engine)
    sig = kp.sign(b"A random message", rng, engine)
    assert sig.verify(b"A random message", kp.public_key, engine) is None
    assert ssa.Signature.from_bytes(sig.to_bytes()) == sig
    with pytest.raises(ssa.SignatureError) as ei:
        sig.verify(b"A random messagf", kp.public_key, engine)
    assert ei.value.kind == "InvalidSignature"
    import pymodel as m
    f = m.FIXTURE_SMALL_ORDER_PK
    wrong = ssa.PublicKey(m.fp6_to_bytes48(f[0]) + m.fp6_to_bytes48(f[1]))
    with pytest.raises(ssa.SignatureError) as ei:
        sig.verify(b"A random message", wrong, engine)
    assert ei.value.kind == "InvalidPublicKey" and repr(ei.value) == "Err(InvalidPublicKey)"
    msgs = [b"Message1", b"Message2", b"Message3", b"Message4", b"Message5"]
    kps = [ssa.KeyPair.new(rng, engine) for _ in range(3)]
    kps += [kps[0], kps[0]]
    sigs = [k.sign(mm, rng, engine) for k, mm in zip(kps, msgs)]
    pks = [k.public_key for k in kps]
    assert ssa.verify_batch(sigs, pks, msgs, rng, engine) is None
    pks[1], pks[2] = pks[2], pks[1]
    with pytest.raises(ssa.SignatureError):
        ssa.verify_batch(sigs, pks, msgs, rng, engine)


# ---------------------------------------------------------------- BASELINE sizes (configs 3 and 5)
def test_full_size_2pow20_with_one_percent_corruption(engine, oracle):
    """2^20 signatures, 1% corrupted (SURVEY.md §8(d) configs 3+5): the expected accept/reject vector is
    known by construction; every corrupted lane and a random sample of honest lanes are also
    recomputed by the CPU oracle.  Size-independent property: n_fail == #corrupted, status is
    idempotent across runs and independent of the torsion flag except on the non-subgroup keys."""
    import pymodel as m
    rng = np.random.default_rng(0x5C4E0225)
    n = 1 << 20
    sks, nonces = make_scalars(rng, n), make_scalars(rng, n)
    msgs = rng.integers(0, 256, size=(n, 80), dtype=np.uint8)
    pks, sigs = engine.keygen_sign_many(sks, nonces, msgs)
    nbad = n // 100
    idx = rng.permutation(n)[:nbad]
    f = m.FIXTURE_SMALL_ORDER_PK
    fbytes = np.frombuffer(m.fp6_to_bytes48(f[0]) + m.fp6_to_bytes48(f[1]), dtype=np.uint8)
    osigs, opks = sigs.copy(), pks.copy()
    kinds = np.arange(nbad) % 5
    sigs[idx[kinds == 0], 49] ^= 1
    msgs[idx[kinds == 1], 40] ^= 0x10
    pks[idx[kinds == 2]] = opks[(idx[kinds == 2] + 1) % n]
    sigs[idx[kinds == 3], :49] = osigs[(idx[kinds == 3] + 1) % n, :49]
    pks[idx[kinds == 4]] = fbytes
    expect_t = np.zeros(n, dtype=np.uint8)
    expect_t[idx] = 2
    expect_t[idx[kinds == 4]] = 1
    expect_b = np.where(expect_t != 0, 2, 0).astype(np.uint8)
    st_t, nf_t = engine.verify_many(sigs, pks, msgs, check_torsion=True)
    st_b, nf_b = engine.verify_many(sigs, pks, msgs, check_torsion=False)
    assert (st_t == expect_t).all() and nf_t == nbad
    assert (st_b == expect_b).all() and nf_b == nbad
    st_again, _ = engine.verify_many(sigs, pks, msgs, check_torsion=True)
    assert (st_again == st_t).all()
    sample = np.concatenate([idx[:2000], rng.integers(0, n, size=2000)])
    want = oracle.verify_many(sigs[sample], pks[sample], msgs[sample], check_torsion=True)
    assert (st_t[sample] == want).all()
    # keygen/sign kernel vs CPU signer on a 2^12 prefix (byte-exact)
    pk_o, sig_o = oracle.keygen_sign_many(sks[:4096], nonces[:4096], rng.integers(0, 1, size=(4096, 0), dtype=np.uint8))
    assert (pk_o == opks[:4096]).all()


def test_cpp_host_mirror(tmp_path):
    """The C++ mirror of the reference API (schnorr-sig_amd/host/schnorr_sig.hpp) over the C ABI."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "schnorr-sig_amd", "csrc")
    exe = str(tmp_path / "host_api_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(root, "tests", "csrc", "host_api_test.cpp"),
                           "-L" + libdir, "-lschnorr_sig_amd", "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "host_api_test ok" in out.stdout, out.stdout + out.stderr


def test_alternative_parameter_blobs(oracle):
    """The Rescue instance and sponge layout are data: a blob with 8 rounds, full-width MDS
    entries (generic multiplier path), capacity-first layout and Rescue-Prime padding must give
    the oracle's digests and signatures with no kernel change (how upstream's real constants
    would be dropped in)."""
    import struct
    import schnorr_sig_amd as ssa
    from oracle import Oracle
    base = bytearray(ssa.Engine.default_params())
    rng = np.random.default_rng(99)
    variants = []
    b = bytearray(base)                                  # (1) generic MDS entries, 8 rounds
    struct.pack_into("<IIiIII", b, 8, 8, 0, 11, 0, 0, 0)
    mds = rand_felts(rng, 144)
    struct.pack_into("<144Q", b, 32, *[int(v) for v in mds])
    struct.pack_into("<12Q", b, 1184 + 96 * 7, *[int(v) for v in rand_felts(rng, 12)])
    struct.pack_into("<12Q", b, 1952 + 96 * 7, *[int(v) for v in rand_felts(rng, 12)])
    variants.append(bytes(b))
    b = bytearray(base)                                  # (2) capacity-first, padded sponge, digest = state[4..8]
    struct.pack_into("<IIiIII", b, 8, 7, 4, -1, 1, 4, 0)
    variants.append(bytes(b))
    b = bytearray(base)                                  # (3) length in state[0], rate = state[4..12]
    struct.pack_into("<IIiIII", b, 8, 7, 4, 0, 0, 4, 0)
    variants.append(bytes(b))
    b = bytearray(base)                                  # (4) MDS entries of 17..32 bits: the 32-bit-multiplier path
    struct.pack_into("<144Q", b, 32, *[int(v) for v in rng.integers(2**16, 2**32, size=144)])     # (the default
    variants.append(bytes(b))                                                  # blob's entries take the carry-free one)
    for blob in variants:
        orc = Oracle(blob=blob)
        eng = ssa.Engine(0, params=blob)
        try:
            for width in (0, 5, 8, 13, 25):
                felts = rand_felts(rng, (64, width)) if width else np.zeros((64, 0), dtype=np.uint64)
                want = orc.hash_field_many(felts) if width else np.stack([orc.hash_field(np.zeros(0, np.uint64))] * 64)
                assert (eng.rescue_hash_many(felts) == want).all()
            n = 64
            sks, nonces = make_scalars(rng, n), make_scalars(rng, n)
            msgs = rng.integers(0, 256, size=(n, 33), dtype=np.uint8)
            pks, sigs = eng.keygen_sign_many(sks, nonces, msgs)
            pks_o, sigs_o = orc.keygen_sign_many(sks, nonces, msgs)
            assert (pks == pks_o).all() and (sigs == sigs_o).all()
            sigs[5, 50] ^= 2
            st, nf = eng.verify_many(sigs, pks, msgs, check_torsion=True)
            assert nf == 1 and st[5] == 2 and (orc.verify_many(sigs, pks, msgs, check_torsion=True) == st).all()
        finally:
            eng.close()
    Oracle()   # restore the default blob in the shared library's global state


def test_invalid_parameter_blob_is_rejected():
    import schnorr_sig_amd as ssa
    bad = bytearray(ssa.Engine.default_params())
    bad[0] = ord("X")
    with pytest.raises(RuntimeError, match="invalid parameter blob"):
        ssa.Engine(0, params=bytes(bad))
    bad = bytearray(ssa.Engine.default_params())
    bad[32:40] = b"\xff" * 8          # MDS entry >= p
    with pytest.raises(RuntimeError, match="invalid parameter blob"):
        ssa.Engine(0, params=bytes(bad))


# ---------------------------------------------------------------- decompression
def test_decompress_many_vs_oracle(engine, oracle):
    """PublicKey::from_bytes / AffinePoint::from_compressed: random keys round-trip, both sort flags,
    x values off the curve, and the reference's encoding fixtures (src/public.rs:95-156)."""
    import pymodel as m
    import schnorr_sig_amd as ssa
    rng = np.random.default_rng(31)
    n = 2048
    sks = make_scalars(rng, n)
    pks, _ = engine.keygen_sign_many(sks, sks, np.zeros((n, 1), np.uint8))
    comp = np.zeros((n, 49), dtype=np.uint8)
    for i in range(n):
        comp[i] = np.frombuffer(ssa.PublicKey(pks[i].tobytes()).to_bytes(), dtype=np.uint8)
    out, inf, st = engine.decompress_many(comp)
    assert (st == 0).all() and (inf == 0).all() and (out == pks).all()
    for i in range(0, n, 97):
        assert oracle.compress(pks[i].tobytes()) == comp[i].tobytes()
    flipped = comp.copy()
    flipped[:, 48] ^= 0x40                      # the other root: same x, y negated
    out2, _, st2 = engine.decompress_many(flipped)
    assert (st2 == 0).all() and (out2[:, :48] == pks[:, :48]).all() and (out2[:, 48:] != pks[:, 48:]).any()
    for i in range(0, n, 211):
        assert oracle.decompress(flipped[i].tobytes())[0] == out2[i].tobytes()
    # random x: about half are not on the curve
    rx = rand_felts(rng, (1024, 6))
    rc = np.zeros((1024, 49), dtype=np.uint8)
    rc[:, :48] = rx.view(np.uint8).reshape(1024, 48)
    rc[:, 48] = rng.integers(0, 2, size=1024, dtype=np.uint8) * 0x40
    out3, inf3, st3 = engine.decompress_many(rc)
    n_ok = 0
    for i in range(1024):
        want = oracle.decompress(rc[i].tobytes())
        assert (st3[i] == 0) == (want is not None)
        if want is not None:
            n_ok += 1
            assert out3[i].tobytes() == want[0]
    assert 350 < n_ok < 700
    # encoding fixtures of the reference
    fixtures = [bytes(48) + b"\x80",                      # identity (src/public.rs:95-101)          ok, inf
                bytes(49),                                # 49 zero bytes (:115-120)                   None
                b"\xff" * 49,                             # (:122-129)                                 None
                comp[0, :48].tobytes() + b"\xff",         # flag byte 255 (:150-156)                   None
                bytes(48) + b"\xc0",                      # infinity with the sort flag                None
                bytes([1]) + bytes(47) + b"\x80"]         # infinity with x != 0                       None
    fx = np.frombuffer(b"".join(fixtures), dtype=np.uint8).reshape(-1, 49)
    _, inf4, st4 = engine.decompress_many(fx)
    assert list(st4) == [0, 1, 1, 1, 1, 1] and inf4[0] == 1
    pk = ssa.PublicKey.from_bytes(fixtures[0], engine)
    assert pk is not None and pk.is_identity and pk.to_bytes() == fixtures[0]
    assert ssa.PublicKey.from_bytes(fixtures[1], engine) is None
    rt = ssa.PublicKey.from_bytes(comp[5].tobytes(), engine)
    assert rt == ssa.PublicKey(pks[5].tobytes()) and rt.to_bytes() == comp[5].tobytes()


# ---------------------------------------------------------------- MSM-form verify_batch
def _batch(engine, rng, n, msg_len=80):
    sks, nonces = make_scalars(rng, n), make_scalars(rng, n)
    msgs = rng.integers(0, 256, size=(n, msg_len), dtype=np.uint8)
    pks, sigs = engine.keygen_sign_many(sks, nonces, msgs)
    coeffs = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    coeffs[:, 31] &= 0x3F
    return sigs, pks, msgs, coeffs


@pytest.mark.parametrize("n", [1, 2, 3, 17, 64, 300, 1024, 5000])
def test_verify_batch_msm_vs_oracle(engine, oracle, n):
    """The reference's own batch algorithm (src/batch.rs:56-130) on the GPU against its C restatement,
    same coefficients: honest batch, swapped keys (src/batch.rs:177-178), one corrupted scalar."""
    rng = np.random.default_rng(500 + n)
    sigs, pks, msgs, coeffs = _batch(engine, rng, n)
    assert engine.verify_batch_msm(sigs, pks, msgs, coeffs=coeffs) == 0
    if n <= 1024:
        assert oracle.verify_batch_msm(sigs, pks, msgs, coeffs) == 0
    assert engine.verify_batch_msm(sigs, pks, msgs) == 0                  # library-drawn 128-bit coefficients
    bad = sigs.copy()
    bad[n // 2, 55] ^= 8
    assert engine.verify_batch_msm(bad, pks, msgs, coeffs=coeffs) == 2
    assert engine.verify_batch_msm(bad, pks, msgs) == 2
    if n <= 300:
        assert oracle.verify_batch_msm(bad, pks, msgs, coeffs) == 2
    if n >= 3:
        sw = pks.copy()
        sw[[1, 2]] = sw[[2, 1]]
        assert engine.verify_batch_msm(sigs, sw, msgs, coeffs=coeffs) == 2


def test_verify_batch_msm_golden_batch5_and_repeated_keys(engine, oracle):
    g = _gold()["batch5"]
    sigs = np.frombuffer(bytes.fromhex("".join(g["sigs"])), dtype=np.uint8).reshape(5, 81)
    pks = np.frombuffer(bytes.fromhex("".join(g["pks"])), dtype=np.uint8).reshape(5, 96)
    coeffs = np.frombuffer(bytes.fromhex("".join(g["coeffs"])), dtype=np.uint8).reshape(5, 32)
    msgs = [bytes.fromhex(x) for x in g["msgs"]]
    flat = np.frombuffer(b"".join(msgs) + b"\0", dtype=np.uint8)
    off = np.cumsum([0] + [len(x) for x in msgs]).astype(np.uint64)
    assert engine.verify_batch_msm(sigs, pks, flat, offsets=off, coeffs=coeffs) == g["status"] == 0
    sw = pks.copy()
    sw[[1, 2]] = sw[[2, 1]]
    assert engine.verify_batch_msm(sigs, sw, flat, offsets=off, coeffs=coeffs) == g["status_swapped_1_2"] == 2
    # many signatures under ONE key and equal coefficients: equal points meet in one bucket (doubling inside jac_madd)
    rng = np.random.default_rng(77)
    n = 256
    sk = make_scalars(rng, 1).repeat(n, axis=0)
    nonces = make_scalars(rng, n)
    m2 = rng.integers(0, 256, size=(n, 12), dtype=np.uint8)
    pk2, sg2 = engine.keygen_sign_many(sk, nonces, m2)
    c2 = np.tile(make_scalars(rng, 1), (n, 1))
    assert engine.verify_batch_msm(sg2, pk2, m2, coeffs=c2) == oracle.verify_batch_msm(sg2, pk2, m2, c2) == 0
    sg2[9, 49] ^= 1
    assert engine.verify_batch_msm(sg2, pk2, m2, coeffs=c2) == oracle.verify_batch_msm(sg2, pk2, m2, c2) == 2


def test_verify_batch_msm_divergence_classes(engine, oracle):
    """Where the MSM form differs from n x Signature::verify: the flag byte of sig.x matters
    (class iii), an x off the curve panics in the reference (class ii -> SSA_MALFORMED), no torsion check."""
    rng = np.random.default_rng(88)
    sigs, pks, msgs, coeffs = _batch(engine, rng, 40, 24)
    flipped = sigs.copy()
    flipped[7, 48] ^= 0x40                                   # R -> -R: per-lane verify is x-only and still accepts
    st, _ = engine.verify_many(flipped, pks, msgs, check_torsion=True)
    assert (st == 0).all()
    assert engine.verify_batch_msm(flipped, pks, msgs, coeffs=coeffs) == 2
    assert oracle.verify_batch_msm(flipped, pks, msgs, coeffs) == 2
    off = sigs.copy()
    off[3, :8] = np.frombuffer((12345).to_bytes(8, "little"), dtype=np.uint8)   # canonical x, (almost surely) no point
    want = oracle.verify_batch_msm(off, pks, msgs, coeffs)
    got = engine.verify_batch_msm(off, pks, msgs, coeffs=coeffs)
    assert got == want and got in (2, 3)
    bad = sigs.copy()
    bad[5, 0:8] = 0xFF                                        # non-canonical limb: unwrap panics (src/batch.rs:67)
    assert engine.verify_batch_msm(bad, pks, msgs, coeffs=coeffs) == 3
    assert engine.verify_batch_msm(np.zeros((0, 81), np.uint8), np.zeros((0, 96), np.uint8),
                                   np.zeros((0, 4), np.uint8)) == 0


def test_verify_batch_msm_full_size(engine):
    """2^20 signatures through the MSM form: honest batch accepts, a single corrupted signature rejects."""
    rng = np.random.default_rng(0x5C4E0222)
    n = 1 << 20
    sigs, pks, msgs, coeffs = _batch(engine, rng, n)
    assert engine.verify_batch_msm(sigs, pks, msgs, coeffs=coeffs) == 0
    assert engine.verify_batch_msm(sigs, pks, msgs) == 0
    sigs[123456, 60] ^= 1
    assert engine.verify_batch_msm(sigs, pks, msgs, coeffs=coeffs) == 2
    assert engine.verify_batch_msm(sigs, pks, msgs) == 2


# ---------------------------------------------------------------- randomized differential test
def test_fuzz_mixed_batch_vs_oracle(engine, oracle):
    """Ragged messages, every corruption class, small-order / non-subgroup / identity keys and malformed
    encodings mixed in one batch; both torsion settings; GPU status vector == oracle status vector."""
    import pymodel as m
    rng = np.random.default_rng(0xF022)
    n = 6000
    lens = rng.integers(0, 120, size=n)
    lens[:40] = [0, 1, 6, 7, 8, 13, 14, 20, 21, 27] * 4
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    flat = rng.integers(0, 256, size=int(off[-1]) + 1, dtype=np.uint8)
    sks, nonces = make_scalars(rng, n), make_scalars(rng, n)
    pks, sigs = engine.keygen_sign_many(sks, nonces, flat, offsets=off)
    inf = np.zeros(n, dtype=np.uint8)
    special = [m.FIXTURE_SMALL_ORDER_PK] + [m.SMALL_ORDER_POINTS[o] for o in (2, 5, 10)]
    sp_bytes = [np.frombuffer(m.fp6_to_bytes48(p[0]) + m.fp6_to_bytes48(p[1]), dtype=np.uint8) for p in special]
    kinds = rng.integers(0, 14, size=n)
    for i in np.nonzero(kinds < 9)[0]:
        k = kinds[i]
        if k == 0:
            sigs[i, 49 + rng.integers(0, 31)] ^= 1 << rng.integers(0, 8)
        elif k == 1 and lens[i] > 0:
            flat[int(off[i]) + rng.integers(0, lens[i])] ^= 1 << rng.integers(0, 8)
        elif k == 2:
            pks[i] = pks[(i + 7) % n]
        elif k == 3:
            sigs[i, :49] = sigs[(i + 3) % n, :49]
        elif k == 4:
            pks[i] = sp_bytes[rng.integers(0, 4)]
        elif k == 5:
            inf[i] = 1
        elif k == 6:
            sigs[i, rng.integers(0, 6) * 8: rng.integers(0, 6) * 8 + 8] = 0xFF
        elif k == 7:
            sigs[i, 49:81] = 0xFF
        elif k == 8:
            sigs[i, 0] ^= 1           # R.x changed: canonical, (almost surely) not the signer's R
    for torsion in (True, False):
        want = oracle.verify_many(sigs, pks, flat, offsets=off, check_torsion=torsion, pk_inf=inf)
        for mode in ("lane", "coop"):       # throughput kernels and the wave-per-signature kernel
            st, nf = engine.verify_many(sigs, pks, flat, offsets=off, check_torsion=torsion, pk_inf=inf, mode=mode)
            assert (st == want).all(), (mode, np.nonzero(st != want)[0][:10])
            assert nf == int((want != 0).sum())
        assert set(np.unique(st)) <= {0, 1, 2, 3} and (st == 0).sum() > n // 3


def test_config4_size_2pow22_single_gpu(engine):
    """BASELINE.json configs[3] size (2^22 signatures) on ONE GPU: the per-GPU shard of an 8-GPU run is
    2^19, but nothing in the engine depends on that -- workspaces scale to 288 GB.  Verdicts by construction."""
    rng = np.random.default_rng(0x5C4E0224)
    n = 1 << 22
    sks, nonces = make_scalars(rng, n), make_scalars(rng, n)
    msgs = rng.integers(0, 256, size=(n, 80), dtype=np.uint8)
    pks, sigs = engine.keygen_sign_many(sks, nonces, msgs)
    bad = rng.permutation(n)[:1000]
    sigs[bad, 49] ^= 1
    st, nf = engine.verify_many(sigs, pks, msgs, check_torsion=False)
    expect = np.zeros(n, dtype=np.uint8)
    expect[bad] = 2
    assert nf == 1000 and (st == expect).all()
    assert engine.verify_batch_msm(sigs, pks, msgs) == 2
    sigs[bad, 49] ^= 1
    assert engine.verify_batch_msm(sigs, pks, msgs) == 0


def test_keyed_signatures_wire_form(engine, oracle):
    """KeyedSignature (src/signature.rs:55-60, 232-271): 130-byte records pk(49) || sig(81) verified with
    the key decompressed on the GPU; undecodable halves give SSA_MALFORMED; object mirror round trip."""
    import os as _os
    import schnorr_sig_amd as ssa
    rng = np.random.default_rng(41)
    n = 512
    sks, nonces = make_scalars(rng, n), make_scalars(rng, n)
    msgs = rng.integers(0, 256, size=(n, 40), dtype=np.uint8)
    pks, sigs = engine.keygen_sign_many(sks, nonces, msgs)
    keyed = np.zeros((n, 130), dtype=np.uint8)
    for i in range(n):
        keyed[i, :49] = np.frombuffer(oracle.compress(pks[i].tobytes()), dtype=np.uint8)
    keyed[:, 49:] = sigs
    st, nf = engine.verify_keyed_many(keyed, msgs)
    assert nf == 0 and (st == 0).all()
    bad = keyed.copy()
    bad[3, 48] ^= 0x40          # other root: a different (valid) key -> InvalidSignature
    bad[4, 48] = 0xFF           # flag byte 255: decompression failed
    bad[5, 60] ^= 1             # sig.x changed
    bad[6, 49 + 49 + 31] = 0x7F # e >= q
    bad[7, :8] = 0xFF           # pk limb >= p
    st, nf = engine.verify_keyed_many(bad, msgs)
    assert list(st[3:8]) == [2, 3, 2, 3, 3] and nf == 5 and (np.delete(st, [3, 4, 5, 6, 7]) == 0).all()
    # identity key || a signature made with sk = 0 semantics (e = r): accepted (src/signature.rs:462-481 layout)
    idk = np.zeros((1, 130), dtype=np.uint8)
    idk[0, 48] = 0x80
    r = make_scalars(rng, 1)
    rpk, _ = engine.keygen_sign_many(r, r, np.zeros((1, 1), np.uint8))
    idk[0, 49:49 + 48] = rpk[0, :48]
    idk[0, 49 + 49:] = r[0]
    st, nf = engine.verify_keyed_many(idk, np.zeros((1, 3), np.uint8))
    assert st[0] == 0 and nf == 0
    # object mirror
    rnd = lambda k: _os.urandom(k)
    kp = ssa.KeyPair.new(rnd, engine)
    ks = kp.sign_and_bind_pkey(b"bound message", rnd, engine)
    raw = ks.to_bytes()
    assert len(raw) == ssa.KEYED_SIGNATURE_LENGTH
    back = ssa.KeyedSignature.from_bytes(raw, engine)
    assert back is not None and back.public_key == kp.public_key and back.signature == ks.signature
    assert back.verify(b"bound message", engine) is None
    assert ssa.KeyedSignature.from_bytes(raw[:48] + b"\xff" + raw[49:], engine) is None


def test_multi_device_sharding_single_process(engine):
    """ssa_multi_*: the shard/thread/sum logic of the single-process multi-GPU entry point, exercised with
    three contexts on device 0 (ragged shards, variable-length messages); must equal the one-context result."""
    import schnorr_sig_amd as ssa
    rng = np.random.default_rng(51)
    n = 1000
    lens = rng.integers(0, 60, size=n)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    flat = rng.integers(0, 256, size=int(off[-1]) + 1, dtype=np.uint8)
    sks, nonces = make_scalars(rng, n), make_scalars(rng, n)
    pks, sigs = engine.keygen_sign_many(sks, nonces, flat, offsets=off)
    bad = [0, 333, 334, 667, 999]
    sigs[bad, 50] ^= 1
    want, nf_want = engine.verify_many(sigs, pks, flat, offsets=off, check_torsion=True)
    multi = ssa.MultiEngine([0, 0, 0])
    try:
        st, nf = multi.verify_many(sigs, pks, flat, offsets=off, check_torsion=True)
        assert (st == want).all() and nf == nf_want == 5
        dense = rng.integers(0, 256, size=(n, 16), dtype=np.uint8)
        pk2, sg2 = engine.keygen_sign_many(sks, nonces, dense)
        st2, nf2 = multi.verify_many(sg2, pk2, dense, check_torsion=False)
        assert nf2 == 0 and (st2 == 0).all()
    finally:
        multi.close()


# ---------------------------------------------------------------- wave-per-signature (low-latency) kernel
@pytest.mark.parametrize("mode", ["lane", "coop"])
def test_both_kernel_families_on_the_edge_cases(engine, oracle, mode):
    """Every exceptional-case input of this file through BOTH kernel families: one lane per signature
    (throughput) and one wave per signature (latency)."""
    import pymodel as m
    rng = np.random.default_rng(61)
    prm = m.default_params()
    g = prm.generator()
    sigs, pks, msgs, infs = [], [], [], []

    def add(sig, pk96, msg, inf=0):
        sigs.append(sig); pks.append(pk96); msgs.append(msg); infs.append(inf)

    def pkb(p):
        return m.fp6_to_bytes48(p[0]) + m.fp6_to_bytes48(p[1])
    # honest signatures over the chunking-relevant message lengths
    for L in (0, 1, 6, 7, 8, 13, 14, 24, 48, 80, 160):
        msg = bytes(rng.integers(0, 256, size=L, dtype=np.uint8))
        sig, pk = m.sign(int(rng.integers(1, 2**62)), int(rng.integers(1, 2**62)), msg, prm)
        add(sig, pkb(pk), msg)
        add(sig[:49] + bytes([sig[49] ^ 1]) + sig[50:], pkb(pk), msg)
    # adversarial e: [h]P + [e]G lands on / opposite comb-table points
    for k in (1, 2, 65536, Q - 1):
        pk = m.pt_mul(k, g)
        rx = m.pt_mul(12345, g)[0]
        h = m.scalar_from_digest(m.hash_message(rx, pk, b"adv", prm))
        for target in (0, 1, 2, 65536, 65537, Q - 1):
            e = (target - h * k) % Q
            add(m.fp6_to_bytes48(rx) + b"\0" + e.to_bytes(32, "little"), pkb(pk), b"adv")
    # small-order and non-subgroup keys, identity key, malformed encodings
    for p in [m.FIXTURE_SMALL_ORDER_PK] + [m.SMALL_ORDER_POINTS[o] for o in (2, 5, 10)]:
        for t in range(3):
            rx = m.pt_mul(777 + t, g)[0]
            add(m.fp6_to_bytes48(rx) + b"\0" + int(rng.integers(0, 2**62)).to_bytes(32, "little"), pkb(p), b"xyz")
    r = int(rng.integers(1, 2**62))
    add(m.fp6_to_bytes48(m.pt_mul(r, g)[0]) + b"\0" + r.to_bytes(32, "little"), bytes(96), b"identity key", 1)
    good_sig, good_pk = m.sign(5, 7, b"m", prm)
    add(b"\xff" * 8 + good_sig[8:], pkb(good_pk), b"m")
    add(good_sig[:49] + b"\xff" * 32, pkb(good_pk), b"m")
    add(good_sig, pkb(good_pk)[:95] + bytes([pkb(good_pk)[95] ^ 1]), b"m")
    add(bytes(48) + b"\x80" + good_sig[49:], pkb(good_pk), b"m")
    # non-subgroup key AND undecodable sig.x: the subgroup check comes first (src/signature.rs:182 before :186)
    add(b"\xff" * 8 + good_sig[8:], pkb(m.FIXTURE_SMALL_ORDER_PK), b"m")
    n = len(sigs)
    S = np.frombuffer(b"".join(sigs), dtype=np.uint8).reshape(n, 81)
    P_ = np.frombuffer(b"".join(pks), dtype=np.uint8).reshape(n, 96)
    flat = np.frombuffer(b"".join(msgs) + b"\0", dtype=np.uint8)
    off = np.cumsum([0] + [len(x) for x in msgs]).astype(np.uint64)
    inf = np.array(infs, dtype=np.uint8)
    for torsion in (True, False):
        st, nf = engine.verify_many(S, P_, flat, offsets=off, check_torsion=torsion, pk_inf=inf, mode=mode)
        want = oracle.verify_many(S, P_, flat, offsets=off, check_torsion=torsion, pk_inf=inf)
        ok = np.ones(n, dtype=bool)
        ok[n - 3] = False                       # off-curve key: SSA_MALFORMED here, undefined in the reference
        assert (st[ok] == want[ok]).all(), (mode, torsion, np.nonzero(st != want)[0])
        assert (st[-5:-2] == 3).all()           # undecodable x, e >= q, off-curve key
        assert st[-1] == (1 if torsion else 3)  # InvalidPublicKey wins over the undecodable x, as in the reference
        assert nf == int((st != 0).sum())


def test_auto_dispatch_agrees_across_the_threshold(engine):
    """auto mode switches kernel family at the context's threshold; verdicts must not depend on it."""
    rng = np.random.default_rng(62)
    n = 15000
    sks, nonces = make_scalars(rng, n), make_scalars(rng, n)
    msgs = rng.integers(0, 256, size=(n, 33), dtype=np.uint8)
    pks, sigs = engine.keygen_sign_many(sks, nonces, msgs)
    bad = rng.permutation(n)[:100]
    sigs[bad, 52] ^= 4
    expect = np.zeros(n, dtype=np.uint8)
    expect[bad] = 2
    for torsion, cuts in ((True, (1, 100, 14336, 14337, n)), (False, (10240, 10241))):
        for cut in cuts:
            st, nf = engine.verify_many(sigs[:cut], pks[:cut], msgs[:cut], check_torsion=torsion)
            assert (st == expect[:cut]).all() and nf == int((expect[:cut] != 0).sum())


# ---- coefficients of the MSM-form verify_batch: ChaCha20 keystream generated on the device ------------------
def _chacha20_block(key, counter, nonce):
    """RFC 8439 section 2.3, straight from the text (independent of the kernel)."""
    import struct
    def rotl(x, n):
        return ((x << n) | (x >> (32 - n))) & 0xffffffff
    def qr(x, a, b, c, d):
        x[a] = (x[a] + x[b]) & 0xffffffff; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & 0xffffffff; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & 0xffffffff; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & 0xffffffff; x[b] = rotl(x[b] ^ x[c], 7)
    st = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + list(struct.unpack("<8I", key)) + [counter] + \
        list(struct.unpack("<3I", nonce))
    x = list(st)
    for _ in range(10):
        qr(x, 0, 4, 8, 12); qr(x, 1, 5, 9, 13); qr(x, 2, 6, 10, 14); qr(x, 3, 7, 11, 15)
        qr(x, 0, 5, 10, 15); qr(x, 1, 6, 11, 12); qr(x, 2, 7, 8, 13); qr(x, 3, 4, 9, 14)
    return struct.pack("<16I", *[(a + b) & 0xffffffff for a, b in zip(x, st)])


RFC8439_KEY = bytes(range(32))
RFC8439_NONCE = bytes.fromhex("000000090000004a00000000")
RFC8439_BLOCK1 = bytes.fromhex(
    "10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e"
    "d2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e")


def test_chacha20_model_reproduces_rfc8439_block():
    assert _chacha20_block(RFC8439_KEY, 1, RFC8439_NONCE) == RFC8439_BLOCK1      # RFC 8439 section 2.3.2


def test_device_chacha20_keystream(engine):
    """the generator behind `coeffs=None`: RFC 8439 known answer and random keys against the Python model"""
    assert engine.debug_chacha20(RFC8439_KEY, RFC8439_NONCE, 1, 1) == RFC8439_BLOCK1
    rng = np.random.default_rng(77)
    for _ in range(3):
        key, nonce = rng.bytes(32), rng.bytes(12)
        got = engine.debug_chacha20(key, nonce, 5, 300)
        want = b"".join(_chacha20_block(key, 5 + i, nonce) for i in range(300))
        assert got == want


def test_msm_form_with_library_drawn_coefficients(engine):
    """coeffs=None: fresh device-drawn coefficients every call; the verdict does not depend on them"""
    rng = np.random.default_rng(78)
    n = 5000
    sks, nonces = make_scalars(rng, n), make_scalars(rng, n)
    msgs = rng.integers(0, 256, size=(n, 21), dtype=np.uint8)
    pks, sigs = engine.keygen_sign_many(sks, nonces, msgs)
    for _ in range(3):
        assert engine.verify_batch_msm(sigs, pks, msgs) == 0
    sigs[1234, 60] ^= 1
    for _ in range(3):
        assert engine.verify_batch_msm(sigs, pks, msgs) == 2
