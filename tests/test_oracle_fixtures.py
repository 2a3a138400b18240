"""CPU tests pinning the oracle: (1) every reference-owned fixture that constrains the
arithmetic, (2) the independent Python model vs the C restatement on the committed golden
vectors.  Parity against the upstream Rescue constants / generator stays UNPINNED (they live
in the absent crates cheetah/hash, reference Cargo.toml:16,18)."""
import json
import os

import numpy as np
import pytest

import pymodel as m

HERE = os.path.dirname(os.path.abspath(__file__))
P, Q = m.P, m.Q


@pytest.fixture(scope="module")
def gold():
    return json.load(open(os.path.join(HERE, "golden", "vectors.json")))


# ---------------------------------------------------------------- reference-owned fixtures
def test_fixture_point_on_curve_under_canonical_limbs():
    """src/signature.rs:387-404 == src/error.rs:47-64: from_raw_unchecked limbs are canonical
    integers and satisfy y^2 = x^3 + x + (u + 395) in Fp[u]/(u^6 - 7) (README.md:4-9)."""
    f = m.FIXTURE_SMALL_ORDER_PK
    assert all(v < P for v in f[0] + f[1])
    assert m.on_curve(f)
    # the Montgomery reading (limbs * R^-1, R = 2^64) is NOT on the curve
    rinv = pow(2**64, P - 2, P)
    g = (tuple(v * rinv % P for v in f[0]), tuple(v * rinv % P for v in f[1]))
    assert not m.on_curve(g)


def test_group_order_and_subgroup_check_of_fixture(oracle):
    """q is prime, #E(Fp6) = q*h is the only multiple of q in the Hasse interval, [q*h]F = O and
    [q]F != O: the reference returns InvalidPublicKey for F (src/error.rs:66-69)."""
    import sympy
    assert sympy.isprime(Q)
    n = Q * m.COFACTOR
    lo, hi = P**6 + 1 - 2 * P**3, P**6 + 1 + 2 * P**3
    assert lo <= n <= hi and n - Q < lo and n + Q > hi
    f = m.FIXTURE_SMALL_ORDER_PK
    assert m.pt_mul(n, f) is None
    assert m.pt_mul(Q, f) is not None
    assert not m.is_torsion_free(f)
    assert not oracle.is_torsion_free(f)
    assert oracle.on_curve(f)


def test_field_tower_constants():
    assert 7 * m.INV_ALPHA % (P - 1) == 1 and m.INV_ALPHA == 0x92492491B6DB6DB7
    assert m.GAMMA == 0xFFFFFFFE00000002 and pow(m.GAMMA, 3, P) == P - 1
    # u^6 - 7 irreducible: 7 is neither a square nor a cube mod p
    assert pow(7, (P - 1) // 2, P) != 1 and pow(7, (P - 1) // 3, P) != 1
    # B = u + 395 is a non-square: no point has x = 0, so 49 zero bytes decode to None (src/public.rs:115-120)
    assert not m.f6_is_square(m.CURVE_B)


def test_wire_layouts_from_reference_encoding_tests():
    """src/signature.rs:430-460: Signature = x(49) || e(32, LE canonical);
    src/public.rs:95-101: identity = [0;48] || 0x80; src/private.rs:154-160,203-204: 32 x 0xff and
    a top byte of 127 are not canonical scalars."""
    import schnorr_sig_amd as ssa
    assert ssa.SIGNATURE_LENGTH == 81 and ssa.PUBLIC_KEY_LENGTH == 49 and ssa.KEYED_SIGNATURE_LENGTH == 130
    one_e = bytes(49) + (1).to_bytes(32, "little")
    assert ssa.Signature(one_e).to_bytes()[49] == 1
    assert ssa.Signature.from_bytes(bytes([0xFF]) * 81) is None
    with pytest.raises(ValueError):
        ssa.PrivateKey(bytes(32))
    with pytest.raises(ValueError):
        ssa.PrivateKey(bytes([0xFF]) * 32)
    with pytest.raises(ValueError):
        ssa.PrivateKey(bytes(31) + bytes([127]))
    assert Q >> 248 == 0x7A


def test_negative_verify_cases_of_the_reference(oracle, gold):
    """src/signature.rs:363-426 and src/error.rs:40-84: wrong message, pk = generator, non-subgroup
    pk (-> InvalidPublicKey, checked first), sig.x = identity encoding, e = 0: all errors."""
    for rec in gold["signatures"]:
        for neg in rec["negative"]:
            st = oracle.verify(bytes.fromhex(neg["sig"]), bytes.fromhex(neg["pk"]), bytes.fromhex(neg["msg"]), True)
            assert st == neg["status"] != 0, neg["what"]
            if neg["what"] == "non-subgroup pk":
                assert st == 1
                assert oracle.verify(bytes.fromhex(neg["sig"]), bytes.fromhex(neg["pk"]),
                                     bytes.fromhex(neg["msg"]), False) == neg["status_no_torsion"] == 2
            else:
                assert st == 2


# ---------------------------------------------------------------- model vs C oracle on golden vectors
def test_params_blob_matches_independent_derivation():
    from oracle import PARAMS_BLOB
    import struct
    b = open(PARAMS_BLOB, "rb").read()
    assert len(b) == 2816 and b[:8] == b"SSAPARM1"
    prm = m.default_params()
    assert struct.unpack_from("<IIiIII", b, 8) == (prm.n_rounds, prm.rate_off, prm.cap_len_idx, prm.pad_mode,
                                                   prm.digest_off, 0)
    mds = struct.unpack_from("<144Q", b, 32)
    assert [list(mds[12 * i: 12 * i + 12]) for i in range(12)] == prm.mds
    for r in range(prm.n_rounds):
        assert list(struct.unpack_from("<12Q", b, 1184 + 96 * r)) == prm.ark1[r]
        assert list(struct.unpack_from("<12Q", b, 1952 + 96 * r)) == prm.ark2[r]
    g = prm.generator()
    assert struct.unpack_from("<6Q", b, 2720) == g[0] and struct.unpack_from("<6Q", b, 2768) == g[1]
    assert m.on_curve(g) and m.is_torsion_free(g)


def test_mds_is_invertible_and_circulant():
    prm = m.default_params()
    for i in range(12):
        for j in range(12):
            assert prm.mds[i][j] == m.MDS_ROW[(j - i) % 12]
    # non-singular mod p (Gaussian elimination)
    a = [row[:] for row in prm.mds]
    for c in range(12):
        piv = next(r for r in range(c, 12) if a[r][c] % P)
        a[c], a[piv] = a[piv], a[c]
        inv = pow(a[c][c], P - 2, P)
        for r in range(c + 1, 12):
            f = a[r][c] * inv % P
            a[r] = [(x - f * y) % P for x, y in zip(a[r], a[c])]


def test_fp6_golden(oracle, gold):
    for rec in gold["fp6"]:
        assert [int(v) for v in oracle.fp6_mul(rec["a"], rec["b"])] == rec["mul"]
        assert [int(v) for v in oracle.fp6_sqr(rec["a"])] == rec["sqr"]
        if "inv" in rec:
            assert [int(v) for v in oracle.fp6_inv(rec["a"])] == rec["inv"]


def test_scalar_mul_golden(oracle, gold):
    g = (tuple(gold["generator"]["x"]), tuple(gold["generator"]["y"]))
    f = (tuple(gold["fixture_small_order_pk"]["x"]), tuple(gold["fixture_small_order_pk"]["y"]))
    for rec in gold["scalar_mul"]:
        k = int(rec["k"])
        if k >= 2**256:
            continue  # the C entry point takes 32-byte scalars
        got = oracle.point_mul(k, g if rec["base"] == "G" else f)
        want = None if rec["res"] is None else (tuple(rec["res"][0]), tuple(rec["res"][1]))
        assert got == want, rec["k"]


def test_rescue_golden(oracle, gold):
    for rec in gold["rescue_permutation"]:
        assert [int(v) for v in oracle.rescue_permutation(rec["in"])] == rec["out"]
    for rec in gold["hash_field"]:
        assert [int(v) for v in oracle.hash_field(np.array(rec["in"], dtype=np.uint64))] == rec["digest"]


def test_sign_verify_golden(oracle, gold):
    for rec in gold["signatures"]:
        sk, nonce, msg = bytes.fromhex(rec["sk"]), bytes.fromhex(rec["nonce"]), bytes.fromhex(rec["msg"])
        pk, inf = oracle.keygen(sk)
        assert pk.hex() == rec["pk"] and not inf
        assert oracle.sign(sk, nonce, pk, msg).hex() == rec["sig"]
        assert oracle.hash_message(bytes.fromhex(rec["sig"])[:48], pk, msg).hex() == rec["digest"]
        assert oracle.verify(bytes.fromhex(rec["sig"]), pk, msg, True) == 0


def test_message_chunking_rule():
    """src/signature.rs:285-301: 7-byte chunks, 0x01 terminator only on a partial last chunk."""
    assert m.message_to_felts(b"") == []
    assert m.message_to_felts(bytes(7)) == [0]
    assert m.message_to_felts(bytes(14)) == [0, 0]
    assert m.message_to_felts(b"\x05") == [5 | (1 << 8)]
    assert m.message_to_felts(bytes([255] * 8)) == [int.from_bytes(bytes([255] * 7), "little"), 255 | (1 << 8)]
    assert len(m.message_to_felts(bytes(80))) == 12 and len(m.message_to_felts(bytes(160))) == 23


def test_verify_batch_msm_form(oracle, gold):
    """src/batch.rs:152-179: 5 signatures (signers 3, 4 reuse keypair 0) verify; swapping
    public_keys[1], [2] fails.  MSM form (random linear combination) == AND of per-signature verdicts."""
    b = gold["batch5"]
    sigs = np.frombuffer(bytes.fromhex("".join(b["sigs"])), dtype=np.uint8).reshape(5, 81)
    pks = np.frombuffer(bytes.fromhex("".join(b["pks"])), dtype=np.uint8).reshape(5, 96)
    coeffs = np.frombuffer(bytes.fromhex("".join(b["coeffs"])), dtype=np.uint8).reshape(5, 32)
    msgs = [bytes.fromhex(x) for x in b["msgs"]]
    flat = np.frombuffer(b"".join(msgs) + b"\0", dtype=np.uint8)
    off = np.cumsum([0] + [len(x) for x in msgs]).astype(np.uint64)
    assert oracle.verify_batch_msm(sigs, pks, flat, coeffs, offsets=off) == b["status"] == 0
    assert (oracle.verify_many(sigs, pks, flat, offsets=off, check_torsion=False) == 0).all()
    sw = pks.copy()
    sw[[1, 2]] = sw[[2, 1]]
    assert oracle.verify_batch_msm(sigs, sw, flat, coeffs, offsets=off) == b["status_swapped_1_2"] == 2
    st = oracle.verify_many(sigs, sw, flat, offsets=off, check_torsion=False)
    assert list(st) == [0, 2, 2, 0, 0]


def test_config1_1024_signatures_through_cpu_verify_batch(oracle):
    """BASELINE.json configs[0]: 1024 random-keypair signatures through the CPU restatement of
    verify_batch (MSM form) and of n x Signature::verify -- plumbing, no GPU."""
    rng = np.random.default_rng(0x5C4E0222)
    n = 1024
    sks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sks[:, 31] &= 0x3F; sks[:, 0] |= 1
    nonces = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); nonces[:, 31] &= 0x3F; nonces[:, 0] |= 1
    coeffs = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); coeffs[:, 31] &= 0x3F
    msgs = rng.integers(0, 256, size=(n, 80), dtype=np.uint8)
    pks, sigs = oracle.keygen_sign_many(sks, nonces, msgs)
    assert oracle.verify_batch_msm(sigs, pks, msgs, coeffs) == 0
    assert (oracle.verify_many(sigs, pks, msgs, check_torsion=False) == 0).all()
    sigs[517, 60] ^= 4
    assert oracle.verify_batch_msm(sigs, pks, msgs, coeffs) == 2
    st = oracle.verify_many(sigs, pks, msgs, check_torsion=False)
    assert st[517] == 2 and st.sum() == 2


def test_mds_matrix_has_no_singular_minor(tmp_path):
    """The recalled circulant is a valid Rescue MDS layer over Goldilocks: all 2 704 155 square
    submatrices are non-singular (oracle/tools/mds_check.c, exhaustive)."""
    import subprocess
    root = os.path.dirname(HERE)
    exe = str(tmp_path / "mds_check")
    subprocess.check_call(["gcc", "-O2", "-o", exe, os.path.join(root, "oracle", "tools", "mds_check.c")])
    out = subprocess.run([exe, os.path.join(root, "schnorr-sig_amd", "params", "params_default.bin")],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip() == "MDS ok 2704155", out.stdout


def test_oracle_flag_byte_semantics_and_identity_key_in_the_batch(oracle):
    """The oracle's restatement of verify_batch's treatment of sig.x's flag byte (src/batch.rs:104) and of an identity
    public key (src/public.rs:95-101, src/batch.rs:106), which the round-2 GPU tests compare against."""
    rng = np.random.default_rng(77)
    n = 6
    sks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sks[:, 31] &= 0x3F; sks[:, 0] |= 1
    nonces = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); nonces[:, 31] &= 0x3F; nonces[:, 0] |= 1
    msgs = rng.integers(0, 256, size=(n, 20), dtype=np.uint8)
    pks, sigs = oracle.keygen_sign_many(sks, nonces, msgs)
    co = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); co[:, 31] &= 0x3F
    assert (oracle.verify_many(sigs, pks, msgs, check_torsion=False, sig_flag_byte=True) == 0).all()
    assert oracle.verify_batch_msm(sigs, pks, msgs, co) == 0
    s2 = sigs.copy()
    s2[1, 48] ^= 0x40          # R decodes to -R
    s2[2, 48] |= 0x04          # undecodable
    st = oracle.verify_many(s2, pks, msgs, check_torsion=False, sig_flag_byte=True)
    assert list(st) == [0, 2, 3, 0, 0, 0]
    assert list(oracle.verify_many(s2, pks, msgs, check_torsion=True)) == [0] * n      # Signature::verify ignores byte 48
    assert oracle.verify_batch_msm(s2, pks, msgs, co) == 3
    assert oracle.verify_batch_msm(s2[[0, 1, 3]], pks[[0, 1, 3]], msgs[[0, 1, 3]], co[:3]) == 2
    inf = np.zeros(n, np.uint8); inf[4] = 1
    assert oracle.verify_batch_msm(sigs, pks, msgs, co, pk_inf=inf) == 2               # key replaced by the identity
    assert list(oracle.verify_many(sigs, pks, msgs, check_torsion=True, pk_inf=inf)) == [0, 0, 0, 0, 2, 0]
