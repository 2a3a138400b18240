"""The TIMING path of the CPU baseline (oracle/schnorr_oracle_fast.inc: lazy Fp6 products, width-5 NAF, fixed-base table,
bucket MSM -- test infrastructure, like the plain oracle it is checked against) returns what the plain restatement of
src/signature.rs:181-205 / src/batch.rs:31-130 returns on every input: honest, every corruption class, the reference's
non-subgroup fixture (src/signature.rs:387-404), identity keys, both flag-byte semantics, ragged messages."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Q = 0x7AF2599B3B3F22D0563FBF0F990A37B5327AA72330157722D443623EAED4ACCF


def _scalars(rng, n):
    s = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    s[:, 31] &= 0x3F
    s[:, 0] |= 1
    return s


def _fixture_key():
    with open(os.path.join(ROOT, "tests", "golden", "vectors.json")) as fh:
        f = json.load(fh)["fixture_small_order_pk"]
    return np.frombuffer(b"".join(int(x).to_bytes(8, "little") for x in (f["x"] + f["y"])), dtype=np.uint8)


@pytest.fixture(scope="module")
def batch(oracle):
    rng = np.random.default_rng(7700)
    n = 160
    lens = rng.integers(0, 100, size=n)
    lens[:6] = [0, 1, 6, 7, 14, 80]
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    flat = rng.integers(0, 256, size=int(off[-1]) + 1, dtype=np.uint8)
    pks, sigs = oracle.keygen_sign_many(_scalars(rng, n), _scalars(rng, n), flat, offsets=off)
    return rng, n, sigs, pks, flat, off


def test_windowed_verify_equals_the_plain_restatement(oracle, batch):
    rng, n, sigs, pks, flat, off = batch
    sigs, pks, flat = sigs.copy(), pks.copy(), flat.copy()
    inf = np.zeros(n, np.uint8)
    sigs[10, 49] ^= 1                      # e
    flat[int(off[11])] ^= 1                # message (length 80 at index 5; index 11 is random: skip when empty)
    pks[12] = pks[13]                      # someone else's key
    sigs[14, :49] = sigs[15, :49]          # someone else's R
    pks[16] = _fixture_key()               # not in the prime subgroup: InvalidPublicKey with the check
    sigs[17, 48] ^= 0x40                   # wrong sort bit: only verify_batch's semantics care
    sigs[18, 48] |= 0x04                   # undecodable flag byte
    sigs[19, 0] ^= 1                       # an x that is (almost surely) not R.x, maybe not on the curve
    sigs[20, 49:] = np.frombuffer(Q.to_bytes(32, "little"), np.uint8)           # e = q: not canonical
    sigs[21, :8] = 0xFF                    # limb >= p
    pks[22, 48:56] = 0xFF                  # key limb >= p
    pks[23, 0] ^= 1                        # key off the curve
    inf[24] = 1                            # identity key (a valid PublicKey, src/public.rs:95-101)
    sigs[25, :48] = 0
    sigs[25, 48] = 0x80                    # R = identity encoding
    for torsion in (False, True):
        for fb in (False, True):
            a = oracle.verify_many(sigs, pks, flat, offsets=off, check_torsion=torsion, pk_inf=inf, sig_flag_byte=fb, threads=1)
            b = oracle.verify_many_fast(sigs, pks, flat, offsets=off, check_torsion=torsion, pk_inf=inf, sig_flag_byte=fb,
                                        threads=2)
            assert (a == b).all(), (torsion, fb, np.nonzero(a != b)[0], a[a != b], b[a != b])
            assert a[10] == 2 and a[12] == 2 and a[14] == 2 and a[20] == 3 and a[21] == 3 and a[22] == 3 and a[23] == 3
            assert a[16] == (1 if torsion else 2) and a[18] == (3 if fb else 0) and a[17] == (2 if fb else 0)
            assert (a[26:] == 0).all()


def test_windowed_msm_form_equals_the_plain_restatement(oracle, batch):
    rng, n, sigs, pks, flat, off = batch
    co = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    co[:, 31] &= 0x3F
    co128 = co.copy()
    co128[:, 16:] = 0
    for c in (co, co128):
        assert oracle.verify_batch_msm(sigs, pks, flat, c, offsets=off, threads=1) == 0
        assert oracle.verify_batch_msm_fast(sigs, pks, flat, c, offsets=off, threads=2) == 0
    cases = []
    s2 = sigs.copy()
    s2[33, 50] ^= 2
    cases.append((s2, pks, None, 2))                      # one bad signature
    p2 = pks.copy()
    p2[[40, 41]] = p2[[41, 40]]
    cases.append((sigs, p2, None, 2))                     # order matters (src/batch.rs:175-178)
    s3 = sigs.copy()
    s3[7, 48] |= 0x08
    cases.append((s3, pks, None, 3))                      # from_compressed(..).unwrap() panics
    s4 = sigs.copy()
    s4[9, 48] ^= 0x40
    cases.append((s4, pks, None, 2))                      # the other root of R
    inf = np.zeros(n, np.uint8)
    inf[50] = 1
    cases.append((sigs, pks, inf, 2))                     # an identity key contributes nothing: the equation fails
    for s_, p_, i_, want in cases:
        a = oracle.verify_batch_msm(s_, p_, flat, co128, offsets=off, threads=1, pk_inf=i_)
        b = oracle.verify_batch_msm_fast(s_, p_, flat, co128, offsets=off, threads=2, pk_inf=i_)
        assert a == b == want
    # sizes around the window-width steps of the bucket method, single signature, empty batch
    for m in (1, 2, 3, 17, 129):
        assert oracle.verify_batch_msm_fast(sigs[:m], pks[:m], flat, co[:m], offsets=off[:m + 1], threads=1) == 0
