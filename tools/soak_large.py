"""Differential soak at FULL OCCUPANCY: random large batches (2^17 .. 2^20 signatures, fresh seeds) for a wall-clock budget.
tools/soak.py runs thousands of small batches; the hazards of a full machine (round 5: a scratch reload of the compiler's still
in flight when a generated statement starts, profiles/r05/gather_ab.txt) only show when every SIMD is busy and memory is slow.
Per iteration:
  * keygen + sign on the GPU TWICE from the same inputs: the same bytes (throughput signer), and on a 2^15 prefix the
    constant-time signer's bytes too;
  * 1 % of the lanes corrupted in six ways whose statuses are known BY CONSTRUCTION (every lane of the batch is checked against
    that expectation under the three semantics, device-pointer path through the host wrapper);
  * a random sample of 1536 lanes (all corrupted ones of the sample included) against the CPU oracle;
  * the MSM-form verdict of the corrupted batch (must reject) and of the honest one (must accept);
  * every fourth iteration: the same signatures through a keyed context (64 signers) -- ladder tables and per-key combs;
    every fifth: the messages as one flat buffer with an offsets table; every seventh: a batch of more than one slice
    (up to 2.5 * 2^20 signatures: bounded staging, two host threads, the twin context).
    python tools/soak_large.py [seconds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import schnorr_sig_amd as ssa
from oracle import Oracle
import pymodel as m

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
eng, orc = ssa.Engine(0), Oracle()
fixture = np.frombuffer(m.fp6_to_bytes48(m.FIXTURE_SMALL_ORDER_PK[0]) + m.fp6_to_bytes48(m.FIXTURE_SMALL_ORDER_PK[1]), dtype=np.uint8)
t0, it, total, fails = time.time(), 0, 0, 0


TRACE = os.environ.get("SOAK_TRACE") is not None


def trace(what):
    if TRACE:
        print("    [%d] %s" % (it, what), flush=True)


def check(cond, what):
    global fails
    if not cond:
        fails += 1
        print("MISMATCH (iteration %d): %s" % (it, what), flush=True)


while time.time() - t0 < budget:
    rng = np.random.default_rng(0x50A4 + 7919 * it)
    n = int(rng.integers(1 << 17, (1 << 20) + 1))
    if it % 7 == 6:
        n = int(rng.integers((1 << 20) + 1, 5 << 19))          # more than one slice: the two-thread host path, the twin context
    mlen = int(rng.choice([80, 80, 32, 77, 160]))
    sks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sks[:, 31] &= 0x3F; sks[:, 0] |= 1
    nonces = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); nonces[:, 31] &= 0x3F; nonces[:, 0] |= 1
    msgs = rng.integers(0, 256, size=(n, mlen), dtype=np.uint8)
    trace("n = %d, message length %d: sign" % (n, mlen))
    pks, sigs = eng.keygen_sign_many(sks, nonces, msgs)
    trace("sign again")
    pks2, sigs2 = eng.keygen_sign_many(sks, nonces, msgs)
    check((pks == pks2).all() and (sigs == sigs2).all(), "the signer is not deterministic (n = %d)" % n)
    k = 1 << 15
    trace("constant-time signer")
    pk_ct, sig_ct = eng.keygen_sign_many(sks[:k], nonces[:k], msgs[:k], constant_time=True)
    check((pk_ct == pks[:k]).all() and (sig_ct == sigs[:k]).all(), "constant-time signer differs from the throughput signer")
    honest_sigs, honest_pks, honest_msgs = sigs.copy(), pks.copy(), msgs.copy()
    # corruptions with statuses known by construction
    nbad = n // 100
    bad = rng.permutation(n)[:nbad]
    kinds = np.arange(nbad) % 6
    nxt = (bad + 1) % n
    sigs[bad[kinds == 0], 49] ^= 1                                # e: InvalidSignature
    msgs[bad[kinds == 1], mlen // 2] ^= 0x10                      # message: InvalidSignature
    pks[bad[kinds == 2]] = honest_pks[nxt[kinds == 2]]            # someone else's key: InvalidSignature
    sigs[bad[kinds == 3], :49] = honest_sigs[nxt[kinds == 3], :49]   # someone else's R: InvalidSignature
    pks[bad[kinds == 4]] = fixture                                # not in the subgroup: InvalidPublicKey with the check, else InvalidSignature
    sigs[bad[kinds == 5], 48] ^= 0x40                             # sort bit of R: only the flag-byte semantics notices
    for torsion, fb in ((False, False), (True, False), (False, True)):
        want = np.zeros(n, dtype=np.uint8)
        want[bad[kinds <= 3]] = 2
        want[bad[kinds == 4]] = 1 if torsion else 2
        want[bad[kinds == 5]] = 2 if fb else 0
        trace("verify_many torsion %s flag byte %s" % (torsion, fb))
        st, nf = eng.verify_many(sigs, pks, msgs, check_torsion=torsion, sig_flag_byte=fb, mode="lane")
        same = bool((st == want).all()) and nf == int((want != 0).sum())
        check(same, "n = %d, torsion %s, flag byte %s: %d lanes differ from the construction, first %s"
              % (n, torsion, fb, int((st != want).sum()), np.nonzero(st != want)[0][:8]))
        samp = np.unique(np.concatenate([rng.integers(0, n, size=1024), bad[:512]]))
        exp = orc.verify_many(sigs[samp], pks[samp], msgs[samp], check_torsion=torsion, sig_flag_byte=fb)
        check((st[samp] == exp).all(), "n = %d, torsion %s, flag byte %s: oracle sample differs" % (n, torsion, fb))
    trace("MSM form, corrupted")
    check(eng.verify_batch_msm(sigs, pks, msgs) == 2, "MSM form accepted the corrupted batch")
    trace("MSM form, honest")
    check(eng.verify_batch_msm(honest_sigs, honest_pks, honest_msgs) == 0, "MSM form rejected the honest batch")
    if it % 5 == 4:
        # the same corrupted batch as a flat message buffer with an offsets table (the reference's &[&[u8]]): same statuses
        trace("offsets form")
        off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(mlen))
        st, nf = eng.verify_many(sigs, pks, msgs.reshape(-1), offsets=off, check_torsion=True, mode="lane")
        want = np.zeros(n, dtype=np.uint8)
        want[bad[kinds <= 3]] = 2
        want[bad[kinds == 4]] = 1
        check(nf == int((want != 0).sum()) and (st == want).all(), "offsets form differs from the construction")
    if it % 4 == 0:
        # 64 signers sign everything (their keys: the first 64 secret keys): key set with ladder tables, then with per-key combs
        ks = 64
        idx = rng.integers(0, ks, size=n).astype(np.uint32)
        trace("sign for 64 signers")
        kpks, ksigs = eng.keygen_sign_many(sks[idx], nonces, honest_msgs)
        for kind in ("ladder", "comb"):
            trace("key set " + kind)
            keyset = eng.keyset_create(honest_pks[:ks], kind=kind)
            trace("verify_many_indexed")
            st, nf = eng.verify_many_indexed(keyset, idx, ksigs, honest_msgs, check_torsion=True)
            check(nf == 0 and not st.any(), "key set (%s): %d honest signatures rejected" % (kind, nf))
            ksig2 = ksigs.copy()
            ksig2[bad, 49] ^= 1
            st, nf = eng.verify_many_indexed(keyset, idx, ksig2, honest_msgs, check_torsion=True)
            want = np.zeros(n, dtype=np.uint8)
            want[bad] = 2
            check(nf == nbad and (st == want).all(), "key set (%s): corrupted lanes differ from the construction" % kind)
            trace("key set close")
            keyset.close()
    it += 1
    total += n
    print("  ... %d iterations, %d signatures, %.0f s" % (it, total, time.time() - t0), flush=True)
print("large-batch soak %s: %d iterations, %d signatures x 3 semantics (every lane against the construction, oracle samples, MSM "
      "verdicts, signers twice), %d mismatches, %.0f s" % ("ok" if fails == 0 else "FAILED", it, total, fails, time.time() - t0))
sys.exit(0 if fails == 0 else 1)
