"""One-off evidence beyond the test-suite's 2^16 slices: EVERY lane of a 2^20-signature batch with 1 % corruptions
(config 5 = config 3's inputs + the corruption classes) against the CPU oracle on all host threads, under the three
semantics (verify_batch, Signature::verify with the subgroup check, verify_batch's flag-byte treatment), plus the
MSM-form verdict of the honest and of the corrupted batch.  ~2-4 minutes of CPU on the GPU box.
    python tools/full_batch_parity.py [log2 n]"""
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

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
eng, orc = ssa.Engine(0), Oracle()
rng = np.random.default_rng(0x5C4E0225)
sks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sks[:, 31] &= 0x3F; sks[:, 0] |= 1
nonces = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); nonces[:, 31] &= 0x3F; nonces[:, 0] |= 1
msgs = rng.integers(0, 256, size=(n, 80), dtype=np.uint8)
pks, sigs = eng.keygen_sign_many(sks, nonces, msgs)
honest_sigs, honest_pks = sigs.copy(), pks.copy()
nbad = n // 100
bad = rng.permutation(n)[:nbad]
kinds = np.arange(nbad) % 6
f = np.frombuffer(m.fp6_to_bytes48(m.FIXTURE_SMALL_ORDER_PK[0]) + m.fp6_to_bytes48(m.FIXTURE_SMALL_ORDER_PK[1]), dtype=np.uint8)
sigs[bad[kinds == 0], 49] ^= 1
msgs[bad[kinds == 1], 40] ^= 0x10
pks[bad[kinds == 2]] = pks[(bad[kinds == 2] + 1) % n]
sigs[bad[kinds == 3], :49] = sigs[(bad[kinds == 3] + 1) % n, :49]
pks[bad[kinds == 4]] = f
sigs[bad[kinds == 5], 48] ^= 0x40                      # sort bit of sig.x: only the flag-byte semantics notices
ok = True
for torsion, fb in ((False, False), (True, False), (False, True)):
    t0 = time.time()
    st, nf = eng.verify_many(sigs, pks, msgs, check_torsion=torsion, sig_flag_byte=fb)
    t1 = time.time()
    want = orc.verify_many(sigs, pks, msgs, check_torsion=torsion, sig_flag_byte=fb)
    t2 = time.time()
    same = bool((st == want).all()) and nf == int((want != 0).sum())
    ok = ok and same
    print("torsion %-5s flag byte %-5s: %d lanes, %d rejected, GPU call %.2f s, oracle %.1f s (%d threads) -> %s"
          % (torsion, fb, n, nf, t1 - t0, t2 - t1, os.cpu_count(), "IDENTICAL" if same else "MISMATCH at %s" % np.nonzero(st != want)[0][:10]),
          flush=True)
k = min(n, 1 << 16)       # the oracle's MSM is the naive one: a 2^16 prefix
co = rng.integers(0, 256, size=(k, 32), dtype=np.uint8); co[:, 31] &= 0x3F
for name, s_, p_ in (("honest", honest_sigs, honest_pks), ("corrupted", sigs, pks)):
    g = eng.verify_batch_msm(s_[:k], p_[:k], msgs[:k] if name == "corrupted" else msgs[:k], coeffs=co)
    # honest batch: the messages of the corrupted copy differ on kind-1 lanes; recompute with the honest ones
    if name == "honest":
        hm = msgs.copy(); hm[bad[kinds == 1], 40] ^= 0x10
        g = eng.verify_batch_msm(s_[:k], p_[:k], hm[:k], coeffs=co)
        c = orc.verify_batch_msm(s_[:k], p_[:k], hm[:k], co)
    else:
        c = orc.verify_batch_msm(s_[:k], p_[:k], msgs[:k], co)
    ok = ok and g == c
    print("MSM-form verdict, %s 2^16 prefix: GPU %d, oracle %d" % (name, g, c), flush=True)
print("full batch parity:", "ok" if ok else "FAILED")
sys.exit(0 if ok else 1)
