import sys, numpy as np
sys.path.insert(0, '/root/repo')
import schnorr_sig_amd as ssa
eng = ssa.Engine(0)
n = 1 << 20
rng = np.random.default_rng(7)
sks = rng.integers(1, 255, size=(n, 32), dtype=np.uint8); sks[:, 31] &= 0x3f
nonces = rng.integers(1, 255, size=(n, 32), dtype=np.uint8); nonces[:, 31] &= 0x3f
msgs = rng.integers(0, 256, size=(n, 80), dtype=np.uint8)
pks, sigs = eng.keygen_sign_many(sks, nonces, msgs)
for rep in range(3):
    st, nf = eng.verify_many(sigs, pks, msgs, mode="lane")
    bad = np.nonzero(st)[0]
    print("rep", rep, "rejected", nf, "first", bad[:8], "hist by 131072:", np.bincount(bad >> 17, minlength=8))
    if len(bad):
        w = bad >> 6
        uw, cnt = np.unique(w, return_counts=True)
        print("  waves affected", len(uw), "lanes per affected wave: min %d max %d" % (cnt.min(), cnt.max()), "status values", np.unique(st[bad]))
# are the signatures themselves right?  re-sign and compare
pks2, sigs2 = eng.keygen_sign_many(sks, nonces, msgs)
print("signer deterministic:", (sigs2 == sigs).all(), (pks2 == pks).all(), "differing sigs", int((sigs2 != sigs).any(axis=1).sum()))
