import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import schnorr_sig_amd as ssa
eng = ssa.Engine(0)
rng = np.random.default_rng(3)
for n in (1, 64, 256, 1024, 4096, 16384, 65536, 262144):
    sks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sks[:, 31] &= 0x3F; sks[:, 0] |= 1
    msgs = rng.integers(0, 256, size=(n, 80), dtype=np.uint8)
    pks, sigs = eng.keygen_sign_many(sks, sks[::-1].copy(), msgs)
    eng.verify_many(sigs, pks, msgs, check_torsion=True)
    t = time.perf_counter()
    for _ in range(3):
        st, nf = eng.verify_many(sigs, pks, msgs, check_torsion=True)
    dt = (time.perf_counter() - t) / 3
    assert nf == 0
    print("n=%7d  %.2f ms per call  %.0f verifications/s (Signature::verify semantics, host buffers)" % (n, dt * 1e3, n / dt))
