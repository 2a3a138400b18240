"""The reference's own criterion benchmark shapes (benches/schnorr.rs:67-96) on this engine and on the
C restatement: "Verify - {8,80,160}" (one Signature::verify) and "Verify batch - {4,16,32,64,128}
signatures" (one shared 80-byte message).  Host-buffer entry points, wall time per call."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import schnorr_sig_amd as ssa
from oracle import Oracle

eng, orc = ssa.Engine(0), Oracle()
rng = np.random.default_rng(7)


def scal(n):
    s = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    s[:, 31] &= 0x3F
    s[:, 0] |= 1
    return s


def timeit(fn, reps):
    fn()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t) / reps * 1e3


print("| benchmark (benches/schnorr.rs) | MI355X engine | C restatement, 1 core |")
print("|---|---|---|")
for L in (8, 80, 160):
    msg = rng.integers(0, 256, size=(1, L), dtype=np.uint8)
    sk, nonce = scal(1), scal(1)
    pk, sig = eng.keygen_sign_many(sk, nonce, msg)
    g = timeit(lambda: eng.verify_one(sig[0].tobytes(), pk[0].tobytes(), msg[0].tobytes(), check_torsion=True), 20)
    c = timeit(lambda: orc.verify(sig[0].tobytes(), pk[0].tobytes(), msg[0].tobytes(), True), 20)
    print("| Verify - %d bytes (:67-76) | %.2f ms | %.2f ms |" % (L, g, c))
for n in (4, 16, 32, 64, 128):
    msg = np.tile(rng.integers(0, 256, size=(1, 80), dtype=np.uint8), (n, 1))
    sks, nonces = scal(n), scal(n)
    pks, sigs = eng.keygen_sign_many(sks, nonces, msg)
    coeffs = scal(n)
    g1 = timeit(lambda: eng.verify_batch_status(sigs, pks, msg), 10)
    g2 = timeit(lambda: eng.verify_batch_msm(sigs, pks, msg, coeffs=coeffs), 10)
    c = timeit(lambda: orc.verify_batch_msm(sigs, pks, msg, coeffs, threads=1), 3)
    print("| Verify batch - %d signatures (:78-96) | %.2f ms exact per-signature checks, %.2f ms MSM form | %.1f ms (MSM form) |"
          % (n, g1, g2, c))
