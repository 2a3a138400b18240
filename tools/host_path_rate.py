"""PCIe-inclusive rate of the host-buffer entry point ssa_verify_many (never the bench value)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import schnorr_sig_amd as ssa

n = 1 << 20
rng = np.random.default_rng(1)
sks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sks[:, 31] &= 0x3F; sks[:, 0] |= 1
nonces = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); nonces[:, 31] &= 0x3F; nonces[:, 0] |= 1
msgs = rng.integers(0, 256, size=(n, 80), dtype=np.uint8)
eng = ssa.Engine(0)
pks, sigs = eng.keygen_sign_many(sks, nonces, msgs)
eng.verify_many(sigs, pks, msgs, check_torsion=False)
t = time.perf_counter()
for _ in range(3):
    st, nf = eng.verify_many(sigs, pks, msgs, check_torsion=False)
dt = (time.perf_counter() - t) / 3
print("host-buffer ssa_verify_many: %.1f ms per 2^20 = %.2f M verifications/s (PCIe + pageable copies included)"
      % (dt * 1e3, n / dt / 1e6))
t = time.perf_counter()
for _ in range(3):
    v = eng.verify_batch_msm(sigs, pks, msgs)
dt = (time.perf_counter() - t) / 3
print("host-buffer ssa_verify_batch_msm: %.1f ms per 2^20 = %.2f M signatures/s (incl. getrandom + copies)"
      % (dt * 1e3, n / dt / 1e6))
