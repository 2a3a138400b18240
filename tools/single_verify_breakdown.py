"""Single Signature::verify through the host-buffer ABI: wall time per call vs kernel time (HIP events)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import schnorr_sig_amd as ssa

eng = ssa.Engine(0)
rng = np.random.default_rng(5)
for n in (1, 16, 128, 1024):
    sks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sks[:, 31] &= 0x3F; sks[:, 0] |= 1
    msgs = rng.integers(0, 256, size=(n, 80), dtype=np.uint8)
    pks, sigs = eng.keygen_sign_many(sks, sks[::-1].copy(), msgs)
    for tors in (True, False):
        for _ in range(3):
            eng.verify_many(sigs, pks, msgs, check_torsion=tors, mode="coop")
        reps = 20
        t = time.perf_counter()
        for _ in range(reps):
            st, nf = eng.verify_many(sigs, pks, msgs, check_torsion=tors, mode="coop")
        wall = (time.perf_counter() - t) / reps
        eng.enable_timing(True)
        for _ in range(5):
            eng.verify_many(sigs, pks, msgs, check_torsion=tors, mode="coop")
        k, cnt = eng.read_timing("ssa_k_verify_coop")
        eng.enable_timing(False)
        assert nf == 0
        print("n=%5d torsion=%d  wall %.3f ms  kernel %.3f ms (%d launches)" % (n, tors, wall * 1e3, k, cnt))
