"""MSM-form verify_batch: the small-batch path (one cooperative block per signature) against the bucket method, host
call to host call, over the batch sizes around the switch.  Run twice: SSA_MSM_SMALL_MAX=0 (bucket method only) and
SSA_MSM_SMALL_MAX=1000000 (small path only); prints ms per call."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import schnorr_sig_amd as ssa  # noqa: E402


def main():
    eng = ssa.Engine(0)
    rng = np.random.default_rng(9)
    out = []
    for n in (4, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096):
        sks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sks[:, 31] &= 0x3F; sks[:, 0] |= 1
        no = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); no[:, 31] &= 0x3F; no[:, 0] |= 1
        msgs = rng.integers(0, 256, size=(n, 80), dtype=np.uint8)
        pks, sigs = eng.keygen_sign_many(sks, no, msgs)
        for label, co in (("128-bit library-drawn", None), ("32-byte caller", rng.integers(0, 64, size=(n, 32), dtype=np.uint8))):
            assert eng.verify_batch_msm(sigs, pks, msgs, coeffs=co) == 0
            t0 = time.perf_counter()
            reps = 10
            for _ in range(reps):
                eng.verify_batch_msm(sigs, pks, msgs, coeffs=co)
            out.append((n, label, (time.perf_counter() - t0) / reps * 1e3))
    print("SSA_MSM_SMALL_MAX=%s" % os.environ.get("SSA_MSM_SMALL_MAX", "(default)"))
    for n, label, ms in out:
        print("n=%5d  %-22s %.3f ms" % (n, label, ms))


if __name__ == "__main__":
    main()
