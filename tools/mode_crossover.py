"""Latency of the two kernel families (one wave vs one lane per signature) across batch sizes."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import schnorr_sig_amd as ssa

eng = ssa.Engine(0)
rng = np.random.default_rng(3)
for n in (1, 512, 2048, 4096, 8192, 12288, 16384, 24576, 32768, 65536):
    sks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sks[:, 31] &= 0x3F; sks[:, 0] |= 1
    msgs = rng.integers(0, 256, size=(n, 80), dtype=np.uint8)
    pks, sigs = eng.keygen_sign_many(sks, sks[::-1].copy(), msgs)
    row = []
    for mode in ("coop", "lane"):
        for torsion in (True, False):
            eng.verify_many(sigs, pks, msgs, check_torsion=torsion, mode=mode)
            t = time.perf_counter()
            for _ in range(3):
                st, nf = eng.verify_many(sigs, pks, msgs, check_torsion=torsion, mode=mode)
            row.append((time.perf_counter() - t) / 3 * 1e3)
            assert nf == 0
    print("n=%6d  coop %.2f / %.2f ms   lane %.2f / %.2f ms   (torsion on / off)" % (n, row[0], row[1], row[2], row[3]))
