"""Several host threads, each with its OWN context on one device, verifying / signing concurrently (the boundary's threading
rule: a context is used by one thread at a time, any number of contexts may run side by side -- include/schnorr_sig_amd.h;
the contexts share the generator's comb table through a reference-counted registry).  Every lane against statuses known by
construction; contexts are created and destroyed inside the threads as well.
    python tools/soak_threads.py [seconds] [threads]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import schnorr_sig_amd as ssa

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 3
fails, done = [], [0] * n_threads
keep = ssa.Engine(0)          # (holds the shared comb table while the threads' contexts come and go)


def worker(tid):
    t0, it = time.time(), 0
    eng = ssa.Engine(0)
    while time.time() - t0 < budget:
        rng = np.random.default_rng(0x7A3 + 1000 * tid + it)
        if it % 5 == 4:               # a fresh context now and then: create / destroy under load
            eng.close()
            eng = ssa.Engine(0)
        n = int(rng.integers(1 << 12, 1 << 19))
        sks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sks[:, 31] &= 0x3F; sks[:, 0] |= 1
        nonces = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); nonces[:, 31] &= 0x3F; nonces[:, 0] |= 1
        msgs = rng.integers(0, 256, size=(n, 80), dtype=np.uint8)
        pks, sigs = eng.keygen_sign_many(sks, nonces, msgs)
        bad = rng.permutation(n)[: max(1, n // 100)]
        sigs[bad, 49] ^= 1
        want = np.zeros(n, dtype=np.uint8)
        want[bad] = 2
        for torsion in (False, True):
            st, nf = eng.verify_many(sigs, pks, msgs, check_torsion=torsion, mode="lane" if it % 2 else None)
            if nf != len(bad) or not (st == want).all():
                fails.append("thread %d iteration %d: n = %d torsion %s: %d lanes differ" % (tid, it, n, torsion, int((st != want).sum())))
        if eng.verify_batch_msm(sigs, pks, msgs) != 2:
            fails.append("thread %d iteration %d: MSM form accepted a corrupted batch" % (tid, it))
        sigs[bad, 49] ^= 1
        if eng.verify_batch_msm(sigs, pks, msgs) != 0:
            fails.append("thread %d iteration %d: MSM form rejected an honest batch" % (tid, it))
        it += 1
        done[tid] += n
    eng.close()


th = [threading.Thread(target=worker, args=(k,)) for k in range(n_threads)]
t0 = time.time()
for t in th:
    t.start()
for t in th:
    t.join()
for f in fails[:20]:
    print("MISMATCH:", f)
print("thread soak %s: %d threads with their own contexts, %d signatures (x 2 semantics + 2 MSM verdicts each), %d mismatches, %.0f s"
      % ("ok" if not fails else "FAILED", n_threads, sum(done), len(fails), time.time() - t0))
sys.exit(0 if not fails else 1)
