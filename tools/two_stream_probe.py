#!/usr/bin/env python3
"""Do the tails of ssa_k_hash / ssa_k_verify cost anything?  One context over the whole batch against TWO contexts of the
same device (own streams), each over half of the batch, issued back to back: the second half's hash can fill the tail of the
first half's kernels.  Prints ms per 2^20 signatures for both arrangements (alternating, verify_batch flags)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import schnorr_sig_amd as ssa  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    parts = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    first = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0 / parts        # share of the batch the first context takes
    dev = torch.device("cuda", 0)
    eng = ssa.Engine(0)
    others = [ssa.Engine(0) for _ in range(parts - 1)]
    engines = [eng] + others
    sigs, pks, msgs, g = bench.gen_batch(torch, eng, dev, n, 0x5C4E0222)
    status = torch.empty(n, dtype=torch.uint8, device=dev)
    nfail = torch.zeros(parts, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    def whole():
        eng.verify_many_device(sigs.data_ptr(), pks.data_ptr(), msgs.data_ptr(), n, 80, status.data_ptr(), nfail.data_ptr(),
                               check_torsion=False, sig_flag_byte=True)

    def split():
        n0 = (int(n * first) // 256) * 256
        h = (n - n0) // max(parts - 1, 1)
        for k, e in enumerate(engines):
            lo = 0 if k == 0 else n0 + (k - 1) * h
            m = n0 if k == 0 else (h if k + 1 < parts else n - lo)
            e.verify_many_device(sigs[lo:].data_ptr(), pks[lo:].data_ptr(), msgs[lo:].data_ptr(), m, 80,
                                 status[lo:].data_ptr(), nfail[k:].data_ptr(), check_torsion=False, sig_flag_byte=True)

    def timed(fn):
        fn()
        for e in engines:
            e.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        for e in engines:
            e.sync()
        return (time.perf_counter() - t0) / reps * 1e3

    for rnd in range(3):
        a = timed(whole)
        b = timed(split)
        print("round %d: one context %.3f ms, %d contexts on %d streams (first share %.3f) %.3f ms  (rejected %s)"
              % (rnd, a, parts, parts, first, b, nfail.tolist()))


if __name__ == "__main__":
    main()
