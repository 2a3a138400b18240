#!/usr/bin/env python3
"""MSM-form verify_batch on 2^20 signatures resident in HBM, a few repetitions -- the subject of
`rocprofv3 --kernel-trace --stats -- python3 tools/msm_probe.py` (per-kernel times of the MSM stages)."""
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
    eng = ssa.Engine(0)
    dev = torch.device("cuda", 0)
    sigs, pks, msgs, g = bench.gen_batch(torch, eng, dev, n, 0x5C4E0222)
    coeffs = torch.randint(0, 256, (n, 16), dtype=torch.uint8, device=dev, generator=g)
    verdict = torch.full((1,), 255, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for it in range(reps + 1):
        if it == 1:
            eng.sync()
            t0 = time.perf_counter()
        eng.verify_batch_msm_device(sigs.data_ptr(), pks.data_ptr(), msgs.data_ptr(), n, 80, coeffs.data_ptr(), 16,
                                    verdict.data_ptr())
    eng.sync()
    dt = (time.perf_counter() - t0) / reps
    print("n %d: %.3f ms per batch, verdict %d" % (n, dt * 1e3, int(verdict.item())))


if __name__ == "__main__":
    main()
