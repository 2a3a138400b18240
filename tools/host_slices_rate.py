#!/usr/bin/env python3
"""Host-buffer ssa_verify_many on a batch of several slices (default 2^22 signatures = 4 slices of 2^20 lanes): rate and
device memory with the slices alternating between the context and its twin (two host threads) and on one stream.
    [SSA_TWO_STREAMS=0] python3 tools/host_slices_rate.py [log2 n] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import schnorr_sig_amd as ssa  # noqa: E402


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    n = 1 << lg
    dev = torch.device("cuda", 0)
    eng = ssa.Engine(0)
    sigs, pks, msgs, g = bench.gen_batch(torch, eng, dev, n, 0x5C4E0224)
    hs, hp, hm = sigs.cpu().numpy(), pks.cpu().numpy(), msgs.cpu().numpy()
    hs[12345, 49] ^= 1
    del sigs, pks, msgs
    torch.cuda.empty_cache()
    best = None
    for r in range(reps + 1):
        t0 = time.perf_counter()
        st, nf = eng.verify_many(hs, hp, hm, check_torsion=False, sig_flag_byte=True)
        dt = time.perf_counter() - t0
        assert nf == 1 and st[12345] == 2
        if r:
            best = dt if best is None else min(best, dt)
    info = eng.info()
    print("n = 2^%d from host buffers: %.1f ms per call, %.2f M verifications/s; two_streams %s, workspaces %.2f GB"
          % (lg, best * 1e3, n / best / 1e6, info["two_streams"], info["workspace_bytes"] / 1e9))


if __name__ == "__main__":
    main()
