"""PCIe-inclusive rate of the host-buffer entry points (what a Rust shim binds), and what the pieces cost:
pinning the caller's arrays in place (hipHostRegister / hipHostUnregister), and the number of upload chunks.
Run on the GPU box:  python tools/host_path_rate.py [n]"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20

if len(sys.argv) > 2 and sys.argv[2] == "child":
    import schnorr_sig_amd as ssa
    eng = ssa.Engine(0)
    rng = np.random.default_rng(1)
    sks = rng.integers(0, 256, (n, 32), dtype=np.uint8); sks[:, 31] &= 0x3F; sks[:, 0] |= 1
    nonces = rng.integers(0, 256, (n, 32), dtype=np.uint8); nonces[:, 31] &= 0x3F; nonces[:, 0] |= 1
    msgs = rng.integers(0, 256, (n, 80), dtype=np.uint8)
    pks, sigs = eng.keygen_sign_many(sks, nonces, msgs)
    eng.verify_many(sigs, pks, msgs, check_torsion=False, mode="lane")
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        st, nf = eng.verify_many(sigs, pks, msgs, check_torsion=False, mode="lane")
        ts.append(time.perf_counter() - t0)
    assert nf == 0
    print("chunks=%s  ssa_verify_many %.2f ms (min of 5; %.2f M verifications/s)"
          % (os.environ.get("SSA_PIPELINE_CHUNKS", "default"), min(ts) * 1e3, n / min(ts) / 1e6))
    if os.environ.get("SSA_PIPELINE_CHUNKS", "") == "4":
        hip = C.CDLL("libamdhip64.so.7")      # the SONAME: the runtime this process has already loaded, not a second copy
        hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
        hip.hipHostUnregister.argtypes = [C.c_void_p]
        for name, a in (("sigs", sigs), ("pks", pks), ("msgs", msgs)):
            t0 = time.perf_counter()
            rc = hip.hipHostRegister(a.ctypes.data, a.nbytes, 0)
            t1 = time.perf_counter()
            rc2 = hip.hipHostUnregister(a.ctypes.data)
            t2 = time.perf_counter()
            print("  hipHostRegister %-5s %6.1f MB: %.2f ms (rc %d), unregister %.2f ms (rc %d)"
                  % (name, a.nbytes / 1e6, (t1 - t0) * 1e3, rc, (t2 - t1) * 1e3, rc2))
        co = rng.integers(0, 256, (n, 32), dtype=np.uint8); co[:, 31] &= 0x3F
        eng.verify_batch_msm(sigs, pks, msgs, coeffs=co)
        t0 = time.perf_counter()
        for _ in range(3):
            v = eng.verify_batch_msm(sigs, pks, msgs, coeffs=co)
        print("  ssa_verify_batch_msm (host buffers) %.2f ms, verdict %d" % ((time.perf_counter() - t0) / 3 * 1e3, v))
    sys.exit(0)

for chunks in ("1", "2", "4", "8"):
    env = dict(os.environ, SSA_PIPELINE_CHUNKS=chunks)
    subprocess.call([sys.executable, os.path.abspath(__file__), str(n), "child"], env=env)
