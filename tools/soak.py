"""Differential soak: random mixed batches (tests/test_gpu_parity.py::test_fuzz_mixed_batch_vs_oracle
with fresh seeds) for a wall-clock budget; any GPU/oracle disagreement is printed and fails the run."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import schnorr_sig_amd as ssa
from oracle import Oracle
import pymodel as m

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
eng, orc = ssa.Engine(0), Oracle()
special = [m.FIXTURE_SMALL_ORDER_PK] + [m.SMALL_ORDER_POINTS[o] for o in (2, 5, 10)]
sp = [np.frombuffer(m.fp6_to_bytes48(p[0]) + m.fp6_to_bytes48(p[1]), dtype=np.uint8) for p in special]
t0, it, total = time.time(), 0, 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(1000 + it)
    n = int(rng.integers(1, 5000))
    lens = rng.integers(0, 200, size=n)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    flat = rng.integers(0, 256, size=int(off[-1]) + 1, dtype=np.uint8)
    sks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sks[:, 31] &= 0x3F; sks[:, 0] |= 1
    nonces = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); nonces[:, 31] &= 0x3F; nonces[:, 0] |= 1
    pks, sigs = eng.keygen_sign_many(sks, nonces, flat, offsets=off)
    # round 4: the constant-time signer emits the same bytes (and the oracle's signer on a prefix); keyed records;
    # PublicKey::to_bytes inverts from_bytes
    pks_c, recs = eng.keygen_sign_many(sks, nonces, flat, offsets=off, constant_time=True, keyed=True)
    m_o = min(n, 24)
    pk_o, sig_o = orc.keygen_sign_many(sks[:m_o], nonces[:m_o], flat, offsets=off[:m_o + 1])
    comp_k, st_k = eng.compress_many(pks)
    back_k, inf_k, dst_k = eng.decompress_many(comp_k)
    if not ((pks_c == pks).all() and (recs[:, 49:] == sigs).all() and (recs[:, :49] == comp_k).all() and
            (pk_o == pks[:m_o]).all() and (sig_o == sigs[:m_o]).all() and (st_k == 0).all() and (dst_k == 0).all() and
            (back_k == pks).all() and all(comp_k[i].tobytes() == orc.compress(pks[i].tobytes()) for i in range(m_o))):
        print("SIGNING / WIRE-FORM MISMATCH iteration", it)
        sys.exit(1)
    inf = np.zeros(n, dtype=np.uint8)
    kinds = rng.integers(0, 18, size=n)
    for i in np.nonzero(kinds < 10)[0]:
        k = kinds[i]
        if k == 0:
            sigs[i, 49 + rng.integers(0, 31)] ^= 1 << rng.integers(0, 8)
        elif k == 1 and lens[i] > 0:
            flat[int(off[i]) + rng.integers(0, lens[i])] ^= 1 << rng.integers(0, 8)
        elif k == 2:
            pks[i] = pks[(i + 7) % n]
        elif k == 3:
            sigs[i, :49] = sigs[(i + 3) % n, :49]
        elif k == 4:
            pks[i] = sp[rng.integers(0, 4)]
        elif k == 5:
            inf[i] = 1
        elif k == 6:
            j = rng.integers(0, 6) * 8
            sigs[i, j:j + 8] = 0xFF
        elif k == 7:
            sigs[i, 49:81] = 0xFF
        elif k == 8:
            sigs[i, rng.integers(0, 48)] ^= 1 << rng.integers(0, 7)
        elif k == 9:
            sigs[i, 48] ^= 1 << rng.integers(0, 8)          # flag byte of sig.x
    for torsion, fb in ((True, False), (False, False), (False, True)):
        want = orc.verify_many(sigs, pks, flat, offsets=off, check_torsion=torsion, pk_inf=inf, sig_flag_byte=fb)
        for mode in ("lane", "coop"):     # both kernel families
            st, nf = eng.verify_many(sigs, pks, flat, offsets=off, check_torsion=torsion, pk_inf=inf, mode=mode,
                                     sig_flag_byte=fb)
            if not (st == want).all() or nf != int((want != 0).sum()):
                bad = np.nonzero(st != want)[0]
                print("MISMATCH iteration", it, mode, "torsion", torsion, "flag byte", fb, "lanes", bad[:10], st[bad[:10]],
                      want[bad[:10]])
                sys.exit(1)
        # keyed context, both table kinds (every signature its own key here)
        if n <= 2000:
            for kind in ("ladder", "comb"):
                if kind == "comb" and n > 150:          # 100 MB of comb per key since round 4
                    continue
                ks = eng.keyset_create(pks, pk_inf=inf, kind=kind)
                st, nf = eng.verify_many_indexed(ks, np.arange(n, dtype=np.uint32), sigs, flat, offsets=off,
                                                 check_torsion=torsion, sig_flag_byte=fb)
                eng.keyset_destroy(ks)
                if not (st == want).all():
                    bad = np.nonzero(st != want)[0]
                    print("KEYED MISMATCH iteration", it, kind, "torsion", torsion, "flag byte", fb, "lanes", bad[:10],
                          st[bad[:10]], want[bad[:10]])
                    sys.exit(1)
    # MSM-form verdict on a prefix against the oracle's MSM
    k = min(n, 40)
    co = rng.integers(0, 256, size=(k, 32), dtype=np.uint8); co[:, 31] &= 0x3F
    v_gpu = eng.verify_batch_msm(sigs[:k], pks[:k], flat, offsets=off[:k + 1], coeffs=co, pk_inf=inf[:k])
    v_cpu = orc.verify_batch_msm(sigs[:k], pks[:k], flat, co, offsets=off[:k + 1], pk_inf=inf[:k])
    if v_gpu != v_cpu:
        print("MSM VERDICT MISMATCH iteration", it, v_gpu, v_cpu)
        sys.exit(1)
    # every fourth iteration: the whole batch through the bucket method (n above the small-batch switch), library-drawn
    # and caller-supplied coefficients, against the oracle's MSM
    if it % 4 == 0 and n > 3100:
        co = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); co[:, 31] &= 0x3F
        v_cpu = orc.verify_batch_msm(sigs, pks, flat, co, offsets=off, pk_inf=inf)
        v_gpu = eng.verify_batch_msm(sigs, pks, flat, offsets=off, coeffs=co, pk_inf=inf)
        v_gpu2 = eng.verify_batch_msm(sigs, pks, flat, offsets=off, pk_inf=inf)
        if not (v_gpu == v_cpu == v_gpu2):
            print("BUCKET MSM VERDICT MISMATCH iteration", it, v_gpu, v_gpu2, v_cpu)
            sys.exit(1)
    # decompression of the generated keys and of random x
    comp = np.zeros((min(n, 256), 49), dtype=np.uint8)
    comp[:, :48] = rng.integers(0, 256, size=(comp.shape[0], 48), dtype=np.uint8)
    comp[:, 7::8] &= 0x7F
    comp[:, 48] = rng.integers(0, 2, size=comp.shape[0], dtype=np.uint8) * 0x40
    out, infs, sts = eng.decompress_many(comp)
    for i in range(comp.shape[0]):
        w = orc.decompress(comp[i].tobytes())
        if (sts[i] == 0) != (w is not None) or (w is not None and out[i].tobytes() != w[0]):
            print("DECOMPRESS MISMATCH", it, i)
            sys.exit(1)
    total += n
    it += 1
    if it % 5 == 0:
        print("  ... %d iterations, %d signatures, %.0f s" % (it, total, time.time() - t0), flush=True)
print("soak ok: %d iterations, %d signatures x 3 semantics x 2 kernel families (+ keyed contexts of both kinds, "
      "MSM verdicts of both paths, constant-time signer = throughput signer = oracle signer, wire forms), %.0f s"
      % (it, total, time.time() - t0))
