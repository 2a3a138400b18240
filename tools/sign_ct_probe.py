#!/usr/bin/env python3
"""One launch of the signer on n lanes for ONE of three secret sets -- the dynamic side of the constant-time claim
(SSA_FLAG_SIGN_CT; the reference signs with the constant-time `&BASEPOINT_TABLE * r`, src/signature.rs:67,116):

    a  uniformly random scalars         b  sparse: sk = lane + 1, nonce = 2^252 + lane + 1 (60 zero windows or more)
    c  dense: 0x6fff...f minus a few bits of the lane index (nearly every window 15)

Messages are the same in the three sets.  Run under `rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU ...`
once per set (tools/sign_ct_pmc.sh): the counters of ssa_k_sign_ct must not depend on the set, those of the
throughput signer ssa_k_sign (--vartime) do -- it skips zero windows.

The public part of the kernel differs between the sets through R and the key they hash (an S-box lane takes its
flagged fallback about once in 3 * 10^5 hashes): with n = 4096 that is ~1 % of the runs; the counters of the secret-
dependent functions themselves have no such term."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
Q = 0x7AF2599B3B3F22D0563FBF0F990A37B5327AA72330157722D443623EAED4ACCF


def secrets(which, n):
    if which == "a":
        rng = np.random.default_rng(0xC7)
        def draw():
            v = [int.from_bytes(rng.bytes(64), "little") % Q or 1 for _ in range(n)]
            return np.frombuffer(b"".join(x.to_bytes(32, "little") for x in v), np.uint8).reshape(n, 32).copy()
        return draw(), draw()
    def pack(vals):
        assert all(0 < v < Q for v in vals)
        return np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), np.uint8).reshape(n, 32).copy()
    if which == "b":      # sparse: at most four non-zero windows per scalar (the lanes still differ from each other)
        return pack([i + 1 for i in range(n)]), pack([(1 << 252) + i + 1 for i in range(n)])
    dense = (0x6F << 248) | ((1 << 248) - 1)      # dense: every window 15 but the few the lane index clears
    return pack([dense - (i << 40) for i in range(n)]), pack([dense - (i << 100) - 1 for i in range(n)])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--set", choices=["a", "b", "c"], required=True)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--vartime", action="store_true", help="the throughput signer instead (for contrast)")
    args = ap.parse_args()
    import schnorr_sig_amd as ssa
    eng = ssa.Engine(0)
    sks, nonces = secrets(args.set, args.n)
    msgs = np.random.default_rng(5).integers(0, 256, size=(args.n, 80), dtype=np.uint8)
    pks, sigs = eng.keygen_sign_many(sks, nonces, msgs, constant_time=not args.vartime)
    st, nf = eng.verify_many(sigs, pks, msgs, check_torsion=False)
    assert nf == 0
    import hashlib
    print("set %s  n %d  %s  sha256(sigs) %s" % (args.set, args.n, "vartime" if args.vartime else "constant-time",
                                                 hashlib.sha256(sigs.tobytes()).hexdigest()[:16]))


if __name__ == "__main__":
    main()
