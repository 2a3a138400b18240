"""Generates tests/golden/msm_partials.json ON THE GPU BOX: a 41-signature batch cut into three shards, each reduced
to its 24-word record by the HIP engine (ssa_verify_batch_msm_partial), for an honest batch and for three spoiled
ones.  The CPU suite (tests/test_msm_records.py) recomputes every record with the oracle and gathers / combines them
over a world-size-2 gloo group.  Run:  python tests/golden/make_msm_partials.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import schnorr_sig_amd as ssa  # noqa: E402
from schnorr_sig_amd.sharding import shard_range  # noqa: E402


def main():
    eng = ssa.Engine(0)
    rng = np.random.default_rng(0x5C4E0300)
    n, k = 41, 3
    sks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sks[:, 31] &= 0x3F; sks[:, 0] |= 1
    nonces = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); nonces[:, 31] &= 0x3F; nonces[:, 0] |= 1
    msgs = rng.integers(0, 256, size=(n, 24), dtype=np.uint8)
    pks, sigs = eng.keygen_sign_many(sks, nonces, msgs)
    coeffs = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); coeffs[:, 31] &= 0x3F
    inf = np.zeros(n, np.uint8)
    cases = {}

    def case(name, sg, pk_inf=None):
        recs = []
        for r in range(k):
            lo, hi = shard_range(n, r, k)
            recs.append(eng.verify_batch_msm_partial(sg[lo:hi], pks[lo:hi], msgs[lo:hi], coeffs=coeffs[lo:hi],
                                                     pk_inf=None if pk_inf is None else pk_inf[lo:hi]))
        recs = np.stack(recs)
        cases[name] = {"sigs": sg.tobytes().hex(), "pk_inf": None if pk_inf is None else pk_inf.tolist(),
                       "records": [[int(v) for v in r] for r in recs],
                       "verdict": int(eng.msm_combine(recs)),
                       "single_context_verdict": int(eng.verify_batch_msm(sg, pks, msgs, coeffs=coeffs, pk_inf=pk_inf))}

    case("honest", sigs)
    bad = sigs.copy(); bad[n - 2, 55] ^= 0x10
    case("one_corrupted", bad)
    und = sigs.copy(); und[20, 48] |= 2
    case("undecodable", und)
    inf1 = inf.copy(); inf1[3] = 1
    case("identity_key", sigs, inf1)
    out = {"generator": "tests/golden/make_msm_partials.py on MI355X (library sha256 in `lib_sha256`)", "n": n, "shards": k,
           "message_bytes": 24, "pks": pks.tobytes().hex(), "msgs": msgs.tobytes().hex(), "coeffs": coeffs.tobytes().hex(),
           "cases": cases}
    import hashlib
    out["lib_sha256"] = hashlib.sha256(open(ssa.LIB_PATH, "rb").read()).hexdigest()
    path = os.path.join(os.environ.get("MSM_PARTIALS_OUT", os.path.join(ROOT, "tests", "golden")), "msm_partials.json")
    json.dump(out, open(path, "w"))
    print("wrote", path, {k_: (v["verdict"], v["single_context_verdict"]) for k_, v in cases.items()})


if __name__ == "__main__":
    main()
