#!/usr/bin/env python3
"""Generates tests/golden/vectors.json from the independent Python big-int model
(oracle/pymodel.py).  The reference itself (Rust + un-vendored git crates) cannot be built or
imported in this image, so these vectors pin the *restated* algorithm; the reference-owned
fixtures they embed (non-subgroup point, wire encodings) are copied as data from
src/signature.rs:387-404,430-460 and src/public.rs:95-101.

    python3 tests/golden/make_golden.py
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import pymodel as m  # noqa: E402

rnd = random.Random(0x5C4E0220)
P, Q = m.P, m.Q
EDGE = [0, 1, P - 1, 2**32 - 1, 2**32, P - 2**32]


def felt():
    return rnd.choice(EDGE) if rnd.random() < 0.2 else rnd.randrange(P)


def f6():
    return [felt() for _ in range(6)]


def hexb(b):
    return bytes(b).hex()


out = {"p": P, "q": Q, "cofactor": m.COFACTOR}
prm = m.default_params()
g = prm.generator()
out["generator"] = {"x": list(g[0]), "y": list(g[1])}
out["fixture_small_order_pk"] = {"x": list(m.FIXTURE_SMALL_ORDER_PK[0]), "y": list(m.FIXTURE_SMALL_ORDER_PK[1])}

# Fp6 operations
ops = []
for _ in range(40):
    a, b = f6(), f6()
    rec = {"a": a, "b": b, "mul": list(m.f6_mul(tuple(a), tuple(b))), "sqr": list(m.f6_sqr(tuple(a)))}
    if any(a):
        rec["inv"] = list(m.f6_inv_fermat(tuple(a)))
    ops.append(rec)
out["fp6"] = ops

# curve: scalar multiples of G and of the fixture point, additions incl. exceptional cases
pts = []
for k in [0, 1, 2, 3, 15, 16, 17, Q - 1, Q, Q + 1] + [rnd.randrange(2**256) for _ in range(6)]:
    r = m.pt_mul(k % (Q * m.COFACTOR), g) if k else None
    pts.append({"k": str(k), "base": "G", "res": None if r is None else [list(r[0]), list(r[1])]})
f = m.FIXTURE_SMALL_ORDER_PK
for k in [1, 2, 5, Q, Q * m.COFACTOR // 5, Q * m.COFACTOR]:
    r = m.pt_mul(k, f)
    pts.append({"k": str(k), "base": "F", "res": None if r is None else [list(r[0]), list(r[1])]})
out["scalar_mul"] = pts

# Rescue permutation and hash_field
perms = []
for _ in range(6):
    s = [felt() for _ in range(12)]
    perms.append({"in": s, "out": m.rescue_permutation(s, prm)})
out["rescue_permutation"] = perms
hf = []
for n in [0, 1, 7, 8, 9, 16, 17, 25, 36]:
    v = [felt() for _ in range(n)]
    hf.append({"in": v, "digest": m.rescue_hash_field(v, prm)})
out["hash_field"] = hf

# sign / hash_message / verify triples over the message lengths the reference's chunking cares about
sigs = []
for L in [0, 1, 6, 7, 8, 13, 14, 24, 48, 80, 160]:
    sk, r = rnd.randrange(1, Q), rnd.randrange(1, Q)
    msg = bytes(rnd.randrange(256) for _ in range(L))
    sig, pk = m.sign(sk, r, msg, prm)
    pk96 = m.fp6_to_bytes48(pk[0]) + m.fp6_to_bytes48(pk[1])
    rec = {"sk": hexb(sk.to_bytes(32, "little")), "nonce": hexb(r.to_bytes(32, "little")), "msg": hexb(msg),
           "pk": hexb(pk96), "sig": hexb(sig),
           "digest": hexb(m.hash_message(m.fp6_from_bytes48(sig[:48]), pk, msg, prm)),
           "status": m.verify(sig, pk, msg, True, prm)}
    assert rec["status"] == 0
    bad = []
    if L:
        wm = bytes([msg[0] ^ 42]) + msg[1:]
        bad.append({"what": "wrong message", "msg": hexb(wm), "pk": hexb(pk96), "sig": hexb(sig),
                    "status": m.verify(sig, pk, wm, True, prm)})
    gpk = m.fp6_to_bytes48(g[0]) + m.fp6_to_bytes48(g[1])
    bad.append({"what": "pk = generator", "msg": hexb(msg), "pk": hexb(gpk), "sig": hexb(sig),
                "status": m.verify(sig, g, msg, True, prm)})
    fpk = m.fp6_to_bytes48(f[0]) + m.fp6_to_bytes48(f[1])
    bad.append({"what": "non-subgroup pk", "msg": hexb(msg), "pk": hexb(fpk), "sig": hexb(sig),
                "status": m.verify(sig, f, msg, True, prm),
                "status_no_torsion": m.verify(sig, f, msg, False, prm)})
    idsig = bytes(48) + bytes([0x80]) + sig[49:]
    bad.append({"what": "sig.x = identity encoding", "msg": hexb(msg), "pk": hexb(pk96), "sig": hexb(idsig),
                "status": m.verify(idsig, pk, msg, True, prm)})
    e0 = sig[:49] + bytes(32)
    bad.append({"what": "e = 0", "msg": hexb(msg), "pk": hexb(pk96), "sig": hexb(e0),
                "status": m.verify(e0, pk, msg, True, prm)})
    rec["negative"] = bad
    sigs.append(rec)
out["signatures"] = sigs

# a 5-signature batch in the shape of src/batch.rs:152-179 (signers 3,4 reuse keypair 0)
msgs = [b"Message1", b"Message2", b"Message3", b"Message4", b"Message5"]
sks = [rnd.randrange(1, Q) for _ in range(3)]
sks += [sks[0], sks[0]]
bs, bp = [], []
for sk, msg in zip(sks, msgs):
    sig, pk = m.sign(sk, rnd.randrange(1, Q), msg, prm)
    bs.append(sig)
    bp.append(pk)
coeffs = [rnd.randrange(1, Q) for _ in range(5)]
ok = m.verify_batch(bs, bp, msgs, coeffs, prm)
swapped = list(bp)
swapped[1], swapped[2] = swapped[2], swapped[1]
bad = m.verify_batch(bs, swapped, msgs, coeffs, prm)
out["batch5"] = {"msgs": [hexb(x) for x in msgs], "sigs": [hexb(s) for s in bs],
                 "pks": [hexb(m.fp6_to_bytes48(p[0]) + m.fp6_to_bytes48(p[1])) for p in bp],
                 "coeffs": [hexb(c.to_bytes(32, "little")) for c in coeffs], "status": ok, "status_swapped_1_2": bad}
assert ok == 0 and bad == 2

with open(os.path.join(HERE, "vectors.json"), "w") as fh:
    json.dump(out, fh, indent=0, separators=(",", ":"))
print("wrote", os.path.join(HERE, "vectors.json"))
