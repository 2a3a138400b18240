#!/usr/bin/env python3
"""Builds the 2816-byte parameter blob (`ssa_params`, include/schnorr_sig_amd.h) from UPSTREAM's constants given
as plain text -- the one step between this engine and verifying genuine toposware signatures.

The reference takes its Rescue-Prime instance from `hash::rescue_64_12_8` and its generator from
`cheetah::AffinePoint::generator()` (reference src/signature.rs:20-24,116,303-305; src/batch.rs:98-100); both crates
are un-vendored git dependencies (Cargo.toml:16,18) and absent from the build container, so the library ships a
builder-default blob ("parity unpinned").  Nothing here reads or builds the reference: someone with the
crates at hand copies the numbers into a JSON file and runs this tool.

Input (JSON; integers as decimal or 0x-hex strings or plain numbers), either the whole file or its "constants" key:
  {
    "rounds": 7,                                 # number of Rescue rounds (<= 8)
    "mds": [[... 12 ...] x 12]  or  "mds_circulant_first_row": [... 12 ...],
    "ark1": [[... 12 ...] x rounds],             # added after the first (x^7) half-round's MDS
    "ark2": [[... 12 ...] x rounds],             # added after the second (x^(1/7)) half-round's MDS
    "sponge": {
        "rate_offset": 0 | 4,                    # index of the first rate element in the 12-felt state
        "length_index": -1 .. 11,                # state element initialised with the number of absorbed felts (-1: none)
        "pad_one": false,                        # true: a 1 is absorbed after the last felt (always at least one block)
        "digest_offset": 0 .. 8                  # first of the four digest felts
    },
    "generator": {"x": [c0..c5], "y": [c0..c5]}  # AffinePoint::generator(), canonical limbs
  }
Optional known answers for tests/test_gpu_round2.py::test_upstream_vectors_if_present, in the same file:
    "hash_field": [{"input": ["0x..", ...], "digest": "<64 hex chars: Digest::to_bytes()>"}],
    "signatures": [{"public_key": "<98 hex: PublicKey::to_bytes()>", "signature": "<162 hex>", "message": "<hex>",
                    "valid": true}]
Save that file as tests/golden/upstream_vectors.json to arm the test.

Usage:
    python3 tools/blob_from_upstream.py upstream.json -o upstream_params.bin [--check]
    --check  also runs the on-curve / subgroup test of the generator with plain Python integers
Then:  ssa_ctx_create(&ctx, device, blob, 2816)   /   schnorr_sig_amd.Engine(0, params=open(...,'rb').read())
"""
import argparse
import json
import struct
import sys

P = 2**64 - 2**32 + 1
Q = 0x7AF2599B3B3F22D0563FBF0F990A37B5327AA72330157722D443623EAED4ACCF
MAGIC = b"SSAPARM1"
M, MAX_ROUNDS, BLOB_LEN = 12, 8, 2816


def _int(v):
    return int(v, 0) if isinstance(v, str) else int(v)


def _felt(v, what):
    x = _int(v)
    if not 0 <= x < P:
        raise ValueError("%s: %d is not a canonical Goldilocks element" % (what, x))
    return x


def build_blob(spec):
    """spec: the dict described in the module docstring -> 2816 bytes"""
    spec = spec.get("constants", spec)
    rounds = _int(spec["rounds"])
    if not 1 <= rounds <= MAX_ROUNDS:
        raise ValueError("rounds must be in 1..8")
    if "mds" in spec:
        mds = [[_felt(v, "mds") for v in row] for row in spec["mds"]]
    else:
        first = [_felt(v, "mds") for v in spec["mds_circulant_first_row"]]
        if len(first) != M:
            raise ValueError("mds_circulant_first_row needs 12 entries")
        mds = [[first[(j - i) % M] for j in range(M)] for i in range(M)]
    if len(mds) != M or any(len(r) != M for r in mds):
        raise ValueError("mds must be 12 x 12")
    arks = []
    for name in ("ark1", "ark2"):
        t = [[_felt(v, name) for v in row] for row in spec[name]]
        if len(t) != rounds or any(len(r) != M for r in t):
            raise ValueError("%s must be rounds x 12" % name)
        arks.append(t + [[0] * M] * (MAX_ROUNDS - rounds))
    sp = spec.get("sponge", {})
    rate_off = _int(sp.get("rate_offset", 0))
    len_idx = _int(sp.get("length_index", -1))
    pad = 1 if sp.get("pad_one", False) else 0
    dig = _int(sp.get("digest_offset", 0))
    if rate_off not in (0, 4) or not -1 <= len_idx <= 11 or not 0 <= dig <= 8:
        raise ValueError("sponge layout out of range")
    gx = [_felt(v, "generator.x") for v in spec["generator"]["x"]]
    gy = [_felt(v, "generator.y") for v in spec["generator"]["y"]]
    if len(gx) != 6 or len(gy) != 6:
        raise ValueError("generator coordinates need 6 limbs each")
    out = bytearray(MAGIC)
    out += struct.pack("<IIiIII", rounds, rate_off, len_idx, pad, dig, 0)
    for row in mds:
        out += struct.pack("<12Q", *row)
    for t in arks:
        for row in t:
            out += struct.pack("<12Q", *row)
    out += struct.pack("<6Q", *gx) + struct.pack("<6Q", *gy)
    assert len(out) == BLOB_LEN
    return bytes(out)


def parse_blob(blob):
    """inverse of build_blob (for round-trip tests and for printing what a blob contains)"""
    if len(blob) != BLOB_LEN or blob[:8] != MAGIC:
        raise ValueError("not an ssa_params blob")
    rounds, rate_off, len_idx, pad, dig, _ = struct.unpack_from("<IIiIII", blob, 8)
    off = 32
    mds = [list(struct.unpack_from("<12Q", blob, off + 96 * i)) for i in range(M)]
    off += 96 * M
    ark1 = [list(struct.unpack_from("<12Q", blob, off + 96 * i)) for i in range(MAX_ROUNDS)]
    off += 96 * MAX_ROUNDS
    ark2 = [list(struct.unpack_from("<12Q", blob, off + 96 * i)) for i in range(MAX_ROUNDS)]
    off += 96 * MAX_ROUNDS
    gx = list(struct.unpack_from("<6Q", blob, off))
    gy = list(struct.unpack_from("<6Q", blob, off + 48))
    return {"rounds": rounds, "mds": mds, "ark1": ark1[:rounds], "ark2": ark2[:rounds],
            "sponge": {"rate_offset": rate_off, "length_index": len_idx, "pad_one": bool(pad), "digest_offset": dig},
            "generator": {"x": gx, "y": gy}}


# ---- generator sanity with plain integers (--check): y^2 = x^3 + x + (u + 395) over Fp[u]/(u^6 - 7), [q]G = O ----
def _mul(a, b):
    t = [0] * 12
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            t[i + j] += x * y
    return [(t[k] + 7 * t[k + 6]) % P for k in range(6)]


def _add(a, b):
    return [(x + y) % P for x, y in zip(a, b)]


def _sub(a, b):
    return [(x - y) % P for x, y in zip(a, b)]


def _jdbl(p):
    X, Y, Z = p
    if not any(Z):
        return p
    XX, YY, ZZ = _mul(X, X), _mul(Y, Y), _mul(Z, Z)
    S = _mul([4, 0, 0, 0, 0, 0], _mul(X, YY))
    Mm = _add(_mul([3, 0, 0, 0, 0, 0], XX), _mul(ZZ, ZZ))
    X3 = _sub(_mul(Mm, Mm), _add(S, S))
    Y3 = _sub(_mul(Mm, _sub(S, X3)), _mul([8, 0, 0, 0, 0, 0], _mul(YY, YY)))
    Z3 = _mul([2, 0, 0, 0, 0, 0], _mul(Y, Z))
    return (X3, Y3, Z3)


def _jadd(p, q):
    if not any(p[2]):
        return q
    if not any(q[2]):
        return p
    Z1Z1, Z2Z2 = _mul(p[2], p[2]), _mul(q[2], q[2])
    U1, U2 = _mul(p[0], Z2Z2), _mul(q[0], Z1Z1)
    S1, S2 = _mul(_mul(p[1], q[2]), Z2Z2), _mul(_mul(q[1], p[2]), Z1Z1)
    H, R = _sub(U2, U1), _sub(S2, S1)
    if not any(H):
        return _jdbl(p) if not any(R) else ([1, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0], [0] * 6)
    HH = _mul(H, H)
    HHH, V = _mul(H, HH), _mul(U1, HH)
    X3 = _sub(_sub(_mul(R, R), HHH), _add(V, V))
    Y3 = _sub(_mul(R, _sub(V, X3)), _mul(S1, HHH))
    return (X3, Y3, _mul(_mul(p[2], q[2]), H))


def check_generator(gx, gy):
    rhs = _add(_add(_mul(_mul(gx, gx), gx), gx), [395, 1, 0, 0, 0, 0])
    if _mul(gy, gy) != rhs:
        return "generator is not on y^2 = x^3 + x + (u + 395)"
    one = [1, 0, 0, 0, 0, 0]
    acc, base = (one, one, [0] * 6), (gx, gy, one)
    for bit in bin(Q)[2:]:
        acc = _jdbl(acc)
        if bit == "1":
            acc = _jadd(acc, base)
    if any(acc[2]):
        return "[q]G is not the identity: the generator is outside the prime-order subgroup"
    return None


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("spec", help="JSON file with upstream's constants ('-' = stdin)")
    ap.add_argument("-o", "--out", default="upstream_params.bin")
    ap.add_argument("--check", action="store_true", help="verify the generator with plain-integer arithmetic")
    args = ap.parse_args()
    spec = json.load(sys.stdin if args.spec == "-" else open(args.spec))
    blob = build_blob(spec)
    if args.check:
        c = parse_blob(blob)
        err = check_generator(c["generator"]["x"], c["generator"]["y"])
        if err:
            sys.exit("blob_from_upstream: " + err)
        print("generator: on the curve, [q]G = O")
    with open(args.out, "wb") as fh:
        fh.write(blob)
    print("wrote %s (%d bytes)" % (args.out, len(blob)))


if __name__ == "__main__":
    main()
