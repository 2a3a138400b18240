#!/usr/bin/env python3
"""Deterministic generator of the default parameter blob (`ssa_params`, see
include/schnorr_sig_amd.h) shipped inside the library.

Why a blob: the reference takes its Rescue-Prime instance and curve generator from the
un-vendored crates `hash` and `cheetah` (reference Cargo.toml:16,18), whose sources are
not available offline.  The values below are therefore a documented stand-in
("parity unpinned") and are *data*: the upstream constants can be
dropped in through ssa_ctx_create(params_blob) with no kernel change.

  * ARK1/ARK2: Rescue-Prime paper procedure (eprint 2020/1143): SHAKE-256 of
    "Rescue-XLIX(p,m,capacity,security)", 9-byte little-endian chunks mod p.
  * MDS: 12x12 circulant with first row [7,23,8,26,13,10,9,7,6,22,21,8] (recalled from the
    Rp64_256 lineage the `hash` crate is believed to derive from).
  * rounds = 7, rate = state[0..8], capacity = state[8..12], state[11] = input length,
    digest = state[0..4].
  * G: smallest integer k with x = k on y^2 = x^3 + x + (u + 395), the lexicographically
    smaller y, multiplied by the cofactor.

Self-contained on purpose (the product never imports oracle/).  Run:
    python3 gen_params.py            # rewrites params_default.bin and params_default.inc
"""
import hashlib
import os
import struct

P = 2**64 - 2**32 + 1
Q = 0x7AF2599B3B3F22D0563FBF0F990A37B5327AA72330157722D443623EAED4ACCF
H = 708537115134665106932687062569690615370
M, RATE, CAP, ROUNDS, MAX_ROUNDS = 12, 8, 4, 7, 8
MDS_ROW = [7, 23, 8, 26, 13, 10, 9, 7, 6, 22, 21, 8]
MAGIC = b"SSAPARM1"


# --- Fp6 = Fp[u]/(u^6-7), elements as lists of 6 ints -----------------------
def mul(a, b):
    t = [0] * 12
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            t[i + j] += x * y
    return [(t[k] + 7 * t[k + 6]) % P for k in range(6)]


def add(a, b):
    return [(x + y) % P for x, y in zip(a, b)]


def sub(a, b):
    return [(x - y) % P for x, y in zip(a, b)]


def powf(a, e):
    r = [1, 0, 0, 0, 0, 0]
    while e:
        if e & 1:
            r = mul(r, a)
        a = mul(a, a)
        e >>= 1
    return r


ONE = [1, 0, 0, 0, 0, 0]
ZERO = [0] * 6
B = [395, 1, 0, 0, 0, 0]


def sqrt(a):
    """Tonelli-Shanks over Fp6*; returns None for non-squares."""
    n = P**6 - 1
    if powf(a, n // 2) != ONE:
        return None
    s, t = 0, n
    while t % 2 == 0:
        s, t = s + 1, t // 2
    c = 0
    while powf([c, 1, 0, 0, 0, 0], n // 2) == ONE:
        c += 1
    z = powf([c, 1, 0, 0, 0, 0], t)
    r, tt, m = powf(a, (t + 1) // 2), powf(a, t), s
    while tt != ONE:
        i, t2 = 0, tt
        while t2 != ONE:
            t2, i = mul(t2, t2), i + 1
        b = z
        for _ in range(m - i - 1):
            b = mul(b, b)
        z = mul(b, b)
        r, tt, m = mul(r, b), mul(tt, z), i
    return r


# --- Jacobian arithmetic (a = 1); None is the identity ----------------------
def jdbl(p):
    if p is None:
        return None
    x, y, z = p
    if y == ZERO:
        return None
    yy = mul(y, y)
    s = mul([4, 0, 0, 0, 0, 0], mul(x, yy))
    zz = mul(z, z)
    m = add(mul([3, 0, 0, 0, 0, 0], mul(x, x)), mul(zz, zz))
    x3 = sub(mul(m, m), add(s, s))
    y3 = sub(mul(m, sub(s, x3)), mul([8, 0, 0, 0, 0, 0], mul(yy, yy)))
    z3 = mul(add(y, y), z)
    return (x3, y3, z3)


def jadd(p, q):
    if p is None:
        return q
    if q is None:
        return p
    x1, y1, z1 = p
    x2, y2, z2 = q
    z1z1, z2z2 = mul(z1, z1), mul(z2, z2)
    u1, u2 = mul(x1, z2z2), mul(x2, z1z1)
    s1, s2 = mul(y1, mul(z2, z2z2)), mul(y2, mul(z1, z1z1))
    if u1 == u2:
        return jdbl(p) if s1 == s2 else None
    h, r = sub(u2, u1), sub(s2, s1)
    hh = mul(h, h)
    hhh = mul(h, hh)
    v = mul(u1, hh)
    x3 = sub(sub(mul(r, r), hhh), add(v, v))
    y3 = sub(mul(r, sub(v, x3)), mul(s1, hhh))
    return (x3, y3, mul(mul(z1, z2), h))


def jmul(k, p):
    acc = None
    for bit in bin(k)[2:]:
        acc = jdbl(acc)
        if bit == "1":
            acc = jadd(acc, p)
    return acc


def to_affine(p):
    x, y, z = p
    zi = powf(z, P**6 - 2)
    zi2 = mul(zi, zi)
    return mul(x, zi2), mul(y, mul(zi, zi2))


def generator():
    k = 0
    while True:
        x = [k, 0, 0, 0, 0, 0]
        y = sqrt(add(add(mul(mul(x, x), x), x), B))
        if y is not None:
            yn = sub(ZERO, y)
            y = min(y, yn)
            g = jmul(H, (x, y, ONE))
            if g is not None:
                gx, gy = to_affine(g)
                assert jmul(Q, (gx, gy, ONE)) is None
                return gx, gy
        k += 1


def round_constants(rounds):
    nbytes = 9
    count = 2 * M * rounds
    seed = "Rescue-XLIX(%i,%i,%i,%i)" % (P, M, CAP, 128)
    stream = hashlib.shake_256(seed.encode("ascii")).digest(nbytes * count)
    rc = [int.from_bytes(stream[nbytes * i:nbytes * (i + 1)], "little") % P for i in range(count)]
    ark1 = [rc[2 * M * r:2 * M * r + M] for r in range(rounds)]
    ark2 = [rc[2 * M * r + M:2 * M * r + 2 * M] for r in range(rounds)]
    return ark1, ark2


def build_blob():
    ark1, ark2 = round_constants(ROUNDS)
    gx, gy = generator()
    out = bytearray()
    out += MAGIC
    out += struct.pack("<IIiIII", ROUNDS, 0, 11, 0, 0, 0)
    for i in range(M):
        for j in range(M):
            out += struct.pack("<Q", MDS_ROW[(j - i) % M])
    for table in (ark1, ark2):
        for r in range(MAX_ROUNDS):
            row = table[r] if r < ROUNDS else [0] * M
            out += struct.pack("<12Q", *row)
    out += struct.pack("<6Q", *gx)
    out += struct.pack("<6Q", *gy)
    assert len(out) == 2816
    return bytes(out)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    blob = build_blob()
    with open(os.path.join(here, "params_default.bin"), "wb") as f:
        f.write(blob)
    with open(os.path.join(here, "params_default.inc"), "w") as f:
        f.write("/* generated by gen_params.py -- do not edit */\n")
        for i in range(0, len(blob), 16):
            f.write(",".join("0x%02x" % b for b in blob[i:i + 16]) + ",\n")
    print("wrote %d bytes" % len(blob))


if __name__ == "__main__":
    main()
