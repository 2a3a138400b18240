"""Independent Python big-int model of the toposware/schnorr-sig verification path.

TEST INFRASTRUCTURE ONLY.  Nothing under schnorr-sig_amd/ may import this file;
it exists to (1) re-derive the parameter blob independently, (2) reproduce every
reference-owned fixture, (3) generate the golden vectors under tests/golden/.

PARITY STATUS: **parity unpinned** for the Rescue-Prime constants (MDS, ARK,
round count, sponge layout) and the curve generator G: the reference takes them
from the un-vendored git crates `cheetah` / `hash` (Cargo.toml:16,18) which are
not available offline.  Everything the reference's own tests pin (field tower,
curve equation, canonical limb encoding, subgroup order via the non-torsion-free
fixture, wire layouts, check ordering) is reproduced in tests/test_oracle_fixtures.py.

Algorithms are deliberately the plainest ones (affine chord-and-tangent with an
explicit inversion, bit-by-bit double-and-add, schoolbook Fp6) so that this file
shares no structure with the C oracle (oracle/schnorr_oracle.c) or the HIP path.

Reference call sites followed:
  hash_message      src/signature.rs:274-306
  Signature::verify src/signature.rs:181-205
  KeyPair::sign     src/signature.rs:114-129
  verify_batch      src/batch.rs:31-130
  wire layout       src/signature.rs:208-227, src/constants.rs:12-30
  field / curve     README.md:4-9
"""
import hashlib
import math

# ----------------------------------------------------------------------------
# Base field Fp, p = 2^64 - 2^32 + 1 (README.md:4)
# ----------------------------------------------------------------------------
P = 2**64 - 2**32 + 1

# Prime subgroup order q and cofactor (proven in
# tests/test_oracle_fixtures.py::test_group_order from the reference's fixture point).
Q = 0x7AF2599B3B3F22D0563FBF0F990A37B5327AA72330157722D443623EAED4ACCF
COFACTOR = 708537115134665106932687062569690615370

# Fp6 = Fp[u]/(u^6 - 7) (README.md:8)
NONRES = 7


def f6(*c):
    assert len(c) == 6
    return tuple(int(v) % P for v in c)


F6_ZERO = (0, 0, 0, 0, 0, 0)
F6_ONE = (1, 0, 0, 0, 0, 0)


def f6_add(a, b):
    return tuple((x + y) % P for x, y in zip(a, b))


def f6_sub(a, b):
    return tuple((x - y) % P for x, y in zip(a, b))


def f6_neg(a):
    return tuple((-x) % P for x in a)


def f6_mul(a, b):
    r = [0] * 11
    for i in range(6):
        ai = a[i]
        if ai:
            for j in range(6):
                r[i + j] += ai * b[j]
    return tuple((r[k] + (NONRES * r[k + 6] if k < 5 else 0)) % P for k in range(6))


def f6_sqr(a):
    return f6_mul(a, a)


def f6_scale(a, s):
    return tuple(x * s % P for x in a)


def f6_pow(a, e):
    r = F6_ONE
    base = a
    while e:
        if e & 1:
            r = f6_mul(r, base)
        base = f6_mul(base, base)
        e >>= 1
    return r


def f6_inv_fermat(a):
    """a^(p^6-2): the slow, obviously-correct inverse (used to cross-check f6_inv)."""
    return f6_pow(a, P**6 - 2)


# Frobenius: u^p = gamma*u with gamma = 7^((p-1)/6)
GAMMA = pow(NONRES, (P - 1) // 6, P)
_GPOW = [[pow(GAMMA, i * k, P) for i in range(6)] for k in range(6)]


def f6_frob(a, k):
    return tuple(a[i] * _GPOW[k % 6][i] % P for i in range(6))


def f6_inv(a):
    """Inverse through the norm to Fp: a^-1 = prod_{k=1..5} frob_k(a) / N(a)."""
    if a == F6_ZERO:
        raise ZeroDivisionError("Fp6 inverse of zero")
    t = f6_frob(a, 1)
    for k in range(2, 6):
        t = f6_mul(t, f6_frob(a, k))
    n = f6_mul(a, t)
    assert n[1:] == (0, 0, 0, 0, 0)
    return f6_scale(t, pow(n[0], P - 2, P))


def f6_is_square(a):
    if a == F6_ZERO:
        return True
    return f6_pow(a, (P**6 - 1) // 2) == F6_ONE


def f6_sqrt(a):
    """Tonelli-Shanks in Fp6 (2-adicity of p^6-1 is 33). Returns one root or None."""
    if a == F6_ZERO:
        return F6_ZERO
    if not f6_is_square(a):
        return None
    order = P**6 - 1
    s = 0
    t = order
    while t % 2 == 0:
        t //= 2
        s += 1
    # deterministic non-residue: smallest c such that (u + c) is a non-square
    c = 0
    while True:
        z = f6(c, 1, 0, 0, 0, 0)
        if not f6_is_square(z):
            break
        c += 1
    m = s
    cc = f6_pow(z, t)
    tt = f6_pow(a, t)
    r = f6_pow(a, (t + 1) // 2)
    while tt != F6_ONE:
        i = 0
        t2 = tt
        while t2 != F6_ONE:
            t2 = f6_sqr(t2)
            i += 1
        b = cc
        for _ in range(m - i - 1):
            b = f6_sqr(b)
        m = i
        cc = f6_sqr(b)
        tt = f6_mul(tt, cc)
        r = f6_mul(r, b)
    assert f6_sqr(r) == a
    return r


# ----------------------------------------------------------------------------
# Curve  y^2 = x^3 + x + B,  B = u + 395  (README.md:4-9).  Affine points are
# (x, y) tuples of Fp6 tuples; the identity is None.
# ----------------------------------------------------------------------------
CURVE_A = F6_ONE
CURVE_B = f6(395, 1, 0, 0, 0, 0)


def on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return f6_sqr(y) == f6_add(f6_add(f6_mul(f6_sqr(x), x), x), CURVE_B)


def pt_neg(pt):
    if pt is None:
        return None
    return (pt[0], f6_neg(pt[1]))


def pt_add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if f6_add(y1, y2) == F6_ZERO:
            return None
        # doubling (y1 == y2 != 0)
        lam = f6_mul(f6_add(f6_scale(f6_sqr(x1), 3), CURVE_A), f6_inv(f6_scale(y1, 2)))
    else:
        lam = f6_mul(f6_sub(y2, y1), f6_inv(f6_sub(x2, x1)))
    x3 = f6_sub(f6_sub(f6_sqr(lam), x1), x2)
    y3 = f6_sub(f6_mul(lam, f6_sub(x1, x3)), y1)
    return (x3, y3)


def pt_mul(k, pt):
    """Bit-by-bit left-to-right double-and-add; k is a non-negative int."""
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = pt_add(acc, acc)
        if bit == "1":
            acc = pt_add(acc, pt)
    return acc


def is_torsion_free(pt):
    """cheetah AffinePoint::is_torsion_free (called at src/signature.rs:182): [q]P == O."""
    return pt_mul(Q, pt) is None


def lex_largest(y):
    """Sort flag of a compressed point (bit 6 of byte 48): scanning c5 down to c0, the first
    non-zero coefficient is above (p-1)/2.  Convention recalled from the zkcrypto lineage the
    cheetah crate follows (Fp2::lexicographically_largest generalised); UNPINNED -- the reference
    only pins bit 7 (infinity, src/public.rs:95-101) and that 0xff is invalid (:150-156)."""
    for c in reversed(y):
        if c:
            return c > (P - 1) // 2
    return False


def pt_decompress_x(x, want_sign=None):
    """Point from an x coordinate (AffinePoint::from_compressed, src/batch.rs:104).
    Returns (x, y) with the root whose lex_largest() equals want_sign (False by default);
    None when x is not on the curve."""
    rhs = f6_add(f6_add(f6_mul(f6_sqr(x), x), x), CURVE_B)
    y = f6_sqrt(rhs)
    if y is None:
        return None
    if lex_largest(y) != bool(want_sign):
        y = f6_neg(y)
    return (x, y)


def pt_compress(pt):
    """AffinePoint::to_compressed: 48 bytes of x || flag byte (bit 7 infinity, bit 6 sort flag)."""
    if pt is None:
        return bytes(48) + bytes([0x80])
    return fp6_to_bytes48(pt[0]) + bytes([0x40 if lex_largest(pt[1]) else 0])


def pt_decompress(b49):
    """AffinePoint::from_compressed (src/public.rs:54-56, src/batch.rs:104).
    Returns ('ok', point-or-None) or ('invalid', None)."""
    flag = b49[48]
    if flag & 0x3F:
        return "invalid", None
    inf, sort = bool(flag & 0x80), bool(flag & 0x40)
    x = fp6_from_bytes48(b49[:48])
    if x is None:
        return "invalid", None
    if inf:
        return ("ok", None) if (x == F6_ZERO and not sort) else ("invalid", None)
    pt = pt_decompress_x(x, want_sign=sort)
    if pt is None:
        return "invalid", None
    return "ok", pt


# ----------------------------------------------------------------------------
# Parameter derivation (the "constants blob"): UNPINNED values.
# ----------------------------------------------------------------------------
RESCUE_M = 12          # state width     (module name rescue_64_12_8, src/signature.rs:22)
RESCUE_RATE = 8        # rate
RESCUE_CAP = 4
RESCUE_ROUNDS = 7      # recalled (Winterfell Rp64_256 lineage); the blob carries it as data
ALPHA = 7
INV_ALPHA = 10540996611094048183  # 7^-1 mod (p-1)
assert ALPHA * INV_ALPHA % (P - 1) == 1

# Circulant MDS first row: recalled from the Rp64_256 lineage, unpinned.
MDS_ROW = [7, 23, 8, 26, 13, 10, 9, 7, 6, 22, 21, 8]


def rescue_mds():
    return [[MDS_ROW[(j - i) % RESCUE_M] for j in range(RESCUE_M)] for i in range(RESCUE_M)]


def rescue_round_constants(n_rounds=RESCUE_ROUNDS):
    """Rescue-Prime paper (eprint 2020/1143, Algorithm 'get_round_constants'):
    SHAKE-256 of "Rescue-XLIX(p,m,capacity,security)" cut into (ceil(bits/8)+1)-byte
    little-endian integers reduced mod p; 2*m per round, first m = ARK1, next m = ARK2."""
    bytes_per_int = math.ceil(len(bin(P)[2:]) / 8) + 1
    n = 2 * RESCUE_M * n_rounds
    seed = "Rescue-XLIX(%i,%i,%i,%i)" % (P, RESCUE_M, RESCUE_CAP, 128)
    stream = hashlib.shake_256(seed.encode("ascii")).digest(bytes_per_int * n)
    rc = [int.from_bytes(stream[bytes_per_int * i: bytes_per_int * (i + 1)], "little") % P
          for i in range(n)]
    ark1 = [rc[2 * RESCUE_M * r: 2 * RESCUE_M * r + RESCUE_M] for r in range(n_rounds)]
    ark2 = [rc[2 * RESCUE_M * r + RESCUE_M: 2 * RESCUE_M * (r + 1)] for r in range(n_rounds)]
    return ark1, ark2


def derive_generator():
    """Deterministic stand-in for cheetah's AffinePoint::generator() (unpinned):
    smallest k >= 0 with x = k on the curve, lexicographically smaller y, cofactor cleared."""
    k = 0
    while True:
        pt = pt_decompress_x(f6(k, 0, 0, 0, 0, 0))
        if pt is not None:
            if f6_neg(pt[1]) < pt[1]:      # the blob's G was fixed with the tuple-smaller root
                pt = (pt[0], f6_neg(pt[1]))
            g = pt_mul(COFACTOR, pt)
            if g is not None:
                return g, k
        k += 1


class Params:
    """Mirror of the binary parameter blob (include/schnorr_sig_amd.h: ssa_params)."""

    def __init__(self):
        self.n_rounds = RESCUE_ROUNDS
        self.rate_off = 0          # rate = state[0..8], capacity = state[8..12]
        self.cap_len_idx = 11      # state[11] = number of absorbed felts
        self.pad_mode = 0          # 0: none (length lives in the capacity)
        self.digest_off = 0        # digest = state[0..4]
        self.mds = rescue_mds()
        self.ark1, self.ark2 = rescue_round_constants(self.n_rounds)
        self.gen = None            # filled lazily (cofactor clearing costs ~0.1 s)

    def generator(self):
        if self.gen is None:
            self.gen, _ = derive_generator()
        return self.gen


_DEFAULT = None


def default_params():
    global _DEFAULT
    if _DEFAULT is None:
        _DEFAULT = Params()
    return _DEFAULT


# ----------------------------------------------------------------------------
# Rescue-Prime 64/12/8 (hash::rescue_64_12_8, called at src/signature.rs:303-305)
# ----------------------------------------------------------------------------
def rescue_permutation(state, prm=None):
    prm = prm or default_params()
    s = list(state)
    for r in range(prm.n_rounds):
        s = [pow(v, ALPHA, P) for v in s]
        s = [(sum(prm.mds[i][j] * s[j] for j in range(RESCUE_M)) + prm.ark1[r][i]) % P
             for i in range(RESCUE_M)]
        s = [pow(v, INV_ALPHA, P) for v in s]
        s = [(sum(prm.mds[i][j] * s[j] for j in range(RESCUE_M)) + prm.ark2[r][i]) % P
             for i in range(RESCUE_M)]
    return s


def rescue_hash_field(felts, prm=None):
    """Hasher::hash_field: additive absorption into the rate, one permutation per full
    block and one for a trailing partial block; returns the 4-felt digest."""
    prm = prm or default_params()
    state = [0] * RESCUE_M
    if prm.cap_len_idx >= 0:
        state[prm.cap_len_idx] = len(felts) % P
    i = 0
    for v in felts:
        state[prm.rate_off + i] = (state[prm.rate_off + i] + v) % P
        i += 1
        if i == RESCUE_RATE:
            state = rescue_permutation(state, prm)
            i = 0
    if prm.pad_mode == 1:
        state[prm.rate_off + i] = (state[prm.rate_off + i] + 1) % P
        state = rescue_permutation(state, prm)
    elif i > 0:
        state = rescue_permutation(state, prm)
    return state[prm.digest_off: prm.digest_off + 4]


def digest_to_bytes(d):
    """Digest::to_bytes (src/signature.rs:305): 4 canonical felts, 8 bytes LE each."""
    return b"".join(int(v).to_bytes(8, "little") for v in d)


# ----------------------------------------------------------------------------
# schnorr-sig glue
# ----------------------------------------------------------------------------
def message_to_felts(message):
    """src/signature.rs:285-301: 7-byte LE chunks; a final partial chunk gets 0x01 appended
    at index chunk_len; when len % 7 == 0 (incl. empty) there is no terminator felt."""
    out = []
    nb_chunks = len(message) // 7
    for i in range(0, len(message), 7):
        chunk = message[i:i + 7]
        if i // 7 < nb_chunks:
            out.append(int.from_bytes(chunk + b"\x00", "little"))
        else:
            buf = bytearray(8)
            buf[:len(chunk)] = chunk
            buf[len(chunk)] = 1
            out.append(int.from_bytes(bytes(buf), "little"))
    return out


def hash_message(rx, pk, message, prm=None):
    """src/signature.rs:274-306: [R.x c0..c5] || [P.x c0..c5] || [P.y c0] || message felts."""
    px, py = pk
    data = list(rx) + list(px) + [py[0]] + message_to_felts(message)
    return digest_to_bytes(rescue_hash_field(data, prm))


def scalar_from_digest(h32):
    """Scalar::from_bits_vartime(h.as_bits::<Lsb0>()) (src/signature.rs:189-192):
    the 256-bit little-endian integer of the digest bytes, reduced mod q."""
    return int.from_bytes(h32, "little") % Q


OK, INVALID_PUBLIC_KEY, INVALID_SIGNATURE, MALFORMED = 0, 1, 2, 3


def fp6_from_bytes48(b):
    limbs = [int.from_bytes(b[8 * i: 8 * i + 8], "little") for i in range(6)]
    if any(v >= P for v in limbs):
        return None          # the reference's `.unwrap()` would panic (src/signature.rs:186)
    return tuple(limbs)


def fp6_to_bytes48(a):
    return b"".join(int(v).to_bytes(8, "little") for v in a)


def sign(sk, r, message, prm=None):
    """KeyPair::sign with the nonce r supplied (src/signature.rs:114-129).
    Returns (sig81 bytes, pk affine point)."""
    prm = prm or default_params()
    g = prm.generator()
    pk = pt_mul(sk, g)
    rp = pt_mul(r, g)
    h = scalar_from_digest(hash_message(rp[0], pk, message, prm))
    e = (r - sk * h) % Q
    # CompressedPoint: 48 bytes of x || flag byte (bit 7 = infinity; bit 6 = sort flag, unpinned)
    return pt_compress(rp) + e.to_bytes(32, "little"), pk


def verify(sig81, pk, message, check_torsion=True, prm=None):
    """Signature::verify (src/signature.rs:181-205). pk is an affine point (or None)."""
    prm = prm or default_params()
    if check_torsion and not is_torsion_free(pk):
        return INVALID_PUBLIC_KEY
    x_felt = fp6_from_bytes48(sig81[0:48])
    e = int.from_bytes(sig81[49:81], "little")
    if x_felt is None or e >= Q:
        return MALFORMED
    if pk is None:
        # identity public key: get_x() of the identity is taken to be 0 (unpinned)
        pkx, pky = F6_ZERO, F6_ZERO
        h = scalar_from_digest(hash_message(x_felt, (pkx, pky), message, prm))
        r = pt_mul(e, prm.generator())
    else:
        h = scalar_from_digest(hash_message(x_felt, pk, message, prm))
        r = pt_add(pt_mul(h, pk), pt_mul(e, prm.generator()))
    rx = F6_ZERO if r is None else r[0]
    return OK if rx == x_felt else INVALID_SIGNATURE


def verify_batch(sigs, pks, messages, coeffs, prm=None):
    """verify_batch (src/batch.rs:31-130) with the random coefficients supplied:
    sum s_i R_i - sum s_i h_i P_i  ?=  [sum s_i e_i] G, x-only comparison."""
    prm = prm or default_params()
    assert len(sigs) == len(pks) == len(messages) == len(coeffs)
    lin = 0
    left = None
    for sig, pk, msg, s in zip(sigs, pks, messages, coeffs):
        x_felt = fp6_from_bytes48(sig[0:48])
        assert x_felt is not None
        h = scalar_from_digest(hash_message(x_felt, pk, msg, prm))
        e = int.from_bytes(sig[49:81], "little")
        lin = (lin + s * e) % Q
        rp = pt_decompress_x(x_felt, want_sign=bool(sig[48] & 0x40))
        assert rp is not None, "from_compressed().unwrap() would panic (src/batch.rs:104)"
        left = pt_add(left, pt_mul(s % Q, rp))
        left = pt_add(left, pt_mul(s * h % Q, pt_neg(pk)))
    right = pt_mul(lin, prm.generator())
    lx = F6_ZERO if left is None else left[0]
    rx = F6_ZERO if right is None else right[0]
    return OK if lx == rx else INVALID_SIGNATURE


# The non-subgroup public key used by the reference's negative tests
# (src/signature.rs:385-406 == src/error.rs:47-64), canonical limbs.
FIXTURE_SMALL_ORDER_PK = (
    (0x9BFCD3244AFCB637, 0x39005E478830B187, 0x7046F1C03B42C6CC,
     0xB5EEAC99193711E5, 0x7FD272E724307B98, 0xCC371DD6DD5D8625),
    (0x9D03FDC216DFAAE8, 0xBF4ADE2A7665D9B8, 0xF08B022D5B3262B7,
     0x2EAF583A3CF15C6F, 0xA92531E4B1338285, 0x5B8157814141A7A7),
)

# Points of order 2 (y = 0), 5 and 10 on E(Fp6) (cofactor = 2*5*29*...), derived as [N/o]R from a
# decompressed point R; used to exercise identity entries in the per-lane multiples tables.
SMALL_ORDER_POINTS = {
    5: ((16525022225432646804, 13399409047111918848, 12960134507503110866, 9400154066480193298,
         12239491872660442389, 11775795367955543569),
        (16955858041853764987, 2630571789795685815, 7754293585774190388, 2618474437792887236,
         945460206037092674, 5566252518106206811)),
    2: ((16464216994076148022, 10762729315666779701, 13396543320389503071, 6901070379872838024,
         3684827223278792538, 13601246634833184273), (0, 0, 0, 0, 0, 0)),
    10: ((13635412684447980835, 11213008678279131991, 5977087893765706792, 974128823796173302,
          3486677043147322562, 9644894798482641728),
         (8783693457566965682, 3196204548694481917, 8820385477742708675, 5433695813968208649,
          9505825109225200576, 17799668965059873635)),
}
