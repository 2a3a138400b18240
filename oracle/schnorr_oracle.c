/* CPU oracle for the toposware/schnorr-sig verification path -- TEST INFRASTRUCTURE ONLY.
 * See schnorr_oracle.h for the usage rule and the parity status ("parity unpinned" for the
 * Rescue constants / generator; pinned against the reference-owned fixtures).
 *
 * Plain C (gcc, unsigned __int128).  Each section names the reference lines it restates;
 * the arithmetic itself lives upstream in the absent crates `cheetah` / `hash`
 * (reference Cargo.toml:16,18), so those parts restate the published definitions:
 *   Fp        p = 2^64 - 2^32 + 1                               README.md:4
 *   Fp6       Fp[u]/(u^6 - 7)                                   README.md:8
 *   curve     y^2 = x^3 + x + (u + 395)                         README.md:4-5
 *   Rescue    rescue_64_12_8, alpha = 7                         src/signature.rs:21-24,303-305
 */
#include "schnorr_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef uint64_t u64;

#define FP_P 0xffffffff00000001ULL
#define FP_EPS 0xffffffffULL /* 2^64 mod p */

/* ------------------------------------------------------------------ Fp (canonical u64) */
static inline u64 fp_add(u64 a, u64 b) {
    u64 s = a + b;
    if (s < a || s >= FP_P) s -= FP_P;
    return s;
}
static inline u64 fp_sub(u64 a, u64 b) { return a >= b ? a - b : a + (FP_P - b); }
static inline u64 fp_neg(u64 a) { return a ? FP_P - a : 0; }
static inline u64 fp_red128(u128 x) {
    /* x = lo + 2^64*(h0 + 2^32*h1);  2^64 = 2^32 - 1, 2^96 = -1 (mod p) */
    u64 lo = (u64)x, hi = (u64)(x >> 64);
    u64 h0 = hi & 0xffffffffULL, h1 = hi >> 32;
    u64 t = lo - h1;
    if (lo < h1) t -= FP_EPS;
    u64 m = (h0 << 32) - h0;
    u64 r = t + m;
    if (r < m) r += FP_EPS;
    if (r >= FP_P) r -= FP_P;
    return r;
}
static inline u64 fp_mul(u64 a, u64 b) { return fp_red128((u128)a * b); }
static u64 fp_pow(u64 a, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = fp_mul(r, a);
        a = fp_mul(a, a);
        e >>= 1;
    }
    return r;
}
static u64 fp_inv(u64 a) { return fp_pow(a, FP_P - 2); }

/* ------------------------------------------------------------------ Fp6 = Fp[u]/(u^6-7) */
typedef struct { u64 c[6]; } fp6;
#define FP6_ZERO_INIT {{0, 0, 0, 0, 0, 0}}
static const fp6 FP6_ZERO = {{0, 0, 0, 0, 0, 0}};
static const fp6 FP6_ONE = {{1, 0, 0, 0, 0, 0}};

static inline fp6 fp6_add(fp6 a, fp6 b) {
    fp6 r;
    for (int i = 0; i < 6; i++) r.c[i] = fp_add(a.c[i], b.c[i]);
    return r;
}
static inline fp6 fp6_sub(fp6 a, fp6 b) {
    fp6 r;
    for (int i = 0; i < 6; i++) r.c[i] = fp_sub(a.c[i], b.c[i]);
    return r;
}
static inline fp6 fp6_neg(fp6 a) {
    fp6 r;
    for (int i = 0; i < 6; i++) r.c[i] = fp_neg(a.c[i]);
    return r;
}
static inline int fp6_is_zero(fp6 a) {
    return (a.c[0] | a.c[1] | a.c[2] | a.c[3] | a.c[4] | a.c[5]) == 0;
}
static inline int fp6_eq(fp6 a, fp6 b) { return memcmp(a.c, b.c, sizeof a.c) == 0; }
static fp6 fp6_mul(fp6 a, fp6 b) {
    /* schoolbook; each product reduced to < 2^64, sums of <= 6+35 of them fit a u128 */
    u128 t[11];
    for (int k = 0; k < 11; k++) t[k] = 0;
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) t[i + j] += fp_mul(a.c[i], b.c[j]);
    fp6 r;
    for (int k = 0; k < 6; k++) r.c[k] = fp_red128(t[k] + (k < 5 ? 7 * t[k + 6] : 0));
    return r;
}
static fp6 fp6_sqr(fp6 a) { return fp6_mul(a, a); }
static fp6 fp6_muls(fp6 a, u64 s) {
    fp6 r;
    for (int i = 0; i < 6; i++) r.c[i] = fp_mul(a.c[i], s);
    return r;
}
/* Frobenius u -> gamma*u, gamma = 7^((p-1)/6); powers filled by so_init */
static u64 GPOW[6];
static fp6 fp6_frob(fp6 a, int k) {
    fp6 r;
    u64 g = 1, gk = fp_pow(GPOW[1], (u64)k);
    for (int i = 0; i < 6; i++) {
        r.c[i] = fp_mul(a.c[i], g);
        g = fp_mul(g, gk);
    }
    return r;
}
static int fp6_inv(fp6 a, fp6 *out) {
    /* a^-1 = prod_{k=1..5} frob^k(a) / Norm(a), Norm(a) in Fp */
    if (fp6_is_zero(a)) return 0;
    fp6 t = fp6_frob(a, 1);
    for (int k = 2; k < 6; k++) t = fp6_mul(t, fp6_frob(a, k));
    fp6 n = fp6_mul(a, t);
    *out = fp6_muls(t, fp_inv(n.c[0]));
    return 1;
}
static fp6 fp6_pow_limbs(fp6 a, const u64 *e, int nlimbs) {
    fp6 r = FP6_ONE;
    for (int i = nlimbs - 1; i >= 0; i--)
        for (int b = 63; b >= 0; b--) {
            r = fp6_sqr(r);
            if ((e[i] >> b) & 1) r = fp6_mul(r, a);
        }
    return r;
}
/* p^6 - 1 = 2^33 * T */
static const u64 TS_T[6] = {0x0000000a7ffffffdULL, 0x0000002cffffffe7ULL, 0x000000467fffffc1ULL,
                            0x0000002cffffffc1ULL, 0x0000000a7fffffe7ULL, 0x000000007ffffffdULL};
static const u64 TS_T1[6] = {0x800000053fffffffULL, 0x800000167ffffff3ULL, 0x800000233fffffe0ULL,
                             0x800000167fffffe0ULL, 0x800000053ffffff3ULL, 0x000000003ffffffeULL};
static const u64 TS_HALF[6] = {0x7ffffffd00000000ULL, 0xffffffe70000000aULL, 0x7fffffc10000002cULL,
                               0xffffffc100000046ULL, 0x7fffffe70000002cULL, 0x7ffffffd0000000aULL};
static int fp6_is_square(fp6 a) {
    if (fp6_is_zero(a)) return 1;
    return fp6_eq(fp6_pow_limbs(a, TS_HALF, 6), FP6_ONE);
}
static fp6 TS_Z; /* non-residue ^ T, filled by so_init */
static int fp6_sqrt(fp6 a, fp6 *out) {
    if (fp6_is_zero(a)) {
        *out = FP6_ZERO;
        return 1;
    }
    if (!fp6_is_square(a)) return 0;
    int m = 33;
    fp6 c = TS_Z, t = fp6_pow_limbs(a, TS_T, 6), r = fp6_pow_limbs(a, TS_T1, 6);
    while (!fp6_eq(t, FP6_ONE)) {
        int i = 0;
        fp6 t2 = t;
        while (!fp6_eq(t2, FP6_ONE)) {
            t2 = fp6_sqr(t2);
            i++;
        }
        fp6 b = c;
        for (int k = 0; k < m - i - 1; k++) b = fp6_sqr(b);
        m = i;
        c = fp6_sqr(b);
        t = fp6_mul(t, c);
        r = fp6_mul(r, b);
    }
    *out = r;
    return 1;
}

/* ------------------------------------------------------------------ scalars mod q (4 x u64 LE) */
typedef struct { u64 w[4]; } sc256;
static const sc256 SC_Q = {{0xd443623eaed4accfULL, 0x327aa72330157722ULL, 0x563fbf0f990a37b5ULL,
                            0x7af2599b3b3f22d0ULL}};
static int sc_geq(const sc256 *a, const sc256 *b) {
    for (int i = 3; i >= 0; i--) {
        if (a->w[i] > b->w[i]) return 1;
        if (a->w[i] < b->w[i]) return 0;
    }
    return 1;
}
static void sc_sub_raw(sc256 *a, const sc256 *b) {
    u64 borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->w[i] - b->w[i] - borrow;
        a->w[i] = (u64)d;
        borrow = (u64)(d >> 64) & 1;
    }
}
static void sc_add_raw(sc256 *a, const sc256 *b) {
    u64 carry = 0;
    for (int i = 0; i < 4; i++) {
        u128 s = (u128)a->w[i] + b->w[i] + carry;
        a->w[i] = (u64)s;
        carry = (u64)(s >> 64);
    }
}
static sc256 sc_from_bytes(const uint8_t b[32]) {
    sc256 r;
    for (int i = 0; i < 4; i++) {
        u64 v = 0;
        for (int k = 7; k >= 0; k--) v = (v << 8) | b[8 * i + k];
        r.w[i] = v;
    }
    return r;
}
static void sc_to_bytes(const sc256 *a, uint8_t b[32]) {
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 8; k++) b[8 * i + k] = (uint8_t)(a->w[i] >> (8 * k));
}
/* any 256-bit value mod q: floor(2^256/q) = 2, so at most two subtractions.
 * Scalar::from_bits_vartime(h.as_bits::<Lsb0>()), src/signature.rs:189-192 */
static sc256 sc_reduce256(sc256 a) {
    while (sc_geq(&a, &SC_Q)) sc_sub_raw(&a, &SC_Q);
    return a;
}
static sc256 sc_addmod(sc256 a, const sc256 *b) {
    sc_add_raw(&a, b); /* both < q < 2^255: no overflow */
    if (sc_geq(&a, &SC_Q)) sc_sub_raw(&a, &SC_Q);
    return a;
}
static sc256 sc_submod(sc256 a, const sc256 *b) {
    if (sc_geq(&a, b)) {
        sc_sub_raw(&a, b);
    } else {
        sc_add_raw(&a, &SC_Q);
        sc_sub_raw(&a, b);
    }
    return a;
}
static sc256 sc_mulmod(const sc256 *a, const sc256 *b) {
    u64 t[8] = {0};
    for (int i = 0; i < 4; i++) {
        u64 carry = 0;
        for (int j = 0; j < 4; j++) {
            u128 s = (u128)a->w[i] * b->w[j] + t[i + j] + carry;
            t[i + j] = (u64)s;
            carry = (u64)(s >> 64);
        }
        t[i + 4] = carry;
    }
    /* bitwise reduction of the 512-bit product (test infrastructure: clarity over speed) */
    sc256 r = {{0, 0, 0, 0}};
    for (int bit = 511; bit >= 0; bit--) {
        u64 top = r.w[3] >> 63;
        (void)top; /* r < q < 2^255, so the shift never overflows */
        r.w[3] = (r.w[3] << 1) | (r.w[2] >> 63);
        r.w[2] = (r.w[2] << 1) | (r.w[1] >> 63);
        r.w[1] = (r.w[1] << 1) | (r.w[0] >> 63);
        r.w[0] = (r.w[0] << 1) | ((t[bit >> 6] >> (bit & 63)) & 1);
        if (sc_geq(&r, &SC_Q)) sc_sub_raw(&r, &SC_Q);
    }
    return r;
}

/* ------------------------------------------------------------------ curve (Jacobian, a = 1) */
typedef struct { fp6 x, y, z; } jpt; /* z == 0: identity */
static fp6 CURVE_B;                  /* u + 395 */
static jpt J_ID;

static int j_is_id(const jpt *p) { return fp6_is_zero(p->z); }
static jpt j_dbl(const jpt *p) {
    if (j_is_id(p) || fp6_is_zero(p->y)) return J_ID; /* order-2 points double to O */
    fp6 yy = fp6_sqr(p->y);
    fp6 s = fp6_muls(fp6_mul(p->x, yy), 4);
    fp6 zz = fp6_sqr(p->z);
    fp6 m = fp6_add(fp6_muls(fp6_sqr(p->x), 3), fp6_sqr(zz)); /* 3x^2 + a z^4, a = 1 */
    jpt r;
    r.x = fp6_sub(fp6_sqr(m), fp6_add(s, s));
    r.y = fp6_sub(fp6_mul(m, fp6_sub(s, r.x)), fp6_muls(fp6_sqr(yy), 8));
    r.z = fp6_mul(fp6_add(p->y, p->y), p->z);
    return r;
}
static jpt j_add(const jpt *p, const jpt *q) {
    if (j_is_id(p)) return *q;
    if (j_is_id(q)) return *p;
    fp6 z1z1 = fp6_sqr(p->z), z2z2 = fp6_sqr(q->z);
    fp6 u1 = fp6_mul(p->x, z2z2), u2 = fp6_mul(q->x, z1z1);
    fp6 s1 = fp6_mul(p->y, fp6_mul(q->z, z2z2)), s2 = fp6_mul(q->y, fp6_mul(p->z, z1z1));
    if (fp6_eq(u1, u2)) {
        if (fp6_eq(s1, s2)) return j_dbl(p);
        return J_ID;
    }
    fp6 h = fp6_sub(u2, u1), r = fp6_sub(s2, s1);
    fp6 hh = fp6_sqr(h), hhh, v;
    hhh = fp6_mul(h, hh);
    v = fp6_mul(u1, hh);
    jpt o;
    o.x = fp6_sub(fp6_sub(fp6_sqr(r), hhh), fp6_add(v, v));
    o.y = fp6_sub(fp6_mul(r, fp6_sub(v, o.x)), fp6_mul(s1, hhh));
    o.z = fp6_mul(fp6_mul(p->z, q->z), h);
    return o;
}
static jpt j_neg(const jpt *p) {
    jpt r = *p;
    r.y = fp6_neg(p->y);
    return r;
}
static jpt j_from_affine(const u64 x[6], const u64 y[6], int inf) {
    if (inf) return J_ID;
    jpt r;
    memcpy(r.x.c, x, 48);
    memcpy(r.y.c, y, 48);
    r.z = FP6_ONE;
    return r;
}
/* returns 1 and (x,y) for finite points; 0 for the identity (x,y zeroed:
 * the identity's get_x() is taken to be 0 -- unpinned detail) */
static int j_to_affine(const jpt *p, fp6 *x, fp6 *y) {
    if (j_is_id(p)) {
        *x = FP6_ZERO;
        *y = FP6_ZERO;
        return 0;
    }
    fp6 zi;
    fp6_inv(p->z, &zi);
    fp6 zi2 = fp6_sqr(zi);
    *x = fp6_mul(p->x, zi2);
    *y = fp6_mul(p->y, fp6_mul(zi, zi2));
    return 1;
}
static void j_table16(const jpt *p, jpt tab[16]) {
    tab[0] = J_ID;
    tab[1] = *p;
    for (int i = 2; i < 16; i++) tab[i] = (i & 1) ? j_add(&tab[i - 1], p) : j_dbl(&tab[i / 2]);
}
static inline int sc_nibble(const sc256 *k, int w) { return (int)((k->w[w >> 4] >> ((w & 15) * 4)) & 15); }
static jpt G_TAB[16];
static jpt G_J;
/* [a]P + [b]G by Straus-Shamir with shared doublings and 4-bit fixed windows
 * (AffinePoint::multiply_double_with_basepoint_vartime, called at src/signature.rs:196-198) */
static jpt j_double_mul(const sc256 *a, const jpt *p, const sc256 *b) {
    jpt tab[16];
    j_table16(p, tab);
    jpt acc = J_ID;
    for (int w = 63; w >= 0; w--) {
        for (int k = 0; k < 4; k++) acc = j_dbl(&acc);
        int da = sc_nibble(a, w), db = sc_nibble(b, w);
        if (da) acc = j_add(&acc, &tab[da]);
        if (db) acc = j_add(&acc, &G_TAB[db]);
    }
    return acc;
}
static jpt j_mul(const sc256 *k, const jpt *p) {
    jpt tab[16];
    j_table16(p, tab);
    jpt acc = J_ID;
    for (int w = 63; w >= 0; w--) {
        for (int i = 0; i < 4; i++) acc = j_dbl(&acc);
        int d = sc_nibble(k, w);
        if (d) acc = j_add(&acc, &tab[d]);
    }
    return acc;
}
static jpt j_mul_base(const sc256 *k) {
    jpt acc = J_ID;
    for (int w = 63; w >= 0; w--) {
        for (int i = 0; i < 4; i++) acc = j_dbl(&acc);
        int d = sc_nibble(k, w);
        if (d) acc = j_add(&acc, &G_TAB[d]);
    }
    return acc;
}
static int pt_on_curve(fp6 x, fp6 y) {
    fp6 rhs = fp6_add(fp6_add(fp6_mul(fp6_sqr(x), x), x), CURVE_B);
    return fp6_eq(fp6_sqr(y), rhs);
}

/* ------------------------------------------------------------------ parameters / Rescue */
static struct {
    uint32_t n_rounds, rate_off;
    int32_t cap_len_idx;
    uint32_t pad_mode, digest_off;
    u64 mds[12][12], ark1[8][12], ark2[8][12];
    int ready;
} PRM;

#define INV_ALPHA 0x92492491b6db6db7ULL /* 7^-1 mod (p-1) */

static u64 rd64(const uint8_t *p) {
    u64 v = 0;
    for (int k = 7; k >= 0; k--) v = (v << 8) | p[k];
    return v;
}
static void wr64(uint8_t *p, u64 v) {
    for (int k = 0; k < 8; k++) p[k] = (uint8_t)(v >> (8 * k));
}

static void fast_init(void);
int so_init(const uint8_t *blob, size_t len) {
    if (len != 2816 || memcmp(blob, "SSAPARM1", 8) != 0) return -1;
    uint32_t hdr[6];
    memcpy(hdr, blob + 8, 24);
    PRM.n_rounds = hdr[0];
    PRM.rate_off = hdr[1];
    PRM.cap_len_idx = (int32_t)hdr[2];
    PRM.pad_mode = hdr[3];
    PRM.digest_off = hdr[4];
    if (PRM.n_rounds == 0 || PRM.n_rounds > 8 || PRM.rate_off > 4 || PRM.digest_off > 8) return -1;
    const uint8_t *p = blob + 32;
    for (int i = 0; i < 12; i++)
        for (int j = 0; j < 12; j++, p += 8) PRM.mds[i][j] = rd64(p) % FP_P;
    for (int r = 0; r < 8; r++)
        for (int j = 0; j < 12; j++, p += 8) PRM.ark1[r][j] = rd64(p) % FP_P;
    for (int r = 0; r < 8; r++)
        for (int j = 0; j < 12; j++, p += 8) PRM.ark2[r][j] = rd64(p) % FP_P;
    u64 gx[6], gy[6];
    for (int i = 0; i < 6; i++, p += 8) gx[i] = rd64(p);
    for (int i = 0; i < 6; i++, p += 8) gy[i] = rd64(p);

    u64 g = fp_pow(7, (FP_P - 1) / 6);
    GPOW[0] = 1;
    for (int i = 1; i < 6; i++) GPOW[i] = fp_mul(GPOW[i - 1], g);
    memset(&J_ID, 0, sizeof J_ID);
    J_ID.x = FP6_ONE;
    J_ID.y = FP6_ONE;
    CURVE_B = FP6_ZERO;
    CURVE_B.c[0] = 395;
    CURVE_B.c[1] = 1;
    /* Tonelli-Shanks non-residue: smallest c with (u + c) a non-square */
    for (u64 c = 0;; c++) {
        fp6 z = FP6_ZERO;
        z.c[0] = c;
        z.c[1] = 1;
        if (!fp6_is_square(z)) {
            TS_Z = fp6_pow_limbs(z, TS_T, 6);
            break;
        }
    }
    G_J = j_from_affine(gx, gy, 0);
    if (!pt_on_curve(G_J.x, G_J.y)) return -2;
    j_table16(&G_J, G_TAB);
    PRM.ready = 1;
    fast_init();      /* tables of the timing path (schnorr_oracle_fast.inc) */
    return 0;
}

void so_rescue_permutation(u64 s[12]) {
    /* round = x^7, MDS, +ARK1, x^(1/7), MDS, +ARK2 (Rescue-Prime, eprint 2020/1143 §2) */
    u64 t[12];
    for (uint32_t r = 0; r < PRM.n_rounds; r++) {
        for (int i = 0; i < 12; i++) {
            u64 x2 = fp_mul(s[i], s[i]), x4 = fp_mul(x2, x2);
            s[i] = fp_mul(fp_mul(x4, x2), s[i]);
        }
        for (int i = 0; i < 12; i++) {
            u128 acc = 0;
            for (int j = 0; j < 12; j++) acc += fp_mul(PRM.mds[i][j], s[j]);
            t[i] = fp_add(fp_red128(acc), PRM.ark1[r][i]);
        }
        for (int i = 0; i < 12; i++) s[i] = fp_pow(t[i], INV_ALPHA);
        for (int i = 0; i < 12; i++) {
            u128 acc = 0;
            for (int j = 0; j < 12; j++) acc += fp_mul(PRM.mds[i][j], s[j]);
            t[i] = fp_add(fp_red128(acc), PRM.ark2[r][i]);
        }
        memcpy(s, t, sizeof t);
    }
}

/* Hasher::hash_field (called at src/signature.rs:303): rate-8 additive sponge */
void so_hash_field(const u64 *felts, size_t n, u64 digest[4]) {
    u64 st[12] = {0};
    if (PRM.cap_len_idx >= 0) st[PRM.cap_len_idx] = (u64)n % FP_P;
    size_t i = 0;
    for (size_t k = 0; k < n; k++) {
        st[PRM.rate_off + i] = fp_add(st[PRM.rate_off + i], felts[k] % FP_P);
        if (++i == 8) {
            so_rescue_permutation(st);
            i = 0;
        }
    }
    if (PRM.pad_mode == 1) {
        st[PRM.rate_off + i] = fp_add(st[PRM.rate_off + i], 1);
        so_rescue_permutation(st);
    } else if (i > 0) {
        so_rescue_permutation(st);
    }
    for (int k = 0; k < 4; k++) digest[k] = st[PRM.digest_off + k];
}

/* hash_message, src/signature.rs:274-306 */
void so_hash_message(const uint8_t rx48[48], const uint8_t pk96[96], const uint8_t *msg, size_t len,
                     uint8_t out32[32]) {
    size_t nmsg = (len + 6) / 7; /* chunks(7): full chunks + one partial chunk */
    u64 *data = (u64 *)malloc((13 + nmsg) * sizeof(u64));
    for (int i = 0; i < 6; i++) data[i] = rd64(rx48 + 8 * i);       /* R.x      :278 */
    for (int i = 0; i < 6; i++) data[6 + i] = rd64(pk96 + 8 * i);   /* P.x      :279 */
    data[12] = rd64(pk96 + 48);                                     /* P.y[0]   :282 */
    size_t full = len / 7;
    for (size_t i = 0; i < nmsg; i++) {
        uint8_t buf[8] = {0};
        if (i < full) {
            memcpy(buf, msg + 7 * i, 7);                            /* :291-292 */
        } else {
            size_t cl = len - 7 * i;                                /* :294-298 */
            memcpy(buf, msg + 7 * i, cl);
            buf[cl] = 1;
        }
        data[13 + i] = rd64(buf);
    }
    u64 d[4];
    so_hash_field(data, 13 + nmsg, d);
    free(data);
    for (int k = 0; k < 4; k++) wr64(out32 + 8 * k, d[k]);          /* Digest::to_bytes :305 */
}

void so_scalar_from_digest(const uint8_t h32[32], uint8_t out32[32]) {
    sc256 h = sc_reduce256(sc_from_bytes(h32));
    sc_to_bytes(&h, out32);
}

/* ------------------------------------------------------------------ exported field / curve helpers */
static fp6 ld6(const u64 a[6]) {
    fp6 r;
    memcpy(r.c, a, 48);
    return r;
}
void so_fp6_mul(const u64 a[6], const u64 b[6], u64 out[6]) {
    fp6 r = fp6_mul(ld6(a), ld6(b));
    memcpy(out, r.c, 48);
}
void so_fp6_sqr(const u64 a[6], u64 out[6]) {
    fp6 r = fp6_sqr(ld6(a));
    memcpy(out, r.c, 48);
}
int so_fp6_inv(const u64 a[6], u64 out[6]) {
    fp6 r = FP6_ZERO;
    int ok = fp6_inv(ld6(a), &r);
    memcpy(out, r.c, 48);
    return ok;
}
int so_fp6_sqrt(const u64 a[6], u64 out[6]) {
    fp6 r = FP6_ZERO;
    int ok = fp6_sqrt(ld6(a), &r);
    memcpy(out, r.c, 48);
    return ok;
}
void so_point_mul(const uint8_t k32[32], const u64 px[6], const u64 py[6], int p_inf, u64 ox[6],
                  u64 oy[6], int *o_inf) {
    sc256 k = sc_from_bytes(k32);
    jpt p = j_from_affine(px, py, p_inf);
    jpt r = j_mul(&k, &p);
    fp6 x, y;
    *o_inf = !j_to_affine(&r, &x, &y);
    memcpy(ox, x.c, 48);
    memcpy(oy, y.c, 48);
}
void so_point_add(const u64 ax[6], const u64 ay[6], int a_inf, const u64 bx[6], const u64 by[6],
                  int b_inf, u64 ox[6], u64 oy[6], int *o_inf) {
    jpt a = j_from_affine(ax, ay, a_inf), b = j_from_affine(bx, by, b_inf);
    jpt r = j_add(&a, &b);
    fp6 x, y;
    *o_inf = !j_to_affine(&r, &x, &y);
    memcpy(ox, x.c, 48);
    memcpy(oy, y.c, 48);
}
/* AffinePoint::is_torsion_free (called at src/signature.rs:182): [q]P == O */
int so_is_torsion_free(const u64 px[6], const u64 py[6], int p_inf) {
    jpt p = j_from_affine(px, py, p_inf);
    jpt r = j_mul(&SC_Q, &p);
    return j_is_id(&r);
}
int so_on_curve(const u64 px[6], const u64 py[6]) { return pt_on_curve(ld6(px), ld6(py)); }

/* ------------------------------------------------------------------ schnorr-sig */
/* sort flag of a compressed point: from c5 down, the first non-zero coefficient exceeds (p-1)/2
 * (zkcrypto-style lexicographically_largest; unpinned) */
static int fp6_lex_largest(fp6 y) {
    for (int i = 5; i >= 0; i--)
        if (y.c[i]) return y.c[i] > (FP_P - 1) / 2;
    return 0;
}
static int fp6_from_bytes48(const uint8_t *b, fp6 *out) {
    for (int i = 0; i < 6; i++) {
        out->c[i] = rd64(b + 8 * i);
        if (out->c[i] >= FP_P) return 0; /* Fp6::from_bytes(..).unwrap() panics, :186 */
    }
    return 1;
}
static void fp6_to_bytes48(const fp6 *a, uint8_t *b) {
    for (int i = 0; i < 6; i++) wr64(b + 8 * i, a->c[i]);
}

/* PublicKey::from(&PrivateKey): [sk]G, src/public.rs:26-32 */
void so_keygen(const uint8_t sk32[32], uint8_t pk96[96], int *pk_inf) {
    sc256 sk = sc_reduce256(sc_from_bytes(sk32));
    jpt p = j_mul_base(&sk);
    fp6 x, y;
    *pk_inf = !j_to_affine(&p, &x, &y);
    fp6_to_bytes48(&x, pk96);
    fp6_to_bytes48(&y, pk96 + 48);
}

/* KeyPair::sign, src/signature.rs:114-129, nonce supplied instead of Scalar::random */
int so_sign(const uint8_t sk32[32], const uint8_t nonce32[32], const uint8_t pk96[96],
            const uint8_t *msg, size_t len, uint8_t sig81[81]) {
    sc256 sk = sc_reduce256(sc_from_bytes(sk32));
    sc256 r = sc_reduce256(sc_from_bytes(nonce32));
    jpt rp = j_mul_base(&r);                                        /* :116 */
    fp6 rx, ry;
    int finite = j_to_affine(&rp, &rx, &ry);
    uint8_t rx48[48], h32[32];
    fp6_to_bytes48(&rx, rx48);
    so_hash_message(rx48, pk96, msg, len, h32);                     /* :118 */
    sc256 h = sc_reduce256(sc_from_bytes(h32));                     /* :122 */
    sc256 skh = sc_mulmod(&sk, &h);
    sc256 e = sc_submod(r, &skh);                                   /* :124 */
    memcpy(sig81, rx48, 48);
    /* CompressedPoint flag byte: bit 7 infinity (src/public.rs:95-101), bit 6 sort flag (unpinned) */
    int sign = fp6_lex_largest(ry);
    sig81[48] = finite ? (sign ? 0x40 : 0) : 0x80;
    sc_to_bytes(&e, sig81 + 49);
    return 0;
}

/* Signature::verify, src/signature.rs:181-205.
 * flags bit 0: the subgroup check (:182-184; off = verify_batch semantics, src/batch.rs has none)
 * flags bit 3: verify_batch's treatment of the signature's flag byte: R is what
 *              AffinePoint::from_compressed(&sig.x).unwrap() yields (src/batch.rs:104) -- None (undecodable flag
 *              byte, x not on the curve) is the reference's panic -> SO_MALFORMED -- and the equation must hold
 *              for that point, R == [h]P + [e]G, not only for its x */
int so_verify(const uint8_t sig81[81], const uint8_t pk96[96], int pk_inf, const uint8_t *msg,
              size_t len, int flags) {
    fp6 px, py, x_felt;
    if (!fp6_from_bytes48(pk96, &px) || !fp6_from_bytes48(pk96 + 48, &py)) return SO_MALFORMED;
    if (!pk_inf && !pt_on_curve(px, py)) return SO_MALFORMED;      /* PublicKey's constructors never yield this */
    jpt p = j_from_affine(px.c, py.c, pk_inf);
    if (flags & 1) {                                                /* :182-184 */
        jpt t = j_mul(&SC_Q, &p);
        if (!j_is_id(&t)) return SO_INVALID_PUBLIC_KEY;
    }
    if (!fp6_from_bytes48(sig81, &x_felt)) return SO_MALFORMED;     /* :186 */
    sc256 e = sc_from_bytes(sig81 + 49);
    if (sc_geq(&e, &SC_Q)) return SO_MALFORMED;                     /* Scalar::from_bytes is_none */
    uint8_t rdec[96];
    int r_inf = 0;
    if ((flags & 8) && !so_decompress(sig81, rdec, &r_inf)) return SO_MALFORMED;   /* src/batch.rs:104 */
    uint8_t h32[32];
    so_hash_message(sig81, pk96, msg, len, h32);                    /* :188 */
    sc256 h = sc_reduce256(sc_from_bytes(h32));                     /* :189-192 */
    jpt r = j_double_mul(&h, &p, &e);                               /* :196-198 */
    fp6 rx, ry;
    int finite = j_to_affine(&r, &rx, &ry);
    if (flags & 8) {
        if (r_inf || !finite) return (r_inf && !finite) ? SO_OK : SO_INVALID_SIGNATURE;
        fp6 dy = FP6_ZERO_INIT;
        fp6_from_bytes48(rdec + 48, &dy);
        return (fp6_eq(rx, x_felt) && fp6_eq(ry, dy)) ? SO_OK : SO_INVALID_SIGNATURE;
    }
    return fp6_eq(rx, x_felt) ? SO_OK : SO_INVALID_SIGNATURE;       /* :200-204 */
}

static const uint8_t *msg_at(const uint8_t *msgs, const u64 *off, size_t stride, size_t msg_len,
                             size_t i, size_t *len) {
    if (off) {
        *len = (size_t)(off[i + 1] - off[i]);
        return msgs + off[i];
    }
    *len = msg_len;
    return msgs + i * stride;
}

int so_hw_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void so_verify_many(const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf,
                    const uint8_t *msgs, const u64 *off, size_t stride, size_t msg_len, size_t n,
                    int flags, int threads, uint8_t *status) {
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads)
#endif
    for (long i = 0; i < (long)n; i++) {
        size_t len;
        const uint8_t *m = msg_at(msgs, off, stride, msg_len, (size_t)i, &len);
        status[i] = (uint8_t)so_verify(sigs + 81 * i, pks + 96 * i, pk_inf ? pk_inf[i] : 0, m, len, flags);
    }
}

void so_keygen_sign_many(const uint8_t *sks, const uint8_t *nonces, const uint8_t *msgs,
                         const u64 *off, size_t stride, size_t msg_len, size_t n, int threads,
                         uint8_t *pks, uint8_t *sigs) {
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads)
#endif
    for (long i = 0; i < (long)n; i++) {
        size_t len;
        const uint8_t *m = msg_at(msgs, off, stride, msg_len, (size_t)i, &len);
        int inf;
        so_keygen(sks + 32 * i, pks + 96 * i, &inf);
        so_sign(sks + 32 * i, nonces + 32 * i, pks + 96 * i, m, len, sigs + 81 * i);
    }
}

/* verify_batch, src/batch.rs:31-130: sum s_i R_i - sum (s_i h_i) P_i ?= [sum s_i e_i] G (x-only) */
int so_verify_batch_msm(const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf, const uint8_t *msgs,
                        const u64 *off, size_t stride, size_t msg_len, size_t n, const uint8_t *coeffs,
                        int threads) {
    sc256 lin = {{0, 0, 0, 0}};
    jpt left = J_ID;
    int bad = 0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
#endif
    {
        sc256 lin_l = {{0, 0, 0, 0}};
        jpt left_l = J_ID;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 8)
#endif
        for (long i = 0; i < (long)n; i++) {
            const uint8_t *sig = sigs + 81 * i, *pk = pks + 96 * i;
            size_t len;
            const uint8_t *m = msg_at(msgs, off, stride, msg_len, (size_t)i, &len);
            fp6 x_felt, px, py;
            if (!fp6_from_bytes48(sig, &x_felt) || !fp6_from_bytes48(pk, &px) ||
                !fp6_from_bytes48(pk + 48, &py)) {
                bad = 1;                                            /* :67 unwrap panics */
                continue;
            }
            uint8_t h32[32];
            so_hash_message(sig, pk, m, len, h32);                  /* :68 */
            sc256 h = sc_reduce256(sc_from_bytes(h32));             /* :71 */
            sc256 s = sc_reduce256(sc_from_bytes(coeffs + 32 * i)); /* :77 */
            sc256 e = sc_from_bytes(sig + 49);
            const int p_inf = pk_inf && pk_inf[i];                  /* the identity is a valid PublicKey */
            if (sc_geq(&e, &SC_Q) || (!p_inf && !pt_on_curve(px, py))) {
                bad = 1;                                            /* not constructible in the reference */
                continue;
            }
            /* AffinePoint::from_compressed(&sig.x).unwrap(), :104 */
            uint8_t rdec[96];
            int r_inf = 0;
            if (!so_decompress(sig, rdec, &r_inf)) {
                bad = 1;
                continue;
            }
            sc256 se = sc_mulmod(&s, &e);                           /* :92-97 */
            lin_l = sc_addmod(lin_l, &se);
            fp6 ry = FP6_ZERO_INIT;
            fp6_from_bytes48(rdec + 48, &ry);
            jpt rp = j_from_affine(x_felt.c, ry.c, r_inf);
            jpt np = j_from_affine(px.c, py.c, p_inf);
            np = j_neg(&np);                                        /* :106 */
            sc256 sh = sc_mulmod(&s, &h);                           /* :109-111 */
            jpt t1 = j_mul(&s, &rp), t2 = j_mul(&sh, &np);          /* :123 (MSM, naive) */
            left_l = j_add(&left_l, &t1);
            left_l = j_add(&left_l, &t2);
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            lin = sc_addmod(lin, &lin_l);
            left = j_add(&left, &left_l);
        }
    }
    if (bad) return SO_MALFORMED;
    jpt right = j_mul_base(&lin);                                   /* :98-100 */
    fp6 lx, ly, rx, ry;
    j_to_affine(&left, &lx, &ly);
    j_to_affine(&right, &rx, &ry);
    return fp6_eq(lx, rx) ? SO_OK : SO_INVALID_SIGNATURE;           /* :125-129 */
}

/* AffinePoint::from_compressed (src/public.rs:54-56, src/batch.rs:104): returns 1 and the affine
 * point (pk_inf = 1 for the identity encoding [0;48] || 0x80), 0 when decompression fails. */
int so_decompress(const uint8_t c49[49], uint8_t pk96[96], int *pk_inf) {
    memset(pk96, 0, 96);
    *pk_inf = 0;
    uint8_t flag = c49[48];
    if (flag & 0x3f) return 0;
    int inf = flag >> 7, sort = (flag >> 6) & 1;
    fp6 x;
    if (!fp6_from_bytes48(c49, &x)) return 0;
    if (inf) {
        if (!fp6_is_zero(x) || sort) return 0;
        *pk_inf = 1;
        return 1;
    }
    fp6 rhs = fp6_add(fp6_add(fp6_mul(fp6_sqr(x), x), x), CURVE_B), y;
    if (!fp6_sqrt(rhs, &y)) return 0;
    if (fp6_lex_largest(y) != sort) y = fp6_neg(y);
    fp6_to_bytes48(&x, pk96);
    fp6_to_bytes48(&y, pk96 + 48);
    return 1;
}
void so_compress(const uint8_t pk96[96], int pk_inf, uint8_t c49[49]) {
    memset(c49, 0, 49);
    if (pk_inf) {
        c49[48] = 0x80;
        return;
    }
    fp6 y;
    memcpy(c49, pk96, 48);
    fp6_from_bytes48(pk96 + 48, &y);
    c49[48] = fp6_lex_largest(y) ? 0x40 : 0;
}

/* ------------------------------------------------------------------ the timing path of the CPU baseline */
#include "schnorr_oracle_fast.inc"
