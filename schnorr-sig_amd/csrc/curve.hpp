// Cheetah curve  y^2 = x^3 + x + (u + 395)  over Fp6 (reference README.md:4-9) and 256-bit
// scalars mod q, for the verification path (cheetah::AffinePoint::
// multiply_double_with_basepoint_vartime / is_torsion_free, called at
// src/signature.rs:182,196-198).  Jacobian coordinates, a = 1; Z == 0 encodes the identity.
//
// E(Fp6) has even order, so every addition handles P == +-Q and the identity explicitly
// (adversarial e can steer the accumulator onto a table point); doublings need no special
// case (Y == 0 or Z == 0 both give Z3 == 0).
#pragma once
#include "fp6.hpp"

namespace ssa {

struct jac {
    fp6 X, Y, Z;
};
struct aff {
    fp6 x, y;
};

SSA_DEV jac jac_identity() {
    jac r;
    r.X = f6_one();
    r.Y = f6_one();
    r.Z = f6_zero();
    return r;
}
SSA_DEV bool jac_is_identity(const jac &p) { return f6_is_zero(p.Z); }

// dbl-2007-bl with a = 1: 1M + 8S.  The additions of the formula ride in the accumulators of the products they
// follow (f6_sqr_sub2 & co.); three modular additions are left (t, yz, D).  With S' = S / 2 = 2 X Y^2:
//   S' = (X + YY)^2 - XX - YYYY      M = ZZ^2 + 3 XX      X3 = M^2 - 4 S'
//   Y3 = M (2 S' - X3) - 8 YYYY      Z3 = (Y + Z)^2 - YY - ZZ
#ifdef SSA_DBL_PLAIN
SSA_DEV jac jac_dbl(const jac &p) {
    fp6 XX = f6_sqr(p.X);
    fp6 YY = f6_sqr(p.Y);
    fp6 YYYY = f6_sqr(YY);
    fp6 ZZ = f6_sqr(p.Z);
    fp6 t = f6_add(p.X, YY);
    fp6 S = f6_dbl(f6_sub(f6_sub(f6_sqr(t), XX), YYYY));
    fp6 M = f6_add(f6_add(f6_dbl(XX), XX), f6_sqr(ZZ));
    jac r;
    r.X = f6_sub(f6_sqr(M), f6_dbl(S));
    fp6 y8 = f6_dbl(f6_dbl(f6_dbl(YYYY)));
    r.Y = f6_sub(f6_mul(M, f6_sub(S, r.X)), y8);
    fp6 yz = f6_add(p.Y, p.Z);
    r.Z = f6_sub(f6_sub(f6_sqr(yz), YY), ZZ);
    return r;
}
#else
SSA_DEV jac jac_dbl(const jac &p) {
#ifdef SSA_DBL_INLINE
    const fp6 XX = f6_sqr_inl(p.X);
    const fp6 YY = f6_sqr_inl(p.Y);
    const fp6 YYYY = f6_sqr_inl(YY);
    const fp6 ZZ = f6_sqr_inl(p.Z);
#else
    const fp6 XX = f6_sqr(p.X);
    const fp6 YY = f6_sqr(p.Y);
    const fp6 YYYY = f6_sqr(YY);
    const fp6 ZZ = f6_sqr(p.Z);
#endif
    const fp6 Sh = f6_sqr_sub2(f6_add(p.X, YY), XX, YYYY);
    const fp6 M = f6_sqr_add3x(ZZ, XX);
    jac r;
    r.X = f6_sqr_sub4x(M, Sh);
    r.Y = f6_mul_sub8x(M, f6_sub(f6_dbl(Sh), r.X), YYYY);
    r.Z = f6_sqr_sub2(f6_add(p.Y, p.Z), YY, ZZ);
    return r;
}
#endif

// n >= 1 consecutive doublings.  Device code: ONE generated asm statement (jac_asm.inc, tools/gen_jac_asm.py: the
// point in pinned VGPRs, hand-assigned temporaries, pre-scaled operands shared between the blocks, guarded short
// forms of 2a / 7a).  -DSSA_NO_JAC_ASM and the host build: the compiled doubling in a loop.
#if defined(SSA_F6_ASM) && !defined(SSA_NO_JAC_ASM)
#define SSA_JAC_ASM 1
#ifdef SSA_JAC_ASM_INC           // an alternative generated file (A/B builds: tools/build_variants.sh)
#include SSA_JAC_ASM_INC
#else
#include "jac_asm.inc"
#endif
#endif
SSA_DEV jac jac_dbl_n(jac p, u32 n) {
    if (n == 0) return p;     // the generated loop counts down to zero: n == 0 would wrap to 2^32 iterations
#ifdef SSA_JAC_ASM
    jac_dbl_n_asm(p.X.c, p.Y.c, p.Z.c, n);
    return p;
#else
#pragma unroll 1
    for (u32 d = 0; d < n; d++) p = jac_dbl(p);
    return p;
#endif
}

// the doubling as a shared out-of-line body for the rare P == Q branches of the additions (-DSSA_MADD_COLD_DBL).
// Measured and NOT the default: the call in the middle of the addition costs the register allocator more than the
// never-executed inlined copy costs in code size (ssa_k_verify 36.6 vs 35.5 ms, msm_k_buckets 4.5 vs 3.6 ms).
SSA_FN void jac_dbl_cold(jac *__restrict__ r, const jac *__restrict__ p) { *r = jac_dbl(*p); }

// mixed addition p + (x2, y2): 7M + 4S on the generic path.  The affine pair (0, 0) -- not a
// curve point since B != 0 -- is the table's encoding of the identity.  As in jac_dbl the subtractions ride in the
// accumulators of the products they follow, and R (V - X3) - Y1 HHH is ONE twelve-product accumulation.
SSA_DEV jac jac_madd(const jac &p, const aff &q) {
#ifdef SSA_MADD_PLAIN
    fp6 Z1Z1 = f6_sqr(p.Z);
    fp6 U2 = f6_mul(q.x, Z1Z1);
    fp6 S2 = f6_mul(f6_mul(q.y, p.Z), Z1Z1);
    fp6 H = f6_sub(U2, p.X);
    fp6 R = f6_sub(S2, p.Y);
#else
    const fp6 Z1Z1 = f6_sqr(p.Z);
    u64 zz7[6];
    f6_mul_prescale(Z1Z1, zz7);
    const fp6 H = f6_mul_subx_pre(q.x, Z1Z1, zz7, p.X);                   // U2 - X1
    const fp6 R = f6_mul_subx_pre(f6_mul(q.y, p.Z), Z1Z1, zz7, p.Y);      // S2 - Y1
#endif
    const bool p_inf = f6_is_zero(p.Z);
    const bool q_inf = f6_is_zero(q.x) && f6_is_zero(q.y);
    const bool h0 = f6_is_zero(H);
    const bool r0 = f6_is_zero(R);
    if (!p_inf && !q_inf && h0 && r0) {  // p == q (rare, divergent)
#ifndef SSA_MADD_COLD_DBL
        return jac_dbl(p);
#else
        jac d;
        jac_dbl_cold(&d, &p);
        return d;
#endif
    }
    fp6 HH = f6_sqr(H);
    fp6 HHH = f6_mul(H, HH);
    fp6 V = f6_mul(p.X, HH);
    jac r;
#ifdef SSA_MADD_PLAIN
    r.X = f6_sub(f6_sub(f6_sqr(R), HHH), f6_dbl(V));
    r.Y = f6_sub(f6_mul(R, f6_sub(V, r.X)), f6_mul(p.Y, HHH));
#else
    r.X = f6_sqr_subx_sub2y(R, HHH, V);
    r.Y = f6_mul2_add(R, f6_sub(V, r.X), f6_neg(p.Y), HHH);
#endif
    r.Z = f6_mul(p.Z, H);  // H == 0, R != 0  =>  Z3 = 0: p == -q gives the identity
    if (p_inf) {
        r.X = q.x;
        r.Y = q.y;
        r.Z = q_inf ? f6_zero() : f6_one();
    }
    if (q_inf && !p_inf) r = p;
    return r;
}

// The mixed addition of the hot loops: the generated asm block (jac_asm.inc) for the generic case; when it reports a
// possible exceptional input (identity, P == +-Q: a first coefficient = 0 mod p, never for honest inputs) the whole
// wave runs the compiled, exact jac_madd above on the untouched point.
SSA_DEV jac jac_madd_fast(const jac &p, const aff &q) {
#ifdef SSA_JAC_ASM
    jac r = p;
    if (jac_madd_asm(r.X.c, r.Y.c, r.Z.c, q.x.c, q.y.c)) return r;
    return jac_madd(r, q);          // r is untouched (at most canonicalised): no copy of p has to be kept
#else
    return jac_madd(p, q);
#endif
}

// general addition p + q (both Jacobian): 11M + 5S on the generic path
SSA_DEV jac jac_add(const jac &p, const jac &q) {
    fp6 Z1Z1 = f6_sqr(p.Z);
    fp6 Z2Z2 = f6_sqr(q.Z);
    fp6 U1 = f6_mul(p.X, Z2Z2);
    fp6 U2 = f6_mul(q.X, Z1Z1);
    fp6 S1 = f6_mul(f6_mul(p.Y, q.Z), Z2Z2);
    fp6 S2 = f6_mul(f6_mul(q.Y, p.Z), Z1Z1);
    fp6 H = f6_sub(U2, U1);
    fp6 R = f6_sub(S2, S1);
    const bool p_inf = f6_is_zero(p.Z);
    const bool q_inf = f6_is_zero(q.Z);
    if (!p_inf && !q_inf && f6_is_zero(H) && f6_is_zero(R)) {
#ifndef SSA_MADD_COLD_DBL
        return jac_dbl(p);
#else
        jac d;
        jac_dbl_cold(&d, &p);
        return d;
#endif
    }
    fp6 HH = f6_sqr(H);
    fp6 HHH = f6_mul(H, HH);
    fp6 V = f6_mul(U1, HH);
    jac r;
    r.X = f6_sub(f6_sub(f6_sqr(R), HHH), f6_dbl(V));
    r.Y = f6_sub(f6_mul(R, f6_sub(V, r.X)), f6_mul(S1, HHH));
    r.Z = f6_mul(f6_mul(p.Z, q.Z), H);
    if (p_inf) r = q;
    if (q_inf && !p_inf) r = p;
    return r;
}

SSA_DEV jac jac_neg_if(const jac &p, bool neg) {
    jac r = p;
    fp6 ny = f6_neg(p.Y);
    r.Y = f6_select(neg, p.Y, ny);
    return r;
}

SSA_DEV jac jac_from_aff(const aff &a) {
    jac r;
    r.X = a.x;
    r.Y = a.y;
    r.Z = f6_one();
    return r;
}

// affine x, y of a finite point (canonical limbs); the identity maps to (0, 0)
SSA_DEV aff jac_to_aff(const jac &p) {
    aff r;
    if (f6_is_zero(p.Z)) {
        r.x = f6_zero();
        r.y = f6_zero();
        return r;
    }
    fp6 zi = f6_inv(p.Z);
    fp6 zi2 = f6_sqr(zi);
    r.x = f6_canon(f6_mul(p.X, zi2));
    r.y = f6_canon(f6_mul(p.Y, f6_mul(zi, zi2)));
    return r;
}

// ------------------------------------------------------------------ scalars (4 x u64, LE)
struct sc256 {
    u64 w[4];
};
// q = 0x7af2599b3b3f22d0563fbf0f990a37b5327aa72330157722d443623eaed4accf (prime subgroup order)
SSA_DEV u64 sc_q_limb(int i) {
    switch (i) {
        case 0: return 0xd443623eaed4accfULL;
        case 1: return 0x327aa72330157722ULL;
        case 2: return 0x563fbf0f990a37b5ULL;
        default: return 0x7af2599b3b3f22d0ULL;
    }
}
#define SC_Q(i) sc_q_limb(i)

SSA_DEV bool sc_geq_q(const sc256 &a) {
    bool ge = true;  // equal so far => >=
#pragma unroll
    for (int i = 0; i < 4; i++) {  // from least to most significant: later limbs override
        if (a.w[i] > SC_Q(i)) ge = true;
        if (a.w[i] < SC_Q(i)) ge = false;
    }
    return ge;
}
SSA_DEV sc256 sc_sub_q(const sc256 &a) {
    sc256 r;
    u64 borrow = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u64 d = a.w[i] - SC_Q(i);
        u64 b1 = a.w[i] < SC_Q(i);
        u64 d2 = d - borrow;
        u64 b2 = d < borrow;
        r.w[i] = d2;
        borrow = b1 | b2;
    }
    return r;
}
// any 256-bit value mod q: floor(2^256 / q) = 2 (Scalar::from_bits_vartime, src/signature.rs:189-192)
SSA_DEV sc256 sc_reduce256(sc256 a) {
    if (sc_geq_q(a)) a = sc_sub_q(a);
    if (sc_geq_q(a)) a = sc_sub_q(a);
    return a;
}

// Offset recoding for signed 4-bit windows of a scalar k < 2^255 (true for h, e and q):
// k' = k + 0x0888...8 (63 nibbles of 8).  digit_w = nibble_w(k') - 8 in [-8, 7] for w < 63 and the
// top digit nibble_63(k') in [0, 8] is unsigned:  k = sum digit_w 16^w, no carry-out.
SSA_DEV sc256 sc_recode_offset(const sc256 &k) {
    sc256 r;
    u64 carry = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const u64 off = (i == 3) ? 0x0888888888888888ULL : 0x8888888888888888ULL;
        u64 s = k.w[i] + off;
        u64 c1 = s < k.w[i];
        u64 s2 = s + carry;
        u64 c2 = s2 < s;
        r.w[i] = s2;
        carry = c1 | c2;
    }
    return r;
}
// nibble w (0..63) of a 256-bit value with a dynamic index, registers only
SSA_DEV u32 sc_nibble(const sc256 &k, u32 w) {
    const u32 wi = w >> 4;
    u64 word = k.w[0];
    if (wi == 1) word = k.w[1];
    if (wi == 2) word = k.w[2];
    if (wi == 3) word = k.w[3];
    return (u32)(word >> ((w & 15u) * 4u)) & 15u;
}
// The same recoding for signed 5-bit windows (the ladder of the per-lane kernels, round 4: 16-entry tables):
// k' = k + sum_{w < 50} 16 * 32^w.  digit_w = window5_w(k') - 16 in [-16, 15] for w < 50; what is left above bit 250
// -- at most 32 for k < 2^255, at most 31 for a scalar below q -- is the unsigned top digit.
SSA_DEV sc256 sc_recode_offset5(const sc256 &k) {
    const u64 OFF[4] = {0x0842108421084210ULL, 0x1084210842108421ULL, 0x2108421084210842ULL, 0x0210842108421084ULL};
    sc256 r;
    u64 carry = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u64 s = k.w[i] + OFF[i];
        u64 c1 = s < k.w[i];
        u64 s2 = s + carry;
        u64 c2 = s2 < s;
        r.w[i] = s2;
        carry = c1 | c2;
    }
    return r;
}
// bits [5 w, 5 w + 5) of a 256-bit value with a dynamic index (w <= 50), registers only
SSA_DEV u32 sc_win5(const sc256 &k, u32 w) {
    const u32 bit = 5u * w, wi = bit >> 6, sh = bit & 63u;
    u64 lo = k.w[0], hi = k.w[1];
    if (wi == 1) { lo = k.w[1]; hi = k.w[2]; }
    if (wi == 2) { lo = k.w[2]; hi = k.w[3]; }
    if (wi == 3) { lo = k.w[3]; hi = 0ull; }
    u64 v = lo >> sh;
    if (sh > 59u) v |= hi << (64u - sh);
    return (u32)v & 31u;
}
SSA_DEV u32 sc_top5(const sc256 &k) { return (u32)(k.w[3] >> 58); }   // bits 250..255 of the recoded value

// bits [bit, bit + n) of a 256-bit value (n <= 32) with a dynamic position, registers only; bits above 255 read 0
SSA_DEV u32 sc_bits(const sc256 &k, u32 bit, u32 n) {
    const u32 wi = bit >> 6, sh = bit & 63u;
    u64 lo = k.w[0], hi = k.w[1];
    if (wi == 1) { lo = k.w[1]; hi = k.w[2]; }
    if (wi == 2) { lo = k.w[2]; hi = k.w[3]; }
    if (wi == 3) { lo = k.w[3]; hi = 0ull; }
    if (wi > 3) { lo = 0ull; hi = 0ull; }
    u64 v = lo >> sh;
    if (sh + n > 64u) v |= hi << (64u - sh);
    return (u32)(v & ((1ull << n) - 1ull));
}

// Curve equation check y^2 == x^3 + x + (u + 395)
SSA_DEV bool aff_on_curve(const aff &p) {
    fp6 rhs = f6_add(f6_mul(f6_sqr(p.x), p.x), p.x);
    rhs.c[0] = fp_add(rhs.c[0], 395ull);
    rhs.c[1] = fp_add(rhs.c[1], 1ull);
    return f6_eq(f6_sqr(p.y), rhs);
}

}  // namespace ssa
