// Fp6 = Fp[u]/(u^6 - 7) (reference README.md:8), coefficients c0..c5, loose Goldilocks limbs.
// Multiplication is schoolbook with lazily reduced column accumulators: 36 (mul) / 21 (sqr)
// 64x64 products, 6 reductions.  The wrap-around terms (u^6 = 7) use a pre-scaled copy of
// one operand so that every output coefficient is a single accumulation chain.
//
// Device code runs the product and the square as one generated inline-asm block each
// (fp6_asm.inc, tools/gen_f6_asm.py: accumulators in fixed caller-saved VGPRs, carries in SGPR pairs,
// the reductions interleaved three at a time so that no carry needs s_nop padding).  Measured against
// the C++ formulation below, which stays as the host build (tests/csrc/host_arith.cpp) and as the
// -DSSA_NO_F6_ASM fallback: ssa_k_verify 41.25 -> 38.85 ms at 2^20, lazy-Fp6 probe 2.80 -> 2.96e12 Fp-mul/s.
#pragma once
#include "fp.hpp"

namespace ssa {

#if defined(__HIP_DEVICE_COMPILE__) && !defined(SSA_NO_F6_ASM)
#define SSA_F6_ASM 1
#include "fp6_asm.inc"
#endif

struct fp6 {
    u64 c[6];
};

SSA_DEV fp6 f6_zero() {
    fp6 r;
#pragma unroll
    for (int i = 0; i < 6; i++) r.c[i] = 0ull;
    return r;
}
SSA_DEV fp6 f6_one() {
    fp6 r = f6_zero();
    r.c[0] = 1ull;
    return r;
}
SSA_DEV fp6 f6_add(const fp6 &a, const fp6 &b) {
    fp6 r;
#pragma unroll
    for (int i = 0; i < 6; i++) r.c[i] = fp_add(a.c[i], b.c[i]);
    return r;
}
SSA_DEV fp6 f6_sub(const fp6 &a, const fp6 &b) {
    fp6 r;
#pragma unroll
    for (int i = 0; i < 6; i++) r.c[i] = fp_sub(a.c[i], b.c[i]);
    return r;
}
SSA_DEV fp6 f6_dbl(const fp6 &a) {
    fp6 r;
#pragma unroll
    for (int i = 0; i < 6; i++) r.c[i] = fp_dbl(a.c[i]);
    return r;
}
SSA_DEV fp6 f6_neg(const fp6 &a) {
    fp6 r;
#pragma unroll
    for (int i = 0; i < 6; i++) r.c[i] = fp_neg(a.c[i]);
    return r;
}
SSA_DEV bool f6_is_zero(const fp6 &a) {
    bool z = true;
#pragma unroll
    for (int i = 0; i < 6; i++) z = z && fp_is_zero(a.c[i]);
    return z;
}
SSA_DEV bool f6_eq(const fp6 &a, const fp6 &b) {
    bool e = true;
#pragma unroll
    for (int i = 0; i < 6; i++) e = e && fp_eq(a.c[i], b.c[i]);
    return e;
}
SSA_DEV fp6 f6_canon(const fp6 &a) {
    fp6 r;
#pragma unroll
    for (int i = 0; i < 6; i++) r.c[i] = fp_canon(a.c[i]);
    return r;
}
SSA_DEV fp6 f6_select(bool pick_b, const fp6 &a, const fp6 &b) {
    fp6 r;
#pragma unroll
    for (int i = 0; i < 6; i++) r.c[i] = pick_b ? b.c[i] : a.c[i];
    return r;
}
SSA_DEV fp6 f6_mul_small(const fp6 &a, u32 k) {
    fp6 r;
#pragma unroll
    for (int i = 0; i < 6; i++) r.c[i] = fp_mul_small(a.c[i], k);
    return r;
}
SSA_DEV fp6 f6_mul_fp(const fp6 &a, u64 s) {
    fp6 r;
#pragma unroll
    for (int i = 0; i < 6; i++) r.c[i] = fp_mul(a.c[i], s);
    return r;
}

// c_k = sum_{i+j=k} a_i b_j + 7 sum_{i+j=k+6} a_i b_j
// (flat scalar arguments: a by-value struct pair would be passed through scratch memory)
SSA_FN fp6 f6_mul_flat(u64 a0, u64 a1, u64 a2, u64 a3, u64 a4, u64 a5, u64 b0, u64 b1, u64 b2, u64 b3,
                       u64 b4, u64 b5) {
    fp6 a, b;
    a.c[0] = a0; a.c[1] = a1; a.c[2] = a2; a.c[3] = a3; a.c[4] = a4; a.c[5] = a5;
    b.c[0] = b0; b.c[1] = b1; b.c[2] = b2; b.c[3] = b3; b.c[4] = b4; b.c[5] = b5;
    u64 b7[6];
    b7[0] = 0ull;
#pragma unroll
    for (int j = 1; j < 6; j++) b7[j] = fp_mul_small(b.c[j], 7u);
    fp6 r;
#ifdef SSA_F6_ASM
    f6_mul_core_asm(a.c, b.c, b7, r.c);
#else
#pragma unroll
    for (int k = 0; k < 6; k++) {
        fp_acc s;
        acc_init(s, a.c[0], b.c[k]);
#pragma unroll
        for (int i = 1; i < 6; i++) {
            if (i <= k)
                acc_mac(s, a.c[i], b.c[k - i]);
            else
                acc_mac(s, a.c[i], b7[k + 6 - i]);
        }
        r.c[k] = acc_reduce(s);
    }
#endif
    return r;
}
SSA_DEV fp6 f6_mul(const fp6 &a, const fp6 &b) {
    return f6_mul_flat(a.c[0], a.c[1], a.c[2], a.c[3], a.c[4], a.c[5], b.c[0], b.c[1], b.c[2], b.c[3],
                       b.c[4], b.c[5]);
}

// Pre-scaled operands of the squaring blocks: a2[j] = 2 a[j] (j = 1..5), a7[j] = 7 a[j] (j = 3..5), a14[j] = 14 a[j]
// (j = 4, 5).  Doubling a loose value is a shift plus EPS for the top bit, and the sum wraps a second time only for
// a >= 2^64 - 2^31, i.e. when the high word is all ones: one max over the seven high words guards the two-instruction
// doubling, the three-instruction second fix-up of fp_dbl runs in the (never taken) cold branch.
SSA_DEV u64 fp_dbl_nowrap(u64 a) { return (a << 1) + (u64)(u32)((int)hi32(a) >> 31); }
SSA_DEV void f6_sqr_scale(const u64 (&a)[6], u64 (&a2)[6], u64 (&a7)[6], u64 (&a14)[6]) {
    a2[0] = 0ull;
#pragma unroll
    for (int j = 0; j < 3; j++) a7[j] = a14[j] = 0ull;
    a14[3] = 0ull;     // no term uses it
#pragma unroll
    for (int j = 3; j < 6; j++) a7[j] = fp_mul_small(a[j], 7u);
#ifdef SSA_PLAIN_PRESCALE
#pragma unroll
    for (int j = 1; j < 6; j++) a2[j] = fp_dbl(a[j]);
#pragma unroll
    for (int j = 4; j < 6; j++) a14[j] = fp_dbl(a7[j]);
#else
    u32 g = hi32(a[1]);
#pragma unroll
    for (int j = 2; j < 6; j++) g = g > hi32(a[j]) ? g : hi32(a[j]);
#pragma unroll
    for (int j = 4; j < 6; j++) g = g > hi32(a7[j]) ? g : hi32(a7[j]);
#pragma unroll
    for (int j = 1; j < 6; j++) a2[j] = fp_dbl_nowrap(a[j]);
#pragma unroll
    for (int j = 4; j < 6; j++) a14[j] = fp_dbl_nowrap(a7[j]);
    if (__builtin_expect(g == 0xffffffffu, 0)) {
#pragma unroll
        for (int j = 1; j < 6; j++) a2[j] = fp_dbl(a[j]);
#pragma unroll
        for (int j = 4; j < 6; j++) a14[j] = fp_dbl(a7[j]);
    }
#endif
}

// squaring: 21 products; off-diagonal terms use 2a_j (direct) or 14a_j (wrapped),
// diagonal terms a_i (direct) or 7a_i (wrapped).
SSA_FN fp6 f6_sqr_flat(u64 a0, u64 a1, u64 a2_, u64 a3, u64 a4, u64 a5) {
    fp6 a;
    a.c[0] = a0; a.c[1] = a1; a.c[2] = a2_; a.c[3] = a3; a.c[4] = a4; a.c[5] = a5;
    u64 a2[6], a7[6], a14[6];
    f6_sqr_scale(a.c, a2, a7, a14);
#ifdef SSA_F6_ASM
    fp6 r;
    f6_sqr_core_asm(a.c, a2, a7, a14, r.c);
    return r;
#else
    fp6 r;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        fp_acc s;
        bool first = true;
        // direct terms i + j = k, i <= j
#pragma unroll
        for (int i = 0; i <= k / 2; i++) {
            int j = k - i;
            u64 y = (i == j) ? a.c[j] : a2[j];
            if (first) {
                acc_init(s, a.c[i], y);
                first = false;
            } else {
                acc_mac(s, a.c[i], y);
            }
        }
        // wrapped terms i + j = k + 6, i <= j <= 5
#pragma unroll
        for (int i = k + 1; i <= (k + 6) / 2; i++) {
            int j = k + 6 - i;
            if (j > 5) continue;
            u64 y = (i == j) ? a7[j] : a14[j];
            acc_mac(s, a.c[i], y);
        }
        r.c[k] = acc_reduce(s);
    }
    return r;
#endif
}
SSA_DEV fp6 f6_sqr(const fp6 &a) { return f6_sqr_flat(a.c[0], a.c[1], a.c[2], a.c[3], a.c[4], a.c[5]); }

// the plain product and square as inlined blocks (no call, no argument moves): for the few hottest call sites only --
// every inlined copy is ~3 KB of code (SSA_DBL_INLINE, measured in DESIGN.md)
SSA_DEV fp6 f6_sqr_inl(const fp6 &a) {
#ifdef SSA_F6_ASM
    u64 a2[6], a7[6], a14[6];
    f6_sqr_scale(a.c, a2, a7, a14);
    fp6 r;
    f6_sqr_core_asm(a.c, a2, a7, a14, r.c);
    return r;
#else
    return f6_sqr(a);
#endif
}

// ---- products fused with the additions that follow them in the point formulas -------------------------------
// r = a^2 (or a*b) + sum of small multiples of other elements, the linear terms added to the accumulator columns
// before the single reduction (fp6_asm.inc, "fused linear terms": 4-6 instructions per coefficient and term instead
// of the ~10 of a modular addition).  Inlined at their call sites (one each in jac_dbl / jac_madd): no call, no
// argument moves.  Host build and -DSSA_NO_F6_ASM: the same value from the plain operations.
SSA_DEV void f6_sqr_prescale(const fp6 &a, u64 (&a2)[6], u64 (&a7)[6], u64 (&a14)[6]) {
    f6_sqr_scale(a.c, a2, a7, a14);
}
SSA_DEV void f6_mul_prescale(const fp6 &b, u64 (&b7)[6]) {
    b7[0] = 0ull;
#pragma unroll
    for (int j = 1; j < 6; j++) b7[j] = fp_mul_small(b.c[j], 7u);
}
// a^2 - x - y
SSA_DEV fp6 f6_sqr_sub2(const fp6 &a, const fp6 &x, const fp6 &y) {
#ifdef SSA_F6_ASM
    u64 a2[6], a7[6], a14[6];
    f6_sqr_prescale(a, a2, a7, a14);
    fp6 r;
    f6_sqr_sub2_core_asm(a.c, a2, a7, a14, x.c, y.c, r.c);
    return r;
#else
    return f6_sub(f6_sub(f6_sqr(a), x), y);
#endif
}
// a^2 - x - 2y
SSA_DEV fp6 f6_sqr_subx_sub2y(const fp6 &a, const fp6 &x, const fp6 &y) {
#ifdef SSA_F6_ASM
    u64 a2[6], a7[6], a14[6];
    f6_sqr_prescale(a, a2, a7, a14);
    fp6 r;
    f6_sqr_subx_sub2y_core_asm(a.c, a2, a7, a14, x.c, y.c, r.c);
    return r;
#else
    return f6_sub(f6_sub(f6_sqr(a), x), f6_dbl(y));
#endif
}
// a^2 + 3x
SSA_DEV fp6 f6_sqr_add3x(const fp6 &a, const fp6 &x) {
#ifdef SSA_F6_ASM
    u64 a2[6], a7[6], a14[6];
    f6_sqr_prescale(a, a2, a7, a14);
    fp6 r;
    f6_sqr_add3x_core_asm(a.c, a2, a7, a14, x.c, r.c);
    return r;
#else
    return f6_add(f6_sqr(a), f6_add(f6_dbl(x), x));
#endif
}
// a^2 - 4x
SSA_DEV fp6 f6_sqr_sub4x(const fp6 &a, const fp6 &x) {
#ifdef SSA_F6_ASM
    u64 a2[6], a7[6], a14[6];
    f6_sqr_prescale(a, a2, a7, a14);
    fp6 r;
    f6_sqr_sub4x_core_asm(a.c, a2, a7, a14, x.c, r.c);
    return r;
#else
    return f6_sub(f6_sqr(a), f6_dbl(f6_dbl(x)));
#endif
}
// a*b - 8x
SSA_DEV fp6 f6_mul_sub8x(const fp6 &a, const fp6 &b, const fp6 &x) {
#ifdef SSA_F6_ASM
    u64 b7[6];
    f6_mul_prescale(b, b7);
    fp6 r;
    f6_mul_sub8x_core_asm(a.c, b.c, b7, x.c, r.c);
    return r;
#else
    return f6_sub(f6_mul(a, b), f6_dbl(f6_dbl(f6_dbl(x))));
#endif
}
// a*b - x
SSA_DEV fp6 f6_mul_subx(const fp6 &a, const fp6 &b, const fp6 &x) {
#ifdef SSA_F6_ASM
    u64 b7[6];
    f6_mul_prescale(b, b7);
    fp6 r;
    f6_mul_subx_core_asm(a.c, b.c, b7, x.c, r.c);
    return r;
#else
    return f6_sub(f6_mul(a, b), x);
#endif
}

// a*b - x with the prescaled second operand supplied by the caller (shared between several products by the same b)
SSA_DEV fp6 f6_mul_subx_pre(const fp6 &a, const fp6 &b, const u64 (&b7)[6], const fp6 &x) {
#ifdef SSA_F6_ASM
    fp6 r;
    f6_mul_subx_core_asm(a.c, b.c, b7, x.c, r.c);
    return r;
#else
    (void)b7;
    return f6_sub(f6_mul(a, b), x);
#endif
}
// a*b + c*d: twelve products per coefficient in one accumulator, one reduction
SSA_DEV fp6 f6_mul2_add(const fp6 &a, const fp6 &b, const fp6 &c, const fp6 &d) {
#ifdef SSA_F6_ASM
    u64 b7[6], d7[6];
    f6_mul_prescale(b, b7);
    f6_mul_prescale(d, d7);
    fp6 r;
    f6_mul2_add_core_asm(a.c, b.c, b7, c.c, d.c, d7, r.c);
    return r;
#else
    return f6_add(f6_mul(a, b), f6_mul(c, d));
#endif
}

// Frobenius x -> x^(p^k): c_i *= gamma^(i k), gamma = 7^((p-1)/6) = 2^64-2^33+2 (mod p).
// Powers of gamma: 1, g, -2^32, -1, 2^32-1 (= 2^64), 2^32.
SSA_DEV u64 fp_mul_gpow(u64 x, int e) {
    e %= 6;
    switch (e) {
        case 0: return x;
        case 1: return fp_mul(x, 0xfffffffe00000002ULL);
        case 2: return fp_neg(fp_mul(x, 0x100000000ULL));
        case 3: return fp_neg(x);
        case 4: return fp_mul(x, 0xffffffffULL);
        default: return fp_mul(x, 0x100000000ULL);
    }
}
template <int K>
SSA_DEV fp6 f6_frob(const fp6 &a) {
    fp6 r;
#pragma unroll
    for (int i = 0; i < 6; i++) r.c[i] = fp_mul_gpow(a.c[i], i * K);
    return r;
}

// norm to Fp: N(a) = a * prod_{k=1..5} frob_k(a) (only c0 of the product is non-zero); *conj receives the product
SSA_DEV u64 f6_norm(const fp6 &a, fp6 *conj = nullptr) {
    fp6 t = f6_mul(f6_frob<1>(a), f6_frob<2>(a));
    t = f6_mul(t, f6_frob<3>(a));
    t = f6_mul(t, f6_frob<4>(a));
    t = f6_mul(t, f6_frob<5>(a));
    fp_acc s;
    acc_init(s, a.c[0], t.c[0]);
#pragma unroll
    for (int i = 1; i < 6; i++) acc_mac(s, a.c[i], fp_mul_small(t.c[6 - i], 7u));
    if (conj) *conj = t;
    return acc_reduce(s);
}
// Euler's criterion in Fp: n^((p-1)/2), (p-1)/2 = 2^31 (2^32 - 1)
SSA_DEV bool fp_is_square(u64 n) {
    if (fp_is_zero(n)) return true;
    u64 x = n;                                   // n^(2^32 - 1) by the 2^k - 1 ladder
    u64 x2 = fp_mul(fp_sqr(x), x);
    u64 x4 = fp_mul(fp_sqr(fp_sqr(x2)), x2);
    u64 x8 = x4;
    for (int i = 0; i < 4; i++) x8 = fp_sqr(x8);
    x8 = fp_mul(x8, x4);
    u64 x16 = x8;
    for (int i = 0; i < 8; i++) x16 = fp_sqr(x16);
    x16 = fp_mul(x16, x8);
    u64 x32 = x16;
    for (int i = 0; i < 16; i++) x32 = fp_sqr(x32);
    x32 = fp_mul(x32, x16);
    for (int i = 0; i < 31; i++) x32 = fp_sqr(x32);
    return fp_canon(x32) == 1ull;
}
// a is a square in Fp6  <=>  its norm is a square in Fp (the norm maps Fp6* onto Fp* and squares onto squares)
SSA_DEV bool f6_is_square(const fp6 &a) { return fp_is_square(f6_norm(a)); }

// a^-1 = (prod_{k=1..5} frob_k(a)) / N(a), with N(a) = a * prod in Fp.  a != 0 required.
// The product of the five conjugates by three multiplications: b = a^p, c = b b^p = a^(p + p^2),
// d = c c^(p^2) = a^(p + .. + p^4), t = d a^(p^5).
SSA_DEV fp6 f6_inv(const fp6 &a) {
    const fp6 b = f6_frob<1>(a);
    const fp6 c = f6_mul(b, f6_frob<1>(b));
    fp6 t = f6_mul(f6_mul(c, f6_frob<2>(c)), f6_frob<5>(a));
    // only c0 of a*t is non-zero
    fp_acc s;
    acc_init(s, a.c[0], t.c[0]);
#pragma unroll
    for (int i = 1; i < 6; i++) acc_mac(s, a.c[i], fp_mul_small(t.c[6 - i], 7u));
    u64 n = acc_reduce(s);
    return f6_mul_fp(t, fp_inv(n));
}

}  // namespace ssa
