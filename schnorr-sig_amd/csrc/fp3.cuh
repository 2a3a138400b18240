// Fp3 = Fp[t]/(t^3 - 7) with t = u^2, and square roots in Fp6 = Fp3[u]/(u^2 - t) by descent
// ("complex method"): one Fp3 norm, two Fp3 square roots, no Fp6 exponentiation.  Used by point
// decompression (cheetah AffinePoint::from_compressed, called at reference src/public.rs:54-56 and
// src/batch.rs:104).  The 2-Sylow subgroup of Fp3* has order 2^32 and lies inside Fp*, so the
// Tonelli-Shanks discrete logarithm runs on plain Goldilocks elements.
#pragma once
#include "fp6.cuh"

namespace ssa {

struct fp3 {
    u64 c[3];
};

SSA_DEV fp3 f3_zero() { return fp3{{0ull, 0ull, 0ull}}; }
SSA_DEV fp3 f3_one() { return fp3{{1ull, 0ull, 0ull}}; }
SSA_DEV bool f3_is_zero(const fp3 &a) { return fp_is_zero(a.c[0]) && fp_is_zero(a.c[1]) && fp_is_zero(a.c[2]); }
SSA_DEV fp3 f3_add(const fp3 &a, const fp3 &b) {
    return fp3{{fp_add(a.c[0], b.c[0]), fp_add(a.c[1], b.c[1]), fp_add(a.c[2], b.c[2])}};
}
SSA_DEV fp3 f3_sub(const fp3 &a, const fp3 &b) {
    return fp3{{fp_sub(a.c[0], b.c[0]), fp_sub(a.c[1], b.c[1]), fp_sub(a.c[2], b.c[2])}};
}
SSA_DEV fp3 f3_neg(const fp3 &a) { return fp3{{fp_neg(a.c[0]), fp_neg(a.c[1]), fp_neg(a.c[2])}}; }
SSA_DEV fp3 f3_mul_fp(const fp3 &a, u64 s) {
    return fp3{{fp_mul(a.c[0], s), fp_mul(a.c[1], s), fp_mul(a.c[2], s)}};
}
// a * t: (a0 + a1 t + a2 t^2) t = 7 a2 + a0 t + a1 t^2
SSA_DEV fp3 f3_mul_t(const fp3 &a) { return fp3{{fp_mul_small(a.c[2], 7u), a.c[0], a.c[1]}}; }
// a / 2
SSA_DEV fp3 f3_half(const fp3 &a) {
    const u64 inv2 = 0x7fffffff80000001ULL;  // (p + 1) / 2
    return f3_mul_fp(a, inv2);
}

// schoolbook with lazy accumulation: 9 products, 3 reductions
SSA_FN fp3 f3_mul(fp3 a, fp3 b) {
    const u64 b1s = fp_mul_small(b.c[1], 7u), b2s = fp_mul_small(b.c[2], 7u);
    fp3 r;
    fp_acc s;
    acc_init(s, a.c[0], b.c[0]);
    acc_mac(s, a.c[1], b2s);
    acc_mac(s, a.c[2], b1s);
    r.c[0] = acc_reduce(s);
    acc_init(s, a.c[0], b.c[1]);
    acc_mac(s, a.c[1], b.c[0]);
    acc_mac(s, a.c[2], b2s);
    r.c[1] = acc_reduce(s);
    acc_init(s, a.c[0], b.c[2]);
    acc_mac(s, a.c[1], b.c[1]);
    acc_mac(s, a.c[2], b.c[0]);
    r.c[2] = acc_reduce(s);
    return r;
}
SSA_DEV fp3 f3_sqr(const fp3 &a) { return f3_mul(a, a); }

// a^-1 = a^p a^(p^2) / N(a);  Frobenius t -> w t with w = 7^((p-1)/3) a primitive cube root of unity
SSA_DEV fp3 f3_inv(const fp3 &a) {
    const u64 W = 0xfffffffe00000001ULL;   // gamma^2 = 7^((p-1)/3)
    const u64 W2 = 0x00000000ffffffffULL;  // gamma^4
    const fp3 a1 = fp3{{a.c[0], fp_mul(a.c[1], W), fp_mul(a.c[2], W2)}};   // a^p
    const fp3 a2 = fp3{{a.c[0], fp_mul(a.c[1], W2), fp_mul(a.c[2], W)}};   // a^(p^2)
    const fp3 m = f3_mul(a1, a2);
    const fp3 n = f3_mul(a, m);            // norm: only c0 is non-zero
    return f3_mul_fp(m, fp_inv(n.c[0]));
}

// Frobenius t -> w t (w = 7^((p-1)/3), a primitive cube root of unity): two multiplications by constants
SSA_DEV fp3 f3_frob1(const fp3 &a) {
    return fp3{{a.c[0], fp_mul(a.c[1], 0xfffffffe00000001ULL), fp_mul(a.c[2], 0x00000000ffffffffULL)}};
}
SSA_DEV fp3 f3_frob2(const fp3 &a) {
    return fp3{{a.c[0], fp_mul(a.c[1], 0x00000000ffffffffULL), fp_mul(a.c[2], 0xfffffffe00000001ULL)}};
}
template <int N>
SSA_DEV fp3 f3_sqr_n_mul(fp3 x, const fp3 &tail) {
#pragma unroll 1
    for (int i = 0; i < N; i++) x = f3_sqr(x);
    return f3_mul(x, tail);
}

// Tonelli-Shanks constants: p^3 - 1 = 2^32 * T3, zeta = t^T3 generates the 2-Sylow subgroup (in Fp)
constexpr u64 TS_ZETA_INV = 0x76b6b635b6fc8719ULL;     // zeta^-1, zeta = 0x185629dcda58878c
constexpr u64 TS_CT = 0x676669cb3be57916ULL;           // t^-((T3+1)/2) = TS_CT * t

// a^((T3-1)/2) without a 159-bit square-and-multiply: T3 = (2^32 - 1) N with N = p^2 + p + 1, hence
//   (T3 - 1) / 2 = (2^31 - 1) N + p (p + 1) / 2      and      a^N = norm(a) in Fp,
//   a^((T3-1)/2) = norm(a)^(2^31 - 1) * frob(a^((p+1)/2)),   (p + 1) / 2 = (2^32 - 1) 2^31 + 1:
// 62 squarings + 6 products in Fp3, one norm and a 31-bit power in Fp (was 158 squarings + ~100 products).
SSA_DEV fp3 f3_pow_ts(const fp3 &a) {
    // norm
    const fp3 m = f3_mul(f3_frob1(a), f3_frob2(a));
    fp_acc s;
    acc_init(s, a.c[0], m.c[0]);
    acc_mac(s, a.c[1], fp_mul_small(m.c[2], 7u));
    acc_mac(s, a.c[2], fp_mul_small(m.c[1], 7u));
    const u64 n = acc_reduce(s);
    // n^(2^31 - 1) (the ladder of fp_inv)
    u64 n2 = fp_mul(fp_sqr(n), n);
    u64 n4 = n2;
    for (int i = 0; i < 2; i++) n4 = fp_sqr(n4);
    n4 = fp_mul(n4, n2);
    u64 n8 = n4;
    for (int i = 0; i < 4; i++) n8 = fp_sqr(n8);
    n8 = fp_mul(n8, n4);
    u64 n16 = n8;
#pragma unroll 1
    for (int i = 0; i < 8; i++) n16 = fp_sqr(n16);
    n16 = fp_mul(n16, n8);
    u64 n12 = n8;
    for (int i = 0; i < 4; i++) n12 = fp_sqr(n12);
    n12 = fp_mul(n12, n4);
    u64 n14 = fp_sqr(fp_sqr(n12));
    n14 = fp_mul(n14, n2);
    const u64 n15 = fp_mul(fp_sqr(n14), n);
    u64 n31 = n16;
#pragma unroll 1
    for (int i = 0; i < 15; i++) n31 = fp_sqr(n31);
    n31 = fp_mul(n31, n15);
    // a^((p+1)/2)
    const fp3 x2 = f3_sqr_n_mul<1>(a, a);
    const fp3 x4 = f3_sqr_n_mul<2>(x2, x2);
    const fp3 x8 = f3_sqr_n_mul<4>(x4, x4);
    const fp3 x16 = f3_sqr_n_mul<8>(x8, x8);
    const fp3 x32 = f3_sqr_n_mul<16>(x16, x16);      // a^(2^32 - 1)
    const fp3 h = f3_sqr_n_mul<31>(x32, a);          // a^((2^32 - 1) 2^31 + 1)
    return f3_mul_fp(f3_frob1(h), n31);
}

// y with y^2 == a (returns true) or y^2 == a / t (returns false: a is a non-square).  a != 0.
SSA_DEV bool f3_sqrt_or_nonres(const fp3 &a, fp3 &y) {
    const fp3 w = f3_pow_ts(a);          // a^((T3-1)/2)
    const fp3 x = f3_mul(a, w);          // a^((T3+1)/2)
    const fp3 b3 = f3_mul(x, w);         // a^T3, an element of Fp of 2-power order
    u64 c = b3.c[0];
    // discrete log of c to base zeta, bit by bit: k = sum k_i 2^i with (c zeta^-k)^(2^(31-i)) == 1
    u64 zi = TS_ZETA_INV;                // zeta^(-2^i)
    u64 corr = 1ull;                     // zeta^(-floor(k/2))
    u64 zhalf = 1ull;                    // zeta^(-2^(i-1)) for i >= 1
    bool odd = false;
#pragma unroll 1
    for (int i = 0; i < 32; i++) {
        u64 d = c;
#pragma unroll 1
        for (int s = 0; s < 31 - i; s++) d = fp_sqr(d);
        const bool bit = fp_canon(d) != 1ull;
        if (bit) {
            c = fp_mul(c, zi);
            if (i == 0) odd = true;
            else corr = fp_mul(corr, zhalf);
        }
        zhalf = zi;
        zi = fp_sqr(zi);
    }
    fp3 r = x;
    if (odd) r = f3_mul_fp(f3_mul_t(x), TS_CT);  // x * t^-((T3+1)/2)
    y = f3_mul_fp(r, corr);
    return !odd;
}

// Square root in Fp6 of a = A + B u (A = (c0, c2, c4), B = (c1, c3, c5)); false for non-squares.
SSA_DEV bool f6_sqrt(const fp6 &a, fp6 &out) {
    const fp3 A = fp3{{a.c[0], a.c[2], a.c[4]}}, B = fp3{{a.c[1], a.c[3], a.c[5]}};
    fp3 x0, x1;
    if (f3_is_zero(A) && f3_is_zero(B)) {
        out = f6_zero();
        return true;
    }
    if (f3_is_zero(B)) {
        fp3 y;
        const bool sq = f3_sqrt_or_nonres(A, y);   // A = y^2, or A = t y^2 = (y u)^2
        x0 = sq ? y : f3_zero();
        x1 = sq ? f3_zero() : y;
    } else {
        const fp3 alpha = f3_sub(f3_sqr(A), f3_mul_t(f3_sqr(B)));   // norm to Fp3
        fp3 s;
        if (!f3_sqrt_or_nonres(alpha, s)) return false;
        fp3 delta = f3_half(f3_add(A, s));
        if (f3_is_zero(delta)) delta = f3_half(f3_sub(A, s));
        const fp3 hb = f3_half(B);
        fp3 y;
        if (f3_sqrt_or_nonres(delta, y)) {
            x0 = y;                                  // delta = x0^2, x1 = B / (2 x0)
            x1 = f3_mul(hb, f3_inv(y));
        } else {
            x1 = y;                                  // delta = t y^2: the other root has x0 = (B/2)/y, x1 = y
            x0 = f3_mul(hb, f3_inv(y));
        }
    }
    out.c[0] = x0.c[0]; out.c[2] = x0.c[1]; out.c[4] = x0.c[2];
    out.c[1] = x1.c[0]; out.c[3] = x1.c[1]; out.c[5] = x1.c[2];
    return f6_eq(f6_sqr(out), a);
}

// sort flag of a compressed point (bit 6): from c5 down, the first non-zero coefficient of the
// canonical y exceeds (p-1)/2 (zkcrypto-style lexicographically_largest; unpinned upstream detail)
SSA_DEV bool f6_lex_largest(const fp6 &y) {
    bool res = false, decided = false;
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        const u64 c = fp_canon(y.c[i]);
        if (!decided && c != 0ull) {
            res = c > (FP_P - 1) / 2;
            decided = true;
        }
    }
    return res;
}

}  // namespace ssa
