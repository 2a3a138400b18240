// Fp3 = Fp[t]/(t^3 - 7) with t = u^2, and square roots in Fp6 = Fp3[u]/(u^2 - t) by descent
// ("complex method"): one Fp3 norm, two Fp3 square roots, no Fp6 exponentiation.  Used by point
// decompression (cheetah AffinePoint::from_compressed, called at reference src/public.rs:54-56 and
// src/batch.rs:104).  The 2-Sylow subgroup of Fp3* has order 2^32 and lies inside Fp*, so the
// Tonelli-Shanks discrete logarithm runs on plain Goldilocks elements.
#pragma once
#include "fp6.hpp"

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

// schoolbook with lazy accumulation: 9 products, 3 reductions.  Device code: ONE generated asm block (fp6_asm.inc,
// tools/gen_f6_asm.py f3_mul_core_asm: 9 x 8 + 33 instructions; the compiled form below is ~190).
SSA_FN fp3 f3_mul(fp3 a, fp3 b) {
    const u64 b1s = fp_mul_small(b.c[1], 7u), b2s = fp_mul_small(b.c[2], 7u);
    fp3 r;
#if defined(SSA_F6_ASM) && !defined(SSA_NO_F3_ASM)
    const u64 b7[3] = {0ull, b1s, b2s};
    f3_mul_core_asm(a.c, b.c, b7, r.c);
#else
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
#endif
    return r;
}
// the square: 6 products (round 4; it was f3_mul(a, a)): c0 = a0^2 + a1 (14 a2), c1 = a0 (2 a1) + a2 (7 a2), c2 = a0 (2 a2) + a1^2
SSA_FN fp3 f3_sqr(fp3 a) {
#if defined(SSA_F6_ASM) && !defined(SSA_NO_F3_ASM)
    const u64 a7_2 = fp_mul_small(a.c[2], 7u);
    const u64 a2[3] = {0ull, fp_dbl(a.c[1]), fp_dbl(a.c[2])};
    const u64 a7[3] = {0ull, 0ull, a7_2};
    const u64 a14[3] = {0ull, 0ull, fp_dbl(a7_2)};
    fp3 r;
    f3_sqr_core_asm(a.c, a2, a7, a14, r.c);
    return r;
#else
    return f3_mul(a, a);
#endif
}

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
// zeta = 0x185629dcda58878c, zeta^-1 = 0x76b6b635b6fc8719 (tables in f3_sqrt_or_nonres)
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
    // Discrete log of c to base zeta (order 2^32) by Pohlig-Hellman with 4-bit digits, least significant first:
    // digit e_j is read off c_j^(2^(28-4j)), an element of the subgroup of order 16 (TS_G16[e] = zeta^(2^28 e)),
    // then c_{j+1} = c_j zeta^(-e_j 16^j) (TS_ZI) and the correction zeta^(-floor(k/2)) picks up its factor
    // (TS_CO).  112 squarings instead of the 496 of the bit-by-bit loop.  Tables: pow(zeta, ., p) in Python.
    static constexpr u64 TS_G16[16] = {
        0x0000000000000001ULL, 0xefffffff00000001ULL, 0xfffffffeff000001ULL, 0x000ffffffff00000ULL,
        0x0001000000000000ULL, 0x0000000000001000ULL, 0xfffffeff00000101ULL, 0xffffffef00000001ULL,
        0xffffffff00000000ULL, 0x1000000000000000ULL, 0x0000000001000000ULL, 0xffefffff00100001ULL,
        0xfffeffff00000001ULL, 0xfffffffefffff001ULL, 0x000000ffffffff00ULL, 0x0000001000000000ULL,
    };
    static constexpr u64 TS_ZI[128] = {
        0x0000000000000001ULL, 0x76b6b635b6fc8719ULL, 0x95c0ec9a7ab50701ULL, 0xa1a99678c9550900ULL,
        0xe2434909eec4f00bULL, 0x8591acb30f040081ULL, 0xd7c3a0e4a311b3e0ULL, 0xc290be950f34a87bULL,
        0xe4d14a114454645dULL, 0xae54163c414a7873ULL, 0x223bc8feb7654c30ULL, 0x139371776614b71cULL,
        0xd2d6b46a60f2151fULL, 0xb2a7c5e865c9db7fULL, 0xb5486db4d65b7474ULL, 0xfd76d3040e86a1c1ULL,
        0x0000000000000001ULL, 0x3ea7eab8d8857184ULL, 0x91f3853f38e675d9ULL, 0x8373b8d70892cbf3ULL,
        0xea52f593bb20759aULL, 0xdc1459a39f5334e0ULL, 0x03c924a686b9e39dULL, 0x5eb28021686c5010ULL,
        0xc01f93fc71bb0b9bULL, 0xf9900f6d916356a4ULL, 0xbfaca1357c2db314ULL, 0xe9d4336e9e933b16ULL,
        0xdc6fa652a5544befULL, 0x1fffc3399d868e04ULL, 0x741b01338c3c403eULL, 0x7b54462bb1eb6efcULL,
        0x0000000000000001ULL, 0x10eb845263814db7ULL, 0x4bb9aee372cf655eULL, 0x946421e5fe0dc1b7ULL,
        0xd46e5a4c36458c11ULL, 0xd35eb476e48a8c67ULL, 0x699089649f3c059aULL, 0xb6e7295b51b92476ULL,
        0xa52008ac564a2368ULL, 0xf3b36e189ba16676ULL, 0x3a18fdd17243ef21ULL, 0xd2295810b068690fULL,
        0x0ebe0b715b38443bULL, 0x2e519608003cc576ULL, 0x980783935f60ca23ULL, 0x3a0a19e0d9fd41a1ULL,
        0x0000000000000001ULL, 0xef8856969fe6ed7bULL, 0x46a23c48234c7df9ULL, 0x4b9ea14aea49c430ULL,
        0x22e1fbf03f8b95d6ULL, 0xa1bd2a25959d53bcULL, 0xc0747847c0037794ULL, 0x8a1a6f3cdf0577f2ULL,
        0xcc9e5a57b8343b3fULL, 0xcf6e62d8fd93060fULL, 0xda9b90bbb92ccf0aULL, 0x97731813d8b72e5aULL,
        0x89bad6229b157586ULL, 0x8246431ad205f082ULL, 0xb92ba6d1d00153cfULL, 0xc9f0dc453d026a88ULL,
        0x0000000000000001ULL, 0x6d341b1c9a04ed19ULL, 0x158ee068c8241329ULL, 0xa2cd245731f0a1e7ULL,
        0x409730a1895adfb6ULL, 0x9242ea239873ad37ULL, 0xe4ce2f3569f8ec06ULL, 0x13eba67512257fc4ULL,
        0x3712791d9eb0314aULL, 0xf6ddb6337dfd732cULL, 0xb9d31fdde6a95865ULL, 0x9e3dc4482624fe87ULL,
        0x24b0ab371d4c8ce8ULL, 0xa0bb333d74c5c896ULL, 0x4af72f7f55024f8eULL, 0xf8eeeea7abe8b566ULL,
        0x0000000000000001ULL, 0x9af01e431fbd6ea0ULL, 0x76a40e0866a8e50dULL, 0xc75a40a196d99d6bULL,
        0x3b9ae9d1d8d87589ULL, 0xccd995a189591249ULL, 0xa902d3354e7f6542ULL, 0x0f1aaed36ded4360ULL,
        0x3de19c67cf496a74ULL, 0xd67571f7d9bfe905ULL, 0x88faac55bfee9b74ULL, 0x1751c4ad8907625cULL,
        0xaf7ef29b0b3a11f2ULL, 0xde1bfb2b80eede7cULL, 0x9c9fb1a8cf5f0698ULL, 0x0711cdf9749c5f45ULL,
        0x0000000000000001ULL, 0x1d62e30fa4a4eeb0ULL, 0xffefffff00000011ULL, 0xba33e6ac7b4b0b7cULL,
        0xfdffffff00000001ULL, 0x80b6b6221f840fa4ULL, 0xdffffffeffffe001ULL, 0xaf0969e85a6afde5ULL,
        0xfffffffefffc0001ULL, 0x73c0f7e04540758cULL, 0x0000003fffbfffc0ULL, 0x654b2a03d212e8d0ULL,
        0x000007fffffff800ULL, 0x27757f14c17202dbULL, 0x000080007fff8000ULL, 0x585bda2e086ebc26ULL,
        0x0000000000000001ULL, 0x0000001000000000ULL, 0x000000ffffffff00ULL, 0xfffffffefffff001ULL,
        0xfffeffff00000001ULL, 0xffefffff00100001ULL, 0x0000000001000000ULL, 0x1000000000000000ULL,
        0xffffffff00000000ULL, 0xffffffef00000001ULL, 0xfffffeff00000101ULL, 0x0000000000001000ULL,
        0x0001000000000000ULL, 0x000ffffffff00000ULL, 0xfffffffeff000001ULL, 0xefffffff00000001ULL,
    };
    static constexpr u64 TS_CO[128] = {
        0x0000000000000001ULL, 0x0000000000000001ULL, 0x76b6b635b6fc8719ULL, 0x76b6b635b6fc8719ULL,
        0x95c0ec9a7ab50701ULL, 0x95c0ec9a7ab50701ULL, 0xa1a99678c9550900ULL, 0xa1a99678c9550900ULL,
        0xe2434909eec4f00bULL, 0xe2434909eec4f00bULL, 0x8591acb30f040081ULL, 0x8591acb30f040081ULL,
        0xd7c3a0e4a311b3e0ULL, 0xd7c3a0e4a311b3e0ULL, 0xc290be950f34a87bULL, 0xc290be950f34a87bULL,
        0x0000000000000001ULL, 0xe4d14a114454645dULL, 0x3ea7eab8d8857184ULL, 0xaec48e58daa821c6ULL,
        0x91f3853f38e675d9ULL, 0xdcd72bcb2a91bddaULL, 0x8373b8d70892cbf3ULL, 0x648efbcabad83c4cULL,
        0xea52f593bb20759aULL, 0x2d6e3e557c82733fULL, 0xdc1459a39f5334e0ULL, 0xb596819ad42f468bULL,
        0x03c924a686b9e39dULL, 0x2c093db44caafb62ULL, 0x5eb28021686c5010ULL, 0x3f7016a916c96cb4ULL,
        0x0000000000000001ULL, 0xc01f93fc71bb0b9bULL, 0x10eb845263814db7ULL, 0xfc65e9728faa47deULL,
        0x4bb9aee372cf655eULL, 0xeb0155a3ae434f44ULL, 0x946421e5fe0dc1b7ULL, 0x8c82f6edd49d9c68ULL,
        0xd46e5a4c36458c11ULL, 0xf90e3b961a5543a8ULL, 0xd35eb476e48a8c67ULL, 0xf7db9122e4f89799ULL,
        0x699089649f3c059aULL, 0x7dd584a38d2b54b5ULL, 0xb6e7295b51b92476ULL, 0x76e59b4705b0a3b8ULL,
        0x0000000000000001ULL, 0xa52008ac564a2368ULL, 0xef8856969fe6ed7bULL, 0x9c5ed9f97a659040ULL,
        0x46a23c48234c7df9ULL, 0x86c7191908406364ULL, 0x4b9ea14aea49c430ULL, 0xb00750c39968d455ULL,
        0x22e1fbf03f8b95d6ULL, 0x33b685f6211966ebULL, 0xa1bd2a25959d53bcULL, 0xbd9d7a68f5529351ULL,
        0xc0747847c0037794ULL, 0xea2e7668ccca821fULL, 0x8a1a6f3cdf0577f2ULL, 0x3d46e2a8095a93aaULL,
        0x0000000000000001ULL, 0xcc9e5a57b8343b3fULL, 0x6d341b1c9a04ed19ULL, 0x8a3e70a190720daaULL,
        0x158ee068c8241329ULL, 0xf16f58c03e4582abULL, 0xa2cd245731f0a1e7ULL, 0xc8fa9ce65714a9a4ULL,
        0x409730a1895adfb6ULL, 0x79eac6a9c62836c7ULL, 0x9242ea239873ad37ULL, 0x49236d6cebe73732ULL,
        0xe4ce2f3569f8ec06ULL, 0x5fa126b166462af9ULL, 0x13eba67512257fc4ULL, 0x4276284a2ce942e1ULL,
        0x0000000000000001ULL, 0x3712791d9eb0314aULL, 0x9af01e431fbd6ea0ULL, 0x3c0568652383ea48ULL,
        0x76a40e0866a8e50dULL, 0xf1f58be156abeb3bULL, 0xc75a40a196d99d6bULL, 0x0d5a00845e983e56ULL,
        0x3b9ae9d1d8d87589ULL, 0x5cb38f10c8138e40ULL, 0xccd995a189591249ULL, 0xd472858f6f255af4ULL,
        0xa902d3354e7f6542ULL, 0x19cc97ae16b205daULL, 0x0f1aaed36ded4360ULL, 0x315a53f3abbab8f3ULL,
        0x0000000000000001ULL, 0x3de19c67cf496a74ULL, 0x1d62e30fa4a4eeb0ULL, 0x98d73e94c6b9494eULL,
        0xffefffff00000011ULL, 0x705cd1e9bb1779edULL, 0xba33e6ac7b4b0b7cULL, 0x0f477dc154ea8ddfULL,
        0xfdffffff00000001ULL, 0x48616d2ad01a560eULL, 0x80b6b6221f840fa4ULL, 0x3a728d6d2abf2110ULL,
        0xdffffffeffffe001ULL, 0x5289d10bd456e898ULL, 0xaf0969e85a6afde5ULL, 0xbf562ae382c86418ULL,
        0x0000000000000001ULL, 0xfffffffefffc0001ULL, 0x0000001000000000ULL, 0xffbfffff00000001ULL,
        0x000000ffffffff00ULL, 0xfbffffff04000001ULL, 0xfffffffefffff001ULL, 0x0000000040000000ULL,
        0xfffeffff00000001ULL, 0x00000003fffffffcULL, 0xffefffff00100001ULL, 0xfffffffeffffffc1ULL,
        0x0000000001000000ULL, 0xfffffbff00000001ULL, 0x1000000000000000ULL, 0xffffbfff00004001ULL,
    };
    u64 corr = 1ull;                     // zeta^(-floor(k/2))
    bool odd = false;
#pragma unroll 1
    for (int j = 0; j < 8; j++) {
        u64 d = c;
#pragma unroll 1
        for (int sq = 0; sq < 28 - 4 * j; sq++) d = fp_sqr(d);
        d = fp_canon(d);
        int e = 0;
#pragma unroll
        for (int t = 1; t < 16; t++) e = d == TS_G16[t] ? t : e;
        c = fp_mul(c, TS_ZI[16 * j + e]);
        corr = fp_mul(corr, TS_CO[16 * j + e]);
        if (j == 0) odd = (e & 1) != 0;
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
