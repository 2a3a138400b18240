// Goldilocks field Fp, p = 2^64 - 2^32 + 1 (reference README.md:4), for gfx950 VALU.
//
// Representation: "loose" u64 -- any value in [0, 2^64) stands for its residue mod p.
// Values are canonicalised (fp_canon) only where bits are compared or leave the GPU.
// Identities used everywhere:  2^64 = EPS = 2^32 - 1,  2^96 = -1  (mod p).
//
// The hot primitive is the 32x32+64 multiply-add v_mad_u64_u32 (4 per 64x64 product).
// Fp6 products accumulate partial products lazily in three 96-bit columns (fp_acc) so
// that one Goldilocks reduction is paid per output coefficient, not per product.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ssa {

typedef uint64_t u64;
typedef unsigned int u32;

// __host__ too: tests/csrc/host_arith.cpp compiles these headers for the CPU to unit-test the
// limb logic without a GPU; the library itself never runs them on the host.
#define SSA_DEV __host__ __device__ __forceinline__
// out-of-line: the Fp6 product/square bodies are ~4 KB each; one copy keeps the scalar-
// multiplication loop inside the 64 KB instruction cache.
#define SSA_FN inline __host__ __device__ __attribute__((noinline))

constexpr u64 FP_P = 0xffffffff00000001ULL;
constexpr u32 FP_EPS = 0xffffffffu;

SSA_DEV u32 lo32(u64 x) { return (u32)x; }
SSA_DEV u32 hi32(u64 x) { return (u32)(x >> 32); }
SSA_DEV u64 mk64(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }

// a + b (mod p), loose in / loose out.  A carry out of 2^64 is worth EPS; the corrected
// sum can wrap once more only when both inputs are above p.
// (Measured: hand-written v_add_co/v_addc_co/v_subb VCC carry chains for fp_add, fp_sub and the
// reductions have fewer instructions than what hipcc emits from this C++ -- v_lshl_add_u64 +
// v_cmp_lt_u64 + v_cndmask -- but ran 7-16 % slower: dependent VCC hops serialise.)
SSA_DEV u64 fp_add(u64 a, u64 b) {
    u64 s = a + b;
    u32 c = s < a;
    u64 t = s + (c ? (u64)FP_EPS : 0ull);
    u32 c2 = t < s;
    return t + (c2 ? (u64)FP_EPS : 0ull);
}

// a - b (mod p), loose in / loose out.
SSA_DEV u64 fp_sub(u64 a, u64 b) {
    u64 d = a - b;
    u32 bw = a < b;
    u64 t = d - (bw ? (u64)FP_EPS : 0ull);
    u32 b2 = t > d;
    return t - (b2 ? (u64)FP_EPS : 0ull);
}

SSA_DEV u64 fp_dbl(u64 a) { return fp_add(a, a); }

// canonical representative in [0, p)
SSA_DEV u64 fp_canon(u64 a) { return a >= FP_P ? a - FP_P : a; }
SSA_DEV u64 fp_neg(u64 a) { return fp_sub(0ull, a); }
SSA_DEV bool fp_is_zero(u64 a) { return a == 0ull || a == FP_P; }
SSA_DEV bool fp_eq(u64 a, u64 b) { return fp_canon(a) == fp_canon(b); }

// (2^32 - 1) * h as a 64-bit value.  hipcc emits one v_mad_u64_u32 by 0xffffffff for this; a
// hand-written v_sub_co/v_subbrev pair measured 8 % slower in the Fp-mul probe (VCC dependency), and the
// carry-free form (0 - h, h - min(h, 1)) 13 % slower (ssa_k_hash 14.6 -> 16.5 ms).
SSA_DEV u64 eps_times(u32 h) { return ((u64)h << 32) - h; }

// Reduce lo + 2^64*(h0 + 2^32*h1), h1 given as a 64-bit "top" value (top < 2^63):
//   V = lo + EPS*h0 - top.
// One multiply-add computes lo + EPS*h0 (the 64-bit addend is lo; a carry c is worth +2^64 = +EPS), the
// subtraction of top may borrow (b, worth -EPS), and ONE correction by (c - b)*EPS finishes:
//   c only:  r <= 2^64 - 2^33, adding EPS cannot carry again;   b only:  r >= 2^64 - top > EPS, no second borrow;
//   both: they cancel.  (Round 1 fixed the borrow, added EPS*h0 separately and fixed the carry: three 64-bit
//   additions and two compares more -- the same restructuring as in the generated asm blocks.)
SSA_DEV u64 fp_reduce_parts(u64 lo, u32 h0, u64 top) {
    const u64 r = (u64)h0 * (u64)FP_EPS + lo;    // v_mad_u64_u32 with a 64-bit addend
    const bool c = r < lo;
    const u64 d = r - top;
    const bool b = r < top;
    u64 corr = 0ull;
    if (c && !b) corr = (u64)FP_EPS;
    if (b && !c) corr = 0ull - (u64)FP_EPS;      // = p (mod 2^64)
    return d + corr;
}

SSA_DEV u64 fp_reduce128(u64 lo, u64 hi) { return fp_reduce_parts(lo, lo32(hi), (u64)hi32(hi)); }

// full 64x64 -> 128 product out of four v_mad_u64_u32
SSA_DEV void mul64x64(u64 a, u64 b, u64 &lo, u64 &hi) {
    u32 a0 = lo32(a), a1 = hi32(a), b0 = lo32(b), b1 = hi32(b);
    u64 t0 = (u64)a0 * b0;
    u64 t1 = (u64)a0 * b1 + hi32(t0);
    u64 t2 = (u64)a1 * b0 + lo32(t1);
    u64 t3 = (u64)a1 * b1 + ((u64)hi32(t1) + hi32(t2));
    lo = mk64(lo32(t0), lo32(t2));
    hi = t3;
}

SSA_DEV u64 fp_mul(u64 a, u64 b) {
    u64 lo, hi;
    mul64x64(a, b, lo, hi);
    return fp_reduce128(lo, hi);
}
// a^2 with three multiplies instead of four: a = a0 + 2^32 a1, a^2 = a0^2 + 2^33 a0 a1 + 2^64 a1^2.
//   u  = a0*a1 + (a0^2 >> 33)            (no overflow: (2^32-1)^2 + 2^31 < 2^64)
//   lo = (a0^2 mod 2^33) | (u mod 2^31) << 33     (disjoint bits: no carry)
//   hi = a1^2 + (u >> 31)                (no overflow)
// Measured on MI355X (DESIGN.md, round 2 experiments): v_mad_u64_u32 issues about as fast as the
// shift/merge instructions this trades it for, so the saving is small; kept because it is never slower.
SSA_DEV u64 fp_sqr3(u64 a) {
    const u32 a0 = lo32(a), a1 = hi32(a);
    const u64 t0 = (u64)a0 * a0;
    const u64 u = (u64)a0 * a1 + (t0 >> 33);
    const u64 lo = (t0 & 0x1ffffffffULL) | (u << 33);
    const u64 hi = (u64)a1 * a1 + (u >> 31);
    return fp_reduce128(lo, hi);
}
#ifdef SSA_FP_SQR4
SSA_DEV u64 fp_sqr(u64 a) { return fp_mul(a, a); }
#else
SSA_DEV u64 fp_sqr(u64 a) { return fp_sqr3(a); }
#endif

// a * k for a 32-bit constant k: two mads, 96-bit result
SSA_DEV u64 fp_mul_small(u64 a, u32 k) {
    u64 t0 = (u64)lo32(a) * k;
    u64 t1 = (u64)hi32(a) * k + hi32(t0);
    return fp_reduce_parts(mk64(lo32(t0), lo32(t1)), hi32(t1), 0ull);
}

// ---------------------------------------------------------------------------------------
// Lazy accumulator: sum of up to 15 full 64x64 products kept as three columns
//   c0 (weight 2^0)  <- a0*b0      c1 (weight 2^32) <- a0*b1 + a1*b0      c2 (weight 2^64) <- a1*b1
// each a 64-bit mad accumulator plus a carry counter (v_mad_u64_u32 carry-out -> v_addc).
// ---------------------------------------------------------------------------------------
struct fp_acc {
    u64 c0, c1, c2;
    u32 k0, k1, k2;
};

SSA_DEV void acc_zero(fp_acc &s) {
    s.c0 = s.c1 = s.c2 = 0ull;
    s.k0 = s.k1 = s.k2 = 0u;
}

// s += a*b.  One asm block per 64x64 product: four v_mad_u64_u32 whose carry-outs feed
// (measured on MI355X, lazy Fp6 probe: this block 2.82e12 products/s; the four mads issued first and
// the carry adds last 2.82e12; plain C++ carry detection 1.77e12; NO carry tracking -- wrong results,
// upper bound -- 3.78e12: the carries are the price of 64-bit columns)
// v_addc_co_u32 on the column counters; the independent mads are placed between each
// carry's producer and consumer, and two scratch SGPR pairs keep three carries in flight.
SSA_DEV void acc_mac(fp_acc &s, u64 a, u64 b) {
    u32 a0 = lo32(a), a1 = hi32(a), b0 = lo32(b), b1 = hi32(b);
#if defined(__HIP_DEVICE_COMPILE__)
    u64 t0, t1;
    asm("v_mad_u64_u32 %0, %6, %8, %10, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %8, %11, %1\n\t"
        "v_mad_u64_u32 %2, %7, %9, %11, %2\n\t"
        "v_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"
        "v_mad_u64_u32 %1, vcc, %9, %10, %1\n\t"
        "v_addc_co_u32 %3, %6, 0, %3, %6\n\t"
        "v_addc_co_u32 %5, %7, 0, %5, %7\n\t"
        "v_addc_co_u32 %4, vcc, 0, %4, vcc"
        : "+v"(s.c0), "+v"(s.c1), "+v"(s.c2), "+v"(s.k0), "+v"(s.k1), "+v"(s.k2), "=&s"(t0), "=&s"(t1)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1)
        : "vcc");
#else
    u64 p, t;
    p = (u64)a0 * b0; t = s.c0 + p; s.k0 += t < p; s.c0 = t;
    p = (u64)a0 * b1; t = s.c1 + p; s.k1 += t < p; s.c1 = t;
    p = (u64)a1 * b0; t = s.c1 + p; s.k1 += t < p; s.c1 = t;
    p = (u64)a1 * b1; t = s.c2 + p; s.k2 += t < p; s.c2 = t;
#endif
}

// s += a*m for a 32-bit multiplier (small MDS entries): two mads instead of four
SSA_DEV void acc_mac32(fp_acc &s, u64 a, u32 m) {
    u32 a0 = lo32(a), a1 = hi32(a);
#if defined(__HIP_DEVICE_COMPILE__)
    u64 t0;
    asm("v_mad_u64_u32 %0, %4, %5, %7, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %6, %7, %1\n\t"
        "v_addc_co_u32 %2, %4, 0, %2, %4\n\t"
        "v_addc_co_u32 %3, vcc, 0, %3, vcc"
        : "+v"(s.c0), "+v"(s.c1), "+v"(s.k0), "+v"(s.k1), "=&s"(t0)
        : "v"(a0), "v"(a1), "v"(m)
        : "vcc");
#else
    u64 p, t;
    p = (u64)a0 * m; t = s.c0 + p; s.k0 += t < p; s.c0 = t;
    p = (u64)a1 * m; t = s.c1 + p; s.k1 += t < p; s.c1 = t;
#endif
}

// first product into a fresh accumulator: no carries possible on c0/c2, one on c1
SSA_DEV void acc_init(fp_acc &s, u64 a, u64 b) {
    u32 a0 = lo32(a), a1 = hi32(a), b0 = lo32(b), b1 = hi32(b);
    s.c0 = (u64)a0 * b0;
    s.c2 = (u64)a1 * b1;
    const u64 p = (u64)a0 * b1, q = (u64)a1 * b0 + p;
    s.c1 = q;
    s.k0 = s.k2 = 0u;
    s.k1 = q < p;
}

// value = c0 + 2^32 c1 + 2^64 c2 with c_i = col_i + 2^64 k_i, reduced mod p (loose).
SSA_DEV u64 acc_reduce(const fp_acc &s) {
    // five 32-bit words w0..w4 by carry propagation
    u32 w0 = lo32(s.c0);
    u64 t1 = (u64)hi32(s.c0) + lo32(s.c1);
    u32 w1 = lo32(t1);
    u64 t2 = (u64)hi32(s.c1) + lo32(s.c2) + s.k0 + hi32(t1);
    u32 w2 = lo32(t2);
    u64 t3 = (u64)hi32(s.c2) + s.k1 + hi32(t2);
    u32 w3 = lo32(t3);
    u32 w4 = s.k2 + hi32(t3);
    // w0 + 2^32 w1 + 2^64 w2 + 2^96 w3 + 2^128 w4 = (w1:w0) + EPS*w2 - (w4:w3)
    return fp_reduce_parts(mk64(w0, w1), w2, mk64(w3, w4));
}

// x^(p-2) by a fixed chain: p - 2 = 2^64 - 2^32 - 1 = (2^31 - 1) 2^33 + (2^32 - 1): 64 squarings + 10 products
// (round 1's chain built 2^32 - 1 and 2^31 - 1 separately: 86 + 12)
SSA_DEV u64 fp_sqr_times(u64 x, int n) {
#pragma unroll 1
    for (int i = 0; i < n; i++) x = fp_sqr(x);
    return x;
}
SSA_DEV u64 fp_inv(u64 x) {
    const u64 x2 = fp_mul(fp_sqr(x), x);                    // x^(2^2 - 1)
    const u64 x4 = fp_mul(fp_sqr_times(x2, 2), x2);         // 2^4 - 1
    const u64 x8 = fp_mul(fp_sqr_times(x4, 4), x4);         // 2^8 - 1
    const u64 x16 = fp_mul(fp_sqr_times(x8, 8), x8);        // 2^16 - 1
    const u64 x24 = fp_mul(fp_sqr_times(x16, 8), x8);       // 2^24 - 1
    const u64 x28 = fp_mul(fp_sqr_times(x24, 4), x4);       // 2^28 - 1
    const u64 x30 = fp_mul(fp_sqr_times(x28, 2), x2);       // 2^30 - 1
    const u64 x31 = fp_mul(fp_sqr(x30), x);                 // 2^31 - 1
    const u64 x32 = fp_mul(fp_sqr(x31), x);                 // 2^32 - 1
    return fp_mul(fp_sqr_times(x31, 33), x32);
}

}  // namespace ssa
