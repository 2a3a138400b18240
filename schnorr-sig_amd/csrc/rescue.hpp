// Rescue-Prime 64/12/8 over Goldilocks (reference: hash::rescue_64_12_8::RescueHash, imported
// at src/signature.rs:21-24 and called at :303-305) and the hash_message felt packing
// (src/signature.rs:274-306).  All constants come from the parameter blob (ssa_params): the
// upstream values are unpinned, so nothing here is hard-coded except alpha = 7.
#pragma once
#include "fp.hpp"

namespace ssa {

// Device image of the parameter blob (include/schnorr_sig_amd.h: ssa_params, 2816 bytes).
struct DevParams {
    char magic[8];
    u32 n_rounds, rate_off;
    int cap_len_idx;
    u32 pad_mode, digest_off, flags;
    u64 mds[144];
    u64 ark1[96];
    u64 ark2[96];
    u64 gen_x[6], gen_y[6];
};
static_assert(sizeof(DevParams) == 2816, "blob layout");

// x^7: 2 squarings + 2 products
SSA_DEV u64 sbox(u64 x) {
    u64 x2 = fp_sqr(x);
    u64 x4 = fp_sqr(x2);
    return fp_mul(fp_mul(x4, x2), x);
}

template <int N>
SSA_DEV u64 sqr_n_mul(u64 base, u64 tail) {
    u64 r = base;
#pragma unroll 1
    for (int i = 0; i < N; i++) r = fp_sqr(r);
    return fp_mul(r, tail);
}

// x^(1/7) = x^0x92492491b6db6db7: 63 squarings + 9 products
SSA_DEV u64 inv_sbox(u64 x) {
    u64 t1 = fp_sqr(x);
    u64 t2 = fp_sqr(t1);
    u64 t3 = sqr_n_mul<3>(t2, t2);
    u64 t4 = sqr_n_mul<6>(t3, t3);
    u64 t5 = sqr_n_mul<12>(t4, t4);
    u64 t6 = sqr_n_mul<6>(t5, t3);
    u64 t7 = sqr_n_mul<31>(t6, t6);
    u64 a = fp_mul(fp_sqr(t7), t6);
    a = fp_sqr(fp_sqr(a));
    u64 b = fp_mul(fp_mul(t1, t2), x);
    return fp_mul(a, b);
}

// Two state elements at a time: the five `square n times, multiply once` runs of the chain (58 of its 63
// squarings) go through one asm loop that squares both values with 17 instructions each instead of hipcc's 22 + 2
// s_nop (fp_chain_asm.inc, tools/gen_fp_chain_asm.py); the rest is the plain code above.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SSA_NO_FP_CHAIN_ASM)
#include "fp_chain_asm.inc"
SSA_DEV void fp_sqr2_n(u64 &x, u64 &y, int n) { fp_sqr2_n_asm(x, y, n); }
#else
SSA_DEV void fp_sqr2_n(u64 &x, u64 &y, int n) {
    for (int i = 0; i < n; i++) {
        x = fp_sqr(x);
        y = fp_sqr(y);
    }
}
#endif
SSA_DEV void inv_sbox2(u64 &x, u64 &y) {
    const u64 t1x = fp_sqr(x), t1y = fp_sqr(y);
    const u64 t2x = fp_sqr(t1x), t2y = fp_sqr(t1y);
    u64 ax = t2x, ay = t2y;
    fp_sqr2_n(ax, ay, 3);
    const u64 t3x = fp_mul(ax, t2x), t3y = fp_mul(ay, t2y);
    ax = t3x; ay = t3y;
    fp_sqr2_n(ax, ay, 6);
    const u64 t4x = fp_mul(ax, t3x), t4y = fp_mul(ay, t3y);
    ax = t4x; ay = t4y;
    fp_sqr2_n(ax, ay, 12);
    ax = fp_mul(ax, t4x); ay = fp_mul(ay, t4y);                  // t5
    fp_sqr2_n(ax, ay, 6);
    const u64 t6x = fp_mul(ax, t3x), t6y = fp_mul(ay, t3y);
    ax = t6x; ay = t6y;
    fp_sqr2_n(ax, ay, 31);
    ax = fp_mul(ax, t6x); ay = fp_mul(ay, t6y);                  // t7
    ax = fp_mul(fp_sqr(ax), t6x); ay = fp_mul(fp_sqr(ay), t6y);
    ax = fp_sqr(fp_sqr(ax)); ay = fp_sqr(fp_sqr(ay));
    const u64 bx = fp_mul(fp_mul(t1x, t2x), x), by = fp_mul(fp_mul(t1y, t2y), y);
    x = fp_mul(ax, bx);
    y = fp_mul(ay, by);
}

// The 12-felt sponge state of a lane lives in LDS ("LDS-staged"): element i of lane t is
// st[i * RS_STRIDE + t], so dynamic indexing costs a ds_read/ds_write instead of forcing the
// whole permutation to be unrolled (17k instructions when it was register-resident; the
// rolled form is ~2k and stays in the instruction cache).  ONE plane: the MDS layer loads the
// whole state into registers before it writes the first output, so it runs in place.
// 12 x 8 B = 96 B per lane, 24 KB per 256-thread block (two ping-pong planes: 48 KB, three blocks
// per CU and 14.6 ms for ssa_k_hash at 2^20; one plane: five blocks, 14.1 ms -- the kernel is
// VALU-bound, what the extra blocks buy is a shorter tail of the last block wave).
constexpr int RS_STRIDE = 256;  // == blockDim.x of every kernel that hashes
constexpr int RS_LDS_U64 = 12 * RS_STRIDE;

// dst <- MDS * src + ark (lazy 12-term accumulation per output, one reduction each).
// SMALL: every MDS entry is below 2^32 (flag set by the host when it loads the blob), which
// halves the multiplier work of the layer.
template <bool SMALL>
SSA_DEV void mds_ark(const u64 *src, u64 *dst, const u64 *__restrict__ mds, const u64 *__restrict__ ark) {
    u64 v[12];
#pragma unroll
    for (int j = 0; j < 12; j++) v[j] = src[j * RS_STRIDE];
#pragma unroll 1
    for (int i = 0; i < 12; i++) {
        const u64 *row = mds + i * 12;
        fp_acc acc;
        if (SMALL) {
            acc_zero(acc);
#pragma unroll
            for (int j = 0; j < 12; j++) acc_mac32(acc, v[j], (u32)row[j]);
        } else {
            acc_init(acc, v[0], row[0]);
#pragma unroll
            for (int j = 1; j < 12; j++) acc_mac(acc, v[j], row[j]);
        }
        dst[i * RS_STRIDE] = fp_add(acc_reduce(acc), ark[i]);
    }
}

// permutation of the state in plane A (plane B is scratch); result back in plane A
constexpr u32 PRM_FLAG_SMALL_MDS = 1u;

SSA_DEV void rescue_permutation(u64 *A, u64 *B, const DevParams *__restrict__ prm) {
    const u32 nr = prm->n_rounds;
    const bool small = (prm->flags & PRM_FLAG_SMALL_MDS) != 0;
#pragma unroll 1
    for (u32 r = 0; r < nr; r++) {
#pragma unroll 1
        for (int i = 0; i < 6; i++) {  // two independent chains per iteration
            const u64 x = sbox(A[i * RS_STRIDE]), y = sbox(A[(i + 6) * RS_STRIDE]);
            A[i * RS_STRIDE] = x;
            A[(i + 6) * RS_STRIDE] = y;
        }
        if (small) mds_ark<true>(A, B, prm->mds, prm->ark1 + 12 * r);
        else mds_ark<false>(A, B, prm->mds, prm->ark1 + 12 * r);
#pragma unroll 1
        for (int i = 0; i < 6; i++) {
            u64 x = B[i * RS_STRIDE], y = B[(i + 6) * RS_STRIDE];
            inv_sbox2(x, y);
            B[i * RS_STRIDE] = x;
            B[(i + 6) * RS_STRIDE] = y;
        }
        if (small) mds_ark<true>(B, A, prm->mds, prm->ark2 + 12 * r);
        else mds_ark<false>(B, A, prm->mds, prm->ark2 + 12 * r);
    }
}

// Rate-8 additive sponge (Hasher::hash_field) driven by a felt source `src(idx)`.
//   A, B     : this lane's columns of the two LDS planes
//   n_felts  : number of felts absorbed by this lane (13 + message felts for hash_message)
template <class Src>
SSA_DEV void sponge_hash(u64 *A, u64 *B, const DevParams *__restrict__ prm, u32 n_felts, Src src,
                         u64 (&digest)[4]) {
#pragma unroll
    for (int i = 0; i < 12; i++) A[i * RS_STRIDE] = 0ull;
    if (prm->cap_len_idx >= 0) A[prm->cap_len_idx * RS_STRIDE] = (u64)n_felts;
    const u32 rate_off = prm->rate_off;
    const bool pad1 = prm->pad_mode == 1;
    const u32 n_blocks = pad1 ? n_felts / 8 + 1 : (n_felts + 7) / 8;
#pragma unroll 1
    for (u32 b = 0; b < n_blocks; b++) {
#pragma unroll 1
        for (u32 j = 0; j < 8; j++) {
            const u32 idx = 8 * b + j;
            u64 *slot = A + (rate_off + j) * RS_STRIDE;
            if (idx < n_felts)
                *slot = fp_add(*slot, src(idx));
            else if (pad1 && idx == n_felts)
                *slot = fp_add(*slot, 1ull);
        }
        rescue_permutation(A, B, prm);
    }
    const u32 off = prm->digest_off;
#pragma unroll
    for (int k = 0; k < 4; k++) digest[k] = fp_canon(A[(off + k) * RS_STRIDE]);
}

}  // namespace ssa
