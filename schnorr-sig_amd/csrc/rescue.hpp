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

// SSA_FP_CHAINS (three) state elements at a time, the whole chain of all of them in ONE asm block (fp_chain_asm.inc,
// tools/gen_fp_chain_asm.py): 11 instructions per squaring and 13 per product instead of hipcc's 22-26 + s_nop
// padding, the values' instructions interleaved by a list scheduler so that the SGPR carries get their wait states and a
// dependent instruction sits three or more positions behind its producer; the values are pinned to the blocks' own
// registers (no moves in or out) and the chain's "copies" are register renamings.
// The blocks do not repair the one rare event of their reduction (a borrow with probability ~2^-32 per squaring, see
// the generator): they OR the lanes that met it into a wave-wide mask.  The caller tests the mask on the scalar unit
// (one compare and one branch per block, no vector instruction) and recomputes the S-boxes of a flagged lane -- about one
// lane in 3 * 10^5 hashes -- from its inputs, which are still in the LDS state, with the compiled exact chain.
// (A first form of this round re-hashed a flagged lane from scratch at the END of its hash: fewer instructions still, but
// a 2^20-signature launch has ~3 such lanes and a wave that hashes twice in the last wave generation holds the whole
// kernel back: 8.44 instead of 8.12 ms.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SSA_NO_FP_CHAIN_ASM)
#define SSA_FP_CHAIN_ASM 1
#ifdef SSA_FP_CHAIN_INC          // an alternative generated file (A/B builds: tools/build_variants.sh)
#include SSA_FP_CHAIN_INC
#else
#include "fp_chain_asm.inc"
#endif
// this lane's bit of a wave-wide mask
SSA_DEV bool lane_bit(u64 mask) {
    const u32 lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    return ((mask >> lane) & 1ull) != 0;
}
#else
#define SSA_FP_CHAINS 3
#endif
static_assert(12 % SSA_FP_CHAINS == 0 && (SSA_FP_CHAINS == 2 || SSA_FP_CHAINS == 3), "S-box blocks of 2 or 3 state elements");

// One S-box block IN PLACE on SSA_FP_CHAINS state elements (p[0], p[1], ...: the inputs are read here and, for a lane the
// asm block flags, read again).  INV: x^(1/7), else x^7.  Returns the asm block's mask (0 on the compiled path).
template <bool INV>
SSA_DEV u64 sbox_block(u64 *const (&p)[SSA_FP_CHAINS]) {
    u64 st = 0;
#ifdef SSA_FP_CHAIN_ASM
    u64 v[SSA_FP_CHAINS];
#pragma unroll
    for (int k = 0; k < SSA_FP_CHAINS; k++) v[k] = *p[k];
#if SSA_FP_CHAINS == 3
    if (INV) inv_sbox_n_asm(v[0], v[1], v[2], st);
    else sbox_n_asm(v[0], v[1], v[2], st);
#else
    if (INV) inv_sbox_n_asm(v[0], v[1], st);
    else sbox_n_asm(v[0], v[1], st);
#endif
    if (st != 0) {                 // wave-uniform: some lane of the wave met the rare borrow
        if (lane_bit(st)) {
#pragma unroll
            for (int k = 0; k < SSA_FP_CHAINS; k++) v[k] = INV ? inv_sbox(*p[k]) : sbox(*p[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < SSA_FP_CHAINS; k++) *p[k] = v[k];
#else
#pragma unroll
    for (int k = 0; k < SSA_FP_CHAINS; k++) *p[k] = INV ? inv_sbox(*p[k]) : sbox(*p[k]);
#endif
    return st;
}
// ONE value: the cooperative kernels' sponge (a state element per lane of one wave: latency, not throughput).  The single-chain
// programs pad the wait states of their carries with s_nop: 811 + 288 issue slots for x^(1/7) where the compiled chain has
// ~1800; a lane the block flags is recomputed from its input.
template <bool INV>
SSA_DEV u64 sbox_one(u64 x) {
#ifdef SSA_FP_CHAIN_ASM
    u64 v = x, st = 0;
    if (INV) inv_sbox_1_asm(v, st);
    else sbox_1_asm(v, st);
    if (st != 0) {
        if (lane_bit(st)) v = INV ? inv_sbox(x) : sbox(x);
    }
    return v;
#else
    return INV ? inv_sbox(x) : sbox(x);
#endif
}

// the S-box layer of the 12-element state at `st` (element i at st[i * RS_STRIDE]): blocks (i, i + n, i + 2n), n = 12 / chains
template <bool INV>
SSA_DEV void sbox_layer(u64 *state, int stride) {
    constexpr int NB = 12 / SSA_FP_CHAINS;
#pragma unroll 1
    for (int i = 0; i < NB; i++) {
        u64 *p[SSA_FP_CHAINS];
#pragma unroll
        for (int k = 0; k < SSA_FP_CHAINS; k++) p[k] = state + (i + k * NB) * stride;
        sbox_block<INV>(p);
    }
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

// The same layer for an MDS matrix whose entries are below 2^16 (PRM_FLAG_TINY_MDS; upstream-style circulants of
// one- and two-digit integers): the low and the high words of the state are accumulated separately,
//   lo = ark_lo + sum v_j.lo * m_ij < 2^52,  hi = ark_hi + sum v_j.hi * m_ij < 2^52,  row = lo + 2^32 hi,
// 24 multiply-adds chained through their 64-bit addend with NO carry bookkeeping (the general path pays a carry add
// per multiply and a three-column fold), the round constant rides in the two opening addends, and the reduction is
// one add, one multiply-add by EPS for bits 64.. and one wrap-around fix-up.
SSA_DEV void mds_ark_tiny(const u64 *src, u64 *dst, const u64 *__restrict__ mds, const u64 *__restrict__ ark) {
    u64 v[12];
#pragma unroll
    for (int j = 0; j < 12; j++) v[j] = src[j * RS_STRIDE];
#ifdef SSA_MDS_FREE      // TIMING EXPERIMENT ONLY (wrong digests): the layer for free -- one add per row, the LDS round trip
    (void)mds;            // kept -- bounds what ANY other evaluation of the matrix product (MFMA, dot4) could save
#pragma unroll
    for (int i = 0; i < 12; i++) dst[i * RS_STRIDE] = v[i] + ark[i];
    return;
#endif
#pragma unroll 1
    for (int i = 0; i < 12; i++) {
        const u64 *row = mds + i * 12;
        const u64 k = ark[i];
        u64 lo = (u64)lo32(k), hi = (u64)hi32(k);
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const u32 m = (u32)row[j];
            lo += (u64)lo32(v[j]) * m;
            hi += (u64)hi32(v[j]) * m;
        }
        // lo + 2^32 hi = (lo + 2^32 hi_l) + 2^64 hi_h;  2^64 = EPS
        const u64 t = lo + ((u64)lo32(hi) << 32);
        const u32 top = hi32(hi) + (t < lo ? 1u : 0u);        // < 2^21
        dst[i * RS_STRIDE] = fp_reduce_parts(t, top, 0ull);
    }
}

// permutation of the state in plane A (plane B is scratch); result back in plane A
constexpr u32 PRM_FLAG_SMALL_MDS = 1u, PRM_FLAG_TINY_MDS = 2u;

SSA_DEV void rescue_permutation(u64 *A, u64 *B, const DevParams *__restrict__ prm) {
    const u32 nr = prm->n_rounds;
    const bool small = (prm->flags & PRM_FLAG_SMALL_MDS) != 0;
    const bool tiny = (prm->flags & PRM_FLAG_TINY_MDS) != 0;
#pragma unroll 1
    for (u32 r = 0; r < nr; r++) {
        sbox_layer<false>(A, RS_STRIDE);
        if (tiny) mds_ark_tiny(A, B, prm->mds, prm->ark1 + 12 * r);
        else if (small) mds_ark<true>(A, B, prm->mds, prm->ark1 + 12 * r);
        else mds_ark<false>(A, B, prm->mds, prm->ark1 + 12 * r);
        sbox_layer<true>(B, RS_STRIDE);
        if (tiny) mds_ark_tiny(B, A, prm->mds, prm->ark2 + 12 * r);
        else if (small) mds_ark<true>(B, A, prm->mds, prm->ark2 + 12 * r);
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
