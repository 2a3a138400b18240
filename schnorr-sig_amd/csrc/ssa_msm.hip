// MSM-form batch verification on the GPU: the algorithm the reference's verify_batch actually runs
// (src/batch.rs:31-130).
//
//     sum_i s_i R_i  -  sum_i (s_i h_i) P_i   ?=   [ sum_i s_i e_i ] G        (x-only comparison)
//
// with R_i decompressed from sig.x (src/batch.rs:104), h_i = hash_message scalars (:64-73), s_i the
// random coefficients (:75-78).  The 2n-point multi-scalar multiplication is a bucket method laid out
// for the GPU:
//   1. msm_k_prepare   per signature: decompress R, form the points R_i, -P_i and the scalars
//                      a_i = s_i, b_i = s_i h_i mod q, and block-reduce s_i e_i mod q
//   2. msm_k_digits    (window, digit) sort keys for every point; hipCUB radix sort groups the point
//                      indices of each bucket; msm_k_bounds finds the bucket extents
//   3. msm_k_buckets   ONE BUCKET PER LANE: a lane adds up the points of its bucket with mixed
//                      additions (every exceptional case handled: equal public keys land in one bucket)
//   4. msm_k_chunks    running-sum trick on chunks of 8 buckets (short chains, 2^17 lanes);
//                      msm_k_tree (pairwise, x13) sums the chunk sums of a window; msm_k_finish is ONE cooperative
//                      block: wave 0 combines the windows by Horner's rule (the only long sequential chain
//                      of the method, ~240 doublings, wave-cooperative Fp6 arithmetic), wave 1 computes
//                      [lin]G from the comb table meanwhile; then the x coordinates are compared
// Panics of the reference (undecodable x, x not on the curve: src/batch.rs:67,104) give SSA_MALFORMED.
#define SSA_NO_KERNELS 1
#include "ssa_ctx.hpp"

#include <hipcub/hipcub.hpp>

#include <sys/random.h>

namespace ssa {

constexpr int MSM_CHUNK = 8;   // buckets per lane in the running-sum pass (short chains, many lanes)
constexpr u32 MSM_TREE_GROUP = 2;   // pairwise: the tree is latency-bound (lone waves), 13 levels of ONE addition beat 5 of eight

struct MsmShape {
    u32 c;        // window bits
    u32 windows;  // ceil(255 / c)
    u32 buckets;  // 2^c
    u32 chunks;   // per window
};

SSA_DEV u32 sc_window(const u64 *__restrict__ k, u32 bit, u32 c) {
    const u32 wi = bit >> 6, sh = bit & 63u;
    u64 v = k[wi] >> sh;
    if (sh + c > 64 && wi < 3) v |= k[wi + 1] << (64 - sh);
    return (u32)(v & ((1ull << c) - 1ull));
}

SSA_DEV void st_aff_row(u64 *__restrict__ row, const aff &p) {
    st_f6(row, p.x);
    st_f6(row + 6, p.y);
}

// ---- 1. points and scalars ---------------------------------------------------------------------
__global__ void __launch_bounds__(256)
msm_k_prepare(const u8 *__restrict__ sigs, const u8 *__restrict__ pks, const u8 *__restrict__ pk_inf,
              const u64 *__restrict__ h_in,
              const u8 *__restrict__ coeffs, u32 coeff_bytes, size_t n, u64 *__restrict__ points,
              u64 *__restrict__ scalars, u64 *__restrict__ partials, u32 *__restrict__ malformed) {
    __shared__ u64 red[256 * 4];
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    sc256 se;
#pragma unroll
    for (int k = 0; k < 4; k++) se.w[k] = 0;
    if (i < n) {
        bool ok = true;
        aff P;
        P.x = ld_fp6(pks + 96 * i, ok);
        P.y = ld_fp6(pks + 96 * i + 48, ok);
        const sc256 e = ld_sc(sigs + 81 * i + 49);
        ok = ok && !sc_geq_q(e);
        // an identity key is a valid PublicKey (src/public.rs:95-101); negated and fed to the MSM it adds nothing
        // (src/batch.rs:106): the (0, 0) sentinel jac_madd skips
        const bool p_inf = pk_inf && pk_inf[i];
        if (ok && !p_inf) ok = aff_on_curve(P);
        aff R;
        bool r_inf = false;
        if (ok) ok = decompress_lane(sigs + 81 * i, R, r_inf) == 0;   // from_compressed(..).unwrap(), :104
        if (!ok) {
            atomicOr(malformed, 1u);
            R.x = f6_zero(); R.y = f6_zero();
            P.x = f6_zero(); P.y = f6_zero();
        }
        if (r_inf) {  // identity R: the (0, 0) sentinel jac_madd skips
            R.x = f6_zero();
            R.y = f6_zero();
        }
        if (p_inf) {
            P.x = f6_zero();
            P.y = f6_zero();
        }
        sc256 s;
#pragma unroll
        for (int k = 0; k < 4; k++) s.w[k] = 0;
        const u8 *cp = coeffs + (size_t)coeff_bytes * i;
        for (u32 b = 0; b < coeff_bytes; b++) s.w[b >> 3] |= (u64)cp[b] << (8 * (b & 7u));
        s = sc_reduce256(s);                                           // Scalar::random, :75-78
        sc256 h;
#pragma unroll
        for (int k = 0; k < 4; k++) h.w[k] = h_in[4 * i + k];
        const sc256 sh = sc_mul_mod(s, h);                             // hashes[i] *= scalars[i], :109-111
        if (ok) se = sc_mul_mod(s, e);                                 // s * e, :92-97
        P.y = f6_canon(f6_neg(P.y));                                   // k.0.neg(), :106
        st_aff_row(points + 12 * i, R);
        st_aff_row(points + 12 * (n + i), P);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            scalars[4 * i + k] = s.w[k];
            scalars[4 * (n + i) + k] = sh.w[k];
        }
    }
    // block reduction of s_i e_i mod q
#pragma unroll
    for (int k = 0; k < 4; k++) red[threadIdx.x * 4 + k] = se.w[k];
    __syncthreads();
    for (u32 stride = 128; stride > 0; stride >>= 1) {
        if (threadIdx.x < stride) {
            sc256 a, b;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                a.w[k] = red[threadIdx.x * 4 + k];
                b.w[k] = red[(threadIdx.x + stride) * 4 + k];
            }
            a = sc_add_mod(a, b);
#pragma unroll
            for (int k = 0; k < 4; k++) red[threadIdx.x * 4 + k] = a.w[k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 4; k++) partials[4 * blockIdx.x + k] = red[k];
    }
}

// ---- 2. sort keys ------------------------------------------------------------------------------
// The first n points (the R_i) carry the coefficients s_i themselves: with coefficients of `coeff_bytes` bytes only
// their lowest wa windows can be non-zero, and the sort is spared the rest (a quarter of the items for the 128-bit
// coefficients the library draws); the other n points (the P_i) carry s_i h_i mod q and occupy every window.
__global__ void msm_k_digits(const u64 *__restrict__ scalars, size_t n, u32 wa, MsmShape sh,
                             u32 *__restrict__ keys, u32 *__restrict__ vals) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t first = n * wa;
    if (t >= first + n * sh.windows) return;
    u32 j;
    size_t i;
    if (t < first) {
        j = (u32)(t / n);
        i = t - (size_t)j * n;
    } else {
        const size_t u = t - first;
        j = (u32)(u / n);
        i = n + (u - (size_t)j * n);
    }
    const u32 d = sc_window(scalars + 4 * i, j * sh.c, sh.c);
    keys[t] = j * sh.buckets + d;
    vals[t] = (u32)i;
}

// bounds[2*key] = first position, bounds[2*key+1] = one past the last (both 0 for empty buckets)
__global__ void msm_k_bounds(const u32 *__restrict__ keys, size_t total, u32 *__restrict__ bounds) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const u32 k = keys[t];
    if (t == 0 || keys[t - 1] != k) bounds[2 * (size_t)k] = (u32)t;
    if (t + 1 == total || keys[t + 1] != k) bounds[2 * (size_t)k + 1] = (u32)(t + 1);
}

// ---- 3. one bucket per lane ---------------------------------------------------------------------
// Bucket sizes are Poisson-distributed (mean 16 or 32 at n = 2^20) and a wave waits for its largest bucket:
// the buckets are handed to the lanes in order of size (a 1 M-item radix sort, ~0.2 ms), so that the 64
// buckets of a wave hold the same number of points.
__global__ void msm_k_counts(const u32 *__restrict__ bounds, MsmShape sh, u32 *__restrict__ cnt,
                             u32 *__restrict__ ids) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)sh.windows * sh.buckets) return;
    cnt[t] = (t & (sh.buckets - 1)) != 0 ? bounds[2 * t + 1] - bounds[2 * t] : 0u;   // digit 0 contributes nothing
    ids[t] = (u32)t;
}

__global__ void __launch_bounds__(256, 2)
msm_k_buckets(const u64 *__restrict__ points, const u32 *__restrict__ vals, const u32 *__restrict__ bounds,
              const u32 *__restrict__ order, MsmShape sh, u64 *__restrict__ bsum) {
    const size_t lane_id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nb = (size_t)sh.windows * sh.buckets;
    if (lane_id >= nb) return;
    const size_t t = order[lane_id];
    jac acc = jac_identity();
    if ((t & (sh.buckets - 1)) != 0) {
        const u32 lo = bounds[2 * t], hi = bounds[2 * t + 1];
#pragma unroll 1
        for (u32 p = lo; p < hi; p++) {
            const aff q = ld_aff(points + 12 * (size_t)vals[p]);
            acc = jac_madd_fast(acc, q);      // asm block; identity / equal points fall back to the exact addition
        }
    }
    st_jac(bsum + 18 * t, acc);
}

// ---- 4. bucket reduction ------------------------------------------------------------------------
// chunk of MSM_CHUNK buckets [v0, v0 + L): sum_v v B_v = sum_v (v - v0 + 1) B_v + (v0 - 1) sum_v B_v
__global__ void __launch_bounds__(256, 2)
msm_k_chunks(const u64 *__restrict__ bsum, MsmShape sh, u64 *__restrict__ chunk_out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)sh.windows * sh.chunks) return;
    const u32 j = (u32)(t / sh.chunks), ch = (u32)(t % sh.chunks);
    const u32 len = sh.buckets < (u32)MSM_CHUNK ? sh.buckets : (u32)MSM_CHUNK;
    const u32 v0 = ch * len;
    // the top bucket opens both sums (two additions to the identity saved per chunk)
    jac running = ld_jac(bsum + 18 * ((size_t)j * sh.buckets + (v0 + len - 1))), total = running;
#pragma unroll 1
    for (int v = (int)(v0 + len) - 2; v >= (int)v0; v--) {
        if (v == 0) break;
        const jac b = ld_jac(bsum + 18 * ((size_t)j * sh.buckets + (u32)v));
        running = jac_add(running, b);
        total = jac_add(total, running);
    }
    // + [v0 - 1] running (v0 >= 1 here unless this is the first chunk, where the weight offset is 0): double-and-add from
    // the top set bit of the weight, the doublings through the ladder's generated statement
    if (v0 > 1) {
        const u32 m = v0 - 1;
        jac acc = running;
#pragma unroll 1
        for (int bit = 30 - __builtin_clz(m); bit >= 0; bit--) {
            acc = jac_dbl_n(acc, 1u);
            if ((m >> bit) & 1u) acc = jac_add(acc, running);
        }
        total = jac_add(total, acc);
    }
    st_jac(chunk_out + 18 * t, total);
}

// tree step: out[j][g] = sum of `group` consecutive points of window j's `count` inputs
__global__ void __launch_bounds__(64, 2)
msm_k_tree(const u64 *__restrict__ in, u32 windows, u32 count, u32 group, u64 *__restrict__ out) {
    const u32 groups = (count + group - 1) / group;
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= windows * groups) return;
    const u32 j = t / groups, g = t % groups;
    const u32 lo = g * group, hi = (lo + group < count) ? lo + group : count;
    jac acc = ld_jac(in + 18 * ((size_t)j * count + lo));        // (lo < hi always: groups = ceil(count / group))
#pragma unroll 1
    for (u32 k = lo + 1; k < hi; k++) acc = jac_add(acc, ld_jac(in + 18 * ((size_t)j * count + k)));
    st_jac(out + 18 * (size_t)t, acc);
}

// ---- coefficients ---------------------------------------------------------------------------------
// Scalar::random(rng) (src/batch.rs:75-78) when the caller supplies none: 128-bit coefficients from a
// ChaCha20 keystream (RFC 8439 block function, 32-bit block counter) keyed per call with 44 bytes of
// getrandom(2).  One 64-byte block = four coefficients per lane; drawing 16 MB on the host took ~20 ms.
struct ChaChaKey {
    u32 key[8];
    u32 nonce[3];
};
SSA_DEV u32 rotl32(u32 x, int n) { return (x << n) | (x >> (32 - n)); }
#define SSA_QR(a, b, c, d)          \
    a += b; d ^= a; d = rotl32(d, 16); \
    c += d; b ^= c; b = rotl32(b, 12); \
    a += b; d ^= a; d = rotl32(d, 8);  \
    c += d; b ^= c; b = rotl32(b, 7)
__global__ void __launch_bounds__(256)
msm_k_chacha20(ChaChaKey kn, u32 counter0, size_t n_blocks, u32 *__restrict__ out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_blocks) return;
    u32 st[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, kn.key[0], kn.key[1], kn.key[2], kn.key[3],
                  kn.key[4], kn.key[5], kn.key[6], kn.key[7], counter0 + (u32)t, kn.nonce[0], kn.nonce[1], kn.nonce[2]};
    u32 x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = st[i];
#pragma unroll 1
    for (int r = 0; r < 10; r++) {
        SSA_QR(x[0], x[4], x[8], x[12]);
        SSA_QR(x[1], x[5], x[9], x[13]);
        SSA_QR(x[2], x[6], x[10], x[14]);
        SSA_QR(x[3], x[7], x[11], x[15]);
        SSA_QR(x[0], x[5], x[10], x[15]);
        SSA_QR(x[1], x[6], x[11], x[12]);
        SSA_QR(x[2], x[7], x[8], x[13]);
        SSA_QR(x[3], x[4], x[9], x[14]);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) out[16 * t + i] = x[i] + st[i];   // little-endian words = the keystream bytes
}

// ---- the one sequential chain of the reduction ---------------------------------------------------
// left = sum_j 2^(c j) W_j by Horner's rule: (windows - 1) x (c doublings + one addition), a chain of ~240
// dependent doublings.  A lone lane ran it at ~80 us per doubling; here wave 0 of the block works on the
// one point (wave-cooperative Fp6 arithmetic, ssa_coop.hpp: ~2.5 us per doubling) while wave 1 adds up
// lin = sum of the blocks' partial sums and computes right = [lin] G from the comb table.
// Verdict: x-only comparison, left.get_x() == right.get_x() (src/batch.rs:98-100, :125-129).
__global__ void __launch_bounds__(128)
msm_k_finish(const u64 *__restrict__ win_in, MsmShape sh, const u64 *__restrict__ partials, u32 n_partials,
             const u64 *__restrict__ gtab, const u32 *__restrict__ malformed, u32 *__restrict__ verdict,
             u64 *__restrict__ partial_out) {
    // partial_out != nullptr: this device holds one shard of the batch (ssa_multi_verify_batch_msm): emit the
    // shard's left-hand point (X, Y, Z: words 0..17), its sum s_i e_i (18..21) and the malformed flag (22) instead
    // of a verdict; device 0 adds the shards up with this same kernel (sh.c = 0: no doublings between the "windows")
    __shared__ CoopLds L;
    __shared__ u64 lin_sh[64][4];
    const u32 lane = threadIdx.x & 63u;
    const int ws = (int)(threadIdx.x >> 6);
    if (*malformed) {   // block-uniform
        if (partial_out) {   // a whole, well-formed record: no word is left to whatever the buffer held before
            if (threadIdx.x < 24)
                partial_out[threadIdx.x] = threadIdx.x == 22 ? 1ull : threadIdx.x == 23 ? SSA_MSM_RECORD_MAGIC : 0ull;
        } else if (threadIdx.x == 0) {
            *verdict = ST_MALFORMED;
        }
        return;
    }
    // slots: wave 0 accumulator 0..3 (X, Y, Z, W), addend 4..6, scratch 7..15; wave 1 accumulator 20..23,
    // addend 24..25, scratch 26..34
    int t[9];
#pragma unroll
    for (int k = 0; k < 9; k++) t[k] = (ws ? 26 : 7) + k;
    if (ws == 0) {
        auto load = [&](int s0, u32 j) {   // X, Y, Z of window j with their 7x halves
            if (lane < 36) {
                const u32 v = lane / 12u, c = lane % 12u;
                const u64 w = win_in[18 * (size_t)j + 6u * v + c % 6u];
                L.slot[s0 + (int)v][c] = c < 6 ? w : fp_mul_small(w, 7u);
            }
            coop_sync();
        };
        load(0, sh.windows - 1);
        coop_mul(L, 3, 2, 2, lane, ws);    // W = Z^4 of the accumulator
        coop_mul(L, 3, 3, 3, lane, ws);
#pragma unroll 1
        for (int j = (int)sh.windows - 2; j >= 0; j--) {
#pragma unroll 1
            for (u32 d = 0; d < sh.c; d++) coop_jac_dbl(L, 0, t, lane, ws);
            load(4, (u32)j);
            coop_jac_add(L, 0, 4, t, lane, ws);
        }
    } else {
        sc256 acc;
#pragma unroll
        for (int k = 0; k < 4; k++) acc.w[k] = 0;
#pragma unroll 1
        for (u32 b = lane; b < n_partials; b += 64) {
            sc256 p;
#pragma unroll
            for (int k = 0; k < 4; k++) p.w[k] = partials[4 * b + k];
            acc = sc_add_mod(acc, p);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) lin_sh[lane][k] = acc.w[k];
        coop_sync();
#pragma unroll 1
        for (u32 stride = 32; stride >= 1; stride >>= 1) {
            if (lane < stride) {
                sc256 a, b;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    a.w[k] = lin_sh[lane][k];
                    b.w[k] = lin_sh[lane + stride][k];
                }
                a = sc_add_mod(a, b);
#pragma unroll
                for (int k = 0; k < 4; k++) lin_sh[lane][k] = a.w[k];
            }
            coop_sync();
        }
        sc256 lin;
#pragma unroll
        for (int k = 0; k < 4; k++) lin.w[k] = lin_sh[0][k];
        if (partial_out && lane < 4) partial_out[18 + lane] = lin.w[lane];
        coop_set(L, 20, 1ull, lane, ws);
        coop_set(L, 21, 1ull, lane, ws);
        coop_set(L, 22, 0ull, lane, ws);
        coop_set(L, 23, 0ull, lane, ws);
#pragma unroll 1
        for (int w = 0; w < (partial_out ? 0 : GW_COUNT); w++) {      // BASEPOINT_TABLE.multiply_vartime
            const u32 d = sc_win16(lin, (u32)w);
            if (d != 0) {
                const u64 *rowp = gtab + (((size_t)w << GW_BITS) + d) * 12;
                if (lane < 24) {
                    const u32 half = lane / 12u, c = lane % 12u;
                    const u64 v = rowp[6u * half + c % 6u];
                    L.slot[half ? 25 : 24][c] = c < 6 ? v : fp_mul_small(v, 7u);
                }
                coop_sync();
                coop_jac_madd(L, 20, 24, 25, t, lane, ws);
            }
        }
    }
    __syncthreads();
    if (partial_out) {
        if (ws == 0 && lane < 18) partial_out[lane] = fp_canon(L.slot[(int)(lane / 6u)][lane % 6u]);
        if (threadIdx.x == 0) {
            partial_out[22] = 0;
            partial_out[23] = SSA_MSM_RECORD_MAGIC;
        }
        return;
    }
    if (ws == 0) {
        // X_l Z_r^2 == X_r Z_l^2; the identity's x is taken as 0
        const bool li = coop_is_zero(L, 2, lane, ws), ri = coop_is_zero(L, 22, lane, ws);
        bool eq;
        if (li || ri) {
            eq = (li && ri) || (li && coop_is_zero(L, 20, lane, ws)) || (ri && coop_is_zero(L, 0, lane, ws));
        } else {
            coop_mul(L, 7, 22, 22, lane, ws);
            coop_mul(L, 7, 0, 7, lane, ws);
            coop_mul(L, 8, 2, 2, lane, ws);
            coop_mul(L, 8, 20, 8, lane, ws);
            eq = coop_eq(L, 7, 8, lane, ws);
        }
        if (lane == 0) *verdict = eq ? ST_OK : ST_INVALID_SIG;
    }
}

// ---- small batches: Straus on cooperating waves ---------------------------------------------------
// The bucket method pays ~20 launches of lone, latency-bound waves however small the batch is (2.5 ms at the
// reference's own bench sizes, benches/schnorr.rs:78-96: 4..128 signatures).  Below MSM_SMALL_MAX signatures the
// same equation is evaluated term by term instead: ONE two-wave block per signature computes
//     T_i = [s_i] R_i - [s_i h_i] P_i      and      s_i e_i mod q
// with the wave-cooperative arithmetic of ssa_coop.hpp (table of eight multiples + signed 4-bit windows for each
// of the two points, the same code as the low-latency verification kernel), and writes them as one 24-word record
// -- the very record a shard of a multi-process batch produces (include/schnorr_sig_amd.h), so the records are summed
// by the same combination kernel.
//   wave 0: stage, P on the curve, table of P      ||  wave 1: hash_message -> h, s h mod q, s e mod q
//   wave 0: decompress R, table of R, [s] R        ||  wave 1: [s h] P from P's table, negate
//   wave 0: T = [s]R + (-[s h]P), record
struct MsmSmallShared {
    u32 ok, r_ok, r_inf;
    u64 s[4], b[4], se[4];
};

__global__ void __launch_bounds__(128)
msm_k_small(const DevParams *__restrict__ prm, const u8 *__restrict__ sigs, const u8 *__restrict__ pks,
            const u8 *__restrict__ pk_inf, MsgView mv, const u8 *__restrict__ coeffs, u32 coeff_bytes, size_t n,
            u64 *__restrict__ records) {
    __shared__ CoopLds L;
    __shared__ MsmSmallShared sh;
    using namespace coop_slots;
    const size_t i = blockIdx.x;
    if (i >= n) return;
    const u32 lane = threadIdx.x & 63u;
    const int ws = (int)(threadIdx.x >> 6);
    COOP_WORKING_SET(ws);
    int t[9];
#pragma unroll
    for (int k = 0; k < 9; k++) t[k] = T0 + k;
    const u8 *sig = sigs + 81 * i, *pk = pks + 96 * i;
    const bool inf = pk_inf && pk_inf[i];
    u64 *rec = records + 24 * i;
    u32 len;
    const u8 *m = msg_ptr(mv, i, len);
    const sc256 e = ld_sc(sig + 49);
    if (ws == 0) {   // stage the inputs, canonical-limb checks (the reference panics on these: src/batch.rs:67,104)
        bool ok = true;
        if (lane < 12) {
            const u32 c = lane % 6u;
            const u64 xs = ld_u64_le(sig + 8 * c), px = ld_u64_le(pk + 8 * c), py = ld_u64_le(pk + 48 + 8 * c);
            coop_store7(L, SX, xs, lane);
            coop_store7(L, PX, px, lane);
            coop_store7(L, PY, py, lane);
            ok = px < FP_P && py < FP_P && xs < FP_P;
        }
        ok = __all(ok) && !sc_geq_q(e);
        if (lane == 0) {
            sh.ok = ok;
            sh.r_ok = 0;
            sh.r_inf = 0;
        }
    }
    __syncthreads();
    // wave 0's share of R: decompression (one lane: the Fp6 square root is a long serial chain either way) and the
    // table of its multiples.  With 128-bit coefficients [s]R is half a ladder and all of this fits beside wave 1's
    // [s h]P; with full-width coefficients (a shim passing Scalar::random) both ladders are equally long and R's table
    // is built BEFORE the first barrier instead, beside wave 1's hash.
    const bool wide = coeff_bytes > 16;
    auto r_tables = [&]() {
        if (lane == 0) {   // R = from_compressed(sig.x).unwrap() (:104)
            aff R;
            bool r_inf = false;
            const u32 st = decompress_lane(sig, R, r_inf);
            sh.r_ok = st == 0;
            sh.r_inf = r_inf;
#pragma unroll
            for (int c = 0; c < 6; c++) {
                L.slot[RX][c] = R.x.c[c];
                L.slot[RX][6 + c] = fp_mul_small(R.x.c[c], 7u);
                L.slot[RY][c] = R.y.c[c];
                L.slot[RY][6 + c] = fp_mul_small(R.y.c[c], 7u);
            }
        }
        coop_sync();
        if (sh.r_ok) coop_build_table(L, sh.r_inf != 0, lane, ws, TAB2, RX, RY);   // (lane 0 wrote it before the fence)
    };
    if (ws == 0) {
        bool ok = sh.ok != 0;
        if (ok && !inf) {   // y^2 == x^3 + x + (u + 395)
            coop_mul(L, T0, PX, PX, lane, ws);
            coop_mul(L, T0, T0, PX, lane, ws);
            coop_add(L, T0, T0, PX, lane, ws);
            if (lane < 2) L.slot[T0][lane] = fp_add(L.slot[T0][lane], lane == 0 ? 395ull : 1ull);   // compared only
            coop_sync();
            coop_mul(L, T0 + 1, PY, PY, lane, ws);
            ok = coop_eq(L, T0, T0 + 1, lane, ws);
        }
        if (lane == 0) sh.ok = ok;
        if (ok) {
            coop_build_table(L, inf, lane, ws);
            if (wide) r_tables();
        }
    } else {
        const sc256 h = coop_hash_message(L, prm, m, len, lane, ws);   // reads SX, PX, PY only; h_i, src/batch.rs:64-73
        if (lane == 0) {
            sc256 s;
#pragma unroll
            for (int k = 0; k < 4; k++) s.w[k] = 0;
            const u8 *cp = coeffs + (size_t)coeff_bytes * i;
            for (u32 b = 0; b < coeff_bytes; b++) s.w[b >> 3] |= (u64)cp[b] << (8 * (b & 7u));
            s = sc_reduce256(s);                                       // Scalar::random, :75-78
            const sc256 sb = sc_mul_mod(s, h), se = sc_mul_mod(s, e);  // :109-111, :92-97
#pragma unroll
            for (int k = 0; k < 4; k++) {
                sh.s[k] = s.w[k];
                sh.b[k] = sb.w[k];
                sh.se[k] = se.w[k];
            }
        }
    }
    __syncthreads();
    if (!sh.ok) {   // block-uniform
        if (threadIdx.x < 24) rec[threadIdx.x] = threadIdx.x == 22 ? 1ull : threadIdx.x == 23 ? SSA_MSM_RECORD_MAGIC : 0ull;
        return;
    }
    if (ws == 0) {
        if (!wide) r_tables();
        if (sh.r_ok) {
            sc256 s;
#pragma unroll
            for (int k = 0; k < 4; k++) s.w[k] = sh.s[k];
            coop_mul_table(L, s, lane, ws, TAB2);                           // [s_i] R_i
        }
    } else {
        sc256 b;
#pragma unroll
        for (int k = 0; k < 4; k++) b.w[k] = sh.b[k];
        coop_mul_table(L, b, lane, ws);                                     // [s_i h_i] P_i  (the identity for an identity key)
        coop_neg(L, AY, AY, lane, ws);                                      // k.0.neg(), :106
    }
    __syncthreads();
    if (!sh.r_ok) {
        if (threadIdx.x < 24) rec[threadIdx.x] = threadIdx.x == 22 ? 1ull : threadIdx.x == 23 ? SSA_MSM_RECORD_MAGIC : 0ull;
        return;
    }
    if (ws == 0) {
        // wave 1's accumulator as the second operand (X, Y, Z with their 7x halves) in this wave's I0..I2
        if (lane < 18) {
            const u32 v = lane / 6u, c = lane % 6u;
            const u64 w = L.slot[WS_SLOTS + (int)v][c];
            L.slot[I0 + (int)v][c] = w;
            L.slot[I0 + (int)v][6 + c] = fp_mul_small(w, 7u);
        }
        coop_sync();
        coop_jac_add(L, AX, I0, t, lane, ws);
        if (lane < 18) rec[lane] = fp_canon(L.slot[AX + (int)(lane / 6u)][lane % 6u]);
        else if (lane < 22) rec[lane] = sh.se[lane - 18u];
        else if (lane == 22) rec[lane] = 0ull;
        else if (lane == 23) rec[lane] = SSA_MSM_RECORD_MAGIC;
    }
}

// records [g * group, (g + 1) * group) -> one record (one wave per group): the points by cooperative general
// additions, the scalars mod q, the malformed flags OR-ed
__global__ void __launch_bounds__(64)
msm_k_sum_records(const u64 *__restrict__ in, u32 count, u32 group, u64 *__restrict__ out) {
    __shared__ CoopLds L;
    const u32 lane = threadIdx.x, g = blockIdx.x;
    const u32 lo = g * group, hi = lo + group < count ? lo + group : count;
    if (lo >= hi) return;
    int t[9];
#pragma unroll
    for (int k = 0; k < 9; k++) t[k] = 7 + k;
    bool bad = false;
    sc256 lin;
#pragma unroll
    for (int k = 0; k < 4; k++) lin.w[k] = 0;
    // accumulator 0..3 (X, Y, Z, W = Z^4), addend 4..6
    auto load = [&](int s0, u32 j) {
        if (lane < 36) {
            const u32 v = lane / 12u, c = lane % 12u;
            const u64 w = in[24 * (size_t)j + 6u * v + c % 6u];
            L.slot[s0 + (int)v][c] = c < 6 ? w : fp_mul_small(w, 7u);
        }
        coop_sync();
    };
    load(0, lo);
    coop_mul(L, 3, 2, 2, lane, 0);
    coop_mul(L, 3, 3, 3, lane, 0);
#pragma unroll 1
    for (u32 j = lo; j < hi; j++) {
        bad = bad || in[24 * (size_t)j + 22] != 0;
        sc256 p;
#pragma unroll
        for (int k = 0; k < 4; k++) p.w[k] = in[24 * (size_t)j + 18 + k];
        lin = sc_add_mod(lin, p);
        if (j > lo) {
            load(4, j);
            coop_jac_add(L, 0, 4, t, lane, 0);
        }
    }
    u64 *rec = out + 24 * (size_t)g;
    if (lane < 18) rec[lane] = bad ? 0ull : fp_canon(L.slot[(int)(lane / 6u)][lane % 6u]);
    else if (lane < 22) rec[lane] = lin.w[lane - 18u];
    else if (lane == 22) rec[lane] = bad ? 1ull : 0ull;
    else if (lane == 23) rec[lane] = SSA_MSM_RECORD_MAGIC;
}

// the record of an empty shard: the identity (Z = 0), sum s_i e_i = 0, not malformed -- and the magic word: a buffer that
// nobody wrote (all zero) is NOT a record
__global__ void msm_k_empty_record(u64 *__restrict__ rec) {
    if (threadIdx.x < 24) rec[threadIdx.x] = threadIdx.x == 23 ? SSA_MSM_RECORD_MAGIC : 0ull;
}

}  // namespace ssa

// ------------------------------------------------------------------------------------------------
static MsmShape msm_shape(size_t n) {
    // Window width: 16 bits, or 8 for small batches.  Both divide 128 (library-drawn coefficients) and
    // leave a wide top window for 255-bit scalars (255 mod 16 = 15, 255 mod 8 = 7): a narrow partial
    // window would have a handful of digits and therefore a handful of enormous buckets (measured:
    // c = 14 put n/4 points into single lanes and took 0.9 s at n = 2^18).
    MsmShape sh;
    sh.c = n >= 4096 ? 16u : 8u;
    sh.windows = (255 + sh.c - 1) / sh.c;
    sh.buckets = 1u << sh.c;
    sh.chunks = sh.buckets <= (u32)MSM_CHUNK ? 1u : sh.buckets / (u32)MSM_CHUNK;
    return sh;
}

// n_blocks 64-byte ChaCha20 blocks into d_out (device), on the context's stream
static int msm_chacha20(ssa_ctx *ctx, const uint8_t key[32], const uint8_t nonce[12], uint32_t counter0,
                        size_t n_blocks, void *d_out) {
    ChaChaKey kn;
    memcpy(kn.key, key, 32);
    memcpy(kn.nonce, nonce, 12);
    if (n_blocks == 0) return 0;
    hipLaunchKernelGGL(msm_k_chacha20, dim3(grid_for(n_blocks, 256)), dim3(256), 0, ctx->stream, kn, counter0, n_blocks,
                       (u32 *)d_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

// fresh 128-bit coefficients for n signatures in ctx->st_coeffs (device)
static int msm_draw_coefficients(ssa_ctx *ctx, size_t n, const void **d_out) {
    uint8_t seed[44];
    size_t got = 0;
    while (got < sizeof seed) {
        ssize_t r = getrandom(seed + got, sizeof seed - got, 0);
        if (r <= 0) return SSA_ERR_ARG;
        got += (size_t)r;
    }
    const size_t n_blocks = (n * 16 + 63) / 64;
    if (ctx->st_coeffs.reserve(n_blocks * 64)) return SSA_ERR_HIP;
    if (int rc = msm_chacha20(ctx, seed, seed + 32, 0u, n_blocks, ctx->st_coeffs.p)) return rc;
    *d_out = ctx->st_coeffs.p;
    return 0;
}

extern "C" int ssa_debug_chacha20(ssa_ctx *ctx, const uint8_t key[32], const uint8_t nonce[12], uint32_t counter0,
                                  size_t n_blocks, uint8_t *out) {
    if (!ctx || !key || !nonce || (n_blocks && !out)) return SSA_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->st_coeffs.reserve(n_blocks * 64 + 64)) return SSA_ERR_HIP;
    if (int rc = msm_chacha20(ctx, key, nonce, counter0, n_blocks, ctx->st_coeffs.p)) return rc;
    HIP_TRY(hipMemcpyAsync(out, ctx->st_coeffs.p, n_blocks * 64, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

static int msm_run(ssa_ctx *ctx, const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_pk_inf,
                   const uint8_t *d_msgs, const uint64_t *d_msg_off, size_t msg_stride, size_t msg_len, size_t n,
                   const uint8_t *d_coeffs, uint32_t coeff_bytes, uint32_t *d_verdict_out, u64 *d_partial_out,
                   bool hashed = false);
static int msm_run_one(ssa_ctx *ctx, const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_pk_inf,
                       const uint8_t *d_msgs, const uint64_t *d_msg_off, size_t msg_stride, size_t msg_len, size_t n,
                       const uint8_t *d_coeffs, uint32_t coeff_bytes, uint32_t *d_verdict_out, u64 *d_partial_out,
                       const u64 *d_h);
static int msm_run_small(ssa_ctx *ctx, const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_pk_inf,
                         const uint8_t *d_msgs, const uint64_t *d_msg_off, size_t msg_stride, size_t msg_len, size_t n,
                         const uint8_t *d_coeffs, uint32_t coeff_bytes, uint32_t *d_verdict_out, u64 *d_partial_out);

extern "C" int ssa_verify_batch_msm_device(ssa_ctx *ctx, const uint8_t *d_sigs, const uint8_t *d_pks,
                                           const uint8_t *d_pk_inf, const uint8_t *d_msgs,
                                           const uint64_t *d_msg_off, size_t msg_stride,
                                           size_t msg_len, size_t n, const uint8_t *d_coeffs, uint32_t coeff_bytes,
                                           uint32_t *d_verdict_out) {
    if (!d_verdict_out) return SSA_ERR_ARG;
    return msm_run(ctx, d_sigs, d_pks, d_pk_inf, d_msgs, d_msg_off, msg_stride, msg_len, n, d_coeffs, coeff_bytes,
                   d_verdict_out, nullptr);
}

// check_points: the records come from outside this call (ssa_msm_combine*): scalars and points are validated too
static int msm_combine_records(ssa_ctx *ctx, const u64 *d_records, size_t k, uint32_t *d_verdict_out, u64 *d_partial_out,
                               bool check_points = false);

// Small batch (n <= ctx->msm_small_max): one cooperative block per signature, then the records are summed -- in
// groups of 16 by one wave each while there are more than 16 of them, the rest by the combination kernel, which also
// computes [sum s_i e_i] G and compares (or emits the shard's record).
static int msm_run_small(ssa_ctx *ctx, const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_pk_inf,
                         const uint8_t *d_msgs, const uint64_t *d_msg_off, size_t msg_stride, size_t msg_len, size_t n,
                         const uint8_t *d_coeffs, uint32_t coeff_bytes, uint32_t *d_verdict_out, u64 *d_partial_out) {
    const size_t groups = (n + 15) / 16;
    if (ctx->msm_buckets.reserve(n * 24 * sizeof(u64)) || ctx->msm_chunks.reserve(groups * 24 * sizeof(u64)) ||
        ctx->msm_windows.reserve(groups * 24 * sizeof(u64)))
        return SSA_ERR_HIP;
    MsgView mv{d_msgs, d_msg_off, msg_stride, msg_len};
    int rc = timed_launch(ctx, "msm_k_small", [&] {
        hipLaunchKernelGGL(msm_k_small, dim3((unsigned)n), dim3(128), 0, ctx->stream, ctx->d_params, d_sigs, d_pks, d_pk_inf,
                           mv, d_coeffs, coeff_bytes, n, (u64 *)ctx->msm_buckets.p);
    });
    if (rc) return rc;
    const u64 *recs = (const u64 *)ctx->msm_buckets.p;
    size_t count = n;
    u64 *ping = (u64 *)ctx->msm_chunks.p, *pong = (u64 *)ctx->msm_windows.p;
    while (count > 16) {     // the combination kernel adds its records one after the other: hand it at most 16
        const size_t g = (count + 15) / 16;
        hipLaunchKernelGGL(msm_k_sum_records, dim3((unsigned)g), dim3(64), 0, ctx->stream, recs, (u32)count, 16u, ping);
        HIP_TRY(hipGetLastError());
        recs = ping;
        u64 *tmp = ping;
        ping = pong;
        pong = tmp;
        count = g;
    }
    return msm_combine_records(ctx, recs, count, d_verdict_out, d_partial_out);
}

// A batch of any size (n <= SSA_MAX_BATCH) in bounded memory: more than ctx->msm_slice signatures run slice after slice,
// every slice reduced to its 24-word record exactly as a shard of a multi-GPU batch is (src/batch.rs:98-129: one point
// and one scalar per part), and the records are added up by the combination kernel -- one point addition per slice.
// hashed: ctx->ws_h already holds the challenge scalars of the WHOLE batch (the host-buffer pipeline computed them).
static int msm_run(ssa_ctx *ctx, const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_pk_inf,
                   const uint8_t *d_msgs, const uint64_t *d_msg_off, size_t msg_stride, size_t msg_len, size_t n,
                   const uint8_t *d_coeffs, uint32_t coeff_bytes, uint32_t *d_verdict_out, u64 *d_partial_out,
                   bool hashed) {
    if (!ctx || (!d_verdict_out && !d_partial_out)) return SSA_ERR_ARG;
    if (n && (!d_sigs || !d_pks)) return SSA_ERR_ARG;
    if (d_coeffs && (coeff_bytes == 0 || coeff_bytes > 32)) return SSA_ERR_ARG;
    if (int rc = check_msgs(d_msgs, d_msg_off, msg_stride, msg_len, n)) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    const u64 *d_h = hashed ? (const u64 *)ctx->ws_h.p : nullptr;
    if (n <= ctx->msm_slice)
        return msm_run_one(ctx, d_sigs, d_pks, d_pk_inf, d_msgs, d_msg_off, msg_stride, msg_len, n, d_coeffs, coeff_bytes,
                           d_verdict_out, d_partial_out, d_h);
    const size_t slice = ctx->msm_slice, k = (n + slice - 1) / slice;
    if (ctx->msm_slice_recs.reserve(k * 24 * sizeof(u64))) return SSA_ERR_HIP;
    u64 *recs = (u64 *)ctx->msm_slice_recs.p;
    for (size_t j = 0; j < k; j++) {
        const size_t lo = j * slice, cnt = n - lo < slice ? n - lo : slice;
        if (int rc = msm_run_one(ctx, d_sigs + 81 * lo, d_pks + 96 * lo, d_pk_inf ? d_pk_inf + lo : nullptr,
                                 d_msg_off ? d_msgs : (d_msgs ? d_msgs + lo * msg_stride : nullptr),
                                 d_msg_off ? d_msg_off + lo : nullptr, msg_stride, msg_len, cnt,
                                 d_coeffs ? d_coeffs + (size_t)coeff_bytes * lo : nullptr, coeff_bytes, nullptr,
                                 recs + 24 * j, d_h ? d_h + 4 * lo : nullptr))
            return rc;
    }
    return msm_combine_records(ctx, recs, k, d_verdict_out, d_partial_out);
}

// the kernels of one MSM-form slice on ctx->stream: a verdict (d_partial_out == nullptr) or the slice's / shard's record
// d_h: the challenge scalars if they exist already, else nullptr (they are computed into ctx->ws_h)
static int msm_run_one(ssa_ctx *ctx, const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_pk_inf,
                       const uint8_t *d_msgs, const uint64_t *d_msg_off, size_t msg_stride, size_t msg_len, size_t n,
                       const uint8_t *d_coeffs, uint32_t coeff_bytes, uint32_t *d_verdict_out, u64 *d_partial_out,
                       const u64 *d_h) {
    if (n > (1ull << 25)) return SSA_ERR_ARG;   // 2n * 16 sort items must fit the sort's 32-bit positions
    if (n == 0) {   // empty batch: Ok (src/batch.rs); an empty shard adds the identity and 0
        if (d_partial_out) {
            hipLaunchKernelGGL(msm_k_empty_record, dim3(1), dim3(64), 0, ctx->stream, d_partial_out);
            HIP_TRY(hipGetLastError());
            return 0;
        }
        HIP_TRY(hipMemsetAsync(d_verdict_out, 0, sizeof(uint32_t), ctx->stream));
        return 0;
    }
    if (!d_coeffs) {   // Scalar::random(rng): the library draws 128-bit coefficients
        const void *p;
        if (int rc = msm_draw_coefficients(ctx, n, &p)) return rc;
        d_coeffs = (const uint8_t *)p;
        coeff_bytes = 16;
    }
    if (n <= ctx->msm_small_max) return msm_run_small(ctx, d_sigs, d_pks, d_pk_inf, d_msgs, d_msg_off, msg_stride, msg_len, n,
                                                      d_coeffs, coeff_bytes, d_verdict_out, d_partial_out);
    const MsmShape sh = msm_shape(n);
    // windows the coefficients themselves can reach (they are reduced mod q when they are as wide as q)
    const u32 wa = coeff_bytes >= 32 ? sh.windows : (8u * coeff_bytes + sh.c - 1) / sh.c;
    const size_t npts = 2 * n, total = n * wa + n * sh.windows, nb = (size_t)sh.windows * sh.buckets;
    const unsigned n_blocks = grid_for(n, 256);
    size_t sort_tmp = 0;
    const int end_bit = 32 - __builtin_clz((unsigned)(nb - 1) | 1u);
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, sort_tmp, (const u32 *)nullptr, (u32 *)nullptr,
                                           (const u32 *)nullptr, (u32 *)nullptr, (int)total, 0, end_bit,
                                           ctx->stream) != hipSuccess)
        return SSA_ERR_HIP;
    size_t sort_tmp2 = 0;
    const int cnt_bits = 33 - __builtin_clz((unsigned)npts | 1u);   // a bucket holds at most 2n points
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, sort_tmp2, (const u32 *)nullptr, (u32 *)nullptr,
                                           (const u32 *)nullptr, (u32 *)nullptr, (int)nb, 0, cnt_bits > 32 ? 32 : cnt_bits,
                                           ctx->stream) != hipSuccess)
        return SSA_ERR_HIP;
    if (sort_tmp2 > sort_tmp) sort_tmp = sort_tmp2;
    if (ctx->msm_cnt.reserve(nb * 4) || ctx->msm_cnt2.reserve(nb * 4) || ctx->msm_ids.reserve(nb * 4) ||
        ctx->msm_ids2.reserve(nb * 4))
        return SSA_ERR_HIP;
    if ((!d_h && ctx->ws_h.reserve(n * 32)) || ctx->msm_points.reserve(npts * 96) || ctx->msm_scalars.reserve(npts * 32) ||
        ctx->msm_keys.reserve(total * 4) || ctx->msm_vals.reserve(total * 4) || ctx->msm_keys2.reserve(total * 4) ||
        ctx->msm_vals2.reserve(total * 4) || ctx->msm_sort_tmp.reserve(sort_tmp + 16) ||
        ctx->msm_bounds.reserve(nb * 8) || ctx->msm_buckets.reserve(nb * 144) ||
        ctx->msm_chunks.reserve((size_t)sh.windows * sh.chunks * 144) ||
        ctx->msm_windows.reserve((size_t)sh.windows * sh.chunks * 144) ||
        ctx->msm_partials.reserve((size_t)n_blocks * 32) || ctx->msm_flags.reserve(64))
        return SSA_ERR_HIP;
    HIP_TRY(hipMemsetAsync(ctx->msm_flags.p, 0, 64, ctx->stream));
    HIP_TRY(hipMemsetAsync(ctx->msm_bounds.p, 0, nb * 8, ctx->stream));
    // challenge scalars h_i with the kernel of the per-lane path
    if (!d_h) {
        if (int rc = ssa_internal_hash_scalars(ctx, d_sigs, d_pks, d_msgs, d_msg_off, msg_stride, msg_len, n)) return rc;
        d_h = (const u64 *)ctx->ws_h.p;
    }
    int rc = timed_launch(ctx, "msm_k_prepare", [&] {
        hipLaunchKernelGGL(msm_k_prepare, dim3(n_blocks), dim3(256), 0, ctx->stream, d_sigs, d_pks, d_pk_inf,
                           d_h, d_coeffs, coeff_bytes, n, (u64 *)ctx->msm_points.p,
                           (u64 *)ctx->msm_scalars.p, (u64 *)ctx->msm_partials.p, (u32 *)ctx->msm_flags.p);
    });
    if (rc) return rc;
    rc = timed_launch(ctx, "msm_sort", [&] {
        hipLaunchKernelGGL(msm_k_digits, dim3(grid_for(total, 256)), dim3(256), 0, ctx->stream,
                           (const u64 *)ctx->msm_scalars.p, n, wa, sh, (u32 *)ctx->msm_keys.p, (u32 *)ctx->msm_vals.p);
        (void)hipcub::DeviceRadixSort::SortPairs(ctx->msm_sort_tmp.p, sort_tmp, (const u32 *)ctx->msm_keys.p,
                                                 (u32 *)ctx->msm_keys2.p, (const u32 *)ctx->msm_vals.p,
                                                 (u32 *)ctx->msm_vals2.p, (int)total, 0, end_bit, ctx->stream);
        hipLaunchKernelGGL(msm_k_bounds, dim3(grid_for(total, 256)), dim3(256), 0, ctx->stream,
                           (const u32 *)ctx->msm_keys2.p, total, (u32 *)ctx->msm_bounds.p);
    });
    if (rc) return rc;
    rc = timed_launch(ctx, "msm_k_buckets", [&] {
        hipLaunchKernelGGL(msm_k_counts, dim3(grid_for(nb, 256)), dim3(256), 0, ctx->stream,
                           (const u32 *)ctx->msm_bounds.p, sh, (u32 *)ctx->msm_cnt.p, (u32 *)ctx->msm_ids.p);
        (void)hipcub::DeviceRadixSort::SortPairs(ctx->msm_sort_tmp.p, sort_tmp2, (const u32 *)ctx->msm_cnt.p,
                                                 (u32 *)ctx->msm_cnt2.p, (const u32 *)ctx->msm_ids.p,
                                                 (u32 *)ctx->msm_ids2.p, (int)nb, 0, cnt_bits > 32 ? 32 : cnt_bits, ctx->stream);
        hipLaunchKernelGGL(msm_k_buckets, dim3(grid_for(nb, 256)), dim3(256), 0, ctx->stream,
                           (const u64 *)ctx->msm_points.p, (const u32 *)ctx->msm_vals2.p,
                           (const u32 *)ctx->msm_bounds.p, (const u32 *)ctx->msm_ids2.p, sh,
                           (u64 *)ctx->msm_buckets.p);
    });
    if (rc) return rc;
    return timed_launch(ctx, "msm_reduce", [&] {
        // per window: running sums on chunks of MSM_CHUNK buckets, a tree over the chunk sums, then one
        // cooperative block: Horner over the windows || [lin]G, and the comparison
        hipLaunchKernelGGL(msm_k_chunks, dim3(grid_for((size_t)sh.windows * sh.chunks, 256)), dim3(256), 0,
                           ctx->stream, (const u64 *)ctx->msm_buckets.p, sh, (u64 *)ctx->msm_chunks.p);
        u64 *ping = (u64 *)ctx->msm_chunks.p, *pong = (u64 *)ctx->msm_windows.p;
        u32 count = sh.chunks;
        while (count > 1) {
            const u32 groups = (count + MSM_TREE_GROUP - 1) / MSM_TREE_GROUP;
            hipLaunchKernelGGL(msm_k_tree, dim3(grid_for((size_t)sh.windows * groups, 64)), dim3(64), 0, ctx->stream,
                               (const u64 *)ping, sh.windows, count, MSM_TREE_GROUP, pong);
            u64 *tmp = ping;
            ping = pong;
            pong = tmp;
            count = groups;
        }
        hipLaunchKernelGGL(msm_k_finish, dim3(1), dim3(128), 0, ctx->stream, (const u64 *)ping, sh,
                           (const u64 *)ctx->msm_partials.p, n_blocks, (const u64 *)ctx->d_gtab,
                           (const u32 *)ctx->msm_flags.p, d_verdict_out, d_partial_out);
    });
}

// One shard of a batch that spans several devices: stage the host buffers, run the MSM pipeline, return the shard's
// 24-word partial record (left-hand point, sum s_i e_i, malformed flag) in host memory.
extern "C" int ssa_verify_batch_msm_partial(ssa_ctx *ctx, const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf,
                                            const uint8_t *msgs, const uint64_t *msg_off, size_t msg_stride,
                                            size_t msg_len, size_t n, const uint8_t *coeffs,
                                            uint64_t out24[SSA_MSM_PARTIAL_WORDS]) {
    if (!ctx || !out24 || (n && (!sigs || !pks))) return SSA_ERR_ARG;
    if (int rc = check_msgs(msgs, msg_off, msg_stride, msg_len, n)) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    std::memset(out24, 0, SSA_MSM_PARTIAL_WORDS * sizeof(uint64_t));
    if (n == 0) {
        out24[23] = SSA_MSM_RECORD_MAGIC;    // the empty shard's record: identity, 0
        return 0;
    }
    StagedInputs s;
    const void *p;
    if (int rc = stage_up(ctx, ctx->st_sigs, sigs, n * 81, &p)) return rc;
    s.sigs = (const u8 *)p;
    if (int rc = stage_up(ctx, ctx->st_pks, pks, n * 96, &p)) return rc;
    s.pks = (const u8 *)p;
    if (pk_inf) {
        if (int rc = stage_up(ctx, ctx->st_inf, pk_inf, n, &p)) return rc;
        s.inf = (const u8 *)p;
    }
    if (int rc = stage_msgs(ctx, msgs, msg_off, msg_stride, msg_len, n, s)) return rc;
    p = nullptr;
    if (coeffs) {
        if (int rc = stage_up(ctx, ctx->st_coeffs, coeffs, n * 32, &p)) return rc;
    }
    if (ctx->st_aux2.reserve(SSA_MSM_PARTIAL_WORDS * sizeof(u64))) return SSA_ERR_HIP;
    if (int rc = msm_run(ctx, s.sigs, s.pks, s.inf, s.msgs, s.off, msg_stride, msg_len, n, (const u8 *)p, 32, nullptr,
                         (u64 *)ctx->st_aux2.p))
        return rc;
    HIP_TRY(hipMemcpyAsync(out24, ctx->st_aux2.p, SSA_MSM_PARTIAL_WORDS * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

// the device-buffer form: what one rank of a process-per-GPU job calls on its shard (the records then travel by
// all-gather, 24 words per rank)
extern "C" int ssa_verify_batch_msm_partial_device(ssa_ctx *ctx, const uint8_t *d_sigs, const uint8_t *d_pks,
                                                   const uint8_t *d_pk_inf, const uint8_t *d_msgs,
                                                   const uint64_t *d_msg_off, size_t msg_stride, size_t msg_len, size_t n,
                                                   const uint8_t *d_coeffs, uint32_t coeff_bytes,
                                                   uint64_t *d_partial_out) {
    if (!d_partial_out) return SSA_ERR_ARG;
    return msm_run(ctx, d_sigs, d_pks, d_pk_inf, d_msgs, d_msg_off, msg_stride, msg_len, n, d_coeffs, coeff_bytes, nullptr,
                   (u64 *)d_partial_out);
}

namespace ssa {
// k partial records (24-word stride) -> the layout msm_k_finish reads: k points of 18 words, k scalars of 4 words, and
// the OR of the malformed flags
__global__ void msm_k_unpack_parts(const u64 *__restrict__ parts, u32 k, u64 *__restrict__ pts, u64 *__restrict__ lins,
                                   u32 *__restrict__ malformed) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= k * 24u) return;
    const u32 j = t / 24u, w = t % 24u;
    const u64 v = parts[t];
    if (w < 18) {
        pts[18u * j + w] = v;
        if (v >= FP_P) atomicOr(malformed, 1u);                 // limbs of a record are canonical
    } else if (w < 22) {
        lins[4u * j + (w - 18u)] = v;
    } else if (w == 22) {
        if (v != 0) atomicOr(malformed, 1u);
    } else if (v != SSA_MSM_RECORD_MAGIC) {
        atomicOr(malformed, 1u);   // not a record of this format: never written (all zero), foreign version, garbled
    }
}

// One lane per record: the scalar is canonical (< q) and the point is the identity (Z = 0) or satisfies the Jacobian
// curve equation Y^2 = X^3 + X Z^4 + (u + 395) Z^6.  Records cross process boundaries (all-gather): a corrupted or
// foreign one gives SSA_MALFORMED, never an arbitrary verdict.  k <= 4096: the cost is one short launch.
__global__ void __launch_bounds__(64)
msm_k_check_parts(const u64 *__restrict__ parts, u32 k, u32 *__restrict__ malformed) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    const u64 *r = parts + 24u * (size_t)j;
    bool ok = true;
    fp6 X, Y, Z;
#pragma unroll
    for (int c = 0; c < 6; c++) {
        X.c[c] = r[c];
        Y.c[c] = r[6 + c];
        Z.c[c] = r[12 + c];
        ok = ok && X.c[c] < FP_P && Y.c[c] < FP_P && Z.c[c] < FP_P;
    }
    sc256 lin;
#pragma unroll
    for (int c = 0; c < 4; c++) lin.w[c] = r[18 + c];
    ok = ok && !sc_geq_q(lin);
    if (ok && !f6_is_zero(Z)) {
        const fp6 z2 = f6_sqr(Z), z4 = f6_sqr(z2), z6 = f6_mul(z4, z2);
        fp6 b = f6_zero();
        b.c[0] = 395ull;
        b.c[1] = 1ull;
        const fp6 rhs = f6_add(f6_add(f6_mul(f6_sqr(X), X), f6_mul(X, z4)), f6_mul(b, z6));
        ok = f6_eq(f6_sqr(Y), rhs);
    }
    if (!ok) atomicOr(malformed, 1u);
}
}  // namespace ssa

// The shards added up on one device: one Jacobian addition per shard, the scalars mod q, [lin]G from the comb table and
// the x-only comparison (src/batch.rs:98-100,123-129) -- msm_k_finish with no doublings between its "windows".
static int msm_combine_records(ssa_ctx *ctx, const u64 *d_records, size_t k, uint32_t *d_verdict_out, u64 *d_partial_out,
                               bool check_points) {
    if (ctx->msm_comb_pts.reserve(18 * k * sizeof(u64)) || ctx->msm_comb_lins.reserve(4 * k * sizeof(u64)) ||
        ctx->msm_flags.reserve(64))
        return SSA_ERR_HIP;
    HIP_TRY(hipMemsetAsync(ctx->msm_flags.p, 0, 64, ctx->stream));
    hipLaunchKernelGGL(msm_k_unpack_parts, dim3(grid_for(k * 24, 256)), dim3(256), 0, ctx->stream, d_records, (u32)k,
                       (u64 *)ctx->msm_comb_pts.p, (u64 *)ctx->msm_comb_lins.p, (u32 *)ctx->msm_flags.p);
    HIP_TRY(hipGetLastError());
    if (check_points) {
        hipLaunchKernelGGL(msm_k_check_parts, dim3(grid_for(k, 64)), dim3(64), 0, ctx->stream, d_records, (u32)k,
                           (u32 *)ctx->msm_flags.p);
        HIP_TRY(hipGetLastError());
    }
    MsmShape sh;
    sh.c = 0;
    sh.windows = (u32)k;
    sh.buckets = 1;
    sh.chunks = 1;
    return timed_launch(ctx, "msm_combine", [&] {
        hipLaunchKernelGGL(msm_k_finish, dim3(1), dim3(128), 0, ctx->stream, (const u64 *)ctx->msm_comb_pts.p, sh,
                           (const u64 *)ctx->msm_comb_lins.p, (u32)k, (const u64 *)ctx->d_gtab,
                           (const u32 *)ctx->msm_flags.p, d_verdict_out, d_partial_out);
    });
}

extern "C" int ssa_msm_combine_device(ssa_ctx *ctx, const uint64_t *d_parts24, size_t k, uint32_t *d_verdict_out) {
    if (!ctx || !d_parts24 || !d_verdict_out || k == 0 || k > 4096) return SSA_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    return msm_combine_records(ctx, (const u64 *)d_parts24, k, d_verdict_out, nullptr, true);
}

extern "C" int ssa_msm_combine(ssa_ctx *ctx, const uint64_t *parts24, size_t k) {
    if (!ctx || !parts24 || k == 0 || k > 4096) return SSA_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    const void *d_parts;
    if (int rc = stage_up(ctx, ctx->st_aux, parts24, k * SSA_MSM_PARTIAL_WORDS * sizeof(uint64_t), &d_parts)) return rc;
    uint32_t *d_verdict = (uint32_t *)((char *)ctx->ws_fail.p + 32);
    if (int rc = ssa_msm_combine_device(ctx, (const uint64_t *)d_parts, k, d_verdict)) return rc;
    uint32_t v = SSA_MALFORMED;
    HIP_TRY(hipMemcpyAsync(&v, d_verdict, sizeof v, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return (int)v;
}

extern "C" int ssa_verify_batch_msm(ssa_ctx *ctx, const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf,
                                    const uint8_t *msgs, const uint64_t *msg_off, size_t msg_stride, size_t msg_len,
                                    size_t n, const uint8_t *coeffs) {
    if (!ctx || (n && (!sigs || !pks))) return SSA_ERR_ARG;
    if (int rc = check_msgs(msgs, msg_off, msg_stride, msg_len, n)) return rc;
    if (n == 0) return SSA_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    // Scalar::random(rng) (src/batch.rs:75-78): caller-supplied 32-byte scalars, or (coeffs == NULL) 128-bit
    // coefficients drawn on the device from a ChaCha20 stream keyed with getrandom(2)
    if (n >= ctx->pipeline_min_n && ctx->pipeline_chunks > 1) {
        // large batch: uploads pinned in place and chunked, the hashes (62 % of this form) run behind them
        PipelinedInputs pin;      // its destructor drains the side streams on every error return below
        bool used = false;
        if (pin.r_coeffs.pin(coeffs, coeffs ? n * 32 : 0)) {
            if (coeffs && ctx->st_coeffs.reserve(n * 32)) return SSA_ERR_HIP;
            if (int rc = pipelined_upload_hash(ctx, sigs, pks, pk_inf, msgs, msg_off, msg_stride, msg_len, n, pin, &used))
                return rc;
            if (used) {
                const void *pc = nullptr;
                if (coeffs) {
                    // behind the chunk copies on the copy stream; ctx->stream waits for this copy explicitly
                    HIP_TRY(hipMemcpyAsync(ctx->st_coeffs.p, coeffs, n * 32, hipMemcpyHostToDevice, ctx->copy_stream));
                    HIP_TRY(hipEventRecord(ctx->pipe_start, ctx->copy_stream));
                    HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->pipe_start, 0));
                    pc = ctx->st_coeffs.p;
                }
                uint32_t *d_verdict = (uint32_t *)((char *)ctx->ws_fail.p + 32);
                if (int rc = msm_run(ctx, pin.s.sigs, pin.s.pks, pin.s.inf, pin.s.msgs, pin.s.off, msg_stride, msg_len, n,
                                     (const u8 *)pc, 32, d_verdict, nullptr, true))
                    return rc;
                uint32_t v = SSA_MALFORMED;
                HIP_TRY(hipMemcpyAsync(&v, d_verdict, sizeof v, hipMemcpyDeviceToHost, ctx->stream));
                HIP_TRY(hipStreamSynchronize(ctx->stream));
                pin.done();
                return (int)v;
            }
        }
    }
    StagedInputs s;
    const void *p;
    if (int rc = stage_up(ctx, ctx->st_sigs, sigs, n * 81, &p)) return rc;
    s.sigs = (const u8 *)p;
    if (int rc = stage_up(ctx, ctx->st_pks, pks, n * 96, &p)) return rc;
    s.pks = (const u8 *)p;
    if (pk_inf) {
        if (int rc = stage_up(ctx, ctx->st_inf, pk_inf, n, &p)) return rc;
        s.inf = (const u8 *)p;
    }
    if (int rc = stage_msgs(ctx, msgs, msg_off, msg_stride, msg_len, n, s)) return rc;
    p = nullptr;
    if (coeffs) {
        if (int rc = stage_up(ctx, ctx->st_coeffs, coeffs, n * 32, &p)) return rc;
    }
    uint32_t *d_verdict = (uint32_t *)((char *)ctx->ws_fail.p + 32);
    if (int rc = ssa_verify_batch_msm_device(ctx, s.sigs, s.pks, s.inf, s.msgs, s.off, msg_stride, msg_len, n,
                                             (const u8 *)p, 32, d_verdict))
        return rc;
    uint32_t v = SSA_MALFORMED;
    HIP_TRY(hipMemcpyAsync(&v, d_verdict, sizeof v, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return (int)v;
}
