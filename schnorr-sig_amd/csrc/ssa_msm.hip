// MSM-form batch verification on the GPU: the algorithm the reference's verify_batch actually runs
// (src/batch.rs:31-130).
//
//     sum_i s_i R_i  -  sum_i (s_i h_i) P_i   ?=   [ sum_i s_i e_i ] G        (x-only comparison)
//
// with R_i decompressed from sig.x (src/batch.rs:104), h_i = hash_message scalars (:64-73), s_i the
// random coefficients (:75-78).  The 2n-point multi-scalar multiplication is a bucket method laid out
// for the GPU:
//   1. msm_k_prepare   per signature: decompress R, form the points R_i, -P_i and the scalars a_i = s_i,
//                      b_i = s_i h_i mod q -- written as SIGNED c-bit digits (|d| <= 2^(c-1): half the buckets
//                      of the unsigned form, a negative digit adds the negated point) --, block-reduce s_i e_i
//   2. grouping the (point, window) items by bucket -- the library's own two-level counting sort (round 4; hipCUB's
//      radix sort and its ~40 small launches are gone): the items of one window are consecutive, so a tile of 4096
//      items belongs to ONE window:
//        msm_k_hist     per tile (16384 items): LDS histogram of the HIGH 8 bits of the bucket number
//        msm_k_rowsum / msm_k_rowscan   exclusive scan of the (bin, tile) counts -> every tile's write positions
//        msm_k_scatter  per tile: items -> their (window, high-bits) group, ranks by LDS atomics (order inside a
//                       bucket means nothing to a sum: no stable pass is needed)
//        msm_k_group    per group (~4096 items, 128 buckets): LDS counting sort by the low 7 bits; the buckets'
//                       extents fall out of it (no keys are stored, no bounds pass, no global atomic anywhere)
//        msm_k_sizes / msm_k_size_ranks / msm_k_order   the buckets in global order of size, largest first: the 64
//                       buckets of a wave hold the same number of points (sizes are Poisson-distributed and a wave
//                       waits for its largest) and the launch ends on its shortest waves
//   3. msm_k_buckets   ONE BUCKET PER LANE: a lane adds up the points of its bucket with mixed
//                      additions (every exceptional case handled: equal public keys land in one bucket)
//   4. msm_k_chunks    running-sum trick on chunks of 8 buckets (short chains, 2^16 lanes);
//                      msm_k_tree sums the chunk sums of a window 16 at a time, one cooperating wave per sum
//                      (three launches); msm_k_finish is ONE cooperative
//                      block: wave 0 combines the windows by Horner's rule (the only long sequential chain
//                      of the method, ~240 doublings, wave-cooperative Fp6 arithmetic), wave 1 computes
//                      [lin]G from the comb table meanwhile; then the x coordinates are compared
// Panics of the reference (undecodable x, x not on the curve: src/batch.rs:67,104) give SSA_MALFORMED.
#define SSA_NO_KERNELS 1
#include "ssa_ctx.hpp"

#include <sys/random.h>

namespace ssa {

constexpr int MSM_CHUNK = 8;   // buckets per lane in the running-sum pass (short chains, many lanes)
// (the tree: one cooperating wave sums ctx->msm_tree_group = 16 chunk sums -- 15 additions of ~2.2 us --: 4096 -> 256 -> 16 -> 1)

struct MsmShape {
    u32 c;        // window bits
    u32 windows;  // ceil(255 / c) (+ 0: the top window of a 255-bit scalar has c - 1 bits, the last carry fits)
    u32 buckets;  // per window: 2^(c-1), for |digit| = 1 .. 2^(c-1) (signed digits; digit 0 contributes nothing)
    u32 chunks;   // per window
};

constexpr u32 MSM_TILE = 16384;         // items per tile of the grouping passes (256 threads x 64): a tile writes runs of
                                        // ~64 items (256 B) per group
constexpr u32 MSM_HI_BINS = 256;        // groups per window: the bucket number's bits above the low 7
constexpr u32 MSM_LO_BITS = 7, MSM_LO_BINS = 1u << MSM_LO_BITS;
constexpr u32 MSM_MAX_WINDOWS = 32;

// the (point, window) items in window-major order: window j holds the P points (n .. 2n-1) always and the R points
// (0 .. n-1) while j < wa (a coefficient of `coeff_bytes` bytes reaches only its lowest wa windows)
struct MsmItems {
    u32 n, wa, windows;
    u32 tile0[MSM_MAX_WINDOWS + 1];     // first tile of window j (tiles never straddle two windows)
    u32 base[MSM_MAX_WINDOWS + 1];      // first position of window j's region in the item arrays
};
SSA_DEV u32 items_of_window(const MsmItems &it, u32 j) { return j < it.wa ? 2u * it.n : it.n; }
// item k of window j -> point index
SSA_DEV u32 item_point(const MsmItems &it, u32 j, u32 k) { return j < it.wa ? k : it.n + k; }

SSA_DEV u32 sc_window(const u64 *__restrict__ k, u32 bit, u32 c) {
    const u32 wi = bit >> 6, sh = bit & 63u;
    if (wi > 3) return 0u;
    u64 v = k[wi] >> sh;
    if (sh + c > 64 && wi < 3) v |= k[wi + 1] << (64 - sh);
    return (u32)(v & ((1ull << c) - 1ull));
}

// Signed c-bit digits, window-major: digits[j * npts + pt] in [-2^(c-1), 2^(c-1) - 1] -- every window read as a
// two's-complement number with the carry of the window below:
//   raw = window_j(k) + carry;  raw >= 2^(c-1): digit = raw - 2^c, carry 1.
// Returns the carry out of the LAST window: the digits represent  k - carry_out * 2^(windows * c).  A scalar mod q
// (< 2^255: its top window has c - 1 bits and a value below 2^(c-1) - 1) never carries out; a narrow coefficient that
// fills its windows to the last bit may (see msm_k_prepare: the coefficient then IS the value its digits represent).
SSA_DEV u32 write_signed_digits(const sc256 &k, u32 c, u32 windows, short *__restrict__ digits, size_t npts, size_t pt) {
    const int half = 1 << (c - 1);
    int carry = 0;
#pragma unroll 1
    for (u32 j = 0; j < windows; j++) {
        int raw = (int)sc_window(k.w, j * c, c) + carry;
        carry = 0;
        if (raw >= half) {
            raw -= 2 * half;
            carry = 1;
        }
        digits[(size_t)j * npts + pt] = (short)raw;
    }
    return (u32)carry;
}

SSA_DEV void st_aff_row(u64 *__restrict__ row, const aff &p) {
    st_f6(row, p.x);
    st_f6(row + 6, p.y);
}

// ---- 1. points and scalars ---------------------------------------------------------------------
__global__ void __launch_bounds__(256)
msm_k_prepare(const u8 *__restrict__ sigs, const u8 *__restrict__ pks, const u8 *__restrict__ pk_inf,
              const u64 *__restrict__ h_in,
              const u8 *__restrict__ coeffs, u32 coeff_bytes, size_t n, MsmShape shp, u32 wa, u64 *__restrict__ points,
              short *__restrict__ digits, u64 *__restrict__ partials, u32 *__restrict__ malformed,
              u64 *__restrict__ s_out) {
    // h_in == nullptr (round 5): everything that does not need the challenge scalars -- the checks, R's square root,
    // the coefficient's digits, s_i e_i -- so that this kernel can run on a second stream UNDER ssa_k_hash (it fills the
    // hash kernel's tail and its own); the coefficient s_i is left in s_out for msm_k_prepare_h, which writes the digits
    // of s_i h_i once the hashes exist.
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
        // R_i carries the coefficient itself.  A 32-byte coefficient is taken mod q (Scalar::random, :75-78) and recoded
        // over all the windows.  A narrower one is recoded over ITS OWN wa windows only -- a carry window on top would
        // hold the digit 1 for half of the points: one enormous bucket -- so when its signed digits carry out of the
        // last window (a coefficient that fills its windows to the top bit, about half of the 128-bit ones) they
        // represent  raw - 2^(wa c), and THAT value is the coefficient: it multiplies R_i (through the digits), h_i and
        // e_i (through s below) alike, so the equation is the reference's with another, equally random, coefficient.
        if (coeff_bytes >= 32u) {
            s = sc_reduce256(s);
            (void)write_signed_digits(s, shp.c, wa, digits, 2 * n, i);
        } else if (write_signed_digits(s, shp.c, wa, digits, 2 * n, i)) {      // -(2^(wa c) - raw) mod q
            const u32 bits = wa * shp.c;        // (< 256: a narrow coefficient has fewer windows than a scalar)
            sc256 s_abs;
            u64 borrow = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const u32 lo_bit = 64u * (u32)k;
                const u64 pw = (bits >= lo_bit && bits < lo_bit + 64u) ? 1ull << (bits - lo_bit) : 0ull;
                const u64 d = pw - s.w[k];
                const u64 b1 = pw < s.w[k];
                const u64 d2 = d - borrow;
                const u64 b2 = d < borrow;
                s_abs.w[k] = d2;
                borrow = b1 | b2;
            }
            s = sc_neg_mod(s_abs);
        }
        if (ok) se = sc_mul_mod(s, e);                                 // s * e, :92-97
        P.y = f6_canon(f6_neg(P.y));                                   // k.0.neg(), :106
        st_aff_row(points + 12 * i, R);
        st_aff_row(points + 12 * (n + i), P);
        if (h_in) {
            sc256 h;
#pragma unroll
            for (int k = 0; k < 4; k++) h.w[k] = h_in[4 * i + k];
            const sc256 sh = sc_mul_mod(s, h);                         // hashes[i] *= scalars[i], :109-111
            (void)write_signed_digits(sh, shp.c, shp.windows, digits, 2 * n, n + i);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) s_out[4 * i + k] = s.w[k];
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

// the half of msm_k_prepare that needs the challenge scalars: the digits of s_i h_i for the point -P_i
__global__ void __launch_bounds__(256)
msm_k_prepare_h(const u64 *__restrict__ h_in, const u64 *__restrict__ s_in, size_t n, MsmShape shp,
                short *__restrict__ digits) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    sc256 s, h;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        s.w[k] = s_in[4 * i + k];
        h.w[k] = h_in[4 * i + k];
    }
    (void)write_signed_digits(sc_mul_mod(s, h), shp.c, shp.windows, digits, 2 * n, n + i);   // hashes[i] *= scalars[i], :109-111
}

// ---- 2. grouping the items by bucket ------------------------------------------------------------------
// bucket number of a digit: |d| - 1 in [0, 2^(c-1)); hi = its bits above the low 7 (the group), lo = the low 7
SSA_DEV bool tile_of_block(const MsmItems &it, u32 blk, u32 &j, u32 &k0, u32 &cnt) {
    j = 0;
#pragma unroll 1
    while (j + 1 < it.windows && blk >= it.tile0[j + 1]) j++;
    const u32 items = items_of_window(it, j);
    k0 = (blk - it.tile0[j]) * MSM_TILE;
    if (k0 >= items) return false;
    cnt = items - k0 < MSM_TILE ? items - k0 : MSM_TILE;
    return true;
}

// tile_hist[(j * 256 + bin) * tmax + tile] = items of this tile whose bucket lies in group `bin`
__global__ void __launch_bounds__(256)
msm_k_hist(const short *__restrict__ digits, MsmItems it, u32 tmax, u32 *__restrict__ tile_hist) {
    __shared__ u32 h[MSM_HI_BINS];
    h[threadIdx.x] = 0u;
    __syncthreads();
    u32 j, k0, cnt;
    if (!tile_of_block(it, blockIdx.x, j, k0, cnt)) return;        // (block-uniform)
    const short *dj = digits + (size_t)j * (2u * (size_t)it.n);
    for (u32 k = threadIdx.x; k < cnt; k += 256u) {
        const int d = dj[item_point(it, j, k0 + k)];
        if (d != 0) atomicAdd(&h[(u32)((d < 0 ? -d : d) - 1) >> MSM_LO_BITS], 1u);
    }
    __syncthreads();
    tile_hist[((size_t)j * MSM_HI_BINS + threadIdx.x) * tmax + (blockIdx.x - it.tile0[j])] = h[threadIdx.x];
}

// Exclusive scan of a window's (bin, tile) counts in bin-major order, in place, in two steps:
//   msm_k_rowsum  one wave per (window, bin) row: the row's total
//   msm_k_rowscan one block per row: base = the window's region + the totals of the bins before it, then the row's
//                 exclusive scan in place -> the position at which tile `tile` writes its first item of group `bin`;
//                 gstart[j * 257 + bin] = first position of the group, gstart[j * 257 + 256] = one past the window's last
__global__ void __launch_bounds__(64)
msm_k_rowsum(MsmItems it, u32 tmax, const u32 *__restrict__ tile_hist, u32 *__restrict__ rowsum) {
    const u32 row = blockIdx.x, j = row / MSM_HI_BINS;
    const u32 tiles = it.tile0[j + 1] - it.tile0[j];
    const u32 *th = tile_hist + (size_t)row * tmax;
    u32 sum = 0;
    for (u32 t = threadIdx.x; t < tiles; t += 64u) sum += th[t];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_down(sum, off, 64);
    if (threadIdx.x == 0) rowsum[row] = sum;
}

__global__ void __launch_bounds__(256)
msm_k_rowscan(MsmItems it, u32 tmax, const u32 *__restrict__ rowsum, u32 *__restrict__ tile_hist,
              u32 *__restrict__ gstart) {
    __shared__ u32 sh[256];
    const u32 row = blockIdx.x, j = row / MSM_HI_BINS, bin = row % MSM_HI_BINS;
    const u32 tiles = it.tile0[j + 1] - it.tile0[j];
    // totals of the bins before this one (and of all of them, for the sentinel)
    const u32 mine = rowsum[j * MSM_HI_BINS + threadIdx.x];
    sh[threadIdx.x] = mine;
    __syncthreads();
    for (u32 off = 1; off < 256u; off <<= 1) {
        const u32 v = threadIdx.x >= off ? sh[threadIdx.x - off] : 0u;
        __syncthreads();
        sh[threadIdx.x] += v;
        __syncthreads();
    }
    const u32 before = bin ? sh[bin - 1] : 0u, all = sh[255];
    __syncthreads();
    u32 run = it.base[j] + before;
    if (threadIdx.x == 0) {
        gstart[j * (MSM_HI_BINS + 1) + bin] = run;
        if (bin == MSM_HI_BINS - 1) gstart[j * (MSM_HI_BINS + 1) + MSM_HI_BINS] = it.base[j] + all;
    }
    u32 *th = tile_hist + (size_t)row * tmax;
    for (u32 t0 = 0; t0 < tiles; t0 += 256u) {           // the row in pieces of 256 tiles, the running total carried
        const u32 t = t0 + threadIdx.x;
        const u32 c = t < tiles ? th[t] : 0u;
        sh[threadIdx.x] = c;
        __syncthreads();
        for (u32 off = 1; off < 256u; off <<= 1) {
            const u32 v = threadIdx.x >= off ? sh[threadIdx.x - off] : 0u;
            __syncthreads();
            sh[threadIdx.x] += v;
            __syncthreads();
        }
        if (t < tiles) th[t] = run + sh[threadIdx.x] - c;
        const u32 piece = sh[255];
        __syncthreads();
        run += piece;
    }
}

// items -> their group, ONE word per item: v1[pos] = point index (24 bits: 2n <= 2^24 per slice) | low 7 bits of the
// bucket number << 24 | sign << 31
__global__ void __launch_bounds__(256)
msm_k_scatter(const short *__restrict__ digits, MsmItems it, u32 tmax, const u32 *__restrict__ tile_hist,
              u32 *__restrict__ v1) {
    __shared__ u32 cur[MSM_HI_BINS];
    u32 j, k0, cnt;
    if (!tile_of_block(it, blockIdx.x, j, k0, cnt)) return;
    cur[threadIdx.x] = tile_hist[((size_t)j * MSM_HI_BINS + threadIdx.x) * tmax + (blockIdx.x - it.tile0[j])];
    __syncthreads();
    const short *dj = digits + (size_t)j * (2u * (size_t)it.n);
    for (u32 k = threadIdx.x; k < cnt; k += 256u) {
        const u32 pt = item_point(it, j, k0 + k);
        const int d = dj[pt];
        if (d != 0) {
            const u32 b = (u32)((d < 0 ? -d : d) - 1);
            const u32 pos = atomicAdd(&cur[b >> MSM_LO_BITS], 1u);
            v1[pos] = pt | ((b & (MSM_LO_BINS - 1u)) << 24) | (d < 0 ? 0x80000000u : 0u);
        }
    }
}

// one block per (window, group): counting sort of the group's items by the low 7 bits of their bucket number; the
// 128 buckets' extents are the by-product: bstart[t], cnt[t] for t = j * buckets + group * 128 + lo
__global__ void __launch_bounds__(256)
msm_k_group(const u32 *__restrict__ gstart, const u32 *__restrict__ v1, MsmShape sh, u32 *__restrict__ v2,
            u32 *__restrict__ bstart, u32 *__restrict__ cnt) {
    __shared__ u32 h[MSM_LO_BINS], pre[MSM_LO_BINS];
    const u32 j = blockIdx.x / MSM_HI_BINS, bin = blockIdx.x % MSM_HI_BINS;
    if (bin * MSM_LO_BINS >= sh.buckets) return;                    // (narrow windows have fewer groups)
    const u32 gs = gstart[j * (MSM_HI_BINS + 1) + bin], ge = gstart[j * (MSM_HI_BINS + 1) + bin + 1];
    if (threadIdx.x < MSM_LO_BINS) h[threadIdx.x] = 0u;
    __syncthreads();
    for (u32 p = gs + threadIdx.x; p < ge; p += 256u) atomicAdd(&h[(v1[p] >> 24) & (MSM_LO_BINS - 1u)], 1u);
    __syncthreads();
    if (threadIdx.x < 64u) {          // exclusive scan of the 128 counts on one wave: two per lane
        const u32 a = h[2u * threadIdx.x], b = h[2u * threadIdx.x + 1u];
        u32 incl = a + b;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 v = __shfl_up(incl, off, 64);
            if ((int)threadIdx.x >= off) incl += v;
        }
        pre[2u * threadIdx.x] = incl - a - b;
        pre[2u * threadIdx.x + 1u] = incl - b;
    }
    __syncthreads();
    if (threadIdx.x < MSM_LO_BINS && bin * MSM_LO_BINS + threadIdx.x < sh.buckets) {
        const size_t t = (size_t)j * sh.buckets + bin * MSM_LO_BINS + threadIdx.x;
        bstart[t] = gs + pre[threadIdx.x];
        cnt[t] = h[threadIdx.x];
    }
    __syncthreads();
    for (u32 p = gs + threadIdx.x; p < ge; p += 256u) {
        const u32 v = v1[p];
        v2[gs + atomicAdd(&pre[(v >> 24) & (MSM_LO_BINS - 1u)], 1u)] = v & 0x80ffffffu;
    }
}

// Bucket sizes are Poisson-distributed (mean ~48 at n = 2^20) and a wave waits for its largest bucket: the lanes take
// the buckets in GLOBAL order of size, largest first (sizes clamped at 1023) -- the 64 buckets of a wave hold the same
// number of points, and the long waves start first, so the launch ends on its shortest ones (with four waves per
// slot at 2^20 signatures a long wave started late would idle most of the chip: measured 3.5 ms against 3.0 for the
// same additions when the order was only local).  Counting sort in three small launches; the only global atomics are
// one per (block, size class present in the block): ~60 per block.
__global__ void __launch_bounds__(1024)
msm_k_sizes(const u32 *__restrict__ cnt, u32 nb, u32 *__restrict__ ghist) {
    __shared__ u32 h[1024];
    const u32 t = blockIdx.x * 1024u + threadIdx.x;
    h[threadIdx.x] = 0u;
    __syncthreads();
    if (t < nb) atomicAdd(&h[cnt[t] < 1023u ? cnt[t] : 1023u], 1u);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&ghist[threadIdx.x], h[threadIdx.x]);
}
// gcur[c] = first rank of size class c when the classes are laid out from 1023 down to 0
__global__ void __launch_bounds__(1024)
msm_k_size_ranks(const u32 *__restrict__ ghist, u32 *__restrict__ gcur) {
    __shared__ u32 h[1024];
    const u32 mine = ghist[1023u - threadIdx.x];
    h[threadIdx.x] = mine;
    __syncthreads();
    for (u32 off = 1; off < 1024u; off <<= 1) {
        const u32 v = threadIdx.x >= off ? h[threadIdx.x - off] : 0u;
        __syncthreads();
        h[threadIdx.x] += v;
        __syncthreads();
    }
    gcur[1023u - threadIdx.x] = h[threadIdx.x] - mine;
}
__global__ void __launch_bounds__(1024)
msm_k_order(const u32 *__restrict__ cnt, u32 nb, u32 *__restrict__ gcur, u32 *__restrict__ order) {
    __shared__ u32 h[1024], base[1024];
    const u32 t = blockIdx.x * 1024u + threadIdx.x;
    h[threadIdx.x] = 0u;
    __syncthreads();
    const u32 c = t < nb ? (cnt[t] < 1023u ? cnt[t] : 1023u) : 0u;
    if (t < nb) atomicAdd(&h[c], 1u);
    __syncthreads();
    if (h[threadIdx.x]) base[threadIdx.x] = atomicAdd(&gcur[threadIdx.x], h[threadIdx.x]);   // this block's run of the class
    __syncthreads();
    h[threadIdx.x] = 0u;
    __syncthreads();
    if (t < nb) order[base[c] + atomicAdd(&h[c], 1u)] = t;
}

// ---- 3. one bucket per lane ---------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2)
msm_k_buckets(const u64 *__restrict__ points, const u32 *__restrict__ vals, const u32 *__restrict__ bstart,
              const u32 *__restrict__ cnt, const u32 *__restrict__ order, size_t nb, u64 *__restrict__ bsum) {
    const size_t lane_id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lane_id >= nb) return;
    const size_t t = order[lane_id];
    jac acc = jac_identity();
    const u32 lo = bstart[t], hi = lo + cnt[t];
#pragma unroll 1
    for (u32 p = lo; p < hi; p++) {
        const u32 item = vals[p];
        aff q = ld_aff(points + 12 * (size_t)(item & 0x7fffffffu));
        if (item >> 31) q.y = f6_neg(q.y);                // a negative digit adds the negated point ((0, 0) stays (0, 0))
        acc = jac_madd_fast(acc, q);      // asm block; identity / equal points fall back to the exact addition
    }
    st_jac(bsum + 18 * t, acc);
}

// ---- 4. bucket reduction ------------------------------------------------------------------------
// chunk of MSM_CHUNK buckets [k0, k0 + L) of a window, bucket k weighing k + 1 (|digit|):
//   sum_k (k + 1) B_k = sum_k (k - k0 + 1) B_k + k0 sum_k B_k
// (2^16 lanes are one wave per SIMD anyway: with one wave per SIMD allowed the general addition keeps its values in
//  registers -- 256 VGPRs + 384 B of scratch before)
__global__ void __launch_bounds__(256, 1)
msm_k_chunks(const u64 *__restrict__ bsum, MsmShape sh, u64 *__restrict__ chunk_out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)sh.windows * sh.chunks) return;
    const u32 j = (u32)(t / sh.chunks), ch = (u32)(t % sh.chunks);
    const u32 len = sh.buckets < (u32)MSM_CHUNK ? sh.buckets : (u32)MSM_CHUNK;
    const u32 k0 = ch * len;
    // the top bucket opens both sums (two additions to the identity saved per chunk)
    jac running = ld_jac(bsum + 18 * ((size_t)j * sh.buckets + (k0 + len - 1))), total = running;
#pragma unroll 1
    for (int k = (int)(k0 + len) - 2; k >= (int)k0; k--) {
        const jac b = ld_jac(bsum + 18 * ((size_t)j * sh.buckets + (u32)k));
        running = jac_add(running, b);
        total = jac_add(total, running);
    }
    // + [k0] running: double-and-add from the top set bit of the weight, the doublings through the ladder's
    // generated statement
    if (k0 > 0) {
        const u32 m = k0;
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

// tree step: out[j][g] = sum of `group` consecutive points of window j's `count` inputs.  ONE WAVE per output, the
// general additions on the wave-cooperative arithmetic (ssa_coop.hpp: ~2.2 us per addition, where a lone lane takes ~15):
// three launches of 16-way sums replace round 3's twelve pairwise ones (0.20 -> 0.1 ms).
__global__ void __launch_bounds__(64)
msm_k_tree(const u64 *__restrict__ in, u32 windows, u32 count, u32 group, u64 *__restrict__ out) {
    __shared__ CoopLds L;
    const u32 lane = threadIdx.x;
    const u32 groups = (count + group - 1) / group;
    const u32 t = blockIdx.x;
    if (t >= windows * groups) return;
    const u32 j = t / groups, g = t % groups;
    const u32 lo = g * group, hi = (lo + group < count) ? lo + group : count;    // (lo < hi: groups = ceil(count / group))
    int tt[9];
#pragma unroll
    for (int k = 0; k < 9; k++) tt[k] = 7 + k;
    // accumulator 0..3 (X, Y, Z, W = Z^4), addend 4..6
    auto load = [&](int s0, u32 k) {
        if (lane < 36) {
            const u32 v = lane / 12u, c = lane % 12u;
            const u64 w = in[18 * ((size_t)j * count + k) + 6u * v + c % 6u];
            L.slot[s0 + (int)v][c] = c < 6 ? w : fp_mul_small(w, 7u);
        }
        coop_sync();
    };
    load(0, lo);
    coop_mul(L, 3, 2, 2, lane, 0);
    coop_mul(L, 3, 3, 3, lane, 0);
#pragma unroll 1
    for (u32 k = lo + 1; k < hi; k++) {
        load(4, k);
        coop_jac_add(L, 0, 4, tt, lane, 0);
    }
    if (lane < 18) out[18 * (size_t)t + lane] = fp_canon(L.slot[(int)(lane / 6u)][lane % 6u]);
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
        const GtabGeom gg = gtab_geom(gtab);
#pragma unroll 1
        for (u32 w = 0; w < (partial_out ? 0u : gg.count); w++) {     // BASEPOINT_TABLE.multiply_vartime
            const u32 d = sc_bits(lin, w * gg.bits, gg.bits);
            if (d != 0) {
                const u64 *rowp = gtab + (((size_t)w << gg.bits) + d) * 12;
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
        // The record carries the point in its CANONICAL form: affine (x, y, 1), or (0, 0, 0) for the identity.  The
        // grouping passes rank the items of a bucket by LDS atomics, so the order of the additions -- and with it
        // the Jacobian representative -- differs from run to run; the point does not, and equal shards give equal
        // bytes (records are compared, committed as fixtures and cross process boundaries).
        if (ws == 0) {
            const bool inf = coop_is_zero(L, 2, lane, ws);
            if (!inf) {
                coop_inv(L, 7, 2, 8, 9, 10, lane, ws);     // 1 / Z
                coop_mul(L, 8, 7, 7, lane, ws);            // 1 / Z^2
                coop_mul(L, 0, 0, 8, lane, ws);            // x
                coop_mul(L, 8, 8, 7, lane, ws);            // 1 / Z^3
                coop_mul(L, 1, 1, 8, lane, ws);            // y
            }
            if (lane < 18) {
                const u32 v = lane / 6u, c = lane % 6u;
                u64 w = v == 2 ? (c == 0 ? 1ull : 0ull) : fp_canon(L.slot[(int)v][c]);
                partial_out[lane] = inf ? 0ull : w;
            }
        }
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
    // c = 14 put n/4 points into single lanes and took 0.9 s at n = 2^18).  Signed digits (round 4): 2^(c-1)
    // buckets per window, and the top window's c - 1 bits absorb the last carry.
    MsmShape sh;
    sh.c = n >= 4096 ? 16u : 8u;
    sh.windows = (255 + sh.c - 1) / sh.c;
    sh.buckets = 1u << (sh.c - 1);
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
    if (n > (1ull << 23)) return SSA_ERR_ARG;   // an item carries its point index in 24 bits: 2n <= 2^24 (msm_run slices)
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
    // windows the coefficients themselves can reach (32-byte ones are reduced mod q: all of them).  A narrower coefficient
    // whose width is a whole number of windows is read as a TWO'S-COMPLEMENT integer (msm_k_prepare): its signed digits
    // then need no carry window -- which would hold the digit 1 for half of the points, one enormous bucket -- and a
    // random coefficient is as good signed as unsigned (the same value multiplies R_i, h_i and e_i).
    const u32 wa = coeff_bytes >= 32 ? sh.windows : (8u * coeff_bytes + sh.c - 1) / sh.c;
    MsmItems it;
    it.n = (u32)n;
    it.wa = wa;
    it.windows = sh.windows;
    it.tile0[0] = it.base[0] = 0;
    for (u32 j = 0; j < sh.windows; j++) {
        const size_t items = j < wa ? 2 * n : n;
        it.tile0[j + 1] = it.tile0[j] + (u32)((items + MSM_TILE - 1) / MSM_TILE);
        it.base[j + 1] = it.base[j] + (u32)items;
    }
    const u32 n_tiles = it.tile0[sh.windows], tmax = (u32)((2 * n + MSM_TILE - 1) / MSM_TILE);
    const size_t npts = 2 * n, total = it.base[sh.windows], nb = (size_t)sh.windows * sh.buckets;
    const unsigned n_blocks = grid_for(n, 256);
    if ((!d_h && ctx->ws_h.reserve(n * 32)) || ctx->msm_points.reserve(npts * 96) ||
        ctx->msm_scalars.reserve(npts * sh.windows * sizeof(short)) ||                 // signed digits, window-major
        ctx->msm_keys.reserve((size_t)sh.windows * MSM_HI_BINS * tmax * 4) ||          // per-tile group counts / positions
        ctx->msm_vals.reserve(total * 4) ||                                            // items by group: point | low bits | sign
        ctx->msm_keys2.reserve((size_t)sh.windows * MSM_HI_BINS * 4) ||                // row totals of the scan
        ctx->msm_ids.reserve(2048 * 4) ||                                              // size classes: counts, first ranks
        ctx->msm_vals2.reserve(total * 4) ||                                           // items by bucket
        ctx->msm_bounds.reserve(nb * 4) || ctx->msm_cnt.reserve(nb * 4) ||             // bucket extents
        ctx->msm_ids2.reserve((nb + 1024) * 4) ||                                      // buckets in order of size
        ctx->msm_cnt2.reserve((size_t)sh.windows * (MSM_HI_BINS + 1) * 4) ||           // group extents
        ctx->msm_buckets.reserve(nb * 144) ||
        ctx->msm_chunks.reserve((size_t)sh.windows * sh.chunks * 144) ||
        ctx->msm_windows.reserve((size_t)sh.windows * sh.chunks * 144) ||
        ctx->msm_partials.reserve((size_t)n_blocks * 32) || ctx->msm_flags.reserve(64))
        return SSA_ERR_HIP;
    HIP_TRY(hipMemsetAsync(ctx->msm_flags.p, 0, 64, ctx->stream));
    int rc = 0;
    if (!d_h && ctx->msm_overlap) {
        // The challenge hashes are 62 % of this form and nothing but the digits of s_i h_i needs them: the rest of the
        // preparation (R's square roots above all) runs on a second stream UNDER ssa_k_hash -- its waves fill the hash
        // kernel's tail (the last wave of every SIMD alone, 0.35 ms) and the hash fills theirs --, ctx->stream joins it
        // after the hash and writes the digits that were missing.
        if (ctx->msm_sbuf.reserve(n * 32)) return SSA_ERR_HIP;
        hipStream_t side = ctx->hash_stream[0];
        HIP_TRY(hipEventRecord(ctx->pipe_start, ctx->stream));
        HIP_TRY(hipStreamWaitEvent(side, ctx->pipe_start, 0));
        hipLaunchKernelGGL(msm_k_prepare, dim3(n_blocks), dim3(256), 0, side, d_sigs, d_pks, d_pk_inf,
                           (const u64 *)nullptr, d_coeffs, coeff_bytes, n, sh, wa, (u64 *)ctx->msm_points.p,
                           (short *)ctx->msm_scalars.p, (u64 *)ctx->msm_partials.p, (u32 *)ctx->msm_flags.p,
                           (u64 *)ctx->msm_sbuf.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ctx->hash_done[0], side));
        if (int hrc = ssa_internal_hash_scalars(ctx, d_sigs, d_pks, d_msgs, d_msg_off, msg_stride, msg_len, n)) {
            (void)hipStreamSynchronize(side);
            return hrc;
        }
        d_h = (const u64 *)ctx->ws_h.p;
        HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->hash_done[0], 0));
        rc = timed_launch(ctx, "msm_k_prepare", [&] {
            hipLaunchKernelGGL(msm_k_prepare_h, dim3(n_blocks), dim3(256), 0, ctx->stream, d_h, (const u64 *)ctx->msm_sbuf.p,
                               n, sh, (short *)ctx->msm_scalars.p);
        });
        if (rc) return rc;
    } else {
        // challenge scalars h_i with the kernel of the per-lane path
        if (!d_h) {
            if (int hrc = ssa_internal_hash_scalars(ctx, d_sigs, d_pks, d_msgs, d_msg_off, msg_stride, msg_len, n)) return hrc;
            d_h = (const u64 *)ctx->ws_h.p;
        }
        rc = timed_launch(ctx, "msm_k_prepare", [&] {
            hipLaunchKernelGGL(msm_k_prepare, dim3(n_blocks), dim3(256), 0, ctx->stream, d_sigs, d_pks, d_pk_inf,
                               d_h, d_coeffs, coeff_bytes, n, sh, wa, (u64 *)ctx->msm_points.p,
                               (short *)ctx->msm_scalars.p, (u64 *)ctx->msm_partials.p, (u32 *)ctx->msm_flags.p,
                               (u64 *)nullptr);
        });
        if (rc) return rc;
    }
    rc = timed_launch(ctx, "msm_sort", [&] {
        hipLaunchKernelGGL(msm_k_hist, dim3(n_tiles), dim3(256), 0, ctx->stream, (const short *)ctx->msm_scalars.p, it, tmax,
                           (u32 *)ctx->msm_keys.p);
        hipLaunchKernelGGL(msm_k_rowsum, dim3(sh.windows * MSM_HI_BINS), dim3(64), 0, ctx->stream, it, tmax,
                           (const u32 *)ctx->msm_keys.p, (u32 *)ctx->msm_keys2.p);
        hipLaunchKernelGGL(msm_k_rowscan, dim3(sh.windows * MSM_HI_BINS), dim3(256), 0, ctx->stream, it, tmax,
                           (const u32 *)ctx->msm_keys2.p, (u32 *)ctx->msm_keys.p, (u32 *)ctx->msm_cnt2.p);
        hipLaunchKernelGGL(msm_k_scatter, dim3(n_tiles), dim3(256), 0, ctx->stream, (const short *)ctx->msm_scalars.p, it,
                           tmax, (const u32 *)ctx->msm_keys.p, (u32 *)ctx->msm_vals.p);
        hipLaunchKernelGGL(msm_k_group, dim3(sh.windows * MSM_HI_BINS), dim3(256), 0, ctx->stream,
                           (const u32 *)ctx->msm_cnt2.p, (const u32 *)ctx->msm_vals.p, sh, (u32 *)ctx->msm_vals2.p,
                           (u32 *)ctx->msm_bounds.p, (u32 *)ctx->msm_cnt.p);
    });
    if (rc) return rc;
    rc = timed_launch(ctx, "msm_k_buckets", [&] {
        u32 *ghist = (u32 *)ctx->msm_ids.p, *gcur = ghist + 1024;
        (void)hipMemsetAsync(ghist, 0, 1024 * 4, ctx->stream);
        hipLaunchKernelGGL(msm_k_sizes, dim3(grid_for(nb, 1024)), dim3(1024), 0, ctx->stream, (const u32 *)ctx->msm_cnt.p,
                           (u32)nb, ghist);
        hipLaunchKernelGGL(msm_k_size_ranks, dim3(1), dim3(1024), 0, ctx->stream, (const u32 *)ghist, gcur);
        hipLaunchKernelGGL(msm_k_order, dim3(grid_for(nb, 1024)), dim3(1024), 0, ctx->stream, (const u32 *)ctx->msm_cnt.p,
                           (u32)nb, gcur, (u32 *)ctx->msm_ids2.p);
        hipLaunchKernelGGL(msm_k_buckets, dim3(grid_for(nb, 256)), dim3(256), 0, ctx->stream,
                           (const u64 *)ctx->msm_points.p, (const u32 *)ctx->msm_vals2.p,
                           (const u32 *)ctx->msm_bounds.p, (const u32 *)ctx->msm_cnt.p, (const u32 *)ctx->msm_ids2.p, nb,
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
            const u32 groups = (count + ctx->msm_tree_group - 1) / ctx->msm_tree_group;
            hipLaunchKernelGGL(msm_k_tree, dim3(sh.windows * groups), dim3(64), 0, ctx->stream,
                               (const u64 *)ping, sh.windows, count, ctx->msm_tree_group, pong);
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

static int msm_host_one(ssa_ctx *ctx, const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf, const uint8_t *msgs,
                        const uint64_t *msg_off, size_t msg_stride, size_t msg_len, size_t n, const uint8_t *coeffs,
                        int *verdict_out, uint64_t *out24);
static int msm_host_sliced(ssa_ctx *ctx, const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf, const uint8_t *msgs,
                           const uint64_t *msg_off, size_t msg_stride, size_t msg_len, size_t n, const uint8_t *coeffs,
                           int *verdict_out, uint64_t *out24);

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
    if (n > ctx->msm_slice)
        return msm_host_sliced(ctx, sigs, pks, pk_inf, msgs, msg_off, msg_stride, msg_len, n, coeffs, nullptr, out24);
    return msm_host_one(ctx, sigs, pks, pk_inf, msgs, msg_off, msg_stride, msg_len, n, coeffs, nullptr, out24);
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
    int verdict = SSA_MALFORMED;
    const int rc = n > ctx->msm_slice
                       ? msm_host_sliced(ctx, sigs, pks, pk_inf, msgs, msg_off, msg_stride, msg_len, n, coeffs, &verdict, nullptr)
                       : msm_host_one(ctx, sigs, pks, pk_inf, msgs, msg_off, msg_stride, msg_len, n, coeffs, &verdict, nullptr);
    return rc ? rc : verdict;
}

// A host batch of more than one MSM slice in bounded device memory (round 5): slice after slice through the one-slice
// form -- staging sized for a slice, only the slice in flight pinned, on the context and its twin alternately -- each
// reduced to its 24-word record in HOST memory, and the records combined like the shards of a multi-GPU batch
// (src/batch.rs:98-129: one point and one scalar per part): a verdict, or the whole batch's own record.
static int msm_host_sliced(ssa_ctx *ctx, const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf, const uint8_t *msgs,
                           const uint64_t *msg_off, size_t msg_stride, size_t msg_len, size_t n, const uint8_t *coeffs,
                           int *verdict_out, uint64_t *out24) {
    if (msg_off)
        for (size_t i = 0; i < n; i++)
            if (msg_off[i + 1] < msg_off[i] || msg_off[i + 1] - msg_off[i] > 0xffffffffull) return SSA_ERR_ARG;
    const size_t slice = ctx->msm_slice, k = (n + slice - 1) / slice;
    if (k > 4096) return SSA_ERR_ARG;
    std::vector<uint64_t> recs(k * SSA_MSM_PARTIAL_WORDS, 0);
    int rc = run_host_slices(ctx, n, slice, [&](ssa_ctx *c, size_t lo, size_t cnt) {
        const HostMsgSlice ms(msgs, msg_off, msg_stride, lo, cnt);
        return msm_host_one(c, sigs + 81 * lo, pks + 96 * lo, pk_inf ? pk_inf + lo : nullptr, ms.msgs, ms.offp, msg_stride,
                            msg_len, cnt, coeffs ? coeffs + 32 * lo : nullptr, nullptr,
                            recs.data() + (lo / slice) * SSA_MSM_PARTIAL_WORDS);
    });
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    const void *d_parts;
    if (int r = stage_up(ctx, ctx->st_aux, recs.data(), recs.size() * sizeof(uint64_t), &d_parts)) return r;
    uint32_t *d_verdict = (uint32_t *)((char *)ctx->ws_fail.p + 32);
    if (out24 && ctx->st_aux2.reserve(SSA_MSM_PARTIAL_WORDS * sizeof(u64))) return SSA_ERR_HIP;
    if (int r = msm_combine_records(ctx, (const u64 *)d_parts, k, out24 ? nullptr : d_verdict,
                                    out24 ? (u64 *)ctx->st_aux2.p : nullptr, true))
        return r;
    uint32_t v = SSA_MALFORMED;
    if (out24)
        HIP_TRY(hipMemcpyAsync(out24, ctx->st_aux2.p, SSA_MSM_PARTIAL_WORDS * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    else
        HIP_TRY(hipMemcpyAsync(&v, d_verdict, sizeof v, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (verdict_out) *verdict_out = (int)v;
    return 0;
}

// ONE slice (n <= ctx->msm_slice) from host buffers: the verdict (out24 == nullptr) or the slice's 24-word record
static int msm_host_one(ssa_ctx *ctx, const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf, const uint8_t *msgs,
                        const uint64_t *msg_off, size_t msg_stride, size_t msg_len, size_t n, const uint8_t *coeffs,
                        int *verdict_out, uint64_t *out24) {
    HIP_TRY(hipSetDevice(ctx->device));
    uint32_t *d_verdict = out24 ? nullptr : (uint32_t *)((char *)ctx->ws_fail.p + 32);
    u64 *d_rec = nullptr;
    if (out24) {
        if (ctx->st_aux2.reserve(SSA_MSM_PARTIAL_WORDS * sizeof(u64))) return SSA_ERR_HIP;
        d_rec = (u64 *)ctx->st_aux2.p;
    }
    auto fetch = [&]() -> int {       // the result of the launches queued on ctx->stream
        uint32_t v = SSA_MALFORMED;
        if (out24)
            HIP_TRY(hipMemcpyAsync(out24, d_rec, SSA_MSM_PARTIAL_WORDS * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
        else
            HIP_TRY(hipMemcpyAsync(&v, d_verdict, sizeof v, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (verdict_out) *verdict_out = (int)v;
        return 0;
    };
    // Scalar::random(rng) (src/batch.rs:75-78): caller-supplied 32-byte scalars, or (coeffs == NULL) 128-bit
    // coefficients drawn on the device from a ChaCha20 stream keyed with getrandom(2)
    if (n >= ctx->pipeline_min_n && ctx->pipeline_chunks > 1) {
        // large batch: uploads pinned in place and chunked, the hashes (62 % of this form) run behind them
        PipelinedInputs pin;      // its destructor drains the side streams on every error return below
        bool used = false;
        if (!coeffs || ctx->pin_coeffs.reserve(n * 32) == 0) {
            if (coeffs && ctx->st_coeffs.reserve(n * 32)) return SSA_ERR_HIP;
            if (int rc = pipelined_upload_hash(ctx, sigs, pks, pk_inf, msgs, msg_off, msg_stride, msg_len, n, pin, &used))
                return rc;
            if (used) {
                const void *pc = nullptr;
                if (coeffs) {
                    // behind the chunk copies on the copy stream; ctx->stream waits for this copy explicitly
                    host_copy(ctx->pin_coeffs.p, coeffs, n * 32);
                    HIP_TRY(hipMemcpyAsync(ctx->st_coeffs.p, ctx->pin_coeffs.p, n * 32, hipMemcpyHostToDevice, ctx->copy_stream));
                    HIP_TRY(hipEventRecord(ctx->pipe_start, ctx->copy_stream));
                    HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->pipe_start, 0));
                    pc = ctx->st_coeffs.p;
                }
                if (int rc = msm_run(ctx, pin.s.sigs, pin.s.pks, pin.s.inf, pin.s.msgs, pin.s.off, msg_stride, msg_len, n,
                                     (const u8 *)pc, 32, d_verdict, d_rec, true))
                    return rc;
                if (int rc = fetch()) return rc;
                pin.done();
                return 0;
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
    if (int rc = msm_run(ctx, s.sigs, s.pks, s.inf, s.msgs, s.off, msg_stride, msg_len, n, (const u8 *)p, 32, d_verdict,
                         d_rec, false))
        return rc;
    return fetch();
}
