// Wave-cooperative arithmetic: ONE WAVE works on ONE point / ONE signature.
//
// The per-lane kernels (ssa_kernels.hpp) need ~65 k lanes to fill the chip and take 5-8 ms per call however
// small the batch is, because a lane runs its ~330 point operations serially.  Here a wave shares the work of
// one point operation: Fp6 values live in LDS slots; the independent products of a formula run side by side
// (twelve lanes per product, one LDS round trip per round) with the formula's additions folded into the tail
// of the round that produces their operands; a lone product (inversions, comparisons) is spread over 36 lanes.
// Used for (1) the sequential tail of the MSM reduction (ssa_msm.hip) and (2) the low-latency verification
// kernel ssa_k_verify_coop (small batches, single Signature::verify calls).
//
// Every function here must be called by ALL 64 lanes of the wave that owns the working set; Fp6 values are
// LDS slots addressed by index, booleans returned are wave-uniform.
#pragma once
// (included at the end of ssa_kernels.hpp: uses its byte loaders, msg_felt and status codes)

namespace ssa {

constexpr int COOP_SLOTS = 105;     // 71 of the verification kernel + a second point and its 32-slot table (small-batch MSM form)

// A slot holds an Fp6 value v (words 0..5) and 7*v (words 6..11): the product rounds read the wrapped
// terms (u^6 = 7) of their second operand from the upper half, so nobody scales on the critical path.
// The generic primitives keep the upper half valid (coop_store7 / coop_put); the hand-scheduled point
// operations skip it for values that are only ever first operands or tail operands (noted where they do).
struct CoopLds {
    u64 slot[COOP_SLOTS][12];
    u64 part[2][6][6][3];     // coop_mul: per cooperating wave, products grouped by output coefficient
    u64 st[2][12];            // Rescue state planes
};

#define COOP_FN __device__ __forceinline__

// Ordering point between LDS writes and reads of different lanes.  The code here runs on ONE wave per
// working set, so no s_barrier is needed: LDS executes a wave's instructions in order; the fence keeps the
// compiler from moving accesses across it and waits for outstanding LDS operations.
COOP_FN void coop_sync() {
#ifdef SSA_COOP_USE_BARRIER
    __syncthreads();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// lanes 0..11 store v (lanes 0..5: coefficient `lane`) and 7 v (lanes 6..11) of a value given per coefficient
COOP_FN void coop_store7(CoopLds &L, int dst, u64 coef, u32 lane) {
    if (lane < 12) L.slot[dst][lane] = lane < 6 ? coef : fp_mul_small(coef, 7u);
}

// One product on 36 lanes (one 64x64 product each), six lanes add up the columns and reduce: the lowest
// latency for a lone product (two LDS round trips).
COOP_FN void coop_mul(CoopLds &L, int dst, int a, int b, u32 lane, int ws = 0) {
    if (lane < 36) {
        const u32 i = lane / 6, j = lane % 6;
        u32 k = i + j;
        u64 lo, hi;
        if (k >= 6) {
            k -= 6;
            mul64x64(L.slot[a][i], L.slot[b][6 + j], lo, hi);
        } else {
            mul64x64(L.slot[a][i], L.slot[b][j], lo, hi);
        }
        L.part[ws][k][i][0] = lo;
        L.part[ws][k][i][1] = hi;
    }
    coop_sync();
    // 6 lanes: column sums (6 * 2^128 < 2^131) and one Goldilocks reduction each
    if (lane < 6) {
        u64 lo = 0, hi = 0, top = 0;
#pragma unroll
        for (int t = 0; t < 6; t++) {
            const u64 plo = L.part[ws][lane][t][0], phi = L.part[ws][lane][t][1];
            const u64 nlo = lo + plo;
            const u64 c0 = nlo < plo;
            const u64 nh1 = hi + phi;
            const u64 c1 = nh1 < phi;
            const u64 nh2 = nh1 + c0;
            const u64 c2 = nh2 < c0;
            lo = nlo;
            hi = nh2;
            top += c1 + c2;
        }
        const u64 r = fp_reduce_parts(lo, lo32(hi), (u64)hi32(hi) + (top << 32));
        L.slot[dst][lane] = r;
        L.slot[dst][6 + lane] = fp_mul_small(r, 7u);
    }
    coop_sync();
}

// linear steps: the same operation on both halves of a slot (7 (a + b) = 7a + 7b)
COOP_FN void coop_add(CoopLds &L, int dst, int a, int b, u32 lane, int ws = 0) {
    if (lane < 12) L.slot[dst][lane] = fp_add(L.slot[a][lane], L.slot[b][lane]);
    coop_sync();
}
COOP_FN void coop_sub(CoopLds &L, int dst, int a, int b, u32 lane, int ws = 0) {
    if (lane < 12) L.slot[dst][lane] = fp_sub(L.slot[a][lane], L.slot[b][lane]);
    coop_sync();
}
COOP_FN void coop_neg(CoopLds &L, int dst, int a, u32 lane, int ws = 0) {
    if (lane < 12) L.slot[dst][lane] = fp_neg(L.slot[a][lane]);
    coop_sync();
}
COOP_FN void coop_copy(CoopLds &L, int dst, int a, u32 lane, int ws = 0) {
    if (lane < 12) L.slot[dst][lane] = L.slot[a][lane];
    coop_sync();
}
COOP_FN void coop_set(CoopLds &L, int dst, u64 c0, u32 lane, int ws = 0) {  // dst = c0 (element of Fp)
    if (lane < 12) L.slot[dst][lane] = lane == 0 ? c0 : (lane == 6 ? fp_mul_small(c0, 7u) : 0ull);
    coop_sync();
}
COOP_FN void coop_mul_fp(CoopLds &L, int dst, int a, u64 s, u32 lane, int ws = 0) {
    if (lane < 12) L.slot[dst][lane] = fp_mul(L.slot[a][lane], s);
    coop_sync();
}
COOP_FN bool coop_is_zero(CoopLds &L, int a, u32 lane, int ws = 0) {
    const bool z = lane < 6 ? fp_is_zero(L.slot[a][lane]) : true;
    return __all(z);
}
COOP_FN bool coop_eq(CoopLds &L, int a, int b, u32 lane, int ws = 0) {
    const bool e = lane < 6 ? fp_eq(L.slot[a][lane], L.slot[b][lane]) : true;
    return __all(e);
}

// dst = a^-1 through the norm to Fp (a != 0); uses scratch slots t0, t1, t2 (all distinct from a)
COOP_FN void coop_inv(CoopLds &L, int dst, int a, int t0, int t1, int t2, u32 lane, int ws = 0) {
    // t0 = frob_1(a) * frob_2(a) * ... * frob_5(a)
    if (lane < 12) {
        const int c = (int)(lane % 6u);
        L.slot[t0][lane] = fp_mul_gpow(L.slot[a][lane], c * 1);
        L.slot[t1][lane] = fp_mul_gpow(L.slot[a][lane], c * 2);
    }
    coop_sync();
    coop_mul(L, t0, t0, t1, lane, ws);
#pragma unroll 1
    for (int k = 3; k <= 5; k++) {
        if (lane < 12) L.slot[t1][lane] = fp_mul_gpow(L.slot[a][lane], (int)(lane % 6u) * k);
        coop_sync();
        coop_mul(L, t0, t0, t1, lane, ws);
    }
    coop_mul(L, t2, a, t0, lane, ws);                 // norm: only c0 is non-zero
    const u64 ninv = fp_inv(L.slot[t2][0]);       // every lane computes the same Fp inverse
    coop_mul_fp(L, dst, t0, ninv, lane, ws);
}

// Building blocks of a hand-scheduled round: lane k of a six-lane group computes coefficient k of A * B
// (lazy accumulation, one reduction -- the arithmetic of the per-lane kernels), then the group applies the
// additions that follow the product in the formula and stores the value -- with its 7x half (coop_put) when
// the value will be the SECOND operand of a later product, which is the one read through the wrapped terms.
// COOP_LPC lanes share a coefficient (terms i = q, q + COOP_LPC, ...): shorter multiply chains per lane; the
// partial results meet through a DPP swap of neighbouring lanes.  Lane layout: lane = COOP_LPC * (6 g + k) + q.
#ifndef COOP_LPC
#define COOP_LPC 2
#endif
COOP_FN u64 coop_swap_neighbour(u64 v) {   // value of lane ^ 1 (quad_perm [1, 0, 3, 2])
    const int lo = __builtin_amdgcn_update_dpp(0, (int)lo32(v), 0xB1, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)hi32(v), 0xB1, 0xf, 0xf, false);
    return mk64((u32)lo, (u32)hi);
}
COOP_FN u64 coop_group_mul(const u64 *A, const u64 *B, u32 k, u32 q) {
    constexpr int NT = 6 / COOP_LPC;
    u64 x[NT], y[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) {
        const int i = (int)q + COOP_LPC * j;
        int idx = (int)k - i;          // b_{k-i}, or 7 b_{k-i+6} from the upper half
        if (idx < 0) idx += 12;
        x[j] = A[i];
        y[j] = B[idx];
    }
#if defined(SSA_F6_ASM) && COOP_LPC == 2 && !defined(SSA_NO_ACC3_ASM)
    // round 4: the lane's three products and their reduction as one generated block (fp6_asm.inc fp_acc3_core_asm: 32
    // instructions + 7 padded wait states; the compiled fp_acc sequence is ~58) -- this is the critical path of every round
    u64 r = fp_acc3_core_asm(x, y);
#else
    fp_acc acc;
    acc_init(acc, x[0], y[0]);
#pragma unroll
    for (int j = 1; j < NT; j++) acc_mac(acc, x[j], y[j]);
    u64 r = acc_reduce(acc);
#endif
#if COOP_LPC == 2
    r = fp_add(r, coop_swap_neighbour(r));
#endif
    return r;
}
// lane -> (group g, coefficient k, sub-lane q)
#define COOP_LANE_ROLES(lane)                                              \
    const u32 q = (lane) % (u32)COOP_LPC, hl_ = (lane) / (u32)COOP_LPC;    \
    const u32 g = hl_ / 6u, k = hl_ - 6u * g
// lane holding coefficient k of group g (source of a cross-group shuffle)
COOP_FN int coop_lane_of(u32 g, u32 k) { return (int)((6u * g + k) * (u32)COOP_LPC); }
COOP_FN void coop_put(CoopLds &L, int dst, u32 k, u64 r) {
    L.slot[dst][k] = r;
    L.slot[dst][6 + k] = fp_mul_small(r, 7u);
}
COOP_FN int coop_pick(u32 g, int s0, int s1, int s2 = 0, int s3 = 0) {
    return g == 0 ? s0 : (g == 1 ? s1 : (g == 2 ? s2 : s3));
}

// A cooperative point is FOUR consecutive slots P .. P+3 = (X, Y, Z, W) in modified Jacobian coordinates,
// W = Z^4 (the curve's a is 1, so the doubling needs a Z^4 = W): carrying W takes one level off the doubling's
// dependency chain -- three rounds instead of four -- and costs the additions nothing, their last two rounds
// have idle lane groups for Z3^2 and Z3^4.  The identity is Z = 0 (W = 0).
//
// doubling, three rounds (t[0..3] scratch):
//   R1  X^2 -> M = 3 X^2 + W | YY = Y^2 | Y Z -> Z3 = 2 Y Z
//   R2  YY^2 -> E = 8 Y^4 | X YY -> S = 4 X YY | M^2 -> X3 = M^2 - 2 S, D = S - X3
//       (+ with `fuse_qy >= 0`: Z3^2 and QY Z3, the first round of the mixed addition that follows)
//   R3  D M -> Y3 = D M - E | E W -> W3 = 2 E W
// Within a round the product part (uniform code) reads, the per-group tails write: no tail reads a slot that
// another group's tail writes.  coop_put stores value and 7x value; a plain store is used where the value is
// only ever a first operand or a tail operand.
COOP_FN void coop_jac_dbl(CoopLds &L, int P, const int *t, u32 lane, int ws = 0, int fuse_qy = -1,
                          int fuse_z1z1 = 0, int fuse_qyz = 0) {
    const int X = P, Y = P + 1, Z = P + 2, W = P + 3;
    const int M = t[0], YY = t[1], E = t[2], D = t[3];
    COOP_LANE_ROLES(lane);
    (void)q;
    u64 r = 0;
    u64 pre = L.slot[W][k];     // tail operands are fetched before the products: their LDS latency hides
    if (g < 3) r = coop_group_mul(L.slot[coop_pick(g, X, Y, Y)], L.slot[coop_pick(g, X, Y, Z)], k, q);
    if (g == 0) coop_put(L, M, k, fp_add(fp_add(fp_dbl(r), r), pre));
    else if (g == 1) coop_put(L, YY, k, r);
    else if (g == 2) coop_put(L, Z, k, fp_dbl(r));
    coop_sync();
    if (g < 3) {
        r = coop_group_mul(L.slot[coop_pick(g, YY, X, M)], L.slot[coop_pick(g, YY, YY, M)], k, q);
    } else if (fuse_qy >= 0 && g < 5) {
        r = coop_group_mul(L.slot[g == 3 ? Z : fuse_qy], L.slot[Z], k, q);
    }
    {
        const u64 s4 = fp_dbl(fp_dbl(__shfl(r, coop_lane_of(1, k))));     // S = 4 X YY, seen by every group
        if (g == 0) {
            L.slot[E][k] = fp_dbl(fp_dbl(fp_dbl(r)));
        } else if (g == 2) {
            const u64 x3 = fp_sub(r, fp_dbl(s4));
            coop_put(L, X, k, x3);
            L.slot[D][k] = fp_sub(s4, x3);
        } else if (fuse_qy >= 0 && g == 3) {
            coop_put(L, fuse_z1z1, k, r);
        } else if (fuse_qy >= 0 && g == 4) {
            L.slot[fuse_qyz][k] = r;
        }
    }
    coop_sync();
    if (g == 0) pre = L.slot[E][k];
    if (g < 2) r = coop_group_mul(L.slot[coop_pick(g, D, E)], L.slot[coop_pick(g, M, W)], k, q);
    if (g == 0) coop_put(L, Y, k, fp_sub(r, pre));
    else if (g == 1) coop_put(L, W, k, fp_dbl(r));
    coop_sync();
}

// P <- P + (QX, QY) affine, (0, 0) = identity; same case analysis as jac_madd.  Five rounds (four when the
// doubling before it computed the first); where an addition needs the product of a neighbouring group the
// value crosses by a wave shuffle.  t[0..8] scratch; with `prefused` t[4] = Z^2 and t[5] = QY Z are given.
//   R1  Z1Z1 = Z^2 | QY Z
//   R2  QX Z1Z1 -> H = U2 - X | (QY Z) Z1Z1 -> R = S2 - Y           (H = 0: doubling or the identity)
//   R3  HH = H^2 | Z3 = Z H | RR = R^2
//   R4  HHH = HH H | V = HH X -> X3 = RR - HHH - 2 V, D = V - X3 | ZZ = Z3^2
//   R5  D R | HHH Y -> Y3 = D R - HHH Y | W3 = ZZ^2
COOP_FN void coop_jac_madd(CoopLds &L, int P, int QX, int QY, const int *t, u32 lane, int ws = 0,
                           bool prefused = false) {
    const int X = P, Y = P + 1, Z = P + 2, W = P + 3;
    const int HH = t[0], RR = t[1], HHH = t[2], D = t[3], Z1Z1 = t[4], QYZ = t[5], H = t[6], R = t[7], ZZ = t[8];
    COOP_LANE_ROLES(lane);
    (void)q;
    const bool p_inf = coop_is_zero(L, Z, lane, ws);
    const bool q_inf = coop_is_zero(L, QX, lane, ws) && coop_is_zero(L, QY, lane, ws);
    if (q_inf) return;
    if (p_inf) {
        if (g == 0) coop_put(L, X, k, L.slot[QX][k]);
        else if (g == 1) coop_put(L, Y, k, L.slot[QY][k]);
        else if (g == 2) coop_put(L, Z, k, k == 0 ? 1ull : 0ull);
        else if (g == 3) coop_put(L, W, k, k == 0 ? 1ull : 0ull);
        coop_sync();
        return;
    }
    u64 r = 0;
    if (!prefused) {
        if (g < 2) r = coop_group_mul(L.slot[coop_pick(g, Z, QY)], L.slot[Z], k, q);
        if (g == 0) coop_put(L, Z1Z1, k, r);
        else if (g == 1) L.slot[QYZ][k] = r;
        coop_sync();
    }
    u64 pre = L.slot[coop_pick(g, X, Y)][k];
    if (g < 2) r = coop_group_mul(L.slot[coop_pick(g, QX, QYZ)], L.slot[Z1Z1], k, q);
    if (g == 0) coop_put(L, H, k, fp_sub(r, pre));
    else if (g == 1) coop_put(L, R, k, fp_sub(r, pre));
    coop_sync();
    if (coop_is_zero(L, H, lane, ws)) {
        if (coop_is_zero(L, R, lane, ws)) {
            coop_jac_dbl(L, P, t, lane, ws);            // p == q
        } else {                                        // p == -q
            if (g == 0) coop_put(L, Z, k, 0ull);
            else if (g == 1) coop_put(L, W, k, 0ull);
            coop_sync();
        }
        return;
    }
    if (g < 3) r = coop_group_mul(L.slot[coop_pick(g, H, Z, R)], L.slot[coop_pick(g, H, H, R)], k, q);
    if (g == 0) L.slot[HH][k] = r;
    else if (g == 1) coop_put(L, Z, k, r);
    else if (g == 2) L.slot[RR][k] = r;
    coop_sync();
    pre = L.slot[RR][k];
    if (g < 3) r = coop_group_mul(L.slot[coop_pick(g, HH, HH, Z)], L.slot[coop_pick(g, H, X, Z)], k, q);
    {
        const u64 hhh = __shfl(r, coop_lane_of(0, k));  // group 0's product, seen by every group
        if (g == 0) {
            L.slot[HHH][k] = r;
        } else if (g == 1) {
            const u64 x3 = fp_sub(fp_sub(pre, hhh), fp_dbl(r));
            coop_put(L, X, k, x3);
            L.slot[D][k] = fp_sub(r, x3);
        } else if (g == 2) {
            coop_put(L, ZZ, k, r);
        }
    }
    coop_sync();
    if (g < 3) r = coop_group_mul(L.slot[coop_pick(g, D, HHH, ZZ)], L.slot[coop_pick(g, R, Y, ZZ)], k, q);
    {
        const u64 yh = __shfl(r, coop_lane_of(1, k));   // group 1's product
        if (g == 0) coop_put(L, Y, k, fp_sub(r, yh));
        else if (g == 2) coop_put(L, W, k, r);
    }
    coop_sync();
}

// P1 <- P1 + P2, both Jacobian (add-1998-cmo-2; P2 = slots X2, Y2, Z2, its W is not needed), Z = 0 is the
// identity.  Five rounds; t[0..8] scratch.
//   R1  A = Z1^2 | B = Z2^2 | C = Y1 Z2 | D = Y2 Z1 | E = Z1 Z2
//   R2  U1 = X1 B | X2 A -> H = U2 - U1 | S1 = C B | D A -> R = S2 - S1      (H = 0: doubling or the identity)
//   R3  HH = H^2 | Z3 = E H | RR = R^2
//   R4  HHH = H HH | V = U1 HH -> X3 = RR - HHH - 2 V, F = V - X3 | ZZ = Z3^2
//   R5  R F | S1 HHH -> Y3 = R F - S1 HHH | W3 = ZZ^2
COOP_FN void coop_jac_add(CoopLds &L, int P1, int P2, const int *t, u32 lane, int ws = 0) {
    const int X1 = P1, Y1 = P1 + 1, Z1 = P1 + 2, W1 = P1 + 3, X2 = P2, Y2 = P2 + 1, Z2 = P2 + 2;
    const int A = t[0], B = t[1], C = t[2], Dd = t[3], E = t[4], U1 = t[5], S1 = t[6], H = t[7], R = t[8];
    const int HH = t[0], RR = t[1], HHH = t[2], F = t[3], ZZ = t[4];
    COOP_LANE_ROLES(lane);
    (void)q;
    if (coop_is_zero(L, Z2, lane, ws)) return;
    if (coop_is_zero(L, Z1, lane, ws)) {
        // copy X, Y, Z and rebuild W = Z^4
        if (g == 0) coop_put(L, X1, k, L.slot[X2][k]);
        else if (g == 1) coop_put(L, Y1, k, L.slot[Y2][k]);
        else if (g == 2) coop_put(L, Z1, k, L.slot[Z2][k]);
        coop_sync();
        coop_mul(L, W1, Z1, Z1, lane, ws);
        coop_mul(L, W1, W1, W1, lane, ws);
        return;
    }
    u64 r = 0;
    if (g < 5) {
        const int sa = g == 0 ? Z1 : (g == 1 ? Z2 : (g == 2 ? Y1 : (g == 3 ? Y2 : Z1)));
        const int sb = g == 0 ? Z1 : (g == 1 ? Z2 : (g == 2 ? Z2 : (g == 3 ? Z1 : Z2)));
        r = coop_group_mul(L.slot[sa], L.slot[sb], k, q);
        coop_put(L, g == 0 ? A : (g == 1 ? B : (g == 2 ? C : (g == 3 ? Dd : E))), k, r);
    }
    coop_sync();
    if (g < 4) r = coop_group_mul(L.slot[coop_pick(g, X1, X2, C, Dd)], L.slot[coop_pick(g, B, A, B, A)], k, q);
    {
        const u64 u1 = __shfl(r, coop_lane_of(0, k)), s1 = __shfl(r, coop_lane_of(2, k));
        if (g == 0) coop_put(L, U1, k, r);
        else if (g == 1) coop_put(L, H, k, fp_sub(r, u1));
        else if (g == 2) coop_put(L, S1, k, r);
        else if (g == 3) coop_put(L, R, k, fp_sub(r, s1));
    }
    coop_sync();
    if (coop_is_zero(L, H, lane, ws)) {
        if (coop_is_zero(L, R, lane, ws)) {
            coop_jac_dbl(L, P1, t, lane, ws);           // same point
        } else {                                        // opposite points
            if (g == 0) coop_put(L, Z1, k, 0ull);
            else if (g == 1) coop_put(L, W1, k, 0ull);
            coop_sync();
        }
        return;
    }
    if (g < 3) r = coop_group_mul(L.slot[coop_pick(g, H, E, R)], L.slot[coop_pick(g, H, H, R)], k, q);
    if (g == 0) coop_put(L, HH, k, r);
    else if (g == 1) coop_put(L, Z1, k, r);
    else if (g == 2) coop_put(L, RR, k, r);
    coop_sync();
    if (g < 3) r = coop_group_mul(L.slot[coop_pick(g, H, U1, Z1)], L.slot[coop_pick(g, HH, HH, Z1)], k, q);
    {
        const u64 hhh = __shfl(r, coop_lane_of(0, k));
        if (g == 0) {
            coop_put(L, HHH, k, r);
        } else if (g == 1) {
            const u64 x3 = fp_sub(fp_sub(L.slot[RR][k], hhh), fp_dbl(r));
            coop_put(L, X1, k, x3);
            coop_put(L, F, k, fp_sub(r, x3));
        } else if (g == 2) {
            coop_put(L, ZZ, k, r);
        }
    }
    coop_sync();
    if (g < 3) r = coop_group_mul(L.slot[coop_pick(g, R, S1, ZZ)], L.slot[coop_pick(g, F, HHH, ZZ)], k, q);
    {
        const u64 sh = __shfl(r, coop_lane_of(1, k));
        if (g == 0) coop_put(L, Y1, k, fp_sub(r, sh));
        else if (g == 2) coop_put(L, W1, k, r);
    }
    coop_sync();
}

// ------------------------------------------------------------------------------------------------
// Signature::verify for ONE signature by ONE wave (reference src/signature.rs:181-205): the same
// algorithm as ssa_k_hash + ssa_k_verify (affine table 1P..8P, signed 4-bit windows, comb for G,
// x-only compare, optional [q]P == O first), every Fp6 operation spread over the wave.
// LDS slot map: two per-wave working sets (accumulator X, Y, Z, W, addend, nine temporaries, three for
// inversions) and the slots both waves share (public key, signature x, the 8-entry table with rows X, Y, Z, C;
// C holds W = Z^4 while the table is being built and the prefix products of the normalisation afterwards).
namespace coop_slots {
enum : int { WS_SLOTS = 18, PX = 2 * WS_SLOTS, PY, SX, TAB, TAB2 = TAB + 32, RX = TAB2 + 32, RY = RX + 1 };
}
#define COOP_WORKING_SET(ws)                                                                               \
    const int AX = (ws) * coop_slots::WS_SLOTS, AY = AX + 1, AZ = AX + 2, AW = AX + 3, QX = AX + 4, QY = AX + 5,  \
              T0 = AX + 6, I0 = AX + 15, I1 = AX + 16, I2 = AX + 17;                                           \
    (void)AY; (void)AZ; (void)AW; (void)QX; (void)QY; (void)T0; (void)I0; (void)I1; (void)I2

// Rescue-Prime permutation on the 12 lanes 0..11; state in plane L.st[0], L.st[1] is scratch
COOP_FN void coop_rescue_permutation(CoopLds &L, const DevParams *__restrict__ prm, u32 lane, int ws = 0) {
    const u32 nr = prm->n_rounds;
    const bool small = (prm->flags & PRM_FLAG_SMALL_MDS) != 0;
    u64 *S = L.st[0], *T = L.st[1];
#pragma unroll 1
    for (u32 r = 0; r < nr; r++) {
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
            u64 *src = half ? T : S, *dst = half ? S : T;
            if (lane < 12) src[lane] = half ? sbox_one<true>(src[lane]) : sbox_one<false>(src[lane]);
            coop_sync();
            if (lane < 12) {
                const u64 *row = prm->mds + lane * 12;
                const u64 *ark = (half ? prm->ark2 : prm->ark1) + 12 * r;
                fp_acc acc;
                acc_zero(acc);
#pragma unroll 1
                for (int j = 0; j < 12; j++) {
                    if (small) acc_mac32(acc, src[j], (u32)row[j]);
                    else acc_mac(acc, src[j], row[j]);
                }
                dst[lane] = fp_add(acc_reduce(acc), ark[lane]);
            }
            coop_sync();
        }
    }
}

// hash_message (src/signature.rs:274-306) of the signature staged in slots SX / PX / PY -> scalar h
COOP_FN sc256 coop_hash_message(CoopLds &L, const DevParams *__restrict__ prm, const u8 *m, u32 len, u32 lane, int ws = 0) {
    using namespace coop_slots;
    const u32 nmsg = (len + 6u) / 7u, n_felts = 13u + nmsg;
    u64 *S = L.st[0];
    if (lane < 12) S[lane] = ((int)lane == prm->cap_len_idx) ? (u64)n_felts : 0ull;
    coop_sync();
    const u32 rate_off = prm->rate_off;
    const bool pad1 = prm->pad_mode == 1;
    const u32 n_blocks = pad1 ? n_felts / 8 + 1 : (n_felts + 7) / 8;
#pragma unroll 1
    for (u32 b = 0; b < n_blocks; b++) {
        if (lane < 8) {
            const u32 idx = 8 * b + lane;
            u64 v = 0;
            bool have = false;
            if (idx < n_felts) {
                have = true;
                if (idx < 6) v = L.slot[SX][idx];
                else if (idx < 12) v = L.slot[PX][idx - 6];
                else if (idx == 12) v = L.slot[PY][0];
                else v = msg_felt(m, len, idx - 13u);
            } else if (pad1 && idx == n_felts) {
                have = true;
                v = 1ull;
            }
            if (have) S[rate_off + lane] = fp_add(S[rate_off + lane], v);
        }
        coop_sync();
        coop_rescue_permutation(L, prm, lane, ws);
    }
    sc256 h;
#pragma unroll
    for (int k = 0; k < 4; k++) h.w[k] = fp_canon(S[prm->digest_off + k]);
    coop_sync();
    return sc_reduce256(h);    // Scalar::from_bits_vartime, src/signature.rs:189-192
}

// affine multiples 1P..8P into the TAB slots (rows of X, Y, Z, C); identity multiples become (0, 0)
// (tab, px, py: the table rows and the slots of the affine point they are built from -- TAB / PX / PY of the
// verification kernel by default, TAB2 / RX / RY for the second point of the small-batch MSM form)
COOP_FN void coop_build_table(CoopLds &L, bool p_inf, u32 lane, int ws = 0, int tab = coop_slots::TAB,
                              int px = coop_slots::PX, int py = coop_slots::PY) {
    using namespace coop_slots;
    COOP_WORKING_SET(ws);
    int t[9];
#pragma unroll
    for (int k = 0; k < 9; k++) t[k] = T0 + k;
    auto row = [tab](int e, int f) { return tab + 4 * e + f; };
    if (p_inf) {
#pragma unroll 1
        for (int e = 0; e < 8; e++) {
            coop_set(L, row(e, 0), 0ull, lane, ws);
            coop_set(L, row(e, 1), 0ull, lane, ws);
        }
        return;
    }
    coop_copy(L, row(0, 0), px, lane, ws);
    coop_copy(L, row(0, 1), py, lane, ws);
    coop_set(L, row(0, 2), 1ull, lane, ws);
    coop_set(L, row(0, 3), 1ull, lane, ws);
    // (source row, operation): 2P = dbl 1P, 3P = 2P + P, 4P = dbl 2P, 5P = 4P + P, 6P = dbl 3P, 7P = 6P + P, 8P = dbl 4P
    const int src[7] = {0, 1, 1, 3, 2, 5, 3};
    const bool is_add[7] = {false, true, false, true, false, true, false};
#pragma unroll 1
    for (int e = 1; e < 8; e++) {
        const int s = src[e - 1];
#pragma unroll 1
        for (int f = 0; f < 4; f++) coop_copy(L, AX + f, row(s, f), lane, ws);
        if (is_add[e - 1]) coop_jac_madd(L, AX, px, py, t, lane, ws);
        else coop_jac_dbl(L, AX, t, lane, ws);
#pragma unroll 1
        for (int f = 0; f < 4; f++) coop_copy(L, row(e, f), AX + f, lane, ws);
    }
    // Montgomery's trick over the (non-zero) Z's
    coop_set(L, QX, 1ull, lane, ws);                             // running prefix product
#pragma unroll 1
    for (int e = 1; e < 8; e++) {
        if (coop_is_zero(L, row(e, 2), lane, ws)) coop_set(L, QY, 1ull, lane, ws);
        else coop_copy(L, QY, row(e, 2), lane, ws);
        coop_mul(L, QX, QX, QY, lane, ws);
        coop_copy(L, row(e, 3), QX, lane, ws);
    }
    coop_inv(L, QX, QX, I0, I1, I2, lane, ws);                   // QX = 1 / prod Z
#pragma unroll 1
    for (int e = 7; e >= 1; e--) {
        const bool zero = coop_is_zero(L, row(e, 2), lane, ws);
        if (zero) coop_set(L, QY, 1ull, lane, ws);
        else coop_copy(L, QY, row(e, 2), lane, ws);
        if (e > 1) coop_mul(L, I0, QX, row(e - 1, 3), lane, ws); // 1 / Z_e
        else coop_copy(L, I0, QX, lane, ws);
        coop_mul(L, QX, QX, QY, lane, ws);
        coop_mul(L, I1, I0, I0, lane, ws);                       // Zinv^2
        coop_mul(L, row(e, 0), row(e, 0), I1, lane, ws);
        coop_mul(L, I1, I1, I0, lane, ws);                       // Zinv^3
        coop_mul(L, row(e, 1), row(e, 1), I1, lane, ws);
        if (zero) {
            coop_set(L, row(e, 0), 0ull, lane, ws);
            coop_set(L, row(e, 1), 0ull, lane, ws);
        }
    }
}

// (AX, AY, AZ) <- [k] P from the table, k < 2^255.  Leading windows whose digit is zero are skipped while the
// accumulator is still the identity (short scalars: the 128-bit coefficients of the MSM form).
COOP_FN void coop_mul_table(CoopLds &L, const sc256 &k, u32 lane, int ws = 0, int tab = coop_slots::TAB) {
    using namespace coop_slots;
    const int TAB = tab;
    COOP_WORKING_SET(ws);
    int t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = T0 + i;
    const sc256 kr = sc_recode_offset(k);
    coop_set(L, AX, 1ull, lane, ws);
    coop_set(L, AY, 1ull, lane, ws);
    coop_set(L, AZ, 0ull, lane, ws);
    coop_set(L, AW, 0ull, lane, ws);
    const u32 top = sc_nibble(kr, 63u);
    bool started = false;      // wave-uniform: some non-zero digit has been consumed
    if (top != 0) {
        started = true;
        const int e = (int)top - 1;
        if (!(coop_is_zero(L, TAB + 4 * e, lane, ws) && coop_is_zero(L, TAB + 4 * e + 1, lane, ws))) {
            coop_copy(L, AX, TAB + 4 * e, lane, ws);
            coop_copy(L, AY, TAB + 4 * e + 1, lane, ws);
            coop_set(L, AZ, 1ull, lane, ws);
            coop_set(L, AW, 1ull, lane, ws);
        }
    }
#pragma unroll 1
    for (int w = 62; w >= 0; w--) {
        const int digit = (int)sc_nibble(kr, (u32)w) - 8;
        if (digit == 0 && !started) continue;      // doublings of the identity
        started = true;
        if (digit != 0) {   // the addend of this window (second operands only: no 7x halves needed)
            const int e = (digit < 0 ? -digit : digit) - 1;
            if (lane < 6) {
                L.slot[QX][lane] = L.slot[TAB + 4 * e][lane];
                const u64 y = L.slot[TAB + 4 * e + 1][lane];
                L.slot[QY][lane] = digit < 0 ? fp_neg(y) : y;
            }
            coop_sync();
        }
#pragma unroll 1
        for (int d = 0; d < 3; d++) coop_jac_dbl(L, AX, t, lane, ws);
        if (digit != 0) {   // the fourth doubling also computes the first round of the addition
            coop_jac_dbl(L, AX, t, lane, ws, QY, t[4], t[5]);
            coop_jac_madd(L, AX, QX, QY, t, lane, ws, true);
        } else {
            coop_jac_dbl(L, AX, t, lane, ws);
        }
    }
}

// Two waves per signature (a 128-thread block).  After the inputs are staged:
//   wave 0: key checks + table build        ||  wave 1: hash_message -> h
//   wave 0: [q]P == O (with the flag)       ||  wave 1: [h]P + [e]G and the x comparison
// The waves share the table and the staged inputs, keep separate working sets, and meet at three
// workgroup barriers.  Check order as in the reference (src/signature.rs:181-205): key decoding,
// subgroup check (InvalidPublicKey), then the signature's x (the reference panics: SSA_MALFORMED).
struct CoopShared {
    u32 ok_pk, ok_sig, tors_bad, eq, x_bad;
    u64 h[4];
};

COOP_FN u32 coop_verify_two_waves(CoopLds &L, CoopShared &sh, const DevParams *__restrict__ prm,
                                  const u8 *__restrict__ sig, const u8 *__restrict__ pk, bool inf,
                                  const u8 *__restrict__ m, u32 len, const u64 *__restrict__ gtab, u32 flags,
                                  u32 lane, int ws) {
    using namespace coop_slots;
    COOP_WORKING_SET(ws);
    int t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = T0 + i;
    const sc256 e = ld_sc(sig + 49);
    if (ws == 0) {   // stage the inputs, canonical-limb checks
        bool pk_ok = true, sig_ok = true;
        if (lane < 12) {
            const u32 c = lane % 6u;
            const u64 xs = ld_u64_le(sig + 8 * c), px = ld_u64_le(pk + 8 * c), py = ld_u64_le(pk + 48 + 8 * c);
            coop_store7(L, SX, xs, lane);
            coop_store7(L, PX, px, lane);
            coop_store7(L, PY, py, lane);
            pk_ok = px < FP_P && py < FP_P;
            sig_ok = xs < FP_P;
        }
        pk_ok = __all(pk_ok);
        sig_ok = __all(sig_ok) && !sc_geq_q(e);
        if ((flags & VF_SIG_FLAG_BYTE) && sig_ok) {
            const bool x0 = __all(lane < 6 ? ld_u64_le(sig + 8 * lane) == 0ull : true);
            sig_ok = sig_flag_precheck(sig[48], x0) == ST_OK;
        }
        if (lane == 0) {
            sh.ok_pk = pk_ok;
            sh.ok_sig = sig_ok;
            sh.tors_bad = 0;
            sh.eq = 0;
            sh.x_bad = 0;
        }
    }
    __syncthreads();
    if (ws == 0) {
        bool ok = sh.ok_pk != 0;
        if (ok && !inf) {   // y^2 == x^3 + x + (u + 395)
            coop_mul(L, T0, PX, PX, lane, ws);
            coop_mul(L, T0, T0, PX, lane, ws);
            coop_add(L, T0, T0, PX, lane, ws);
            if (lane < 2) L.slot[T0][lane] = fp_add(L.slot[T0][lane], lane == 0 ? 395ull : 1ull);   // compared only
            coop_sync();
            coop_mul(L, T0 + 1, PY, PY, lane, ws);
            ok = coop_eq(L, T0, T0 + 1, lane, ws);
        }
        if (lane == 0) sh.ok_pk = ok;
        if (ok) coop_build_table(L, inf, lane, ws);
    } else {
        const sc256 h = coop_hash_message(L, prm, m, len, lane, ws);   // reads SX, PX, PY only
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 4; k++) sh.h[k] = h.w[k];
        }
    }
    __syncthreads();
    if (sh.ok_pk) {
        if (ws == 0) {
            if (flags & 1u) {   // is_torsion_free, src/signature.rs:182-184
                sc256 q;
#pragma unroll
                for (int k = 0; k < 4; k++) q.w[k] = SC_Q(k);
                coop_mul_table(L, q, lane, ws);
                const bool at_identity = coop_is_zero(L, AZ, lane, ws);
                if (lane == 0) sh.tors_bad = at_identity ? 0u : 1u;
            }
        } else if (sh.ok_sig) {
            sc256 h;
#pragma unroll
            for (int k = 0; k < 4; k++) h.w[k] = sh.h[k];
            coop_mul_table(L, h, lane, ws);                               // [h]P
            const GtabGeom gg = gtab_geom(gtab);
#pragma unroll 1
            for (u32 w = 0; w < gg.count; w++) {                          // + [e]G, src/signature.rs:196-198
                const u32 d = sc_bits(e, w * gg.bits, gg.bits);
                if (d != 0) {
                    const u64 *rowp = gtab + (((size_t)w << gg.bits) + d) * 12;
                    if (lane < 24) {   // lanes 0..11: x, 7x; lanes 12..23: y, 7y
                        const u32 half = lane / 12u, c = lane % 12u;
                        const u64 v = rowp[6u * half + c % 6u];
                        L.slot[half ? QY : QX][c] = c < 6 ? v : fp_mul_small(v, 7u);
                    }
                    coop_sync();
                    coop_jac_madd(L, AX, QX, QY, t, lane, ws);
                }
            }
            bool eq;
            const bool r_inf = coop_is_zero(L, AZ, lane, ws);
            const u32 fbyte = sig[48];
            if ((flags & VF_SIG_FLAG_BYTE) && (fbyte & 0x80u)) {
                eq = r_inf;                                               // R decodes to the identity
            } else if (r_inf) {
                eq = !(flags & VF_SIG_FLAG_BYTE) && coop_is_zero(L, SX, lane, ws);   // the identity's x is taken as 0
            } else {
                coop_mul(L, T0, AZ, AZ, lane, ws);
                coop_mul(L, T0, SX, T0, lane, ws);
                eq = coop_eq(L, AX, T0, lane, ws);                        // X == x * Z^2, src/signature.rs:200
                if (eq && (flags & VF_SIG_FLAG_BYTE)) {                   // the y the flag byte selects (src/batch.rs:104)
                    coop_inv(L, T0, AZ, T0 + 1, T0 + 2, T0 + 3, lane, ws);
                    coop_mul(L, T0 + 1, T0, T0, lane, ws);
                    coop_mul(L, T0 + 1, T0 + 1, T0, lane, ws);
                    coop_mul(L, T0 + 1, AY, T0 + 1, lane, ws);            // y = Y / Z^3
                    fp6 y;
#pragma unroll
                    for (int c = 0; c < 6; c++) y.c[c] = L.slot[T0 + 1][c];
                    eq = f6_lex_largest(y) == ((fbyte & 0x40u) != 0);
                }
            }
            if ((flags & VF_SIG_FLAG_BYTE) && !eq && !(fbyte & 0x80u)) {
                // an x with no curve point (x^3 + x + u + 395 a non-square: its norm is a non-residue of Fp):
                // from_compressed is None and the reference panics
                coop_mul(L, T0, SX, SX, lane, ws);
                coop_mul(L, T0, T0, SX, lane, ws);
                coop_add(L, T0, T0, SX, lane, ws);
                if (lane < 12 && (lane % 6u) < 2u) {
                    const u64 add = (lane % 6u) == 0 ? 395ull : 1ull;
                    L.slot[T0][lane] = fp_add(L.slot[T0][lane], lane < 6 ? add : 7ull * add);
                }
                coop_sync();
                if (!coop_is_zero(L, T0, lane, ws)) {
                    coop_inv(L, T0 + 1, T0, T0 + 2, T0 + 3, T0 + 4, lane, ws);
                    if (!fp_is_square(L.slot[T0 + 4][0]) && lane == 0) sh.x_bad = 1;
                }
            }
            if (lane == 0) sh.eq = eq;
        }
    }
    __syncthreads();
    if (!sh.ok_pk) return ST_MALFORMED;
    if ((flags & 1u) && sh.tors_bad) return ST_INVALID_PK;
    if (!sh.ok_sig || sh.x_bad) return ST_MALFORMED;
    return sh.eq ? ST_OK : ST_INVALID_SIG;
}

}  // namespace ssa
