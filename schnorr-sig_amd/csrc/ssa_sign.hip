// The signing side as a product feature: KeyPair::new / sign / sign_and_bind_pkey
// (reference src/keypair.rs:57-65, src/signature.rs:65-156), PublicKey::to_bytes (src/public.rs:49-51) and
// KeyedSignature::to_bytes (src/signature.rs:237-245).
//
//   ssa_k_sign_ct     CONSTANT-TIME keygen + sign (SSA_FLAG_SIGN_CT).  The reference signs with the constant-time table
//                     product `&BASEPOINT_TABLE * r` and Scalar::from_bits (src/signature.rs:67,116,123); the
//                     throughput signer ssa_k_sign (ssa_kernels.hpp) skips zero windows and indexes a 100 MB table
//                     with secret 16-bit windows.  Here:
//                       * fixed 4-bit windows, 64 per scalar, over a 98 KB table [d 16^w]G (d = 1..15) that is READ IN
//                         FULL for every window -- wave-uniform addresses, the same for every secret -- and the entry is
//                         picked with v_cndmask selects;
//                       * no window is skipped: digit 0 adds entry 1 and the sum is discarded by a select;
//                       * the accumulator starts at a public offset point B = [b]G (and -B is added at the end), so that
//                         it is never the identity and meets +-(table entry) only with probability ~2^-250 per
//                         addition: the generic addition formulas run with NO exceptional-case branch;
//                       * a lane whose addition did hit an exceptional input (Z3 = 0: only k = 0 gets there for sure --
//                         B - B) is flagged and recomputed by the exact variable-time code AFTER the constant-time
//                         pass: the one data-dependent branch of the kernel, taken with negligible probability;
//                       * this translation unit is compiled with the COMPILED Fp6 blocks (SSA_NO_F6_ASM): the generated
//                         blocks of the verification kernels send their rare reduction borrow to a cold path behind a
//                         branch -- harmless for public data, not acceptable around secrets;
//                       * scalar arithmetic mod q (e = r - sk h, src/signature.rs:124) by masked subtractions.
//                     What is NOT secret and therefore not treated: the message, the public key, R and the challenge
//                     hash (the S-box blocks keep their flagged-lane fallback: their inputs are public).
//   ssa_k_ctab        the 4-bit comb table and +-B, built once per context from the 16-bit comb table
//   ssa_k_compress    AffinePoint::to_compressed: 96-byte affine -> 49-byte wire form
//   ssa_k_pack_keyed  (pk, sig) -> the 130-byte KeyedSignature record pk(49) || sig(81)
#define SSA_NO_KERNELS 1
#define SSA_NO_COOP 1
#define SSA_NO_F6_ASM 1
#define SSA_PLAIN_PRESCALE 1   // the squaring's pre-scaled operands without the guarded short form (a data-dependent branch)
#include "ssa_ctx.hpp"

namespace ssa {

constexpr int CT_WINDOWS = 64, CT_ENTRIES = 16;
constexpr size_t CTAB_ROWS = (size_t)CT_WINDOWS * CT_ENTRIES + 2;     // + B, -B
constexpr size_t CTAB_B = (size_t)CT_WINDOWS * CT_ENTRIES, CTAB_NEG_B = CTAB_B + 1;

// the public offset b of the accumulator ("SSA_CT_BLIND_OFFSET_ROUND4_----", 31 ASCII bytes: < 2^248 < q)
SSA_DEV sc256 ct_offset_scalar() {
    sc256 b;
    b.w[0] = 0x425f54435f415353ULL;   // the bytes of the string, little-endian (any fixed non-zero value below q serves)
    b.w[1] = 0x46464f5f444e494cULL;
    b.w[2] = 0x4e554f525f544553ULL;
    b.w[3] = 0x002d2d2d2d5f3444ULL;
    return b;
}

__global__ void __launch_bounds__(64)
ssa_k_ctab(const u64 *__restrict__ gtab, u64 *__restrict__ ctab) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= CTAB_ROWS) return;
    sc256 s;
#pragma unroll
    for (int k = 0; k < 4; k++) s.w[k] = 0;
    if (t < CTAB_B) {
        const u32 w = (u32)(t >> 4), d = (u32)(t & 15u);
        const u64 v = (u64)d << ((w & 15u) * 4u);
        const u32 wi = w >> 4;
        if (wi == 0) s.w[0] = v;
        if (wi == 1) s.w[1] = v;
        if (wi == 2) s.w[2] = v;
        if (wi == 3) s.w[3] = v;
    } else {
        s = ct_offset_scalar();
    }
    aff a = jac_to_aff(add_base_mul(jac_identity(), gtab, s));       // public data: the variable-time comb walk
    if (t == CTAB_NEG_B) a.y = f6_canon(f6_neg(a.y));
    st_aff(ctab + 12 * t, a);
}

// ---- constant-time pieces -------------------------------------------------------------------------------
SSA_DEV u64 ct_sel(bool pick_b, u64 a, u64 b) { return pick_b ? b : a; }     // v_cndmask x2, never a branch
SSA_DEV fp6 ct_sel6(bool pick_b, const fp6 &a, const fp6 &b) {
    fp6 r;
#pragma unroll
    for (int i = 0; i < 6; i++) r.c[i] = ct_sel(pick_b, a.c[i], b.c[i]);
    return r;
}

// zero test without short-circuit evaluation (f6_is_zero's && chain compiles to exec-mask branches)
SSA_DEV bool f6_is_zero_ct(const fp6 &a) {
    u32 z = 1u;
#pragma unroll
    for (int i = 0; i < 6; i++) z &= (u32)(a.c[i] == 0ull) | (u32)(a.c[i] == FP_P);
    return z != 0u;
}

// generic mixed addition p + (x2, y2) with NO exceptional-case handling: p = O, p = +-q give Z3 = 0 (and X3, Y3
// that mean nothing), reported through `bad`
SSA_DEV jac jac_madd_ct(const jac &p, const aff &q, bool &bad) {
    const fp6 Z1Z1 = f6_sqr(p.Z);
    const fp6 H = f6_sub(f6_mul(q.x, Z1Z1), p.X);
    const fp6 R = f6_sub(f6_mul(f6_mul(q.y, p.Z), Z1Z1), p.Y);
    const fp6 HH = f6_sqr(H);
    const fp6 HHH = f6_mul(H, HH);
    const fp6 V = f6_mul(p.X, HH);
    jac r;
    r.X = f6_sub(f6_sub(f6_sqr(R), HHH), f6_dbl(V));
    r.Y = f6_sub(f6_mul(R, f6_sub(V, r.X)), f6_mul(p.Y, HHH));
    r.Z = f6_mul(p.Z, H);
    bad = bad | f6_is_zero_ct(r.Z);
    return r;
}

// [k]G for a SECRET k < 2^256: 64 windows, every table entry of a window read and selected, 64 additions + the
// closing -B, identical instruction stream whatever k is.  Out of line: one copy of the loop for sk and the nonce.
SSA_FN void ct_base_mul(jac *__restrict__ out, bool *__restrict__ bad_out, const u64 *__restrict__ ctab, const sc256 *__restrict__ kp) {
    const sc256 k = *kp;
    jac acc = jac_from_aff(ld_aff(ctab + 12 * CTAB_B));
    bool bad = false;
#pragma unroll 1
    for (int w = 0; w < CT_WINDOWS; w++) {
        const u32 wi = (u32)w >> 4;                       // (w is the loop counter: uniform and public)
        u64 word = k.w[0];
        word = wi == 1 ? k.w[1] : word;
        word = wi == 2 ? k.w[2] : word;
        word = wi == 3 ? k.w[3] : word;
        const u32 d = (u32)(word >> (((u32)w & 15u) * 4u)) & 15u;
        const u32 dsel = d | (u32)(d == 0u);              // digit 0 walks through the addition with entry 1
        const u64 *row = ctab + 12 * ((size_t)w * CT_ENTRIES);
        aff sel = ld_aff(row + 12);
#pragma unroll 1
        for (u32 e = 2; e < (u32)CT_ENTRIES; e++) {
            const aff t = ld_aff(row + 12 * e);           // address depends on (w, e) only
            const bool hit = dsel == e;
            sel.x = ct_sel6(hit, sel.x, t.x);
            sel.y = ct_sel6(hit, sel.y, t.y);
        }
        bool bad_w = false;
        const jac nx = jac_madd_ct(acc, sel, bad_w);
        const bool take = d != 0u;
        acc.X = ct_sel6(take, acc.X, nx.X);
        acc.Y = ct_sel6(take, acc.Y, nx.Y);
        acc.Z = ct_sel6(take, acc.Z, nx.Z);
        bad = bad | (bad_w & take);
    }
    acc = jac_madd_ct(acc, ld_aff(ctab + 12 * CTAB_NEG_B), bad);
    *out = acc;
    *bad_out = bad;
}

// affine (x, y), canonical limbs, with no branch on Z: 0^-1 comes out as 0 and the identity maps to (0, 0).
// Out of line like ct_base_mul: the secret-dependent code of the kernel lives in three functions whose bodies the CPU
// suite disassembles and checks for data-dependent branches (tests/test_sign_ct_static.py).
SSA_FN void ct_to_aff(aff *__restrict__ out, const jac *__restrict__ pp) {
    const jac p = *pp;
    const fp6 zi = f6_inv(p.Z);
    const fp6 zi2 = f6_sqr(zi);
    aff r;
    r.x = f6_canon(f6_mul(p.X, zi2));
    r.y = f6_canon(f6_mul(p.Y, f6_mul(zi, zi2)));
    *out = r;
}

// ---- scalars mod q by masked subtractions ----------------------------------------------------------------
// r (five limbs, < 3q) -> r mod q: two conditional subtractions of q, both always computed
SSA_DEV sc256 sc_fold_ct(u64 (&r)[5]) {
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        u64 d[5];
        u64 bw = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const u64 t = r[i] - SC_Q(i);
            const u64 b1 = r[i] < SC_Q(i);
            const u64 t2 = t - bw;
            const u64 b2 = t < bw;
            d[i] = t2;
            bw = b1 | b2;
        }
        d[4] = r[4] - bw;
        const bool ge = r[4] >= bw;          // no borrow out of the top limb: r >= q
#pragma unroll
        for (int i = 0; i < 5; i++) r[i] = ct_sel(ge, r[i], d[i]);
    }
    sc256 out;
#pragma unroll
    for (int i = 0; i < 4; i++) out.w[i] = r[i];
    return out;
}
SSA_DEV sc256 sc_reduce256_ct(const sc256 &a) {       // any 256-bit value: floor(2^256 / q) = 2
    u64 r[5] = {a.w[0], a.w[1], a.w[2], a.w[3], 0ull};
    return sc_fold_ct(r);
}
// a * b mod q, a, b < q: schoolbook + Barrett as sc_mul_mod (ssa_kernels.hpp), the final corrections masked
SSA_DEV sc256 sc_mul_mod_ct(const sc256 &a, const sc256 &b) {
    const u64 MU[4] = {0xdfd9f45eab999731ULL, 0x3314f7c7edb24b7dULL, 0x8c4072a8b88f9d66ULL, 0x8542d23b3c0cc598ULL};
    const u64 QL[4] = {SC_Q(0), SC_Q(1), SC_Q(2), SC_Q(3)};
    u64 x[8];
    sc_mul_4x4(a.w, b.w, x);
    u64 q1[4];
#pragma unroll
    for (int i = 0; i < 4; i++) q1[i] = (x[3 + i] >> 62) | (x[4 + i] << 2);
    u64 q2[8];
    sc_mul_4x4(q1, MU, q2);
    const u64 q3[4] = {q2[4], q2[5], q2[6], q2[7]};
    u64 t[8];
    sc_mul_4x4(q3, QL, t);
    u64 r[5];
    u64 borrow = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const u64 d = x[i] - t[i];
        const u64 b1 = x[i] < t[i];
        const u64 d2 = d - borrow;
        const u64 b2 = d < borrow;
        r[i] = d2;
        borrow = b1 | b2;
    }
    return sc_fold_ct(r);
}
// a - b mod q (a, b < q): a + (q - b), one masked subtraction
SSA_DEV sc256 sc_sub_mod_ct(const sc256 &a, const sc256 &b) {
    u64 r[5];
    u64 bw = 0, carry = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {            // nb = q - b  (b < q: no borrow out)
        const u64 t = SC_Q(i) - b.w[i];
        const u64 b1 = SC_Q(i) < b.w[i];
        const u64 t2 = t - bw;
        const u64 b2 = t < bw;
        bw = b1 | b2;
        const u64 s = a.w[i] + t2;           // a + nb
        const u64 c1 = s < t2;
        const u64 s2 = s + carry;
        const u64 c2 = s2 < s;
        r[i] = s2;
        carry = c1 | c2;
    }
    r[4] = carry;
    return sc_fold_ct(r);                    // a + q - b < 2q
}

// e = r - sk h mod q (src/signature.rs:124); sk, r secret, h public
SSA_FN void ct_response(sc256 *__restrict__ e, const sc256 *__restrict__ r, const sc256 *__restrict__ sk,
                        const sc256 *__restrict__ h) {
    *e = sc_sub_mod_ct(*r, sc_mul_mod_ct(*sk, *h));
}
// secret scalar bytes -> value mod q
SSA_FN void ct_load_scalar(sc256 *__restrict__ out, const u8 *__restrict__ p) { *out = sc_reduce256_ct(ld_sc(p)); }

__global__ void __launch_bounds__(256, 2)      // two waves per SIMD: 256 registers (the default allocation took 266 and ran one)
ssa_k_sign_ct(const DevParams *__restrict__ prm, const u64 *__restrict__ ctab, const u64 *__restrict__ gtab,
              const u8 *__restrict__ sks, const u8 *__restrict__ nonces, MsgView mv, size_t n,
              u8 *__restrict__ pks_out, u8 *__restrict__ sigs_out) {
    __shared__ u64 lds[RS_LDS_U64];
    u64 *A = lds + threadIdx.x, *B = A;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    sc256 sk, r;
    ct_load_scalar(&sk, sks + 32 * i);
    ct_load_scalar(&r, nonces + 32 * i);
    jac pj, rj;
    bool bad_p, bad_r;
    ct_base_mul(&pj, &bad_p, ctab, &sk);                                   // src/public.rs:29
    ct_base_mul(&rj, &bad_r, ctab, &r);                                    // src/signature.rs:116
    // the one data-dependent branch: an addition met an exceptional input (k = 0 -- excluded by the host entry point --
    // or an event of probability ~2^-250): that lane's product is recomputed by the exact code
    if (bad_p) pj = add_base_mul(jac_identity(), gtab, sk);
    if (bad_r) rj = add_base_mul(jac_identity(), gtab, r);
    aff pk, rp;
    ct_to_aff(&pk, &pj);
    ct_to_aff(&rp, &rj);
    // from here on everything but sk, r and e is public
    st_fp6(pks_out + 96 * i, pk.x);
    st_fp6(pks_out + 96 * i + 48, pk.y);
    u32 len;
    const u8 *m = msg_ptr(mv, i, len);
    u64 d[4];
    hash_message_lane(A, B, prm, rp.x, pk.x, pk.y.c[0], m, len, d);        // src/signature.rs:118
    sc256 h;
#pragma unroll
    for (int k = 0; k < 4; k++) h.w[k] = d[k];
    h = sc_reduce256_ct(h);      // :122 (the digest is public; the masked form keeps the kernel's instruction counts
                                 //       independent of it too, which makes the PMC comparison of secret sets exact)
    sc256 e;
    ct_response(&e, &r, &sk, &h);                                          // :124
    u8 *sig = sigs_out + 81 * i;
    st_fp6(sig, rp.x);
    // (R is public, but its flag is formed without branches as well: whether the 64 sort bits of a wave agree would
    //  otherwise show in the kernel's instruction counts and blur the PMC comparison of secret sets)
    u32 lex = 0u, decided = 0u;
#pragma unroll
    for (int k = 5; k >= 0; k--) {
        const u64 c = rp.y.c[k];                       // canonical (ct_to_aff)
        const u32 nz = (u32)(c != 0ull), take = nz & ~decided & 1u;
        lex = take ? (u32)(c > (FP_P - 1) / 2) : lex;
        decided |= nz;
    }
    const u32 r_inf = (u32)f6_is_zero_ct(rj.Z);
    sig[48] = (u8)(r_inf ? 0x80u : (lex << 6));
#pragma unroll
    for (int k = 0; k < 4; k++) st_u64_le(sig + 49 + 8 * k, e.w[k]);
}

// PublicKey::from(&PrivateKey) (src/public.rs:26-32): pk = [sk]G and nothing else -- ONE constant-time base
// multiplication, no nonce, no hash, no response scalar: nothing derived from sk but the public key leaves the lane.
// (Round 4's mirrors obtained the key from the signer with nonce = sk, whose discarded response e = sk (1 - h) is as
// good as the private key: ADVICE r4.)
__global__ void __launch_bounds__(256, 2)
ssa_k_pubkey_ct(const u64 *__restrict__ ctab, const u64 *__restrict__ gtab, const u8 *__restrict__ sks, size_t n,
                u8 *__restrict__ pks_out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    sc256 sk;
    ct_load_scalar(&sk, sks + 32 * i);
    jac pj;
    bool bad_p;
    ct_base_mul(&pj, &bad_p, ctab, &sk);
    if (bad_p) pj = add_base_mul(jac_identity(), gtab, sk);      // sk = 0 (refused by the host form) or a 2^-250 event
    aff pk;
    ct_to_aff(&pk, &pj);
    st_fp6(pks_out + 96 * i, pk.x);
    st_fp6(pks_out + 96 * i + 48, pk.y);
}

// AffinePoint::to_compressed + CompressedPoint::to_bytes (PublicKey::to_bytes, src/public.rs:49-51): x || flag byte
// (bit 7 infinity -- [0; 48] || 0x80, src/public.rs:95-101 --, bit 6 the sort flag of y).  status: 0, or SSA_MALFORMED
// for a limb that is not canonical (an AffinePoint cannot hold one)
SSA_DEV u32 compress_lane(const u8 *__restrict__ pk, bool inf, u8 *__restrict__ out) {
    bool ok = true;
    const fp6 x = ld_fp6(pk, ok);
    const fp6 y = ld_fp6(pk + 48, ok);
    if (inf || !ok) {
        for (int k = 0; k < 48; k++) out[k] = 0;
        out[48] = inf ? 0x80 : 0x00;
        return inf ? ST_OK : ST_MALFORMED;
    }
    for (int k = 0; k < 48; k++) out[k] = pk[k];
    out[48] = f6_lex_largest(y) ? 0x40 : 0x00;
    (void)x;
    return ST_OK;
}

__global__ void __launch_bounds__(256)
ssa_k_compress(const u8 *__restrict__ pks, const u8 *__restrict__ pk_inf, size_t n, u8 *__restrict__ out,
               u8 *__restrict__ status_out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 st = compress_lane(pks + 96 * i, pk_inf && pk_inf[i], out + 49 * i);
    if (status_out) status_out[i] = (u8)st;
}

// KeyedSignature::to_bytes (src/signature.rs:237-245): pk.to_bytes() (49) || signature.to_bytes() (81)
__global__ void __launch_bounds__(256)
ssa_k_pack_keyed(const u8 *__restrict__ pks, const u8 *__restrict__ sigs, size_t n, u8 *__restrict__ keyed) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u8 *rec = keyed + 130 * i;
    // a key pair's public key [sk]G is the identity only for sk = 0, which the signers reduce to and the reference's
    // PrivateKey never holds; its affine bytes are then (0, 0)
    bool zero = true;
    for (int k = 0; k < 96; k++) zero = zero && pks[96 * i + k] == 0;
    (void)compress_lane(pks + 96 * i, zero, rec);
    for (int k = 0; k < 81; k++) rec[49 + k] = sigs[81 * i + k];
}

}  // namespace ssa

// ------------------------------------------------------------------------------------------------------------------
int ssa_internal_sign_vartime(ssa_ctx *ctx, const uint8_t *d_sks, const uint8_t *d_nonces, const uint8_t *d_msgs,
                              const uint64_t *d_msg_off, size_t msg_stride, size_t msg_len, size_t n, uint8_t *d_pks_out,
                              uint8_t *d_sigs_out);     // ssa_api.hip: the throughput signer's launch

// the 98 KB table of the constant-time signer, built once per context -- and COMPLETE before the call returns: a later
// call may run on another stream (ssa_ctx_set_stream), and a failed build must not leave a table marked ready
static int ensure_ctab(ssa_ctx *ctx) {
    if (ctx->ctab_ready) return 0;
    if (ctx->ctab.reserve(CTAB_ROWS * 12 * sizeof(u64))) return SSA_ERR_HIP;
    hipLaunchKernelGGL(ssa_k_ctab, dim3(grid_for(CTAB_ROWS, 64)), dim3(64), 0, ctx->stream, (const u64 *)ctx->d_gtab,
                       (u64 *)ctx->ctab.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->ctab_ready = true;
    return 0;
}

// PublicKey::from(&PrivateKey) for n secret keys (device buffers; the caller guarantees canonical non-zero scalars --
// the host form checks them: a zero or non-canonical value is still computed correctly, [sk mod q]G, but through the
// variable-time fallback)
extern "C" int ssa_pubkey_many_device(ssa_ctx *ctx, const uint8_t *d_sks, size_t n, uint8_t *d_pks_out) {
    if (!ctx || (n && (!d_sks || !d_pks_out)) || n > SSA_MAX_BATCH) return SSA_ERR_ARG;
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(ctx->device));
    if (int rc = ensure_ctab(ctx)) return rc;
    return timed_launch(ctx, "ssa_k_pubkey_ct", [&] {
        hipLaunchKernelGGL(ssa_k_pubkey_ct, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, (const u64 *)ctx->ctab.p,
                           (const u64 *)ctx->d_gtab, d_sks, n, d_pks_out);
    });
}

extern "C" int ssa_keygen_sign_many_ex_device(ssa_ctx *ctx, const uint8_t *d_sks, const uint8_t *d_nonces,
                                              const uint8_t *d_msgs, const uint64_t *d_msg_off, size_t msg_stride,
                                              size_t msg_len, size_t n, uint32_t flags, uint8_t *d_pks_out,
                                              uint8_t *d_sigs_out) {
    if (!ctx || (flags & ~(SSA_FLAG_SIGN_CT | SSA_FLAG_SIGN_KEYED))) return SSA_ERR_ARG;
    const bool keyed = (flags & SSA_FLAG_SIGN_KEYED) != 0;
    if (n && (!d_sks || !d_nonces || !d_sigs_out || (!keyed && !d_pks_out))) return SSA_ERR_ARG;
    if (int rc = check_msgs(d_msgs, d_msg_off, msg_stride, msg_len, n)) return rc;
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(ctx->device));
    // keyed output: the 81-byte signatures (and the keys, when the caller does not want them) pass through the context
    uint8_t *pks = d_pks_out, *sigs = d_sigs_out;
    if (keyed) {
        if (ctx->sg_sigs.reserve(n * 81) || (!d_pks_out && ctx->sg_pks.reserve(n * 96))) return SSA_ERR_HIP;
        sigs = (uint8_t *)ctx->sg_sigs.p;
        if (!d_pks_out) pks = (uint8_t *)ctx->sg_pks.p;
    }
    if (flags & SSA_FLAG_SIGN_CT) {
        if (int rc = ensure_ctab(ctx)) return rc;
        MsgView mv{d_msgs, d_msg_off, msg_stride, msg_len};
        int rc = timed_launch(ctx, "ssa_k_sign_ct", [&] {
            hipLaunchKernelGGL(ssa_k_sign_ct, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, ctx->d_params,
                               (const u64 *)ctx->ctab.p, (const u64 *)ctx->d_gtab, d_sks, d_nonces, mv, n, pks, sigs);
        });
        if (rc) return rc;
    } else if (int rc = ssa_internal_sign_vartime(ctx, d_sks, d_nonces, d_msgs, d_msg_off, msg_stride, msg_len, n, pks, sigs)) {
        return rc;
    }
    if (keyed) {
        hipLaunchKernelGGL(ssa_k_pack_keyed, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, (const u8 *)pks,
                           (const u8 *)sigs, n, d_sigs_out);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

extern "C" int ssa_compress_many_device(ssa_ctx *ctx, const uint8_t *d_pks, const uint8_t *d_pk_inf, size_t n,
                                        uint8_t *d_out, uint8_t *d_status_out) {
    if (!ctx || (n && (!d_pks || !d_out)) || n > SSA_MAX_BATCH) return SSA_ERR_ARG;
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(ctx->device));
    return timed_launch(ctx, "ssa_k_compress", [&] {
        hipLaunchKernelGGL(ssa_k_compress, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, d_pks, d_pk_inf, n, d_out,
                           d_status_out);
    });
}

extern "C" int ssa_compress_many(ssa_ctx *ctx, const uint8_t *pks, const uint8_t *pk_inf, size_t n, uint8_t *out,
                                 uint8_t *status_out) {
    if (!ctx || (n && (!pks || !out))) return SSA_ERR_ARG;
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(ctx->device));
    const void *p, *pi = nullptr;
    if (int rc = stage_up(ctx, ctx->st_pks, pks, n * 96, &p)) return rc;
    if (pk_inf)
        if (int rc = stage_up(ctx, ctx->st_inf, pk_inf, n, &pi)) return rc;
    if (ctx->st_aux.reserve(n * 49) || ctx->st_status.reserve(n + 16)) return SSA_ERR_HIP;
    if (int rc = ssa_compress_many_device(ctx, (const u8 *)p, (const u8 *)pi, n, (u8 *)ctx->st_aux.p, (u8 *)ctx->st_status.p))
        return rc;
    HIP_TRY(hipMemcpyAsync(out, ctx->st_aux.p, n * 49, hipMemcpyDeviceToHost, ctx->stream));
    if (status_out) HIP_TRY(hipMemcpyAsync(status_out, ctx->st_status.p, n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

// secret keys and nonces are canonical non-zero scalars: PrivateKey::new / Scalar::random never yield 0 or a value >= q
// (src/private.rs:49-57); the comparison runs over all 32 bytes whatever they are (the buffers hold secrets)
static bool scalars_canonical_nonzero(const uint8_t *v, size_t n) {
    static const uint8_t q_le[32] = {0xcf, 0xac, 0xd4, 0xae, 0x3e, 0x62, 0x43, 0xd4, 0x22, 0x77, 0x15,
                                     0x30, 0x23, 0xa7, 0x7a, 0x32, 0xb5, 0x37, 0x0a, 0x99, 0x0f, 0xbf,
                                     0x3f, 0x56, 0xd0, 0x22, 0x3f, 0x3b, 0x9b, 0x59, 0xf2, 0x7a};
    unsigned all_ok = 1;
    for (size_t i = 0; i < n; i++) {
        const uint8_t *s = v + 32 * i;
        unsigned nonzero = 0, borrow = 0;
        for (int k = 0; k < 32; k++) {            // s - q: the final borrow says s < q
            nonzero |= s[k];
            const unsigned d = (unsigned)s[k] - (unsigned)q_le[k] - borrow;
            borrow = (d >> 8) & 1u;
        }
        all_ok &= borrow & (unsigned)(nonzero != 0);
    }
    return all_ok != 0;
}

extern "C" int ssa_pubkey_many(ssa_ctx *ctx, const uint8_t *sks, size_t n, uint8_t *pks_out) {
    if (!ctx || (n && (!sks || !pks_out)) || n > SSA_MAX_BATCH) return SSA_ERR_ARG;
    if (n == 0) return 0;
    if (!scalars_canonical_nonzero(sks, n)) return SSA_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    struct Wipe {      // the staged secrets do not outlive the call, whichever way it returns
        ssa_ctx *ctx;
        size_t bytes;
        ~Wipe() {
            if (ctx->st_sigs.p && ctx->st_sigs.cap >= bytes) (void)hipMemsetAsync(ctx->st_sigs.p, 0, bytes, ctx->stream);
            (void)hipStreamSynchronize(ctx->stream);
        }
    } wipe{ctx, n * 32};
    const void *p_sk;
    if (int rc = stage_up(ctx, ctx->st_sigs, sks, n * 32, &p_sk)) return rc;
    if (ctx->st_aux.reserve(n * 96)) return SSA_ERR_HIP;
    if (int rc = ssa_pubkey_many_device(ctx, (const u8 *)p_sk, n, (u8 *)ctx->st_aux.p)) return rc;
    HIP_TRY(hipMemcpyAsync(pks_out, ctx->st_aux.p, n * 96, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int ssa_keygen_sign_many_ex(ssa_ctx *ctx, const uint8_t *sks, const uint8_t *nonces, const uint8_t *msgs,
                                       const uint64_t *msg_off, size_t msg_stride, size_t msg_len, size_t n,
                                       uint32_t flags, uint8_t *pks_out, uint8_t *sigs_out) {
    if (!ctx || (flags & ~(SSA_FLAG_SIGN_CT | SSA_FLAG_SIGN_KEYED))) return SSA_ERR_ARG;
    const bool keyed = (flags & SSA_FLAG_SIGN_KEYED) != 0;
    if (n && (!sks || !nonces || !sigs_out || (!keyed && !pks_out))) return SSA_ERR_ARG;
    if (int rc = check_msgs(msgs, msg_off, msg_stride, msg_len, n)) return rc;
    if (n == 0) return 0;
    if (!scalars_canonical_nonzero(sks, n) || !scalars_canonical_nonzero(nonces, n)) return SSA_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    // the staged secrets do not outlive the call, whichever way it returns
    struct Wipe {
        ssa_ctx *ctx;
        size_t bytes;
        ~Wipe() {
            if (ctx->st_sigs.p && ctx->st_sigs.cap >= bytes) (void)hipMemsetAsync(ctx->st_sigs.p, 0, bytes, ctx->stream);
            if (ctx->st_pks.p && ctx->st_pks.cap >= bytes) (void)hipMemsetAsync(ctx->st_pks.p, 0, bytes, ctx->stream);
            (void)hipStreamSynchronize(ctx->stream);
        }
    } wipe{ctx, n * 32};
    StagedInputs s;
    const void *p_sk, *p_nonce;
    if (int rc = stage_up(ctx, ctx->st_sigs, sks, n * 32, &p_sk)) return rc;
    if (int rc = stage_up(ctx, ctx->st_pks, nonces, n * 32, &p_nonce)) return rc;
    if (int rc = stage_msgs(ctx, msgs, msg_off, msg_stride, msg_len, n, s)) return rc;
    const size_t sig_bytes = keyed ? 130 : 81;
    if (ctx->st_aux.reserve(n * 96) || ctx->st_aux2.reserve(n * sig_bytes)) return SSA_ERR_HIP;
    if (int rc = ssa_keygen_sign_many_ex_device(ctx, (const u8 *)p_sk, (const u8 *)p_nonce, s.msgs, s.off, msg_stride,
                                                msg_len, n, flags, (u8 *)ctx->st_aux.p, (u8 *)ctx->st_aux2.p))
        return rc;
    if (pks_out) HIP_TRY(hipMemcpyAsync(pks_out, ctx->st_aux.p, n * 96, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(sigs_out, ctx->st_aux2.p, n * sig_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}
