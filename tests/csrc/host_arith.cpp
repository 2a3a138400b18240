// Host-side compile of the device arithmetic headers (tests only).  The limb logic of
// fp.hpp / fp6.hpp / curve.hpp / rescue.hpp is plain integer C++ apart from one inline-asm
// multiply-accumulate that has a C twin, so it can be unit-tested against the oracle on a
// machine without a GPU.  The shipped library never executes this code on the CPU.
//   hipcc --cuda-host-only -x hip -O2 -shared -fPIC host_arith.cpp -o libhost_arith.so
#define SSA_NO_KERNELS 1
#define SSA_NO_COOP 1
#include <cstring>
#include "../../schnorr-sig_amd/csrc/ssa_kernels.hpp"

using namespace ssa;

extern "C" {
uint64_t ha_fp_mul(uint64_t a, uint64_t b) { return fp_canon(fp_mul(a, b)); }
uint64_t ha_fp_sqr3(uint64_t a) { return fp_canon(fp_sqr3(a)); }
uint64_t ha_fp_add(uint64_t a, uint64_t b) { return fp_canon(fp_add(a, b)); }
uint64_t ha_fp_sub(uint64_t a, uint64_t b) { return fp_canon(fp_sub(a, b)); }
uint64_t ha_fp_inv(uint64_t a) { return fp_canon(fp_inv(a)); }
uint64_t ha_fp_mul_small(uint64_t a, uint32_t k) { return fp_canon(fp_mul_small(a, k)); }
uint64_t ha_inv_sbox(uint64_t a) { return fp_canon(inv_sbox(a)); }
uint64_t ha_inv_sbox2(uint64_t a, uint64_t b, uint64_t *ob) {
    uint64_t t[3] = {a, b, a};
    uint64_t *p[SSA_FP_CHAINS] = {t, t + 1, t + 2};
    sbox_block<true>(p);
    *ob = fp_canon(t[1]);
    return fp_canon(t[0]);
}
static fp6 ld(const uint64_t *p) {
    fp6 r;
    for (int i = 0; i < 6; i++) r.c[i] = p[i];
    return r;
}
static void st(uint64_t *p, const fp6 &a) {
    fp6 c = f6_canon(a);
    for (int i = 0; i < 6; i++) p[i] = c.c[i];
}
void ha_f6_mul(const uint64_t *a, const uint64_t *b, uint64_t *o) { st(o, f6_mul(ld(a), ld(b))); }
void ha_f6_sqr(const uint64_t *a, uint64_t *o) { st(o, f6_sqr(ld(a))); }
void ha_f6_inv(const uint64_t *a, uint64_t *o) { st(o, f6_inv(ld(a))); }
int ha_f6_sqrt(const uint64_t *a, uint64_t *o) {
    fp6 r = f6_zero();
    bool ok = f6_sqrt(ld(a), r);
    st(o, r);
    return ok;
}
int ha_decompress(const uint8_t *c49, uint64_t *o12, int *inf) {
    aff p;
    bool is_inf;
    u32 s = decompress_lane(c49, p, is_inf);
    st(o12, p.x);
    st(o12 + 6, p.y);
    *inf = is_inf;
    return (int)s;
}
// [k]P through build_ptab + mul_ptab; tab must hold ha_ptab_words() u64
int ha_ptab_words(void) { return PTAB_ENTRIES * PTAB_ENTRY_U64; }
int ha_mul_ptab(const uint64_t *k4, const uint64_t *p12, int inf, uint64_t *tab, uint64_t *o12) {
    sc256 k;
    for (int i = 0; i < 4; i++) k.w[i] = k4[i];
    aff p;
    p.x = ld(p12);
    p.y = ld(p12 + 6);
    build_ptab(tab, p, inf != 0);
    jac r = mul_ptab(tab, k);
    aff a = jac_to_aff(r);
    st(o12, a.x);
    st(o12 + 6, a.y);
    return jac_is_identity(r);
}
int ha_point_add(const uint64_t *a12, int a_inf, const uint64_t *b12, int b_inf, int general, uint64_t *o12) {
    aff p, q;
    p.x = ld(a12); p.y = ld(a12 + 6);
    q.x = ld(b12); q.y = ld(b12 + 6);
    jac pj = a_inf ? jac_identity() : jac_from_aff(p);
    jac r;
    if (b_inf) r = pj;
    else if (general) r = jac_add(pj, jac_from_aff(q));
    else r = jac_madd(pj, q);
    aff a = jac_to_aff(r);
    st(o12, a.x);
    st(o12 + 6, a.y);
    return jac_is_identity(r);
}
// hash_field with the LDS plane emulated by a host array (one plane: the MDS layer works in place)
// flags != 0 forces the small-MDS path (the library derives the flag itself at ctx_create)
void ha_hash_field(const void *params, int flags, const uint64_t *felts, uint32_t n, uint64_t *digest) {
    static uint64_t planes[RS_LDS_U64];
    uint64_t d[4];
    DevParams prm;
    memcpy(&prm, params, sizeof prm);
    prm.flags = (u32)flags;
    sponge_hash(planes, planes, &prm, n,
                [&](u32 idx) -> u64 { return felts[idx]; }, d);
    for (int i = 0; i < 4; i++) digest[i] = d[i];
}
void ha_hash_message(const void *params, const uint8_t *sig, const uint8_t *pk, const uint8_t *msg,
                     uint32_t len, uint64_t *digest) {
    static uint64_t planes[RS_LDS_U64];
    bool ok = true;
    fp6 rx = ld_fp6(sig, ok), px = ld_fp6(pk, ok);
    uint64_t d[4];
    hash_message_lane(planes, planes, (const DevParams *)params, rx, px,
                      ld_u64_le(pk + 48), msg, len, d);
    for (int i = 0; i < 4; i++) digest[i] = d[i];
}
void ha_sc_mul_sub(const uint64_t *r4, const uint64_t *sk4, const uint64_t *h4, uint64_t *e4) {
    sc256 r, sk, h;
    for (int i = 0; i < 4; i++) { r.w[i] = r4[i]; sk.w[i] = sk4[i]; h.w[i] = h4[i]; }
    sc256 e = sc_add_mod(r, sc_neg_mod(sc_mul_mod(sk, h)));
    for (int i = 0; i < 4; i++) e4[i] = e.w[i];
}
void ha_sc_mul(const uint64_t *a4, const uint64_t *b4, uint64_t *o4) {
    sc256 a, b;
    for (int i = 0; i < 4; i++) { a.w[i] = a4[i]; b.w[i] = b4[i]; }
    const sc256 r = sc_mul_mod(a, b);
    for (int i = 0; i < 4; i++) o4[i] = r.w[i];
}
void ha_sc_reduce(const uint64_t *a4, uint64_t *o4) {
    sc256 a;
    for (int i = 0; i < 4; i++) a.w[i] = a4[i];
    a = sc_reduce256(a);
    for (int i = 0; i < 4; i++) o4[i] = a.w[i];
}
}
