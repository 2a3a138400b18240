/* Sanitizer self-test of the CPU oracle ("-fsanitize=address,undefined CPU test build").
 * TEST INFRASTRUCTURE: built by `make -C oracle asan` into oracle/_asan/ and run by the CPU suite
 * (tests/test_sanitizers.py); never on the GPU box.  It drives every exported entry point of schnorr_oracle.c over
 * honest, corrupted, ragged and degenerate inputs; AddressSanitizer / UBSan abort on the first finding, the checks
 * below catch plain wrong answers.   usage: oracle_selftest <params blob> */
#include "schnorr_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(c)                                                     \
    do {                                                             \
        if (!(c)) {                                                  \
            printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c);       \
            return 1;                                                \
        }                                                            \
    } while (0)

static uint64_t rng_state = 0x5C4E0333ull;
static uint64_t splitmix(void) {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static void fill(uint8_t *p, size_t n) {
    for (size_t i = 0; i < n; i++) p[i] = (uint8_t)splitmix();
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    uint8_t blob[2816];
    size_t got = fread(blob, 1, sizeof blob, f);
    fclose(f);
    CHECK(got == sizeof blob);
    CHECK(so_init(blob, sizeof blob - 1) != 0);      /* wrong length refused */
    CHECK(so_init(blob, sizeof blob) == 0);

    enum { N = 9, ML = 24 };
    /* exact-size heap blocks: an out-of-bounds read or write of any entry point lands in a red zone */
    uint8_t *sks = malloc(N * 32), *nonces = malloc(N * 32), *msgs = malloc(N * ML), *pks = malloc(N * 96),
            *sigs = malloc(N * 81), *st = malloc(N), *inf = calloc(N, 1), *co = malloc(N * 32);
    fill(sks, N * 32); fill(nonces, N * 32); fill(msgs, N * ML); fill(co, N * 32);
    for (int i = 0; i < N; i++) {
        sks[32 * i + 31] &= 0x3f; sks[32 * i] |= 1;
        nonces[32 * i + 31] &= 0x3f; nonces[32 * i] |= 1;
        co[32 * i + 31] &= 0x3f;
    }
    so_keygen_sign_many(sks, nonces, msgs, NULL, ML, ML, N, 2, pks, sigs);
    so_verify_many(sigs, pks, inf, msgs, NULL, ML, ML, N, 1, 2, st);
    for (int i = 0; i < N; i++) CHECK(st[i] == SO_OK);
    CHECK(so_verify_batch_msm(sigs, pks, inf, msgs, NULL, ML, ML, N, co, 2) == SO_OK);
    CHECK(so_verify_batch_msm(sigs, pks, NULL, msgs, NULL, ML, ML, 0, co, 1) == SO_OK);   /* empty batch */
    /* the timing path (schnorr_oracle_fast.inc): the same answers out of exact-size blocks */
    so_verify_many_fast(sigs, pks, inf, msgs, NULL, ML, ML, N, 1, 2, st);
    for (int i = 0; i < N; i++) CHECK(st[i] == SO_OK);
    CHECK(so_verify_batch_msm_fast(sigs, pks, inf, msgs, NULL, ML, ML, N, co, 2) == SO_OK);
    CHECK(so_verify_batch_msm_fast(sigs, pks, NULL, msgs, NULL, ML, ML, 0, co, 1) == SO_OK);
    CHECK(so_verify_batch_msm_fast(sigs, pks, inf, msgs, NULL, ML, ML, 1, co, 1) == SO_OK);

    /* single-signature entry points agree with the batch ones */
    uint8_t pk1[96], sig1[81];
    int pinf = 0;
    so_keygen(sks, pk1, &pinf);
    CHECK(!pinf && memcmp(pk1, pks, 96) == 0);
    CHECK(so_sign(sks, nonces, pk1, msgs, ML, sig1) == 0 && memcmp(sig1, sigs, 81) == 0);
    CHECK(so_verify(sig1, pk1, 0, msgs, ML, 1) == SO_OK);
    CHECK(so_verify(sig1, pk1, 0, msgs, ML, 1 | 8) == SO_OK);
    CHECK(so_verify(sig1, pk1, 0, msgs, ML - 1, 1) == SO_INVALID_SIGNATURE);
    CHECK(so_verify(sig1, pk1, 0, NULL, 0, 0) == SO_INVALID_SIGNATURE);                   /* empty message */

    /* corruptions: e bit, message bit, swapped key, non-canonical limb, e >= q, flag byte */
    sigs[81 * 1 + 49] ^= 1;
    msgs[ML * 2 + 5] ^= 0x40;
    memcpy(pks + 96 * 3, pks + 96 * 4, 96);
    memset(sigs + 81 * 5, 0xff, 8);
    memset(sigs + 81 * 6 + 49, 0xff, 32);
    so_verify_many(sigs, pks, inf, msgs, NULL, ML, ML, N, 0, 3, st);
    CHECK(st[0] == 0 && st[1] == 2 && st[2] == 2 && st[3] == 2 && st[4] == 0 && st[5] == 3 && st[6] == 3);
    CHECK(so_verify_batch_msm(sigs, pks, inf, msgs, NULL, ML, ML, N, co, 2) == SO_MALFORMED);
    {
        uint8_t *st2 = malloc(N);
        so_verify_many_fast(sigs, pks, inf, msgs, NULL, ML, ML, N, 0, 3, st2);
        CHECK(memcmp(st, st2, N) == 0);
        so_verify_many_fast(sigs, pks, inf, msgs, NULL, ML, ML, N, 1 | 8, 1, st2);
        so_verify_many(sigs, pks, inf, msgs, NULL, ML, ML, N, 1 | 8, 1, st);
        CHECK(memcmp(st, st2, N) == 0);
        free(st2);
        CHECK(so_verify_batch_msm_fast(sigs, pks, inf, msgs, NULL, ML, ML, N, co, 2) == SO_MALFORMED);
    }
    uint8_t keep = sigs[81 * 7 + 48];
    sigs[81 * 7 + 48] = 0xff;                                                              /* src/public.rs:150-156 */
    CHECK(so_verify(sigs + 81 * 7, pks + 96 * 7, 0, msgs + ML * 7, ML, 8) == SO_MALFORMED);
    CHECK(so_verify(sigs + 81 * 7, pks + 96 * 7, 0, msgs + ML * 7, ML, 0) == SO_OK);       /* verify ignores byte 48 */
    sigs[81 * 7 + 48] = keep;

    /* ragged messages through offsets, zero-length ones included; identity key */
    uint64_t off[N + 1];
    off[0] = 0;
    for (int i = 0; i < N; i++) off[i + 1] = off[i] + (uint64_t)((i * 5) % 23);
    uint8_t *flat = malloc(off[N] + 1);
    fill(flat, off[N] + 1);
    so_keygen_sign_many(sks, nonces, flat, off, 0, 0, N, 1, pks, sigs);
    inf[8] = 1;
    so_verify_many(sigs, pks, inf, flat, off, 0, 0, N, 1 | 8, 0, st);
    for (int i = 0; i < 8; i++) CHECK(st[i] == SO_OK);
    CHECK(st[8] == SO_INVALID_SIGNATURE);
    so_verify_many_fast(sigs, pks, inf, flat, off, 0, 0, N, 1 | 8, 0, st);
    for (int i = 0; i < 8; i++) CHECK(st[i] == SO_OK);
    CHECK(st[8] == SO_INVALID_SIGNATURE);
    CHECK(so_verify_batch_msm_fast(sigs, pks, inf, flat, off, 0, 0, N, co, 1) ==
          so_verify_batch_msm(sigs, pks, inf, flat, off, 0, 0, N, co, 1));
    inf[8] = 0;

    /* compression round trip, identity encoding (src/public.rs:95-101), undecodable x = 0 (:115-120) */
    uint8_t c49[49], back[96];
    so_compress(pks, 0, c49);
    CHECK(so_decompress(c49, back, &pinf) == 1 && !pinf && memcmp(back, pks, 96) == 0);
    memset(c49, 0, 49);
    c49[48] = 0x80;
    CHECK(so_decompress(c49, back, &pinf) == 1 && pinf == 1);
    c49[48] = 0;
    CHECK(so_decompress(c49, back, &pinf) == 0);

    /* field / curve / sponge entry points on edge values */
    const uint64_t P = 0xffffffff00000001ull;
    uint64_t a[6] = {P - 1, 0, 1, 0xffffffffull, 0x100000000ull, P - 0x100000000ull}, b[6], c[6], one[6] = {1, 0, 0, 0, 0, 0};
    CHECK(so_fp6_inv(a, b) == 1);
    so_fp6_mul(a, b, c);
    CHECK(memcmp(c, one, sizeof one) == 0);
    so_fp6_sqr(a, b);
    so_fp6_mul(a, a, c);
    CHECK(memcmp(b, c, sizeof b) == 0);
    CHECK(so_fp6_sqrt(b, c) == 1);
    so_fp6_sqr(c, one);
    CHECK(memcmp(one, b, sizeof b) == 0);
    uint64_t zero[6] = {0};
    CHECK(so_fp6_inv(zero, b) == 0);
    uint64_t px[6], py[6], qx[6], qy[6], rx[6], ry[6];
    memcpy(px, pks, 48); memcpy(py, pks + 48, 48);
    CHECK(so_on_curve(px, py) == 1 && so_is_torsion_free(px, py, 0) == 1);
    int oinf = 0;
    uint8_t k32[32] = {2};
    so_point_mul(k32, px, py, 0, qx, qy, &oinf);
    CHECK(!oinf);
    so_point_add(px, py, 0, px, py, 0, rx, ry, &oinf);                                    /* doubling branch */
    CHECK(!oinf && memcmp(qx, rx, 48) == 0 && memcmp(qy, ry, 48) == 0);
    for (int i = 0; i < 6; i++) ry[i] = py[i] ? P - py[i] : 0;
    so_point_add(px, py, 0, px, ry, 0, rx, qy, &oinf);                                    /* P + (-P) */
    CHECK(oinf == 1);
    so_point_add(px, py, 0, zero, zero, 1, rx, ry, &oinf);                                /* P + O */
    CHECK(!oinf && memcmp(rx, px, 48) == 0);
    uint64_t felts[27], d0[4], d1[4];
    for (int i = 0; i < 27; i++) felts[i] = splitmix() % P;
    for (size_t n = 0; n <= 27; n++) so_hash_field(felts, n, d0);                         /* every block raggedness */
    so_hash_field(felts, 8, d0);
    so_hash_field(felts, 9, d1);
    CHECK(memcmp(d0, d1, sizeof d0) != 0);
    uint64_t state[12] = {0};
    so_rescue_permutation(state);
    uint8_t h32[32], s32[32];
    memset(h32, 0xff, 32);
    so_scalar_from_digest(h32, s32);
    CHECK(s32[31] < 0x7b);
    CHECK(so_hw_threads() >= 1);

    free(sks); free(nonces); free(msgs); free(pks); free(sigs); free(st); free(inf); free(co); free(flat);
    printf("oracle_selftest ok\n");
    return 0;
}
