/* CPU oracle for the toposware/schnorr-sig verification path -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the reference algorithm (see schnorr_oracle.c for the
 * reference file:line each function follows).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (schnorr-sig_amd/) never links, imports or calls it.
 *
 * PARITY STATUS: "parity unpinned" for the Rescue-Prime constants and the generator
 * (un-vendored third-party crates cheetah/hash, reference Cargo.toml:16,18); pinned
 * against every reference-owned fixture (tests/test_oracle_fixtures.py).
 */
#ifndef SCHNORR_ORACLE_H
#define SCHNORR_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status vocabulary: reference src/error.rs:13-18 (+3 where the reference panics) */
#define SO_OK 0
#define SO_INVALID_PUBLIC_KEY 1
#define SO_INVALID_SIGNATURE 2
#define SO_MALFORMED 3

/* Load the 2816-byte parameter blob (MDS, ARK, sponge layout, generator). */
int so_init(const uint8_t *blob, size_t len);

/* field-level entry points (unit parity with the HIP path) */
void so_fp6_mul(const uint64_t a[6], const uint64_t b[6], uint64_t out[6]);
void so_fp6_sqr(const uint64_t a[6], uint64_t out[6]);
int so_fp6_inv(const uint64_t a[6], uint64_t out[6]);
int so_fp6_sqrt(const uint64_t a[6], uint64_t out[6]);

/* curve: affine in / affine out, *_inf = 1 for the identity */
void so_point_mul(const uint8_t k32[32], const uint64_t px[6], const uint64_t py[6], int p_inf,
                  uint64_t ox[6], uint64_t oy[6], int *o_inf);
void so_point_add(const uint64_t ax[6], const uint64_t ay[6], int a_inf, const uint64_t bx[6],
                  const uint64_t by[6], int b_inf, uint64_t ox[6], uint64_t oy[6], int *o_inf);
int so_is_torsion_free(const uint64_t px[6], const uint64_t py[6], int p_inf);
int so_on_curve(const uint64_t px[6], const uint64_t py[6]);

/* Rescue-Prime 64/12/8 */
void so_rescue_permutation(uint64_t state[12]);
void so_hash_field(const uint64_t *felts, size_t n, uint64_t digest[4]);
void so_hash_message(const uint8_t rx48[48], const uint8_t pk96[96], const uint8_t *msg, size_t len,
                     uint8_t out32[32]);
/* digest bytes -> canonical scalar mod q (32 bytes LE) */
void so_scalar_from_digest(const uint8_t h32[32], uint8_t out32[32]);

/* schnorr-sig */
void so_keygen(const uint8_t sk32[32], uint8_t pk96[96], int *pk_inf);
int so_sign(const uint8_t sk32[32], const uint8_t nonce32[32], const uint8_t pk96[96],
            const uint8_t *msg, size_t len, uint8_t sig81[81]);
/* flags: bit 0 = subgroup check (Signature::verify), bit 3 = verify_batch's flag-byte semantics */
int so_verify(const uint8_t sig81[81], const uint8_t pk96[96], int pk_inf, const uint8_t *msg,
              size_t len, int flags);
/* n independent Signature::verify calls; msg i = msgs + off[i] .. off[i+1] (off != NULL) or
 * msgs + i*stride, length msg_len.  threads <= 0 -> all hardware threads. */
void so_verify_many(const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf,
                    const uint8_t *msgs, const uint64_t *off, size_t stride, size_t msg_len,
                    size_t n, int flags, int threads, uint8_t *status);
void so_keygen_sign_many(const uint8_t *sks, const uint8_t *nonces, const uint8_t *msgs,
                         const uint64_t *off, size_t stride, size_t msg_len, size_t n, int threads,
                         uint8_t *pks, uint8_t *sigs);
/* verify_batch exactly as src/batch.rs: random-linear-combination + MSM, x-only compare.
 * coeffs = n x 32-byte canonical scalars standing in for Scalar::random(rng). */
int so_verify_batch_msm(const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf, const uint8_t *msgs,
                        const uint64_t *off, size_t stride, size_t msg_len, size_t n,
                        const uint8_t *coeffs, int threads);
/* The same two calls with the algorithms a CPU library would use (schnorr_oracle_fast.inc: lazy Fp6 products, width-5
 * NAF over affine odd multiples, a fixed-base table for G, a bucket MSM) -- the TIMING path of bench.py's cpu_baseline
 * leg, cross-checked against the plain functions above on every sample it times.  Identical results on every input. */
void so_verify_many_fast(const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf,
                         const uint8_t *msgs, const uint64_t *off, size_t stride, size_t msg_len,
                         size_t n, int flags, int threads, uint8_t *status);
int so_verify_batch_msm_fast(const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf, const uint8_t *msgs,
                             const uint64_t *off, size_t stride, size_t msg_len, size_t n,
                             const uint8_t *coeffs, int threads);
int so_hw_threads(void);
int so_decompress(const uint8_t c49[49], uint8_t pk96[96], int *pk_inf);
void so_compress(const uint8_t pk96[96], int pk_inf, uint8_t c49[49]);

#ifdef __cplusplus
}
#endif
#endif
