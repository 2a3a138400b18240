/* schnorr_sig_amd.h -- C ABI of the MI355X-native batched Schnorr verification engine.
 *
 * Drop-in boundary for the verification path of toposware/schnorr-sig.  The reference has
 * no FFI (`#![deny(unsafe_code)]`, src/lib.rs:152); these entry points are what a Rust
 * `-sys` shim would bind to replace, one for one:
 *
 *   ssa_verify              <- Signature::verify            src/signature.rs:181-205
 *                              (KeyPair::verify_signature   src/signature.rs:159-165,
 *                               PublicKey::verify_signature src/signature.rs:170-176,
 *                               KeyedSignature::verify      src/signature.rs:232-234)
 *   ssa_verify_batch        <- verify_batch                 src/batch.rs:31-50
 *   ssa_verify_batch_msm    <- verify_batch, the reference's own algorithm (random linear
 *                              combination + 2n-point MSM)     src/batch.rs:56-130
 *   ssa_verify_many         <- n x Signature::verify (the per-signature accept/reject vector
 *                              BASELINE.json's north_star asks for)
 *   ssa_hash_message_many   <- hash_message                 src/signature.rs:274-306
 *   ssa_rescue_hash_many    <- RescueHash::hash_field       src/signature.rs:303
 *   ssa_verify_keyed_many   <- KeyedSignature::{from_bytes, verify}  src/signature.rs:232-271
 *   ssa_decompress_many     <- PublicKey::from_bytes / AffinePoint::from_compressed
 *                                                           src/public.rs:54-56, src/batch.rs:104
 *   ssa_keygen_sign_many    <- KeyPair::new / KeyPair::sign src/keypair.rs:57-65,
 *                                                           src/signature.rs:114-129
 *   ssa_keygen_sign_many_ex <- the same, constant-time (SSA_FLAG_SIGN_CT) and / or as KeyedSignature records
 *                              (sign_and_bind_pkey + KeyedSignature::to_bytes, src/signature.rs:132-156,237-245)
 *   ssa_pubkey_many         <- PublicKey::from(&PrivateKey) src/public.rs:26-32
 *   ssa_compress_many       <- PublicKey::to_bytes          src/public.rs:49-51
 *   status codes            <- SignatureError               src/error.rs:13-18
 *   record sizes            <- src/constants.rs:12-30
 *
 * Data layout (all little-endian, canonical limbs -- pinned by the reference's fixtures,
 * src/signature.rs:387-404, :430-460):
 *   signature  81 B = R.x c0..c5 (6 x u64) | flag byte (ignored by verify, src/signature.rs:186)
 *                     | e (32 B, < q)
 *   public key 96 B = affine x c0..c5 | y c0..c5   (PublicKey.0 is an AffinePoint in memory,
 *                     src/public.rs:24); the 49-B compressed wire form (src/public.rs:49-56) goes through
 *                     ssa_decompress_many / ssa_compress_many, or directly into ssa_verify_keyed_many
 *   messages   either a dense array with a fixed stride, or concatenated bytes + (n+1) offsets
 *
 * Ownership: the caller owns every buffer for the duration of the call; the library keeps
 * no pointer.  A context owns its device memory and one HIP stream; calls on one context
 * must be serialised by the caller, several contexts may be used concurrently.
 * No entry point aborts: inputs on which the reference would panic (non-canonical limbs,
 * src/signature.rs:186; e >= q) get status SSA_MALFORMED.
 */
#ifndef SCHNORR_SIG_AMD_H
#define SCHNORR_SIG_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSA_SIGNATURE_LENGTH 81   /* src/constants.rs:27 */
#define SSA_AFFINE_PK_LENGTH 96   /* 2 x BASEFIELD_LENGTH, src/constants.rs:18 */
#define SSA_SCALAR_LENGTH 32      /* src/constants.rs:12 */
#define SSA_PUBLIC_KEY_LENGTH 49  /* compressed wire form, src/constants.rs:21 */
#define SSA_DIGEST_LENGTH 32      /* hash_message output, src/signature.rs:274 */
#define SSA_PARAMS_LENGTH 2816

/* per-signature status (src/error.rs:13-18) */
#define SSA_OK 0
#define SSA_INVALID_PUBLIC_KEY 1  /* SignatureError::InvalidPublicKey */
#define SSA_INVALID_SIGNATURE 2   /* SignatureError::InvalidSignature */
#define SSA_MALFORMED 3           /* the reference would panic on this input */

/* API return codes (negative = the call itself failed) */
#define SSA_ERR_ARG (-1)
#define SSA_ERR_HIP (-2)
#define SSA_ERR_PARAMS (-3)
#define SSA_ERR_NO_DEVICE (-4)

/* flags */
#define SSA_MAX_BATCH ((size_t)1 << 30)   /* signatures per call; larger n returns SSA_ERR_ARG.  Device memory does not grow
                                             with n, for device-pointer AND host-buffer entry points: workspaces and
                                             staging are sized for slices of 2^20 lanes (4.4 GB of tables + 0.3 GB of
                                             staging per slice in flight, at most two; SSA_LANE_SLICE), resp. 2^23
                                             signatures of the MSM form (SSA_MSM_SLICE).  The host-buffer entry points
                                             copy the caller's bytes through page-locked bounce buffers of the library's,
                                             the size of one slice's inputs (0.3 GB, resp. 2.2 GB for a full MSM slice);
                                             the caller's memory is never registered with the runtime */
#define SSA_FLAG_FORCE_LANE 2u    /* always the throughput kernels (one signature per lane) */
#define SSA_FLAG_FORCE_COOP 4u    /* always the low-latency kernel (one wave per signature) */
#define SSA_FLAG_CHECK_TORSION 1u /* Signature::verify semantics (src/signature.rs:182-184);
                                     off = verify_batch semantics (src/batch.rs has no check) */
#define SSA_FLAG_SIG_FLAG_BYTE 8u /* verify_batch semantics for byte 48 of the signature: the reference decompresses R
                                     with its flag byte (from_compressed(&sig.x).unwrap(), src/batch.rs:104), so a
                                     signature only verifies for the R the flags select: wrong sort bit (bit 6) or an
                                     infinity bit (bit 7) that does not match the recomputed R -> SSA_INVALID_SIGNATURE,
                                     an undecodable flag byte -> SSA_MALFORMED (the reference panics).  Off =
                                     Signature::verify, which ignores the byte (src/signature.rs:186) */

typedef struct ssa_ctx ssa_ctx;

/* Parameter blob (Rescue-Prime instance + generator), see schnorr-sig_amd/params/gen_params.py.
 *   char magic[8] = "SSAPARM1"; u32 n_rounds, rate_off; i32 cap_len_idx; u32 pad_mode,
 *   digest_off, flags; u64 mds[144]; u64 ark1[8][12]; u64 ark2[8][12]; u64 gen_x[6], gen_y[6]
 * params == NULL selects the built-in default blob -- the builder's own Rescue constants and generator, NOT
 * upstream's (they live in un-vendored crates; DESIGN.md "parity unpinned"): such a context rejects every genuine
 * toposware signature and ssa_ctx_uses_default_params() returns 1 for it.  tools/blob_from_upstream.py builds the
 * blob from upstream's constants.  The generator is validated on the device (on the curve, [q]G == O):
 * SSA_ERR_PARAMS otherwise.  Device memory: the fixed-base comb table of the generator (the reference's const
 * BASEPOINT_TABLE, src/signature.rs:20,116, src/batch.rs:98-100) is sized by the context -- see ssa_ctx_create_ex --,
 * one per device, generator and geometry, shared by all the contexts of the process; the per-lane workspaces grow to
 * 4.4 GB with the first large batch (slices of 2^20 lanes, whatever the batch size), twice that once a call of more than
 * one slice has used the second internal stream. */
int ssa_ctx_create(ssa_ctx **out, int device, const void *params, size_t params_len);
/* The same with the speed-for-memory trade chosen by the caller.
 *   gtab_bits         window width of the comb for G: 24 (11 windows, 17.7 GB: [e]G = 11 additions), 22 (12, 4.8 GB),
 *                     20 (13, 1.3 GB) or 16 (16, 100 MB; ssa_k_verify +1.9 %); 0 = the widest whose table fits the budget
 *                     (environment: SSA_GTAB_BITS).  Results are identical for every width.
 *   hbm_budget_bytes  what the tables that only buy speed may take on the device -- the comb for G, the per-key combs of
 *                     an SSA_KEYSET_AUTO key set; 0 = a tenth of the memory that is free when the context is created
 *                     (environment: SSA_HBM_BUDGET_MB).
 * An allocation that fails is not an error while a smaller table exists: the context falls back width by width down
 * to the 100 MB comb (ssa_ctx_info says what it got). */
int ssa_ctx_create_ex(ssa_ctx **out, int device, const void *params, size_t params_len, uint32_t gtab_bits,
                      uint64_t hbm_budget_bytes);
/* What the context holds on the device: out[0] window bits and out[1] windows of the comb for G, out[2] its bytes,
 * out[3] bytes of this context's workspaces and staging buffers as reserved so far, out[4] lanes per slice of the
 * per-lane kernels, out[5] signatures per slice of the MSM form, out[6] the HBM budget, out[7] 1 when calls of more than
 * one slice alternate their slices between two internal streams (SSA_TWO_STREAMS=0 turns that off). */
int ssa_ctx_info(const ssa_ctx *ctx, uint64_t out[8]);
/* 1 when the context was created from the built-in blob (parity with upstream unpinned), 0 for a caller-supplied one */
int ssa_ctx_uses_default_params(const ssa_ctx *ctx);
void ssa_ctx_destroy(ssa_ctx *ctx);
const char *ssa_strerror(int rc);
/* the built-in blob (SSA_PARAMS_LENGTH bytes) */
const void *ssa_default_params(void);
/* Ordering against a stream of the caller WITHOUT a host synchronisation.  Every *_device entry point only ENQUEUES
 * on the context's stream (its own non-blocking stream unless ssa_ctx_set_stream changed it): its outputs are valid for
 * other streams only after one of ssa_ctx_sync, ssa_ctx_stream_release, or a shared stream.
 *   ssa_ctx_stream_release(ctx, s): stream s waits for everything enqueued on the context so far (publish outputs to s);
 *   ssa_ctx_stream_acquire(ctx, s): the context's stream waits for everything enqueued on s so far (inputs written on s,
 *                                   e.g. by a collective, are visible to the next call on the context). */
int ssa_ctx_stream_release(ssa_ctx *ctx, void *consumer_hip_stream);
int ssa_ctx_stream_acquire(ssa_ctx *ctx, void *producer_hip_stream);
/* make the context issue its work on an existing hipStream_t.  NULL = back to the context's own (non-blocking)
 * stream; to select the legacy null stream pass HIP's own handle for it, hipStreamLegacy (or hipStreamPerThread). */
int ssa_ctx_set_stream(ssa_ctx *ctx, void *hip_stream);
/* average duration (ms) of the `ssa_k_verify` launches since the last call, measured with
 * hipEvents on the context's stream when profiling is on; resets the statistics. */
int ssa_ctx_enable_timing(ssa_ctx *ctx, int on);
int ssa_ctx_read_timing(ssa_ctx *ctx, const char *kernel, double *avg_ms, uint64_t *launches);

/* ---- host-buffer entry points (what the Rust shim binds) ----------------------------- */

/* Signature::verify for one signature.  Returns a status code. */
int ssa_verify(ssa_ctx *ctx, const uint8_t sig[SSA_SIGNATURE_LENGTH],
               const uint8_t pk[SSA_AFFINE_PK_LENGTH], const uint8_t *msg, size_t msg_len,
               uint32_t flags);

/* n independent verifications; status_out[i] in {0,1,2,3}; *n_fail_out = #(status != 0).
 * msg_off != NULL: message i = msgs[msg_off[i] .. msg_off[i+1]);
 * msg_off == NULL: message i = msgs[i*msg_stride .. i*msg_stride + msg_len).
 * pk_inf (optional, n bytes): non-zero marks pk i as the identity. */
int ssa_verify_many(ssa_ctx *ctx, const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf,
                    const uint8_t *msgs, const uint64_t *msg_off, size_t msg_stride,
                    size_t msg_len, size_t n, uint32_t flags, uint8_t *status_out,
                    uint64_t *n_fail_out);

/* verify_batch: one verdict for the batch = AND of the per-signature verdicts (equal to the
 * reference's MSM verdict on honest and on corrupted-but-well-formed inputs; divergence
 * classes are listed in DESIGN.md).  Returns SSA_OK, SSA_INVALID_SIGNATURE (or
 * SSA_INVALID_PUBLIC_KEY with SSA_FLAG_CHECK_TORSION, SSA_MALFORMED) -- the smallest
 * non-zero status present.  n == 0 returns SSA_OK like the reference.  SSA_FLAG_SIG_FLAG_BYTE is always on
 * here (the reference's verify_batch honours the flag byte); pk_inf as in ssa_verify_many: an identity key is a
 * valid PublicKey (src/public.rs:95-101) and contributes nothing to the equation (src/batch.rs:106). */
int ssa_verify_batch(ssa_ctx *ctx, const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf,
                     const uint8_t *msgs, const uint64_t *msg_off, size_t msg_stride, size_t msg_len,
                     size_t n, uint32_t flags);

/* verify_batch exactly as src/batch.rs:31-130: sum s_i R_i - sum (s_i h_i) P_i ?= [sum s_i e_i] G with
 * R_i decompressed from sig.x (flag byte honoured), a 2n-point bucket MSM on the GPU and an x-only
 * comparison.  coeffs: n x 32-byte scalars standing in for Scalar::random(rng) (reduced mod q), or
 * NULL for 128-bit coefficients from a ChaCha20 stream (RFC 8439) generated on the device and keyed per
 * call with getrandom(2).  Returns SSA_OK, SSA_INVALID_SIGNATURE, or
 * SSA_MALFORMED where the reference panics (undecodable sig.x, src/batch.rs:67,104).  No torsion
 * check, like the reference.  Any n <= SSA_MAX_BATCH: above 2^23 signatures (SSA_MSM_SLICE) the batch runs slice after slice,
 * each reduced to its record like a shard, the records added up (src/batch.rs:98-129) -- bounded device memory. */
int ssa_verify_batch_msm(ssa_ctx *ctx, const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf,
                         const uint8_t *msgs, const uint64_t *msg_off, size_t msg_stride, size_t msg_len,
                         size_t n, const uint8_t *coeffs);

/* hash_message for n (R.x, pk, message) triples -> n x 32-byte digests */
int ssa_hash_message_many(ssa_ctx *ctx, const uint8_t *sigs, const uint8_t *pks,
                          const uint8_t *msgs, const uint64_t *msg_off, size_t msg_stride,
                          size_t msg_len, size_t n, uint8_t *digests_out);

/* Rescue-Prime hash_field over n rows of `felts_per_row` canonical u64 -> n x 4 felts */
int ssa_rescue_hash_many(ssa_ctx *ctx, const uint64_t *felts, uint32_t felts_per_row, size_t n,
                         uint64_t *digests_out);

/* pk_i = [sk_i]G (affine, 96 B) and sig_i = sign(sk_i, nonce_i, msg_i).  Secret keys and nonces are 32-byte LE
 * CANONICAL scalars in [1, q): 0 and values >= q return SSA_ERR_ARG (PrivateKey::new / Scalar::random never
 * produce them, src/private.rs:49-57).  Drawing them uniformly is the caller's job -- 64 random bytes reduced
 * mod q, as the C++ and Python mirrors do; 32 random bytes reduced mod q are biased (2^256 / q ~ 2.08) and leak the
 * key through the nonces.  The kernel is VARIABLE-TIME in the secrets (comb windows equal to zero are skipped and the
 * table is indexed by secret windows), like the reference's *_vartime verification calls but unlike its signing:
 * use it where timing side channels are out of scope (test-input generation, trusted hosts).
 * The _device form cannot report errors per lane: non-canonical inputs are reduced mod q there. */
int ssa_keygen_sign_many(ssa_ctx *ctx, const uint8_t *sks, const uint8_t *nonces,
                         const uint8_t *msgs, const uint64_t *msg_off, size_t msg_stride,
                         size_t msg_len, size_t n, uint8_t *pks_out, uint8_t *sigs_out);

/* The signing side with options (KeyPair::sign / sign_and_bind_pkey, src/signature.rs:114-156; PrivateKey::sign,
 * :65-110).  flags:
 *   SSA_FLAG_SIGN_CT     constant-time in the secrets, like the reference's `&BASEPOINT_TABLE * r` and Scalar::from_bits
 *                        (src/signature.rs:67,116,123): fixed 4-bit windows over a 98 KB table that is read in full for
 *                        every window (the entry is selected, never indexed), no skipped window, generic additions from a
 *                        public offset point with no exceptional-case branch, masked scalar arithmetic, compiled (branch-
 *                        free) field blocks.  Same bytes out as the throughput signer; about 5x its time.  A lane whose
 *                        addition meets an exceptional input (probability ~2^-250; surely only for a scalar that reduces
 *                        to 0, which the host form refuses) is recomputed by the exact variable-time code.
 *   SSA_FLAG_SIGN_KEYED  sigs_out receives n x 130-byte KeyedSignature records pk(49, compressed) || sig(81)
 *                        (KeyedSignature::to_bytes, src/signature.rs:237-245) instead of n x 81 bytes; pks_out may be
 *                        NULL then.
 * The host form checks the scalars (canonical, non-zero: SSA_ERR_ARG otherwise) in time independent of their values
 * and wipes its device copies of them before it returns.  ssa_keygen_sign_many[_device] = flags 0. */
#define SSA_FLAG_SIGN_CT 16u
#define SSA_FLAG_SIGN_KEYED 32u
int ssa_keygen_sign_many_ex(ssa_ctx *ctx, const uint8_t *sks, const uint8_t *nonces, const uint8_t *msgs,
                            const uint64_t *msg_off, size_t msg_stride, size_t msg_len, size_t n, uint32_t flags,
                            uint8_t *pks_out, uint8_t *sigs_out);

/* PublicKey::from(&PrivateKey) (src/public.rs:26-32; KeyPair::from_bytes / from_seed re-derive the key the same way,
 * src/keypair.rs:73-103): pks_out[i] = [sks[i]]G as 96-byte affine points -- ONE constant-time base multiplication per
 * key (the signer's table and window code) and nothing else derived from the secret: no nonce, no response scalar.
 * The host form checks the scalars (canonical, non-zero: SSA_ERR_ARG otherwise) in time independent of their values and
 * wipes its device copy before it returns; the _device form computes [sk mod q]G for any 32 bytes, through the
 * variable-time fallback when sk reduces to 0. */
int ssa_pubkey_many(ssa_ctx *ctx, const uint8_t *sks, size_t n, uint8_t *pks_out);

/* PublicKey::to_bytes (AffinePoint::to_compressed, src/public.rs:49-51): n x 96-byte affine points -> n x 49 bytes
 * x || flag byte (bit 7: the identity, [0; 48] || 0x80, src/public.rs:95-101 -- marked by pk_inf[i] != 0, optional; bit 6:
 * the sort flag of y).  status_out (optional): 0, or SSA_MALFORMED for a limb that is not canonical. */
int ssa_compress_many(ssa_ctx *ctx, const uint8_t *pks, const uint8_t *pk_inf, size_t n, uint8_t *out,
                      uint8_t *status_out);

#define SSA_KEYED_SIGNATURE_LENGTH 130  /* src/constants.rs:30 */
/* n x KeyedSignature::verify on the 130-byte wire form pk(49, compressed) || sig(81)
 * (src/signature.rs:232-271): the key is decompressed on the GPU, then verified as ssa_verify_many.
 * A record whose key or scalar does not decode (KeyedSignature::from_bytes is_none) gets SSA_MALFORMED. */
int ssa_verify_keyed_many(ssa_ctx *ctx, const uint8_t *keyed, const uint8_t *msgs, const uint64_t *msg_off,
                          size_t msg_stride, size_t msg_len, size_t n, uint32_t flags, uint8_t *status_out,
                          uint64_t *n_fail_out);

/* PublicKey::from_bytes (src/public.rs:54-56): n x 49-byte compressed points (48 bytes of x, flag
 * byte: bit 7 = infinity, bit 6 = sort flag, other bits clear) -> n x 96-byte affine points.
 * status_out[i] = 0 ok, 1 "decompression failed" (CtOption is_none); pk_inf_out[i] (optional) = 1
 * for the identity encoding [0;48] || 0x80 (src/public.rs:95-101). */
int ssa_decompress_many(ssa_ctx *ctx, const uint8_t *compressed, size_t n, uint8_t *pks_out,
                        uint8_t *pk_inf_out, uint8_t *status_out);

/* ---- device-buffer entry points: same semantics, every pointer is a device pointer ----
 * (work is enqueued on the context's stream; outputs are valid after ssa_ctx_sync) */
int ssa_verify_many_device(ssa_ctx *ctx, const uint8_t *d_sigs, const uint8_t *d_pks,
                           const uint8_t *d_pk_inf, const uint8_t *d_msgs,
                           const uint64_t *d_msg_off, size_t msg_stride, size_t msg_len, size_t n,
                           uint32_t flags, uint8_t *d_status_out, uint64_t *d_n_fail_out);
int ssa_hash_message_many_device(ssa_ctx *ctx, const uint8_t *d_sigs, const uint8_t *d_pks,
                                 const uint8_t *d_msgs, const uint64_t *d_msg_off,
                                 size_t msg_stride, size_t msg_len, size_t n,
                                 uint8_t *d_digests_out);
int ssa_rescue_hash_many_device(ssa_ctx *ctx, const uint64_t *d_felts, uint32_t felts_per_row,
                                size_t n, uint64_t *d_digests_out);
int ssa_keygen_sign_many_device(ssa_ctx *ctx, const uint8_t *d_sks, const uint8_t *d_nonces,
                                const uint8_t *d_msgs, const uint64_t *d_msg_off,
                                size_t msg_stride, size_t msg_len, size_t n, uint8_t *d_pks_out,
                                uint8_t *d_sigs_out);
int ssa_decompress_many_device(ssa_ctx *ctx, const uint8_t *d_compressed, size_t n, uint8_t *d_pks_out,
                               uint8_t *d_pk_inf_out, uint8_t *d_status_out);
/* (cannot report errors per lane: scalars are reduced mod q; d_pks_out may be NULL with SSA_FLAG_SIGN_KEYED.
 *  PRECONDITION of SSA_FLAG_SIGN_CT on device buffers: every sk and nonce is canonical and non-zero -- what the host form
 *  checks; a scalar that reduces to 0 is still signed correctly, but by the variable-time fallback) */
int ssa_keygen_sign_many_ex_device(ssa_ctx *ctx, const uint8_t *d_sks, const uint8_t *d_nonces, const uint8_t *d_msgs,
                                   const uint64_t *d_msg_off, size_t msg_stride, size_t msg_len, size_t n,
                                   uint32_t flags, uint8_t *d_pks_out, uint8_t *d_sigs_out);
int ssa_pubkey_many_device(ssa_ctx *ctx, const uint8_t *d_sks, size_t n, uint8_t *d_pks_out);
int ssa_compress_many_device(ssa_ctx *ctx, const uint8_t *d_pks, const uint8_t *d_pk_inf, size_t n, uint8_t *d_out,
                             uint8_t *d_status_out);
/* coeff_bytes in 1..32: little-endian coefficient width (d_coeffs == NULL: the library draws 128-bit
 * coefficients as above); *d_verdict_out receives the status.  A 32-byte coefficient is taken mod q (Scalar::random,
 * src/batch.rs:75-78).  A narrower one is recoded into signed digits over its own windows only, so a value that fills
 * them to the top (above 0x7fff...7fff with 16-bit windows, n >= 4096; above 0x7f7f...7f with 8-bit ones) stands for
 * raw - 2^(8 coeff_bytes): the SAME value multiplies R_i, h_i and e_i, so the equation is the reference's with another,
 * equally random, coefficient (tests/test_gpu_round4.py pins the rule against the oracle). */
int ssa_verify_batch_msm_device(ssa_ctx *ctx, const uint8_t *d_sigs, const uint8_t *d_pks,
                                const uint8_t *d_pk_inf, const uint8_t *d_msgs, const uint64_t *d_msg_off, size_t msg_stride,
                                size_t msg_len, size_t n, const uint8_t *d_coeffs, uint32_t coeff_bytes,
                                uint32_t *d_verdict_out);
int ssa_ctx_sync(ssa_ctx *ctx);

/* ---- keyed context: many signatures by few signers (validator sets) ---------------------------------
 * A key set runs once per KEY what Signature::verify runs per signature on the key: canonical-limb and on-curve
 * checks, the subgroup check [q]P == O (src/signature.rs:182-184) and the table of multiples the ladder needs.
 * ssa_verify_many_indexed then verifies signature i against key key_idx[i] starting at the ladder.  Same
 * statuses as ssa_verify_many (a key that failed its subgroup check gives SSA_INVALID_PUBLIC_KEY under
 * SSA_FLAG_CHECK_TORSION; an index >= m gives SSA_MALFORMED).  A key set belongs to the context it was created
 * on.  Destroy it before the context; one that outlives its context is orphaned by ssa_ctx_destroy (its tables are
 * freed there, every call on it but ssa_keyset_destroy returns SSA_ERR_ARG), never a dangling pointer. */
typedef struct ssa_keyset ssa_keyset;
/* table kind: SSA_KEYSET_LADDER keeps sixteen multiples per key (4 KB; verification runs the 250-doubling ladder),
 * SSA_KEYSET_COMB a comb of [d * 2^(16w)]P, w < 16, d < 65536 per key (100 MB; [h]P becomes 16 mixed additions, no
 * doublings -- about 8x less curve work per signature), SSA_KEYSET_AUTO the comb while all tables fit 16 GB (160 keys) */
#define SSA_KEYSET_AUTO 0u
#define SSA_KEYSET_COMB 1u
#define SSA_KEYSET_LADDER 2u
int ssa_keyset_create(ssa_ctx *ctx, const uint8_t *pks, const uint8_t *pk_inf, size_t m, uint32_t flags,
                      ssa_keyset **out);
int ssa_keyset_create_device(ssa_ctx *ctx, const uint8_t *d_pks, const uint8_t *d_pk_inf, size_t m, uint32_t flags,
                             ssa_keyset **out);
void ssa_keyset_destroy(ssa_keyset *ks);
/* per-key status (m bytes): 0 usable, 1 not in the prime subgroup, 3 malformed */
int ssa_keyset_status(ssa_keyset *ks, uint8_t *status_out);
int ssa_verify_many_indexed(ssa_ctx *ctx, ssa_keyset *ks, const uint32_t *key_idx, const uint8_t *sigs,
                            const uint8_t *msgs, const uint64_t *msg_off, size_t msg_stride, size_t msg_len,
                            size_t n, uint32_t flags, uint8_t *status_out, uint64_t *n_fail_out);
int ssa_verify_many_indexed_device(ssa_ctx *ctx, ssa_keyset *ks, const uint32_t *d_key_idx, const uint8_t *d_sigs,
                                   const uint8_t *d_msgs, const uint64_t *d_msg_off, size_t msg_stride,
                                   size_t msg_len, size_t n, uint32_t flags, uint8_t *d_status_out,
                                   uint64_t *d_n_fail_out);

/* ---- several GPUs of one node from a single process --------------------------------------------
 * The batch shards by signature (contiguous ranges, sizes differ by at most one) over the listed
 * devices -- one context and one host thread per device, no collective: every verification reads
 * only its own record (src/signature.rs:181-205).  Same semantics as ssa_verify_many. */
typedef struct ssa_multi ssa_multi;
int ssa_multi_create(ssa_multi **out, const int *devices, int n_devices, const void *params, size_t params_len);
void ssa_multi_destroy(ssa_multi *m);
int ssa_multi_verify_many(ssa_multi *m, const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf,
                          const uint8_t *msgs, const uint64_t *msg_off, size_t msg_stride, size_t msg_len,
                          size_t n, uint32_t flags, uint8_t *status_out, uint64_t *n_fail_out);

/* verify_batch exactly as src/batch.rs (ssa_verify_batch_msm) with the batch sharded over the devices: each device
 * reduces its shard to one point and one scalar, device 0 adds the shards up (one point addition per shard), computes
 * [sum s_i e_i]G and compares x coordinates.  coeffs: n x 32 bytes or NULL (every device draws its own). */
int ssa_multi_verify_batch_msm(ssa_multi *m, const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf,
                               const uint8_t *msgs, const uint64_t *msg_off, size_t msg_stride, size_t msg_len,
                               size_t n, const uint8_t *coeffs);

/* ---- MSM-form verify_batch across PROCESSES (one process per GPU, torch.distributed / RCCL) ----------------
 * The same split as ssa_multi_verify_batch_msm, with the exchange left to the caller ("in the MSM
 * form, one point addition per shard + one compare", src/batch.rs:98-129): every rank reduces ITS shard to one
 * record of SSA_MSM_PARTIAL_WORDS u64 --
 *     words 0..17  the shard's left-hand point  sum s_i R_i - sum (s_i h_i) P_i  as X, Y, Z in canonical limbs; the
 *                  library writes the canonical form -- affine (x, y, 1), or (0, 0, 0) for the identity -- so equal
 *                  shards give equal bytes; ssa_msm_combine accepts any Jacobian representative
 *     words 18..21 sum s_i e_i mod q            word 22  1: the shard holds an input the reference panics on, else 0
 *     word 23      SSA_MSM_RECORD_MAGIC -- every record the library produces carries it, the empty shard's too
 * -- the ranks all-gather the records (24 words per rank: the only traffic), and ssa_msm_combine adds the k points
 * up (one Jacobian addition per shard), computes [sum]G from the comb table and compares x coordinates exactly as
 * the single-context call does.  An empty shard (n == 0) gives the identity (all limbs 0), 0 and the magic word.
 * Coefficients as in ssa_verify_batch_msm_device (NULL: drawn per call on the device; every rank draws its own).
 *
 * ORDERING (the _device forms are asynchronous): ssa_verify_batch_msm_partial_device only enqueues on the context's
 * stream; the record is valid for a collective on another stream after ssa_ctx_stream_release(ctx, that stream) (or
 * ssa_ctx_sync, or a shared stream via ssa_ctx_set_stream), and the gathered records are visible to
 * ssa_msm_combine_device after ssa_ctx_stream_acquire(ctx, the collective's stream).  The combination FAILS CLOSED:
 * a record that was never written (all zero), comes from another format version, has a non-canonical limb or scalar,
 * or a point off the curve makes the verdict SSA_MALFORMED -- never SSA_OK. */
#define SSA_MSM_PARTIAL_WORDS 24
#define SSA_MSM_RECORD_MAGIC 0x5353415245430004ull   /* "SSAREC", format 4 */
int ssa_verify_batch_msm_partial_device(ssa_ctx *ctx, const uint8_t *d_sigs, const uint8_t *d_pks,
                                        const uint8_t *d_pk_inf, const uint8_t *d_msgs, const uint64_t *d_msg_off,
                                        size_t msg_stride, size_t msg_len, size_t n, const uint8_t *d_coeffs,
                                        uint32_t coeff_bytes, uint64_t *d_partial_out);
/* host-buffer form (coeffs: n x 32 bytes or NULL); out24 is host memory */
int ssa_verify_batch_msm_partial(ssa_ctx *ctx, const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_inf,
                                 const uint8_t *msgs, const uint64_t *msg_off, size_t msg_stride, size_t msg_len,
                                 size_t n, const uint8_t *coeffs, uint64_t out24[SSA_MSM_PARTIAL_WORDS]);
/* k records (k x 24 words, device memory) -> *d_verdict_out = SSA_OK / SSA_INVALID_SIGNATURE / SSA_MALFORMED,
 * enqueued on the context's stream; 1 <= k <= 4096 */
int ssa_msm_combine_device(ssa_ctx *ctx, const uint64_t *d_parts24, size_t k, uint32_t *d_verdict_out);
/* the same from host memory; returns the status */
int ssa_msm_combine(ssa_ctx *ctx, const uint64_t *parts24, size_t k);

/* ---- ABI version -------------------------------------------------------------------------------------------
 * Bumped whenever an exported signature changes (round 2 inserted pk_inf into the batch entry points under the same
 * symbol names: a shim built against the older header would still link and pass msgs as pk_inf).  A binding checks
 * ssa_abi_version() == SSA_ABI_VERSION at load; the Python and C++ mirrors do. */
#define SSA_ABI_VERSION 5
int ssa_abi_version(void);

/* ---- arithmetic probes (unit parity with the oracle; not part of the reference API) --- */
/* op: 0 = Fp6 mul, 1 = Fp6 sqr, 2 = Fp6 inv, 3 = point add (affine 12+12 -> 12 felts + inf),
 *     4 = scalar mul [k]P (k in a[0..4], P in b), 5 = Fp mul (a[0]*b[0]), 6 = Fp inv,
 *     7 = the cooperative (one wave per point) operations: a = (x, y, inf, mode), b = (x, y, inf);
 *         mode 0 mixed addition, 1 general addition of two scaled Jacobian points, 2 doubling of a
 *     8..14 = products fused with linear terms, a = (u, v) 12 felts, b = (x, y) 12 felts -> 6 felts:
 *         8 u^2 - x - y, 9 u^2 + 3x, 10 u^2 - 4x, 11 u*v - 8x, 12 u*v - x, 13 u^2 - x - 2y, 14 u*v + x*y
 *     15..17 = the generated point operations of the ladder on RAW loose limbs (any 64-bit values, nothing is
 *         canonicalised on the way in or out): a = X, Y, Z (18 words), a[18] = act, a[19] = n; b = x2, y2 (12 words)
 *         -> X, Y, Z (18 words) + out[18] = flag.  15 one ladder window (n doublings, then the mixed addition where
 *         act != 0; flag 0 = the addition met a possible exceptional input and was left to the caller),
 *         16 mixed addition with its exact fallback, 17 n doublings
 *     18 = the Rescue S-boxes on raw loose values: a = (x, y) -> x^7, y^7, x^(1/7), y^(1/7) (canonical) and the flags of
 *         the two generated blocks (non-zero: the block reported its rare reduction borrow and the lane was recomputed) */
int ssa_debug_arith(ssa_ctx *ctx, int op, const uint64_t *a, const uint64_t *b, size_t n,
                    size_t a_stride, size_t b_stride, uint64_t *out, size_t out_stride);
/* error-path tests: the next pipelined host-buffer upload on this context fails with SSA_ERR_HIP after chunk `chunk`
 * has been enqueued (one shot; chunk < 0 disarms).  Replaces round 3's SSA_FAULT_AFTER_CHUNK environment variable: the
 * production path reads no environment per call. */
int ssa_debug_fault_after_chunk(ssa_ctx *ctx, int chunk);
/* host logic of ssa_k_verify's end game, no context and no device needed: the launch plan for n lanes when `waves`
 * waves are resident (pieces / gens / uniform / min_main: what SSA_TAIL_PIECES, _GENS, _UNIFORM, _MIN_MAIN set).
 * out: number of pieces (0: no end game), 64-lane tail groups, ordinary workgroups, workgroups of the grid, the eight
 * piece descriptors (pass | first-of-pass << 1 | last-of-pass << 2 | first window << 8 | end window << 16) and the two
 * whole-pass descriptors ([q]P of the subgroup check, [h]P). */
int ssa_debug_tail_plan(unsigned waves, unsigned pieces, unsigned gens, int uniform, unsigned min_main, size_t n,
                        uint32_t flags, uint32_t out[14]);
/* n_blocks 64-byte blocks of the ChaCha20 keystream the MSM coefficients come from (RFC 8439 known answers) */
int ssa_debug_chacha20(ssa_ctx *ctx, const uint8_t key[32], const uint8_t nonce[12], uint32_t counter0,
                       size_t n_blocks, uint8_t *out);
/* register-resident Fp-mul throughput probe: returns Fp multiplications per second (variants 0..2: fp_mul chains
 * with 1/4/8 independent streams, 3: the lazy Fp6 product + square, 4 / 5: squaring chains with the four- / three-
 * multiply square); variants 10..12: dependent
 * cooperative doublings / mixed additions / general additions per second (one wave) */
int ssa_bench_fpmul(ssa_ctx *ctx, int variant, double *fpmul_per_s);

#ifdef __cplusplus
}
#endif
#endif /* SCHNORR_SIG_AMD_H */
