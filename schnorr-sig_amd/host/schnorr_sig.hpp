// schnorr_sig.hpp -- C++17 host-side mirror of the reference's Rust API over the C ABI
// (include/schnorr_sig_amd.h).  Rust is not available in this image; this header keeps the
// reference's names, argument meaning and error behaviour so that a Rust `-sys` shim (see
// INTEGRATION.md) and these classes are interchangeable callers of the same entry points.
//
//   schnorr_sig::Signature::verify          <- src/signature.rs:181-205
//   schnorr_sig::KeyPair::{create, sign, sign_and_bind_pkey, verify_signature}
//                                           <- src/keypair.rs:57-65, src/signature.rs:114-165 (signing is CONSTANT-TIME,
//                                              SSA_FLAG_SIGN_CT, like the reference's `&BASEPOINT_TABLE * r`)
//   schnorr_sig::PublicKey::{to_bytes, from_bytes}          <- src/public.rs:49-56
//   schnorr_sig::KeyedSignature::{to_bytes, from_bytes, verify}  <- src/signature.rs:232-271
//   schnorr_sig::PublicKey::verify_signature <- src/signature.rs:170-176
//   schnorr_sig::verify_batch               <- src/batch.rs:31-50
//   schnorr_sig::SignatureError             <- src/error.rs:13-31
//
// All compute happens on the GPU behind ssa_*; nothing here does field or curve arithmetic.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <functional>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/schnorr_sig_amd.h"

namespace schnorr_sig {

constexpr size_t SCALAR_LENGTH = 32, BASEFIELD_LENGTH = 48, PUBLIC_KEY_LENGTH = 49, SIGNATURE_LENGTH = 81,
                 KEYED_SIGNATURE_LENGTH = 130, AFFINE_PUBLIC_KEY_LENGTH = 96, PRIVATE_KEY_LENGTH = SCALAR_LENGTH,
                 KEY_PAIR_LENGTH = PRIVATE_KEY_LENGTH;  // src/constants.rs:12-30

enum class SignatureError { InvalidPublicKey = 1, InvalidSignature = 2 };  // src/error.rs:13-18
inline const char *to_string(SignatureError e) {                           // src/error.rs:20-31
    return e == SignatureError::InvalidPublicKey ? "The public key is not an element of the prime subgroup."
                                                 : "The signature is invalid or was incorrectly computed.";
}
// Result<(), SignatureError>: empty optional == Ok(())
using Result = std::optional<SignatureError>;

// Inputs the reference panics on (src/signature.rs:186, src/batch.rs:37-44,67,104)
struct Panic : std::runtime_error {
    using std::runtime_error::runtime_error;
};

using Rng = std::function<void(uint8_t *, size_t)>;  // fills a buffer with random bytes

class Context {
  public:
    // gtab_bits / hbm_budget_bytes: the comb for G (the reference's const BASEPOINT_TABLE) as a speed-for-memory choice of
    // the context -- 0 / 0 = the widest table that fits a tenth of the free device memory (ssa_ctx_create_ex)
    explicit Context(int device = 0, const void *params = nullptr, size_t params_len = 0, uint32_t gtab_bits = 0,
                     uint64_t hbm_budget_bytes = 0) {
        // a library from another revision of the header links just as well and reads its arguments shifted
        if (ssa_abi_version() != SSA_ABI_VERSION)
            throw std::runtime_error("schnorr_sig_amd: library ABI version " + std::to_string(ssa_abi_version()) +
                                     ", header " + std::to_string(SSA_ABI_VERSION));
        int rc = ssa_ctx_create_ex(&ctx_, device, params, params_len, gtab_bits, hbm_budget_bytes);
        if (rc != 0) throw std::runtime_error(std::string("ssa_ctx_create: ") + ssa_strerror(rc));
    }
    ~Context() { ssa_ctx_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    ssa_ctx *get() const { return ctx_; }

  private:
    ssa_ctx *ctx_ = nullptr;
};

inline Result status_to_result(int st) {
    if (st == SSA_OK) return std::nullopt;
    if (st == SSA_INVALID_PUBLIC_KEY) return SignatureError::InvalidPublicKey;
    if (st == SSA_INVALID_SIGNATURE) return SignatureError::InvalidSignature;
    if (st == SSA_MALFORMED) throw Panic("undecodable field element or scalar (the reference panics)");
    throw std::runtime_error(std::string("schnorr_sig_amd: ") + ssa_strerror(st));
}

struct Signature;
struct KeyedSignature;
struct PublicKey;

// a little-endian integer of 64 bytes reduced mod q (Scalar::from_bytes_wide): binary long division, host glue
inline void reduce_wide_mod_q(const uint8_t wide[64], uint64_t r[4]) {
    static const uint64_t Q[4] = {0xd443623eaed4accfULL, 0x327aa72330157722ULL, 0x563fbf0f990a37b5ULL, 0x7af2599b3b3f22d0ULL};
    r[0] = r[1] = r[2] = r[3] = 0;
    for (int bit = 511; bit >= 0; bit--) {
        const uint64_t top = r[3] >> 63;
        r[3] = (r[3] << 1) | (r[2] >> 63);
        r[2] = (r[2] << 1) | (r[1] >> 63);
        r[1] = (r[1] << 1) | (r[0] >> 63);
        r[0] = (r[0] << 1) | ((wide[bit >> 3] >> (bit & 7)) & 1u);
        bool ge = top != 0;
        if (!ge) {
            ge = true;
            for (int i = 3; i >= 0; i--) {
                if (r[i] != Q[i]) {
                    ge = r[i] > Q[i];
                    break;
                }
            }
        }
        if (ge) {
            unsigned __int128 borrow = 0;
            for (int i = 0; i < 4; i++) {
                const unsigned __int128 d = (unsigned __int128)r[i] - Q[i] - borrow;
                r[i] = (uint64_t)d;
                borrow = (d >> 64) & 1;
            }
        }
    }
}

struct PrivateKey {  // src/private.rs:25
    std::array<uint8_t, SCALAR_LENGTH> bytes{};
    bool operator==(const PrivateKey &o) const { return bytes == o.bytes; }
    std::array<uint8_t, PRIVATE_KEY_LENGTH> to_bytes() const { return bytes; }  // src/private.rs:69-71
    // PrivateKey::from_bytes, src/private.rs:74-76: nullopt for a non-canonical or zero scalar
    static std::optional<PrivateKey> from_bytes(const std::array<uint8_t, PRIVATE_KEY_LENGTH> &b) {
        static const uint64_t Q[4] = {0xd443623eaed4accfULL, 0x327aa72330157722ULL, 0x563fbf0f990a37b5ULL, 0x7af2599b3b3f22d0ULL};
        uint64_t w[4] = {0, 0, 0, 0};
        for (int i = 0; i < 32; i++) w[i / 8] |= (uint64_t)b[i] << (8 * (i % 8));
        if ((w[0] | w[1] | w[2] | w[3]) == 0) return std::nullopt;
        for (int i = 3; i >= 0; i--) {
            if (w[i] != Q[i]) {
                if (w[i] > Q[i]) return std::nullopt;
                PrivateKey k;
                k.bytes = b;
                return k;
            }
        }
        return std::nullopt;   // == q
    }
    // PrivateKey::from_seed, src/private.rs:79-82
    static std::optional<PrivateKey> from_seed(const std::array<uint8_t, 64> &seed) {
        uint64_t r[4];
        reduce_wide_mod_q(seed.data(), r);
        if ((r[0] | r[1] | r[2] | r[3]) == 0) return std::nullopt;
        PrivateKey k;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 8; j++) k.bytes[8 * i + j] = (uint8_t)(r[i] >> (8 * j));
        return k;
    }
    // PrivateKey::sign / sign_and_bind_pkey, src/signature.rs:62-110 ("it is faster to sign with a KeyPair": the public key
    // is recomputed first, PublicKey::from(self)) -- defined after KeyPair
    Signature sign(Context &cx, const uint8_t *msg, size_t len, Rng rng) const;
    KeyedSignature sign_and_bind_pkey(Context &cx, const uint8_t *msg, size_t len, Rng rng) const;
};

struct PublicKey {  // src/public.rs:24 -- the in-memory AffinePoint (x, y), canonical LE limbs
    std::array<uint8_t, AFFINE_PUBLIC_KEY_LENGTH> affine{};
    bool is_identity = false;  // AffinePoint::identity() is a valid PublicKey (src/public.rs:95-101)
    bool operator==(const PublicKey &o) const { return affine == o.affine && is_identity == o.is_identity; }
    static PublicKey from_private(Context &cx, const PrivateKey &sk);   // impl From<&PrivateKey>, src/public.rs:26-32
    Result verify_signature(Context &cx, const Signature &sig, const uint8_t *msg, size_t len) const;
    // PublicKey::to_bytes, src/public.rs:49-51: the 49-byte compressed wire form
    std::array<uint8_t, PUBLIC_KEY_LENGTH> to_bytes(Context &cx) const {
        std::array<uint8_t, PUBLIC_KEY_LENGTH> out{};
        const uint8_t inf = is_identity ? 1 : 0;
        uint8_t st = SSA_MALFORMED;
        const int rc = ssa_compress_many(cx.get(), affine.data(), &inf, 1, out.data(), &st);
        if (rc != 0) throw std::runtime_error(std::string("ssa_compress_many: ") + ssa_strerror(rc));
        if (st != SSA_OK) throw Panic("PublicKey holds a non-canonical limb");
        return out;
    }
    // PublicKey::from_bytes, src/public.rs:54-56: nullopt when decompression fails (CtOption is_none)
    static std::optional<PublicKey> from_bytes(Context &cx, const std::array<uint8_t, PUBLIC_KEY_LENGTH> &b) {
        PublicKey pk;
        uint8_t inf = 0, st = 1;
        const int rc = ssa_decompress_many(cx.get(), b.data(), 1, pk.affine.data(), &inf, &st);
        if (rc != 0) throw std::runtime_error(std::string("ssa_decompress_many: ") + ssa_strerror(rc));
        if (st != 0) return std::nullopt;
        pk.is_identity = inf != 0;
        return pk;
    }
};

struct Signature {  // src/signature.rs:34-40, wire layout :208-214
    std::array<uint8_t, SIGNATURE_LENGTH> bytes{};
    // Signature::verify, src/signature.rs:181-205
    Result verify(Context &cx, const uint8_t *msg, size_t len, const PublicKey &pk) const {
        uint8_t st = SSA_MALFORMED;
        const uint8_t inf = pk.is_identity ? 1 : 0, dummy = 0;
        const int rc = ssa_verify_many(cx.get(), bytes.data(), pk.affine.data(), &inf, len ? msg : &dummy, nullptr, len,
                                       len, 1, SSA_FLAG_CHECK_TORSION, &st, nullptr);
        return status_to_result(rc != 0 ? rc : (int)st);
    }
    std::array<uint8_t, SIGNATURE_LENGTH> to_bytes() const { return bytes; }
};

inline Result PublicKey::verify_signature(Context &cx, const Signature &sig, const uint8_t *msg, size_t len) const {
    return sig.verify(cx, msg, len, *this);
}

struct KeyedSignature {  // src/signature.rs:55-60; wire form pk(49) || sig(81), :236-271
    PublicKey public_key;
    Signature signature;
    Result verify(Context &cx, const uint8_t *msg, size_t len) const { return signature.verify(cx, msg, len, public_key); }
    std::array<uint8_t, KEYED_SIGNATURE_LENGTH> to_bytes(Context &cx) const {
        std::array<uint8_t, KEYED_SIGNATURE_LENGTH> out{};
        const auto pk = public_key.to_bytes(cx);
        std::memcpy(out.data(), pk.data(), PUBLIC_KEY_LENGTH);
        std::memcpy(out.data() + PUBLIC_KEY_LENGTH, signature.bytes.data(), SIGNATURE_LENGTH);
        return out;
    }
    // nullopt unless both halves decode (the scalar e must be canonical: Signature::from_bytes, src/signature.rs:217-227)
    static std::optional<KeyedSignature> from_bytes(Context &cx, const std::array<uint8_t, KEYED_SIGNATURE_LENGTH> &b) {
        std::array<uint8_t, PUBLIC_KEY_LENGTH> pkb;
        std::memcpy(pkb.data(), b.data(), PUBLIC_KEY_LENGTH);
        const auto pk = PublicKey::from_bytes(cx, pkb);
        static const uint8_t q_le[32] = {0xcf, 0xac, 0xd4, 0xae, 0x3e, 0x62, 0x43, 0xd4, 0x22, 0x77, 0x15,
                                         0x30, 0x23, 0xa7, 0x7a, 0x32, 0xb5, 0x37, 0x0a, 0x99, 0x0f, 0xbf,
                                         0x3f, 0x56, 0xd0, 0x22, 0x3f, 0x3b, 0x9b, 0x59, 0xf2, 0x7a};
        int cmp = 0;
        for (int k = 31; k >= 0 && cmp == 0; k--)
            if (b[PUBLIC_KEY_LENGTH + 49 + k] != q_le[k]) cmp = b[PUBLIC_KEY_LENGTH + 49 + k] < q_le[k] ? -1 : 1;
        if (!pk || cmp >= 0) return std::nullopt;
        KeyedSignature ks;
        ks.public_key = *pk;
        std::memcpy(ks.signature.bytes.data(), b.data() + PUBLIC_KEY_LENGTH, SIGNATURE_LENGTH);
        return ks;
    }
};

struct KeyPair {  // src/keypair.rs:48-53
    PrivateKey private_key;
    PublicKey public_key;

    // Scalar::random(rng): 64 random bytes reduced mod q (statistical distance from uniform < 2^-256), never 0.
    // (A 32-byte draw reduced mod q, or a masked 254-bit draw, is biased -- fatal for nonces: hidden-number problem.)
    static void random_scalar(Rng &rng, uint8_t out[32]) {
        for (;;) {
            uint8_t wide[64];
            rng(wide, sizeof wide);
            uint64_t r[4];
            reduce_wide_mod_q(wide, r);
            if ((r[0] | r[1] | r[2] | r[3]) == 0) continue;   // PrivateKey::new rejects 0 (src/private.rs:49-57)
            for (int i = 0; i < 4; i++)
                for (int k = 0; k < 8; k++) out[8 * i + k] = (uint8_t)(r[i] >> (8 * k));
            return;
        }
    }
    bool operator==(const KeyPair &o) const { return private_key == o.private_key && public_key == o.public_key; }
    // impl From<&PrivateKey> for KeyPair, src/keypair.rs:21-32: the public key is [sk]G -- one constant-time base
    // multiplication (ssa_pubkey_many) and nothing else derived from the secret
    static KeyPair from_private(Context &cx, const PrivateKey &sk) {
        KeyPair kp;
        kp.private_key = sk;
        int rc = ssa_pubkey_many(cx.get(), sk.bytes.data(), 1, kp.public_key.affine.data());
        if (rc != 0) throw std::runtime_error(std::string("ssa_pubkey_many: ") + ssa_strerror(rc));
        return kp;
    }
    // KeyPair::to_bytes / from_bytes / from_seed, src/keypair.rs:73-103: the private key only, the public key is rebuilt
    std::array<uint8_t, KEY_PAIR_LENGTH> to_bytes() const { return private_key.to_bytes(); }
    static std::optional<KeyPair> from_bytes(Context &cx, const std::array<uint8_t, KEY_PAIR_LENGTH> &b) {
        const auto sk = PrivateKey::from_bytes(b);
        if (!sk) return std::nullopt;
        return from_private(cx, *sk);
    }
    static std::optional<KeyPair> from_seed(Context &cx, const std::array<uint8_t, 64> &seed) {
        const auto sk = PrivateKey::from_seed(seed);
        if (!sk) return std::nullopt;
        return from_private(cx, *sk);
    }
    // KeyPair::new, src/keypair.rs:57-65
    static KeyPair create(Context &cx, Rng rng) {
        PrivateKey sk;
        random_scalar(rng, sk.bytes.data());
        return from_private(cx, sk);
    }
    // KeyPair::sign, src/signature.rs:114-129 (constant-time in the key and the nonce, like the reference)
    Signature sign(Context &cx, const uint8_t *msg, size_t len, Rng rng) const {
        uint8_t nonce[32], pk[AFFINE_PUBLIC_KEY_LENGTH];
        random_scalar(rng, nonce);
        Signature s;
        uint8_t dummy = 0;
        int rc = ssa_keygen_sign_many_ex(cx.get(), private_key.bytes.data(), nonce, len ? msg : &dummy, nullptr, len,
                                         len, 1, SSA_FLAG_SIGN_CT, pk, s.bytes.data());
        if (rc != 0) throw std::runtime_error(std::string("ssa_keygen_sign_many_ex: ") + ssa_strerror(rc));
        return s;
    }
    // KeyPair::sign_and_bind_pkey, src/signature.rs:132-156: the engine emits the 130-byte record itself
    // (SSA_FLAG_SIGN_KEYED); the public key inside it is this pair's
    KeyedSignature sign_and_bind_pkey(Context &cx, const uint8_t *msg, size_t len, Rng rng) const {
        uint8_t nonce[32], rec[KEYED_SIGNATURE_LENGTH], dummy = 0;
        random_scalar(rng, nonce);
        int rc = ssa_keygen_sign_many_ex(cx.get(), private_key.bytes.data(), nonce, len ? msg : &dummy, nullptr, len,
                                         len, 1, SSA_FLAG_SIGN_CT | SSA_FLAG_SIGN_KEYED, nullptr, rec);
        if (rc != 0) throw std::runtime_error(std::string("ssa_keygen_sign_many_ex: ") + ssa_strerror(rc));
        KeyedSignature ks;
        ks.public_key = public_key;
        std::memcpy(ks.signature.bytes.data(), rec + PUBLIC_KEY_LENGTH, SIGNATURE_LENGTH);
        return ks;
    }
    Result verify_signature(Context &cx, const Signature &sig, const uint8_t *msg, size_t len) const {
        return sig.verify(cx, msg, len, public_key);  // src/signature.rs:159-165
    }
};

inline PublicKey PublicKey::from_private(Context &cx, const PrivateKey &sk) { return KeyPair::from_private(cx, sk).public_key; }
inline Signature PrivateKey::sign(Context &cx, const uint8_t *msg, size_t len, Rng rng) const {
    return KeyPair::from_private(cx, *this).sign(cx, msg, len, rng);
}
inline KeyedSignature PrivateKey::sign_and_bind_pkey(Context &cx, const uint8_t *msg, size_t len, Rng rng) const {
    return KeyPair::from_private(cx, *this).sign_and_bind_pkey(cx, msg, len, rng);
}

// Many signatures by few signers (validator sets; the reference's own batch test reuses keys, src/batch.rs:152-175):
// the key checks of Signature::verify -- canonical limbs, on the curve, subgroup check (src/signature.rs:182-184) --
// and the tables the verification needs are computed once per key (ssa_keyset_create); verify() then checks
// signature i against key key_idx[i] with Signature::verify's semantics.
class KeySet {
  public:
    KeySet(Context &cx, const std::vector<PublicKey> &keys, uint32_t kind = SSA_KEYSET_AUTO) : cx_(cx), m_(keys.size()) {
        std::vector<uint8_t> pks(m_ * AFFINE_PUBLIC_KEY_LENGTH), inf(m_);
        for (size_t i = 0; i < m_; i++) {
            std::memcpy(&pks[i * AFFINE_PUBLIC_KEY_LENGTH], keys[i].affine.data(), AFFINE_PUBLIC_KEY_LENGTH);
            inf[i] = keys[i].is_identity ? 1 : 0;
        }
        int rc = ssa_keyset_create(cx.get(), pks.data(), inf.data(), m_, kind, &ks_);
        if (rc != 0) throw std::runtime_error(std::string("ssa_keyset_create: ") + ssa_strerror(rc));
    }
    ~KeySet() { ssa_keyset_destroy(ks_); }
    KeySet(const KeySet &) = delete;
    KeySet &operator=(const KeySet &) = delete;
    size_t size() const { return m_; }
    // one Result per signature (a Panic for inputs the reference would panic on)
    std::vector<Result> verify(const std::vector<Signature> &signatures, const std::vector<uint32_t> &key_idx,
                               const std::vector<std::pair<const uint8_t *, size_t>> &messages) const {
        const size_t n = signatures.size();
        if (key_idx.size() != n || messages.size() != n) throw Panic("one key index and one message per signature");
        std::vector<Result> out(n);
        if (n == 0) return out;
        std::vector<uint8_t> sigs(n * SIGNATURE_LENGTH), flat, status(n);
        std::vector<uint64_t> off(n + 1, 0);
        for (size_t i = 0; i < n; i++) {
            std::memcpy(&sigs[i * SIGNATURE_LENGTH], signatures[i].bytes.data(), SIGNATURE_LENGTH);
            flat.insert(flat.end(), messages[i].first, messages[i].first + messages[i].second);
            off[i + 1] = flat.size();
        }
        flat.push_back(0);
        int rc = ssa_verify_many_indexed(cx_.get(), ks_, key_idx.data(), sigs.data(), flat.data(), off.data(), 0, 0, n,
                                         SSA_FLAG_CHECK_TORSION, status.data(), nullptr);
        if (rc != 0) throw std::runtime_error(std::string("ssa_verify_many_indexed: ") + ssa_strerror(rc));
        for (size_t i = 0; i < n; i++) out[i] = status_to_result(status[i]);
        return out;
    }

  private:
    Context &cx_;
    size_t m_;
    ssa_keyset *ks_ = nullptr;
};

// verify_batch, src/batch.rs:31-50.
//   msm = false: AND of exact per-signature checks (`rng` unused; DESIGN.md lists the divergence classes)
//   msm = true : the reference's own algorithm on the GPU (random linear combination + 2n-point MSM), the
//                coefficients are Scalar::random(rng) per signature (src/batch.rs:75-78); rng == nullptr lets the
//                library draw them (ChaCha20 keyed with getrandom(2))
inline Result verify_batch(Context &cx, const std::vector<Signature> &signatures,
                           const std::vector<PublicKey> &public_keys,
                           const std::vector<std::pair<const uint8_t *, size_t>> &messages, Rng rng = nullptr,
                           bool msm = false) {
    if (signatures.size() != public_keys.size())
        throw Panic("We should have the same number of signatures than public keys");  // src/batch.rs:37-40
    if (messages.size() != public_keys.size())
        throw Panic("We should have the same number of messages than public keys");    // src/batch.rs:41-44
    const size_t n = signatures.size();
    if (n == 0) return std::nullopt;
    std::vector<uint8_t> sigs(n * SIGNATURE_LENGTH), pks(n * AFFINE_PUBLIC_KEY_LENGTH), inf(n), flat;
    std::vector<uint64_t> off(n + 1, 0);
    for (size_t i = 0; i < n; i++) {
        std::memcpy(&sigs[i * SIGNATURE_LENGTH], signatures[i].bytes.data(), SIGNATURE_LENGTH);
        std::memcpy(&pks[i * AFFINE_PUBLIC_KEY_LENGTH], public_keys[i].affine.data(), AFFINE_PUBLIC_KEY_LENGTH);
        inf[i] = public_keys[i].is_identity ? 1 : 0;
        flat.insert(flat.end(), messages[i].first, messages[i].first + messages[i].second);
        off[i + 1] = flat.size();
    }
    flat.push_back(0);
    if (msm) {
        std::vector<uint8_t> coeffs;
        if (rng) {
            coeffs.resize(n * SCALAR_LENGTH);
            for (size_t i = 0; i < n; i++) KeyPair::random_scalar(rng, &coeffs[i * SCALAR_LENGTH]);
        }
        return status_to_result(ssa_verify_batch_msm(cx.get(), sigs.data(), pks.data(), inf.data(), flat.data(),
                                                     off.data(), 0, 0, n, rng ? coeffs.data() : nullptr));
    }
    return status_to_result(
        ssa_verify_batch(cx.get(), sigs.data(), pks.data(), inf.data(), flat.data(), off.data(), 0, 0, n, 0));
}

}  // namespace schnorr_sig
