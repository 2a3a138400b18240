"""schnorr_sig_amd -- host-side mirror of toposware/schnorr-sig's verification API over the
MI355X-native HIP engine (C ABI: include/schnorr_sig_amd.h).

This package is plumbing only: it loads csrc/libschnorr_sig_amd.so with ctypes and forwards
to it.  There is no CPU compute path; if the HIP library is missing, import fails loudly.

Reference API mirrored (names and semantics, reference file:line):
  Signature.verify(message, pkey)            src/signature.rs:181-205
  PublicKey.verify_signature / KeyPair.verify_signature   src/signature.rs:159-176
  KeyPair.new / KeyPair.sign / sign_and_bind_pkey   src/keypair.rs:57-65, src/signature.rs:114-156
  PublicKey.to_bytes / from_bytes            src/public.rs:49-56
  KeyedSignature.to_bytes / from_bytes       src/signature.rs:236-271
  verify_batch(signatures, public_keys, messages, rng)    src/batch.rs:31-50
  SignatureError.{InvalidPublicKey, InvalidSignature}     src/error.rs:13-31
  *_LENGTH constants                         src/constants.rs:12-30

The directory is named `schnorr-sig_amd`; import it as `schnorr_sig_amd` (repo-root shim).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SSA_LIB: an alternative build of the same library (kernel experiments, tools/build_variants.sh); default: in-tree
LIB_PATH = os.environ.get("SSA_LIB") or os.path.join(_HERE, "csrc", "libschnorr_sig_amd.so")

# src/constants.rs:12-30
SCALAR_LENGTH = 32
PRIVATE_KEY_LENGTH = 32
BASEFIELD_LENGTH = 48
PUBLIC_KEY_LENGTH = 49          # compressed wire form (ssa_compress_many / ssa_decompress_many)
AFFINE_PUBLIC_KEY_LENGTH = 96   # in-memory AffinePoint (x, y): what the engine consumes
KEY_PAIR_LENGTH = 32
SIGNATURE_LENGTH = 81
KEYED_SIGNATURE_LENGTH = 130

OK, INVALID_PUBLIC_KEY, INVALID_SIGNATURE, MALFORMED = 0, 1, 2, 3
FLAG_CHECK_TORSION = 1
FLAG_FORCE_LANE = 2   # throughput kernels (one signature per lane) whatever the batch size
FLAG_FORCE_COOP = 4   # low-latency kernel (one wave per signature) whatever the batch size
FLAG_SIG_FLAG_BYTE = 8  # verify_batch's semantics for byte 48 of the signature (src/batch.rs:104)
FLAG_SIGN_CT = 16       # constant-time signing (the reference's `&BASEPOINT_TABLE * r`, src/signature.rs:67,116)
FLAG_SIGN_KEYED = 32    # 130-byte KeyedSignature records out (src/signature.rs:237-245)
KEYSET_KINDS = {"auto": 0, "comb": 1, "ladder": 2}
_MODE_FLAGS = {None: 0, "auto": 0, "lane": FLAG_FORCE_LANE, "coop": FLAG_FORCE_COOP}

Q = 0x7AF2599B3B3F22D0563FBF0F990A37B5327AA72330157722D443623EAED4ACCF


class SignatureError(Exception):
    """src/error.rs:13-31 (Display strings kept verbatim)."""
    InvalidPublicKey = "InvalidPublicKey"
    InvalidSignature = "InvalidSignature"
    _MSG = {
        "InvalidPublicKey": "The public key is not an element of the prime subgroup.",
        "InvalidSignature": "The signature is invalid or was incorrectly computed.",
    }

    def __init__(self, kind):
        super().__init__(self._MSG[kind])
        self.kind = kind

    def __repr__(self):
        return "Err(%s)" % self.kind


class MalformedInput(ValueError):
    """Inputs on which the reference panics (src/signature.rs:186, src/batch.rs:37-44,67)."""


HIP_RUNTIME_BOUND = None      # path of the HIP runtime this module loaded ahead of the library, or a note why none was


def _share_hip_runtime_with_torch():
    """ONE HIP runtime per process.  PyTorch-ROCm wheels carry their own libamdhip64.so (SONAME libamdhip64.so.7) and
    ask for it as `libamdhip64.so`; this library asks for `libamdhip64.so.7`.  With torch imported first the loader
    gives us torch's copy (SONAME match); the other way round torch's name does not match /opt/rocm's copy, a second
    runtime is loaded and finds no device ("No HIP GPUs are available").  A process that will use both (device tensors
    handed to the *_device entry points) must share one: when torch is installed and not imported yet, its copy is
    loaded first.  ONLY libamdhip64.so is loaded -- its RPATH brings torch's own HSA runtime with it; loading the two
    separately and ignoring a failure could bind the system HIP runtime to torch's HSA (or the reverse), a version mix
    that only shows as a GPU initialisation failure much later.  What was bound is kept in HIP_RUNTIME_BOUND (bench.py
    prints it); a failure is a warning, not silence.  SSA_NO_TORCH_HIP_PRELOAD=1 turns this off (a process that never
    imports torch does not need it)."""
    global HIP_RUNTIME_BOUND
    import sys
    if "torch" in sys.modules:
        HIP_RUNTIME_BOUND = "torch was imported first: its runtime serves both (SONAME match)"
        return
    if os.environ.get("SSA_NO_TORCH_HIP_PRELOAD"):
        HIP_RUNTIME_BOUND = "system runtime (SSA_NO_TORCH_HIP_PRELOAD)"
        return
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        HIP_RUNTIME_BOUND = "system runtime (torch is not installed)"
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if not os.path.exists(path):
        HIP_RUNTIME_BOUND = "system runtime (torch carries no libamdhip64.so)"
        return
    try:
        C.CDLL(path, mode=C.RTLD_GLOBAL)
        HIP_RUNTIME_BOUND = path
    except OSError as exc:       # not loadable here (no ROCm userland around it): the system runtime serves
        import warnings
        HIP_RUNTIME_BOUND = "system runtime (torch's copy did not load: %s)" % exc
        warnings.warn("schnorr_sig_amd: torch's HIP runtime %s did not load (%s); importing torch later in this process "
                      "will bring a second runtime" % (path, exc))


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "schnorr_sig_amd: HIP library %s is missing; run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    vp, sz, u32, u64p, i32 = C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_uint64), C.c_int
    sigs = {
        "ssa_ctx_create": (i32, [C.POINTER(vp), i32, vp, sz]),
        "ssa_ctx_create_ex": (i32, [C.POINTER(vp), i32, vp, sz, u32, C.c_uint64]),
        "ssa_ctx_info": (i32, [vp, u64p]),
        "ssa_pubkey_many": (i32, [vp, vp, sz, vp]),
        "ssa_pubkey_many_device": (i32, [vp, vp, sz, vp]),
        "ssa_ctx_destroy": (None, [vp]),
        "ssa_strerror": (C.c_char_p, [i32]),
        "ssa_default_params": (vp, []),
        "ssa_ctx_set_stream": (i32, [vp, vp]),
        "ssa_ctx_uses_default_params": (i32, [vp]),
        "ssa_ctx_sync": (i32, [vp]),
        "ssa_ctx_enable_timing": (i32, [vp, i32]),
        "ssa_ctx_read_timing": (i32, [vp, C.c_char_p, C.POINTER(C.c_double), u64p]),
        "ssa_verify": (i32, [vp, vp, vp, vp, sz, u32]),
        "ssa_verify_many": (i32, [vp, vp, vp, vp, vp, vp, sz, sz, sz, u32, vp, u64p]),
        "ssa_verify_batch": (i32, [vp, vp, vp, vp, vp, vp, sz, sz, sz, u32]),
        "ssa_hash_message_many": (i32, [vp, vp, vp, vp, vp, sz, sz, sz, vp]),
        "ssa_rescue_hash_many": (i32, [vp, vp, u32, sz, vp]),
        "ssa_keygen_sign_many": (i32, [vp, vp, vp, vp, vp, sz, sz, sz, vp, vp]),
        "ssa_verify_many_device": (i32, [vp, vp, vp, vp, vp, vp, sz, sz, sz, u32, vp, vp]),
        "ssa_hash_message_many_device": (i32, [vp, vp, vp, vp, vp, sz, sz, sz, vp]),
        "ssa_rescue_hash_many_device": (i32, [vp, vp, u32, sz, vp]),
        "ssa_keygen_sign_many_device": (i32, [vp, vp, vp, vp, vp, sz, sz, sz, vp, vp]),
        "ssa_verify_batch_msm": (i32, [vp, vp, vp, vp, vp, vp, sz, sz, sz, vp]),
        "ssa_verify_batch_msm_device": (i32, [vp, vp, vp, vp, vp, vp, sz, sz, sz, vp, u32, vp]),
        "ssa_verify_keyed_many": (i32, [vp, vp, vp, vp, sz, sz, sz, u32, vp, u64p]),
        "ssa_keyset_create": (i32, [vp, vp, vp, sz, u32, C.POINTER(vp)]),
        "ssa_keyset_create_device": (i32, [vp, vp, vp, sz, u32, C.POINTER(vp)]),
        "ssa_keyset_destroy": (None, [vp]),
        "ssa_keyset_status": (i32, [vp, vp]),
        "ssa_verify_many_indexed": (i32, [vp, vp, vp, vp, vp, vp, sz, sz, sz, u32, vp, u64p]),
        "ssa_verify_many_indexed_device": (i32, [vp, vp, vp, vp, vp, vp, sz, sz, sz, u32, vp, vp]),
        "ssa_multi_create": (i32, [C.POINTER(vp), C.POINTER(i32), i32, vp, sz]),
        "ssa_multi_destroy": (None, [vp]),
        "ssa_multi_verify_many": (i32, [vp, vp, vp, vp, vp, vp, sz, sz, sz, u32, vp, u64p]),
        "ssa_multi_verify_batch_msm": (i32, [vp, vp, vp, vp, vp, vp, sz, sz, sz, vp]),
        "ssa_decompress_many": (i32, [vp, vp, sz, vp, vp, vp]),
        "ssa_decompress_many_device": (i32, [vp, vp, sz, vp, vp, vp]),
        "ssa_debug_arith": (i32, [vp, i32, vp, vp, sz, sz, sz, vp, sz]),
        "ssa_bench_fpmul": (i32, [vp, i32, C.POINTER(C.c_double)]),
        "ssa_debug_chacha20": (i32, [vp, vp, vp, u32, sz, vp]),
        "ssa_verify_batch_msm_partial_device": (i32, [vp, vp, vp, vp, vp, vp, sz, sz, sz, vp, u32, vp]),
        "ssa_verify_batch_msm_partial": (i32, [vp, vp, vp, vp, vp, vp, sz, sz, sz, vp, vp]),
        "ssa_msm_combine_device": (i32, [vp, vp, sz, vp]),
        "ssa_msm_combine": (i32, [vp, vp, sz]),
        "ssa_abi_version": (i32, []),
        "ssa_keygen_sign_many_ex": (i32, [vp, vp, vp, vp, vp, sz, sz, sz, u32, vp, vp]),
        "ssa_keygen_sign_many_ex_device": (i32, [vp, vp, vp, vp, vp, sz, sz, sz, u32, vp, vp]),
        "ssa_compress_many": (i32, [vp, vp, vp, sz, vp, vp]),
        "ssa_compress_many_device": (i32, [vp, vp, vp, sz, vp, vp]),
        "ssa_ctx_stream_release": (i32, [vp, vp]),
        "ssa_ctx_stream_acquire": (i32, [vp, vp]),
        "ssa_debug_fault_after_chunk": (i32, [vp, i32]),
        "ssa_debug_tail_plan": (i32, [u32, u32, u32, i32, u32, sz, u32, vp]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)      # AttributeError here == ABI symbol missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    # a library built from another revision of include/schnorr_sig_amd.h would still resolve every symbol and then
    # read its arguments shifted (round 2 inserted pk_inf into the batch entry points): refuse it at load
    got = lib.ssa_abi_version()
    if got != ABI_VERSION:
        raise ImportError("schnorr_sig_amd: %s has ABI version %d, this binding was written for %d -- rebuild "
                          "(python -c 'import __graft_entry__ as g; g.build(force=True)')" % (LIB_PATH, got, ABI_VERSION))
    return lib, list(sigs)


ABI_VERSION = 5        # SSA_ABI_VERSION of include/schnorr_sig_amd.h
MSM_PARTIAL_WORDS = 24
MSM_RECORD_MAGIC = 0x5353415245430004     # SSA_MSM_RECORD_MAGIC: word 23 of every shard record
_lib, ABI_SYMBOLS = _load()


def _check(rc, what):
    if rc < 0:
        raise RuntimeError("%s failed: %s (%d)" % (what, _lib.ssa_strerror(rc).decode(), rc))
    return rc


def _np_u8(a, cols=None):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if cols is not None:
        a = a.reshape(-1, cols)
    return a


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def pack_messages(messages):
    """list of bytes -> (concatenated uint8 array, uint64 offsets[n+1])."""
    off = np.zeros(len(messages) + 1, dtype=np.uint64)
    if messages:
        off[1:] = np.cumsum([len(m) for m in messages], dtype=np.uint64)
    flat = np.frombuffer(b"".join(bytes(m) for m in messages) + b"\0", dtype=np.uint8).copy()
    return flat, off


class Engine:
    """One ssa_ctx: a device, a stream, the comb table for G and the workspaces."""

    def __init__(self, device=0, params=None, gtab_bits=0, hbm_budget_bytes=0):
        """gtab_bits: window width of the comb for G (16 / 20 / 22 / 24; 0 = the widest whose table fits the budget);
        hbm_budget_bytes: device memory the speed-for-memory tables may take (0 = a tenth of what is free)."""
        self._ctx = C.c_void_p()
        blob = None
        if params is not None:
            blob = (C.c_uint8 * len(params)).from_buffer_copy(bytes(params))
        _check(_lib.ssa_ctx_create_ex(C.byref(self._ctx), int(device), blob, len(params) if params else 0,
                                      int(gtab_bits), int(hbm_budget_bytes)), "ssa_ctx_create_ex")
        self.device = int(device)

    def info(self):
        """what the context holds on the device (ssa_ctx_info)"""
        out = (C.c_uint64 * 8)()
        _check(_lib.ssa_ctx_info(self._ctx, out), "ssa_ctx_info")
        return {"gtab_bits": int(out[0]), "gtab_windows": int(out[1]), "gtab_bytes": int(out[2]),
                "workspace_bytes": int(out[3]), "lane_slice": int(out[4]), "msm_slice": int(out[5]),
                "hbm_budget_bytes": int(out[6]), "two_streams": bool(out[7])}

    def close(self):
        if self._ctx:
            _lib.ssa_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def uses_default_params(self):
        """True when the context runs the builder-default (unpinned) Rescue constants and generator."""
        return bool(_check(_lib.ssa_ctx_uses_default_params(self._ctx), "ssa_ctx_uses_default_params"))

    @staticmethod
    def default_params():
        return C.string_at(_lib.ssa_default_params(), 2816)

    # ---- host-buffer entry points ------------------------------------------------------
    def _msg_args(self, msgs, offsets, n):
        if offsets is not None:
            m = _np_u8(msgs)
            off = np.ascontiguousarray(offsets, dtype=np.uint64)
            assert off.size == n + 1
            return m, off, 0, 0
        m = _np_u8(msgs)
        assert m.ndim == 2 and m.shape[0] == n, "dense messages must be an (n, len) array"
        return m, None, m.shape[1], m.shape[1]

    def verify_many(self, sigs, pks, msgs, offsets=None, check_torsion=True, pk_inf=None, mode=None,
                    sig_flag_byte=False):
        """n x Signature::verify -> (status uint8[n], n_fail).  mode: None/"auto" (wave-per-signature
        kernel for small batches, lane-per-signature kernels for large ones), "lane" or "coop".
        sig_flag_byte: honour byte 48 of the signature as verify_batch does (src/batch.rs:104)."""
        sigs, pks = _np_u8(sigs, 81), _np_u8(pks, 96)
        n = sigs.shape[0]
        assert pks.shape[0] == n
        m, off, stride, mlen = self._msg_args(msgs, offsets, n)
        inf = _np_u8(pk_inf) if pk_inf is not None else None
        status = np.full(n, 255, dtype=np.uint8)
        nfail = C.c_uint64(0)
        flags = (FLAG_CHECK_TORSION if check_torsion else 0) | _MODE_FLAGS[mode] | \
            (FLAG_SIG_FLAG_BYTE if sig_flag_byte else 0)
        _check(_lib.ssa_verify_many(self._ctx, _ptr(sigs), _ptr(pks), _ptr(inf), _ptr(m), _ptr(off), stride,
                                    mlen, n, flags, _ptr(status), C.byref(nfail)), "ssa_verify_many")
        return status, int(nfail.value)

    def verify_batch_status(self, sigs, pks, msgs, offsets=None, check_torsion=False, pk_inf=None):
        sigs, pks = _np_u8(sigs, 81), _np_u8(pks, 96)
        n = sigs.shape[0]
        if pks.shape[0] != n:
            raise MalformedInput("We should have the same number of signatures than public keys")
        m, off, stride, mlen = self._msg_args(msgs, offsets, n)
        inf = _np_u8(pk_inf) if pk_inf is not None else None
        return _check(_lib.ssa_verify_batch(self._ctx, _ptr(sigs), _ptr(pks), _ptr(inf), _ptr(m), _ptr(off), stride,
                                            mlen, n, FLAG_CHECK_TORSION if check_torsion else 0), "ssa_verify_batch")

    def verify_batch_msm(self, sigs, pks, msgs, offsets=None, coeffs=None, pk_inf=None):
        """verify_batch as the reference runs it (random linear combination + MSM); one status."""
        sigs, pks = _np_u8(sigs, 81), _np_u8(pks, 96)
        n = sigs.shape[0]
        if pks.shape[0] != n:
            raise MalformedInput("We should have the same number of signatures than public keys")
        m, off, stride, mlen = self._msg_args(msgs, offsets, n)
        c = _np_u8(coeffs, 32) if coeffs is not None else None
        inf = _np_u8(pk_inf) if pk_inf is not None else None
        return _check(_lib.ssa_verify_batch_msm(self._ctx, _ptr(sigs), _ptr(pks), _ptr(inf), _ptr(m), _ptr(off),
                                                stride, mlen, n, _ptr(c)), "ssa_verify_batch_msm")

    def verify_batch_msm_device(self, d_sigs, d_pks, d_msgs, n, msg_len, d_coeffs, coeff_bytes, d_verdict,
                                msg_stride=None, d_offsets=0, d_pk_inf=0):
        _check(_lib.ssa_verify_batch_msm_device(self._ctx, d_sigs, d_pks, d_pk_inf or None, d_msgs, d_offsets or None,
                                                msg_stride if msg_stride is not None else msg_len, msg_len, n,
                                                d_coeffs, coeff_bytes, d_verdict), "ssa_verify_batch_msm_device")

    # ---- MSM-form verdict across processes: per-shard records + combination (include/schnorr_sig_amd.h) ----
    def verify_batch_msm_partial_device(self, d_sigs, d_pks, d_msgs, n, msg_len, d_coeffs, coeff_bytes, d_partial24,
                                        msg_stride=None, d_offsets=0, d_pk_inf=0):
        """this rank's shard -> one 24-word record at device address d_partial24 (enqueued on the stream)"""
        _check(_lib.ssa_verify_batch_msm_partial_device(
            self._ctx, d_sigs, d_pks, d_pk_inf or None, d_msgs, d_offsets or None,
            msg_stride if msg_stride is not None else msg_len, msg_len, n, d_coeffs or None, coeff_bytes, d_partial24),
            "ssa_verify_batch_msm_partial_device")

    def verify_batch_msm_partial(self, sigs, pks, msgs, offsets=None, coeffs=None, pk_inf=None):
        """host buffers -> the shard's record as uint64[24]"""
        sigs, pks = _np_u8(sigs, 81), _np_u8(pks, 96)
        n = sigs.shape[0]
        if n:
            m, off, stride, mlen = self._msg_args(msgs, offsets, n)
        else:
            m, off, stride, mlen = None, None, 0, 0
        c = _np_u8(coeffs, 32) if coeffs is not None else None
        inf = _np_u8(pk_inf) if pk_inf is not None else None
        out = np.zeros(MSM_PARTIAL_WORDS, dtype=np.uint64)
        _check(_lib.ssa_verify_batch_msm_partial(self._ctx, _ptr(sigs) if n else None, _ptr(pks) if n else None,
                                                 _ptr(inf), _ptr(m), _ptr(off), stride, mlen, n, _ptr(c), _ptr(out)),
               "ssa_verify_batch_msm_partial")
        return out

    def msm_combine_device(self, d_parts24, k, d_verdict):
        _check(_lib.ssa_msm_combine_device(self._ctx, d_parts24, k, d_verdict), "ssa_msm_combine_device")

    def msm_combine(self, parts):
        """uint64[k, 24] records (host) -> status of the whole batch"""
        parts = np.ascontiguousarray(parts, dtype=np.uint64).reshape(-1, MSM_PARTIAL_WORDS)
        return _check(_lib.ssa_msm_combine(self._ctx, _ptr(parts), parts.shape[0]), "ssa_msm_combine")

    def debug_chacha20(self, key32, nonce12, counter0, n_blocks):
        """keystream blocks of the generator the MSM coefficients come from (RFC 8439 block function)"""
        key, nonce = _np_u8(bytearray(key32)), _np_u8(bytearray(nonce12))
        assert key.size == 32 and nonce.size == 12
        out = np.zeros(64 * n_blocks, dtype=np.uint8)
        _check(_lib.ssa_debug_chacha20(self._ctx, _ptr(key), _ptr(nonce), C.c_uint32(counter0), C.c_size_t(n_blocks),
                                       _ptr(out)), "ssa_debug_chacha20")
        return out.tobytes()

    def verify_one(self, sig81, pk96, message, check_torsion=True, pk_is_identity=False):
        sig, pk = _np_u8(bytearray(sig81)), _np_u8(bytearray(pk96))
        msg = _np_u8(bytearray(bytes(message) + b"\0"))
        if pk_is_identity:   # ssa_verify has no identity marker: one-element ssa_verify_many
            st, _ = self.verify_many(sig, pk, msg, offsets=np.array([0, len(message)], np.uint64),
                                     check_torsion=check_torsion, pk_inf=np.ones(1, np.uint8))
            return int(st[0])
        return _check(_lib.ssa_verify(self._ctx, _ptr(sig), _ptr(pk), _ptr(msg), len(message),
                                      FLAG_CHECK_TORSION if check_torsion else 0), "ssa_verify")

    def hash_message_many(self, sigs, pks, msgs, offsets=None):
        sigs, pks = _np_u8(sigs, 81), _np_u8(pks, 96)
        n = sigs.shape[0]
        m, off, stride, mlen = self._msg_args(msgs, offsets, n)
        out = np.zeros((n, 32), dtype=np.uint8)
        _check(_lib.ssa_hash_message_many(self._ctx, _ptr(sigs), _ptr(pks), _ptr(m), _ptr(off), stride, mlen, n,
                                          _ptr(out)), "ssa_hash_message_many")
        return out

    def rescue_hash_many(self, felts):
        f = np.ascontiguousarray(felts, dtype=np.uint64)
        assert f.ndim == 2
        out = np.zeros((f.shape[0], 4), dtype=np.uint64)
        _check(_lib.ssa_rescue_hash_many(self._ctx, _ptr(f), f.shape[1], f.shape[0], _ptr(out)),
               "ssa_rescue_hash_many")
        return out

    def keygen_sign_many(self, sks, nonces, msgs, offsets=None, constant_time=False, keyed=False):
        """pk_i = [sk_i]G and sig_i = sign(sk_i, nonce_i, msg_i) -> (pks uint8[n, 96], sigs uint8[n, 81]).
        constant_time: SSA_FLAG_SIGN_CT (same bytes, no secret-dependent branch or address).
        keyed: the second array holds 130-byte KeyedSignature records pk(49) || sig(81) instead."""
        sks, nonces = _np_u8(sks, 32), _np_u8(nonces, 32)
        n = sks.shape[0]
        m, off, stride, mlen = self._msg_args(msgs, offsets, n)
        pks = np.zeros((n, 96), dtype=np.uint8)
        sigs = np.zeros((n, 130 if keyed else 81), dtype=np.uint8)
        flags = (FLAG_SIGN_CT if constant_time else 0) | (FLAG_SIGN_KEYED if keyed else 0)
        _check(_lib.ssa_keygen_sign_many_ex(self._ctx, _ptr(sks), _ptr(nonces), _ptr(m), _ptr(off), stride, mlen, n,
                                            flags, _ptr(pks), _ptr(sigs)), "ssa_keygen_sign_many_ex")
        return pks, sigs

    def pubkey_many(self, sks):
        """PublicKey::from(&PrivateKey) for n canonical non-zero scalars -> uint8[n, 96]: one constant-time base
        multiplication per key, nothing else derived from the secret (ssa_pubkey_many)"""
        sks = _np_u8(sks, 32)
        n = sks.shape[0]
        pks = np.zeros((n, 96), dtype=np.uint8)
        _check(_lib.ssa_pubkey_many(self._ctx, _ptr(sks), n, _ptr(pks)), "ssa_pubkey_many")
        return pks

    def pubkey_many_device(self, d_sks, n, d_pks):
        _check(_lib.ssa_pubkey_many_device(self._ctx, C.c_void_p(d_sks), n, C.c_void_p(d_pks)), "ssa_pubkey_many_device")

    def compress_many(self, pks, pk_inf=None):
        """PublicKey::to_bytes for n affine keys -> (uint8[n, 49], status uint8[n])"""
        pks = _np_u8(pks, 96)
        n = pks.shape[0]
        inf = _np_u8(pk_inf) if pk_inf is not None else None
        out = np.zeros((n, 49), dtype=np.uint8)
        st = np.full(n, 255, dtype=np.uint8)
        _check(_lib.ssa_compress_many(self._ctx, _ptr(pks), _ptr(inf), n, _ptr(out), _ptr(st)), "ssa_compress_many")
        return out, st

    def verify_keyed_many(self, keyed, msgs, offsets=None, check_torsion=True):
        """n x KeyedSignature::verify on 130-byte records pk(49) || sig(81) -> (status, n_fail)."""
        kd = _np_u8(keyed, 130)
        n = kd.shape[0]
        m, off, stride, mlen = self._msg_args(msgs, offsets, n)
        status = np.full(n, 255, dtype=np.uint8)
        nfail = C.c_uint64(0)
        _check(_lib.ssa_verify_keyed_many(self._ctx, _ptr(kd), _ptr(m), _ptr(off), stride, mlen, n,
                                          FLAG_CHECK_TORSION if check_torsion else 0, _ptr(status),
                                          C.byref(nfail)), "ssa_verify_keyed_many")
        return status, int(nfail.value)

    def decompress_many(self, compressed):
        """n x 49-byte compressed points -> (pks uint8[n,96], is_identity uint8[n], status uint8[n])."""
        c = _np_u8(compressed, 49)
        n = c.shape[0]
        pks = np.zeros((n, 96), dtype=np.uint8)
        inf = np.zeros(n, dtype=np.uint8)
        st = np.full(n, 255, dtype=np.uint8)
        _check(_lib.ssa_decompress_many(self._ctx, _ptr(c), n, _ptr(pks), _ptr(inf), _ptr(st)),
               "ssa_decompress_many")
        return pks, inf, st

    # ---- keyed context (many signatures by few signers) ---------------------------------
    def keyset_create(self, pks, pk_inf=None, kind="auto"):
        """-> KeySet: subgroup check and tables done once per key.  kind: "auto" (combs while they fit the context's HBM
        budget, else -- or when their allocation fails -- the ladder tables), "comb" (100 MB per key, no doublings at
        verification time) or "ladder" (4 KB per key)"""
        pks = _np_u8(pks, 96)
        inf = _np_u8(pk_inf) if pk_inf is not None else None
        ks = C.c_void_p()
        _check(_lib.ssa_keyset_create(self._ctx, _ptr(pks), _ptr(inf), pks.shape[0], KEYSET_KINDS[kind], C.byref(ks)),
               "ssa_keyset_create")
        return KeySet(self, ks, pks.shape[0])

    def keyset_create_device(self, d_pks, m, d_pk_inf=0, kind="auto"):
        ks = C.c_void_p()
        _check(_lib.ssa_keyset_create_device(self._ctx, d_pks, d_pk_inf or None, m, KEYSET_KINDS[kind], C.byref(ks)),
               "ssa_keyset_create_device")
        return KeySet(self, ks, m)

    def keyset_destroy(self, ks):
        ks.close()

    def keyset_status(self, ks, m=None):
        st = np.full(ks.m if m is None else m, 255, dtype=np.uint8)
        _check(_lib.ssa_keyset_status(ks.handle, _ptr(st)), "ssa_keyset_status")
        return st

    def verify_many_indexed(self, ks, key_idx, sigs, msgs, offsets=None, check_torsion=True, sig_flag_byte=False):
        """n x Signature::verify against key key_idx[i] of the key set -> (status uint8[n], n_fail)"""
        sigs = _np_u8(sigs, 81)
        n = sigs.shape[0]
        idx = np.ascontiguousarray(key_idx, dtype=np.uint32)
        assert idx.size == n
        m, off, stride, mlen = self._msg_args(msgs, offsets, n)
        status = np.full(n, 255, dtype=np.uint8)
        nfail = C.c_uint64(0)
        flags = (FLAG_CHECK_TORSION if check_torsion else 0) | (FLAG_SIG_FLAG_BYTE if sig_flag_byte else 0)
        _check(_lib.ssa_verify_many_indexed(self._ctx, ks.handle, _ptr(idx), _ptr(sigs), _ptr(m), _ptr(off), stride, mlen, n,
                                            flags, _ptr(status), C.byref(nfail)), "ssa_verify_many_indexed")
        return status, int(nfail.value)

    def verify_many_indexed_device(self, ks, d_key_idx, d_sigs, d_msgs, n, msg_len, d_status, d_nfail, msg_stride=None,
                                   d_offsets=0, check_torsion=True):
        _check(_lib.ssa_verify_many_indexed_device(self._ctx, ks.handle, d_key_idx, d_sigs, d_msgs, d_offsets or None,
                                                   msg_stride if msg_stride is not None else msg_len, msg_len, n,
                                                   FLAG_CHECK_TORSION if check_torsion else 0, d_status, d_nfail),
               "ssa_verify_many_indexed_device")

    # ---- device-buffer entry points (raw device addresses, e.g. torch.Tensor.data_ptr()) ----
    def set_stream(self, hip_stream):
        _check(_lib.ssa_ctx_set_stream(self._ctx, C.c_void_p(hip_stream or 0)), "ssa_ctx_set_stream")

    def sync(self):
        _check(_lib.ssa_ctx_sync(self._ctx), "ssa_ctx_sync")

    def stream_release(self, consumer_stream):
        """stream `consumer_stream` (a hipStream_t handle, e.g. torch.cuda.current_stream().cuda_stream) waits for
        everything this engine has enqueued so far: its outputs become visible there without a host synchronisation"""
        _check(_lib.ssa_ctx_stream_release(self._ctx, C.c_void_p(consumer_stream or 0)), "ssa_ctx_stream_release")

    def stream_acquire(self, producer_stream):
        """this engine's stream waits for everything enqueued on `producer_stream` so far (inputs written there, e.g.
        by a collective, are visible to the next call on the engine)"""
        _check(_lib.ssa_ctx_stream_acquire(self._ctx, C.c_void_p(producer_stream or 0)), "ssa_ctx_stream_acquire")

    def debug_fault_after_chunk(self, chunk):
        """tests: the NEXT pipelined host-buffer upload fails after chunk `chunk` (one shot; < 0 disarms)"""
        _check(_lib.ssa_debug_fault_after_chunk(self._ctx, int(chunk)), "ssa_debug_fault_after_chunk")

    def verify_many_device(self, d_sigs, d_pks, d_msgs, n, msg_len, d_status, d_nfail, msg_stride=None,
                           d_offsets=0, d_pk_inf=0, check_torsion=False, mode=None, sig_flag_byte=False):
        """sig_flag_byte=True with check_torsion=False is what ssa_verify_batch runs (src/batch.rs:104: R is decompressed
        with the flag byte of sig.x, no subgroup check)"""
        flags = (FLAG_CHECK_TORSION if check_torsion else 0) | _MODE_FLAGS[mode] | \
            (FLAG_SIG_FLAG_BYTE if sig_flag_byte else 0)
        _check(_lib.ssa_verify_many_device(self._ctx, d_sigs, d_pks, d_pk_inf or None, d_msgs, d_offsets or None,
                                           msg_stride if msg_stride is not None else msg_len, msg_len, n,
                                           flags, d_status, d_nfail), "ssa_verify_many_device")

    def keygen_sign_many_device(self, d_sks, d_nonces, d_msgs, n, msg_len, d_pks, d_sigs, msg_stride=None,
                                d_offsets=0, constant_time=False, keyed=False):
        flags = (FLAG_SIGN_CT if constant_time else 0) | (FLAG_SIGN_KEYED if keyed else 0)
        _check(_lib.ssa_keygen_sign_many_ex_device(self._ctx, d_sks, d_nonces, d_msgs, d_offsets or None,
                                                   msg_stride if msg_stride is not None else msg_len, msg_len, n, flags,
                                                   d_pks or None, d_sigs), "ssa_keygen_sign_many_ex_device")

    def compress_many_device(self, d_pks, n, d_out, d_pk_inf=0, d_status=0):
        _check(_lib.ssa_compress_many_device(self._ctx, d_pks, d_pk_inf or None, n, d_out, d_status or None),
               "ssa_compress_many_device")

    def rescue_hash_many_device(self, d_felts, per_row, n, d_out):
        _check(_lib.ssa_rescue_hash_many_device(self._ctx, d_felts, per_row, n, d_out),
               "ssa_rescue_hash_many_device")

    def hash_message_many_device(self, d_sigs, d_pks, d_msgs, n, msg_len, d_out, msg_stride=None, d_offsets=0):
        _check(_lib.ssa_hash_message_many_device(self._ctx, d_sigs, d_pks, d_msgs, d_offsets or None,
                                                 msg_stride if msg_stride is not None else msg_len, msg_len, n,
                                                 d_out), "ssa_hash_message_many_device")

    def enable_timing(self, on=True):
        _check(_lib.ssa_ctx_enable_timing(self._ctx, int(on)), "ssa_ctx_enable_timing")

    def read_timing(self, kernel):
        avg = C.c_double(0)
        cnt = C.c_uint64(0)
        _check(_lib.ssa_ctx_read_timing(self._ctx, kernel.encode(), C.byref(avg), C.byref(cnt)),
               "ssa_ctx_read_timing")
        return avg.value, int(cnt.value)

    def host_path_probe(self, sigs_t, pks_t, msgs_t, n, reps=3):
        """PCIe-inclusive rate of the host-buffer entry point ssa_verify_many (what a Rust shim binds): the
        device tensors are copied to ordinary pageable host arrays first, then `reps` timed calls."""
        import time
        hs, hp, hm = (t[:n].cpu().numpy() for t in (sigs_t, pks_t, msgs_t))
        st, nf = self.verify_many(hs, hp, hm, check_torsion=False, mode="lane", sig_flag_byte=True)     # warm-up: workspaces
        t0 = time.perf_counter()
        for _ in range(reps):
            st, nf = self.verify_many(hs, hp, hm, check_torsion=False, mode="lane", sig_flag_byte=True)
        dt = (time.perf_counter() - t0) / reps
        verdict = self.verify_batch_msm(hs, hp, hm)                                  # MSM form, library-drawn coefficients
        t0 = time.perf_counter()
        for _ in range(reps):
            verdict = self.verify_batch_msm(hs, hp, hm)
        dt_msm = (time.perf_counter() - t0) / reps
        return {"entry_point": "ssa_verify_many (host buffers in pageable memory, copied through the library's page-locked "
                               "bounce buffers, uploads in chunks overlapped with the hash kernel)",
                "ms_per_batch": dt * 1e3, "verifications_per_sec": n / dt, "rejected": int(nf),
                "verify_batch_msm_ms_per_batch": dt_msm * 1e3, "verify_batch_msm_signatures_per_sec": n / dt_msm,
                "verify_batch_msm_verdict": int(verdict)}

    # ---- probes --------------------------------------------------------------------------
    def debug_arith(self, op, a, b, out_cols):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64) if b is not None else None
        n = a.shape[0]
        out = np.zeros((n, out_cols), dtype=np.uint64)
        _check(_lib.ssa_debug_arith(self._ctx, op, _ptr(a), _ptr(b), n, a.shape[1],
                                    b.shape[1] if b is not None else 0, _ptr(out), out_cols), "ssa_debug_arith")
        return out

    def bench_fpmul(self, variant):
        v = C.c_double(0)
        _check(_lib.ssa_bench_fpmul(self._ctx, variant, C.byref(v)), "ssa_bench_fpmul")
        return v.value


class KeySet:
    """ssa_keyset handle tied to its Engine: the engine cannot be collected before the key set, and the handle is
    destroyed exactly once (close(), or when the object goes away)."""

    def __init__(self, engine, handle, m):
        self.engine = engine      # keeps the context alive
        self.handle = handle
        self.m = int(m)

    def close(self):
        if self.handle:
            _lib.ssa_keyset_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiEngine:
    """Several GPUs of one node from a single process (ssa_multi_*): contiguous shards, one context
    and one host thread per device, no collective."""

    def __init__(self, devices, params=None):
        self._m = C.c_void_p()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        blob = (C.c_uint8 * len(params)).from_buffer_copy(bytes(params)) if params is not None else None
        _check(_lib.ssa_multi_create(C.byref(self._m), devs, len(devices), blob, len(params) if params else 0),
               "ssa_multi_create")

    def close(self):
        if self._m:
            _lib.ssa_multi_destroy(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def verify_many(self, sigs, pks, msgs, offsets=None, check_torsion=True, pk_inf=None):
        sigs, pks = _np_u8(sigs, 81), _np_u8(pks, 96)
        n = sigs.shape[0]
        m = _np_u8(msgs)
        off = np.ascontiguousarray(offsets, dtype=np.uint64) if offsets is not None else None
        stride = mlen = 0 if off is not None else m.shape[1]
        inf = _np_u8(pk_inf) if pk_inf is not None else None
        status = np.full(n, 255, dtype=np.uint8)
        nfail = C.c_uint64(0)
        _check(_lib.ssa_multi_verify_many(self._m, _ptr(sigs), _ptr(pks), _ptr(inf), _ptr(m), _ptr(off), stride, mlen,
                                          n, FLAG_CHECK_TORSION if check_torsion else 0, _ptr(status),
                                          C.byref(nfail)), "ssa_multi_verify_many")
        return status, int(nfail.value)


    def verify_batch_msm(self, sigs, pks, msgs, offsets=None, coeffs=None, pk_inf=None):
        """verify_batch as the reference runs it, the batch sharded over the devices (one status)."""
        sigs, pks = _np_u8(sigs, 81), _np_u8(pks, 96)
        n = sigs.shape[0]
        m = _np_u8(msgs)
        off = np.ascontiguousarray(offsets, dtype=np.uint64) if offsets is not None else None
        stride = mlen = 0 if off is not None else m.shape[1]
        inf = _np_u8(pk_inf) if pk_inf is not None else None
        c = _np_u8(coeffs, 32) if coeffs is not None else None
        return _check(_lib.ssa_multi_verify_batch_msm(self._m, _ptr(sigs), _ptr(pks), _ptr(inf), _ptr(m), _ptr(off),
                                                      stride, mlen, n, _ptr(c)), "ssa_multi_verify_batch_msm")


_default_engine = None


def debug_tail_plan(waves, n, check_torsion=False, pieces=5, gens=1, uniform=False, min_main=0):
    """host logic of ssa_k_verify's end game (no device needed): the launch plan for n lanes with `waves` resident waves.
    Returns a dict: n_pieces, tail_groups, main_blocks, grid_blocks, pieces = [(pass, first, last, lo, hi)], whole"""
    out = np.zeros(14, dtype=np.uint32)
    flags = 1 if check_torsion else 0            # SSA_FLAG_CHECK_TORSION
    _check(_lib.ssa_debug_tail_plan(int(waves), int(pieces), int(gens), int(bool(uniform)), int(min_main), int(n), flags,
                                    out.ctypes.data), "ssa_debug_tail_plan")

    def dec(d):
        d = int(d)
        return (d & 1, bool(d & 2), bool(d & 4), (d >> 8) & 0xff, (d >> 16) & 0xff)
    return {"n_pieces": int(out[0]), "tail_groups": int(out[1]), "main_blocks": int(out[2]), "grid_blocks": int(out[3]),
            "pieces": [dec(out[4 + k]) for k in range(int(out[0]))], "whole": [dec(out[12]), dec(out[13])]}


def default_engine():
    global _default_engine
    if _default_engine is None:
        _default_engine = Engine(0)
    return _default_engine


# ------------------------------------------------------------------------------------------
# Object mirror of the reference's types (thin; bytes in, bytes out)
# ------------------------------------------------------------------------------------------
class PublicKey:
    """PublicKey(AffinePoint) (src/public.rs:24): 96 bytes affine x || y, canonical LE limbs."""

    def __init__(self, affine96, is_identity=False):
        b = bytes(affine96)
        if len(b) != AFFINE_PUBLIC_KEY_LENGTH:
            raise ValueError("PublicKey needs 96 bytes of affine coordinates")
        self.affine = b
        self.is_identity = bool(is_identity)   # AffinePoint::identity() is a valid key (src/public.rs:95-101)

    def verify_signature(self, signature, message):  # src/signature.rs:170-176
        return signature.verify(message, self)

    @classmethod
    def from_private(cls, sk, engine=None):  # impl From<&PrivateKey> for PublicKey, src/public.rs:26-32: [sk]G on the GPU
        return KeyPair.from_private(sk, engine).public_key

    @classmethod
    def from_bytes(cls, b49, engine=None):
        """PublicKey::from_bytes (src/public.rs:54-56): None when decompression fails.  The identity
        encoding decodes to a key whose affine bytes are zero and `is_identity` is set."""
        b = bytes(b49)
        if len(b) != PUBLIC_KEY_LENGTH:
            raise ValueError("compressed public key needs 49 bytes")
        pks, inf, st = (engine or default_engine()).decompress_many(np.frombuffer(b, np.uint8))
        if st[0] != 0:
            return None
        return cls(pks[0].tobytes(), is_identity=bool(inf[0]))

    def to_bytes(self, engine=None):
        """PublicKey::to_bytes (src/public.rs:49-51): x || flag byte, through ssa_compress_many."""
        out, st = (engine or default_engine()).compress_many(np.frombuffer(self.affine, np.uint8),
                                                             pk_inf=np.array([1 if self.is_identity else 0], np.uint8))
        if st[0] != OK:
            raise MalformedInput("PublicKey holds a non-canonical limb")
        return out[0].tobytes()

    def __eq__(self, o):
        return isinstance(o, PublicKey) and o.affine == self.affine and o.is_identity == self.is_identity


class PrivateKey:
    """PrivateKey(Scalar) (src/private.rs:25): 32 bytes LE, canonical, non-zero."""

    def __init__(self, scalar32):
        b = bytes(scalar32)
        v = int.from_bytes(b, "little")
        if len(b) != 32 or v == 0 or v >= Q:
            raise ValueError("invalid private key encoding")   # from_bytes is_none, src/private.rs:74-76
        self.bytes = b

    @classmethod
    def new(cls, rng):  # src/private.rs:49-57
        while True:
            v = int.from_bytes(rng(64), "little") % Q
            if v:
                return cls(v.to_bytes(32, "little"))

    def to_bytes(self):
        return self.bytes

    @classmethod
    def from_bytes(cls, b32):  # src/private.rs:74-76: None (CtOption is_none) for a non-canonical or zero scalar
        b = bytes(b32)
        if len(b) != PRIVATE_KEY_LENGTH:
            raise ValueError("private key needs 32 bytes")
        v = int.from_bytes(b, "little")
        return None if v == 0 or v >= Q else cls(b)

    @classmethod
    def from_seed(cls, seed64):  # src/private.rs:79-82: Scalar::from_bytes_wide, None for 0
        b = bytes(seed64)
        if len(b) != 64:
            raise ValueError("seed needs 64 bytes")
        v = int.from_bytes(b, "little") % Q
        return None if v == 0 else cls(v.to_bytes(32, "little"))

    # PrivateKey::sign / sign_and_bind_pkey (src/signature.rs:62-110): "it is faster to sign with a KeyPair" -- here too:
    # the public key is recomputed on the GPU first (PublicKey::from(self))
    def sign(self, message, rng, engine=None):
        return KeyPair.from_private(self, engine).sign(message, rng, engine)

    def sign_and_bind_pkey(self, message, rng, engine=None):
        return KeyPair.from_private(self, engine).sign_and_bind_pkey(message, rng, engine)

    def __eq__(self, o):
        return isinstance(o, PrivateKey) and o.bytes == self.bytes

    def __hash__(self):
        return hash(self.bytes)


class Signature:
    """Signature{x: CompressedPoint, e: Scalar} (src/signature.rs:34-40), 81 bytes."""

    def __init__(self, sig81):
        b = bytes(sig81)
        if len(b) != SIGNATURE_LENGTH:
            raise ValueError("Signature needs 81 bytes")
        self.bytes = b

    @classmethod
    def from_bytes(cls, b):  # src/signature.rs:217-227: None when e is not canonical
        if int.from_bytes(bytes(b)[49:81], "little") >= Q:
            return None
        return cls(b)

    def to_bytes(self):
        return self.bytes

    def verify(self, message, pkey, engine=None):
        """Ok -> None; otherwise raises SignatureError (src/signature.rs:181-205)."""
        st = (engine or default_engine()).verify_one(self.bytes, pkey.affine, message, check_torsion=True,
                                                     pk_is_identity=pkey.is_identity)
        if st == OK:
            return None
        if st == INVALID_PUBLIC_KEY:
            raise SignatureError(SignatureError.InvalidPublicKey)
        if st == INVALID_SIGNATURE:
            raise SignatureError(SignatureError.InvalidSignature)
        raise MalformedInput("non-canonical field element or scalar (the reference panics here)")

    def __eq__(self, o):
        return isinstance(o, Signature) and o.bytes == self.bytes


class KeyedSignature:
    """KeyedSignature{public_key, signature} (src/signature.rs:55-60); wire form pk(49) || sig(81)."""

    def __init__(self, public_key, signature):
        self.public_key = public_key
        self.signature = signature

    def to_bytes(self, engine=None):  # src/signature.rs:236-243
        return self.public_key.to_bytes(engine) + self.signature.to_bytes()

    @classmethod
    def from_bytes(cls, b130, engine=None):  # src/signature.rs:246-271: None unless both halves decode
        b = bytes(b130)
        if len(b) != KEYED_SIGNATURE_LENGTH:
            raise ValueError("KeyedSignature needs 130 bytes")
        pk = PublicKey.from_bytes(b[:49], engine)
        sig = Signature.from_bytes(b[49:])
        if pk is None or sig is None:
            return None
        return cls(pk, sig)

    def verify(self, message, engine=None):  # src/signature.rs:232-234
        return self.signature.verify(message, self.public_key, engine)

    def __eq__(self, o):
        return isinstance(o, KeyedSignature) and o.public_key == self.public_key and o.signature == self.signature


class KeyPair:
    """KeyPair{private_key, public_key} (src/keypair.rs:48-53)."""

    def __init__(self, private_key, public_key):
        self.private_key = private_key
        self.public_key = public_key

    @classmethod
    def new(cls, rng, engine=None):  # src/keypair.rs:57-65
        return cls.from_private(PrivateKey.new(rng), engine)

    def to_bytes(self):  # src/keypair.rs:73-75: the private key only; the public key is rebuilt when decoding
        return self.private_key.to_bytes()

    @classmethod
    def from_bytes(cls, b32, engine=None):  # src/keypair.rs:78-89
        sk = PrivateKey.from_bytes(b32)
        return None if sk is None else cls.from_private(sk, engine)

    @classmethod
    def from_seed(cls, seed64, engine=None):  # src/keypair.rs:92-103
        sk = PrivateKey.from_seed(seed64)
        return None if sk is None else cls.from_private(sk, engine)

    def __eq__(self, o):
        return isinstance(o, KeyPair) and o.private_key == self.private_key and o.public_key == self.public_key

    @classmethod
    def from_private(cls, sk, engine=None):  # PublicKey::from(&PrivateKey), src/public.rs:26-32
        eng = engine or default_engine()
        pks = eng.pubkey_many(np.frombuffer(sk.bytes, np.uint8))     # [sk]G and nothing else (no nonce, no response)
        return cls(sk, PublicKey(pks[0].tobytes()))

    def _nonce(self, rng):  # Scalar::random: 64 random bytes mod q, never 0
        while True:
            v = int.from_bytes(rng(64), "little") % Q
            if v:
                return v.to_bytes(32, "little")

    def sign(self, message, rng, engine=None):  # src/signature.rs:114-129 (constant-time, like the reference)
        eng = engine or default_engine()
        msg = np.frombuffer(bytes(message) + b"\0", np.uint8).copy()
        off = np.array([0, len(message)], dtype=np.uint64)
        _, sigs = eng.keygen_sign_many(np.frombuffer(self.private_key.bytes, np.uint8),
                                       np.frombuffer(self._nonce(rng), np.uint8), msg, offsets=off, constant_time=True)
        return Signature(sigs[0].tobytes())

    def verify_signature(self, signature, message):  # src/signature.rs:159-165
        return signature.verify(message, self.public_key)

    def sign_and_bind_pkey(self, message, rng, engine=None):  # src/signature.rs:132-156
        """the engine emits the 130-byte record pk(49) || sig(81) itself (SSA_FLAG_SIGN_KEYED)"""
        eng = engine or default_engine()
        msg = np.frombuffer(bytes(message) + b"\0", np.uint8).copy()
        off = np.array([0, len(message)], dtype=np.uint64)
        _, recs = eng.keygen_sign_many(np.frombuffer(self.private_key.bytes, np.uint8),
                                       np.frombuffer(self._nonce(rng), np.uint8), msg, offsets=off, constant_time=True,
                                       keyed=True)
        return KeyedSignature(self.public_key, Signature(recs[0, 49:].tobytes()))


def verify_batch(signatures, public_keys, messages, rng=None, engine=None, msm=False):
    """verify_batch (src/batch.rs:31-50): Ok -> None, else raises SignatureError.
    msm=False: AND of exact per-signature checks (`rng` unused; DESIGN.md, divergence classes).
    msm=True: the reference's own algorithm on the GPU -- random linear combination (coefficients from
    `rng(32)` per signature, or getrandom when rng is None) and a 2n-point MSM."""
    if len(signatures) != len(public_keys):
        raise MalformedInput("We should have the same number of signatures than public keys")
    if len(messages) != len(public_keys):
        raise MalformedInput("We should have the same number of messages than public keys")
    if not signatures:
        return None
    eng = engine or default_engine()
    sigs = np.frombuffer(b"".join(s.bytes for s in signatures), np.uint8)
    pks = np.frombuffer(b"".join(p.affine for p in public_keys), np.uint8)
    inf = np.array([1 if p.is_identity else 0 for p in public_keys], np.uint8)
    flat, off = pack_messages(messages)
    if msm:
        coeffs = None
        if rng is not None:
            coeffs = np.frombuffer(b"".join((int.from_bytes(rng(64), "little") % Q).to_bytes(32, "little")
                                            for _ in signatures), np.uint8)
        st = eng.verify_batch_msm(sigs, pks, flat, offsets=off, coeffs=coeffs, pk_inf=inf)
    else:
        st = eng.verify_batch_status(sigs, pks, flat, offsets=off, check_torsion=False, pk_inf=inf)
    if st == OK:
        return None
    if st == MALFORMED:
        raise MalformedInput("undecodable signature in batch (the reference panics here)")
    raise SignatureError(SignatureError.InvalidSignature)
