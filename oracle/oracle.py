"""ctypes front-end of the C oracle (oracle/schnorr_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
PARAMS_BLOB = os.path.join(_ROOT, "schnorr-sig_amd", "params", "params_default.bin")

_u8p = C.POINTER(C.c_uint8)
_u64p = C.POINTER(C.c_uint64)


def build(native=False):
    target = "native" if native else "all"
    subprocess.check_call(["make", "-s", "-C", _HERE, target])
    name = "libschnorr_oracle_native.so" if native else "libschnorr_oracle.so"
    return os.path.join(_HERE, name)


def _ptr(a, typ):
    return a.ctypes.data_as(typ) if a is not None else None


class Oracle:
    def __init__(self, blob_path=PARAMS_BLOB, native=False, blob=None):
        path = os.path.join(_HERE, "libschnorr_oracle_native.so" if native else "libschnorr_oracle.so")
        src = os.path.join(_HERE, "schnorr_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            path = build(native)
        self.lib = C.CDLL(path)
        if blob is None:
            blob = open(blob_path, "rb").read()
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        self.lib.so_init.argtypes = [C.c_void_p, C.c_size_t]
        rc = self.lib.so_init(buf, len(blob))
        if rc != 0:
            raise RuntimeError("so_init failed: %d" % rc)
        self.lib.so_hw_threads.restype = C.c_int

    # ---- field / curve ------------------------------------------------------
    def fp6_mul(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        o = np.zeros(6, dtype=np.uint64)
        self.lib.so_fp6_mul(_ptr(a, _u64p), _ptr(b, _u64p), _ptr(o, _u64p))
        return o

    def fp6_sqr(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        o = np.zeros(6, dtype=np.uint64)
        self.lib.so_fp6_sqr(_ptr(a, _u64p), _ptr(o, _u64p))
        return o

    def fp6_inv(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        o = np.zeros(6, dtype=np.uint64)
        ok = self.lib.so_fp6_inv(_ptr(a, _u64p), _ptr(o, _u64p))
        return o if ok else None

    def fp6_sqrt(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        o = np.zeros(6, dtype=np.uint64)
        ok = self.lib.so_fp6_sqrt(_ptr(a, _u64p), _ptr(o, _u64p))
        return o if ok else None

    def point_mul(self, k, pt):
        """k: int; pt: (x6, y6) or None -> (x6, y6) tuples of ints or None."""
        kb = np.frombuffer(int(k).to_bytes(32, "little"), dtype=np.uint8).copy()
        inf = pt is None
        px = np.array(pt[0] if not inf else [0] * 6, dtype=np.uint64)
        py = np.array(pt[1] if not inf else [0] * 6, dtype=np.uint64)
        ox, oy = np.zeros(6, np.uint64), np.zeros(6, np.uint64)
        oinf = C.c_int(0)
        self.lib.so_point_mul(_ptr(kb, _u8p), _ptr(px, _u64p), _ptr(py, _u64p), int(inf),
                              _ptr(ox, _u64p), _ptr(oy, _u64p), C.byref(oinf))
        if oinf.value:
            return None
        return tuple(int(v) for v in ox), tuple(int(v) for v in oy)

    def point_add(self, a, b):
        def unpack(p):
            if p is None:
                return np.zeros(6, np.uint64), np.zeros(6, np.uint64), 1
            return np.array(p[0], np.uint64), np.array(p[1], np.uint64), 0
        ax, ay, ai = unpack(a)
        bx, by, bi = unpack(b)
        ox, oy = np.zeros(6, np.uint64), np.zeros(6, np.uint64)
        oinf = C.c_int(0)
        self.lib.so_point_add(_ptr(ax, _u64p), _ptr(ay, _u64p), ai, _ptr(bx, _u64p), _ptr(by, _u64p),
                              bi, _ptr(ox, _u64p), _ptr(oy, _u64p), C.byref(oinf))
        if oinf.value:
            return None
        return tuple(int(v) for v in ox), tuple(int(v) for v in oy)

    def is_torsion_free(self, pt):
        px = np.array(pt[0], np.uint64)
        py = np.array(pt[1], np.uint64)
        return bool(self.lib.so_is_torsion_free(_ptr(px, _u64p), _ptr(py, _u64p), 0))

    def on_curve(self, pt):
        px = np.array(pt[0], np.uint64)
        py = np.array(pt[1], np.uint64)
        return bool(self.lib.so_on_curve(_ptr(px, _u64p), _ptr(py, _u64p)))

    # ---- Rescue -------------------------------------------------------------
    def rescue_permutation(self, state):
        s = np.array(state, dtype=np.uint64)
        self.lib.so_rescue_permutation(_ptr(s, _u64p))
        return s

    def hash_field(self, felts):
        f = np.ascontiguousarray(felts, dtype=np.uint64)
        d = np.zeros(4, np.uint64)
        self.lib.so_hash_field(_ptr(f, _u64p), C.c_size_t(f.size), _ptr(d, _u64p))
        return d

    def hash_field_many(self, felts2d):
        f = np.ascontiguousarray(felts2d, dtype=np.uint64)
        out = np.zeros((f.shape[0], 4), np.uint64)
        for i in range(f.shape[0]):
            self.lib.so_hash_field(_ptr(f[i], _u64p), C.c_size_t(f.shape[1]), _ptr(out[i], _u64p))
        return out

    def hash_message(self, rx48, pk96, msg):
        rx = np.frombuffer(bytes(rx48), np.uint8).copy()
        pk = np.frombuffer(bytes(pk96), np.uint8).copy()
        m = np.frombuffer(bytes(msg) + b"\0", np.uint8).copy()
        o = np.zeros(32, np.uint8)
        self.lib.so_hash_message(_ptr(rx, _u8p), _ptr(pk, _u8p), _ptr(m, _u8p), C.c_size_t(len(msg)),
                                 _ptr(o, _u8p))
        return o.tobytes()

    def scalar_from_digest(self, h32):
        h = np.frombuffer(bytes(h32), np.uint8).copy()
        o = np.zeros(32, np.uint8)
        self.lib.so_scalar_from_digest(_ptr(h, _u8p), _ptr(o, _u8p))
        return o.tobytes()

    # ---- schnorr-sig ----------------------------------------------------------
    def keygen(self, sk32):
        sk = np.frombuffer(bytes(sk32), np.uint8).copy()
        pk = np.zeros(96, np.uint8)
        inf = C.c_int(0)
        self.lib.so_keygen(_ptr(sk, _u8p), _ptr(pk, _u8p), C.byref(inf))
        return pk.tobytes(), bool(inf.value)

    def sign(self, sk32, nonce32, pk96, msg):
        sk = np.frombuffer(bytes(sk32), np.uint8).copy()
        nn = np.frombuffer(bytes(nonce32), np.uint8).copy()
        pk = np.frombuffer(bytes(pk96), np.uint8).copy()
        m = np.frombuffer(bytes(msg) + b"\0", np.uint8).copy()
        sig = np.zeros(81, np.uint8)
        self.lib.so_sign(_ptr(sk, _u8p), _ptr(nn, _u8p), _ptr(pk, _u8p), _ptr(m, _u8p),
                         C.c_size_t(len(msg)), _ptr(sig, _u8p))
        return sig.tobytes()

    def verify(self, sig81, pk96, msg, check_torsion=True, pk_inf=False):
        sig = np.frombuffer(bytes(sig81), np.uint8).copy()
        pk = np.frombuffer(bytes(pk96), np.uint8).copy()
        m = np.frombuffer(bytes(msg) + b"\0", np.uint8).copy()
        return int(self.lib.so_verify(_ptr(sig, _u8p), _ptr(pk, _u8p), int(pk_inf), _ptr(m, _u8p),
                                      C.c_size_t(len(msg)), int(check_torsion)))

    @staticmethod
    def _msgs(msgs, offsets, n):
        msgs = np.ascontiguousarray(msgs, dtype=np.uint8)
        if offsets is not None:
            off = np.ascontiguousarray(offsets, dtype=np.uint64)
            return msgs, off, 0, 0
        assert msgs.ndim == 2 and msgs.shape[0] == n
        return msgs, None, msgs.shape[1], msgs.shape[1]

    def verify_many(self, sigs, pks, msgs, offsets=None, check_torsion=True, threads=0, pk_inf=None,
                    sig_flag_byte=False):
        """sig_flag_byte: verify_batch's semantics for byte 48 of the signature (src/batch.rs:104)"""
        sigs = np.ascontiguousarray(sigs, dtype=np.uint8).reshape(-1, 81)
        pks = np.ascontiguousarray(pks, dtype=np.uint8).reshape(-1, 96)
        n = sigs.shape[0]
        msgs, off, stride, mlen = self._msgs(msgs, offsets, n)
        inf = np.ascontiguousarray(pk_inf, dtype=np.uint8) if pk_inf is not None else None
        st = np.zeros(n, np.uint8)
        self.lib.so_verify_many(_ptr(sigs, _u8p), _ptr(pks, _u8p), _ptr(inf, _u8p), _ptr(msgs, _u8p),
                                _ptr(off, _u64p), C.c_size_t(stride), C.c_size_t(mlen), C.c_size_t(n),
                                int(bool(check_torsion)) | (8 if sig_flag_byte else 0), int(threads), _ptr(st, _u8p))
        return st

    def verify_many_fast(self, sigs, pks, msgs, offsets=None, check_torsion=True, threads=0, pk_inf=None,
                         sig_flag_byte=False):
        """verify_many through the TIMING path (windowed / lazy-reduction algorithms): same statuses"""
        sigs = np.ascontiguousarray(sigs, dtype=np.uint8).reshape(-1, 81)
        pks = np.ascontiguousarray(pks, dtype=np.uint8).reshape(-1, 96)
        n = sigs.shape[0]
        msgs, off, stride, mlen = self._msgs(msgs, offsets, n)
        inf = np.ascontiguousarray(pk_inf, dtype=np.uint8) if pk_inf is not None else None
        st = np.zeros(n, np.uint8)
        self.lib.so_verify_many_fast(_ptr(sigs, _u8p), _ptr(pks, _u8p), _ptr(inf, _u8p), _ptr(msgs, _u8p),
                                     _ptr(off, _u64p), C.c_size_t(stride), C.c_size_t(mlen), C.c_size_t(n),
                                     int(bool(check_torsion)) | (8 if sig_flag_byte else 0), int(threads), _ptr(st, _u8p))
        return st

    def verify_batch_msm_fast(self, sigs, pks, msgs, coeffs, offsets=None, threads=0, pk_inf=None):
        """verify_batch_msm through the TIMING path (bucket MSM): same verdict"""
        sigs = np.ascontiguousarray(sigs, dtype=np.uint8).reshape(-1, 81)
        pks = np.ascontiguousarray(pks, dtype=np.uint8).reshape(-1, 96)
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint8).reshape(-1, 32)
        n = sigs.shape[0]
        msgs, off, stride, mlen = self._msgs(msgs, offsets, n)
        inf = np.ascontiguousarray(pk_inf, dtype=np.uint8) if pk_inf is not None else None
        return int(self.lib.so_verify_batch_msm_fast(_ptr(sigs, _u8p), _ptr(pks, _u8p), _ptr(inf, _u8p), _ptr(msgs, _u8p),
                                                     _ptr(off, _u64p), C.c_size_t(stride), C.c_size_t(mlen),
                                                     C.c_size_t(n), _ptr(coeffs, _u8p), int(threads)))

    def keygen_sign_many(self, sks, nonces, msgs, offsets=None, threads=0):
        sks = np.ascontiguousarray(sks, dtype=np.uint8).reshape(-1, 32)
        nonces = np.ascontiguousarray(nonces, dtype=np.uint8).reshape(-1, 32)
        n = sks.shape[0]
        msgs, off, stride, mlen = self._msgs(msgs, offsets, n)
        pks = np.zeros((n, 96), np.uint8)
        sigs = np.zeros((n, 81), np.uint8)
        self.lib.so_keygen_sign_many(_ptr(sks, _u8p), _ptr(nonces, _u8p), _ptr(msgs, _u8p),
                                     _ptr(off, _u64p), C.c_size_t(stride), C.c_size_t(mlen),
                                     C.c_size_t(n), int(threads), _ptr(pks, _u8p), _ptr(sigs, _u8p))
        return pks, sigs

    def verify_batch_msm(self, sigs, pks, msgs, coeffs, offsets=None, threads=0, pk_inf=None):
        sigs = np.ascontiguousarray(sigs, dtype=np.uint8).reshape(-1, 81)
        pks = np.ascontiguousarray(pks, dtype=np.uint8).reshape(-1, 96)
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint8).reshape(-1, 32)
        n = sigs.shape[0]
        msgs, off, stride, mlen = self._msgs(msgs, offsets, n)
        inf = np.ascontiguousarray(pk_inf, dtype=np.uint8) if pk_inf is not None else None
        return int(self.lib.so_verify_batch_msm(_ptr(sigs, _u8p), _ptr(pks, _u8p), _ptr(inf, _u8p), _ptr(msgs, _u8p),
                                                _ptr(off, _u64p), C.c_size_t(stride), C.c_size_t(mlen),
                                                C.c_size_t(n), _ptr(coeffs, _u8p), int(threads)))

    def decompress(self, c49):
        """-> (pk96 bytes, is_identity) or None when decompression fails."""
        c = np.frombuffer(bytes(c49), np.uint8).copy()
        pk = np.zeros(96, np.uint8)
        inf = C.c_int(0)
        ok = self.lib.so_decompress(_ptr(c, _u8p), _ptr(pk, _u8p), C.byref(inf))
        return (pk.tobytes(), bool(inf.value)) if ok else None

    def compress(self, pk96, pk_inf=False):
        pk = np.frombuffer(bytes(pk96), np.uint8).copy()
        c = np.zeros(49, np.uint8)
        self.lib.so_compress(_ptr(pk, _u8p), int(pk_inf), _ptr(c, _u8p))
        return c.tobytes()

    def hw_threads(self):
        return int(self.lib.so_hw_threads())
