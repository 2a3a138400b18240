"""CPU unit tests of the device arithmetic headers (host-compiled twin, tests/csrc/host_arith.cpp)
against the oracle.  Catches limb-logic errors without a GPU; the product never runs this code
on the host."""
import ctypes as C
import os
import random
import shutil
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "host_arith.cpp")
LIB = os.path.join(HERE, "csrc", "libhost_arith.so")
P = 2**64 - 2**32 + 1
Q = 0x7AF2599B3B3F22D0563FBF0F990A37B5327AA72330157722D443623EAED4ACCF
EDGE = [0, 1, P - 1, P, P + 1, 2**32 - 1, 2**32, P - 2**32, 2**63, 2**64 - 1, 2**64 - 2**32]
u64p = C.POINTER(C.c_uint64)


@pytest.fixture(scope="module")
def ha():
    if os.environ.get("HOST_ARITH_LIB"):      # tests/test_sanitizers.py: the ASan + UBSan build of the same source
        lib = C.CDLL(os.environ["HOST_ARITH_LIB"])
        return _bind(lib)
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    csrc = os.path.join(os.path.dirname(HERE), "schnorr-sig_amd", "csrc")
    deps = [SRC] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hpp")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.check_call(["hipcc", "--cuda-host-only", "-x", "hip", "-O2", "-shared", "-fPIC", SRC, "-o", LIB])
    return _bind(C.CDLL(LIB))


def _bind(lib):
    for f in ("ha_fp_mul", "ha_fp_add", "ha_fp_sub"):
        getattr(lib, f).restype = C.c_uint64
        getattr(lib, f).argtypes = [C.c_uint64, C.c_uint64]
    lib.ha_fp_sqr3.restype = C.c_uint64
    lib.ha_fp_sqr3.argtypes = [C.c_uint64]
    lib.ha_fp_inv.restype = C.c_uint64
    lib.ha_fp_inv.argtypes = [C.c_uint64]
    lib.ha_inv_sbox.restype = C.c_uint64
    lib.ha_inv_sbox.argtypes = [C.c_uint64]
    lib.ha_inv_sbox2.restype = C.c_uint64
    lib.ha_inv_sbox2.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.ha_fp_mul_small.restype = C.c_uint64
    lib.ha_fp_mul_small.argtypes = [C.c_uint64, C.c_uint32]
    return lib


def arr(v):
    return np.array([int(x) for x in v], dtype=np.uint64)


def p_(a):
    return a.ctypes.data_as(u64p)


def test_fp_ops_loose_inputs(ha):
    rnd = random.Random(1)
    vals = EDGE + [2**33 - 1, 2**33, 2**31, (2**32 - 1) << 32, 0x1ffffffff] + [rnd.randrange(2**64) for _ in range(200)]
    for a in vals:
        for b in EDGE + [rnd.randrange(2**64) for _ in range(8)]:
            assert ha.ha_fp_mul(a, b) == a * b % P
            assert ha.ha_fp_add(a, b) == (a + b) % P
            assert ha.ha_fp_sub(a, b) == (a - b) % P
        assert ha.ha_fp_mul_small(a, 7) == a * 7 % P
        assert ha.ha_fp_sqr3(a) == a * a % P
        assert ha.ha_fp_mul_small(a, 0xFFFFFFFF) == a * 0xFFFFFFFF % P
        if a % P:
            assert ha.ha_fp_inv(a) == pow(a, P - 2, P)
        assert ha.ha_inv_sbox(a) == pow(a, 10540996611094048183, P)
        ob = C.c_uint64(0)
        assert ha.ha_inv_sbox2(a, a ^ 0x5555, C.byref(ob)) == pow(a, 10540996611094048183, P)
        assert ob.value == pow(a ^ 0x5555, 10540996611094048183, P)


def test_fp6_ops(ha, oracle):
    rnd = random.Random(2)
    rows = [[P - 1] * 6, [2**64 - 1] * 6, [0] * 6, [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 2**64 - 1]]
    rows += [[rnd.randrange(2**64) for _ in range(6)] for _ in range(200)]
    for a in rows:
        b = rows[rnd.randrange(len(rows))]
        ac, bc = [x % P for x in a], [x % P for x in b]
        o = np.zeros(6, np.uint64)
        ha.ha_f6_mul(p_(arr(a)), p_(arr(b)), p_(o))
        assert (o == oracle.fp6_mul(ac, bc)).all()
        ha.ha_f6_sqr(p_(arr(a)), p_(o))
        assert (o == oracle.fp6_sqr(ac)).all()
        if any(ac):
            ha.ha_f6_inv(p_(arr(a)), p_(o))
            assert (o == oracle.fp6_inv(ac)).all()


def _pt(p):
    if p is None:
        return np.zeros(12, np.uint64), 1
    return arr(list(p[0]) + list(p[1])), 0


def _unpt(o, inf):
    return None if inf else (tuple(int(v) for v in o[:6]), tuple(int(v) for v in o[6:]))


def test_point_add_all_branches(ha):
    import pymodel as m
    g = m.default_params().generator()
    f = m.FIXTURE_SMALL_ORDER_PK
    p2, p3 = m.pt_mul(2, g), m.pt_mul(3, g)
    cases = [(g, p2), (g, g), (g, m.pt_neg(g)), (None, g), (g, None), (p3, p2), (f, f), (f, m.pt_neg(f)), (f, g)]
    for general in (0, 1):
        for a, b in cases:
            av, ai = _pt(a)
            bv, bi = _pt(b)
            o = np.zeros(12, np.uint64)
            inf = ha.ha_point_add(p_(av), ai, p_(bv), bi, general, p_(o))
            assert _unpt(o, inf) == m.pt_add(a, b)


def test_scalar_mul_table_path(ha, oracle):
    import pymodel as m
    rnd = random.Random(3)
    g = m.default_params().generator()
    f = m.FIXTURE_SMALL_ORDER_PK
    ks = [0, 1, 2, 7, 8, 9, 15, 16, 17, Q - 1, Q, Q + 1, 2**255 - 1, int("7" + "8" * 63, 16),
          int("7" * 64, 16), int("f" * 63, 16)] + [rnd.randrange(2**255) for _ in range(12)]
    tab = np.zeros(ha.ha_ptab_words(), np.uint64)
    small = [m.SMALL_ORDER_POINTS[o] for o in (2, 5, 10)]
    for o, p in m.SMALL_ORDER_POINTS.items():
        assert m.on_curve(p) and m.pt_mul(o, p) is None
    for p in (g, f, None) + tuple(small):
        for k in (ks if p not in small else list(range(0, 23)) + ks[9:22]):
            pv, pi = _pt(p)
            o = np.zeros(12, np.uint64)
            inf = ha.ha_mul_ptab(p_(arr([(k >> (64 * i)) & (2**64 - 1) for i in range(4)])), p_(pv), pi, p_(tab), p_(o))
            assert _unpt(o, inf) == oracle.point_mul(k, p), (k, p is g)


def test_hash_field_and_hash_message(ha, oracle):
    import schnorr_sig_amd  # noqa: F401  (only for the blob path)
    from oracle import PARAMS_BLOB
    blob = np.frombuffer(open(PARAMS_BLOB, "rb").read(), dtype=np.uint8).copy()
    rnd = random.Random(4)
    for n in (0, 1, 7, 8, 9, 16, 17, 25, 40):
        felts = arr([rnd.randrange(P) for _ in range(n)]) if n else np.zeros(1, np.uint64)
        for flags in (0, 1):     # generic and small-entry MDS paths
            d = np.zeros(4, np.uint64)
            ha.ha_hash_field(blob.ctypes.data_as(C.c_void_p), flags, p_(felts), n, p_(d))
            assert (d == oracle.hash_field(felts[:n])).all(), (n, flags)
    sig = np.frombuffer(bytes(rnd.randrange(256) for _ in range(81)), np.uint8).copy()
    pk = np.frombuffer(bytes(rnd.randrange(256) for _ in range(96)), np.uint8).copy()
    for i in range(0, 48, 8):
        sig[i + 7] = 0x7F
        pk[i + 7] = 0x7F
        pk[48 + i + 7] = 0x7F
    for L in (0, 1, 6, 7, 8, 13, 14, 24, 48, 80, 160):
        msg = np.frombuffer(bytes(rnd.randrange(256) for _ in range(L)) + b"\0", np.uint8).copy()
        d = np.zeros(4, np.uint64)
        ha.ha_hash_message(blob.ctypes.data_as(C.c_void_p), sig.ctypes.data_as(C.c_void_p),
                           pk.ctypes.data_as(C.c_void_p), msg.ctypes.data_as(C.c_void_p), L, p_(d))
        assert d.tobytes() == oracle.hash_message(sig[:48].tobytes(), pk.tobytes(), msg[:L].tobytes()), L


def test_scalar_arith(ha):
    rnd = random.Random(5)
    for _ in range(50):
        r, sk, h = (rnd.randrange(Q) for _ in range(3))
        e = np.zeros(4, np.uint64)
        lim = lambda v: arr([(v >> (64 * i)) & (2**64 - 1) for i in range(4)])
        ha.ha_sc_mul_sub(p_(lim(r)), p_(lim(sk)), p_(lim(h)), p_(e))
        assert int.from_bytes(e.tobytes(), "little") == (r - sk * h) % Q
    # sc_mul_mod: schoolbook + Barrett (mu = floor(2^510 / q)); edge operands stress the <= 2 corrections
    edge = [0, 1, 2, Q - 1, Q - 2, (Q + 1) // 2, 2**254, 2**254 - 1, 2**128, 2**128 - 1, 2**64, 2**192 + 1]
    pairs = [(a, b) for a in edge for b in edge] + [(rnd.randrange(Q), rnd.randrange(Q)) for _ in range(3000)]
    for a, b in pairs:
        o = np.zeros(4, np.uint64)
        ha.ha_sc_mul(p_(lim(a)), p_(lim(b)), p_(o))
        assert int.from_bytes(o.tobytes(), "little") == a * b % Q, (hex(a), hex(b))
    for v in (0, Q - 1, Q, Q + 1, 2 * Q, 2 * Q + 5, 2**256 - 1):
        o = np.zeros(4, np.uint64)
        ha.ha_sc_reduce(p_(arr([(v >> (64 * i)) & (2**64 - 1) for i in range(4)])), p_(o))
        assert int.from_bytes(o.tobytes(), "little") == v % Q


def test_fp6_sqrt_and_decompress(ha, oracle):
    """Tower-descent square root (two Fp3 Tonelli-Shanks runs) vs the oracle's generic Fp6
    Tonelli-Shanks, and from_compressed incl. the reference's encoding fixtures."""
    import pymodel as m
    rnd = random.Random(6)
    cases = [[rnd.randrange(P) for _ in range(6)] for _ in range(60)]
    cases += [[5, 0, 0, 0, 0, 0], [0, 0, 3, 0, 0, 0], [7, 0, 0, 0, 0, 0], [0, 0, 1, 0, 0, 0], [3, 0, 4, 0, 9, 0],
              [0, 1, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0], [0, 5, 0, 7, 0, 9]]
    cases += [list(m.f6_sqr(tuple(c))) for c in cases[:30]]
    nsq = 0
    for a in cases:
        o = np.zeros(6, np.uint64)
        ok = ha.ha_f6_sqrt(p_(arr(a)), p_(o))
        ref = oracle.fp6_sqrt(a)
        assert bool(ok) == (ref is not None), a
        if ok:
            nsq += 1
            assert tuple(int(v) for v in oracle.fp6_sqr(o)) == tuple(a)
    assert nsq >= 40
    g = m.default_params().generator()
    pts = [m.pt_mul(k, g) for k in (1, 2, 3, 99, m.Q - 1)] + [m.FIXTURE_SMALL_ORDER_PK] + list(m.SMALL_ORDER_POINTS.values())
    enc = [m.pt_compress(p) for p in pts] + [m.pt_compress(None), bytes(49), b"\xff" * 49,
                                              m.pt_compress(pts[0])[:48] + b"\xff", bytes(48) + b"\xc0",
                                              bytes([1]) + bytes(47) + b"\x80", (2).to_bytes(8, "little") + bytes(41)]
    enc += [bytes(c ^ (0x40 if i == 48 else 0) for i, c in enumerate(e)) for e in enc[:4]]
    for e in enc:
        o = np.zeros(12, np.uint64)
        inf = C.c_int(0)
        buf = np.frombuffer(e, np.uint8).copy()
        st = ha.ha_decompress(buf.ctypes.data_as(C.c_void_p), p_(o), C.byref(inf))
        want = oracle.decompress(e)
        st_m, pt_m = m.pt_decompress(e)
        assert (st == 0) == (want is not None) == (st_m == "ok"), e.hex()
        if want is not None:
            assert bool(inf.value) == want[1]
            assert o.tobytes() == want[0]
