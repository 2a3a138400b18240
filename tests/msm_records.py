"""CPU model (test infrastructure, uses the oracle) of the MSM-form shard records of include/schnorr_sig_amd.h
(ssa_verify_batch_msm_partial / ssa_msm_combine): what one shard contributes to the reference's verify_batch
equation (src/batch.rs:84-129) and how k records combine into the verdict."""
import numpy as np

P = 2**64 - 2**32 + 1
Q = 0x7AF2599B3B3F22D0563FBF0F990A37B5327AA72330157722D443623EAED4ACCF
MAGIC = 0x5353415245430004      # SSA_MSM_RECORD_MAGIC (include/schnorr_sig_amd.h): word 23 of every record


def empty_record():
    """the record of an empty shard: the identity, 0, not malformed -- and the magic word"""
    r = np.zeros(24, np.uint64)
    r[23] = MAGIC
    return r


def record_is_wellformed(orc, rec):
    """what ssa_msm_combine checks before it trusts a record: the magic word, canonical limbs and scalar, flag in
    {0, 1}, and the point is the identity (Z = 0) or on the curve (Jacobian: Y^2 = X^3 + X Z^4 + (u + 395) Z^6)"""
    if int(rec[23]) != MAGIC or int(rec[22]) > 1:
        return False
    if any(int(v) >= P for v in rec[:18]) or record_lin(rec) >= Q:
        return False
    X, Y, Z = (tuple(int(v) for v in rec[6 * k:6 * k + 6]) for k in range(3))
    if not any(Z):
        return True
    z2 = _f6_mul(orc, Z, Z)
    z4 = _f6_mul(orc, z2, z2)
    z6 = _f6_mul(orc, z4, z2)
    rhs = [(a + b + c) % P for a, b, c in zip(_f6_mul(orc, _f6_mul(orc, X, X), X), _f6_mul(orc, X, z4),
                                               _f6_mul(orc, (395, 1, 0, 0, 0, 0), z6))]
    return tuple(rhs) == _f6_mul(orc, Y, Y)


def _neg(pt):
    return None if pt is None else (pt[0], tuple((P - v) % P for v in pt[1]))


def _aff(b96):
    v = np.frombuffer(bytes(b96), dtype="<u8")
    return tuple(int(x) for x in v[:6]), tuple(int(x) for x in v[6:])


def cpu_partial(orc, sigs, pks, msgs, coeffs, pk_inf=None):
    """(left point (affine tuple or None), sum s_i e_i mod q, malformed) of one shard, following src/batch.rs:
    R_i = from_compressed(sig.x) (:104), h_i = hash_message scalars (:64-73), left = sum s_i R_i - sum s_i h_i P_i"""
    left, lin = None, 0
    for i in range(len(sigs)):
        sig, pk, msg = bytes(sigs[i]), bytes(pks[i]), bytes(msgs[i])
        dec = orc.decompress(sig[:49])
        e = int.from_bytes(sig[49:], "little")
        if dec is None or e >= Q:
            return None, 0, True
        r = None if dec[1] else _aff(dec[0])
        s = int.from_bytes(bytes(coeffs[i]), "little") % Q
        h = int.from_bytes(orc.scalar_from_digest(orc.hash_message(sig[:48], pk, msg)), "little")
        left = orc.point_add(left, orc.point_mul(s, r))
        if not (pk_inf is not None and pk_inf[i]):
            left = orc.point_add(left, orc.point_mul(s * h % Q, _neg(_aff(pk))))
        lin = (lin + s * e) % Q
    return left, lin, False


def _f6_mul(orc, a, b):
    return tuple(int(v) for v in orc.fp6_mul(np.array(a, np.uint64), np.array(b, np.uint64)))


def record_point(orc, rec):
    """the Jacobian (X, Y, Z) of words 0..17 as an affine tuple (None for Z = 0)"""
    X, Y, Z = (tuple(int(v) for v in rec[6 * k:6 * k + 6]) for k in range(3))
    if not any(Z):
        return None
    zi = tuple(int(v) for v in orc.fp6_inv(np.array(Z, np.uint64)))
    zi2 = _f6_mul(orc, zi, zi)
    return _f6_mul(orc, X, zi2), _f6_mul(orc, Y, _f6_mul(orc, zi2, zi))


def record_lin(rec):
    return sum(int(rec[18 + k]) << (64 * k) for k in range(4))


def cpu_combine(orc, records):
    """ssa_msm_combine on the CPU: 3 if any record is flagged malformed, else compare the x coordinates of the sum of
    the shard points and of [sum lin]G (src/batch.rs:98-100, :123-129; the identity's x is taken as 0)"""
    records = np.asarray(records, dtype=np.uint64).reshape(-1, 24)
    if any(int(r[22]) for r in records) or not all(record_is_wellformed(orc, r) for r in records):
        return 3
    left, lin = None, 0
    for r in records:
        left = orc.point_add(left, record_point(orc, r))
        lin = (lin + record_lin(r)) % Q
    gen = _aff(orc.keygen((1).to_bytes(32, "little"))[0])
    right = orc.point_mul(lin, gen)
    lx = left[0] if left is not None else (0,) * 6
    rx = right[0] if right is not None else (0,) * 6
    return 0 if lx == rx else 2
