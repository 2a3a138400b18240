"""GPU tests added in round 5 (all through the C ABI):
 * the comb for G -- the reference's const BASEPOINT_TABLE (src/signature.rs:20,116, src/batch.rs:98-100) -- is a
   property of the CONTEXT: window width chosen from an HBM budget, forced by the caller or the environment, and a
   failed allocation falls back to the next smaller table; every width gives the same bytes and verdicts;
 * host-buffer entry points in bounded device memory (src/batch.rs:31-36 takes slices of any length): staging sized
   for one slice, slices alternating between the context and its twin;
 * PublicKey::from(&PrivateKey) (src/public.rs:26-32) as one constant-time base multiplication (ssa_pubkey_many)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Q = 0x7AF2599B3B3F22D0563FBF0F990A37B5327AA72330157722D443623EAED4ACCF
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_round4 import _sliced_engine, _spoil, honest, make_scalars  # noqa: E402


def _fixture_key():
    """the reference's non-subgroup point (src/signature.rs:387-404) as 96 affine bytes"""
    with open(os.path.join(ROOT, "tests", "golden", "vectors.json")) as fh:
        f = json.load(fh)["fixture_small_order_pk"]
    return np.frombuffer(b"".join(int(x).to_bytes(8, "little") for x in (f["x"] + f["y"])), dtype=np.uint8)


# ---------------------------------------------------------------- comb geometry
@pytest.mark.parametrize("bits", [16, 20, 22])
def test_comb_geometry_is_a_property_of_the_context(engine, oracle, bits):
    """a context forced to `bits`-bit windows: same status vector as the default context and as the oracle on a
    config-5 slice (both kernel families, subgroup check and flag byte on and off), same signature bytes from both
    signers, same MSM-form verdicts, same keyed-context results"""
    import schnorr_sig_amd as ssa
    eng = ssa.Engine(0, gtab_bits=bits)
    try:
        info = eng.info()
        assert info["gtab_bits"] == bits and info["gtab_windows"] == (255 + bits) // bits
        assert info["gtab_bytes"] == info["gtab_windows"] * (1 << bits) * 96
        d = engine.info()
        assert d["gtab_bits"] in (16, 20, 22, 24) and d["gtab_bytes"] <= max(d["hbm_budget_bytes"], 16 * 65536 * 96)
        rng = np.random.default_rng(5000 + bits)
        n = 12000
        sigs, pks, msgs = honest(engine, rng, n)
        sigs, pks, msgs, inf, bad = _spoil(rng, sigs, pks, msgs, 120)        # 1 %: e / message / key / R
        pks[bad[5]] = _fixture_key()                                          # InvalidPublicKey with the subgroup check
        samp = np.unique(np.concatenate([bad, np.arange(0, n, 61)]))
        for torsion in (False, True):
            for fb in (False, True):
                ref, nf_ref = engine.verify_many(sigs, pks, msgs, check_torsion=torsion, pk_inf=inf, mode="lane",
                                                 sig_flag_byte=fb)
                got, nf = eng.verify_many(sigs, pks, msgs, check_torsion=torsion, pk_inf=inf, mode="lane", sig_flag_byte=fb)
                assert nf == nf_ref and (got == ref).all()
                exp = oracle.verify_many(sigs[samp], pks[samp], msgs[samp], check_torsion=torsion, pk_inf=inf[samp],
                                         sig_flag_byte=fb)
                assert (got[samp] == exp).all()
            sub = slice(0, 600)
            got_c, _ = eng.verify_many(sigs[sub], pks[sub], msgs[sub], check_torsion=torsion, pk_inf=inf[sub], mode="coop")
            ref_c, _ = engine.verify_many(sigs[sub], pks[sub], msgs[sub], check_torsion=torsion, pk_inf=inf[sub], mode="lane")
            assert (got_c == ref_c).all()
        # the signers walk the comb (throughput) or a table built from it (constant-time): same bytes as the default
        # context and the oracle
        sks, nonces = make_scalars(rng, 300), make_scalars(rng, 300)
        m = rng.integers(0, 256, size=(300, 33), dtype=np.uint8)
        for ct in (False, True):
            p1, s1 = eng.keygen_sign_many(sks, nonces, m, constant_time=ct)
            p0, s0 = engine.keygen_sign_many(sks, nonces, m, constant_time=ct)
            assert (p1 == p0).all() and (s1 == s0).all()
        po, so = oracle.keygen_sign_many(sks[:24], nonces[:24], m[:24])
        assert (p1[:24] == po).all() and (s1[:24] == so).all()
        assert (eng.pubkey_many(sks) == p1).all()
        # MSM form ([sum s_i e_i]G comes from the comb): honest Ok, one bad signature rejected; small and bucket paths
        hs, hp, hm = honest(engine, rng, 5000)
        for cnt in (64, 5000):
            assert eng.verify_batch_msm(hs[:cnt], hp[:cnt], hm[:cnt]) == 0
            b2 = hs[:cnt].copy()
            b2[cnt // 2, 50] ^= 4
            assert eng.verify_batch_msm(b2, hp[:cnt], hm[:cnt]) == 2
        # keyed context on this engine (ladder tables and per-key combs; [e]G from the context's comb)
        keys = hp[:8]
        idx = (np.arange(2000) % 8).astype(np.uint32)
        ksk, knn = make_scalars(rng, 8), make_scalars(rng, 2000)
        km = rng.integers(0, 256, size=(2000, 80), dtype=np.uint8)
        kp, _ = engine.keygen_sign_many(ksk, make_scalars(rng, 8), km[:8])
        _, ksig = engine.keygen_sign_many(ksk[idx], knn, km)
        ksig[77, 49] ^= 1
        for kind in ("ladder", "comb"):
            ks = eng.keyset_create(kp, kind=kind)
            st, nf = eng.verify_many_indexed(ks, idx, ksig, km)
            assert nf == 1 and st[77] == 2 and (np.delete(st, 77) == 0).all()
            ks.close()
        del keys
    finally:
        eng.close()


_CHILD = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import schnorr_sig_amd as ssa
out = {}
MB = 1 << 20
for name, kw in (("b200MB", dict(hbm_budget_bytes=200 * MB)), ("b2GB", dict(hbm_budget_bytes=2048 * MB)),
                 ("b6GB", dict(hbm_budget_bytes=6144 * MB)), ("forced22_tiny_budget", dict(gtab_bits=22, hbm_budget_bytes=MB))):
    e = ssa.Engine(0, **kw)
    out[name] = e.info()["gtab_bits"]
    e.close()
os.environ["SSA_GTAB_BITS"] = "20"
e = ssa.Engine(0)
out["env20"] = e.info()["gtab_bits"]
e.close()
del os.environ["SSA_GTAB_BITS"]
os.environ["SSA_HBM_BUDGET_MB"] = "300"
e = ssa.Engine(0)
out["env_budget_300MB"] = e.info()["gtab_bits"]
e.close()
del os.environ["SSA_HBM_BUDGET_MB"]
# allocation failure: leave ~9 GB free, then ask for the 17.7 GB table -- the context must come up on a smaller one
free, total = torch.cuda.mem_get_info(0)
hog = torch.empty(max(0, free - 9 * 1024 * MB), dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize()
e = ssa.Engine(0, gtab_bits=24)
out["after_failed_24"] = e.info()["gtab_bits"]
rng = np.random.default_rng(7)
sk = rng.integers(0, 256, size=(500, 32), dtype=np.uint8); sk[:, 31] &= 0x3f; sk[:, 0] |= 1
nn = rng.integers(0, 256, size=(500, 32), dtype=np.uint8); nn[:, 31] &= 0x3f; nn[:, 0] |= 1
m = rng.integers(0, 256, size=(500, 80), dtype=np.uint8)
pk, sg = e.keygen_sign_many(sk, nn, m)
sg[3, 49] ^= 1
st, nf = e.verify_many(sg, pk, m, check_torsion=True)
out["verify_after_fallback"] = [int(nf), int(st[3]), int((st != 0).sum())]
e.close()
del hog
print("RESULT " + json.dumps(out))
"""


def test_budget_picks_the_width_and_a_failed_allocation_falls_back():
    """ssa_ctx_create_ex: the widest table within the budget (tenth of the free HBM by default), a forced width ignores the
    budget, the environment overrides the default -- and a hipMalloc that fails (most of the HBM taken by the process)
    moves on to a smaller table instead of failing the context.  Fresh child process: the registry of this process may
    already share a 24-bit table."""
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    out = json.loads(line[7:])
    assert out["b200MB"] == 16 and out["b2GB"] == 20 and out["b6GB"] == 22
    assert out["forced22_tiny_budget"] == 22 and out["env20"] == 20 and out["env_budget_300MB"] == 16
    assert out["after_failed_24"] in (16, 20, 22)
    assert out["verify_after_fallback"] == [1, 2, 1]


# ---------------------------------------------------------------- bounded staging of the host forms
def test_host_forms_stage_one_slice_at_a_time(engine, oracle):
    """ssa_verify_many / ssa_verify_keyed_many on host buffers with the slice forced to 65 536 lanes at a ragged
    n = 200 001 (offset table): results equal the unsliced engine's and the oracle's sample, ONE status array and ONE
    counter, and the device memory the context (and its twin) reserved stays that of two slices"""
    rng = np.random.default_rng(5100)
    n = 200001
    lens = rng.integers(0, 40, size=n)
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    flat = rng.integers(0, 256, size=int(off[-1]) + 1, dtype=np.uint8)
    sks, nonces = make_scalars(rng, n), make_scalars(rng, n)
    pks, sigs = engine.keygen_sign_many(sks, nonces, flat, offsets=off)
    bad = rng.permutation(n)[:300]
    sigs[bad, 49] ^= 1
    eng2 = _sliced_engine()
    try:
        ref, nf_ref = engine.verify_many(sigs, pks, flat, offsets=off, check_torsion=False, mode="lane")
        got, nf = eng2.verify_many(sigs, pks, flat, offsets=off, check_torsion=False, mode="lane")
        assert nf == nf_ref == 300 and (got == ref).all() and (got[bad] == 2).all()
        samp = np.unique(np.concatenate([bad[:40], [0, 65535, 65536, 131071, 131072, 196608, n - 1]]))
        msgs_s = [bytes(flat[int(off[i]):int(off[i + 1])]) for i in samp]
        import schnorr_sig_amd as ssa
        fl, of = ssa.pack_messages(msgs_s)
        exp = oracle.verify_many(sigs[samp], pks[samp], fl, offsets=of, check_torsion=False)
        assert (got[samp] == exp).all()
        info = eng2.info()
        assert info["lane_slice"] == 65536
        per_slice = 65536 * (16 * 256 + 32 + 81 + 96 + 40 + 8 + 1 + 1)          # table + scalar + staged inputs + status
        assert info["workspace_bytes"] < 2 * 1.2 * per_slice + (64 << 20), info
        assert info["workspace_bytes"] < n * 16 * 256, "staging must not grow with n"
        # the AND form and the keyed wire form ride the same slices
        assert eng2.verify_batch_status(sigs, pks, flat, offsets=off) == 2
        m2 = 150001
        comp, stc = engine.compress_many(pks[:m2])
        assert (stc == 0).all()
        keyed = np.concatenate([comp, sigs[:m2]], axis=1)
        rk, nk_ref = engine.verify_keyed_many(keyed, flat, offsets=off[:m2 + 1], check_torsion=True)
        gk, nk = eng2.verify_keyed_many(keyed, flat, offsets=off[:m2 + 1], check_torsion=True)
        assert nk == nk_ref == int((bad < m2).sum()) and (gk == rk).all()
    finally:
        eng2.close()


def test_slices_on_two_streams_equal_one_stream(engine):
    """device-pointer calls of more than one slice alternate their slices between the context's stream and its twin's:
    same statuses and count as with SSA_TWO_STREAMS=0, results valid after one ssa_ctx_sync, and the call stays
    ordered behind work queued on the context's stream before it"""
    import torch
    import schnorr_sig_amd as ssa
    rng = np.random.default_rng(5200)
    n = 5 * 65536 + 77
    sigs, pks, msgs = honest(engine, rng, n)
    sigs, pks, msgs, inf, bad = _spoil(rng, sigs, pks, msgs, 200)
    dev = torch.device("cuda", 0)
    ds, dp, dm, di = (torch.from_numpy(a).to(dev) for a in (sigs, pks, msgs, inf))
    ref, nf_ref = engine.verify_many(sigs, pks, msgs, check_torsion=False, pk_inf=inf, mode="lane", sig_flag_byte=True)
    os.environ["SSA_TWO_STREAMS"] = "0"
    try:
        one = _sliced_engine()
    finally:
        del os.environ["SSA_TWO_STREAMS"]
    two = _sliced_engine()
    try:
        assert two.info()["two_streams"] and not one.info()["two_streams"]
        for eng in (one, two):
            for rep in range(2):                 # second call: the twin exists, its buffers are reused
                dst = torch.full((n,), 255, dtype=torch.uint8, device=dev)
                dnf = torch.zeros(1, dtype=torch.int64, device=dev)
                eng.verify_many_device(ds.data_ptr(), dp.data_ptr(), dm.data_ptr(), n, 80, dst.data_ptr(), dnf.data_ptr(),
                                       d_pk_inf=di.data_ptr(), check_torsion=False, mode="lane", sig_flag_byte=True)
                eng.sync()
                assert int(dnf.item()) == nf_ref == 200 and (dst.cpu().numpy() == ref).all()
        assert two.info()["workspace_bytes"] > one.info()["workspace_bytes"]       # the second set of workspaces exists
    finally:
        one.close()
        two.close()


# ---------------------------------------------------------------- PublicKey::from(&PrivateKey)
def test_pubkey_many_is_one_base_multiplication(engine, oracle):
    """ssa_pubkey_many = PublicKey::from(&PrivateKey) (src/public.rs:26-32): the keys of the signer and of the oracle,
    zero and non-canonical scalars refused by the host form, reduced by the device form; the mirrors use it"""
    import torch
    import schnorr_sig_amd as ssa
    rng = np.random.default_rng(5300)
    n = 3000
    sks = make_scalars(rng, n)
    sks[0] = np.frombuffer((1).to_bytes(32, "little"), np.uint8)
    sks[1] = np.frombuffer((Q - 1).to_bytes(32, "little"), np.uint8)
    pks = engine.pubkey_many(sks)
    ref, _ = engine.keygen_sign_many(sks, make_scalars(rng, n), rng.integers(0, 256, size=(n, 8), dtype=np.uint8))
    assert (pks == ref).all()
    po, _ = oracle.keygen_sign_many(sks[:16], sks[:16], np.zeros((16, 1), np.uint8))
    assert (pks[:16] == po).all()
    for bad in (0, Q, Q + 5, (1 << 256) - 1):
        b = sks[:4].copy()
        b[2] = np.frombuffer(bad.to_bytes(32, "little"), np.uint8)
        with pytest.raises(RuntimeError):
            engine.pubkey_many(b)
    # device form: any 32 bytes, reduced mod q ([0]G is the identity: the (0, 0) record)
    dev = torch.device("cuda", 0)
    raw = sks[:8].copy()
    raw[3] = np.frombuffer((Q + 7).to_bytes(32, "little"), np.uint8)
    raw[4] = 0
    d_in = torch.from_numpy(raw).to(dev)
    d_out = torch.zeros((8, 96), dtype=torch.uint8, device=dev)
    engine.pubkey_many_device(d_in.data_ptr(), 8, d_out.data_ptr())
    engine.sync()
    out = d_out.cpu().numpy()
    seven = engine.pubkey_many(np.frombuffer((7).to_bytes(32, "little"), np.uint8))
    assert (out[3] == seven[0]).all() and (out[4] == 0).all() and (out[:3] == pks[:3]).all()
    # mirrors
    sk = ssa.PrivateKey(bytes(sks[9]))
    kp = ssa.KeyPair.from_private(sk, engine)
    assert kp.public_key.affine == bytes(pks[9]) and ssa.PublicKey.from_private(sk, engine).affine == bytes(pks[9])
    sig = kp.sign(b"round five", lambda k: bytes(rng.integers(0, 256, size=k, dtype=np.uint8)), engine)
    assert sig.verify(b"round five", kp.public_key, engine) is None


# ---------------------------------------------------------------- the end game of ssa_k_verify
def _tail_engine(pieces, waves=32, gens=1, uniform=False, reversed_grid=False):
    """an engine whose "generation" is `waves` waves, so that batches of thousands run the end game: the last generation of
    lanes in `pieces` pieces with the accumulator parked in between (read at ssa_ctx_create)"""
    import schnorr_sig_amd as ssa
    env = {"SSA_TAIL_WAVES": str(waves), "SSA_TAIL_PIECES": str(pieces), "SSA_TAIL_GENS": str(gens),
           "SSA_TAIL_UNIFORM": "1" if uniform else "0", "SSA_TAIL_REVERSED": "1" if reversed_grid else "0"}
    os.environ.update(env)
    try:
        return ssa.Engine(0)
    finally:
        for k in env:
            del os.environ[k]


@pytest.mark.parametrize("pieces,gens,uniform,reversed_grid",
                         [(2, 1, False, False), (3, 1, False, False), (5, 2, False, False), (8, 1, False, False),
                          (4, 1, True, False), (7, 3, True, False), (5, 1, False, True), (8, 2, True, True)])
def test_end_game_pieces_equal_whole_lanes(engine, oracle, pieces, gens, uniform, reversed_grid):
    """ssa_k_verify with its last generation(s) of lanes cut into pieces (accumulator and status parked between them, flags
    with release / acquire) gives the status vector of the plain launch and of the oracle: every corruption class, malformed
    inputs, identity keys and the non-subgroup fixture INSIDE the tail groups, subgroup check (two passes: a piece never
    spans them) and flag byte on and off, a ragged last group.  reversed_grid deals the roles from the END of the grid -- the
    closing pieces' workgroups start first, the worst order a dispatcher could choose: a wave then takes the unclaimed
    earlier pieces of its group along with its own and the late-comers leave (no wave waits for one that has not started)"""
    rng = np.random.default_rng(5400 + pieces)
    waves = 32
    n = (2 + gens) * waves * 64 + 1000 + 37                 # tail = the last gens * 2048 lanes (+ the ragged remainder)
    sigs, pks, msgs = honest(engine, rng, n)
    sigs, pks, msgs, inf, bad = _spoil(rng, sigs, pks, msgs, 160)
    tail0 = n - gens * waves * 64 - 500
    t = np.arange(tail0, n)
    rng.shuffle(t)
    sigs[t[0], 49] ^= 1                                       # inside the tail: e
    msgs[t[1], 3] ^= 8                                        # message
    pks[t[2]] = pks[t[3]]                                     # someone else's key
    pks[t[4]] = _fixture_key()                                # not in the prime subgroup
    sigs[t[5], :8] = 0xFF                                     # limb >= p: malformed signature
    pks[t[6], 90:96] = 0xFF                                   # malformed key
    inf[t[7]] = 1                                             # identity key
    sigs[t[8], 48] ^= 0x40                                    # the other root of R (flag-byte semantics only)
    sigs[t[9], 48] |= 2                                       # undecodable flag byte
    sigs[t[10], 49:] = np.frombuffer(Q.to_bytes(32, "little"), np.uint8)   # e = q
    eng = _tail_engine(pieces, waves, gens, uniform, reversed_grid)
    try:
        for torsion in (False, True):
            for fb in (False, True):
                ref, nf_ref = engine.verify_many(sigs, pks, msgs, check_torsion=torsion, pk_inf=inf, mode="lane", sig_flag_byte=fb)
                got, nf = eng.verify_many(sigs, pks, msgs, check_torsion=torsion, pk_inf=inf, mode="lane", sig_flag_byte=fb)
                assert nf == nf_ref and (got == ref).all(), (torsion, fb, np.nonzero(got != ref)[0][:10])
                samp = np.unique(np.concatenate([bad, t[:40], np.arange(0, n, 97), [n - 1]]))
                exp = oracle.verify_many(sigs[samp], pks[samp], msgs[samp], check_torsion=torsion, pk_inf=inf[samp],
                                         sig_flag_byte=fb)
                assert (got[samp] == exp).all()
            assert got[t[4]] == (1 if torsion else 2) and got[t[5]] == 3 and got[t[6]] == 3 and got[t[9]] == 3
        # the same call again (flags and parked state are reset per launch), and a batch too small for an end game
        got2, _ = eng.verify_many(sigs, pks, msgs, check_torsion=True, pk_inf=inf, mode="lane", sig_flag_byte=True)
        assert (got2 == got).all()
        small = slice(0, 3 * waves * 64 - 1)
        a, _ = eng.verify_many(sigs[small], pks[small], msgs[small], check_torsion=True, pk_inf=inf[small], mode="lane")
        b, _ = engine.verify_many(sigs[small], pks[small], msgs[small], check_torsion=True, pk_inf=inf[small], mode="lane")
        assert (a == b).all()
    finally:
        eng.close()


# ---------------------------------------------------------------- the statements that gather their own operands
def test_signer_and_verifier_are_deterministic_at_full_occupancy(engine):
    """2^19 keygen + sign twice from the same inputs: the same bytes, and every signature verifies under both kernel
    families' large-batch path.  The generated statements of jac_asm.inc gather their own table entries since round 5; the
    compiler does not wait for its scratch reloads into registers such a statement only clobbers, and before every statement
    started with its own wait the signer's second comb ran with reloads in flight: wrong R on some waves, different ones
    from run to run (profiles/r05/gather_ab.txt).  Only a full machine shows it: the reloads must be slow enough."""
    n = 1 << 19
    rng = np.random.default_rng(77)
    sks = rng.integers(1, 255, size=(n, 32), dtype=np.uint8)
    sks[:, 31] &= 0x3F
    nonces = rng.integers(1, 255, size=(n, 32), dtype=np.uint8)
    nonces[:, 31] &= 0x3F
    msgs = rng.integers(0, 256, size=(n, 80), dtype=np.uint8)
    pks, sigs = engine.keygen_sign_many(sks, nonces, msgs)
    pks2, sigs2 = engine.keygen_sign_many(sks, nonces, msgs)
    assert (pks == pks2).all() and (sigs == sigs2).all()
    for torsion in (False, True):
        st, nf = engine.verify_many(sigs, pks, msgs, check_torsion=torsion, mode="lane")
        assert nf == 0 and not st.any()
    st2, nf2 = engine.verify_many(sigs, pks, msgs, mode="lane", sig_flag_byte=True)
    assert nf2 == 0 and not st2.any()
