"""CPU tests of the boundary: the C-ABI library loads without a GPU and exports every symbol
include/schnorr_sig_amd.h declares; host-side mirror logic; loud failure without a device."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import schnorr_sig_amd as ssa
    hdr = open(os.path.join(ROOT, "include", "schnorr_sig_amd.h")).read()
    declared = set(re.findall(r"\b(ssa_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = ctypes.CDLL(ssa.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "missing export: " + name
    assert declared == set(ssa.ABI_SYMBOLS)


def test_header_constants_match_reference_lengths():
    hdr = open(os.path.join(ROOT, "include", "schnorr_sig_amd.h")).read()
    vals = dict(re.findall(r"#define (SSA_[A-Z_]+) +\(?(-?\d+)u?\)?", hdr))
    assert int(vals["SSA_SIGNATURE_LENGTH"]) == 81 and int(vals["SSA_AFFINE_PK_LENGTH"]) == 96
    assert int(vals["SSA_SCALAR_LENGTH"]) == 32 and int(vals["SSA_PARAMS_LENGTH"]) == 2816
    assert (int(vals["SSA_OK"]), int(vals["SSA_INVALID_PUBLIC_KEY"]), int(vals["SSA_INVALID_SIGNATURE"]),
            int(vals["SSA_MALFORMED"])) == (0, 1, 2, 3)


def test_default_params_blob_is_the_committed_one():
    import schnorr_sig_amd as ssa
    blob = ssa.Engine.default_params()
    assert blob == open(os.path.join(ROOT, "schnorr-sig_amd", "params", "params_default.bin"), "rb").read()


def test_no_silent_cpu_fallback():
    """Without a GPU the product path must fail loudly, never compute on the host."""
    import torch
    import schnorr_sig_amd as ssa
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        ssa.Engine(0)
    with pytest.raises(RuntimeError):
        ssa.Signature(bytes(81)).verify(b"m", ssa.PublicKey(bytes(96)))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "schnorr-sig_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "schnorr_oracle" not in text and "pymodel" not in text, f
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f


def test_error_vocabulary_and_messages():
    import schnorr_sig_amd as ssa
    assert str(ssa.SignatureError(ssa.SignatureError.InvalidPublicKey)) == \
        "The public key is not an element of the prime subgroup."          # src/error.rs:24
    assert str(ssa.SignatureError(ssa.SignatureError.InvalidSignature)) == \
        "The signature is invalid or was incorrectly computed."            # src/error.rs:27
    assert repr(ssa.SignatureError(ssa.SignatureError.InvalidPublicKey)) == "Err(InvalidPublicKey)"


def test_verify_batch_length_mismatch_panics_like_the_reference():
    import schnorr_sig_amd as ssa
    with pytest.raises(ssa.MalformedInput, match="same number of signatures"):
        ssa.verify_batch([ssa.Signature(bytes(81))], [], [b""])
    with pytest.raises(ssa.MalformedInput, match="same number of messages"):
        ssa.verify_batch([ssa.Signature(bytes(81))], [ssa.PublicKey(bytes(96))], [])
    assert ssa.verify_batch([], [], []) is None      # empty batch is Ok (src/batch.rs)


def test_private_key_codecs_like_the_reference():
    """src/private.rs:118-185 (test_encoding, test_from_seed) on the mirror's host-side scalar codecs: known encodings,
    round trips, the invalid encodings the reference lists (zero, all ones, top byte 127), a seed whose upper half is zero"""
    import schnorr_sig_amd as ssa
    one = bytes([1] + [0] * 31)
    assert ssa.PrivateKey(one).to_bytes() == one                       # from_scalar(Scalar::one()).to_bytes()
    rng = lambda k: os.urandom(k)
    for _ in range(100):
        key = ssa.PrivateKey.new(rng)
        b = key.to_bytes()
        assert len(b) == ssa.PRIVATE_KEY_LENGTH and key == ssa.PrivateKey.from_bytes(b)
        assert key == ssa.PrivateKey.from_seed(b + bytes(32))         # seed[0..32] = the key's bytes (:164-176)
    assert ssa.PrivateKey.from_bytes(bytes(32)) is None
    assert ssa.PrivateKey.from_bytes(b"\xff" * 32) is None
    assert ssa.PrivateKey.from_bytes(ssa.PrivateKey.new(rng).to_bytes()[:31] + bytes([127])) is None   # :203-204
    assert ssa.PrivateKey.from_seed(bytes(64)) is None
    q = ssa.Q
    assert ssa.PrivateKey.from_bytes(q.to_bytes(32, "little")) is None
    assert ssa.PrivateKey.from_bytes((q - 1).to_bytes(32, "little")).to_bytes() == (q - 1).to_bytes(32, "little")
    wide = int.from_bytes(os.urandom(64), "little")
    assert int.from_bytes(ssa.PrivateKey.from_seed(wide.to_bytes(64, "little")).to_bytes(), "little") == wide % q


def test_pack_messages():
    import schnorr_sig_amd as ssa
    flat, off = ssa.pack_messages([b"ab", b"", b"cde"])
    assert list(off) == [0, 2, 2, 5] and bytes(flat[:5]) == b"abcde"


def test_shard_ranges_cover_batch():
    from schnorr_sig_amd.sharding import shard_range, batch_verdict
    for n in (0, 1, 7, 8, 1 << 20, (1 << 22) + 3):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert batch_verdict(0) == 0 and batch_verdict(3) == 2


def test_blob_from_upstream_round_trips_the_default_blob():
    """tools/blob_from_upstream.py: text constants -> blob.  Parsing the built-in blob and rebuilding it must give
    the same 2816 bytes (so the day upstream's constants are typed in, the path is already exercised); the
    plain-integer generator check accepts the default G and rejects the reference's non-subgroup fixture."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import blob_from_upstream as bfu
    import pymodel as m
    blob = open(os.path.join(ROOT, "schnorr-sig_amd", "params", "params_default.bin"), "rb").read()
    spec = bfu.parse_blob(blob)
    assert spec["rounds"] == 7 and spec["sponge"] == {"rate_offset": 0, "length_index": 11, "pad_one": False,
                                                       "digest_offset": 0}
    assert bfu.build_blob(spec) == blob
    as_text = {"constants": {**spec, "mds": [[hex(v) for v in row] for row in spec["mds"]],
                             "ark1": [[str(v) for v in row] for row in spec["ark1"]]}}
    assert bfu.build_blob(as_text) == blob
    circ = dict(spec)
    circ.pop("mds")
    circ["mds_circulant_first_row"] = spec["mds"][0]
    assert bfu.build_blob(circ) == blob
    assert bfu.check_generator(spec["generator"]["x"], spec["generator"]["y"]) is None
    f = m.FIXTURE_SMALL_ORDER_PK
    assert "outside the prime-order subgroup" in bfu.check_generator(list(f[0]), list(f[1]))
    assert "not on" in bfu.check_generator([1, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0])
    with pytest.raises(ValueError):
        bfu.build_blob({**spec, "rounds": 9})
    with pytest.raises(ValueError):
        bad = dict(spec)
        bad["generator"] = {"x": [2**64 - 1] + [0] * 5, "y": [0] * 6}
        bfu.build_blob(bad)


def test_library_never_registers_caller_memory():
    """The host-buffer entry points copy the caller's bytes into page-locked memory of the library's own; they must not
    change the mapping state of memory they do not own.  Rounds 3-5 registered the caller's arrays in place
    (hipHostRegister) for the duration of a call: with the arrays in the process heap a later plain copy faulted on the GPU
    side (tools/soak_large.py).  The library must not even import the symbols."""
    import subprocess
    import schnorr_sig_amd as ssa
    out = subprocess.run(["nm", "-D", "--undefined-only", ssa.LIB_PATH], capture_output=True, text=True, check=True).stdout
    imported = {ln.split()[-1].split("@")[0] for ln in out.splitlines() if ln.strip()}
    assert "hipHostMalloc" in imported and "hipMemcpyAsync" in imported          # (the listing is what it should be)
    assert not ({"hipHostRegister", "hipHostUnregister"} & imported)
