"""Static check of the constant-time signer (SSA_FLAG_SIGN_CT, schnorr-sig_amd/csrc/ssa_sign.hip): the reference signs
with the constant-time `&BASEPOINT_TABLE * r` and Scalar::from_bits (src/signature.rs:67,116,123), so the code that
touches sk, the nonce and e = r - sk h must have NO data-dependent control flow.

The secret-dependent work of ssa_k_sign_ct lives in out-of-line functions -- ct_load_scalar, ct_base_mul, ct_to_aff,
ct_response and the compiled Fp6 product / square they call.  This test compiles the unit to gfx950 assembly (hipcc
cross-compiles without a GPU) and asserts that in those bodies
  * no branch depends on EXEC or VCC (s_cbranch_execz/execnz/vccz/vccnz): lane values never steer the wave;
  * nothing moves a lane value to the scalar unit (v_readfirstlane; v_readlane only reloads the spilled return address),
    no EXEC mask is narrowed (s_and_saveexec & co.; EXEC is only widened to all lanes around the prologue / epilogue
    spills), so the scalar compares in front of the remaining branches can only see loop counters;
  * every conditional branch that is left (s_cbranch_scc*) follows an s_cmp of an SGPR with an immediate: the counters
    of the window loop, the table scan and the fixed exponent chains of the inversion;
  * every load reads the table or the scalar through an address that is no function of a loaded value's lane
    (table rows are indexed by the loop counters only: the selected entry is chosen with v_cndmask).
The dynamic counterpart runs on the GPU (tests/test_gpu_round4.py: byte-equality with the oracle signer; the PMC
instruction counts of two secret sets are recorded by tools/sign_ct_probe.py under profiles/r04/)."""
import hashlib
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "schnorr-sig_amd", "csrc")
CACHE = os.path.join(ROOT, "build", "sign_ct_static")
SECRET_FUNCS = ("ct_load_scalar", "ct_base_mul", "ct_to_aff", "ct_response", "f6_mul_flat", "f6_sqr_flat")


def _asm():
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    deps = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp", ".inc"))]
    deps.append(os.path.join(ROOT, "include", "schnorr_sig_amd.h"))
    h = hashlib.sha256()
    for p in deps:
        h.update(os.path.basename(p).encode() + b"\0" + open(p, "rb").read() + b"\0")
    os.makedirs(CACHE, exist_ok=True)
    out, stamp = os.path.join(CACHE, "ssa_sign.s"), os.path.join(CACHE, "ssa_sign.s.srchash")
    if not (os.path.exists(out) and os.path.exists(stamp) and open(stamp).read().strip() == h.hexdigest()):
        subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "--cuda-device-only", "-S", "-o", out,
                               os.path.join(CSRC, "ssa_sign.hip")], stderr=subprocess.DEVNULL)
        open(stamp, "w").write(h.hexdigest() + "\n")
    return open(out).read()


def _functions(text):
    """{demangled-ish short name: body text} for every function of namespace ssa"""
    out = {}
    for ch in re.split(r"^(?=_ZN3ssa\w+:)", text, flags=re.M):
        m = re.match(r"_ZN3ssa(\d+)(\w+):", ch)
        if m:
            out[m.group(2)[:int(m.group(1))]] = ch.split(".Lfunc_end")[0]
    return out


def test_secret_dependent_code_has_no_data_dependent_control_flow():
    fns = _functions(_asm())
    for name in SECRET_FUNCS:
        assert name in fns, "function %s not found out of line (inlined? the check below needs its own body)" % name
        body = fns[name]
        lines = [ln.strip() for ln in body.splitlines() if ln.strip() and not ln.strip().startswith((";", "."))]
        assert len(lines) > 20, name
        for bad in ("s_cbranch_execz", "s_cbranch_execnz", "s_cbranch_vccz", "s_cbranch_vccnz", "v_readfirstlane",
                    "s_and_saveexec", "s_andn2_saveexec", "s_xor_saveexec", "s_cbranch_cdbg", "v_cmpx"):
            hits = [ln for ln in lines if bad in ln]
            assert not hits, "%s: %s (%d occurrences), first: %s" % (name, bad, len(hits), hits[0])
        # v_readlane only as the reload of a spilled SGPR (the return address): from a VGPR that nothing but
        # v_writelane writes in this function, at a constant lane
        spill = {m for m in re.findall(r"v_writelane_b32 (v\d+),", body)}
        for ln in lines:
            if ln.startswith("v_readlane"):
                m = re.match(r"v_readlane_b32 s\d+, (v\d+), \d+$", ln)
                assert m and m.group(1) in spill, (name, ln)
                others = [x for x in lines if re.match(r"v_\w+ %s[, ]" % m.group(1), x) and not x.startswith("v_writelane")]
                assert not others, (name, ln, others[:2])
        # EXEC is only ever widened to all lanes around the prologue / epilogue spills of callee-saved VGPRs
        # (s_or_saveexec_b64 s[a:b], -1 ... s_mov_b64 exec, s[a:b]) -- no lane value is involved
        saved = set()
        for ln in lines:
            m = re.match(r"s_or_saveexec_b64 (s\[\d+:\d+\]), (.+)$", ln)
            if m:
                assert m.group(2) == "-1", (name, ln)
                saved.add(m.group(1))
            elif re.match(r"s_\w+ exec", ln):
                m = re.match(r"s_mov_b64 exec, (s\[\d+:\d+\])$", ln)
                assert m and m.group(1) in saved, (name, ln)
        # the conditional branches that remain: scalar compares of an SGPR with an immediate (loop counters)
        for i, ln in enumerate(lines):
            if ln.startswith("s_cbranch_scc"):
                prev = [x for x in lines[max(0, i - 400):i] if x.startswith(("s_cmp", "s_and", "s_or", "s_xor", "s_bitcmp"))]
                assert prev and re.match(r"s_cmpk?_(eq|lg|lt|gt|le|ge)_[ui]32 s\d+, (0x[0-9a-f]+|-?\d+)$", prev[-1]), (name, ln, prev[-3:])
    # the closure of the secret-dependent code is the checked set: every call out of a checked body goes to a checked body
    # (a compiler or an edit that outlines f6_inv, sc_mul_4x4, ... would otherwise escape the check while it stays green),
    # and every call is a direct one (s_swappc through a pc-relative symbol: one @rel32@lo per s_swappc)
    for name in SECRET_FUNCS:
        body = fns[name]
        pairs = {}                       # SGPR pair -> the function whose address it holds (set up pc-relative, maybe copied)
        for ln in (x.strip() for x in body.splitlines()):
            m = re.match(r"s_add_u32 s(\d+), s\1, _ZN3ssa(\d+)(\w+)@rel32@lo", ln)
            if m:
                pairs["s[%d:%d]" % (int(m.group(1)), int(m.group(1)) + 1)] = m.group(3)[:int(m.group(2))]
                continue
            m = re.match(r"s_mov_b64 (s\[\d+:\d+\]), (s\[\d+:\d+\])$", ln)
            if m and m.group(2) in pairs:
                pairs[m.group(1)] = pairs[m.group(2)]
                continue
            m = re.match(r"s_swappc_b64 s\[30:31\], (s\[\d+:\d+\])$", ln)
            if m:
                assert m.group(1) in pairs, "%s: call through %s, which holds no pc-relative symbol (indirect?)" % (name, m.group(1))
                assert pairs[m.group(1)] in SECRET_FUNCS, "%s calls %s, which is not in the checked set" % (name, pairs[m.group(1)])
            else:
                assert not ln.startswith("s_swappc"), (name, ln)
        assert not re.search(r"^\s*s_setpc_b64 (?!s\[30:31\])", body, flags=re.M), name       # only the return
    # the kernels themselves call them (they were not folded into the kernel bodies, where public-data branches live)
    kern = fns["ssa_k_sign_ct"]
    assert kern.count("s_swappc_b64") >= 6
    pub = fns["ssa_k_pubkey_ct"]         # PublicKey::from(&PrivateKey): load, one base multiplication, to affine
    called = {rest[:int(ln)] for _, ln, rest in re.findall(r"(_ZN3ssa(\d+)(\w+))@rel32@lo", pub)}
    assert {"ct_load_scalar", "ct_base_mul", "ct_to_aff"} <= called, called
    assert "ct_response" not in called and "hash_message_lane" not in called


def test_table_scan_reads_every_entry_and_selects():
    """ct_base_mul: the inner scan is a COUNTED loop (s_cmp of the entry counter with an immediate) whose body loads one 96-byte
    entry -- six 16-byte loads -- and selects its 24 words with v_cndmask; after the addition the 36 words of the
    accumulator are selected too (digit 0 keeps the old point).  No entry is picked by address."""
    body = _functions(_asm())["ct_base_mul"]
    lines = [ln.split(";")[0].strip() for ln in body.splitlines()]
    lines = [ln for ln in lines if ln]
    # the innermost loop: from its label to the backward branch
    loops = []
    for i, ln in enumerate(lines):
        m = re.match(r"s_cbranch_scc[01] (\.LBB\d+_\d+)$", ln)
        if m and (m.group(1) + ":") in lines[:i]:
            j = max(k for k in range(i) if lines[k] == m.group(1) + ":")
            loops.append(lines[j:i + 1])
    assert loops
    inner = min(loops, key=len)
    n_load = sum(1 for ln in inner if ln.startswith(("flat_load_dwordx4", "global_load_dwordx4")))
    n_sel = sum(1 for ln in inner if ln.startswith("v_cndmask_b32"))
    assert n_load == 6 and n_sel == 24, (n_load, n_sel)
    # (the compiler counts the scan in bytes: 14 entries x 96 B = 0x540)
    assert any(re.match(r"s_cmpk?_(eq|lg)_[ui]32 s\d+, (14|16|0x10|0x540)$", ln) for ln in inner), inner[-6:]
    assert not any(ln.startswith(("s_swappc", "v_readfirstlane", "v_readlane")) for ln in inner)
    assert sum(1 for ln in lines if ln.startswith("v_cndmask_b32")) >= 24 + 36
