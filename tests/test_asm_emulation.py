"""CPU test of the generated S-box asm (schnorr-sig_amd/csrc/fp_chain_asm.inc): the instruction strings are
executed by a small gfx950 interpreter (one lane: v_mad_u64_u32, carry chains through SGPR pairs, 64-bit shifts,
loops) and compared with x^7 and x^(1/7) mod p -- the asm itself is otherwise only ever run on the GPU.
The interpreter also asserts that every multiply-add whose carry-out is discarded cannot overflow, and that the
software wait states between a VALU write of an SGPR pair and the VALU read of it are respected."""
import os
import random
import re

P = 2**64 - 2**32 + 1
M32, M64 = 0xFFFFFFFF, (1 << 64) - 1
INC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "schnorr-sig_amd", "csrc",
                   "fp_chain_asm.inc")


def run(txt, fn, x, y, extra=""):
    m = re.search(r"SSA_DEV void %s\(u64 &x, u64 &y%s\) \{.*?asm volatile\((.*?)\n        : \[x0\]" % (fn, extra), txt, re.S)
    lines = [ln.strip().strip('"').replace("\\n\\t", "") for ln in m.group(1).split("\n") if ln.strip()]
    v, sg = [0] * 256, {}
    env = {"%[x0]": x & M32, "%[x1]": x >> 32, "%[y0]": y & M32, "%[y1]": y >> 32, "%[n]": 5}

    def rv(o):
        o = o.strip()
        if o in env:
            return env[o]
        mm = re.match(r"v\[(\d+):(\d+)\]$", o)
        if mm:
            a = int(mm.group(1))
            return v[a] | (v[a + 1] << 32)
        mm = re.match(r"v(\d+)$", o)
        if mm:
            return v[int(mm.group(1))]
        mm = re.match(r"s\[(\d+):(\d+)\]$", o)
        if mm:
            return sg.get(int(mm.group(1)), 0)
        return M32 if o == "-1" else int(o)

    def wv(o, val):
        o = o.strip()
        if o in env:
            env[o] = val & M32
            return
        mm = re.match(r"v\[(\d+):(\d+)\]$", o)
        if mm:
            a = int(mm.group(1))
            v[a], v[a + 1] = val & M32, (val >> 32) & M32
            return
        v[int(re.match(r"v(\d+)$", o).group(1))] = val & M32

    labels = {ln[:-1]: i for i, ln in enumerate(lines) if ln.endswith(":")}
    pc, s26, scc, issued = 0, 0, 0, 0
    sg_written_at = {}

    def ws(o, val):
        k = int(re.match(r"s\[(\d+):(\d+)\]$", o.strip()).group(1))
        sg[k] = val
        sg_written_at[k] = issued

    def rs(o):
        k = int(re.match(r"s\[(\d+):(\d+)\]$", o.strip()).group(1))
        assert issued - sg_written_at.get(k, -10) >= 3, "SGPR pair %d read %d slots after its VALU write" % (k, issued - sg_written_at[k])
        return sg.get(k, 0)

    while pc < len(lines):
        ln = lines[pc]
        pc += 1
        if ln.endswith(":"):
            continue
        issued += 1
        if ln.startswith("s_nop"):
            continue
        op, rest = ln.split(None, 1)
        a = [t.strip() for t in rest.split(",")]
        if op == "v_mov_b32":
            wv(a[0], rv(a[1]))
        elif op == "s_mov_b32":
            s26 = rv(a[1])
        elif op == "s_sub_u32":
            s26 -= 1
        elif op == "s_cmp_lg_u32":
            scc = s26 != 0
        elif op == "s_cbranch_scc1":
            if scc:
                pc = labels[a[0]]
        elif op == "v_mad_u64_u32":
            r = rv(a[2]) * rv(a[3]) + rv(a[4])
            assert a[1] == "s[24:25]" and r <= M64, "a multiply-add with a discarded carry-out overflowed"
            wv(a[0], r)
        elif op == "v_lshrrev_b32":
            wv(a[0], rv(a[2]) >> int(a[1]))
        elif op == "v_and_b32":
            wv(a[0], rv(a[1]) & rv(a[2]))
        elif op == "v_lshrrev_b64":
            wv(a[0], rv(a[2]) >> int(a[1]))
        elif op == "v_lshl_or_b32":
            wv(a[0], ((rv(a[1]) << int(a[2])) & M32) | rv(a[3]))
        elif op == "v_sub_co_u32":
            d = rv(a[2]) - rv(a[3])
            ws(a[1], 1 if d < 0 else 0)
            wv(a[0], d & M32)
        elif op == "v_subbrev_co_u32":
            d = rv(a[3]) - rv(a[2]) - rs(a[4])
            ws(a[1], 1 if d < 0 else 0)
            wv(a[0], d & M32)
        elif op == "v_cndmask_b32":
            wv(a[0], rv(a[2]) if rs(a[3]) else rv(a[1]))
        elif op == "v_lshl_add_u64":
            wv(a[0], ((rv(a[1]) << int(a[2])) + rv(a[3])) & M64)
        elif op == "v_add_co_u32":
            t = rv(a[2]) + rv(a[3])
            ws(a[1], t >> 32)
            wv(a[0], t & M32)
        elif op == "v_addc_co_u32":
            t = rv(a[2]) + rv(a[3]) + rs(a[4])
            ws(a[1], t >> 32)
            wv(a[0], t & M32)
        else:
            raise AssertionError("unknown instruction: " + ln)
    return env["%[x0]"] | (env["%[x1]"] << 32), env["%[y0]"] | (env["%[y1]"] << 32)


def test_generated_sbox_asm_on_the_cpu():
    txt = open(INC).read()
    rnd = random.Random(5)
    e_inv = 10540996611094048183          # 7^-1 mod (p - 1)
    vals = [0, 1, P - 1, P, P + 1, 2**64 - 1, 2**32, 2**32 - 1, 2**63, 2**64 - 2**32] + [rnd.randrange(2**64) for _ in range(30)]
    for i, a in enumerate(vals):
        b = vals[(i * 7 + 3) % len(vals)]
        rx, ry = run(txt, "inv_sbox2_asm", a, b)
        assert rx % P == pow(a, e_inv, P) and ry % P == pow(b, e_inv, P), (hex(a), hex(b))
        rx, ry = run(txt, "sbox2_asm", a, b)
        assert rx % P == pow(a, 7, P) and ry % P == pow(b, 7, P), (hex(a), hex(b))
        rx, ry = run(txt, "fp_sqr2_n_asm", a, b, extra=", int n")     # n = 5 in the interpreter
        assert rx % P == pow(a, 32, P) and ry % P == pow(b, 32, P)


def test_generated_file_is_up_to_date():
    """fp_chain_asm.inc is what tools/gen_fp_chain_asm.py generates (no hand edits)"""
    import importlib.util
    import io
    from contextlib import redirect_stdout
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    before = open(INC).read()
    spec = importlib.util.spec_from_file_location("gen_fp_chain_asm", os.path.join(root, "tools", "gen_fp_chain_asm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with redirect_stdout(io.StringIO()):
        mod.main()
    assert open(INC).read() == before
