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


import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import asm_interp as ai

F6_INC = os.path.join(os.path.dirname(INC), "fp6_asm.inc")


def run(txt_blocks, fn, vals):
    """-> ([x', ...], flag): the values of the block's chains (PINNED to the chains' value registers) after the program, and
    this lane's bit of the mask operand %[st] -- non-zero means a reduction met its rare borrow and the values are to be
    recomputed by the caller.  The block size is the number of values (the *_n_asm programs: NCH, the *_1_asm programs: 1)"""
    import gen_fp_chain_asm as g
    lines, outs, _ = txt_blocks[fn]
    nch = len(vals)
    assert nch == (1 if fn.endswith("_1_asm") else g.NCH)
    lane = ai.Lane({}, dummy_pairs=(g.DUMMY,))
    for c, x in enumerate(vals):
        r = g.regs(c, nch)["X"]
        lane.v[r], lane.v[r + 1] = x & M32, x >> 32
    lane.run(lines)
    out = [lane.v[g.regs(c, nch)["X"]] | (lane.v[g.regs(c, nch)["X"] + 1] << 32) for c in range(nch)]
    return out, lane.s.get("%[st]", 0)


def test_generated_sbox_asm_on_the_cpu():
    import gen_fp_chain_asm as g
    txt = ai.extract_blocks(INC)
    assert set(txt) == {"inv_sbox_n_asm", "sbox_n_asm", "inv_sbox_1_asm", "sbox_1_asm"}
    rnd = random.Random(5)
    e_inv = 10540996611094048183          # 7^-1 mod (p - 1)
    vals = [0, 1, P - 1, P, P + 1, 2**64 - 1, 2**32, 2**32 - 1, 2**63, 2**64 - 2**32] + [rnd.randrange(2**64) for _ in range(30)]
    flagged = 0
    for i, a in enumerate(vals):
        ins = [vals[(i * (7 + 4 * c) + 3 * c) % len(vals)] for c in range(g.NCH)]
        for fn, want in (("inv_sbox_n_asm", lambda v: pow(v, e_inv, P)), ("sbox_n_asm", lambda v: pow(v, 7, P)),
                         ("inv_sbox_1_asm", lambda v: pow(v, e_inv, P)), ("sbox_1_asm", lambda v: pow(v, 7, P))):
            vals_in = ins[:1] if fn.endswith("_1_asm") else ins
            got, fl = run(txt, fn, vals_in)
            flagged += bool(fl)
            # a flagged lane is recomputed by the caller; an unflagged one must be right
            assert fl or [v % P for v in got] == [want(v) for v in vals_in], (fn, [hex(v) for v in vals_in])
    assert flagged <= 16         # only the hand-picked edge values can get there (2^32 * 2^32 = 2^64, ...)


def test_sbox_programs_execute_no_copy_and_no_loop():
    """round 4: the chains' `cp` steps are register renamings and the squaring runs are straight code; the statement has no
    move in or out (values pinned), so its VALU count is the arithmetic's: 63 squarings x 11 + 9 products x 13 per value"""
    import gen_fp_chain_asm as g
    txt = ai.extract_blocks(INC)
    for fn, n_sq, n_mul in (("inv_sbox_n_asm", 63, 9), ("sbox_n_asm", 2, 2)):
        lines = txt[fn][0]
        valu = [ln for ln in lines if ln.startswith("v_")]
        assert len(valu) == g.NCH * (11 * n_sq + 13 * n_mul + 1), (fn, len(valu))       # + one zero high word per chain
        assert not any(ln.startswith(("s_cbranch", "s_branch")) or ln.endswith(":") for ln in lines), fn
        movs = [ln for ln in valu if ln.startswith("v_mov_b32")]
        assert len(movs) == g.NCH * (1 + 4 * n_mul), (fn, len(movs))                    # only the product head's four word moves
        assert not any(ln.startswith("s_nop") for ln in lines), fn
    # the single-chain programs of the cooperative kernels: the same arithmetic, wait states padded with s_nop
    for fn, n_sq, n_mul in (("inv_sbox_1_asm", 63, 9), ("sbox_1_asm", 2, 2)):
        lines = txt[fn][0]
        assert len([ln for ln in lines if ln.startswith("v_")]) == 11 * n_sq + 13 * n_mul + 1, fn
        assert not any(ln.startswith(("s_cbranch", "s_branch")) or ln.endswith(":") for ln in lines), fn


def test_sbox_asm_reports_the_rare_borrow():
    """a = k 2^48: a^2 = k^2 2^96, so lo = 0, hi.lo = 0, hi.hi = k^2 -- the reduction X - h1 borrows with no carry to
    cancel it.  The block must flag such a lane (in whichever chain) instead of passing a wrong value on silently: the S-box
    programs OR it into their mask operand (the caller recomputes the lane's values from its inputs)."""
    import gen_fp_chain_asm as g
    txt = ai.extract_blocks(INC)
    for k in (1, 3, 0xffff):
        a = k << 48
        for fn in ("inv_sbox_n_asm", "sbox_n_asm"):
            for c in range(g.NCH):
                ins = [5 + j for j in range(g.NCH)]
                ins[c] = a
                assert run(txt, fn, ins)[1] == 1, (fn, k, c)
            assert run(txt, fn, [5 + j for j in range(g.NCH)])[1] == 0
        for fn in ("inv_sbox_1_asm", "sbox_1_asm"):
            assert run(txt, fn, [a])[1] == 1 and run(txt, fn, [5])[1] == 0, (fn, k)


def test_reduction_tail_is_exact_or_flagged():
    """the three-instruction tail of the generated reduction on (lo, hi) pairs at every boundary: with
    V = lo + EPS * hi.lo - hi.hi it must deliver V mod 2^64 + (carry ? EPS : 0) -- congruent to V, no wrap -- or raise
    the sticky flag, and the flag exactly when the true value is negative (no carry, X < hi.hi)"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gen_fp_chain_asm as g
    regs = g.regs(0)
    lines = g.schedule([g.reduce_tail(0, None, "%[st]")])
    rnd = random.Random(77)
    EPS = 2**32 - 1
    h_vals = [0, 1, 2, EPS, EPS - 1, 2**31] + [rnd.randrange(2**32) for _ in range(6)]
    n_flag = n_carry = 0
    for h0 in h_vals:
        for h1 in h_vals:
            base = (2**64 - EPS * h0) % 2**64
            los = {0, 1, h1, max(h1 - 1, 0), h1 + 1, 2**64 - 1, 2**64 - 2, base, (base - 1) % 2**64, (base + 1) % 2**64,
                   (base + h1) % 2**64, (base + h1 - 1) % 2**64, (base + h1 + 1) % 2**64, 2**32 - 1, 2**32}
            los |= {rnd.randrange(2**64) for _ in range(4)}
            for lo in los:
                lane = ai.Lane({}, dummy_pairs=(g.DUMMY,))
                lane.v[regs["T"]], lane.v[regs["T"] + 1] = lo & M32, lo >> 32
                lane.v[regs["H"]], lane.v[regs["H"] + 1] = h0, h1
                lane.s["%[st]"] = 0
                lane.run(lines)
                x = lane.v[regs["X"]] | (lane.v[regs["X"] + 1] << 32)
                full = lo + EPS * h0
                carry, X = full >> 64, full & M64
                assert carry <= 1
                negative = carry == 0 and X < h1
                assert lane.s["%[st]"] == int(negative), (hex(lo), hex(h0), hex(h1))
                n_flag += negative
                n_carry += carry
                if not negative:
                    assert x == X + carry * EPS - h1 and x % P == (lo + (h0 << 64) + (h1 << 96)) % P, (hex(lo), hex(h0), hex(h1))
    assert n_flag >= 20 and n_carry > 500


def test_fp6_reduction_group_is_exact_including_its_cold_path():
    """gen_f6_asm.reduce3 -- three accumulators reduced round-robin, the rare negative result repaired behind ONE branch
    per group -- on raw accumulator contents: random columns and counters, and contents built to land on every
    boundary (X + c EPS < top with and without the carry, th - c wrapping, all three chains at once, one chain only)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gen_f6_asm as g6
    accs = [g6.Acc(j) for j in range(3)]
    outs = [("v%d" % (200 + 2 * j), "v%d" % (201 + 2 * j)) for j in range(3)]
    hot, cold = g6.reduce3(accs, outs, "t")
    lines = hot + ["s_branch L_end_%="] + cold + ["L_end_%=:"]
    assert sum(ln.startswith("v_") for ln in hot) == 33          # 11 per coefficient
    rnd = random.Random(99)
    EPS = 2**32 - 1

    def value(c0, c1, c2, k0, k1, k2):
        return c0 + (c1 << 32) + ((c2 + k0) << 64) + (k1 << 96) + (k2 << 128)

    crafted = [
        (0, 0, 5 << 32, 0, 0, 0),                 # X = 0, no carry, top = 5: negative
        (0, 0, 0, 0, 7, 0),                       # top from the counter k1
        (P, 0, 0, 1, 0, 1),                       # carry, X = 0, X + EPS < top = 2^32: negative with c = 1
        (P + 1, 0, 0, 1, 0, 1),                   # carry, X + EPS = top: exactly zero, not negative
        (P, 0, 0, 1, 0, 0),                       # carry and th = 0: th - c wraps, must NOT be flagged
        (2**64 - 1, 2**64 - 1, 2**64 - 1, 15, 15, 15),
        (0, 0, 0, 0, 0, 0),
        (1, 0, (2**32 - 1) << 32, 0, 15, 15),     # largest top against a tiny X
        (0, 1 << 32, 0, 0, 0, 3),                 # sl = 1: X = EPS, top = 3 * 2^32
    ]
    n_cold = 0
    for trial in range(400):
        sets = []
        for j in range(3):
            if trial < 60:
                sets.append(crafted[(trial + j * (trial // 9 + 1)) % len(crafted)] if (trial + j) % 4 else
                            tuple([rnd.randrange(2**64) for _ in range(3)] + [rnd.randrange(16) for _ in range(3)]))
            else:
                c = [rnd.randrange(2**64) if rnd.random() < 0.8 else rnd.choice([0, 1, P, 2**64 - 1, 2**32]) for _ in range(3)]
                sets.append(tuple(c + [rnd.randrange(16) for _ in range(3)]))
        lane = ai.Lane({}, dummy_pairs=())
        for a, (c0, c1, c2, k0, k1, k2) in zip(accs, sets):
            for i, cv in enumerate((c0, c1, c2)):
                lane.v[a.c[i][0]], lane.v[a.c[i][1]] = cv & M32, cv >> 32
            for i, kv in enumerate((k0, k1, k2)):
                lane.v[a.k[i]] = kv
        lane.run(lines)
        for j, st in enumerate(sets):
            got = lane.v[200 + 2 * j] | (lane.v[201 + 2 * j] << 32)
            assert got % P == value(*st) % P, (trial, j, [hex(x) for x in st])
        n_cold += any(lane.s.get(14 + 2 * j, 0) for j in range(3))
    assert n_cold >= 20


def _f6_mulmod(u, v):
    t = [0] * 12
    for i, x in enumerate(u):
        for j, y in enumerate(v):
            t[i + j] += x * y
    return [(t[k] + 7 * t[k + 6]) % P for k in range(6)]


def test_generated_fp6_blocks_on_the_cpu():
    """every block of fp6_asm.inc (plain product / square and the fused product + linear-term blocks) through the
    interpreter, loose and edge operands, against plain integers; carry wait states checked on the way"""
    blocks = ai.extract_blocks(F6_INC)
    assert len(blocks) == 11         # nine Fp6 blocks + the two Fp3 blocks (test_generated_fp3_blocks_on_the_cpu); the
    # single-accumulator block returns u64 and is read by test_generated_acc3_block_on_the_cpu
    rnd = random.Random(11)
    edge = [0, 1, P - 1, P, 2**64 - 1, 2**32 - 1, 2**32, 2**64 - 2**32]

    def elem(kind):
        if kind == "edge":
            return [rnd.choice(edge) for _ in range(6)]
        if kind == "max":
            return [2**64 - 1] * 6
        if kind == "zero":
            return [0] * 6
        if kind == "sparse":     # k 2^48 in one coefficient: products k k' 2^96 = -k k' -- the reductions' cold paths
            out = [0] * 6
            out[rnd.randrange(6)] = rnd.randrange(1, 2**16) << 48
            return out
        return [rnd.randrange(2**64) for _ in range(6)]

    spec = {   # name -> (is square, second product, linear terms [(sign, c, operand)])
        "f6_mul_core_asm": (False, False, []),
        "f6_sqr_core_asm": (True, False, []),
        "f6_sqr_sub2_core_asm": (True, False, [(-1, 1, "x"), (-1, 1, "y")]),
        "f6_sqr_add3x_core_asm": (True, False, [(1, 3, "x")]),
        "f6_sqr_sub4x_core_asm": (True, False, [(-1, 4, "x")]),
        "f6_mul_sub8x_core_asm": (False, False, [(-1, 8, "x")]),
        "f6_mul_subx_core_asm": (False, False, [(-1, 1, "x")]),
        "f6_mul2_add_core_asm": (False, True, []),
        "f6_sqr_subx_sub2y_core_asm": (True, False, [(-1, 1, "x"), (-1, 2, "y")]),
    }
    kinds = ["rand"] * 6 + ["edge"] * 4 + ["max", "zero"] + ["sparse"] * 6
    cold_taken = 0
    for name, (is_sqr, two, terms) in spec.items():
        lines, outs, ins = blocks[name]
        for kind in kinds:
            arr = {"a": elem(kind), "b": elem("rand" if kind == "zero" else kind), "c": elem(kind), "d": elem("rand"),
                   "x": elem(kind), "y": elem("edge" if kind == "rand" else kind)}
            if kind == "sparse":
                arr["d"], arr["x"], arr["y"] = elem("sparse"), [0] * 6, [0] * 6
            # the pre-scaled operands the C++ wrappers hand over (any representative mod p is allowed: canonical here)
            arr["b7"] = [7 * t % P for t in arr["b"]]
            arr["d7"] = [7 * t % P for t in arr["d"]]
            arr["a2"] = [2 * t % P for t in arr["a"]]
            arr["a7"] = [7 * t % P for t in arr["a"]]
            arr["a14"] = [14 * t % P for t in arr["a"]]
            env = {}
            for nm, (half, an, idx) in ins.items():
                val = arr[an][idx]
                env["%%[%s]" % nm] = val & M32 if half == "lo32" else val >> 32
            lane = ai.Lane(env, dummy_pairs=())
            # the opening multiplies write their (impossible) carry to s[0:1] and nobody reads it before it is rewritten
            e = lane.run(lines)
            cold_taken += any(lb.endswith("_fix_%=") for lb in getattr(lane, "visited", ()))
            got = [(e["%%[r%dl]" % k] | (e["%%[r%dh]" % k] << 32)) % P for k in range(6)]
            want = _f6_mulmod(arr["a"], arr["a"] if is_sqr else arr["b"])
            if two:
                want = [(w + z) % P for w, z in zip(want, _f6_mulmod(arr["c"], arr["d"]))]
            for sign, c, an in terms:
                want = [(w + sign * c * t) % P for w, t in zip(want, arr[an])]
            assert got == want, (name, kind)
    assert cold_taken >= 12       # the sparse operands did take the reductions' cold path (negative result: - EPS there)


def test_generated_fp3_blocks_on_the_cpu():
    """the Fp3 = Fp[t]/(t^3 - 7) product and square of the square-root descent (fp3.hpp; round 4) through the interpreter:
    loose, edge and sparse operands (k 2^48: the reductions' cold path) against plain integers"""
    blocks = ai.extract_blocks(F6_INC)
    rnd = random.Random(12)
    edge = [0, 1, P - 1, P, 2**64 - 1, 2**32 - 1, 2**32, 2**64 - 2**32]

    def elem(kind):
        if kind == "edge":
            return [rnd.choice(edge) for _ in range(3)]
        if kind == "max":
            return [2**64 - 1] * 3
        if kind == "zero":
            return [0] * 3
        if kind == "sparse":
            out = [0] * 3
            out[rnd.randrange(3)] = rnd.randrange(1, 2**16) << 48
            return out
        return [rnd.randrange(2**64) for _ in range(3)]

    def mulmod(u, v):
        t = [0] * 5
        for i, x in enumerate(u):
            for j, y in enumerate(v):
                t[i + j] += x * y
        return [(t[k] + 7 * (t[k + 3] if k + 3 < 5 else 0)) % P for k in range(3)]

    cold_taken = 0
    for name, is_sqr in (("f3_mul_core_asm", False), ("f3_sqr_core_asm", True)):
        lines, outs, ins = blocks[name]
        for kind in ["rand"] * 8 + ["edge"] * 6 + ["max", "zero"] + ["sparse"] * 8:
            arr = {"a": elem(kind), "b": elem("rand" if kind == "zero" else kind)}
            arr["b7"] = [7 * t % P for t in arr["b"]]
            arr["a2"] = [2 * t % P for t in arr["a"]]
            arr["a7"] = [7 * t % P for t in arr["a"]]
            arr["a14"] = [14 * t % P for t in arr["a"]]
            env = {}
            for nm, (half, an, idx) in ins.items():
                val = arr[an][idx]
                env["%%[%s]" % nm] = val & M32 if half == "lo32" else val >> 32
            lane = ai.Lane(env, dummy_pairs=())
            e = lane.run(lines)
            cold_taken += any(lb.endswith("_fix_%=") for lb in getattr(lane, "visited", ()))
            got = [(e["%%[r%dl]" % k] | (e["%%[r%dh]" % k] << 32)) % P for k in range(3)]
            assert got == mulmod(arr["a"], arr["a"] if is_sqr else arr["b"]), (name, kind)
    assert cold_taken >= 4


def test_generated_acc3_block_on_the_cpu():
    """fp_acc3_core_asm (one lane's three products of a cooperative Fp6 product round, one padded reduction): random, edge and
    sparse operands (k 2^48 pairs: the reduction's cold path) against plain integers; wait states enforced by the interpreter"""
    txt = open(F6_INC).read()
    m = re.search(r"SSA_DEV u64 fp_acc3_core_asm\(.*?asm\(\n(.*?)\n        : ", txt, re.S)
    lines = [ln.strip().strip('"').replace("\\n\\t", "") for ln in m.group(1).split("\n") if ln.strip()]
    rnd = random.Random(13)
    edge = [0, 1, P - 1, P, 2**64 - 1, 2**32 - 1, 2**32, 2**64 - 2**32]
    cold = 0
    for it in range(60):
        if it < 25:
            x = [rnd.randrange(2**64) for _ in range(3)]
            y = [rnd.randrange(2**64) for _ in range(3)]
        elif it < 40:
            x = [rnd.choice(edge) for _ in range(3)]
            y = [rnd.choice(edge) for _ in range(3)]
        elif it == 40:
            x, y = [2**64 - 1] * 3, [2**64 - 1] * 3
        else:
            x, y = [0] * 3, [0] * 3
            j = rnd.randrange(3)
            x[j], y[j] = rnd.randrange(1, 2**16) << 48, rnd.randrange(1, 2**16) << 48
        env = {}
        for nm, arr in (("x", x), ("y", y)):
            for j in range(3):
                env["%%[%s%dl]" % (nm, j)] = arr[j] & M32
                env["%%[%s%dh]" % (nm, j)] = arr[j] >> 32
        lane = ai.Lane(env, dummy_pairs=())
        e = lane.run(lines)
        cold += any(lb.startswith("L_fix") for lb in getattr(lane, "visited", ()))
        got = (e["%[rl]"] | (e["%[rh]"] << 32)) % P
        assert got == sum(a * b for a, b in zip(x, y)) % P, (it, x, y)
    assert cold >= 5


def test_asm_blocks_declare_what_they_clobber():
    """A scalar ALU instruction inside a block (s_andn2_b64, s_sub_u32, s_cmp_*) rewrites SCC: the block must say so,
    or the compiler keeps a loop condition alive across it (a GPU hang in round 2 before the clobber was added).
    Every fixed VGPR / SGPR the strings name must be on the clobber list too."""
    import re
    for path in (INC, F6_INC):
        txt = open(path).read()
        for m in re.finditer(r"SSA_DEV (?:void|u32|u64) (\w+)\(.*?asm(?: volatile)?\(\n(.*?)\n        : (.*?)\n        :(.*?)\n        : (\".*?)\);", txt, re.S):
            name, body, clob = m.group(1), m.group(2), m.group(5)
            clobbers = set(re.findall(r'"(\w+)"', clob))
            # registers an operand is PINNED to ("+{v[72:73]}"(x)) are declared by the operand, and must not be clobbers too
            for lo, hi in re.findall(r'"[=+]&?\{v\[(\d+):(\d+)\]\}"', m.group(3) + m.group(4)):
                for r in range(int(lo), int(hi) + 1):
                    assert "v%d" % r not in clobbers, (name, r)
                    clobbers.add("v%d" % r)
            if re.search(r'"s_(andn2|and|xor|or|sub|add|cmp)', body):
                assert "scc" in clobbers, name
            for reg in set(re.findall(r"\bv(\d+)\b", body)):
                assert "v" + reg in clobbers, (name, "v" + reg)
            for lo, hi in set(re.findall(r"\bv\[(\d+):(\d+)\]", body)):
                for r in range(int(lo), int(hi) + 1):
                    assert "v%d" % r in clobbers, (name, r)
            for lo, hi in set(re.findall(r"\bs\[(\d+):(\d+)\]", body)):
                for r in range(int(lo), int(hi) + 1):
                    assert "s%d" % r in clobbers, (name, "s%d" % r)
            if "vcc" in body:
                assert "vcc" in clobbers, name


JAC_INC = os.path.join(os.path.dirname(INC), "jac_asm.inc")


def _jac_lines():
    txt = open(JAC_INC).read()
    am = re.search(r"asm volatile\(\n(.*?)\n        : ", txt, re.S)
    return [ln.strip().strip('"').replace("\\n\\t", "") for ln in am.group(1).split("\n") if ln.strip()], txt


def _jac_dbl_model(X, Y, Z):
    """dbl-2007-bl with a = 1 on plain integers (the formulas of curve.hpp's compiled jac_dbl)"""
    add = lambda u, v: [(a + b) % P for a, b in zip(u, v)]
    sub = lambda u, v: [(a - b) % P for a, b in zip(u, v)]
    sc = lambda c, u: [c * a % P for a in u]
    XX, YY, ZZ = _f6_mulmod(X, X), _f6_mulmod(Y, Y), _f6_mulmod(Z, Z)
    YYYY = _f6_mulmod(YY, YY)
    t = add(X, YY)
    S = sc(2, sub(sub(_f6_mulmod(t, t), XX), YYYY))
    M = add(sc(3, XX), _f6_mulmod(ZZ, ZZ))
    X3 = sub(_f6_mulmod(M, M), sc(2, S))
    Y3 = sub(_f6_mulmod(M, sub(S, X3)), sc(8, YYYY))
    yz = add(Y, Z)
    return X3, Y3, sub(sub(_f6_mulmod(yz, yz), YY), ZZ)


def test_generated_doubling_on_the_cpu():
    """jac_asm.inc (n doublings as one asm statement) through the interpreter against the textbook formulas: random,
    edge and all-ones operands, and operands whose high words are all ones so that every guarded pre-scaling site takes
    its cold path; carry wait states checked on the way"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gen_jac_asm as gj
    lines, _ = _jac_lines()
    rnd = random.Random(21)
    edge = [0, 1, P - 1, P, 2**64 - 1, 2**32 - 1, 2**32, 2**64 - 2**32, 2**64 - 2**31, 2**63, 2**64 - 5]

    def elem(kind):
        if kind == "edge":
            return [rnd.choice(edge) for _ in range(6)]
        if kind == "max":
            return [2**64 - 1] * 6
        if kind == "zero":
            return [0] * 6
        if kind == "hi":
            return [(0xFFFFFFFF << 32) | rnd.randrange(2**32) for _ in range(6)]
        if kind == "near":       # c * a_hi lands within a few units of 2^32 for c = 3, 6, 7, 21: the multiples' guards
            out = []
            for _ in range(6):
                c = rnd.choice([3, 7, 21])
                hi = (2**32 - rnd.randrange(1, 64)) * pow(c, -1, 2**32) % 2**32
                out.append((hi << 32) | rnd.choice([rnd.randrange(2**32), 0xFFFFFFFF, 0]))
            return out
        if kind == "sparse":     # k 2^48 in the first coefficient only: Y^2, Z^2, Y * 2Z are -(small): the reductions' cold paths
            return [rnd.randrange(1, 2**15) << 48] + [0] * 5
        return [rnd.randrange(2**64) for _ in range(6)]

    cold_seen = 0
    red_cold = 0
    for kind in ["rand"] * 3 + ["edge"] * 5 + ["max", "zero", "hi", "hi", "hi", "near", "near", "near", "sparse", "sparse", "sparse"]:
        for n in (1, 3):
            pt = [elem(kind), elem("rand" if kind == "zero" else kind), elem(kind)]
            lane = ai.Lane({"%[n]": n})
            for regs, val in zip((gj.XR, gj.YR, gj.ZR), pt):
                for j in range(6):
                    lane.v[regs[j]], lane.v[regs[j] + 1] = val[j] & M32, val[j] >> 32
            lane.run(lines)
            cold_seen += lane.issued > n * 3000
            red_cold += any("red" in lb and lb.endswith("_fix_%=") for lb in getattr(lane, "visited", ()))
            got = [[(lane.v[r] | (lane.v[r + 1] << 32)) % P for r in regs] for regs in (gj.XR, gj.YR, gj.ZR)]
            want = pt
            for _ in range(n):
                want = _jac_dbl_model(*want)
            assert got == [list(w) for w in want], (kind, n)
    assert cold_seen >= 6         # the cold paths did run
    assert red_cold >= 6          # ... those of the reductions too (sparse operands)


def _f6_sub(u, v):
    return [(a - b) % P for a, b in zip(u, v)]


def _jac_madd_model(X, Y, Z, x2, y2):
    ZZ = _f6_mulmod(Z, Z)
    H = _f6_sub(_f6_mulmod(x2, ZZ), X)
    R = _f6_sub(_f6_mulmod(_f6_mulmod(y2, Z), ZZ), Y)
    HH = _f6_mulmod(H, H)
    HHH, V = _f6_mulmod(H, HH), _f6_mulmod(X, HH)
    X3 = _f6_sub(_f6_sub(_f6_mulmod(R, R), HHH), [2 * v % P for v in V])
    Y3 = _f6_sub(_f6_mulmod(R, _f6_sub(V, X3)), _f6_mulmod(Y, HHH))
    return [X3, Y3, _f6_mulmod(Z, H)], H


def _asm_fn(txt, name):
    fn = txt.split("SSA_DEV u32 %s" % name)[1].split("\n}\n")[0]
    am = re.search(r"asm volatile\(\n(.*?)\n        : ", fn, re.S)
    lines = [ln.strip().strip('"').replace("\\n\\t", "") for ln in am.group(1).split("\n") if ln.strip()]
    ins = [(int(a), nm) for a, _, nm, _ in re.findall(r'"\{v\[(\d+):(\d+)\]\}"\((\w+)\[(\d)\]\)', fn)]
    return lines, [a for a, nm in ins if nm == "x2"], [a for a, nm in ins if nm == "y2"]


def test_generated_mixed_addition_and_window_on_the_cpu():
    """jac_madd_asm and jac_window_asm (n doublings + the addition under a narrowed EXEC) through the interpreter:
    generic inputs against the textbook formulas, cold paths, and the three exceptional inputs -- Z, x2 or H with a first
    coefficient = 0 mod p -- which must leave the point untouched and report 0"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gen_jac_asm as gj
    txt = open(JAC_INC).read()
    rnd = random.Random(31)

    def elem(kind):
        if kind == "hi":
            return [(0xFFFFFFFF << 32) | rnd.randrange(2, 2**32) for _ in range(6)]
        if kind == "max":
            return [2**64 - 1] * 6
        if kind == "edge":
            return [rnd.choice([P - 1, P + 1, 2**64 - 1, 2**32 - 1, 2**32, 2**64 - 2**32, 2**64 - 2**31, 2**63, 1]) for _ in range(6)]
        return [rnd.randrange(1, 2**64) for _ in range(6)]

    def run(lines, x2r, y2r, pt, q, env):
        lane = ai.Lane(dict(env, **{"%[ok]": 7}))
        for regs, val in zip((gj.XR, gj.YR, gj.ZR, x2r, y2r), list(pt) + list(q)):
            for j in range(6):
                lane.v[regs[j]], lane.v[regs[j] + 1] = val[j] & M32, val[j] >> 32
        lane.run(lines)
        assert getattr(lane, "exec_bit", 1) == 1
        return [[(lane.v[r] | (lane.v[r + 1] << 32)) % P for r in regs] for regs in (gj.XR, gj.YR, gj.ZR)], lane.env["%[ok]"]

    lines, x2r, y2r = _asm_fn(txt, "jac_madd_asm")
    generic = 0
    for kind in ["rand"] * 4 + ["edge"] * 6 + ["max", "hi", "hi", "hi"]:
        pt, q = [elem(kind) for _ in range(3)], [elem(kind) for _ in range(2)]
        got, ok = run(lines, x2r, y2r, pt, q, {})
        want, H = _jac_madd_model(*pt, *q)
        if pt[2][0] % P == 0 or q[0][0] % P == 0 or H[0] == 0:
            assert ok == 0 and got == [[c % P for c in v] for v in pt]
        else:
            generic += 1
            assert ok == 1 and got == want, kind
    assert generic >= 10
    for case in range(6):        # exceptional inputs
        pt, q = [elem("rand") for _ in range(3)], [elem("rand") for _ in range(2)]
        if case < 2:
            pt[2][0] = (0, P)[case]
        elif case < 4:
            q[0][0] = (0, P)[case - 2]
        else:
            u = _f6_mulmod(q[0], _f6_mulmod(pt[2], pt[2]))
            pt[0][0] = u[0] if case == 4 or u[0] + P >= 2**64 else u[0] + P
        got, ok = run(lines, x2r, y2r, pt, q, {})
        assert ok == 0 and got == [[c % P for c in v] for v in pt], case
    # the window statement gathers (x2, y2) itself: six 16-byte loads from the lane's table row at its start, one wait
    # behind the doublings and in front of the EXEC narrowing (every path out of the statement passes it)
    lines, x2r, y2r = _asm_fn(txt, "jac_window_asm")
    assert x2r == [] and y2r == []
    loads = [ln for ln in lines if ln.startswith("global_load")]
    assert lines[0] == "s_waitcnt vmcnt(0)"       # nothing of the compiler's in flight when the statement starts (gen_jac_asm.py)
    assert len(loads) == 6 and lines[1:7] == loads
    wait = lines.index("s_waitcnt vmcnt(0)", 1)
    assert lines[wait + 1].startswith("v_cmp_ne_u32 vcc, 0, %[act]") and lines[wait - 1].startswith("s_cbranch_scc1 L_top")
    assert not any(ln.startswith("s_waitcnt") for i, ln in enumerate(lines) if i not in (0, wait))
    in_regs = gj.build_madd("m").IN
    x2r, y2r = in_regs[:6], in_regs[6:]

    def run_window(pt, q, env):
        lane = ai.Lane(dict(env, **{"%[ok]": 7}))
        lane.mem = {"%[row]": [w for coord in q for c in coord for w in (c & M32, c >> 32)]}
        for regs, val in zip((gj.XR, gj.YR, gj.ZR), pt):
            for j in range(6):
                lane.v[regs[j]], lane.v[regs[j] + 1] = val[j] & M32, val[j] >> 32
        for r in x2r + y2r:                       # whatever the registers held before the statement must not matter
            lane.v[r], lane.v[r + 1] = 0xDEADBEEF, 0xFEEDFACE
        lane.run(lines)
        assert getattr(lane, "exec_bit", 1) == 1
        return [[(lane.v[r] | (lane.v[r + 1] << 32)) % P for r in regs] for regs in (gj.XR, gj.YR, gj.ZR)], lane.env["%[ok]"]

    # the mixed addition that gathers its own operand (comb tables): the same cases as jac_madd_asm through the loads; the
    # additions' loads are awaited once, in front of the first use of (x2, y2), the touches of the next entry at the end
    glines, gx, gy = _asm_fn(txt, "jac_madd_gather_asm")
    assert gx == [] and gy == []
    assert glines[0] == "s_waitcnt vmcnt(0)"
    assert [ln.split()[0] for ln in glines[1:9]] == ["global_load_dwordx4"] * 6 + ["global_load_dword"] * 2
    qwait = glines.index("s_waitcnt vmcnt(2)")
    first_use = min(i for i, ln in enumerate(glines) if i >= 9 and re.search(r"\bv(%s)\b" % "|".join(str(r + h) for r in in_regs for h in (0, 1)), ln))
    assert qwait < first_use and glines[-1] == "s_waitcnt vmcnt(0)" and glines[-2].startswith("L_end")
    assert sum(ln.startswith("s_waitcnt") for ln in glines) == 3

    def run_gather(pt, q):
        lane = ai.Lane({"%[ok]": 7})
        words = [w for coord in q for c in coord for w in (c & M32, c >> 32)]
        lane.mem = {"%[row]": words, "%[next]": [0x5A5A5A5A] * 24}
        for regs, val in zip((gj.XR, gj.YR, gj.ZR), pt):
            for j in range(6):
                lane.v[regs[j]], lane.v[regs[j] + 1] = val[j] & M32, val[j] >> 32
        for r in x2r + y2r:
            lane.v[r], lane.v[r + 1] = 0xDEADBEEF, 0xFEEDFACE
        lane.run(glines)
        return [[(lane.v[r] | (lane.v[r + 1] << 32)) % P for r in regs] for regs in (gj.XR, gj.YR, gj.ZR)], lane.env["%[ok]"]

    generic = 0
    for kind in ["rand"] * 4 + ["edge"] * 6 + ["max", "hi", "hi", "hi"]:
        pt, q = [elem(kind) for _ in range(3)], [elem(kind) for _ in range(2)]
        got, ok = run_gather(pt, q)
        want, H = _jac_madd_model(*pt, *q)
        if pt[2][0] % P == 0 or q[0][0] % P == 0 or H[0] == 0:
            assert ok == 0 and got == [[c % P for c in v] for v in pt]
        else:
            generic += 1
            assert ok == 1 and got == want, kind
    assert generic >= 10
    for case in range(6):        # exceptional inputs
        pt, q = [elem("rand") for _ in range(3)], [elem("rand") for _ in range(2)]
        if case < 2:
            pt[2][0] = (0, P)[case]
        elif case < 4:
            q[0][0] = (0, P)[case - 2]
        else:
            u = _f6_mulmod(q[0], _f6_mulmod(pt[2], pt[2]))
            pt[0][0] = u[0] if case == 4 or u[0] + P >= 2**64 else u[0] + P
        got, ok = run_gather(pt, q)
        assert ok == 0 and got == [[c % P for c in v] for v in pt], case

    for kind in ["rand", "rand", "hi", "max"]:
        for act in (0, 5):
            for n in (1, 4):
                pt, q = [elem(kind) for _ in range(3)], [elem(kind) for _ in range(2)]
                got, ok = run_window(pt, q, {"%[act]": act, "%[n]": n})
                want = pt
                for _ in range(n):
                    want = [list(v) for v in _jac_dbl_model(*want)]
                if act:
                    w2, H = _jac_madd_model(*want, *q)
                    if want[2][0] == 0 or q[0][0] % P == 0 or H[0] == 0:
                        assert ok == 0 and got == want
                        continue
                    want = w2
                assert ok == 1 and got == want, (kind, act, n)


def test_doubling_prescale_sites_fast_and_cold():
    """one pre-scaling site in isolation (2a, c a, 2 c a): the short forms and, for operands that trip the guard, the
    exact cold forms"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gen_jac_asm as gj
    rnd = random.Random(4)
    for trial in range(600):
        gen = gj.Gen()
        src, PB = gj.XR, gen.PB
        c = rnd.choice([3, 6, 7, 21])
        gen.prescale(src, [(PB[0], 0)], [(PB[1], 1, c), (PB[2], 2, c)], [(PB[3], PB[1])])
        lines = ["v_mov_b32 v%d, 0" % r for r in gen.zero_regs] + gen.main + ["s_branch L_end"] + gen.cold + ["L_end:"]
        lane = ai.Lane({})
        vals = []
        for j in range(3):
            kind = trial % 3
            if kind == 0:
                v = rnd.randrange(2**64)
            elif kind == 1:
                v = (0xFFFFFFFF << 32) | rnd.choice([rnd.randrange(2**32), 0, 1, 0xFFFFFFFF])
            else:
                hi = (2**32 - rnd.randrange(1, 300)) * pow(c | 1, -1, 2**32) % 2**32
                v = (hi << 32) | rnd.choice([rnd.randrange(2**32), 0xFFFFFFFF, 0])
            vals.append(v)
            lane.v[src[j]], lane.v[src[j] + 1] = v & M32, v >> 32
        lane.run(lines)
        rd = lambda p: lane.v[p] | (lane.v[p + 1] << 32)
        assert rd(src[0]) % P == vals[0] % P                  # canonicalised in place at most
        assert rd(PB[0]) % P == 2 * vals[0] % P and rd(PB[1]) % P == c * vals[1] % P
        assert rd(PB[2]) % P == c * vals[2] % P and rd(PB[3]) % P == 2 * c * vals[1] % P, (trial, c)


def test_doubling_asm_declares_its_registers_and_kernels_leave_room():
    """every VGPR / SGPR the doubling names is pinned or on the clobber list, SCC and VCC are declared, and every
    kernel that inlines it is built for at most two waves per SIMD (the block owns registers up to v255)"""
    _, whole = _jac_lines()
    fns = re.findall(r"SSA_DEV \w+ (\w+)\(.*?asm volatile\(\n(.*?)\n        : (.*?)\);\n", whole, re.S)
    assert [f[0] for f in fns] == ["jac_dbl_n_asm", "jac_madd_asm", "jac_window_asm", "jac_madd_gather_asm"]
    for name, body, tail in fns:
        clob = set(re.findall(r'"(\w+)"', tail.split("\n        : ")[-1]))
        pinned = set()
        for lo, hi in re.findall(r'"\+?\{v\[(\d+):(\d+)\]\}"', tail):
            pinned.update(range(int(lo), int(hi) + 1))
        assert not (pinned & set(int(c[1:]) for c in clob if re.match(r"v\d+$", c))), name
        for reg in set(re.findall(r"\bv(\d+)\b", body)):
            assert "v" + reg in clob or int(reg) in pinned, (name, reg)
        for lo, hi in set(re.findall(r"\bv\[(\d+):(\d+)\]", body)):
            assert int(lo) % 2 == 0 and all("v%d" % r in clob or r in pinned for r in range(int(lo), int(hi) + 1)), (name, lo)
        for lo, hi in set(re.findall(r"\bs\[(\d+):(\d+)\]", body)):
            assert all("s%d" % r in clob for r in range(int(lo), int(hi) + 1)), (name, lo)
        assert {"scc", "vcc"} <= clob and ("s20" in clob or "s20" not in body), name
        if "saveexec" in body or "s_mov_b64 exec" in body:       # EXEC is narrowed and restored inside the statement
            assert body.count("s_and_saveexec_b64") == 1 and "s_mov_b64 exec, s[22:23]" in body
        # a branch on VCC reads a VCC the scalar unit wrote (s_and_b64 vcc, exec, vcc after the VALU compare)
        lines_ = [ln.strip().strip('"').replace("\\n\\t", "") for ln in body.split("\n")]
        for i, ln in enumerate(lines_):
            if ln.startswith("s_cbranch_vcc"):
                assert lines_[i - 1] == "s_and_b64 vcc, exec, vcc", (name, i)
    src = open(os.path.join(os.path.dirname(INC), "ssa_kernels.hpp")).read()
    for m in re.finditer(r"__global__ void\s*(__launch_bounds__\(([^)]*)\))?[^{;]*?\b(ssa_k_\w+)\(.*?\n}\n", src, re.S):
        if re.search(r"mul_ptab\(|jac_dbl_n\(|add_base_mul\(|jac_madd_fast\(", m.group(0)):
            # at most 256 threads per block: 256 VGPRs per lane stay allocatable (no bound = 1024 threads = 128 VGPRs)
            assert m.group(2) and int(m.group(2).split(",")[0]) <= 256, m.group(3)


def test_generated_files_are_up_to_date():
    """fp_chain_asm.inc, jac_asm.inc, fp6_asm.inc and qnaf.inc are what their generators produce (no hand edits, no stale
    generator).  The text is generated IN MEMORY and compared: a test must not write tracked sources (a rewrite bumps
    their mtimes, and a stale generator would overwrite the committed file before the assertion fires)."""
    import importlib.util
    import io
    from contextlib import redirect_stdout
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    for tool, inc in (("gen_fp_chain_asm", "fp_chain_asm.inc"), ("gen_jac_asm", "jac_asm.inc"), ("gen_f6_asm", "fp6_asm.inc"),
                      ("gen_qnaf", "qnaf.inc")):
        path = os.path.join(root, "schnorr-sig_amd", "csrc", inc)
        mtime = os.path.getmtime(path)
        spec = importlib.util.spec_from_file_location(tool, os.path.join(root, "tools", tool + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        with redirect_stdout(io.StringIO()):
            text = mod.generate()
        assert os.path.abspath(mod.OUT_PATH) == os.path.abspath(path)
        assert text == open(path).read(), "%s is not what tools/%s.py generates" % (inc, tool)
        assert os.path.getmtime(path) == mtime, "the freshness test wrote " + inc


def test_order_q_schedule_is_q():
    """qnaf.inc: the (gap, digit) schedule the subgroup check runs evaluates to the subgroup order (plain integers),
    uses only the odd table rows 1P..15P and never asks the window statement for zero doublings"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "schnorr-sig_amd", "csrc", "qnaf.inc")).read()
    digits = [int(v) for v in re.search(r"QNAF_DIGIT\[QNAF_LEN\] = \{(.*?)\}", txt).group(1).split(",")]
    gaps = [int(v) for v in re.search(r"QNAF_GAP\[QNAF_LEN\] = \{(.*?)\}", txt).group(1).split(",")]
    n = int(re.search(r"QNAF_LEN = (\d+)", txt).group(1))
    assert len(digits) == len(gaps) == n
    acc = 0
    for g, d in zip(gaps, digits):
        acc = (acc << g) + d
    assert acc == 0x7AF2599B3B3F22D0563FBF0F990A37B5327AA72330157722D443623EAED4ACCF
    assert gaps[0] == 0 and 1 <= digits[0] <= 16 and all(1 <= g <= 64 for g in gaps[1:])
    assert all(d % 2 and abs(d) <= 15 for d in digits)
    assert n == 44 and sum(gaps) == 255                 # width-5 NAF: 43 additions, 255 doublings


def test_every_generated_block_waits_for_the_compilers_loads_first():
    """Every asm block of the three generated files starts with s_waitcnt vmcnt(0).  The blocks name their registers
    themselves (clobber lists): the compiler keeps values in such registers between blocks, reloads them from scratch behind
    a block, and does NOT wait for such a reload in front of the next block that merely clobbers the register -- a reload
    still in flight lands in the block's temporaries (round 5: wrong signatures from ssa_k_sign on ~7 % of the waves,
    profiles/r05/gather_ab.txt).  The three small statements of fp.hpp take compiler-allocated operands only and are not
    concerned."""
    csrc = os.path.dirname(INC)
    n_blocks = 0
    for name in ("fp6_asm.inc", "fp_chain_asm.inc", "jac_asm.inc"):
        txt = open(os.path.join(csrc, name)).read()
        for m in re.finditer(r"asm(?: volatile)?\(\n\s*\"([^\"]*)\"", txt):
            n_blocks += 1
            first = m.group(1).replace("\\n\\t", "").strip()
            assert first == "s_waitcnt vmcnt(0)", (name, first)
    assert n_blocks >= 16
