"""Generates schnorr-sig_amd/csrc/jac_asm.inc: n consecutive Jacobian doublings of the ladder as ONE asm statement.

Why (round 2, DESIGN.md "Doubling as one block"): in the compiled doubling a quarter of the issue slots were not
arithmetic -- argument moves of the four out-of-line squarings (24 each), the compiler's pre-scaling of every
squaring operand (2a, 7a, 14a: ~60 instructions with their wrap-around fix-ups), three modular additions, s_nop
padding.  Here the whole doubling is generated: registers are assigned by hand (the point lives in pinned VGPRs, the
temporaries in clobbered ones), the blocks of tools/gen_f6_asm.py are instantiated on registers directly (no
parking of results, no moves), and the formulas are arranged so that pre-scaled operands are shared:

    YY = Y^2            ZZ = Z^2            Z3 = Y * (2Z)            YYYY = YY^2
    S' = X * (2 YY)                       (= ((X + YY)^2 - XX - YYYY) / 2: a product instead of a square + 2 additions)
    M  = ZZ^2 + 3 X^2                     (42 products in one accumulator, one reduction; XX is never formed)
    X3 = M^2 - 4 S'     Y3 = M (2 S' - X3) - 8 YYYY

eight reductions instead of nine, one modular subtraction instead of three additions, 2Z / 14Z and 2YY / 14YY are
the squarings' own pre-scaled operands.

Cheap pre-scaling behind a guard.  2a of a loose a is a shift plus EPS for the top bit (2 instructions); the sum
wraps a second time only for a >= 2^64 - 2^31.  c*a for a small constant is  (c a_lo + (c a_hi mod 2^32) 2^32) +
EPS * (c a_hi >> 32): two multiply-adds after the two halves of c * a_hi, each of which overflows only when a 32-bit
intermediate is within c of 2^32.  One v_max3 chain over those words and ONE compare-and-branch per site guards the
short forms; the cold path (canonicalise the doubled operands, exact carry-checked multiples) follows the loop.
tests/test_asm_emulation.py runs the generated text in the one-lane interpreter, cold paths included.
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_f6_asm as g6

# ---- registers ------------------------------------------------------------------------------------------------
# The point: callee-saved VGPRs of the amdgpu calling convention (v136-143, v152-159, ...), so that it survives the
# calls of the compiled mixed addition between two runs of doublings without being moved.
PIN = [136, 138, 140, 142, 152, 154,      # X
       156, 158, 168, 170, 172, 174,      # Y
       184, 186, 188, 190, 200, 202]      # Z
XR, YR, ZR = PIN[0:6], PIN[6:12], PIN[12:18]
_G6_FIXED = set(g6.POOL[:g6.N_FIXED + 4])               # accumulators, counters, complements (gen_f6_asm's registers)
_pin_regs = set(r for p in PIN for r in (p, p + 1))
FREE = [r for r in range(56, 256, 2) if r not in _G6_FIXED and r + 1 not in _G6_FIXED and r not in _pin_regs]


class Alloc:
    def __init__(self):
        self.free = list(FREE)
        self.used = set()

    def pair(self):
        r = self.free.pop(0)
        self.used.update((r, r + 1))
        return r

    def fp6(self):
        return [self.pair() for _ in range(6)]


THRESH = "0xffffff00"


class Gen:
    def __init__(self):
        self.al = Alloc()
        self.main = []          # the loop body
        self.cold = []          # cold paths, after the loop
        self.nsite = 0
        a = self.al
        self.S = [a.fp6() for _ in range(4)]          # Fp6 temporaries
        self.PB = [a.pair() for _ in range(23)]       # pre-scaled operands
        self.MASK = a.pair()                          # (mask, 0)
        self.Q = [a.pair(), a.pair()]                 # (0, low word of c * a_hi)
        self.T = [a.pair(), a.pair()]                 # c * a_lo + Q
        m = a.pair()
        self.UH = [m, m + 1]                          # high word of c * a_hi
        m = a.pair()
        self.G, self.TMP = m, m + 1                   # guard maximum, scratch
        self.zero_regs = [self.MASK + 1, self.Q[0], self.Q[1]]

    # ---- pre-scaled operands ----
    def prescale(self, src, dbl, mulc, dbl_of):
        """dbl: [(dst pair, j)]: dst = 2 src[j];  mulc: [(dst pair, j, c)]: dst = c src[j];
        dbl_of: [(dst pair, pair of a mulc result)]: dst = 2 * that.  Short forms + one guard; cold path appended."""
        site = self.nsite
        self.nsite += 1
        out = self.main
        guard_words = []
        # constant multiples (two sets of temporaries, alternating)
        for n, (dst, j, c) in enumerate(mulc):
            q, t, uh = self.Q[n % 2], self.T[n % 2], self.UH[n % 2]
            a = src[j]
            out += ["v_mul_lo_u32 v%d, v%d, %d" % (q + 1, a + 1, c),
                    "v_mul_hi_u32 v%d, v%d, %d" % (uh, a + 1, c),
                    "v_mad_u64_u32 v[%d:%d], s[0:1], v%d, %d, v[%d:%d]" % (t, t + 1, a, c, q, q + 1),
                    "v_mad_u64_u32 v[%d:%d], s[0:1], v%d, -1, v[%d:%d]" % (dst, dst + 1, uh, t, t + 1)]
            out += [self._max3(n == 0, "v%d" % (q + 1), "v%d" % (t + 1))]
        first = not mulc
        words = ["v%d" % (src[j] + 1) for _, j in dbl]
        for i in range(0, len(words), 2):
            pair = words[i:i + 2]
            if len(pair) == 1:
                pair.append(pair[0])
            out += [self._max3(first, pair[0], pair[1])]
            first = False
        body = []
        for dst, j in dbl:
            body += self._dbl(dst, src[j])
        for dst, sp in dbl_of:
            body += self._dbl(dst, sp)
        out += body
        out += ["v_cmp_le_u32 vcc, %s, v%d" % (THRESH, self.G), "s_cbranch_vccnz L_cold%d_%%=" % site, "L_cont%d_%%=:" % site]
        # cold path
        cold = ["L_cold%d_%%=:" % site]
        for _, j in dbl:
            cold += self._canon(src[j])
        for dst, j, c in mulc:
            cold += self._mulc_exact(dst, src[j], c)
        for _, sp in dbl_of:          # an exact multiple may have any high word: canonicalise before the short doubling
            cold += self._canon(sp)
        cold += body
        cold += ["s_branch L_cont%d_%%=" % site]
        self.cold += cold

    def _max3(self, first, a, b):
        if first:
            return "v_max_u32 v%d, %s, %s" % (self.G, a, b)
        return "v_max3_u32 v%d, v%d, %s, %s" % (self.G, self.G, a, b)

    def _dbl(self, dst, a):
        return ["v_ashrrev_i32 v%d, 31, v%d" % (self.MASK, a + 1),
                "v_lshl_add_u64 v[%d:%d], v[%d:%d], 1, v[%d:%d]" % (dst, dst + 1, a, a + 1, self.MASK, self.MASK + 1)]

    def _canon(self, a):
        """a -= p where a >= p and the high word is all ones (cold): a + EPS mod 2^64 = (lo - 1, 0)"""
        t = self.TMP
        return ["v_cmp_eq_u32 s[0:1], -1, v%d" % (a + 1), "v_cmp_ne_u32 s[2:3], 0, v%d" % a, "s_and_b64 s[0:1], s[0:1], s[2:3]",
                "v_cndmask_b32 v%d, 0, -1, s[0:1]" % t, "v_add_co_u32 v%d, s[2:3], v%d, v%d" % (a, a, t), "s_nop 1",
                "v_addc_co_u32 v%d, s[2:3], v%d, 0, s[2:3]" % (a + 1, a + 1)]

    def _mulc_exact(self, dst, a, c):
        """dst = c * a mod p with every carry checked (cold): (t_l, t_h + u_l) + EPS * (u_h + carry), then the wrap"""
        t, uh, x = self.T[0], self.UH[0], self.TMP
        return ["v_mul_lo_u32 v%d, v%d, %d" % (x, a + 1, c), "v_mul_hi_u32 v%d, v%d, %d" % (uh, a + 1, c),
                "v_mad_u64_u32 v[%d:%d], s[0:1], v%d, %d, 0" % (t, t + 1, a, c),
                "v_add_co_u32 v%d, s[2:3], v%d, v%d" % (t + 1, t + 1, x), "s_nop 1",
                "v_addc_co_u32 v%d, s[2:3], v%d, 0, s[2:3]" % (uh, uh),
                "v_mad_u64_u32 v[%d:%d], s[2:3], v%d, -1, v[%d:%d]" % (dst, dst + 1, uh, t, t + 1), "s_nop 1",
                "v_cndmask_b32 v%d, 0, -1, s[2:3]" % x, "v_add_co_u32 v%d, s[2:3], v%d, v%d" % (dst, dst, x), "s_nop 1",
                "v_addc_co_u32 v%d, s[2:3], v%d, 0, s[2:3]" % (dst + 1, dst + 1)]

    # ---- blocks of gen_f6_asm on registers ----
    def block(self, terms, regs, out, extras=(), extra_regs=None):
        """terms[k]: [(name, name)]; regs: {name: pair}; out: six pairs; extras: [(sign, c, prefix)] with
        extra_regs[prefix] = six pairs"""
        accs = [g6.Acc(j) for j in range(6)]
        lines = []
        bias, bias_setup = g6.extras_bias(extras)
        lines += bias_setup
        for g in range(2):
            for k in range(3 * g, 3 * g + 3):
                t = terms[k]
                lines += g6.init2(accs[k], t[0][0], t[0][1], t[1][0], t[1][1], bias)
                for x, y in t[2:]:
                    lines += g6.mac(accs[k], x, y)
                lines += g6.extra_terms(accs[k], k, extras)
            outs = [("v%d" % out[k], "v%d" % (out[k] + 1)) for k in range(3 * g, 3 * g + 3)]
            lines += g6.reduce3(accs[3 * g:3 * g + 3], outs)
        m = dict(regs)
        for _, _, prefix in extras:
            for k in range(6):
                m["%s%d" % (prefix, k)] = extra_regs[prefix][k]

        def sub(mo):
            nm, half = mo.group(1), mo.group(2)
            return "v%d" % (m[nm] + (1 if half == "h" else 0))
        self.main += [re.sub(r"%\[(\w+?)([lh])\]", sub, ln) for ln in lines]

    # ---- 2a - b, coefficient-wise, in place of a (one guard: the second borrow needs b's high word all ones) ----
    def dbl_sub(self, a, b):
        site = self.nsite
        self.nsite += 1
        out = self.main
        words = ["v%d" % (a[j] + 1) for j in range(6)] + ["v%d" % (b[j] + 1) for j in range(6)]
        for i in range(0, 12, 2):
            out += [self._max3(i == 0, words[i], words[i + 1])]
        body = []
        carr = ["s[0:1]", "s[2:3]", "s[4:5]"]
        for g in range(2):
            js = range(3 * g, 3 * g + 3)
            for j in js:
                body += self._dbl(a[j], a[j])
            for j in js:
                body += ["v_sub_co_u32 v%d, %s, v%d, v%d" % (a[j], carr[j % 3], a[j], b[j])]
            for j in js:
                body += ["v_subb_co_u32 v%d, %s, v%d, v%d, %s" % (a[j] + 1, carr[j % 3], a[j] + 1, b[j] + 1, carr[j % 3])]
            # borrow: - EPS = + 1 - 2^32
            for j in js:
                body += ["v_cndmask_b32 v%d, 0, -1, %s" % (self.UH[0] if j % 3 == 0 else self.UH[1] if j % 3 == 1 else self.TMP, carr[j % 3])]
            for j in js:
                body += ["v_addc_co_u32 v%d, %s, v%d, 0, %s" % (a[j], carr[j % 3], a[j], carr[j % 3])]
            for j in js:
                m = self.UH[0] if j % 3 == 0 else self.UH[1] if j % 3 == 1 else self.TMP
                body += ["v_addc_co_u32 v%d, %s, v%d, v%d, %s" % (a[j] + 1, carr[j % 3], a[j] + 1, m, carr[j % 3])]
        # the guard must see the operands BEFORE they are overwritten: it runs first, the cold path falls into the body
        out += ["v_cmp_le_u32 vcc, %s, v%d" % (THRESH, self.G), "s_cbranch_vccnz L_cold%d_%%=" % site, "L_cont%d_%%=:" % site]
        out += body
        cold = ["L_cold%d_%%=:" % site]
        for j in range(6):
            cold += self._canon(a[j])
            cold += self._canon(b[j])
        cold += ["s_branch L_cont%d_%%=" % site]
        self.cold += cold


def names(prefix, regs, first=0):
    return {"%s%d" % (prefix, j): regs[j] for j in range(first, 6) if regs[j] is not None}


def build():
    gen = Gen()
    S, PB = gen.S, gen.PB
    YY, ZZ, YYYY, SH = S[0], S[1], S[2], S[3]
    M = S[0]                       # YY is dead when M is formed

    def sqr_pres(src, full):
        """2a (j = 1..5, j = 0 too when full), 7a (3..5; 1..5 when full), 14a (4, 5; 1..5 when full) in PB[0..]"""
        d = [None] * 6
        s = [None] * 6
        t = [None] * 6
        n = 0
        for j in range(0 if full else 1, 6):
            d[j] = PB[n]; n += 1
        for j in range(1 if full else 3, 6):
            s[j] = PB[n]; n += 1
        for j in range(1 if full else 4, 6):
            t[j] = PB[n]; n += 1
        gen.prescale(src, [(d[j], j) for j in range(6) if d[j] is not None],
                     [(s[j], j, 7) for j in range(6) if s[j] is not None],
                     [(t[j], s[j]) for j in range(6) if t[j] is not None])
        return d, s, t, n

    def sqr_regs(src, d, s, t):
        m = names("a", src)
        m.update(names("d", d))
        m.update(names("s", s))
        m.update(names("t", t))
        return m

    # 1. YY = Y^2
    d, s, t, _ = sqr_pres(YR, False)
    gen.block(g6.sqr_terms(), sqr_regs(YR, d, s, t), YY)
    # 2. ZZ = Z^2, Z3 = Y * (2Z)   (b = 2Z, 7b = 14Z)
    d, s, t, _ = sqr_pres(ZR, True)
    gen.block(g6.sqr_terms(), sqr_regs(ZR, d, s, t), ZZ)
    m = names("a", YR)
    m.update(names("b", d))
    m.update(names("s", t))
    gen.block(g6.mul_terms(), m, ZR)
    # 3. YYYY = YY^2, S' = X * (2 YY)
    d, s, t, _ = sqr_pres(YY, True)
    gen.block(g6.sqr_terms(), sqr_regs(YY, d, s, t), YYYY)
    m = names("a", XR)
    m.update(names("b", d))
    m.update(names("s", t))
    gen.block(g6.mul_terms(), m, SH)
    # 4. M = ZZ^2 + 3 X^2: the squaring's terms twice, the second time on X with 3X, 6X, 21X, 42X
    d, s, t, n = sqr_pres(ZZ, False)
    x3 = [PB[n + j] for j in range(3)] + [None] * 3
    x6 = [None] + [PB[n + 3 + j] for j in range(5)]
    x21 = [None] * 3 + [PB[n + 8 + j] for j in range(3)]
    x42 = [None] * 4 + [PB[n + 11 + j] for j in range(2)]
    # 6X = 2 (3X) needs 3X for j = 1..5 as well: 3X[3..5] live in the 21X / 42X slots' neighbours -- simply compute 6X
    # as a constant multiple where 3X is not kept
    mulc = [(x3[j], j, 3) for j in range(3)] + [(x6[j], j, 6) for j in range(3, 6)] + [(x21[j], j, 21) for j in range(3, 6)]
    dbl_of = [(x6[j], x3[j]) for j in range(1, 3)] + [(x42[j], x21[j]) for j in range(4, 6)]
    gen.prescale(XR, [], mulc, dbl_of)
    terms = g6.sqr_terms()
    rename = {"a": "x", "d": "e", "s": "f", "t": "g"}
    terms2 = [[(x.replace("a", "x"), rename[y[0]] + y[1:]) for x, y in tk] for tk in terms]
    m = sqr_regs(ZZ, d, s, t)
    m.update(names("x", XR))
    # diagonal direct terms use 3X (j <= 2), cross direct 6X, diagonal wrapped 21X, cross wrapped 42X
    m.update({"x%d" % j: XR[j] for j in range(6)})
    m.update({"e%d" % j: x6[j] for j in range(1, 6)})
    m.update({"f%d" % j: x21[j] for j in range(3, 6)})
    m.update({"g%d" % j: x42[j] for j in range(4, 6)})
    # the diagonal direct term of sqr_terms is (a_i, a_i): on X it must be (x_i, 3 x_i)
    terms2 = [[(x, ("h" + y[1:]) if y[0] == "x" else y) for x, y in tk] for tk in terms2]
    m.update({"h%d" % j: x3[j] for j in range(3)})
    gen.block([a + b for a, b in zip(terms, terms2)], m, M)
    # 5. X3 = M^2 - 4 S'  (into X), with 7M for all j: shared with the product of step 6
    dM = [None] + [PB[j] for j in range(5)]
    sM = [None] + [PB[5 + j] for j in range(5)]
    tM = [None] * 4 + [PB[10], PB[11]]
    gen.prescale(M, [(dM[j], j) for j in range(1, 6)], [(sM[j], j, 7) for j in range(1, 6)], [(tM[j], sM[j]) for j in range(4, 6)])
    gen.block(g6.sqr_terms(), sqr_regs(M, dM, sM, tM), XR, extras=[(-1, 4, "x")], extra_regs={"x": SH})
    # 6. W = 2 S' - X3 (in place of S'), Y3 = W * M - 8 YYYY  (into Y)
    gen.dbl_sub(SH, XR)
    m = names("a", SH)
    m.update(names("b", M))
    m.update(names("s", sM))
    gen.block(g6.mul_terms(), m, YR, extras=[(-1, 8, "x")], extra_regs={"x": YYYY})
    return gen


def emit():
    gen = build()
    pre = ["v_mov_b32 v%d, 0" % r for r in gen.zero_regs] + ["s_mov_b32 s20, %[n]", "L_top_%=:"]
    post = ["s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", "s_cbranch_scc1 L_top_%=", "s_branch L_end_%="]
    body = pre + gen.main + post + gen.cold + ["L_end_%=:"]
    used = set(gen.al.used) | set(g6.POOL[:g6.N_FIXED + 4])
    for ln in body:
        for r in re.findall(r"\bv(\d+)\b", ln):
            assert int(r) in used or int(r) in _pin_regs, ln
        for lo, hi in re.findall(r"\bv\[(\d+):(\d+)\]", ln):
            assert int(lo) % 2 == 0 and int(hi) == int(lo) + 1, ln
            assert int(lo) in used or int(lo) in _pin_regs, ln
    n_valu = sum(1 for ln in gen.main if ln.startswith("v_"))
    n_mad = sum(1 for ln in gen.main if ln.startswith("v_mad"))
    n_nop = sum(1 for ln in gen.main if ln.startswith("s_nop"))
    out = ["// generated by tools/gen_jac_asm.py -- do not edit (see that file for the design notes)",
           "// (X, Y, Z) <- [2^n](X, Y, Z), n >= 1, Jacobian, a = 1 (loose in / loose out; Z == 0 stays Z == 0).",
           "// One doubling: %d VALU instructions (%d multiplies) + %d s_nop on the hot path." % (n_valu, n_mad, n_nop),
           "SSA_DEV void jac_dbl_n_asm(u64 (&X)[6], u64 (&Y)[6], u64 (&Z)[6], u32 n) {", "    asm volatile("]
    for i, ln in enumerate(body):
        out.append('        "%s%s"' % (ln, "\\n\\t" if i + 1 < len(body) else ""))
    ops = []
    for nm, regs in (("X", XR), ("Y", YR), ("Z", ZR)):
        for j in range(6):
            ops.append('"+{v[%d:%d]}"(%s[%d])' % (regs[j], regs[j] + 1, nm, j))
    out.append("        : " + ",\n          ".join(ops))
    out.append('        : [n] "s"(n)')
    clob = ['"v%d"' % r for r in sorted(used)] + ['"s%d"' % i for i in range(21)] + ['"vcc"', '"scc"']
    out.append("        : " + ", ".join(clob) + ");")
    out.append("}")
    print("doubling: %d VALU (%d multiplies), %d s_nop; %d fixed VGPRs + 36 pinned; %d cold-path lines"
          % (n_valu, n_mad, n_nop, len(used), len(gen.cold)))
    return out


def main():
    out = emit()
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "schnorr-sig_amd", "csrc", "jac_asm.inc")
    with open(path, "w") as fh:
        fh.write("\n".join(out) + "\n")
    print("wrote", path)


if __name__ == "__main__":
    main()
