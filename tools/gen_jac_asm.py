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
# between a VALU compare into VCC and the branch on it: the compiler's own idiom (the branch then reads a VCC written by
# the scalar unit, whose read of the VALU result is interlocked; lanes outside EXEC cannot take the wave to a cold path)
VCC_ACTIVE = "s_and_b64 vcc, exec, vcc"


class Gen:
    def __init__(self, n_slots=4, n_pb=23, n_pinned_in=0, tag=""):
        self.al = Alloc()
        self.tag = tag          # label prefix: two generators' code may share one asm statement
        self.main = []          # the loop body
        self.cold = []          # cold paths, after the loop
        self.nsite = 0
        self.nred = 0           # groups of three reductions (one cold path each: gen_f6_asm.reduce3)
        a = self.al
        self.IN = [a.free.pop() for _ in range(n_pinned_in)][::-1]    # further pinned operands (highest free pairs)
        self.S = [a.fp6() for _ in range(n_slots)]    # Fp6 temporaries
        self.PB = [a.pair() for _ in range(n_pb)]     # pre-scaled operands
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
        out += ["v_cmp_le_u32 vcc, %s, v%d" % (THRESH, self.G), VCC_ACTIVE, "s_cbranch_vccnz L_%scold%d_%%=" % (self.tag, site),
                "L_%scont%d_%%=:" % (self.tag, site)]
        # cold path
        cold = ["L_%scold%d_%%=:" % (self.tag, site)]
        for _, j in dbl:
            cold += self._canon(src[j])
        for dst, j, c in mulc:
            cold += self._mulc_exact(dst, src[j], c)
        for _, sp in dbl_of:          # an exact multiple may have any high word: canonicalise before the short doubling
            cold += self._canon(sp)
        cold += body
        cold += ["s_branch L_%scont%d_%%=" % (self.tag, site)]
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
            hot, cold = g6.reduce3(accs[3 * g:3 * g + 3], outs, "%sred%d" % (self.tag, self.nred))
            self.nred += 1
            lines += hot
            self.cold += cold
        m = dict(regs)
        for _, _, prefix in extras:
            for k in range(6):
                m["%s%d" % (prefix, k)] = extra_regs[prefix][k]

        def sub(mo):
            nm, half = mo.group(1), mo.group(2)
            return "v%d" % (m[nm] + (1 if half == "h" else 0))
        self.main += [re.sub(r"%\[(\w+?)([lh])\]", sub, ln) for ln in lines]
        self.nfair = getattr(self, "nfair", 0) + 1
        self.main += fair_now("%sq%d" % (self.tag, self.nfair))

    # ---- dst = (2)a - b coefficient-wise (a = None: -b); one guard: the doubling needs a's high word below all ones,
    # the second borrow of the subtraction b's ----
    def lin_sub(self, dst, a, b, dbl_a=False):
        site = self.nsite
        self.nsite += 1
        out = self.main
        words = ["v%d" % (b[j] + 1) for j in range(6)]
        if a is not None and dbl_a:
            words += ["v%d" % (a[j] + 1) for j in range(6)]
        for i in range(0, len(words), 2):
            out += [self._max3(i == 0, words[i], words[i + 1])]
        body = []
        carr = ["s[0:1]", "s[2:3]", "s[4:5]"]
        msk = [self.UH[0], self.UH[1], self.TMP]
        for g in range(2):
            js = range(3 * g, 3 * g + 3)
            src = {}
            for j in js:
                if a is None:
                    src[j] = ("0", "0")
                elif dbl_a:
                    body += self._dbl(dst[j], a[j])
                    src[j] = ("v%d" % dst[j], "v%d" % (dst[j] + 1))
                else:
                    src[j] = ("v%d" % a[j], "v%d" % (a[j] + 1))
            for j in js:
                body += ["v_sub_co_u32 v%d, %s, %s, v%d" % (dst[j], carr[j % 3], src[j][0], b[j])]
            for j in js:
                body += ["v_subb_co_u32 v%d, %s, %s, v%d, %s" % (dst[j] + 1, carr[j % 3], src[j][1], b[j] + 1, carr[j % 3])]
            # borrow: - EPS = + 1 - 2^32
            for j in js:
                body += ["v_cndmask_b32 v%d, 0, -1, %s" % (msk[j % 3], carr[j % 3])]
            for j in js:
                body += ["v_addc_co_u32 v%d, %s, v%d, 0, %s" % (dst[j], carr[j % 3], dst[j], carr[j % 3])]
            for j in js:
                body += ["v_addc_co_u32 v%d, %s, v%d, v%d, %s" % (dst[j] + 1, carr[j % 3], dst[j] + 1, msk[j % 3], carr[j % 3])]
        # the guard sees the operands BEFORE they are overwritten; the cold path canonicalises them and rejoins
        out += ["v_cmp_le_u32 vcc, %s, v%d" % (THRESH, self.G), VCC_ACTIVE, "s_cbranch_vccnz L_%scold%d_%%=" % (self.tag, site),
                "L_%scont%d_%%=:" % (self.tag, site)]
        out += body
        cold = ["L_%scold%d_%%=:" % (self.tag, site)]
        for j in range(6):
            if a is not None and dbl_a:
                cold += self._canon(a[j])
            cold += self._canon(b[j])
        cold += ["s_branch L_%scont%d_%%=" % (self.tag, site)]
        self.cold += cold

    def dbl_sub(self, a, b):
        self.lin_sub(a, a, b, dbl_a=True)

    # ---- x == 0 (mod p) of one loose 64-bit value, as a word that is zero exactly then ----
    def zero_word(self, dst, a):
        t, u = self.UH[0], self.UH[1]
        return ["v_or_b32 v%d, v%d, v%d" % (dst, a, a + 1),            # 0 iff a == 0
                "v_xor_b32 v%d, 1, v%d" % (t, a), "v_not_b32 v%d, v%d" % (u, a + 1),
                "v_or_b32 v%d, v%d, v%d" % (t, t, u),                    # 0 iff a == p
                "v_min_u32 v%d, v%d, v%d" % (dst, dst, t)]


def names(prefix, regs, first=0):
    return {"%s%d" % (prefix, j): regs[j] for j in range(first, 6) if regs[j] is not None}


def pres(gen, src, d_from, s_from, t_from, base=0):
    """pre-scaled operands of src in gen.PB[base..]: d[j] = 2 src[j] (j >= d_from), s[j] = 7 src[j] (j >= s_from),
    t[j] = 14 src[j] (j >= t_from >= s_from); None = not needed.  Returns (d, s, t, next free PB index)."""
    PB = gen.PB
    d, s, t = [None] * 6, [None] * 6, [None] * 6
    n = base
    for arr, frm in ((d, d_from), (s, s_from), (t, t_from)):
        if frm is None:
            continue
        for j in range(frm, 6):
            arr[j] = PB[n]
            n += 1
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


def mul_regs(a, b, b7):
    m = names("a", a)
    m.update(names("b", b))
    m.update(names("s", b7))
    return m


def build_dbl(tag=""):
    gen = Gen(tag=tag)
    S, PB = gen.S, gen.PB
    YY, ZZ, YYYY, SH = S[0], S[1], S[2], S[3]
    M = S[0]                       # YY is dead when M is formed
    # 1. YY = Y^2
    d, s, t, _ = pres(gen, YR, 1, 3, 4)
    gen.block(g6.sqr_terms(), sqr_regs(YR, d, s, t), YY)
    # 2. ZZ = Z^2, Z3 = Y * (2Z)   (b = 2Z, 7b = 14Z)
    d, s, t, _ = pres(gen, ZR, 0, 1, 1)
    gen.block(g6.sqr_terms(), sqr_regs(ZR, d, s, t), ZZ)
    gen.block(g6.mul_terms(), mul_regs(YR, d, t), ZR)
    # 3. YYYY = YY^2, S' = X * (2 YY)
    d, s, t, _ = pres(gen, YY, 0, 1, 1)
    gen.block(g6.sqr_terms(), sqr_regs(YY, d, s, t), YYYY)
    gen.block(g6.mul_terms(), mul_regs(XR, d, t), SH)
    # 4. M = ZZ^2 + 3 X^2: the squaring's terms twice, the second time on X with 3X (diagonal), 6X (cross), 21X
    #    (wrapped diagonal), 42X (wrapped cross)
    d, s, t, n = pres(gen, ZZ, 1, 3, 4)
    x3 = [PB[n + j] for j in range(3)] + [None] * 3
    x6 = [None] + [PB[n + 3 + j] for j in range(5)]
    x21 = [None] * 3 + [PB[n + 8 + j] for j in range(3)]
    x42 = [None] * 4 + [PB[n + 11 + j] for j in range(2)]
    mulc = [(x3[j], j, 3) for j in range(3)] + [(x6[j], j, 6) for j in range(3, 6)] + [(x21[j], j, 21) for j in range(3, 6)]
    dbl_of = [(x6[j], x3[j]) for j in range(1, 3)] + [(x42[j], x21[j]) for j in range(4, 6)]
    gen.prescale(XR, [], mulc, dbl_of)
    terms = g6.sqr_terms()
    second = {"a": "h", "d": "e", "s": "f", "t": "g"}
    terms2 = [[("x" + x[1:], second[y[0]] + y[1:]) for x, y in tk] for tk in terms]
    m = sqr_regs(ZZ, d, s, t)
    m.update(names("x", XR))
    m.update(names("h", x3))
    m.update(names("e", x6))
    m.update(names("f", x21))
    m.update(names("g", x42))
    gen.block([a + b for a, b in zip(terms, terms2)], m, M)
    # 5. X3 = M^2 - 4 S'  (into X), with 7M for all j: shared with the product of step 6
    dM, sM, tM, _ = pres(gen, M, 1, 1, 4)
    gen.block(g6.sqr_terms(), sqr_regs(M, dM, sM, tM), XR, extras=[(-1, 4, "x")], extra_regs={"x": SH})
    # 6. W = 2 S' - X3 (in place of S'), Y3 = W * M - 8 YYYY  (into Y)
    gen.dbl_sub(SH, XR)
    gen.block(g6.mul_terms(), mul_regs(SH, M, sM), YR, extras=[(-1, 8, "x")], extra_regs={"x": YYYY})
    return gen


# First instruction of every statement.  The compiler keeps values in registers these statements clobber and reloads them from
# scratch behind each statement; it waits for such a reload where the VALUE is next used -- not in front of an inline asm
# that merely clobbers the register (measured: ssa_k_sign entered the gathering addition with four reloads in flight, which
# then landed in the statement's temporaries: wrong signatures on ~7 % of the waves, different ones from run to run).  The
# statements whose operands the compiler loads itself were shielded by its wait for those operands; the gathering ones are
# not.  So: nothing of the compiler's may be in flight when a statement starts.
ENTRY_WAIT = "s_waitcnt vmcnt(0)"
Q_WAIT = "s_waitcnt vmcnt(2)"       # late_q: the statement's own six loads of (x2, y2) have landed; its two touches may be out


def build_madd(tag="", late_q=False):
    """(X, Y, Z) += (x2, y2), the generic path of the mixed addition (7M + 4S as 8 products + 3 squares):
        Z1Z1 = Z^2   T = y2 Z   H = x2 Z1Z1 - X   R = T Z1Z1 - Y   HH = H^2   Z3 = Z H   HHH = HH H   V = X HH
        X3 = R^2 - HHH - 2V      Y3 = (V - X3) R + HHH (-Y)
    The exceptional inputs (Z == 0, (x2, y2) == (0, 0), H == 0) are NOT handled here: a necessary condition of each
    (first coefficient zero mod p) is tested before anything is written, and the block then leaves the point untouched
    and reports 0 -- the caller runs the compiled, exact jac_madd for that wave.  Honest inputs never get there.
    late_q (the statement that gathers (x2, y2) itself, emit_madd_gather): everything that does not need the second
    point comes first -- Z1Z1 = Z^2 and ZZZ = Z1Z1 Z, 57 products --, then the wait for the loads (Q_WAIT), then
    H = x2 Z1Z1 - X and R = y2 ZZZ - Y: the same eight products and three squares, one more pre-scaled operand."""
    gen = Gen(n_slots=5, n_pb=17, n_pinned_in=12, tag=tag)
    S = gen.S
    x2, y2 = gen.IN[0:6], gen.IN[6:12]
    Z1Z1, T, H, R, HH = S[0], S[1], S[2], S[3], S[4]
    Z3, HHH, V, NY = S[0], S[1], S[4], S[2]
    f, w = gen.G, gen.TMP
    if late_q:
        gen.main += gen.zero_word(w, ZR[0])
        gen.main += ["v_cmp_eq_u32 vcc, 0, v%d" % w, VCC_ACTIVE, "s_cbranch_vccnz L_bail_%="]
        # 1. Z1Z1 = Z^2, ZZZ = Z1Z1 * Z (in T's slot)
        d, s, t, _ = pres(gen, ZR, 1, 1, 4)
        gen.block(g6.sqr_terms(), sqr_regs(ZR, d, s, t), Z1Z1)
        gen.block(g6.mul_terms(), mul_regs(Z1Z1, ZR, s), T)
        gen.main += [Q_WAIT]
        gen.main += gen.zero_word(w, x2[0])
        gen.main += ["v_cmp_eq_u32 vcc, 0, v%d" % w, VCC_ACTIVE, "s_cbranch_vccnz L_bail_%="]
        # 2. H = x2 Z1Z1 - X, R = y2 ZZZ - Y
        _, s, _, _ = pres(gen, Z1Z1, None, 1, None)
        gen.block(g6.mul_terms(), mul_regs(x2, Z1Z1, s), H, extras=[(-1, 1, "x")], extra_regs={"x": XR})
        _, s, _, _ = pres(gen, T, None, 1, None)
        gen.block(g6.mul_terms(), mul_regs(y2, T, s), R, extras=[(-1, 1, "x")], extra_regs={"x": YR})
    else:
        # exceptional inputs, part 1 (G doubles as the flag word here; the guards below rewrite it afterwards)
        gen.main += gen.zero_word(w, ZR[0])
        gen.main += ["v_mov_b32 v%d, v%d" % (gen.MASK, w)]
        gen.main += gen.zero_word(w, x2[0])
        gen.main += ["v_min_u32 v%d, v%d, v%d" % (gen.MASK, gen.MASK, w), "v_cmp_eq_u32 vcc, 0, v%d" % gen.MASK, VCC_ACTIVE,
                     "s_cbranch_vccnz L_bail_%="]
        # 1. Z1Z1 = Z^2, T = y2 * Z
        d, s, t, _ = pres(gen, ZR, 1, 1, 4)
        gen.block(g6.sqr_terms(), sqr_regs(ZR, d, s, t), Z1Z1)
        gen.block(g6.mul_terms(), mul_regs(y2, ZR, s), T)
        # 2. H = x2 Z1Z1 - X, R = T Z1Z1 - Y
        _, s, _, _ = pres(gen, Z1Z1, None, 1, None)
        gen.block(g6.mul_terms(), mul_regs(x2, Z1Z1, s), H, extras=[(-1, 1, "x")], extra_regs={"x": XR})
        gen.block(g6.mul_terms(), mul_regs(T, Z1Z1, s), R, extras=[(-1, 1, "x")], extra_regs={"x": YR})
    # exceptional inputs, part 2: H == 0 (P == +-Q)
    gen.main += gen.zero_word(w, H[0])
    gen.main += ["v_cmp_eq_u32 vcc, 0, v%d" % w, VCC_ACTIVE, "s_cbranch_vccnz L_bail_%="]
    # 3. HH = H^2, Z3 = Z H, HHH = HH H
    d, s, t, _ = pres(gen, H, 1, 1, 4)
    gen.block(g6.sqr_terms(), sqr_regs(H, d, s, t), HH)
    gen.block(g6.mul_terms(), mul_regs(ZR, H, s), Z3)
    gen.block(g6.mul_terms(), mul_regs(HH, H, s), HHH)
    # 4. V = X HH (in place of HH: the product reads b = HH, so it goes to H's slot first)
    _, s, _, _ = pres(gen, HH, None, 1, None)
    gen.block(g6.mul_terms(), mul_regs(XR, HH, s), H)
    V = H
    # 5. X3 = R^2 - HHH - 2V  (into X)
    d, sR, t, n = pres(gen, R, 1, 1, 4)
    gen.block(g6.sqr_terms(), sqr_regs(R, d, sR, t), XR, extras=[(-1, 1, "x"), (-1, 2, "y")], extra_regs={"x": HHH, "y": V})
    # 6. W = V - X3 (in place), NY = -Y, Y3 = W R + HHH NY  (into Y)
    gen.lin_sub(V, V, XR)
    NY = S[4]
    gen.lin_sub(NY, None, YR)
    _, sN, _, _ = pres(gen, NY, None, 1, None, base=n)
    m = mul_regs(V, R, sR)
    m.update(names("c", HHH))
    m.update(names("e", NY))
    m.update(names("t", sN))
    gen.block(g6.mul2_terms(), m, YR)
    # 7. Z3 into Z
    for j in range(6):
        gen.main += ["v_mov_b32 v%d, v%d" % (ZR[j], Z3[j]), "v_mov_b32 v%d, v%d" % (ZR[j] + 1, Z3[j] + 1)]
    return gen


# ---- fair turns for the waves of a SIMD: an experiment of round 4, OFF by default (profiles/r04/wave_timeline.txt) ----
# The sequencer serves the oldest wave first: of the two waves of a SIMD the one in slot 0 takes ~70 % of the issue slots,
# the two slot chains run in anti-phase from the first generation on, and the last wave of every SIMD finishes the kernel
# ALONE (a lone wave issues every ~6 cycles instead of every 4): 0.7 ms of a 26.6 ms launch.  The user priority (s_setprio)
# ranks above the age, so the waves could take TURNS at being favoured: a time slice of the 100 MHz counter every wave
# reads (s_memrealtime), offset by the wave's slot on its SIMD.  In tools/arb_probe two waves of pure multiply-add code
# finish after 1.56 / 2.98 ms without it and after 2.60 / 2.60 ms with it (kernel 3.06 -> 2.63 ms).  In ssa_k_verify, with
# the priority refreshed once per doubling (SSA_GEN_FAIR=1: the counter is read at the top of a doubling and used at its
# end) and in the compiled phases, the first generation's spread shrinks from 2.19-3.75 to 2.73-3.78 ms and the kernel
# time does not move (26.70 against 26.65 ms); ssa_k_hash with four rotating priorities gets 4.6 % SLOWER.  Whatever
# favours slot 0 in a kernel whose code does not fit the instruction buffers is not the issue arbitration alone.
FAIR = int(os.environ.get("SSA_GEN_FAIR", "0"))      # 0 off, 1 once per doubling, 2 after every Fp6 block (~400 instructions)
FAIR_SHIFT = int(os.environ.get("SSA_GEN_FAIR_SHIFT", "13"))


def fair_read():
    return ["s_memrealtime s[24:25]"] if FAIR else []


def fair_now(tag):
    """read the clock and set the priority on the spot (FAIR = 2: between two Fp6 blocks)"""
    return ["s_memrealtime s[24:25]"] + fair_set(tag) if FAIR >= 2 else []


def fair_set(tag):
    if not FAIR:
        return []
    return ["s_waitcnt lgkmcnt(0)", "s_lshr_b32 s24, s24, %d" % FAIR_SHIFT, "s_getreg_b32 s21, hwreg(HW_REG_HW_ID, 0, 1)",
            "s_xor_b32 s24, s24, s21", "s_bitcmp1_b32 s24, 0", "s_cbranch_scc1 L_%shi_%%=" % tag, "s_setprio 0",
            "s_branch L_%sset_%%=" % tag, "L_%shi_%%=:" % tag, "s_setprio 1", "L_%sset_%%=:" % tag]


def check_and_stats(gen, body, pinned):
    used = set(gen.al.used) | set(g6.POOL[:g6.N_FIXED + 4])
    for ln in body:
        for r in re.findall(r"\bv(\d+)\b", ln):
            assert int(r) in used or int(r) in pinned, ln
        for lo, hi in re.findall(r"\bv\[(\d+):(\d+)\]", ln):
            assert int(lo) % 2 == 0 and int(hi) == int(lo) + 1, ln
            assert int(lo) in used or int(lo) in pinned, ln
    n_valu = sum(1 for ln in gen.main if ln.startswith("v_"))
    n_mad = sum(1 for ln in gen.main if ln.startswith("v_mad"))
    n_nop = sum(1 for ln in gen.main if ln.startswith("s_nop"))
    return used, n_valu, n_mad, n_nop


def asm_lines(out, body):
    for i, ln in enumerate(body):
        out.append('        "%s%s"' % (ln, "\\n\\t" if i + 1 < len(body) else ""))


def emit_dbl():
    gen = build_dbl()
    pre = [ENTRY_WAIT] + ["v_mov_b32 v%d, 0" % r for r in gen.zero_regs] + ["s_mov_b32 s20, %[n]", "L_top_%=:"] + fair_read()
    post = fair_set("f") + ["s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", "s_cbranch_scc1 L_top_%=", "s_branch L_end_%="]
    body = pre + gen.main + post + gen.cold + ["L_end_%=:"]
    used, n_valu, n_mad, n_nop = check_and_stats(gen, body, _pin_regs)
    out = ["// (X, Y, Z) <- [2^n](X, Y, Z), n >= 1, Jacobian, a = 1 (loose in / loose out; Z == 0 stays Z == 0).",
           "// One doubling: %d VALU instructions (%d multiplies) + %d s_nop on the hot path." % (n_valu, n_mad, n_nop),
           "SSA_DEV void jac_dbl_n_asm(u64 (&X)[6], u64 (&Y)[6], u64 (&Z)[6], u32 n) {", "    asm volatile("]
    asm_lines(out, body)
    ops = []
    for nm, regs in (("X", XR), ("Y", YR), ("Z", ZR)):
        for j in range(6):
            ops.append('"+{v[%d:%d]}"(%s[%d])' % (regs[j], regs[j] + 1, nm, j))
    out.append("        : " + ",\n          ".join(ops))
    out.append('        : [n] "s"(__builtin_amdgcn_readfirstlane(n))      // wave-uniform by contract')
    clob = ['"v%d"' % r for r in sorted(used)] + ['"s%d"' % i for i in range(26 if FAIR else 21)] + ['"vcc"', '"scc"']
    out.append("        : " + ", ".join(clob) + ");")
    out.append("}")
    print("doubling: %d VALU (%d multiplies), %d s_nop; %d fixed VGPRs + 36 pinned; %d cold-path lines"
          % (n_valu, n_mad, n_nop, len(used), len(gen.cold)))
    return out


def emit_madd():
    gen = build_madd()
    pinned = _pin_regs | set(r for p in gen.IN for r in (p, p + 1))
    pre = [ENTRY_WAIT] + ["v_mov_b32 v%d, 0" % r for r in gen.zero_regs] + ["v_mov_b32 %[ok], 1"]
    post = ["s_branch L_end_%="]
    bail = ["L_bail_%=:", "v_mov_b32 %[ok], 0"]
    body = pre + gen.main + post + gen.cold + bail + ["L_end_%=:"]
    used, n_valu, n_mad, n_nop = check_and_stats(gen, [ln for ln in body if "%[ok]" not in ln], pinned)
    out = ["// (X, Y, Z) += (x2, y2): the generic path of the mixed addition; returns 0 with the point untouched when an",
           "// exceptional input is possible (Z, x2 or H with a first coefficient = 0 mod p): the caller then runs jac_madd.",
           "// %d VALU instructions (%d multiplies) + %d s_nop on the hot path." % (n_valu, n_mad, n_nop),
           "SSA_DEV u32 jac_madd_asm(u64 (&X)[6], u64 (&Y)[6], u64 (&Z)[6], const u64 (&x2)[6], const u64 (&y2)[6]) {",
           "    u32 ok;", "    asm volatile("]
    asm_lines(out, body)
    ops = ['[ok] "=&v"(ok)']
    for nm, regs in (("X", XR), ("Y", YR), ("Z", ZR)):
        for j in range(6):
            ops.append('"+{v[%d:%d]}"(%s[%d])' % (regs[j], regs[j] + 1, nm, j))
    out.append("        : " + ",\n          ".join(ops))
    ins = []
    for nm, regs in (("x2", gen.IN[0:6]), ("y2", gen.IN[6:12])):
        for j in range(6):
            ins.append('"{v[%d:%d]}"(%s[%d])' % (regs[j], regs[j] + 1, nm, j))
    out.append("        : " + ",\n          ".join(ins))
    clob = ['"v%d"' % r for r in sorted(used)] + ['"s%d"' % i for i in range(26 if FAIR else 21)] + ['"vcc"', '"scc"']
    out.append("        : " + ", ".join(clob) + ");")
    out += ["    return ok;", "}"]
    print("mixed addition: %d VALU (%d multiplies), %d s_nop; %d fixed VGPRs + 60 pinned; %d cold-path lines"
          % (n_valu, n_mad, n_nop, len(used), len(gen.cold)))
    return out


PF = 117         # the touches' landing register: outside every register the statements name
KEEP = 38        # v[38:39]: outside every register the window / gather statements name


def emit_madd_gather():
    """the mixed addition that gathers its own second point and touches the NEXT one (comb tables: 96-byte entries at random
    places of a table far larger than the caches -- 17.7 GB for G, 100 MB per key --, one addition after the other).  The
    statement issues the six loads of (x2, y2) from `row` and, behind them, two one-word loads of the first and last word
    of the entry at `next` (whatever lines it lies in are then on their way into the L2 while this addition runs); the
    additions' own loads are awaited after the 57 products that do not need them (loads return in order: vmcnt(2)), the
    touches at the end of the statement, long landed."""
    gen = build_madd("g", late_q=True)
    in_regs = set(r for p in gen.IN for r in (p, p + 1))
    assert gen.IN == list(range(gen.IN[0], gen.IN[0] + 24, 2)) and gen.IN[0] % 4 == 0
    pinned = _pin_regs | in_regs
    loads = ["global_load_dwordx4 v[%d:%d], %%[row], off%s" % (gen.IN[0] + 4 * k, gen.IN[0] + 4 * k + 3, " offset:%d" % (16 * k) if k else "")
             for k in range(6)]
    loads += ["global_load_dword v%d, %%[next], off" % PF, "global_load_dword v%d, %%[next], off offset:92" % PF]
    pre = [ENTRY_WAIT] + loads + ["v_mov_b32 v%d, 0" % r for r in gen.zero_regs] + ["v_mov_b32 %[ok], 1"]
    post = ["s_branch L_end_%="]
    bail = ["L_bail_%=:", "v_mov_b32 %[ok], 0"]
    body = pre + gen.main + post + gen.cold + bail + ["L_end_%=:", "s_waitcnt vmcnt(0)"]
    assert body.count(Q_WAIT) == 1
    used, n_valu, n_mad, n_nop = check_and_stats(gen, [ln for ln in body if "%[" not in ln], pinned | {PF})
    used = used | in_regs | {PF}
    assert PF not in _pin_regs and PF not in set(gen.al.used)
    out = ["// (X, Y, Z) += the point at row[0..11], gathered by the statement itself; `next`: the entry the NEXT addition will",
           "// want (its first and last word are loaded and dropped: a prefetch).  Returns 0 with the point untouched when an",
           "// exceptional input is possible: the caller then loads the point and runs jac_madd.",
           "// %d VALU instructions (%d multiplies) + %d s_nop on the hot path." % (n_valu, n_mad, n_nop),
           "// keep: as in jac_window_asm (the table's base, handed through in v[%d:%d])." % (KEEP, KEEP + 1),
           "SSA_DEV u32 jac_madd_gather_asm(u64 (&X)[6], u64 (&Y)[6], u64 (&Z)[6], const u64 *row, const u64 *next, const u64 *&keep) {",
           "    u32 ok;", "    asm volatile("]
    asm_lines(out, body)
    ops = ['[ok] "=&v"(ok)']
    for nm, regs in (("X", XR), ("Y", YR), ("Z", ZR)):
        for j in range(6):
            ops.append('"+{v[%d:%d]}"(%s[%d])' % (regs[j], regs[j] + 1, nm, j))
    assert KEEP not in used and KEEP + 1 not in used and KEEP not in pinned and KEEP + 1 not in pinned
    ops.append('"+{v[%d:%d]}"(keep)' % (KEEP, KEEP + 1))
    out.append("        : " + ",\n          ".join(ops))
    out.append('        : [row] "v"(row), [next] "v"(next)')
    clob = ['"v%d"' % r for r in sorted(used)] + ['"s%d"' % i for i in range(26 if FAIR else 21)] + ['"vcc"', '"scc"', '"memory"']
    out.append("        : " + ", ".join(clob) + ");")
    out += ["    return ok;", "}"]
    print("mixed addition with its gather: %d VALU (%d multiplies), %d s_nop; %d VGPRs + 36 pinned; %d cold-path lines"
          % (n_valu, n_mad, n_nop, len(used), len(gen.cold)))
    return out




def emit_window():
    """n doublings, then the mixed addition on the lanes whose `act` word is non-zero (EXEC narrowed inside the
    statement): one statement per ladder window, the point never leaves its registers in between.
    Round 5: the statement GATHERS its own table entry.  It takes the lane's row address, issues the six 16-byte loads of
    (x2, y2) as its first instructions and waits for them behind the doublings -- 13 000 instructions later --, where the
    compiled loop waited for the whole HBM latency of a lane-private line in front of every window
    (profiles/r05/gather_ab.txt).  The wait stands in front of the EXEC narrowing, on every path out of the statement: the
    loaded registers are the statement's clobbers, the compiler may reuse them right behind it."""
    gd, gm = build_dbl("d"), build_madd("m")
    in_regs = set(r for p in gm.IN for r in (p, p + 1))
    assert not (set(gd.al.used) & in_regs)
    assert gm.IN == list(range(gm.IN[0], gm.IN[0] + 24, 2)) and gm.IN[0] % 4 == 0      # x2, y2: 24 consecutive VGPRs
    pinned = _pin_regs | in_regs
    loads = ["global_load_dwordx4 v[%d:%d], %%[row], off%s" % (gm.IN[0] + 4 * k, gm.IN[0] + 4 * k + 3, " offset:%d" % (16 * k) if k else "")
             for k in range(6)]
    pre = [ENTRY_WAIT] + loads + ["v_mov_b32 v%d, 0" % r for r in gd.zero_regs] + ["v_mov_b32 %[ok], 1", "s_mov_b32 s20, %[n]", "L_top_%=:"] + fair_read()
    loop_end = fair_set("f") + ["s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", "s_cbranch_scc1 L_top_%=", "s_waitcnt vmcnt(0)"]
    narrow = ["v_cmp_ne_u32 vcc, 0, %[act]", "s_and_saveexec_b64 s[22:23], vcc", "s_cbranch_execz L_skip_%="]
    zero_m = ["v_mov_b32 v%d, 0" % r for r in gm.zero_regs if r not in gd.zero_regs]
    skip = ["L_skip_%=:", "s_mov_b64 exec, s[22:23]", "s_branch L_end_%="]
    bail = ["L_bail_%=:", "v_mov_b32 %[ok], 0", "s_branch L_skip_%="]
    body = pre + gd.main + loop_end + narrow + zero_m + gm.main + skip + gd.cold + gm.cold + bail + ["L_end_%=:"]
    plain = [ln for ln in body if "%[" not in ln]
    used_d, nv_d, nm_d, nn_d = check_and_stats(gd, [ln for ln in plain], pinned | set(gm.al.used) | set(g6.POOL[:g6.N_FIXED + 4]))
    used = used_d | set(gm.al.used) | in_regs
    out = ["// n doublings, then (X, Y, Z) += (x2, y2) on the lanes with act != 0: one ladder window as ONE statement.",
           "// (x2, y2) = row[0..11]: the statement loads them itself, under the doublings (row: this lane's table entry).",
           "// Returns 0 when the addition met a possible exceptional input on some lane (the doublings are done, the addition",
           "// is not: the caller runs the compiled jac_madd on the lanes with act != 0).",
           "// keep: a pointer the caller wants to find in a REGISTER behind the statement (the ladder's table base: the",
           "// compiler otherwise parks it in scratch and waits for the reload in front of every window); pinned to",
           "// v[%d:%d], which the statement does not touch." % (KEEP, KEEP + 1),
           "SSA_DEV u32 jac_window_asm(u64 (&X)[6], u64 (&Y)[6], u64 (&Z)[6], const u64 *row, u32 act, u32 n, const u64 *&keep) {",
           "    u32 ok;", "    asm volatile("]
    asm_lines(out, body)
    ops = ['[ok] "=&v"(ok)']
    for nm, regs in (("X", XR), ("Y", YR), ("Z", ZR)):
        for j in range(6):
            ops.append('"+{v[%d:%d]}"(%s[%d])' % (regs[j], regs[j] + 1, nm, j))
    assert KEEP not in used and KEEP + 1 not in used and KEEP not in pinned and KEEP + 1 not in pinned
    ops.append('"+{v[%d:%d]}"(keep)' % (KEEP, KEEP + 1))
    out.append("        : " + ",\n          ".join(ops))
    ins = ['[act] "v"(act)', '[n] "s"(__builtin_amdgcn_readfirstlane(n))', '[row] "v"(row)']
    out.append("        : " + ",\n          ".join(ins))
    clob = ['"v%d"' % r for r in sorted(used)] + ['"s%d"' % i for i in range(26 if FAIR else 24)] + ['"vcc"', '"scc"', '"memory"']
    out.append("        : " + ", ".join(clob) + ");")
    out += ["    return ok;", "}"]
    print("window: %d fixed VGPRs + 36 pinned" % len(used))
    return out


OUT_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "schnorr-sig_amd", "csrc", "jac_asm.inc")


def generate():
    """the text of jac_asm.inc (the freshness test compares it with the committed file without writing anything)"""
    out = ["// generated by tools/gen_jac_asm.py -- do not edit (see that file for the design notes)"] + emit_dbl() + [""] + emit_madd() \
        + [""] + emit_window() + [""] + emit_madd_gather()
    return "\n".join(out) + "\n"


def main():
    text = generate()
    with open(OUT_PATH, "w") as fh:
        fh.write(text)
    print("wrote", OUT_PATH)


if __name__ == "__main__":
    main()
