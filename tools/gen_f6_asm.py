"""Generates schnorr-sig_amd/csrc/fp6_asm.inc: the lazy Fp6 product and square as ONE inline-asm block each.

The multiply-accumulate chains and the six final reductions need two views of the accumulator columns
-- 64-bit register pairs for v_mad_u64_u32, 32-bit halves for the carry chains of the reduction -- and
inline-asm operands cannot name half of a pair.  The accumulators therefore live in fixed VGPRs that
the block declares as clobbers; they are all caller-saved registers of the amdgpu calling convention
above the argument registers and below v128 (v48-55, v64-71, v80-87, v96-103; v0-v39 stay free for the
arguments and the pre-scaled operands the compiler computes), so the
out-of-line f6_mul_flat / f6_sqr_flat that contain the block need no saves and no argument moves.
Carries use vcc and s[0:11] (caller-saved, declared as clobbers).

gfx950 needs two wait states between a VALU write of a carry (vcc / SGPR pair) and the VALU read of it;
the compiler pads its own code with s_nop but nothing inside an asm string is padded.  Every pattern
below keeps at least two other instructions between a carry's producer and its consumer:
  * multiply-accumulate (4 mads + 4 carry adds) and the two-product chain opener: the independent mads
    of the same product sit between producer and consumer;
  * the reductions are issued three at a time, round-robin, one step of each chain at a time (two
    instructions between dependent steps); three accumulators = 27 registers are live at once, which
    keeps the whole block below v128 (a non-kernel function may not use v128 and above).

Reduction of one accumulator (value = c0 + 2^32 c1 + 2^64 (c2 + k0) + 2^96 k1 + 2^128 k2):
  2^64 = EPS = 2^32 - 1, 2^96 = -1, 2^128 = -2^32 (mod p)
  m  = c0h + c1l                (carry cL, weight 2^64)
  sl = c1h + c2l + k0 + cL      (its carries ca, cb weigh 2^96 = -1: folded into t0)
  t0 = c2h + k1 + ca + cb,  th = k2 + carries of t0
  r  = (c0l, m) - (t0, th) [borrow: - EPS] + EPS * sl [carry: + EPS]        -- as fp_reduce_parts
(round 3: the tail is one multiply-add and three carry instructions, the rare negative result goes to a cold path
behind one branch per group of three reductions: see REDUCE_STEPS)
"""
import os

# caller-saved VGPRs above the argument registers and below v128 (a non-kernel function may not touch
# v128+), as even-aligned pairs.  Three accumulators (27 registers) are live at a time; the first
# group's six result words wait in RES until the end.
POOL = []
for lo, hi in ((48, 55), (64, 71), (80, 87), (96, 103), (112, 119)):
    POOL += list(range(lo, hi + 1))
PAIRS = [(POOL[i], POOL[i + 1]) for i in range(0, len(POOL), 2)]
assert all(p[0] % 2 == 0 and p[1] == p[0] + 1 for p in PAIRS)
N_FIXED = 9 * 2 + 9 + 6          # 9 column pairs, 9 counters, 6 parked result words
RES = POOL[27:33]


class Acc:
    def __init__(self, j):
        j %= 3
        self.c = PAIRS[j * 3:(j * 3) + 3]        # three column pairs
        flat = POOL[18:27]                       # the nine counters
        self.k = flat[j * 3:j * 3 + 3]

    def pair(self, i):
        return "v[%d:%d]" % self.c[i]

    def lo(self, i):
        return "v%d" % self.c[i][0]

    def hi(self, i):
        return "v%d" % self.c[i][1]

    def kk(self, i):
        return "v%d" % self.k[i]


S0, S1 = "s[0:1]", "s[2:3]"


def init2(acc, x, y, z, w, bias=None):
    """acc = x*y + z*w (+ bias); operands are names of u64 inputs (halves %[<name>l] / %[<name>h]).
    bias = (c0 addend, c2 addend): constants below 2^33 - 1 ride for free in the zero addends of the opening
    multiplies ((2^32-1)^2 + 2^33 - 2 < 2^64, so those still cannot carry)."""
    b0, b2 = bias if bias else ("0", "0")
    xl, xh, yl, yh = "%%[%sl]" % x, "%%[%sh]" % x, "%%[%sl]" % y, "%%[%sh]" % y
    zl, zh, wl, wh = "%%[%sl]" % z, "%%[%sh]" % z, "%%[%sl]" % w, "%%[%sh]" % w
    c0, c1, c2 = acc.pair(0), acc.pair(1), acc.pair(2)
    k0, k1, k2 = acc.kk(0), acc.kk(1), acc.kk(2)
    return [
        "v_mad_u64_u32 %s, %s, %s, %s, %s" % (c0, S0, xl, yl, b0),
        "v_mad_u64_u32 %s, %s, %s, %s, 0" % (c1, S0, xl, yh),
        "v_mad_u64_u32 %s, %s, %s, %s, %s" % (c2, S0, xh, yh, b2),
        "v_mad_u64_u32 %s, vcc, %s, %s, %s" % (c1, xh, yl, c1),
        "v_mad_u64_u32 %s, %s, %s, %s, %s" % (c0, S0, zl, wl, c0),
        "v_mad_u64_u32 %s, %s, %s, %s, %s" % (c1, S1, zl, wh, c1),
        "v_addc_co_u32 %s, vcc, 0, 0, vcc" % k1,
        "v_mad_u64_u32 %s, vcc, %s, %s, %s" % (c1, zh, wl, c1),
        "v_addc_co_u32 %s, %s, 0, 0, %s" % (k0, S0, S0),
        "v_mad_u64_u32 %s, %s, %s, %s, %s" % (c2, S0, zh, wh, c2),
        "v_addc_co_u32 %s, %s, 0, %s, %s" % (k1, S1, k1, S1),
        "v_addc_co_u32 %s, vcc, 0, %s, vcc" % (k1, k1),
        "v_addc_co_u32 %s, %s, 0, 0, %s" % (k2, S0, S0),
    ]


S2 = "s[4:5]"      # free during the accumulation (the reductions use s[0:11])


def mac(acc, x, y):
    """acc += x*y: the four multiplies first (c1's two separated by one instruction), then the four carry adds --
    every carry is consumed four slots after it was produced (two wait states are required between a VALU write of
    an SGPR pair or VCC and the VALU read of it; round 1's order read VCC after ONE intervening multiply, which the
    hardware tolerated but the rule does not allow -- tests/test_asm_emulation.py enforces the rule now)"""
    xl, xh, yl, yh = "%%[%sl]" % x, "%%[%sh]" % x, "%%[%sl]" % y, "%%[%sh]" % y
    c0, c1, c2 = acc.pair(0), acc.pair(1), acc.pair(2)
    k0, k1, k2 = acc.kk(0), acc.kk(1), acc.kk(2)
    return [
        "v_mad_u64_u32 %s, vcc, %s, %s, %s" % (c1, xl, yh, c1),
        "v_mad_u64_u32 %s, %s, %s, %s, %s" % (c0, S0, xl, yl, c0),
        "v_mad_u64_u32 %s, %s, %s, %s, %s" % (c1, S2, xh, yl, c1),
        "v_mad_u64_u32 %s, %s, %s, %s, %s" % (c2, S1, xh, yh, c2),
        "v_addc_co_u32 %s, vcc, 0, %s, vcc" % (k1, k1),
        "v_addc_co_u32 %s, %s, 0, %s, %s" % (k0, S0, k0, S0),
        "v_addc_co_u32 %s, %s, 0, %s, %s" % (k1, S2, k1, S2),
        "v_addc_co_u32 %s, %s, 0, %s, %s" % (k2, S1, k2, S1),
    ]


# ---- fused linear terms ---------------------------------------------------------------------------------------
# r_k = (products)_k + sum_j sign_j * c_j * x_j[k] with small constants c_j, added to the accumulator columns BEFORE the
# reduction: one multiply per 32-bit half (c0 += c * x_lo, c1 += c * x_hi) and the two carry adds, instead of a
# separate modular addition (about ten instructions with its two wrap-around fix-ups).
# A negative term uses the complement: -x = ~x + 1 - 2^64 = ~x - (2^32 - 2) (mod p), so c * ~x is accumulated (two
# v_not more) and the constants -(2^32 - 2) * C, C = sum of the negative c_j, are paid once per coefficient -- for free:
#   (C + 1) + (2^32 - C) * 2^64 = C + 1 + 2^96 - C * 2^64 = 2C - C * 2^32 = -C (2^32 - 2)   (2^96 = -1, 2^64 = 2^32 - 1)
# i.e. the opening multiplies of columns c0 and c2 start from C + 1 and 2^32 - C instead of 0 (init2's bias).
BIAS_SGPR = "s[12:13]"
XT = POOL[33:37]          # complements of up to two negative operands


def extras_bias(extras):
    c_neg = sum(c for sign, c, _ in extras if sign < 0)
    if c_neg == 0:
        return None, []
    assert c_neg + 1 <= 64
    return (str(c_neg + 1), BIAS_SGPR), ["s_mov_b32 s12, %d" % (-c_neg), "s_mov_b32 s13, 0"]


def extra_terms(acc, k, extras):
    """instruction list for coefficient k; every carry (s[0:1], s[2:3], vcc) has two instructions between its
    producer and its consumer (s_nop where nothing else is left to place)"""
    if not extras:
        return []
    c0, c1 = acc.pair(0), acc.pair(1)
    k0, k1 = acc.kk(0), acc.kk(1)
    pre, mads, adds = [], [], []
    t = 0
    carries0 = [S0, S1]
    for n, (sign, c, name) in enumerate(extras):
        lo, hi = "%%[%s%dl]" % (name, k), "%%[%s%dh]" % (name, k)
        if sign < 0:
            pre += ["v_not_b32 v%d, %s" % (XT[t], lo), "v_not_b32 v%d, %s" % (XT[t + 1], hi)]
            lo, hi = "v%d" % XT[t], "v%d" % XT[t + 1]
            t += 2
        cs = carries0[n % 2]
        mads.append(("v_mad_u64_u32 %s, %s, %s, %d, %s" % (c0, cs, lo, c, c0), "v_addc_co_u32 %s, %s, 0, %s, %s" % (k0, cs, k0, cs)))
        mads.append(("v_mad_u64_u32 %s, vcc, %s, %d, %s" % (c1, hi, c, c1), "v_addc_co_u32 %s, vcc, 0, %s, vcc" % (k1, k1)))
    # schedule: producers in order, each consumer as early as two instructions after its producer; vcc has one
    # producer in flight at a time
    out = list(pre)
    pending = []   # (consumer, index of producer in out)
    for prod, cons in mads:
        is_vcc = ", vcc," in prod
        if is_vcc:   # the previous vcc carry must be consumed first
            for j, (pc, pi) in enumerate(pending):
                if ", vcc" in pc:
                    while len(out) - pi - 1 < 2:
                        out.append("s_nop 0")
                    out.append(pc)
                    pending.pop(j)
                    break
        # flush consumers that are ready (oldest first)
        for pc, pi in list(pending):
            if len(out) - pi - 1 >= 2:
                out.append(pc)
                pending.remove((pc, pi))
        out.append(prod)
        pending.append((cons, len(out) - 1))
    for pc, pi in pending:
        while len(out) - pi - 1 < 2:
            out.append("s_nop 0")
        out.append(pc)
    return out


REDUCE_STEPS = [
    "v_add_co_u32 {c0h}, {A}, {c0h}, {c1l}",          # 1  m
    "v_addc_co_u32 {c1h}, {A}, {c1h}, {c2l}, {A}",    # 2  sa
    "v_add_co_u32 {c1h}, {B}, {c1h}, {k0}",           # 3  sl
    "v_addc_co_u32 {c2h}, {A}, {c2h}, {k1}, {A}",     # 4  t0 = c2h + k1 + ca
    "v_addc_co_u32 {c2h}, {B}, 0, {c2h}, {B}",        # 5  t0 += cb
    "v_addc_co_u32 {k2}, {A}, 0, {k2}, {A}",          # 6  th = k2 + carry(4)
    "v_addc_co_u32 {k2}, {B}, 0, {k2}, {B}",          # 7  th += carry(5)
    # r = (c0l, m) + EPS * sl - (t0, th).  The multiply-add takes (c0l, m) as its 64-bit addend: X and the carry c (worth
    # 2^64 = EPS).  Round 3: X + c EPS - top as ONE 64-bit subtraction with c as the borrow-in of the low word
    # (EPS = 2^32 - 1: low word - 1, high word + 1):  low = X.lo - t0 - c,  high = X.hi - (th - c) - borrow.
    #   c = 1: X <= 2^64 - 2^33, so X + EPS cannot carry;  the difference is negative only when X + c EPS < top < 2^36:
    #   probability ~2^-28 per coefficient.  That rare lane is NOT repaired here: its mask (borrow of the high word,
    #   unless th - c itself wrapped: th = 0, c = 1, where nothing can go wrong) goes to T, the three masks of a group
    #   are OR-ed on the scalar unit and ONE branch per group leads to a cold path that subtracts EPS there
    #   (value + 2^64 - EPS = value + p; value + 2^64 > 2^64 - 2^36 > EPS: no second borrow).
    # 11 VALU instructions per coefficient (round 2: 15 -- borrow and carry settled by three selects and a 64-bit add;
    # round 1: 19) and two scalar ones as before.
    "v_mad_u64_u32 {c0p}, {A}, {c1h}, -1, {c0p}",     # 8  X = (c0l, m) + EPS * sl, carry c
    "v_subb_co_u32 {outl}, {B}, {c0l}, {c2h}, {A}",   # 9  X.lo - t0 - c
    "v_subbrev_co_u32 {k0}, {T}, 0, {k2}, {A}",       # 10 Z = th - c (wraps to 2^32 - 1 only for th = 0, c = 1: mask in T)
    "v_subb_co_u32 {outh}, {B}, {c0h}, {k0}, {B}",    # 11 X.hi - Z - borrow
    "s_andn2_b64 {T}, {B}, {T}",                      #    negative: final borrow and Z did not wrap   (scalar unit)
]


# First instruction of every block (as in tools/gen_jac_asm.py, where the reason was found): the compiler reloads values it
# keeps in registers a block clobbers from scratch behind the block and does not wait for such a reload in front of the
# NEXT block that merely clobbers the register; a reload still in flight would land in the block's accumulators.
ENTRY_WAIT = "s_waitcnt vmcnt(0)"


def reduce3(accs, outs, label):
    """outs[j] = (lo, hi) destination of chain j's result; label: unique name stem of this group's cold path.
    Returns (hot lines, cold lines): the hot lines end in a branch to the cold path (taken ~once in 10^8 groups) and the
    label it returns to; the cold lines are to be placed out of the way by the caller."""
    lines = []
    ms = []
    for j, a in enumerate(accs):
        ms.append({"c0p": a.pair(0), "c0l": a.lo(0), "c0h": a.hi(0), "c1l": a.lo(1), "c1h": a.hi(1), "c2l": a.lo(2), "c2h": a.hi(2),
                   "k0": a.kk(0), "k1": a.kk(1), "k2": a.kk(2), "A": "s[%d:%d]" % (4 * j, 4 * j + 1),
                   "B": "s[%d:%d]" % (4 * j + 2, 4 * j + 3), "T": "s[%d:%d]" % (14 + 2 * j, 15 + 2 * j),
                   "outl": outs[j][0], "outh": outs[j][1]})
    for step in REDUCE_STEPS:
        for m in ms:
            lines.append(step.format(**m))
    # any lane of any of the three chains negative?  (s_or sets SCC = result != 0; A of chain 0 is free by now)
    u = ms[0]["A"]
    lines += ["s_or_b64 %s, %s, %s" % (u, ms[0]["T"], ms[1]["T"]), "s_or_b64 %s, %s, %s" % (u, u, ms[2]["T"]),
              "s_cbranch_scc1 L_%s_fix_%%=" % label, "L_%s_back_%%=:" % label]
    cold = ["L_%s_fix_%%=:" % label]
    for m in ms:        # - EPS = + (1, 2^32 - 1) in the flagged lanes; the counters are free
        cold += ["v_cndmask_b32_e64 {k0}, 0, 1, {T}".format(**m), "v_cndmask_b32_e64 {k1}, 0, -1, {T}".format(**m)]
    for m in ms:
        cold += ["v_add_co_u32 {outl}, {A}, {outl}, {k0}".format(**m)]
    for m in ms:
        cold += ["v_addc_co_u32 {outh}, {A}, {outh}, {k1}, {A}".format(**m)]
    cold += ["s_branch L_%s_back_%%=" % label]
    return lines, cold


def mul_terms():
    """c_k = sum_{i<=k} a_i b_{k-i} + sum_{i>k} a_i (7 b_{k+6-i})"""
    out = []
    for k in range(6):
        t = []
        for i in range(6):
            t.append(("a%d" % i, "b%d" % (k - i)) if i <= k else ("a%d" % i, "s%d" % (k + 6 - i)))
        out.append(t)
    return out


def mul2_terms():
    """c_k of a * b + c * d: the twelve products of both in one accumulator (second pair: operands c, e = d, t = 7 d)"""
    out = []
    for k in range(6):
        t = []
        for i in range(6):
            t.append(("a%d" % i, "b%d" % (k - i)) if i <= k else ("a%d" % i, "s%d" % (k + 6 - i)))
        for i in range(6):
            t.append(("c%d" % i, "e%d" % (k - i)) if i <= k else ("c%d" % i, "t%d" % (k + 6 - i)))
        out.append(t)
    return out


def sqr_terms():
    """direct i + j = k (i <= j): a_i * (a_j | 2a_j); wrapped i + j = k + 6: a_i * (7a_j | 14a_j)"""
    out = []
    for k in range(6):
        t = []
        for i in range(0, k // 2 + 1):
            j = k - i
            t.append(("a%d" % i, ("a%d" if i == j else "d%d") % j))
        for i in range(k + 1, (k + 6) // 2 + 1):
            j = k + 6 - i
            if j > 5:
                continue
            t.append(("a%d" % i, ("s%d" if i == j else "t%d") % j))
        out.append(t)
    return out


def emit(name, terms, inputs, doc, extras=()):
    """extras: fused linear terms (sign, small constant, operand prefix); their operand arrays are appended to inputs"""
    accs = [Acc(j) for j in range(6)]
    lines = [ENTRY_WAIT]
    cold = []
    bias, bias_setup = extras_bias(extras)
    lines += bias_setup
    for g in range(2):
        for k in range(3 * g, 3 * g + 3):
            t = terms[k]
            assert len(t) >= 2
            lines += init2(accs[k], t[0][0], t[0][1], t[1][0], t[1][1], bias)
            for x, y in t[2:]:
                lines += mac(accs[k], x, y)
            lines += extra_terms(accs[k], k, extras)
        if g == 0:   # inputs are still needed: park the results in fixed registers
            outs = [("v%d" % RES[2 * j], "v%d" % RES[2 * j + 1]) for j in range(3)]
        else:
            outs = [("%%[r%dl]" % k, "%%[r%dh]" % k) for k in range(3, 6)]
        hot, cld = reduce3(accs[3 * g:3 * g + 3], outs, "r%d" % g)
        lines += hot
        cold += cld
    for j in range(3):
        lines.append("v_mov_b32 %%[r%dl], v%d" % (j, RES[2 * j]))
        lines.append("v_mov_b32 %%[r%dh], v%d" % (j, RES[2 * j + 1]))
    lines += ["s_branch L_end_%="] + cold + ["L_end_%=:"]
    used = set()
    for ln in lines:
        for tok in ln.replace(",", " ").split():
            if tok.startswith("%["):
                used.add(tok[2:-1])
    out = ["// %s" % doc, "SSA_DEV void %s(%s, u64 (&r)[6]) {" % (name, ", ".join("const u64 (&%s)[6]" % n for n, _ in inputs))]
    out.append("    u32 " + ", ".join("r%dl, r%dh" % (j, j) for j in range(6)) + ";")
    out.append("    asm(")
    for i, ln in enumerate(lines):
        out.append('        "%s%s"' % (ln, "\\n\\t" if i + 1 < len(lines) else ""))
    outs = []
    for j in range(6):
        outs.append('[r%dl] "=v"(r%dl)' % (j, j))
        outs.append('[r%dh] "=v"(r%dh)' % (j, j))
    out.append("        : " + ", ".join(outs))
    ins = []
    for arr, prefix in inputs:
        for j in range(6):
            nm = "%s%d" % (prefix, j)
            if nm + "l" in used or nm + "h" in used:
                ins.append('[%sl] "v"(lo32(%s[%d]))' % (nm, arr, j))
                ins.append('[%sh] "v"(hi32(%s[%d]))' % (nm, arr, j))
    out.append("        : " + ",\n          ".join(ins))
    n_fixed = N_FIXED + (4 if any(sg < 0 for sg, _, _ in extras) else 0)
    n_sgpr = 20       # s[0:11] carries, s[12:13] the bias of the fused blocks, s[14:19] scratch masks of the reductions
    clob = ['"v%d"' % r for r in POOL[:n_fixed]] + ['"s%d"' % i for i in range(n_sgpr)] + ['"vcc"', '"scc"']
    out.append("        : " + ", ".join(clob) + ");")
    for j in range(6):
        out.append("    r[%d] = mk64(r%dl, r%dh);" % (j, j, j))
    out.append("}")
    nm = sum(1 for ln in lines if ln.startswith("v_mad"))
    return out, len(lines), nm


# ---- Fp3 = Fp[t]/(t^3 - 7): the product and the square of the square-root descent (fp3.hpp, round 4) -------------------
# One group of three accumulators, one reduce3.  The compiled f3_mul (fp_acc in C++) costs ~190 instructions and served as
# the square too (9 products); here the product is 9 x 8 + 33 and the square 6 x 8 + 33 instructions.
def f3_mul_terms():
    """c_k = sum_{i+j=k} a_i b_j + sum_{i+j=k+3} a_i (7 b_j)"""
    return [[("a0", "b0"), ("a1", "s2"), ("a2", "s1")],
            [("a0", "b1"), ("a1", "b0"), ("a2", "s2")],
            [("a0", "b2"), ("a1", "b1"), ("a2", "b0")]]


def f3_sqr_terms():
    """c_0 = a0^2 + a1 (14 a2), c_1 = a0 (2 a1) + a2 (7 a2), c_2 = a0 (2 a2) + a1^2"""
    return [[("a0", "a0"), ("a1", "t2")],
            [("a0", "d1"), ("a2", "s2")],
            [("a0", "d2"), ("a1", "a1")]]


def emit3(name, terms, inputs, doc):
    accs = [Acc(j) for j in range(3)]
    lines = [ENTRY_WAIT]
    for k in range(3):
        t = terms[k]
        lines += init2(accs[k], t[0][0], t[0][1], t[1][0], t[1][1], None)
        for x, y in t[2:]:
            lines += mac(accs[k], x, y)
    # the inputs are dead once the last product is issued, but an output operand may share a register with an input the
    # compiler still needs: the results wait in the parking registers until the end, as in the six-coefficient blocks
    outs = [("v%d" % RES[2 * j], "v%d" % RES[2 * j + 1]) for j in range(3)]
    hot, cold = reduce3(accs, outs, "r0")
    lines += hot
    for j in range(3):
        lines.append("v_mov_b32 %%[r%dl], v%d" % (j, RES[2 * j]))
        lines.append("v_mov_b32 %%[r%dh], v%d" % (j, RES[2 * j + 1]))
    lines += ["s_branch L_end_%="] + cold + ["L_end_%=:"]
    used = set()
    for ln in lines:
        for tok in ln.replace(",", " ").split():
            if tok.startswith("%["):
                used.add(tok[2:-1])
    out = ["// %s" % doc, "SSA_DEV void %s(%s, u64 (&r)[3]) {" % (name, ", ".join("const u64 (&%s)[3]" % n for n, _ in inputs))]
    out.append("    u32 " + ", ".join("r%dl, r%dh" % (j, j) for j in range(3)) + ";")
    out.append("    asm(")
    for i, ln in enumerate(lines):
        out.append('        "%s%s"' % (ln, "\\n\\t" if i + 1 < len(lines) else ""))
    outs = []
    for j in range(3):
        outs.append('[r%dl] "=v"(r%dl)' % (j, j))
        outs.append('[r%dh] "=v"(r%dh)' % (j, j))
    out.append("        : " + ", ".join(outs))
    ins = []
    for arr, prefix in inputs:
        for j in range(3):
            nm = "%s%d" % (prefix, j)
            if nm + "l" in used or nm + "h" in used:
                ins.append('[%sl] "v"(lo32(%s[%d]))' % (nm, arr, j))
                ins.append('[%sh] "v"(hi32(%s[%d]))' % (nm, arr, j))
    out.append("        : " + ",\n          ".join(ins))
    clob = ['"v%d"' % r for r in POOL[:N_FIXED]] + ['"s%d"' % i for i in range(20)] + ['"vcc"', '"scc"']
    out.append("        : " + ", ".join(clob) + ");")
    for j in range(3):
        out.append("    r[%d] = mk64(r%dl, r%dh);" % (j, j, j))
    out.append("}")
    nm = sum(1 for ln in lines if ln.startswith("v_mad"))
    return out, len(lines), nm


# ---- one accumulator: r = x0 y0 + x1 y1 + x2 y2 (mod p) -- what ONE lane of the cooperative kernels computes per product
# round (ssa_coop.hpp coop_group_mul: twelve lanes per Fp6 product, three terms each).  The latency path: the reduction
# of a single chain cannot fill the wait states of its carries with a neighbour's instructions, so they are padded (s_nop).
def pad_wait_states(lines, gap=3):
    """insert s_nop so that a VALU read of an SGPR pair / VCC comes at least `gap` positions after its VALU write"""
    import re
    out, written = [], {}
    for ln in lines:
        if ln.startswith("v_"):
            ops = [o.strip() for o in ln.split(None, 1)[1].split(",")]
            mnem = ln.split()[0]
            n_dst = 2 if mnem in ("v_mad_u64_u32", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_subbrev_co_u32") else 1
            reads = [o for o in ops[n_dst:] if re.match(r"(s\[\d+:\d+\]|vcc)$", o)]
            need = 0
            for r in reads:
                if r in written:
                    need = max(need, gap - (len(out) - written[r]))
            if need > 0:
                out.append("s_nop %d" % (need - 1))
                # an s_nop N occupies one position and N + 1 wait states: account for it as `need` positions
                for k in list(written):
                    written[k] -= need - 1
            out.append(ln)
            if n_dst == 2 and re.match(r"(s\[\d+:\d+\]|vcc)$", ops[1]):
                written[ops[1]] = len(out) - 1
        else:
            out.append(ln)
    return out


def emit_acc3(name, doc):
    acc = Acc(0)
    lines = [ENTRY_WAIT] + init2(acc, "x0", "y0", "x1", "y1", None) + mac(acc, "x2", "y2")
    m = {"c0p": acc.pair(0), "c0l": acc.lo(0), "c0h": acc.hi(0), "c1l": acc.lo(1), "c1h": acc.hi(1), "c2l": acc.lo(2),
         "c2h": acc.hi(2), "k0": acc.kk(0), "k1": acc.kk(1), "k2": acc.kk(2), "A": "s[0:1]", "B": "s[2:3]", "T": "s[14:15]",
         "outl": "%[rl]", "outh": "%[rh]"}
    red = [st.format(**m) for st in REDUCE_STEPS]
    lines = pad_wait_states(lines + red)
    # the rare negative result (mask T, SCC from s_andn2): - EPS = + (1, 2^32 - 1) in the flagged lanes
    lines += ["s_cbranch_scc1 L_fix_%=", "L_back_%=:", "s_branch L_end_%=", "L_fix_%=:",
              "v_cndmask_b32_e64 {k0}, 0, 1, {T}".format(**m), "v_cndmask_b32_e64 {k1}, 0, -1, {T}".format(**m),
              "v_add_co_u32 %[rl], {A}, %[rl], {k0}".format(**m), "s_nop 1",
              "v_addc_co_u32 %[rh], {A}, %[rh], {k1}, {A}".format(**m), "s_branch L_back_%=", "L_end_%=:"]
    out = ["// %s" % doc, "SSA_DEV u64 %s(const u64 (&x)[3], const u64 (&y)[3]) {" % name, "    u32 rl, rh;", "    asm("]
    for i, ln in enumerate(lines):
        out.append('        "%s%s"' % (ln, "\\n\\t" if i + 1 < len(lines) else ""))
    out.append('        : [rl] "=&v"(rl), [rh] "=&v"(rh)')
    ins = []
    for arr in ("x", "y"):
        for j in range(3):
            ins.append('[%s%dl] "v"(lo32(%s[%d]))' % (arr, j, arr, j))
            ins.append('[%s%dh] "v"(hi32(%s[%d]))' % (arr, j, arr, j))
    out.append("        : " + ",\n          ".join(ins))
    fixed = sorted(set(r for pr in acc.c for r in pr) | set(acc.k))
    clob = ['"v%d"' % r for r in fixed] + ['"s%d"' % i for i in (0, 1, 2, 3, 4, 5, 14, 15)] + ['"vcc"', '"scc"']
    out.append("        : " + ", ".join(clob) + ");")
    out += ["    return mk64(rl, rh);", "}"]
    n_valu = sum(1 for ln in lines if ln.startswith("v_"))
    n_nop = sum(int(ln.split()[1]) + 1 for ln in lines if ln.startswith("s_nop"))
    print("%s: %d VALU instructions on the hot path + %d wait states" % (name, n_valu - 4, n_nop - 2))
    return out


OUT_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "schnorr-sig_amd", "csrc", "fp6_asm.inc")


def generate():
    """the text of fp6_asm.inc (the freshness test compares it with the committed file without writing anything)"""
    hdr = ["// generated by tools/gen_f6_asm.py -- do not edit (see that file for the design notes)"]
    m, nl, nm = emit("f6_mul_core_asm", mul_terms(), [("a", "a"), ("b", "b"), ("b7", "s")],
                     "r = a * b in Fp[u]/(u^6 - 7); b7[j] = 7 b[j] (j = 1..5)")
    print("mul: %d instructions, %d mads" % (nl, nm))
    s, nl, nm = emit("f6_sqr_core_asm", sqr_terms(), [("a", "a"), ("a2", "d"), ("a7", "s"), ("a14", "t")],
                     "r = a^2; a2[j] = 2 a[j] (j = 1..5), a7[j] = 7 a[j], a14[j] = 14 a[j] (j = 3..5)")
    print("sqr: %d instructions, %d mads" % (nl, nm))
    sqr_in = [("a", "a"), ("a2", "d"), ("a7", "s"), ("a14", "t")]
    mul_in = [("a", "a"), ("b", "b"), ("b7", "s")]
    fused = []
    for nm, terms, ins, ex, doc in (
            ("f6_sqr_sub2_core_asm", sqr_terms(), sqr_in + [("x", "x"), ("y", "y")], [(-1, 1, "x"), (-1, 1, "y")],
             "r = a^2 - x - y (prescaled operands as f6_sqr_core_asm)"),
            ("f6_sqr_add3x_core_asm", sqr_terms(), sqr_in + [("x", "x")], [(+1, 3, "x")], "r = a^2 + 3 x"),
            ("f6_sqr_sub4x_core_asm", sqr_terms(), sqr_in + [("x", "x")], [(-1, 4, "x")], "r = a^2 - 4 x"),
            ("f6_mul_sub8x_core_asm", mul_terms(), mul_in + [("x", "x")], [(-1, 8, "x")], "r = a * b - 8 x"),
            ("f6_mul_subx_core_asm", mul_terms(), mul_in + [("x", "x")], [(-1, 1, "x")], "r = a * b - x"),
            ("f6_mul2_add_core_asm", mul2_terms(), mul_in + [("c", "c"), ("d", "e"), ("d7", "t")], [],
             "r = a * b + c * d; b7[j] = 7 b[j], d7[j] = 7 d[j] (j = 1..5)"),
            ("f6_sqr_subx_sub2y_core_asm", sqr_terms(), sqr_in + [("x", "x"), ("y", "y")], [(-1, 1, "x"), (-1, 2, "y")],
             "r = a^2 - x - 2 y")):
        blk, nl, nmad = emit(nm, terms, ins, doc, ex)
        print("%s: %d instructions, %d mads" % (nm, nl, nmad))
        fused += [""] + blk
    f3 = []
    for nm, terms, ins, doc in (
            ("f3_mul_core_asm", f3_mul_terms(), [("a", "a"), ("b", "b"), ("b7", "s")], "r = a * b in Fp[t]/(t^3 - 7); b7[j] = 7 b[j] (j = 1, 2)"),
            ("f3_sqr_core_asm", f3_sqr_terms(), [("a", "a"), ("a2", "d"), ("a7", "s"), ("a14", "t")],
             "r = a^2 in Fp[t]/(t^3 - 7); a2[j] = 2 a[j] (j = 1, 2), a7[2] = 7 a[2], a14[2] = 14 a[2]")):
        blk, nl, nmad = emit3(nm, terms, ins, doc)
        print("%s: %d instructions, %d mads" % (nm, nl, nmad))
        f3 += [""] + blk
    acc3 = [""] + emit_acc3("fp_acc3_core_asm", "r = x0 y0 + x1 y1 + x2 y2 (mod p, loose): one lane's share of a cooperative Fp6 product")
    return "\n".join(hdr + m + [""] + s + fused + f3 + acc3) + "\n"


def main():
    text = generate()
    with open(OUT_PATH, "w") as fh:
        fh.write(text)
    print("wrote", OUT_PATH)


if __name__ == "__main__":
    main()
