#!/usr/bin/env python3
"""Generates schnorr-sig_amd/csrc/fp_chain_asm.inc: chains of Goldilocks squarings for TWO independent values at a
time, one inline-asm loop -- the shape of the Rescue inverse S-box x^(1/7) (63 squarings + 9 products per state
element, reference hash::rescue_64_12_8 called at src/signature.rs:303), 58 of whose squarings sit in five
`square n times, multiply once` runs.

Why: the kernel is bound by VALU ISSUE, and on gfx950 every VALU instruction in a stream that contains the 64-bit
multiplier costs about 4 cycles per wave whatever it is (tools/isa_probe, DESIGN.md "instruction cost table"), so what
counts is the number of instructions.  hipcc's squaring is 22 VALU instructions + 2 s_nop: it detects the borrow and
the carry of the reduction with separate 64-bit compares (v_cmp_lt_u64 + v_mov to build the zero-extended operand)
although v_subbrev / v_addc already deliver them, and it pads the two VCC hazards of its single dependent chain.
Here: 17 instructions per squaring, carries in SGPR pairs, and the second value's instructions fill the two wait
states gfx950 needs between a VALU write of an SGPR and the VALU read of it (a list scheduler below places them).
Two values only: six at a time (round 1) cost registers, occupancy fell from 4 to 3 waves per SIMD and a lone wave
issues only every ~6 cycles -- slower in spite of fewer instructions.

One squaring, a = a0 + 2^32 a1 (three multiplies, fp.hpp fp_sqr3):
    t0 = a0^2;  u = a0 a1 + (t0 >> 33);  hi = a1^2 + (u >> 31);  lo = (t0 mod 2^33) | (u mod 2^31) << 33
reduction (fp_reduce128): r = lo - hi.hi [borrow: + p] + EPS * hi.lo [carry: + EPS]

Round 3 -- the reduction tail in THREE instructions, 11 VALU per squaring instead of 14 (and two scalar ones as before).
After X = (EPS * h0 + lo) mod 2^64 with carry c (one multiply-add) the value wanted is X + c * EPS - h1 (h1 = hi.hi < 2^32):
  * c = 1: X <= 2^64 - 2^33, so X + EPS - h1 = X + ~h1 can neither carry nor borrow;
  * c = 0: X - h1 borrows only when X < h1 < 2^32 -- probability ~2^-32 for the values of a hash.
Both cases are ONE 64-bit subtraction with c as the borrow-in of the low word (X.lo - h1 - c) and the high word
X.hi - (c ? 0xffffffff : 0) - borrow, i.e. v_subb, v_cndmask, v_subb.  The rare borrow of the c = 0 case (the value would
need + p) is NOT repaired in the chain: the borrow-out of the last v_subb, masked by "not c", is OR-ed into a sticky SGPR
pair per chain (two scalar instructions per squaring, where the old tail spent two on combining its masks), the block
reports the lanes that hit it, and the C++ wrapper recomputes those lanes' S-boxes with the compiled exact code
(rescue.hpp).  Wrong intermediate values in a flagged lane are harmless: nothing of it is kept.
"""
import os

# Chains per block (round 4: THREE).  The squaring is a dependent sequence of eleven instructions; with two chains a reader
# sits two or three positions behind its producer, and even four waves per SIMD do not hide that: a pure squaring stream
# costs 4.73 cycles per instruction with two chains, 4.35 with three, 4.33 with four, 4.39 with six (tools/isa_probe,
# probes sq_chains_N, four waves per SIMD; at two waves three chains cost 4.71).  Three chains of 13 register pairs are 78
# VGPRs (v50..v127): the kernel keeps its 128 registers = four waves per SIMD.  (Round 1 tried six chains and lost a wave
# of occupancy; round 3 costed three as "no instruction disappears" -- true, and beside the point.)  SSA_GEN_CHAINS=2 for A/B.
NCH = int(os.environ.get("SSA_GEN_CHAINS", "3"))
N_PAIRS = 13      # T A U H M E C + the six value pairs X V0..V4
BASE = 128 - 2 * NCH * N_PAIRS      # all above the compiler's own allocation for ssa_k_hash, all below v128


def regs(c, nch=None):
    """chain c's register pairs (interleaved with the other chains': pair k of chain c is BASE + 2 c + 2 n k for a block of n
    chains, the block ending at v127) and its three carry pairs S, S2, S3 (s32 / s33 are the stack and frame pointers)"""
    n = NCH if nch is None else nch
    base = 128 - 2 * n * N_PAIRS

    def pair(k):
        return base + 2 * c + 2 * n * k

    g = {"X": pair(0), "T": pair(1), "A": pair(2), "U": pair(3), "H": pair(4), "M": pair(5), "E": pair(6), "C": pair(7),
         "S": 20 + 2 * c, "S2": 26 + 2 * c, "S3": 36 + 2 * c}
    for k in range(5):
        g["V%d" % k] = pair(8 + k)
    return g


# the head of the 64x64 product.  "moves" (default): operand scanning, five register moves + one 64-bit add between the
# four multiplies (10 instructions); "carries" (SSA_GEN_MUL_HEAD=carries): four carry instructions instead (8).  Measured
# on one box, alternating (profiles/r03/mul_head_ab.txt): ssa_k_hash 8.36 / 8.39 / 8.40 ms with the moves against
# 8.48 / 8.52 / 8.55 ms with the carry chain -- v_mov_b32 issues at twice the rate of the instructions that write or read
# an SGPR carry, so the longer head is the faster one.
MOVES_HEAD = os.environ.get("SSA_GEN_MUL_HEAD", "moves") != "carries"       # "moves" (default), "moves10", "carries"
SQ_MERGE_FAST = os.environ.get("SSA_GEN_SQ_MERGE", "") == "fast"
MUL_HEAD_ADD1 = os.environ.get("SSA_GEN_MUL_HEAD", "moves") != "moves10"     # round 4 default; "moves10" = round 3's head
DUMMY = "s[42:43]"     # carry-outs nobody reads
CUR_NCH = [None]       # chains of the block being generated (None: NCH); square() / multiply() read their registers through it
SGPRS = list(range(20, 32)) + list(range(36, 44))


def vp(r):
    return "v[%d:%d]" % (r, r + 1)


def sp(r):
    return "s[%d:%d]" % (r, r + 1)


def square(c, src=None, dst=None, st=None):
    """(text, sgprs read, sgprs written) of one squaring of chain c, in dependency order.  src / dst: the register pairs
    the value is read from and written to (default: in place in X); st: the asm operand that collects the rare-borrow
    lanes (default: the chain's own sticky pair)"""
    g = regs(c, CUR_NCH[0])
    X, T, A, U, H, M, E, C, S, S2, S3 = (g[k] for k in ("X", "T", "A", "U", "H", "M", "E", "C", "S", "S2", "S3"))
    XS = X if src is None else src          # the three products read the source pair ...
    X = X if dst is None else dst           # ... the reduction writes the destination pair
    assert st is not None                   # the asm operand that collects the lanes of the rare borrow
    s = sp(S)
    return [
        ("v_mad_u64_u32 %s, %s, v%d, v%d, 0" % (vp(T), DUMMY, XS, XS), [], []),
        ("v_lshrrev_b32 v%d, 1, v%d" % (A, T + 1), [], []),                       # (t0 >> 33); v[A+1] stays 0
        ("v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (vp(U), DUMMY, XS, XS + 1, vp(A)), [], []),
        ("v_and_b32 v%d, 1, v%d" % (T + 1, T + 1), [], []),
        ("v_lshrrev_b64 %s, 31, %s" % (vp(H), vp(U)), [], []),
    ] + ([
        ("v_lshl_or_b32 v%d, v%d, 1, v%d" % (T + 1, U, T + 1), [], []),           # lo = v[T:T+1]
    ] if not SQ_MERGE_FAST else [
        # A/B of round 4 (SSA_GEN_SQ_MERGE=fast): the same merge as two instructions of the 32-lanes-per-clock class
        # (v_add_u32, v_or_b32) instead of one of the 16-lanes-per-clock class -- 12 instructions per squaring instead of
        # 11, the same 40 cycles on the cost table; measured in profiles/r04/hash_ab.txt
        ("v_add_u32 v%d, v%d, v%d" % (M, U, U), [], []),                          # (u.lo << 1) mod 2^32
        ("v_or_b32 v%d, v%d, v%d" % (T + 1, M, T + 1), [], []),
    ]) + [
        ("v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (vp(H), DUMMY, XS + 1, XS + 1, vp(H)), [], []),   # hi
        # reduction (see the module docstring): X = EPS * h0 + lo (carry c -> s2), then X + c EPS - h1 as one 64-bit
        # subtraction with borrow-in c; a final borrow while c = 0 is the rare "+ p" case: sticky, repaired by the caller
        ("v_mad_u64_u32 %s, %s, v%d, -1, %s" % (vp(X), sp(S2), H, vp(T)), [], [S2]),
        ("v_subb_co_u32 v%d, %s, v%d, v%d, %s" % (X, s, X, H + 1, sp(S2)), [], [S]),                  # X.lo - h1 - c
        ("v_cndmask_b32 v%d, 0, -1, %s" % (E, sp(S2)), [], []),                                       # c ? 0xffffffff : 0
        ("v_subb_co_u32 v%d, %s, v%d, v%d, %s" % (X + 1, sp(S3), X + 1, E, s), [], [S3]),             # X.hi + c - borrow
    ] + ([] if os.environ.get("SSA_GEN_NO_STICKY") else [       # (timing probe only: no rare-borrow bookkeeping at all)
        ("s_andn2_b64 %s, %s, %s" % (sp(S3), sp(S3), sp(S2)), [], []),                                # borrow and not c
        ("s_or_b64 %s, %s, %s" % (st, st, sp(S3)), [], []),
    ])


def reduce_tail(c, dst=None, st=None):
    """lo = v[T:T+1], hi = v[H:H+1] -> X or dst (the last six instructions of square())"""
    return square(c, None, dst, st)[7:]      # from the multiply-add by EPS on


def multiply(c, saved, src=None, dst=None, st=None):
    """X <- X * saved value (register pair name "V0".."V4"): four multiplies, operand-scanning with the addends in
    zero-extended pairs (v[A+1] and v[C+1] hold 0), then the reduction.  The alternative below the early return -- the
    65-bit middle sum's carry k riding into the top product as the addend (0, k), a three-instruction carry chain for the
    halves: 8 instructions instead of 10 -- measured slower (see MOVES_HEAD)."""
    g = regs(c, CUR_NCH[0])
    X, T, A, U, H, C, E, V, S = g["X"], g["T"], g["A"], g["U"], g["H"], g["C"], g["E"], g[saved] if isinstance(saved, str) else saved, g["S"]
    if src is not None:
        X = src                 # the head reads the source pair; the reduction writes dst (reduce_tail)
    S2, S3 = g["S2"], g["S3"]
    if MOVES_HEAD and not MUL_HEAD_ADD1:        # round 3's head: 10 instructions
        return [
            ("v_mad_u64_u32 %s, %s, v%d, v%d, 0" % (vp(T), DUMMY, X, V), [], []),
            ("v_mov_b32 v%d, v%d" % (A, T + 1), [], []),
            ("v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (vp(U), DUMMY, X, V + 1, vp(A)), [], []),
            ("v_mov_b32 v%d, v%d" % (A, U), [], []),
            ("v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (vp(H), DUMMY, X + 1, V, vp(A)), [], []),
            ("v_mov_b32 v%d, v%d" % (T + 1, H), [], []),
            ("v_mov_b32 v%d, v%d" % (A, U + 1), [], []),
            ("v_mov_b32 v%d, v%d" % (C, H + 1), [], []),
            ("v_lshl_add_u64 %s, %s, 0, %s" % (vp(E), vp(A), vp(C)), [], []),       # v[A+1], v[C+1] hold 0
            ("v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (vp(H), DUMMY, X + 1, V + 1, vp(E)), [], []),
        ] + reduce_tail(c, dst, st)
    if MOVES_HEAD:
        # round 4: NINE instructions.  The high word of the second cross product joins the top product through a
        # multiply-add by the inline constant 1 (S0 may be any VGPR, so no zero-extended pair has to be built for it):
        # x1 s1 + t1.hi + t2.hi <= 2^64 - 1 is the high half of a 128-bit product, it cannot overflow.  One move and the
        # 64-bit add go; the kernel pays per instruction (profiles/r04/hash_ab.txt).
        M = g["M"]
        return [
            ("v_mad_u64_u32 %s, %s, v%d, v%d, 0" % (vp(T), DUMMY, X, V), [], []),                  # t0 = x0 s0
            ("v_mov_b32 v%d, v%d" % (A, T + 1), [], []),
            ("v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (vp(U), DUMMY, X, V + 1, vp(A)), [], []),      # t1 = x0 s1 + t0.hi
            ("v_mov_b32 v%d, v%d" % (A, U), [], []),
            ("v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (vp(M), DUMMY, X + 1, V, vp(A)), [], []),      # t2 = x1 s0 + t1.lo
            ("v_mov_b32 v%d, v%d" % (T + 1, M), [], []),                                           # lo = (t0.lo, t2.lo)
            ("v_mov_b32 v%d, v%d" % (A, U + 1), [], []),
            ("v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (vp(H), DUMMY, X + 1, V + 1, vp(A)), [], []),  # x1 s1 + t1.hi
            ("v_mad_u64_u32 %s, %s, v%d, 1, %s" % (vp(H), DUMMY, M + 1, vp(H)), [], []),           # + t2.hi
        ] + reduce_tail(c, dst, st)
    head = [
        ("v_mad_u64_u32 %s, %s, v%d, v%d, 0" % (vp(T), DUMMY, X, V), [], []),                   # t0 = x0 s0
        ("v_mad_u64_u32 %s, %s, v%d, v%d, 0" % (vp(U), DUMMY, X, V + 1), [], []),               # x0 s1
        ("v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (vp(U), sp(S), X + 1, V, vp(U)), [], [S]),      # mid = x0 s1 + x1 s0, carry k
        ("v_cndmask_b32 v%d, 0, 1, %s" % (C + 1, sp(S)), [], []),                               # (0, k): v[C] stays 0
        ("v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (vp(H), DUMMY, X + 1, V + 1, vp(C)), [], []),   # t3 = x1 s1 + k 2^32
        ("v_add_co_u32 v%d, %s, v%d, v%d" % (T + 1, sp(S2), T + 1, U), [], [S2]),               # lo = (t0.lo, t0.hi + mid.lo)
        ("v_addc_co_u32 v%d, %s, v%d, v%d, %s" % (H, sp(S3), H, U + 1, sp(S2)), [], [S3]),      # hi.lo = t3.lo + mid.hi + carry
        ("v_addc_co_u32 v%d, %s, 0, v%d, %s" % (H + 1, sp(S2), H + 1, sp(S3)), [], [S2]),       # hi.hi += carry (cannot overflow)
    ]
    return head + reduce_tail(c, dst, st)


def zero_inits(g):
    """the registers a block expects to hold 0: the high word of the zero-extended addend pair A, and -- for the heads
    that add two zero-extended words -- of C (low word of C for the carry-chain head)"""
    out = ["v_mov_b32 v%d, 0" % (g["A"] + 1)]
    if not (MOVES_HEAD and MUL_HEAD_ADD1):
        out.append("v_mov_b32 v%d, 0" % (g["C"] + 1 if MOVES_HEAD else g["C"]))
    return out


def copy(c, dst, src):
    g = regs(c)
    return [("v_mov_b32 v%d, v%d" % (g[dst], g[src]), [], []), ("v_mov_b32 v%d, v%d" % (g[dst] + 1, g[src] + 1), [], [])]


# the chains as programs: ("sq", n) n squarings (a loop when n > 2), ("mul", V), ("cp", dst, src)
INV_SBOX = [("cp", "V0", "X"), ("sq", 1), ("cp", "V1", "X"), ("sq", 1), ("cp", "V2", "X"),        # x, x^2, x^4
            ("sq", 3), ("mul", "V2"), ("cp", "V3", "X"),                                         # t3
            ("sq", 6), ("mul", "V3"), ("cp", "V4", "X"),                                         # t4
            ("sq", 12), ("mul", "V4"),                                                           # t5
            ("sq", 6), ("mul", "V3"), ("cp", "V4", "X"),                                         # t6
            ("sq", 31), ("mul", "V4"),                                                           # t7
            ("sq", 1), ("mul", "V4"), ("sq", 2), ("cp", "V4", "X"),                              # a
            ("cp", "X", "V1"), ("mul", "V2"), ("mul", "V0"), ("mul", "V4")]                      # a * (x^2 x^4 x)
SBOX = [("cp", "V0", "X"), ("sq", 1), ("cp", "V1", "X"), ("sq", 1), ("mul", "V1"), ("mul", "V0")]   # x^7 = x^4 x^2 x


VALUE_PAIRS = ["X", "V0", "V1", "V2", "V3", "V4"]     # the six pairs a chain's values live in (the roles move: see rename)


def rename(prog):
    """The programs say `cp V, X` / `cp X, V`; no copy is ever executed (round 4).  Every squaring and product reads its
    source pair in its first instructions and writes its result with its last three, so the result can go to ANY free pair:
    a copy is a second name for the pair the value is in, and the next operation writes somewhere else.  Returns
    [(op, operand index or None, source index, destination index)] over indices into VALUE_PAIRS' registers; the input
    arrives in pair 0 and the result is steered back into pair 0 (the pinned in/out operand of the asm statement)."""
    # names read after op i (a later `cp V, X` re-defines V: its old pair is dead from its last use on)
    live_after, live = [None] * len(prog), set()
    for i in range(len(prog) - 1, -1, -1):
        live_after[i] = set(live)
        op = prog[i]
        if op[0] == "mul":
            live.add(op[1])
        elif op[0] == "cp" and op[2] == "X":
            live.discard(op[1])
        elif op[0] == "cp" and op[1] == "X":
            live.add(op[2])
    idx, out = {"X": 0}, []
    n_ops = sum(op[1] if op[0] == "sq" else 1 for op in prog if op[0] != "cp")
    done = 0
    for i, op in enumerate(prog):
        if op[0] == "cp":
            if op[1] == "X":
                idx["X"] = idx[op[2]]
            else:
                idx[op[1]] = idx["X"]
            continue
        for k in range(op[1] if op[0] == "sq" else 1):
            last_of_op = op[0] != "sq" or k == op[1] - 1
            # pairs that hold a value somebody still needs: names live after this operation (while a run of squarings is
            # in progress, the names live after the whole run)
            busy = {idx[nm] for nm in live_after[i] if nm in idx}
            v = idx[op[1]] if op[0] == "mul" else None
            src = idx["X"]
            done += 1
            if done == n_ops:
                dst = 0                         # the result leaves in the pair the input came in
                assert 0 not in busy
            elif src not in busy:
                dst = src                       # in place
            else:
                dst = min(j for j in range(len(VALUE_PAIRS)) if j not in busy and j != src and j != v)
            out.append((op[0], v, src, dst))
            idx["X"] = dst
            del last_of_op
    return out


def emit_program(name, prog, doc, nch=None):
    nch = NCH if nch is None else nch
    CUR_NCH[0] = nch
    args = ["x", "y", "z", "w"][:nch]
    base = 128 - 2 * nch * N_PAIRS
    gs = [regs(c, nch) for c in range(nch)]
    lines = ["// %s" % doc,
             "// The values are pinned to the blocks' own value registers (no moves in or out; the compiler loads the state",
             "// straight into them); st collects the lanes (bit per lane) where a reduction met its rare borrow: their values are",
             "// then WRONG and the caller recomputes them from its inputs with the compiled exact code (rescue.hpp)",
             "SSA_DEV void %s(%s, u64 &st) {" % (name, ", ".join("u64 &" + a for a in args)), "    asm volatile("]
    body = ["s_waitcnt vmcnt(0)"]       # nothing of the compiler's in flight into the block's registers (tools/gen_jac_asm.py)
    for g in gs:
        body += zero_inits(g)
    counts = {"valu": len(body) - 1, "nop": 0}
    for kind, v, src, dst in rename(prog):
        chains = []
        for c, g in enumerate(gs):
            pr = [g[nm] for nm in VALUE_PAIRS]
            if kind == "sq":
                chains.append(square(c, pr[src], pr[dst], "%[st]"))
            else:
                chains.append(multiply(c, pr[v], pr[src], pr[dst], "%[st]"))
        seg = schedule(chains)
        body += seg
        counts["valu"] += sum(1 for ln in seg if ln.startswith("v_"))
        counts["nop"] += sum(1 for ln in seg if ln.startswith("s_nop"))
    for i, ln in enumerate(body):
        lines.append('        "%s%s"' % (ln, "\\n\\t" if i + 1 < len(body) else ""))
    lines.append("        : " + ", ".join('"+{v[%d:%d]}"(%s)' % (g["X"], g["X"] + 1, a) for g, a in zip(gs, args)) + ', [st] "+s"(st)')
    lines.append("        :")
    pinned = {r for g in gs for r in (g["X"], g["X"] + 1)}
    clob = ['"v%d"' % r for r in range(base, base + 2 * nch * N_PAIRS) if r not in pinned] + ['"s%d"' % r for r in SGPRS] + \
        ['"scc"', '"vcc"']
    lines.append("        : " + ", ".join(clob) + ");")
    lines += ["}"]
    print("%s: %d VALU instructions + %d s_nop for the %d value%s" % (name, counts["valu"], counts["nop"], nch, "" if nch == 1 else "s"))
    CUR_NCH[0] = None
    return lines


def _regs_of(text):
    """(vgprs read, vgprs written, sgpr pairs read, sgpr pairs written) of one instruction, by operand position"""
    import re
    mnem, rest = text.split(None, 1)
    ops = [o.strip() for o in rest.split(",")]

    def expand(o):
        m = re.match(r"v\[(\d+):(\d+)\]$", o)
        if m:
            return [("v", r) for r in range(int(m.group(1)), int(m.group(2)) + 1)]
        m = re.match(r"v(\d+)$", o)
        if m:
            return [("v", int(m.group(1)))]
        m = re.match(r"s\[(\d+):(\d+)\]$", o)
        if m:
            return [("s", int(m.group(1)))]
        return []
    n_dst = 2 if mnem in ("v_mad_u64_u32", "v_sub_co_u32", "v_subb_co_u32", "v_subbrev_co_u32", "v_add_co_u32",
                          "v_addc_co_u32") else 1
    wr = [r for o in ops[:n_dst] for r in expand(o)]
    rd = [r for o in ops[n_dst:] for r in expand(o)]
    return rd, wr


LAT_MAD = int(os.environ.get("SSA_GEN_LAT_MAD", "0"))        # latency model of the list scheduler (0 = none): extra positions a
LAT_OTHER = int(os.environ.get("SSA_GEN_LAT_OTHER", "0"))    # reader of a multiply-add's / another VALU result should keep
VALU_GAP = int(os.environ.get("SSA_GEN_VALU_GAP", "3"))     # positions between a VALU write of an SGPR pair and the VALU read


def schedule(chains):
    """list scheduling of the chains' instructions together: dependencies from the registers (RAW, WAW, WAR), an
    SGPR pair written by a VALU instruction at position i is readable by a VALU instruction at i + 3 at the earliest
    (two wait states), longest remaining path first; s_nop only where nothing else is ready"""
    ins = [t for ch in chains for (t, _, _) in ch]
    info = [_regs_of(t) for t in ins]
    n = len(ins)
    preds = [set() for _ in range(n)]
    for j in range(n):
        rdj, wrj = info[j]
        for i in range(j):
            rdi, wri = info[i]
            if set(wri) & set(rdj) or set(wri) & set(wrj) or set(rdi) & set(wrj):
                if ("s", 24) in (set(wri) & set(wrj)) and not (set(wri) & set(rdj)) and not ((set(wri) & set(wrj)) - {("s", 24)}) \
                        and not (set(rdi) & set(wrj)):
                    continue            # the dummy carry-out pair orders nothing
                preds[j].add(i)
    height = [1] * n
    for j in range(n - 1, -1, -1):
        for i in preds[j]:
            height[i] = max(height[i], height[j] + 1)

    def lat(t):
        return LAT_MAD if t.startswith("v_mad_u64") else (LAT_OTHER if t.startswith("v_") else 0)

    def place(prio):
        placed_at = {}
        out = []
        last_sgpr_write = {}
        vready = {}          # VGPR -> first position at which a reader does not wait for it (latency model, LAT_* > 0)
        while len(placed_at) < n:
            best, best_key = None, None
            for j in range(n):
                if j in placed_at or any(i not in placed_at for i in preds[j]):
                    continue
                rd, wr = info[j]
                # the wait states are a VALU-write -> VALU-read matter (gap 3); the scalar unit's reads of a pair a VALU
                # instruction wrote are interlocked -- one other instruction is put in between anyway (gap 2)
                gap = VALU_GAP if ins[j].startswith("v_") else 2
                if any(r[0] == "s" and r[1] != 24 and len(out) - last_sgpr_write.get(r, -10) < gap for r in rd):
                    continue
                wait = max([vready.get(r, 0) - len(out) for r in rd if r[0] == "v"] + [0])
                key = (-wait, prio[j])
                if best is None or key > best_key:
                    best, best_key = j, key
            if best is None:
                out.append("s_nop 0")
                continue
            out.append(ins[best])
            placed_at[best] = len(out) - 1
            for r in info[best][1]:
                if r[0] == "s" and ins[best].startswith("v_"):
                    last_sgpr_write[r] = len(out) - 1
                if r[0] == "v":
                    vready[r] = len(out) - 1 + 1 + lat(ins[best])
        return out

    def stalls(seq):
        """positions a reader sits closer to the producer of one of its VGPR operands than the latency model wants"""
        ready, tot = {}, 0
        for pos, t in enumerate(seq):
            if t.startswith("s_nop"):
                continue
            rd, wr = _regs_of(t)
            tot += max([ready.get(r, 0) - pos for r in rd if r[0] == "v"] + [0])
            for r in wr:
                if r[0] == "v":
                    ready[r] = pos + 1 + lat(t)
        return tot

    # longest remaining path first; the greedy choice is not always the one without padding, so a few hundred
    # deterministic perturbations of the priorities are tried as well and the schedule with the fewest s_nop (then the
    # fewest modelled stalls) kept
    import random
    rnd = random.Random(len(ins) * 7919 + sum(len(t) for t in ins))
    cost = lambda seq: (sum(ln.startswith("s_nop") for ln in seq), stalls(seq))
    best_out = place([float(h) for h in height])
    for _ in range(300):
        if cost(best_out) == (0, 0):
            break
        cand = place([h + rnd.random() * 2.5 for h in height])
        if cost(cand) < cost(best_out):
            best_out = cand
    return best_out


OUT_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "schnorr-sig_amd", "csrc",
                        "fp_chain_asm.inc")


def generate():
    """the text of fp_chain_asm.inc (the freshness test compares it with the committed file without writing anything)"""
    lines = ["// generated by tools/gen_fp_chain_asm.py -- do not edit (see that file for the design notes)",
             "#define SSA_FP_CHAINS %d      // state elements per S-box block" % NCH, ""]
    progs = emit_program("inv_sbox_n_asm", INV_SBOX, "x <- x^(1/7) for %d values: the whole 63-squaring / 9-product chain of "
                         "all of them in one block, interleaved" % NCH) + [""] + \
        emit_program("sbox_n_asm", SBOX, "x <- x^7 for %d values" % NCH)
    # ONE value per block: the latency path (ssa_coop.hpp: the sponge state on 12 lanes of one wave, a lane per element).  A
    # lone chain cannot fill the wait states of its SGPR carries with another chain's instructions: s_nop where nothing
    # else is ready (4 per squaring) -- still 11 + 4 issue slots where the compiled squaring has 22 + 2.
    single = emit_program("inv_sbox_1_asm", INV_SBOX, "x <- x^(1/7) for ONE value (the cooperative kernels' sponge)", 1) + [""] + \
        emit_program("sbox_1_asm", SBOX, "x <- x^7 for ONE value", 1)
    return "\n".join(lines + progs + [""] + single) + "\n"


def main():
    text = generate()
    with open(OUT_PATH, "w") as fh:
        fh.write(text)
    print("wrote", OUT_PATH)


if __name__ == "__main__":
    main()
