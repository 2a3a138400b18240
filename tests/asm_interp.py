"""A one-lane interpreter for the subset of gfx950 VALU/SALU instructions the generated asm blocks use
(schnorr-sig_amd/csrc/fp6_asm.inc, fp_chain_asm.inc).  Test infrastructure: lets the CPU suite execute the very
instruction strings the GPU runs.  Besides the arithmetic it checks the two things hand-written gfx950 asm gets
wrong silently:
  * a VALU write of an SGPR pair / VCC needs two other instructions before a VALU reads it as carry-in or mask
    (software-managed hazard: no interlock) -- `min_gap`;
  * a v_mad_u64_u32 whose carry-out goes to the "dummy" pair must not overflow.
"""
import re

M32, M64 = 0xFFFFFFFF, (1 << 64) - 1


def extract_blocks(path):
    """{function name: (asm lines, output names, {input name: (half, array, index) or C expression})} for every
    SSA_DEV function of the generated file that contains an asm statement"""
    txt = open(path).read()
    out = {}
    for m in re.finditer(r"SSA_DEV (?:void|u32) (\w+)\(.*?\) \{\n(.*?)\n\}\n", txt, re.S):
        name, body = m.group(1), m.group(2)
        am = re.search(r"asm(?: volatile)?\(\n(.*?)\n        : (.*?)\n        :(.*?)\n        : \"", body, re.S)
        if not am:
            continue
        lines = [ln.strip().strip('"').replace("\\n\\t", "") for ln in am.group(1).split("\n") if ln.strip()]
        outs = re.findall(r"\[(\w+)\] \"[=+]&?v\"", am.group(2))
        ins = {}
        for nm, half, arr, idx in re.findall(r"\[(\w+)\] \"v\"\((lo32|hi32)\((\w+)\[(\d+)\]\)\)", am.group(3)):
            ins[nm] = (half, arr, int(idx))
        out[name] = (lines, outs, ins)
    return out


class Lane:
    def __init__(self, env, dummy_pairs=(), min_gap=3):
        self.v = [0] * 256
        self.s = {}
        self.env = dict(env)
        self.dummy = set(dummy_pairs)
        self.min_gap = min_gap
        self.issued = 0
        self.written_at = {}
        self.salu32 = {}

    # ---- operands
    def rv(self, o):
        o = o.strip()
        if o in self.env:
            return self.env[o]
        m = re.match(r"v\[(\d+):(\d+)\]$", o)
        if m:
            a = int(m.group(1))
            return self.v[a] | (self.v[a + 1] << 32)
        m = re.match(r"v(\d+)$", o)
        if m:
            return self.v[int(m.group(1))]
        m = re.match(r"s\[(\d+):(\d+)\]$", o)
        if m:      # an SGPR pair as a 64-bit DATA operand (set by s_mov_b32)
            a = int(m.group(1))
            return self.salu32.get(a, 0) | (self.salu32.get(a + 1, 0) << 32)
        v = int(o, 0)
        return v & M32 if v < 0 else v

    def wv(self, o, val):
        o = o.strip()
        if o in self.env or o.startswith("%["):
            self.env[o] = val & M32
            return
        m = re.match(r"v\[(\d+):(\d+)\]$", o)
        if m:
            a = int(m.group(1))
            self.v[a], self.v[a + 1] = val & M32, (val >> 32) & M32
            return
        self.v[int(re.match(r"v(\d+)$", o).group(1))] = val & M32

    def _key(self, o):
        o = o.strip()
        if o == "vcc" or o.startswith("%["):        # %[name]: an SGPR-pair operand of the statement (a mask the caller owns)
            return o
        return int(re.match(r"s\[(\d+):(\d+)\]$", o).group(1))

    def wc(self, o, bit):
        k = self._key(o)
        self.s[k] = bit
        self.written_at[k] = self.issued

    def rc(self, o):
        k = self._key(o)
        gap = self.issued - self.written_at.get(k, -100)
        assert gap >= self.min_gap, "carry %s read %d slot(s) after its VALU write" % (o, gap)
        return self.s.get(k, 0)

    # ---- execution
    def run(self, lines):
        labels = {ln[:-1]: i for i, ln in enumerate(lines) if ln.endswith(":")}
        pc, scc = 0, 0
        while pc < len(lines):
            ln = lines[pc]
            pc += 1
            if ln.endswith(":"):
                self.visited = getattr(self, "visited", set())
                self.visited.add(ln[:-1])      # labels passed (tests check that a crafted input took a cold path)
                continue
            self.issued += 1
            if ln.startswith("s_nop"):
                self.issued += int(ln.split()[1])        # s_nop N = N + 1 wait states
                continue
            op, rest = ln.split(None, 1)
            op = op.replace("_e64", "").replace("_e32", "")
            a = [t.strip() for t in rest.split(",")]
            if op == "v_mov_b32":
                self.wv(a[0], self.rv(a[1]))
            elif op == "v_not_b32":
                self.wv(a[0], ~self.rv(a[1]) & M32)
            elif op == "s_mov_b32":
                self.salu32[int(a[0][1:])] = self.rv(a[1]) & M32
            elif op == "s_andn2_b64":       # on carry masks: one lane's bit (SALU read of a VALU-written pair is interlocked)
                kd, ka, kb = self._key(a[0]), self._key(a[1]), self._key(a[2])
                self.s[kd] = self.s.get(ka, 0) & (1 - self.s.get(kb, 0))
                self.written_at[kd] = -100  # written by the scalar unit: no VALU-write wait states
                scc = self.s[kd] != 0       # SCC = (result != 0): one lane's view of the wave-wide mask
            elif op == "s_sub_u32":
                k = int(a[0][1:])
                self.salu32[k] = (self.salu32[k] - self.rv(a[2])) & M32
            elif op == "s_cmp_lg_u32":
                scc = self.salu32[int(a[0][1:])] != self.rv(a[1])
            elif op == "s_cbranch_scc1":
                if scc:
                    pc = labels[a[0]]
            elif op == "s_cbranch_vccnz":    # one lane: the branch is taken when this lane's bit is set
                if self.s.get("vcc", 0):
                    pc = labels[a[0]]
            elif op == "s_and_saveexec_b64":   # one lane: EXEC is this lane's bit
                self.exec_saved = getattr(self, "exec_bit", 1)
                self.exec_bit = self.exec_saved & self.s.get(self._key(a[1]), 0)
            elif op == "s_cbranch_execz":
                if not getattr(self, "exec_bit", 1):
                    pc = labels[a[0]]
            elif op == "s_mov_b64":
                if a[0] == "exec":
                    self.exec_bit = self.exec_saved
                else:                       # a mask pair set to a constant by the scalar unit
                    kd = self._key(a[0])
                    self.s[kd] = 1 if int(a[1], 0) & 1 else 0
                    self.written_at[kd] = -100
            elif op == "s_or_b64":
                kd, ka, kb = self._key(a[0]), self._key(a[1]), self._key(a[2])
                self.s[kd] = self.s.get(ka, 0) | self.s.get(kb, 0)
                self.written_at[kd] = -100
                scc = self.s[kd] != 0
            elif op == "s_branch":
                pc = labels[a[0]]
            # ---- the fair-turn sequence of the ladder statements (tools/gen_jac_asm.py): a time slice of the wall clock XOR
            # the wave's slot picks the priority; one lane has no sibling to share a SIMD with, so the clock (`self.clock`,
            # settable by a test) and the slot (`self.slot`) are plain values and s_setprio is recorded
            elif op == "s_memrealtime":
                k = self._key(a[0])
                t = getattr(self, "clock", 0)
                self.salu32[k], self.salu32[k + 1] = t & M32, (t >> 32) & M32
            elif op == "s_waitcnt":
                pass
            # ---- the window statement gathers its own table entry: `self.mem[operand]` is the list of 32-bit words at the
            # address the operand holds (a test sets it); one lane, no latency: the load lands at once
            elif op == "global_load_dword":     # (the touches of jac_madd_gather_asm: a word loaded to be dropped)
                mo = re.match(r"off(?:\s+offset:(\d+))?$", a[2])
                words = self.mem[a[1]]
                self.v[int(re.match(r"v(\d+)$", a[0]).group(1))] = words[int(mo.group(1) or 0) // 4] & M32
            elif op == "global_load_dwordx4":
                m = re.match(r"v\[(\d+):(\d+)\]$", a[0])
                lo, hi = int(m.group(1)), int(m.group(2))
                assert hi == lo + 3 and lo % 2 == 0
                mo = re.match(r"off(?:\s+offset:(\d+))?$", a[2])
                off = int(mo.group(1) or 0)
                assert off % 4 == 0
                words = self.mem[a[1]]
                for k in range(4):
                    self.v[lo + k] = words[off // 4 + k] & M32
            elif op == "s_lshr_b32":
                self.salu32[int(a[0][1:])] = (self.salu32.get(int(a[1][1:]), 0) >> int(a[2], 0)) & M32
            elif op == "s_getreg_b32":
                assert "HW_REG_HW_ID" in rest
                self.salu32[int(a[0][1:])] = getattr(self, "slot", 0) & 1
            elif op == "s_xor_b32":
                self.salu32[int(a[0][1:])] = self.salu32.get(int(a[1][1:]), 0) ^ self.salu32.get(int(a[2][1:]), 0)
            elif op == "s_bitcmp1_b32":
                scc = (self.salu32.get(int(a[0][1:]), 0) >> int(a[1], 0)) & 1
            elif op == "s_setprio":
                self.prio_log = getattr(self, "prio_log", [])
                self.prio_log.append(int(a[0], 0))
            elif op == "s_and_b64":
                bit = lambda o: getattr(self, "exec_bit", 1) if o.strip() == "exec" else self.s.get(self._key(o), 0)
                kd = self._key(a[0])
                self.s[kd] = bit(a[1]) & bit(a[2])
                self.written_at[kd] = -100
                scc = self.s[kd] != 0
            elif op in ("v_cmp_le_u32", "v_cmp_eq_u32", "v_cmp_ne_u32"):
                x, y = self.rv(a[1]), self.rv(a[2])
                self.wc(a[0], int({"le": x <= y, "eq": x == y, "ne": x != y}[op[6:8]]))
            elif op == "v_mul_lo_u32":
                self.wv(a[0], (self.rv(a[1]) * self.rv(a[2])) & M32)
            elif op == "v_mul_hi_u32":
                self.wv(a[0], (self.rv(a[1]) * self.rv(a[2])) >> 32)
            elif op == "v_add_u32":
                self.wv(a[0], (self.rv(a[1]) + self.rv(a[2])) & M32)
            elif op == "v_or_b32":
                self.wv(a[0], self.rv(a[1]) | self.rv(a[2]))
            elif op == "v_xor_b32":
                self.wv(a[0], self.rv(a[1]) ^ self.rv(a[2]))
            elif op == "v_min_u32":
                self.wv(a[0], min(self.rv(a[1]), self.rv(a[2])))
            elif op == "v_max_u32":
                self.wv(a[0], max(self.rv(a[1]), self.rv(a[2])))
            elif op == "v_max3_u32":
                self.wv(a[0], max(self.rv(a[1]), self.rv(a[2]), self.rv(a[3])))
            elif op == "v_ashrrev_i32":
                x = self.rv(a[2])
                self.wv(a[0], ((x - (1 << 32) if x >> 31 else x) >> int(a[1])) & M32)
            elif op == "v_mad_u64_u32":
                r = self.rv(a[2]) * self.rv(a[3]) + self.rv(a[4])
                if a[1].strip() in self.dummy:
                    assert r <= M64, "a multiply-add with a discarded carry-out overflowed: " + ln
                else:
                    self.wc(a[1], r >> 64)
                self.wv(a[0], r & M64)
            elif op == "v_lshrrev_b32":
                self.wv(a[0], self.rv(a[2]) >> int(a[1]))
            elif op == "v_and_b32":
                self.wv(a[0], self.rv(a[1]) & self.rv(a[2]))
            elif op == "v_lshrrev_b64":
                self.wv(a[0], self.rv(a[2]) >> int(a[1]))
            elif op == "v_lshl_or_b32":
                self.wv(a[0], ((self.rv(a[1]) << int(a[2])) & M32) | self.rv(a[3]))
            elif op == "v_lshl_add_u64":
                self.wv(a[0], ((self.rv(a[1]) << int(a[2])) + self.rv(a[3])) & M64)
            elif op == "v_add_co_u32":
                t = self.rv(a[2]) + self.rv(a[3])
                self.wc(a[1], t >> 32)
                self.wv(a[0], t & M32)
            elif op == "v_addc_co_u32":
                t = self.rv(a[2]) + self.rv(a[3]) + self.rc(a[4])
                self.wc(a[1], t >> 32)
                self.wv(a[0], t & M32)
            elif op == "v_sub_co_u32":
                d = self.rv(a[2]) - self.rv(a[3])
                self.wc(a[1], 1 if d < 0 else 0)
                self.wv(a[0], d & M32)
            elif op == "v_subb_co_u32":
                d = self.rv(a[2]) - self.rv(a[3]) - self.rc(a[4])
                self.wc(a[1], 1 if d < 0 else 0)
                self.wv(a[0], d & M32)
            elif op == "v_subbrev_co_u32":
                d = self.rv(a[3]) - self.rv(a[2]) - self.rc(a[4])
                self.wc(a[1], 1 if d < 0 else 0)
                self.wv(a[0], d & M32)
            elif op == "v_cndmask_b32":
                self.wv(a[0], self.rv(a[2]) if self.rc(a[3]) else self.rv(a[1]))
            else:
                raise AssertionError("unknown instruction: " + ln)
        return self.env
