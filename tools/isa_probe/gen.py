#!/usr/bin/env python3
"""Generates tools/isa_probe/isa_probe.hip: a micro-benchmark of the ISSUE cost of single gfx950 VALU / SALU /
LDS instructions (cycles per wave-instruction per SIMD), the numbers the Fp / Fp6 instruction sequences are
designed against (DESIGN.md "instruction cost table").

Each probe is a kernel whose body is one inline-asm loop of R copies of a short pattern.  Blocks of 256*w
threads (w waves on each of the CU's 4 SIMDs; 96 KB of dynamic LDS pins one block per CU) are timed with HIP
events; cost = elapsed / (blocks per CU * iterations * instructions per wave * w), reported in ns and relative
to v_add_u32 at the same occupancy.
"""
import os

R = 48          # pattern copies per loop iteration
PROBES = []


def probe(name, pattern, n_instr=None, setup=""):
    """pattern: list of asm lines with {i} = copy index (used to rotate registers)."""
    PROBES.append((name, pattern, n_instr if n_instr is not None else len(pattern), setup))


def rot(i, base, n=8, stride=2):
    return base + stride * (i % n)


# independent streams: destinations rotate over 8 register (pairs) v[20..35]; sources v10..v17 fixed
probe("v_add_u32", ["v_add_u32 v{d}, v10, v11"])
probe("v_sub_u32", ["v_sub_u32 v{d}, v10, v11"])
probe("v_and_b32", ["v_and_b32 v{d}, v10, v11"])
probe("v_or_b32", ["v_or_b32 v{d}, v10, v11"])
probe("v_xor_b32", ["v_xor_b32 v{d}, v10, v11"])
probe("v_lshlrev_b32", ["v_lshlrev_b32 v{d}, 3, v10"])
probe("v_lshrrev_b32", ["v_lshrrev_b32 v{d}, 3, v10"])
probe("v_add_u32_e64", ["v_add_u32_e64 v{d}, v10, v11"])
probe("v_max_u32", ["v_max_u32 v{d}, v10, v11"])
probe("v_min_u32", ["v_min_u32 v{d}, v10, v11"])
probe("v_bfe_u32", ["v_bfe_u32 v{d}, v10, 3, 7"])
probe("v_add_lshl_u32", ["v_add_lshl_u32 v{d}, v10, v11, 3"])
probe("v_lshl_add_u32", ["v_lshl_add_u32 v{d}, v10, 3, v11"])
probe("v_xad_u32", ["v_xad_u32 v{d}, v10, v11, v12"])
probe("v_sad_u32", ["v_sad_u32 v{d}, v10, v11, v12"])
probe("v_cndmask_b32_vcc_e32", ["v_cndmask_b32 v{d}, v10, v11, vcc"], setup="vcc")
probe("v_add3_u32", ["v_add3_u32 v{d}, v10, v11, v12"])
probe("v_lshl_or_b32", ["v_lshl_or_b32 v{d}, v10, 3, v11"])
probe("v_and_or_b32", ["v_and_or_b32 v{d}, v10, v11, v12"])
probe("v_alignbit_b32", ["v_alignbit_b32 v{d}, v10, v11, 7"])
probe("v_mov_b32", ["v_mov_b32 v{d}, v10"])
probe("v_add_co_u32_vcc", ["v_add_co_u32 v{d}, vcc, v10, v11"])
probe("v_add_co_u32_sgpr", ["v_add_co_u32 v{d}, s[{s}:{s1}], v10, v11"])
probe("v_addc_co_u32_vcc", ["v_addc_co_u32 v{d}, vcc, v10, v11, vcc"])
probe("v_addc_co_u32_sgpr", ["v_addc_co_u32 v{d}, s[{s}:{s1}], v10, v11, s[{s}:{s1}]"])
probe("v_addc_co_u32_sgpr_fixed_in", ["v_addc_co_u32 v{d}, s[{s}:{s1}], v10, v11, s[60:61]"])
probe("v_sub_co_u32_vcc", ["v_sub_co_u32 v{d}, vcc, v10, v11"])
probe("v_subb_co_u32_vcc", ["v_subb_co_u32 v{d}, vcc, v10, v11, vcc"])
probe("v_lshl_add_u64", ["v_lshl_add_u64 v[{d}:{d1}], v[10:11], 0, v[12:13]"])
probe("v_lshlrev_b64", ["v_lshlrev_b64 v[{d}:{d1}], 5, v[10:11]"])
probe("v_lshrrev_b64", ["v_lshrrev_b64 v[{d}:{d1}], 5, v[10:11]"])
probe("v_cmp_lt_u32_vcc", ["v_cmp_lt_u32 vcc, v10, v11"])
probe("v_cmp_lt_u64_vcc", ["v_cmp_lt_u64 vcc, v[10:11], v[12:13]"])
probe("v_cmp_lt_u64_sgpr", ["v_cmp_lt_u64 s[{s}:{s1}], v[10:11], v[12:13]"])
probe("v_cndmask_b32_vcc", ["v_cndmask_b32 v{d}, v10, v11, vcc"])
probe("v_cndmask_b32_sgpr", ["v_cndmask_b32 v{d}, v10, v11, s[60:61]"])
probe("v_mad_u64_u32_indep", ["v_mad_u64_u32 v[{d}:{d1}], vcc, v10, v11, v[12:13]"])
probe("v_mad_u64_u32_sgprco", ["v_mad_u64_u32 v[{d}:{d1}], s[{s}:{s1}], v10, v11, v[12:13]"])
probe("v_mad_u64_u32_acc", ["v_mad_u64_u32 v[{d}:{d1}], vcc, v10, v11, v[{d}:{d1}]"])
probe("v_mad_u64_u32_zeroadd", ["v_mad_u64_u32 v[{d}:{d1}], vcc, v10, v11, 0"])
probe("v_mul_lo_u32", ["v_mul_lo_u32 v{d}, v10, v11"])
probe("v_mul_hi_u32", ["v_mul_hi_u32 v{d}, v10, v11"])
probe("v_mul_u32_u24", ["v_mul_u32_u24 v{d}, v10, v11"])
probe("v_mul_hi_u32_u24", ["v_mul_hi_u32_u24 v{d}, v10, v11"])
probe("v_mad_u32_u24", ["v_mad_u32_u24 v{d}, v10, v11, v12"])
probe("v_mad_u32_u16", ["v_mad_u32_u16 v{d}, v10, v11, v12"])
probe("v_dot4_u32_u8", ["v_dot4_u32_u8 v{d}, v10, v11, v12"])
probe("v_dot2_u32_u16", ["v_dot2_u32_u16 v{d}, v10, v11, v12"])
probe("v_pk_mul_lo_u16", ["v_pk_mul_lo_u16 v{d}, v10, v11"])
probe("v_pk_mad_u16", ["v_pk_mad_u16 v{d}, v10, v11, v12"])
probe("v_fma_f32", ["v_fma_f32 v{d}, v10, v11, v12"])
probe("v_pk_fma_f32", ["v_pk_fma_f32 v[{d}:{d1}], v[10:11], v[12:13], v[14:15]"])
probe("v_fma_f64", ["v_fma_f64 v[{d}:{d1}], v[10:11], v[12:13], v[14:15]"])
probe("v_add_f64", ["v_add_f64 v[{d}:{d1}], v[10:11], v[12:13]"])
probe("v_mul_f64", ["v_mul_f64 v[{d}:{d1}], v[10:11], v[12:13]"])
probe("v_cvt_f64_u32", ["v_cvt_f64_u32 v[{d}:{d1}], v10"])
probe("v_cvt_u32_f64", ["v_cvt_u32_f64 v{d}, v[10:11]"])
probe("v_accvgpr_write", ["v_accvgpr_write_b32 a{a}, v10"])
probe("v_accvgpr_read", ["v_accvgpr_read_b32 v{d}, a{a}"])
probe("s_xor_b64", ["s_xor_b64 s[{s}:{s1}], s[60:61], s[62:63]"])
probe("s_and_b64", ["s_and_b64 s[{s}:{s1}], s[60:61], s[62:63]"])
probe("ds_read_b64", ["ds_read_b64 v[{d}:{d1}], v18", ], setup="lds")
probe("ds_read_b128", ["ds_read_b128 v[{q}:{q3}], v18"], setup="lds")
probe("ds_write_b64", ["ds_write_b64 v18, v[10:11]"], setup="lds")
probe("ds_write_b128", ["ds_write_b128 v18, v[10:13]"], setup="lds")
# dependent chains
probe("dep_v_add_u32", ["v_add_u32 v20, v20, v11"])
probe("dep_v_mad_u64_u32", ["v_mad_u64_u32 v[20:21], vcc, v10, v11, v[20:21]"])
probe("dep_v_lshl_add_u64", ["v_lshl_add_u64 v[20:21], v[20:21], 0, v[12:13]"])
probe("dep_addc_chain_vcc", ["v_add_co_u32 v20, vcc, v20, v11", "v_addc_co_u32 v21, vcc, v21, v12, vcc"])
probe("dep_addc_chain_sgpr", ["v_add_co_u32 v20, s[40:41], v20, v11", "v_addc_co_u32 v21, s[40:41], v21, v12, s[40:41]"])
# the multiply-accumulate patterns of fp_acc: mad with carry-out + carry add into a counter
probe("mac_pattern_vcc", ["v_mad_u64_u32 v[{d}:{d1}], vcc, v10, v11, v[{d}:{d1}]", "v_addc_co_u32 v{k}, vcc, 0, v{k}, vcc"])
probe("mac_pattern_sgpr_gap", ["v_mad_u64_u32 v[20:21], s[40:41], v10, v11, v[20:21]",
                               "v_mad_u64_u32 v[22:23], s[42:43], v10, v12, v[22:23]",
                               "v_mad_u64_u32 v[24:25], s[44:45], v13, v11, v[24:25]",
                               "v_addc_co_u32 v36, s[40:41], 0, v36, s[40:41]",
                               "v_addc_co_u32 v37, s[42:43], 0, v37, s[42:43]",
                               "v_addc_co_u32 v38, s[44:45], 0, v38, s[44:45]"])
# carries counted on the scalar unit instead: bit-sliced counter planes (s_xor/s_and pairs) beside the mads
probe("mac_pattern_salu_count", ["v_mad_u64_u32 v[20:21], s[40:41], v10, v11, v[20:21]",
                                 "v_mad_u64_u32 v[22:23], s[42:43], v10, v12, v[22:23]",
                                 "s_and_b64 s[46:47], s[48:49], s[40:41]",
                                 "s_xor_b64 s[48:49], s[48:49], s[40:41]",
                                 "s_xor_b64 s[50:51], s[50:51], s[46:47]",
                                 "s_and_b64 s[52:53], s[54:55], s[42:43]",
                                 "s_xor_b64 s[54:55], s[54:55], s[42:43]",
                                 "s_xor_b64 s[56:57], s[56:57], s[52:53]"], n_instr=2)
probe("mads_plus_4salu", ["v_mad_u64_u32 v[{d}:{d1}], vcc, v10, v11, v[12:13]", "s_xor_b64 s[40:41], s[60:61], s[62:63]",
                          "s_and_b64 s[42:43], s[60:61], s[62:63]", "s_xor_b64 s[44:45], s[60:61], s[62:63]",
                          "s_and_b64 s[46:47], s[60:61], s[62:63]"], n_instr=1)
probe("add_plus_1salu", ["v_add_u32 v{d}, v10, v11", "s_xor_b64 s[40:41], s[60:61], s[62:63]"], n_instr=1)
# mixes: 1 mad + k plain ops (is the multiplier pipelined beside the adder?)
probe("mad_plus_1add", ["v_mad_u64_u32 v[{d}:{d1}], vcc, v10, v11, v[12:13]", "v_add_u32 v36, v10, v11"])
probe("mad_plus_2add", ["v_mad_u64_u32 v[{d}:{d1}], vcc, v10, v11, v[12:13]", "v_add_u32 v36, v10, v11", "v_add_u32 v37, v10, v11"])
probe("mad_plus_3add", ["v_mad_u64_u32 v[{d}:{d1}], vcc, v10, v11, v[12:13]", "v_add_u32 v36, v10, v11", "v_add_u32 v37, v10, v11", "v_add_u32 v38, v10, v12"])
probe("mad_plus_3and", ["v_mad_u64_u32 v[{d}:{d1}], vcc, v10, v11, v[12:13]", "v_and_b32 v36, v10, v11", "v_and_b32 v37, v10, v11", "v_and_b32 v38, v10, v12"])
probe("addc_plus_3add", ["v_addc_co_u32 v{d}, s[{s}:{s1}], v10, v11, s[60:61]", "v_add_u32 v36, v10, v11", "v_add_u32 v37, v10, v11", "v_add_u32 v38, v10, v12"])
probe("fma64_plus_3add", ["v_fma_f64 v[{d}:{d1}], v[10:11], v[12:13], v[14:15]", "v_add_u32 v36, v10, v11", "v_add_u32 v37, v10, v11", "v_add_u32 v38, v10, v12"])
probe("mad4_then_add4", ["v_mad_u64_u32 v[20:21], vcc, v10, v11, v[12:13]", "v_mad_u64_u32 v[22:23], vcc, v10, v11, v[12:13]",
                         "v_mad_u64_u32 v[24:25], vcc, v10, v11, v[12:13]", "v_mad_u64_u32 v[26:27], vcc, v10, v11, v[12:13]",
                         "v_add_u32 v36, v10, v11", "v_add_u32 v37, v10, v11", "v_add_u32 v38, v10, v12", "v_add_u32 v39, v10, v12"])
probe("mad_addc_pairs", ["v_mad_u64_u32 v[20:21], s[40:41], v10, v11, v[20:21]", "v_mad_u64_u32 v[22:23], s[42:43], v10, v12, v[22:23]",
                         "v_addc_co_u32 v36, s[40:41], 0, v36, s[40:41]", "v_addc_co_u32 v37, s[42:43], 0, v37, s[42:43]"])
# run lengths: N multiply-accumulates (carry-out to N distinct SGPR pairs), then the N carry adds -- does the switch
# between the multiplier pipe and the plain ALU cost issue slots?  (R = 48 copies of the whole group.)
for _n in (1, 2, 4, 8, 12):
    _pat = []
    for _j in range(_n):
        _pat.append("v_mad_u64_u32 v[%d:%d], s[%d:%d], v10, v11, v[%d:%d]" % (20 + 2 * (_j % 8), 21 + 2 * (_j % 8), 38 + 2 * _j, 39 + 2 * _j, 20 + 2 * (_j % 8), 21 + 2 * (_j % 8)))
    for _j in range(_n):
        _pat.append("v_addc_co_u32 v%d, s[%d:%d], 0, v%d, s[%d:%d]" % (36 + (_j % 4), 38 + 2 * _j, 39 + 2 * _j, 36 + (_j % 4), 38 + 2 * _j, 39 + 2 * _j))
    if _n == 1:
        _pat.insert(1, "s_nop 1")
    elif _n == 2:
        _pat.insert(2, "s_nop 0")
    probe("run%d_mad_then_addc" % _n, _pat, n_instr=2 * _n)
for _n in (2, 4, 8):
    _pat = ["v_mad_u64_u32 v[%d:%d], vcc, v10, v11, v[12:13]" % (20 + 2 * (_j % 8), 21 + 2 * (_j % 8)) for _j in range(_n)]
    _pat += ["v_add_u32 v%d, v10, v11" % (36 + (_j % 4)) for _j in range(_n)]
    probe("run%d_mad_then_add" % _n, _pat)
for _n in (2, 4, 8):
    _pat = ["v_mad_u64_u32 v[%d:%d], vcc, v10, v11, v[12:13]" % (20 + 2 * (_j % 8), 21 + 2 * (_j % 8)) for _j in range(_n)]
    _pat += ["v_lshl_add_u64 v[%d:%d], v[10:11], 0, v[12:13]" % (36 + 2 * (_j % 2), 37 + 2 * (_j % 2)) for _j in range(_n)]
    probe("run%d_mad_then_lshladd64" % _n, _pat)
probe("cmp64_cndmask_lshladd", ["v_lshl_add_u64 v[20:21], v[10:11], 0, v[12:13]", "v_cmp_lt_u64 s[40:41], v[20:21], v[10:11]",
                                 "v_lshl_add_u64 v[22:23], v[14:15], 0, v[12:13]", "v_cmp_lt_u64 s[42:43], v[22:23], v[14:15]",
                                 "v_cndmask_b32 v36, 0, -1, s[40:41]", "v_cndmask_b32 v37, 0, -1, s[42:43]"])
probe("ds_read_b128_plus_4add", ["ds_read_b128 v[{q}:{q3}], v18", "v_add_u32 v36, v10, v11", "v_add_u32 v37, v10, v11", "v_add_u32 v38, v10, v12", "v_add_u32 v39, v10, v12"], n_instr=5, setup="lds")


# ---- round 4 (second session): one Fp6 coefficient's accumulation (6 products = 24 multiply-adds on three 64-bit columns, a
# carry per multiply) + a stand-in for its reduction.  coef_valu: every carry by v_addc (the shipped form, 59 VALU).
# coef_salu_c1: the 12 carries of the middle column counted on the SCALAR unit instead (bit-sliced counter planes P1, P2,
# P4, P8 in SGPR pairs, full-adder steps interleaved with the next product's multiplies), converted to a VGPR count by
# 4 v_cndmask + 2 adds: 53 VALU + 47 SALU.  Does the scalar unit take the work for free (other wave's issue slot)?
def _reduction_standin():
    return ["v_add_co_u32 v27, s[52:53], v27, v28", "v_addc_co_u32 v29, s[52:53], v29, v30, s[52:53]",
            "v_add_co_u32 v29, s[54:55], v29, v36", "v_addc_co_u32 v31, s[52:53], v31, v37, s[52:53]",
            "v_addc_co_u32 v31, s[54:55], 0, v31, s[54:55]", "v_addc_co_u32 v38, s[52:53], 0, v38, s[52:53]",
            "v_addc_co_u32 v38, s[54:55], 0, v38, s[54:55]",
            "v_mad_u64_u32 v[32:33], s[52:53], v29, -1, v[26:27]", "v_subb_co_u32 v32, s[54:55], v32, v31, s[52:53]",
            "v_subbrev_co_u32 v34, s[56:57], 0, v38, s[52:53]", "v_subb_co_u32 v33, s[54:55], v33, v34, s[54:55]"]


def _coef_valu():
    out = []
    for i in range(6):
        out += ["v_mad_u64_u32 v[28:29], vcc, v10, v13, v[28:29]", "v_mad_u64_u32 v[26:27], s[40:41], v10, v12, v[26:27]",
                "v_mad_u64_u32 v[28:29], s[44:45], v11, v12, v[28:29]", "v_mad_u64_u32 v[30:31], s[42:43], v11, v13, v[30:31]",
                "v_addc_co_u32 v37, vcc, 0, v37, vcc", "v_addc_co_u32 v36, s[40:41], 0, v36, s[40:41]",
                "v_addc_co_u32 v37, s[44:45], 0, v37, s[44:45]", "v_addc_co_u32 v38, s[42:43], 0, v38, s[42:43]"]
    return out + _reduction_standin()


def _coef_salu():
    out = []
    P1, P2, P4, P8 = "s[46:47]", "s[48:49]", "s[50:51]", "s[58:59]"
    T, M, U, C, D, E = "s[38:39]", "s[52:53]", "s[54:55]", "s[56:57]", "s[52:53]", "s[54:55]"

    def salu(i):       # fold the two middle-column carries of product i into the counter planes
        a, b = ("s[40:41]", "s[42:43]") if i % 2 == 0 else ("s[44:45]", "vcc")
        if i == 0:
            return ["s_xor_b64 %s, %s, %s" % (P1, a, b), "s_and_b64 %s, %s, %s" % (P2, a, b)]
        ops = ["s_xor_b64 %s, %s, %s" % (T, a, b), "s_and_b64 %s, %s, %s" % (M, a, b), "s_and_b64 %s, %s, %s" % (U, P1, T),
               "s_xor_b64 %s, %s, %s" % (P1, P1, T), "s_or_b64 %s, %s, %s" % (C, M, U)]
        if i == 1:
            return ops + ["s_and_b64 %s, %s, %s" % (P4, P2, C), "s_xor_b64 %s, %s, %s" % (P2, P2, C)]
        ops += ["s_and_b64 %s, %s, %s" % (D, P2, C), "s_xor_b64 %s, %s, %s" % (P2, P2, C)]
        if i == 2:
            return ops + ["s_xor_b64 %s, %s, %s" % (P4, P4, D)]
        if i == 3:
            return ops + ["s_and_b64 %s, %s, %s" % (P8, P4, D), "s_xor_b64 %s, %s, %s" % (P4, P4, D)]
        return ops + ["s_and_b64 %s, %s, %s" % (E, P4, D), "s_xor_b64 %s, %s, %s" % (P4, P4, D), "s_xor_b64 %s, %s, %s" % (P8, P8, E)]

    pend = []
    for i in range(6):
        a, b = ("s[40:41]", "s[42:43]") if i % 2 == 0 else ("s[44:45]", "vcc")
        valu = ["v_mad_u64_u32 v[28:29], %s, v10, v13, v[28:29]" % a, "v_mad_u64_u32 v[26:27], s[60:61], v10, v12, v[26:27]",
                "v_mad_u64_u32 v[28:29], %s, v11, v12, v[28:29]" % b, "v_mad_u64_u32 v[30:31], s[62:63], v11, v13, v[30:31]",
                "v_addc_co_u32 v36, s[60:61], 0, v36, s[60:61]", "v_addc_co_u32 v38, s[62:63], 0, v38, s[62:63]"]
        # interleave the scalar work of the PREVIOUS product between this product's vector instructions
        for j, v in enumerate(valu):
            out.append(v)
            take = (len(pend) + (len(valu) - j) - 1) // (len(valu) - j)
            out += pend[:take]
            pend = pend[take:]
        pend = salu(i)
    out += pend
    out += ["v_cndmask_b32 v37, 0, 1, %s" % P1, "v_cndmask_b32 v34, 0, 2, %s" % P2, "v_cndmask_b32 v35, 0, 4, %s" % P4,
            "v_cndmask_b32 v39, 0, 8, %s" % P8, "v_add3_u32 v37, v37, v34, v35", "v_add_u32 v37, v37, v39"]
    return out + _reduction_standin()


probe("coef_valu", _coef_valu(), n_instr=59)
probe("coef_salu_c1", _coef_salu(), n_instr=59)        # reported per instruction of the SHIPPED form: directly comparable


# ---- round 4 (second session): the Rescue squaring (tools/gen_fp_chain_asm.py) on TWO against THREE interleaved chains:
# does a third independent chain buy issue slots (the S-box blocks are sensitive to the distance between dependent
# instructions: profiles/r04/hash_ab.txt (g))?  Cycles per instruction; 11 instructions per squaring and chain.
def _sq_chains(n):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import gen_fp_chain_asm as g

    def regs_n(c):
        pair = lambda k: 40 + 2 * c + 2 * n * k
        d = {"X": pair(0), "T": pair(1), "A": pair(2), "U": pair(3), "H": pair(4), "R": pair(5), "M": pair(6), "E": pair(7),
             "C": pair(8), "S": 38 + 2 * c}
        return d
    old = (g.regs, g.DUMMY)
    g.regs, g.DUMMY = regs_n, "s[60:61]"
    try:
        chains = [g.square(c, None, None, "s[62:63]") for c in range(n)]
        seq = g.schedule(chains)
    finally:
        g.regs, g.DUMMY = old
    assert not any(ln.startswith("s_nop") for ln in seq), seq
    return seq


for _n in (1, 2, 3, 4, 5, 6):
    try:
        probe("sq_chains_%d" % _n, _sq_chains(_n), n_instr=11 * _n, setup="chains")
    except AssertionError:
        pass


def body(pattern):
    lines = []
    for i in range(R):
        d = 20 + 2 * (i % 8)
        q = 20 + 4 * (i % 4)
        s = 40 + 2 * (i % 8)
        fm = dict(i=i, d=d, d1=d + 1, q=q, q3=q + 3, s=s, s1=s + 1, k=36 + (i % 4), a=i % 16)
        for ln in pattern:
            lines.append(ln.format(**fm))
    return lines


def main():
    out = ["// generated by tools/isa_probe/gen.py -- do not edit", "#include <hip/hip_runtime.h>", "#include <cstdio>",
           "#include <cstring>", "#include <vector>", ""]
    clob = ", ".join('"v%d"' % r for r in range(10, 128)) + ", " + ", ".join('"s%d"' % r for r in range(36, 64)) + \
        ', "vcc", "scc", "memory"'
    for name, pattern, n_instr, setup in PROBES:
        lines = body(pattern)
        out.append("__global__ void __launch_bounds__(1024) k_%s(int iters, unsigned *out) {" % name)
        out.append("    extern __shared__ unsigned lds[];")
        out.append("    unsigned x = threadIdx.x, y;")
        out.append("    asm volatile(")
        pre = ["v_mov_b32 v%d, %%1" % r for r in range(10, 40)]
        pre += ["v_lshlrev_b32 v18, 4, %1", "s_mov_b64 vcc, exec", "s_mov_b64 s[60:61], exec", "s_mov_b64 s[62:63], 0", "s_mov_b32 s36, %2"]
        pre += ["s_mov_b64 s[%d:%d], 0" % (s, s + 1) for s in range(40, 60, 2)]
        if setup == "chains":       # the chains' value pairs = the lane index (any value), every other register 0
            pre += ["v_mov_b32 v%d, 0" % r for r in range(40, 128)]
            pre += ["v_mov_b32 v%d, %%1" % r for r in range(40, 46)]
        for ln in pre:
            out.append('        "%s\\n\\t"' % ln)
        out.append('        "L_%s_%%=:\\n\\t"' % name)
        for ln in lines:
            out.append('        "%s\\n\\t"' % ln)
        if setup == "lds":
            out.append('        "s_waitcnt lgkmcnt(0)\\n\\t"')
        out.append('        "s_sub_u32 s36, s36, 1\\n\\t"')
        out.append('        "s_cmp_lg_u32 s36, 0\\n\\t"')
        out.append('        "s_cbranch_scc1 L_%s_%%=\\n\\t"' % name)
        out.append('        "v_xor_b32 %0, v20, v21"')
        out.append('        : "=v"(y) : "v"(x), "s"(iters) : %s);' % clob)
        out.append("    if (y == 0x12345u) out[threadIdx.x] = y + lds[0];")
        out.append("}")
        out.append("")
    out.append("struct Probe { const char *name; void (*fn)(int, unsigned *); int per_iter; };")
    out.append("static const Probe probes[] = {")
    for name, pattern, n_instr, setup in PROBES:
        out.append('    {"%s", k_%s, %d},' % (name, name, R * n_instr))
    out.append("};")
    out.append(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "main.inc")).read())
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "isa_probe.hip"), "w") as fh:
        fh.write("\n".join(out) + "\n")
    print("%d probes" % len(PROBES))


if __name__ == "__main__":
    main()
