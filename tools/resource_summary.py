#!/usr/bin/env python3
"""Compiler resource summary of every kernel of the library (the code object is authoritative for registers, scratch
and occupancy -- rocprofv3's VGPR_Count column is not): compiles the two translation units to gfx950
assembly and prints NumVgprs / NumAgprs / ScratchSize / Occupancy / code bytes per kernel.
    python3 tools/resource_summary.py > profiles/rNN/resource_summary.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "schnorr-sig_amd", "csrc")


def main():
    print("# hipcc -O3 --offload-arch=gfx950 -std=c++17 --cuda-device-only -S; per kernel: code bytes, VGPRs, AGPRs, "
          "scratch bytes/lane, LDS bytes/workgroup, waves/SIMD the registers allow")
    for unit in ("ssa_api.hip", "ssa_msm.hip"):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "u.s")
            subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "--cuda-device-only", "-S", "-o", out,
                                   os.path.join(CSRC, unit)], stderr=subprocess.DEVNULL)
            s = open(out).read()
        print("## " + unit)
        for m in re.finditer(r"^(_ZN3ssa\w+):.*?; codeLenInByte = (\d+).*?; NumVgprs: (\d+)\n; NumAgprs: (\d+).*?; ScratchSize: (\d+)"
                             r".*?; LDSByteSize: (\d+).*?; Occupancy: (\d+)", s, re.S | re.M):
            name = m.group(1)
            if "_k_" not in name:
                continue
            short = re.sub(r"^_ZN3ssa\d+", "", name)
            short = re.match(r"[a-z0-9_]+", short).group(0)
            print("%-28s code %7d B  vgpr %3d  agpr %3d  scratch %5d B  lds %6d B  waves/SIMD %d"
                  % (short, int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6)), int(m.group(7))))


if __name__ == "__main__":
    main()
