#!/usr/bin/env python3
"""Compiler resource summary of every kernel of the library (the code object is authoritative for registers, scratch
and occupancy -- rocprofv3's VGPR_Count column is not): compiles the three translation units to gfx950
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
    for unit in ("ssa_api.hip", "ssa_msm.hip", "ssa_sign.hip"):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "u.s")
            subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "--cuda-device-only", "-S", "-o", out,
                                   os.path.join(CSRC, unit)], stderr=subprocess.DEVNULL)
            s = open(out).read()
        print("## " + unit)
        # one chunk per function label (a non-kernel function has no "Occupancy" line: a pattern spanning labels would
        # swallow the kernel that follows it)
        chunks = re.split(r"^(?=_ZN3ssa\w+:)", s, flags=re.M)
        for ch in chunks:
            m = re.match(r"(_ZN3ssa\w+):", ch)
            if not m or "_k_" not in m.group(1):
                continue
            r = re.search(r"; codeLenInByte = (\d+).*?; NumVgprs: (\d+)\n; NumAgprs: (\d+).*?; ScratchSize: (\d+)"
                          r".*?; LDSByteSize: (\d+).*?; Occupancy: (\d+)", ch, re.S)
            if not r:
                continue
            short = re.sub(r"^_ZN3ssa\d+", "", m.group(1))
            short = re.match(r"[a-z0-9_]+", short).group(0)
            print("%-28s code %7d B  vgpr %3d  agpr %3d  scratch %5d B  lds %6d B  waves/SIMD %d"
                  % (short, int(r.group(1)), int(r.group(2)), int(r.group(3)), int(r.group(4)), int(r.group(5)), int(r.group(6))))


if __name__ == "__main__":
    main()
