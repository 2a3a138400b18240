"""Register-resident Fp-mul throughput probe (the measured VALU ceiling the roofline uses)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import schnorr_sig_amd as ssa

eng = ssa.Engine(0)
names = {0: "fp_mul ILP1", 1: "fp_mul ILP4", 2: "fp_mul ILP8", 3: "f6_mul+f6_sqr (lazy)"}
out = {}
for v, name in names.items():
    best = max(eng.bench_fpmul(v) for _ in range(3))
    out[name] = best
    print("%-24s %.3e Fp-mul/s" % (name, best))
print(json.dumps(out))
