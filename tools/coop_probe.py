"""Latency of one cooperative point operation (a chain of 2000 dependent operations on one wave)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import schnorr_sig_amd as ssa
eng = ssa.Engine(0)
for v, name in ((10, "doubling"), (11, "mixed addition"), (12, "general addition"),
                (13, "ladder window (4 dbl + madd)")):
    best = max(eng.bench_fpmul(v) for _ in range(3))
    print("%-30s %.2f us" % (name, 1e6 / best))
