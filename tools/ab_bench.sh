#!/bin/bash
# A/B of library variants on ONE box, alternating: ssa_k_hash ms, ssa_k_verify ms, M verifications/s per run.
#   tools/ab_bench.sh name=path/to/variant.so [name2=...]      (default library first in every round; 3 rounds)
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for spec in default= "$@"; do
    name=${spec%%=*}; lib=${spec#*=}
    SSA_LIB=$lib python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --skip-torsion-leg 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())
print('%-10s %.3f %.3f %.2f' % ('$name', j['kernels_ms']['ssa_k_hash'], j['kernels_ms']['ssa_k_verify'], j['value']/1e6))"
  done
done
