#!/bin/bash
# A/B of library variants AND runtime knobs on ONE box, alternating: ssa_k_hash ms, ssa_k_verify ms, M verifications/s.
#   tools/ab_env.sh [-r rounds] [-b "bench args"] "name|path/to/lib.so or empty|ENV=1 ENV2=2" ...
# (a spec with an empty library runs the in-tree one; the environment assignments apply to that run only)
cd "$(dirname "$0")/.."
ROUNDS=3
BARGS="--steps 5 --warmup 1"
while getopts "r:b:" o; do
  case $o in r) ROUNDS=$OPTARG;; b) BARGS=$OPTARG;; esac
done
shift $((OPTIND - 1))
for round in $(seq $ROUNDS); do
  for spec in "$@"; do
    IFS='|' read -r name lib envs <<< "$spec"
    env SSA_LIB=$lib $envs timeout -k 10 300 python3 bench.py $BARGS --no-cpu-baseline --skip-torsion-leg 2>/dev/null | python3 -c "
import json,sys
try:
    j=json.loads(sys.stdin.read())
    print('%-14s hash %.3f  verify %.3f  step %.3f ms  %.2f M/s  rejected %s' % ('$name', j['kernels_ms']['ssa_k_hash'], j['kernels_ms']['ssa_k_verify'], j['ms_per_step'], j['value']/1e6, j.get('rejected')))
except Exception as e:
    print('$name FAILED', e)"
  done
done
