#!/bin/bash
# HBM traffic of ssa_k_verify under runtime knobs, ON THE GPU BOX: one FETCH_SIZE pass and one WRITE_SIZE pass per
# configuration (separate rocprofv3 --pmc runs, the program itself right after `--`; the knobs are exported, not passed
# through env).   tools/traffic_probe.sh out_dir "name|ENV=1 ENV2=2" ...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$1
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  IFS='|' read -r name envs <<< "$spec"
  (
    for kv in $envs; do export "$kv"; done
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${name}_fetch -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --skip-torsion-leg > /dev/null 2> $OUT/${name}_fetch.err
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${name}_write -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --skip-torsion-leg > /dev/null 2> $OUT/${name}_write.err
  )
  f=$(python3 $REPO/tools/pmc_summary.py $OUT/${name}_fetch | grep "'ssa_k_verify'" | tail -1)
  w=$(python3 $REPO/tools/pmc_summary.py $OUT/${name}_write | grep "'ssa_k_verify'" | tail -1)
  echo "$name: $f"
  echo "$name: $w"
  find $OUT -name "*counter_collection.csv" -delete
  find $OUT -name "*kernel_trace.csv" -delete
done
