#!/bin/bash
# Profiles of one round, run ON THE GPU BOX (gpurun): kernel-trace stats of the default bench command, then three
# separate PMC passes (SQ; FETCH_SIZE; WRITE_SIZE + GRBM_GUI_ACTIVE) of the timed steps only, merged into
# hbm_traffic.json.  Output under gpurun_out/prof_$1/ -- copy what is to be judged into profiles/$1/.
# The program itself stands right after `--` (python3 bench.py ...): no env/bash/taskset hop under rocprofv3.
set -e
R=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp      # rocprofv3 wants a writable cwd/TMPDIR
BENCH="python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --skip-torsion-leg"
PMCB="python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --skip-torsion-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/bench_stats_run.json 2> $OUT/stats.err
echo "stats pass done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_full -- python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_full_run.json 2> $OUT/stats_full.err
echo "full-bench stats pass done"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- $PMCB > /dev/null 2> $OUT/pmc_sq.err
echo "sq pass done"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq2 -- $PMCB > /dev/null 2> $OUT/pmc_sq2.err || echo "sq2 pass failed (counters unavailable?)"
echo "sq2 pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $PMCB > /dev/null 2> $OUT/pmc_fetch.err
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_write -- $PMCB > /dev/null 2> $OUT/pmc_write.err
echo "write pass done"
python3 $REPO/tools/pmc_summary.py $OUT/pmc_sq > $OUT/pmc_sq.txt
python3 $REPO/tools/pmc_summary.py $OUT/pmc_sq2 > $OUT/pmc_sq2.txt || true
python3 $REPO/tools/pmc_summary.py $OUT/pmc_fetch > $OUT/pmc_fetch.txt
python3 $REPO/tools/pmc_summary.py $OUT/pmc_write > $OUT/pmc_write.txt
python3 $REPO/tools/pmc_summary.py --sq $OUT/pmc_sq --fetch $OUT/pmc_fetch --write $OUT/pmc_write --batch 1048576 \
    --source "rocprofv3 --kernel-trace --pmc ... -- $PMCB (tools/profile_round.sh $R)" --out $OUT/hbm_traffic.json
find $OUT -name "*kernel_stats.csv" | head
# keep only the small summaries in what is merged back
find $OUT -name "*counter_collection.csv" -size +2000k -delete
find $OUT -name "*kernel_trace.csv" -size +2000k -delete
ls -la $OUT
