#!/bin/bash
# Dynamic check of the constant-time signer ON THE GPU BOX: VALU / SALU / VMEM / LDS instruction counts of ssa_k_sign_ct
# for three secret sets (tools/sign_ct_probe.py), and of the throughput signer ssa_k_sign for contrast.
#   gpurun -- 'bash tools/sign_ct_pmc.sh'   ->  gpurun_out/sign_ct_pmc/summary.txt (copy to profiles/rNN/)
# The program itself stands right after `--` (python3 ...): no env/bash hop under rocprofv3.
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/sign_ct_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in ct vartime; do
  for s in a b c; do
    extra=""; [ $mode = vartime ] && extra="--vartime"
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM \
      --output-format csv -d $OUT/${mode}_$s -- python3 $REPO/tools/sign_ct_probe.py --set $s $extra > $OUT/${mode}_$s.log 2> $OUT/${mode}_$s.err
    echo "$mode $s done"
  done
done
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import collections, csv, glob, sys
out = sys.argv[1]
print("# rocprofv3 --pmc SQ_INSTS_*: wave-instruction counts per launch of the signing kernel, three secret sets")
print("# (a random, b sparse: sk = lane + 1 / nonce = 2^252 + lane + 1, c dense: nearly every 4-bit window 15); n = 4096 lanes, same messages")
rows = {}
for mode, kern in (("ct", "ssa_k_sign_ct"), ("vartime", "ssa_k_sign")):
    for s in "abc":
        fs = glob.glob("%s/%s_%s/**/*_counter_collection.csv" % (out, mode, s), recursive=True)
        acc = collections.OrderedDict()
        for r in csv.DictReader(open(fs[0])):
            name = r["Kernel_Name"]
            short = name.split("(")[0].split("::")[-1]
            if short != kern:
                continue
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        rows[(mode, s)] = acc
        print("%-8s set %s  %s" % (mode, s, "  ".join("%s=%d" % (k, v) for k, v in acc.items())))
ct = [rows[("ct", s)] for s in "abc"]
same = all(ct[0].get(k) == c.get(k) for c in ct[1:] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"))
vt = [rows[("vartime", s)]["SQ_INSTS_VALU"] for s in "abc"]
print("constant-time signer: counters identical across the three secret sets: %s" % same)
print("throughput signer:    SQ_INSTS_VALU differs across the sets: %s (%s)" % (len(set(vt)) > 1, vt))
PY
find $OUT -name "*.csv" -size +500k -delete
