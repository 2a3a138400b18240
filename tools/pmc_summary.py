"""Summarise a rocprofv3 --pmc counter_collection.csv per dispatch for the engine's kernels."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
per = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    if "ssa_k_" not in name:
        continue
    short = name[name.index("ssa_k_"):].split("(")[0]
    key = (int(r["Dispatch_Id"]), short)
    e = per.setdefault(key, {})
    e[r["Counter_Name"]] = float(r["Counter_Value"])
    e["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    e["vgpr"] = r["VGPR_Count"]
    e["scratch"] = r["Scratch_Size"]
for k, v in per.items():
    w = v.get("SQ_WAVE_CYCLES")
    items = []
    for c, x in v.items():
        if c in ("ms", "vgpr", "scratch"):
            continue
        items.append("%s=%.3g%s" % (c, x, " (%.0f%%)" % (100 * x / w) if w and c.startswith("SQ_") else ""))
    print(k, "ms=%.2f vgpr=%s scratch=%s" % (v["ms"], v["vgpr"], v["scratch"]), " ".join(items))
