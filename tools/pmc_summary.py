"""Summarise rocprofv3 --pmc counter_collection.csv files per dispatch for the engine's kernels.

  python tools/pmc_summary.py DIR                      per-dispatch text summary of one pass
  python tools/pmc_summary.py --sq DIR --fetch DIR --write DIR --batch N --out profiles/rNN/hbm_traffic.json
      merges the separate passes (SQ counters; FETCH_SIZE; WRITE_SIZE + GRBM_GUI_ACTIVE -- they do not fit one pass,
      MI355X_MICROARCH.md "rocprofv3 PMC slots") into the file bench.py reads.  The file carries the sha256 of the
      libschnorr_sig_amd.so it was measured on and the batch size: bench.py prints roofline.traffic only when both
      match the library it has loaded, so the numbers cannot go stale silently.
HBM bytes per launch = FETCH_SIZE x 2 (gfx950 tallies the 128-B requests of 16-B-per-lane loads at 64 B: guide,
"HBM") + WRITE_SIZE, both in KiB in the CSV.
"""
import argparse
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "schnorr-sig_amd", "csrc", "libschnorr_sig_amd.so")
sys.path.insert(0, ROOT)


def _work_executed():
    """64x64 products per signature the kernels execute (bench.py's formulas)"""
    try:
        import bench
        return {"ssa_k_verify": bench.W_VERIFY_EXECUTED, "ssa_k_hash": bench.W_HASH}
    except Exception:
        return {}


WORK_EXECUTED = _work_executed()


def load(d):
    fs = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)
    if not fs:
        raise SystemExit("no *_counter_collection.csv under " + d)
    per = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        name = r["Kernel_Name"]
        if "ssa_k_" not in name and "msm_k_" not in name:
            continue
        key = "ssa_k_" if "ssa_k_" in name else "msm_k_"
        short = name[name.index(key):].split("(")[0].split("<")[0]
        e = per.setdefault((int(r["Dispatch_Id"]), short), {})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
        e["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        e["vgpr"] = r.get("VGPR_Count") or r.get("Arch_VGPR_Count")
        e["accum_vgpr"] = r.get("Accum_VGPR_Count")
        e["scratch"] = r.get("Scratch_Size")
        e["grid"] = r.get("Grid_Size")
    return per


def text(per):
    for k, v in per.items():
        w = v.get("SQ_WAVE_CYCLES")
        items = []
        for c, x in v.items():
            if c in ("ms", "vgpr", "accum_vgpr", "scratch", "grid"):
                continue
            items.append("%s=%.4g%s" % (c, x, " (%.0f%%)" % (100 * x / w) if w and c.startswith("SQ_") else ""))
        print(k, "ms=%.2f vgpr=%s accum=%s scratch=%s grid=%s" % (v["ms"], v["vgpr"], v["accum_vgpr"], v["scratch"], v["grid"]),
              " ".join(items))


def mean_over(per, kernel, grid=None):
    """average of every counter over the dispatches of `kernel` (optionally only those with a given grid size)"""
    rows = [v for (d, k), v in per.items() if k == kernel and (grid is None or str(v.get("grid")) == str(grid))]
    if not rows:
        return None
    out = {}
    for c in rows[0]:
        if c in ("vgpr", "accum_vgpr", "scratch", "grid"):
            out[c] = rows[0][c]
        else:
            out[c] = sum(r.get(c, 0.0) for r in rows) / len(rows)
    out["dispatches"] = len(rows)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir", nargs="?")
    ap.add_argument("--sq")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--batch", type=int, default=1 << 20)
    ap.add_argument("--out")
    ap.add_argument("--source", default="")
    args = ap.parse_args()
    if args.dir:
        text(load(args.dir))
        return
    sha = hashlib.sha256(open(LIB, "rb").read()).hexdigest()
    out = {"lib_sha256": sha, "batch": args.batch, "source": args.source,
           "note": "separate rocprofv3 --pmc passes; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts the 128-B "
                   "requests of 16-B/lane loads as 64 B), WRITE_SIZE as read; KiB in the CSV"}
    sq, fe, wr = load(args.sq), load(args.fetch), load(args.write)
    for kern in ("ssa_k_verify", "ssa_k_hash", "ssa_k_verify_keyed"):
        s, f, w = mean_over(sq, kern), mean_over(fe, kern), mean_over(wr, kern)
        if not (s and f and w):
            continue
        e = {"duration_ms": s["ms"], "vgpr": s["vgpr"], "accum_vgpr": s["accum_vgpr"], "scratch": s["scratch"],
             "FETCH_SIZE_KiB": f.get("FETCH_SIZE"), "WRITE_SIZE_KiB": w.get("WRITE_SIZE"),
             "GRBM_GUI_ACTIVE": w.get("GRBM_GUI_ACTIVE")}
        e["hbm_bytes_per_launch"] = int((2 * e["FETCH_SIZE_KiB"] + e["WRITE_SIZE_KiB"]) * 1024)
        for c, x in s.items():
            if c.startswith("SQ_"):
                e[c] = x
        if s.get("SQ_WAVE_CYCLES"):
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY"):
                if c in s:
                    e[c + "_frac"] = s[c] / s["SQ_WAVE_CYCLES"]
        if s.get("SQ_INSTS_VALU"):
            # wave-instructions per SIMD-cycle: 1024 SIMDs, duration x shader clock (GRBM_GUI_ACTIVE counts 8 XCDs)
            if e.get("GRBM_GUI_ACTIVE"):
                cycles = e["GRBM_GUI_ACTIVE"] / 8.0 * (s["ms"] / w["ms"])
                e["clock_GHz"] = e["GRBM_GUI_ACTIVE"] / 8.0 / (w["ms"] * 1e6)
                e["cycles_per_valu_instruction_per_simd"] = cycles * 1024 / s["SQ_INSTS_VALU"]
                # the names bench.py prints (roofline.pmc): VALU utilisation as issue cycles per instruction per SIMD
                # (4.0 = one wave64 instruction every pass of the 16-lane SIMD: back-to-back issue) and the share of
                # wave cycles in which the wave has a VALU instruction in flight
                e["valu_cycles_per_instruction"] = e["cycles_per_valu_instruction_per_simd"]
            if "SQ_ACTIVE_INST_VALU_frac" in e:
                e["valu_active_frac"] = e["SQ_ACTIVE_INST_VALU_frac"]
            wp = WORK_EXECUTED.get(kern)
            if wp:
                # VALU wave-instructions per lane per 64x64 product the kernel executes (4 = multiplies only)
                e["valu_instructions_per_lane"] = s["SQ_INSTS_VALU"] * 64.0 / args.batch
                e["valu_instructions_per_product"] = e["valu_instructions_per_lane"] / wp
            for c in ("SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64"):
                if c in s:
                    e[c + "_share"] = s[c] / s["SQ_INSTS_VALU"]
        out[kern] = e
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
