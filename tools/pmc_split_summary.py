#!/usr/bin/env python3
"""Per-kernel means of the rocprofv3 --pmc passes collected by tools/pmc_split.sh (split-bf16 fp32 products).
SQ_* cycle counters are quad-cycles summed over wavefronts; SQ_VALU_MFMA_BUSY_CYCLES counts cycles (MI355X guide)."""
import collections
import csv
import glob
import json
import re
import statistics
import sys


def main():
    root, out = sys.argv[1], sys.argv[2]
    pat = r"(gemm_[nt][nt]_f32s(<[^>]*>)?|Cijk_\w{4}_\w{4})" if len(sys.argv) < 4 else "(" + sys.argv[3] + r"|Cijk_\w{4}_\w{4})"
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(root + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(pat, r["Kernel_Name"])
            if m:
                key = m.group(1) + " grid=" + r.get("Grid_Size", "?")
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for f in glob.glob(root + "/trace/**/*_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(pat, r["Kernel_Name"])
            if m:
                grid = r.get("Grid_Size") or str(int(r.get("Grid_Size_X", 0)) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1))
                dur[m.group(1) + " grid=" + grid].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    res = {}
    for k, cs in sorted(acc.items()):
        d = {c: statistics.mean(v) for c, v in cs.items()}
        d["launches_seen"] = max(len(v) for v in cs.values())
        if k in dur:
            d["mean_duration_us"] = statistics.mean(dur[k])
        wc = d.get("SQ_WAVE_CYCLES")
        q = {}
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if c in d:
                    q[c + " / SQ_WAVE_CYCLES"] = round(d[c] / wc, 4)
        if "GRBM_GUI_ACTIVE" in d and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs
            q["mfma_busy_fraction_of_all_SIMD_cycles"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
            if "mean_duration_us" in d:
                q["clock_GHz (GRBM_GUI_ACTIVE / 8 / duration)"] = round(d["GRBM_GUI_ACTIVE"] / 8 / d["mean_duration_us"] / 1e3, 3)
        if d.get("SQ_LDS_IDX_ACTIVE"):
            q["SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_LDS_IDX_ACTIVE"], 4)
        d["quotients"] = q
        res[k] = d
    json.dump({"command": "tools/pmc_split.sh" if len(sys.argv) < 4 else "tools/pmc_gemm256.sh", "kernels": res}, open(out, "w"), indent=1)
    print("wrote", out, len(res), "kernels")


if __name__ == "__main__":
    main()
