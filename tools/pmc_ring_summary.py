#!/usr/bin/env python3
"""Per-kernel means of the rocprofv3 --pmc passes of tools/pmc_ring.sh, with the quotients that say where sg::spmm_ring's
time goes.   python tools/pmc_ring_summary.py <dir with sq1/ sq2/ tc1/ trace/> <out.json>
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over all wavefronts;
SQ_WAIT_ANY + SQ_WAIT_INST_ANY + SQ_ACTIVE_INST_ANY ~ SQ_WAVE_CYCLES (disjoint buckets); SQ_VALU_MFMA_BUSY_CYCLES counts
cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import collections
import csv
import glob
import json
import re
import statistics
import sys


def main():
    root, out = sys.argv[1], sys.argv[2]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(root + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(spmm_ring<.+?>)\(", r["Kernel_Name"])
            if m:
                acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = {}
    for f in glob.glob(root + "/trace/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(spmm_ring<.+?>)\(", r["Name"])
            if m:
                dur[m.group(1)] = float(r["AverageNs"]) / 1e3
    res = {}
    for k, cs in sorted(acc.items()):
        d = {c: statistics.mean(v) for c, v in cs.items()}
        d["launches_seen"] = max(len(v) for v in cs.values())
        if k in dur:
            d["mean_duration_us (kernel trace, same command)"] = round(dur[k], 1)
        wc = d.get("SQ_WAVE_CYCLES")
        q = {}
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
                if c in d:
                    q[c + " / SQ_WAVE_CYCLES"] = round(d[c] / wc, 4)
        if d.get("SQ_LDS_IDX_ACTIVE"):
            q["SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"], 4)
        if d.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
            # matrix-core busy cycles summed over 256 CUs x 4 SIMDs against the wall cycles of the launch (GUI_ACTIVE / 8 XCDs)
            q["mfma_busy_fraction_of_all_SIMD_cycles"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
        if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
            q["L2_hit_rate"] = round(d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"]), 4)
        d["quotients"] = q
        res[k] = d
    json.dump({"command": "tools/pmc_ring.sh: rocprofv3 --pmc (three passes, one counter group each) + one --kernel-trace --stats pass "
                          "over python3 tools/agg_bench.py --channels 256 --dtypes bf16 --epilogue 1 --rounds 1 --reps 3 (V = 1 M, E = 6 M)",
               "kernels": res}, open(out, "w"), indent=1)
    print(json.dumps({k: v["quotients"] for k, v in res.items()}, indent=1))


if __name__ == "__main__":
    main()
