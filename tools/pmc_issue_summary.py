#!/usr/bin/env python3
"""Per-kernel means of the rocprofv3 --pmc passes collected by tools/pmc_issue.sh.

    python tools/pmc_issue_summary.py <dir with sq1/ sq2/ tc1/ tc2/> <out.json>
SQ_* cycle counters are in quad-cycles summed over all wavefronts (MI355X_MICROARCH.md, cycle constants);
SQ_WAIT_ANY + SQ_WAIT_INST_ANY + SQ_ACTIVE_INST_ANY ~ SQ_WAVE_CYCLES (disjoint buckets)."""
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
            m = re.search(r"(spmm_\w+<.+?>)\(", r["Kernel_Name"])
            if m:
                acc[m.group(1).replace("sg::(anonymous namespace)::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, cs in sorted(acc.items()):
        d = {c: statistics.mean(v) for c, v in cs.items()}
        d["launches_seen"] = max(len(v) for v in cs.values())
        wc = d.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM"):
                if c in d:
                    d[c + "/SQ_WAVE_CYCLES"] = round(d[c] / wc, 4)
        if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
            d["L2_hit_rate"] = round(d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"]), 4)
        if "TCP_TCC_READ_REQ_LATENCY_sum" in d and d.get("TCP_TCC_READ_REQ_sum"):
            d["mean_L1_to_L2_read_latency_cycles"] = round(d["TCP_TCC_READ_REQ_LATENCY_sum"] / d["TCP_TCC_READ_REQ_sum"], 1)
        res[k] = d
    json.dump({"command": "tools/pmc_issue.sh (rocprofv3 --pmc, four passes) over tools/agg_bench.py --channels 256,64 "
                          "--dtypes bf16,fp32 --epilogue 1", "kernels": res}, open(out, "w"), indent=1)
    print("wrote", out, len(res), "kernels")


if __name__ == "__main__":
    main()
