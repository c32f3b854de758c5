#!/usr/bin/env python3
"""Turn rocprofv3 --pmc counter CSVs (one pass per counter, as MI355X_MICROARCH.md prescribes:
FETCH_SIZE and WRITE_SIZE do not fit one pass) into per-launch HBM-side traffic of the
aggregation kernels.

    python tools/pmc_summary.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/> <out.json> [command string]

Units / corrections (guide, section HBM): both counters are in KiB; on gfx950 FETCH_SIZE reports
exactly 1/2 of the bytes of wide (16 B/lane) coalesced reads -- this kernel's gathers, epilogue
loads and staging loads are all such -- so reads = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact for
16-B/lane streaming stores."""
import collections
import csv
import glob
import json
import re
import statistics
import sys


def load(d):
    rows = []
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def main():
    root, out = sys.argv[1], sys.argv[2]
    cmd = sys.argv[3] if len(sys.argv) > 3 else ""
    commit = sys.argv[4] if len(sys.argv) > 4 else None
    import hashlib
    import os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # the table describes the kernel it was collected from: bench.py nulls `roofline.traffic` once csrc/spmm.hip differs
    sha = hashlib.sha256(open(os.path.join(here, "semigcn_amd", "csrc", "spmm.hip"), "rb").read()).hexdigest()[:16]
    per = {}
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        agg = collections.defaultdict(list)
        for r in load(f"{root}/pmc_{cname}"):
            if r["Counter_Name"] != cname:
                continue
            ring = re.search(r"spmm_ring<(\d+), (\d+), (\d+), (\d+)>", r["Kernel_Name"])      # <channels / 64, epilogue operands, ring depth, store mode>
            if ring:
                agg[("ring", "bfloat16", 8 * int(ring.group(1)), 1, int(ring.group(2)))].append(float(r["Counter_Value"]))
                continue
            rf = re.search(r"spmm_ring_f32<(\d+), (\d+), (\d+), (\d+)>", r["Kernel_Name"])  # <channels / 64, epilogue operands, ring depth, blocks per wavefront>
            if rf:
                agg[("ring_f32", "float32", 16 * int(rf.group(1)), 1, int(rf.group(2)))].append(float(r["Counter_Value"]))
                continue
            m = re.search(r"spmm_(rows|shared)<(.+?), (\d+), (\d+), (\d+)(?:, (?:true|false))?(?:, \d+)?>", r["Kernel_Name"])
            if not m:
                continue
            dt = "bfloat16" if "bf16" in m.group(2) else "float32"
            agg[(m.group(1), dt, int(m.group(3)), int(m.group(4)), int(m.group(5)))].append(float(r["Counter_Value"]))
        per[cname] = agg
    res = []
    for key in sorted(per["FETCH_SIZE"]):
        f = statistics.mean(per["FETCH_SIZE"][key])
        w = statistics.mean(per["WRITE_SIZE"].get(key, [0.0]))
        res.append({"kernel": "spmm_" + key[0], "dtype": key[1], "lanes_per_row": key[2], "vectors_per_lane": key[3],
                    "epilogue_operands": key[4],
                    "launches": len(per["FETCH_SIZE"][key]), "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
                    "read_bytes_corrected": round(2 * f * 1024), "write_bytes": round(w * 1024),
                    "traffic_bytes_per_launch": round(2 * f * 1024 + w * 1024)})
    # float32 rows of 256 channels with epilogue operands run as TWO launches of the 128-channel kernel (csrc/spmm.hip,
    # launch_ring_f32): one sg_spmm call = twice that kernel's per-launch traffic
    for k in list(res):
        if k["kernel"] == "spmm_ring_f32" and k["lanes_per_row"] == 32 and k["epilogue_operands"] >= 1:
            d = dict(k)
            d.update({"kernel": "spmm_ring_f32 x 2 (128-channel halves)", "lanes_per_row": 64, "derived": "2 x the 128-channel launch",
                      "read_bytes_corrected": 2 * k["read_bytes_corrected"], "write_bytes": 2 * k["write_bytes"],
                      "traffic_bytes_per_launch": 2 * k["traffic_bytes_per_launch"]})
            res.append(d)
    json.dump({"command": cmd, "commit": commit, "spmm_hip_sha16": sha, "correction": "reads = 2 * FETCH_SIZE * 1024 (gfx950 wide-read under-count), writes = WRITE_SIZE * 1024",
               "kernels": res}, open(out, "w"), indent=1)
    print("wrote", out, len(res), "kernel variants")


if __name__ == "__main__":
    main()
