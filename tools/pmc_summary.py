#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/profile_gpu.sh: per kernel, mean duration and mean counter value per dispatch."""
import csv, glob, json, os, sys
from collections import defaultdict

def short(name):
    n = name.split("(")[0]
    for key in ("prep_maps", "corr_volume_queue", "corr_volume", "replay_walk", "corr_masked_queue", "corr_masked", "replay_cost",
                "blur_tiles", "od_list", "corr_march", "match_direct", "match_staged", "coverage", "cost_one", "spfit"):
        if key in n:
            return key
    return n[:60]

def main(root):
    out = {"kernels_ms": {}, "counters": defaultdict(dict)}
    for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out["kernels_ms"][short(r["Name"])] = dict(calls=int(r["Calls"]), avg_ms=float(r["AverageNs"]) / 1e6,
                                                       total_ms=float(r["TotalDurationNs"]) / 1e6)
    for f in glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(float))
        disp = defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
        for k in acc:
            for c, v in acc[k].items():
                out["counters"][k][c] = v / max(len(disp[k]), 1)
    print(json.dumps(out, indent=1, sort_keys=True))

if __name__ == "__main__":
    main(sys.argv[1])
