#!/bin/bash
# LDS bank-conflict share of corr_volume per ablation (run on the GPU box): tools/lds_conflicts.sh "0 8 4"
# (0 everything, 8 frame loop only, 4 no products.  NOT 1 -- no DMA -- under the profiler: that run stopped answering.)
# Each pass prints its line as soon as it is done (a silent run is taken for hung after 7 minutes).
export TMPDIR=/tmp
ROOT=$(pwd)
for ab in $1; do
  rm -rf "$ROOT/gpurun_out/pmc_conf_$ab"
  ( cd /tmp && UMPA_HIP_ABLATE=$ab rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d "$ROOT/gpurun_out/pmc_conf_$ab" -o run -- python3 "$ROOT/bench.py" --no-cpu --steps 3 --warmup 1 > /dev/null 2>&1 )
  python3 - "$ROOT/gpurun_out/pmc_conf_$ab" "$ab" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(float); n = set()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "corr_volume" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
d = max(len(n), 1)
print("ablate %s: conflict cycles %.3g, LDS active %.3g, share %.3f, LDS instructions %.3g (per launch)" % (
    sys.argv[2], acc["SQ_LDS_BANK_CONFLICT"] / d, acc["SQ_LDS_IDX_ACTIVE"] / d,
    acc["SQ_LDS_BANK_CONFLICT"] / max(acc["SQ_LDS_IDX_ACTIVE"], 1), acc["SQ_INSTS_LDS"] / d))
PY
done
