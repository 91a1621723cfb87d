#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repository root:
#   [PROF_TAG=name] bash tools/profile_gpu.sh [extra bench.py args]
# Writes gpurun_out/prof[_name]/{stats,pmc*}/ and gpurun_out/prof[_name]/summary.json (tools/pmc_summary.py).
# rocprofv3 gets the program itself after `--` (no env/bash wrappers), counters in their own passes.
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof${PROF_TAG:+_$PROF_TAG}
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--no-cpu --steps 5 --warmup 2 $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1
n=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY"; do
    n=$((n+1))
    rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc$n" -o run -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc$n.log" 2>&1 || echo "pmc pass $n failed"
    echo "pmc pass $n done"
done
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.json"
echo "profile done"
