#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repository root: per-dispatch kernel durations of one bench step
# (rocprofv3 kernel trace), in launch order -> gpurun_out/trace_<tag>.txt.   bash tools/trace_one_step.sh <tag> [bench args]
ROOT=$(pwd)
TAG=$1; shift
OUT=$ROOT/gpurun_out/trace_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o run -- python3 "$ROOT/bench.py" --no-cpu --steps 2 --warmup 1 "$@" > "$OUT/log.txt" 2>&1
cd "$ROOT"
python3 - "$OUT" > "$ROOT/gpurun_out/trace_$TAG.txt" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the last complete step: from the last prep_maps (or the last corr kernel of a new step) on
last = max(i for i, n in enumerate(names) if "prep_maps" in n) if any("prep_maps" in n for n in names) else 0
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:last + 40]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%8.1f us  %s  grid %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:60], r.get("Grid_Size", "?")))
PY
cat "$ROOT/gpurun_out/trace_$TAG.txt"
