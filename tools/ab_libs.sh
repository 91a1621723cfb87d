#!/bin/bash
# A/B of library builds on one GPU box: tools/ab_libs.sh _exp/libA.so _exp/libB.so ...  (corr_volume / step times of C2)
for rep in 1 2; do
  for lib in "$@"; do
    cp "$lib" umpa_amd/libumpa_hip.so
    echo "$lib $(timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernels_ms"])')"
  done
done
