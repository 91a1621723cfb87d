#!/usr/bin/env python3
"""Merge the unconverged-Newton counts recorded by test runs with UMPA_RECORD_UNCONVERGED=1
(gpurun_out/unconverged_{cpu,gpu}.json, written by tests/conftest.py) into tests/golden/unconverged_observed.json:
label -> the largest count any run saw.  oracle/parity.py caps later runs at twice that count."""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
table = os.path.join(root, "tests", "golden", "unconverged_observed.json")
seen = json.load(open(table)) if os.path.exists(table) else {}
for f in sys.argv[1:] or [os.path.join(root, "gpurun_out", "unconverged_cpu.json"), os.path.join(root, "gpurun_out", "unconverged_gpu.json")]:
    if os.path.exists(f):
        for k, v in json.load(open(f)).items():
            seen[k] = max(int(v), int(seen.get(k, 0)))
json.dump(seen, open(table, "w"), indent=0, sort_keys=True)
print("%d labels, %d with unconverged pixels" % (len(seen), sum(1 for v in seen.values() if v)))
