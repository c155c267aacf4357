#!/bin/bash
# step time of some workloads for each value of one environment knob
# usage: tools/env_sweep.sh NAME "v1 v2 .." "workload .."
for v in $2; do
  for w in $3; do
    env $1=$v timeout -k 10 200 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu 2>/dev/null > /tmp/es.json || exit 1
    python - "$1=$v" <<'PY'
import json, sys
d = json.loads(open("/tmp/es.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["config"]["workload"].split(":")[0], "step", d["ms_per_step"], "ms; scans", d["roofline"].get("kernel_ms"), "ms")
PY
  done
done
