#!/bin/bash
# step time of the message-bearing workloads against the number of ranges the shard is cut into (MODES_RANGES)
# usage: tools/ranges_sweep.sh "1 2 3 4 6" "uc8_2400_lut sc16_2400 sc16_2000"
for r in $1; do
  for w in $2; do
    MODES_RANGES=$r timeout -k 10 200 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu 2>/dev/null > /tmp/rs.json || exit 1
    python - "$r" <<'PY'
import json, sys
d = json.loads(open("/tmp/rs.json").read().strip().splitlines()[-1])
print("ranges", sys.argv[1], d["config"]["workload"].split(":")[0], "step", d["ms_per_step"], "ms; scans", d["roofline"].get("kernel_ms"), "ms")
PY
  done
done
