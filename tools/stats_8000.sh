#!/bin/bash
# usage (on the GPU box): tools/stats_8000.sh -- per-kernel times of the 8 MS/s workload, a quick look while tuning
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/st8
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/st8 --output-format csv -- python3 bench.py --workload ${1:-sc16_8000} --steps 5 --warmup 2 --no-cpu > gpurun_out/st8.log 2>&1 || { tail -5 gpurun_out/st8.log; exit 1; }
python3 - > gpurun_out/st8.txt <<PY
import glob,csv
for f in glob.glob("gpurun_out/st8/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "modes::" in r["Name"]: print(r["Name"][:60], r["Calls"], round(float(r["TotalDurationNs"])/1e6,3), round(float(r["AverageNs"])/1e3,1))
PY
find gpurun_out/st8 -type f -delete
head -${2:-8} gpurun_out/st8.txt
grep -o '"ms_per_step": [0-9.]*' gpurun_out/st8.log
