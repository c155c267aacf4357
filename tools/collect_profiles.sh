#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel stats of every bench workload and the HBM counters of the scan kernel.
# Results land in gpurun_out/profiles_raw/; tools/summarise_profiles.py turns them into profiles/r01_*.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/profiles_raw
rm -rf $out && mkdir -p $out
for wl in uc8_2400 sc16_2400 sc16_2000 sc16_8000 sc16_2400ac uc8_2400_dc sc16_2400_dc; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/stats_$wl --output-format csv -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu --no-also > $out/bench_$wl.json 2> $out/bench_$wl.err
  find $out/stats_$wl -type f ! -name '*kernel_stats.csv' -delete      # gpurun copies back at most 64 MiB
done
for wl in uc8_2400 sc16_2400; do
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $out/fetch_$wl --output-format csv -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu --no-also > $out/fetch_$wl.json 2> $out/fetch_$wl.err
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $out/write_$wl --output-format csv -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu --no-also > $out/write_$wl.json 2> $out/write_$wl.err
done
