#!/bin/bash
# Runs on the GPU box (gpurun): for every bench workload one rocprofv3 --kernel-trace --stats run of bench.py (kernel
# stats CSV and the JSON line of THAT run are kept as a pair, or not at all), then FETCH_SIZE and WRITE_SIZE passes
# (separate --pmc runs, no tracing options) for the HBM traffic of the dominant kernel.
# Results land in gpurun_out/profiles_raw/; tools/summarise_profiles.py turns them into profiles/r03_*.
#   usage: tools/collect_profiles.sh [stats|pmc|all] [workload ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
what=${1:-all}; shift
wls=${*:-uc8_2400 uc8_2400_lut sc16_2400 sc16_2000 sc16_8000 sc16_2400ac uc8_2400_dc sc16_2400_dc}
out=gpurun_out/profiles_raw
mkdir -p $out
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu --no-also"
if [ $what = stats ] || [ $what = all ]; then
  for wl in $wls; do
    tmp=$out/tmp_$wl; rm -rf $tmp $out/stats_$wl $out/bench_$wl.json; mkdir -p $tmp
    if timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $tmp/stats --output-format csv -- $BENCH --workload $wl > $tmp/bench.json 2> $tmp/bench.err \
       && python3 -c "import json,sys; json.loads(open('$tmp/bench.json').read().strip().splitlines()[-1])" \
       && ls $tmp/stats/*/*kernel_stats.csv > /dev/null; then
      find $tmp/stats -type f ! -name '*kernel_stats.csv' -delete      # gpurun copies back at most 64 MiB
      tail -1 $tmp/bench.json > $out/bench_$wl.json && mv $tmp/stats $out/stats_$wl     # the pair, from one run
      echo "stats $wl ok"
    else
      echo "stats $wl FAILED"; tail -3 $tmp/bench.err; exit 1
    fi
    rm -rf $tmp
  done
fi
if [ $what = pmc ] || [ $what = all ]; then
  for wl in $wls; do
    [ $wl = sc16_2400ac ] && continue                  # four engines at once: per-kernel bytes are those of sc16_2400
    for ctr in FETCH_SIZE WRITE_SIZE; do
      d=$out/${ctr}_$wl; rm -rf $d
      timeout -s KILL 240 rocprofv3 --pmc $ctr -d $d --output-format csv -- $BENCH --workload $wl > $d.json 2> $d.err || { echo "pmc $ctr $wl FAILED"; tail -3 $d.err; exit 1; }
      find $d -type f ! -name '*counter_collection.csv' -delete
      for f in $d/*/*counter_collection.csv; do                # this library's kernels only (the generator's fill the file)
        python3 -c "import sys; L=open('$f').read().splitlines(); open('$f','w').write('\n'.join([L[0]]+[x for x in L[1:] if 'modes::' in x])+'\n')"
      done
      echo "pmc $ctr $wl ok"
    done
  done
fi
