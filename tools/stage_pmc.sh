#!/bin/bash
# usage (on the GPU box): tools/stage_pmc.sh  -- instruction / cycle counters per stage of the scan kernel, both formats.
# Two passes per format (one --pmc run holds at most 8 SQ counters): issue counts, then lane occupancy of the VALU.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for fmt in sc16 uc8; do
  for pass in a b; do
    d=gpurun_out/pmc_${fmt}_$pass; rm -rf $d $d.log
    if [ $pass = a ]; then ctrs="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD"
    else ctrs="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVES"; fi
    timeout -s KILL 120 rocprofv3 --pmc $ctrs -d $d --output-format csv -- python3 tools/stage_time.py $fmt > $d.log 2>&1 || { echo "pmc $fmt $pass FAILED"; tail -3 $d.log; exit 1; }
    find $d -type f ! -name '*counter_collection.csv' -delete
    for f in $d/*/*counter_collection.csv; do
      python3 -c "import sys; L=open('$f').read().splitlines(); open('$f','w').write('\\n'.join([L[0]]+[x for x in L[1:] if 'scan2400' in x])+'\\n')"
    done
    echo "pmc $fmt $pass ok"
  done
done
