#!/bin/bash
# usage (on the GPU box): tools/pmc_kernel.sh <kernel substring> <bench arguments ...> -- SQ counters of the kernels whose
# name holds the substring, summed over the run (one pass, 8 SQ counters)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
pat=$1; shift
d=gpurun_out/pmck; rm -rf $d $d.log
timeout -s KILL 150 rocprofv3 --pmc ${PMC_CTRS:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT} -d $d --output-format csv -- python3 bench.py "$@" --no-cpu > $d.log 2>&1 || { echo "pmc FAILED"; tail -3 $d.log; exit 1; }
python3 - $d "$pat" <<'PY'
import sys,glob,csv,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.Counter()
for f in glob.glob(sys.argv[1]+'/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:48]
        if sys.argv[2] not in k: continue
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        if r['Counter_Name']=='SQ_WAVES': calls[k]+=1
for k,v in acc.items():
    w=v.get('SQ_WAVES',1) or 1
    print(k, 'launches', calls[k], 'per wave:', {a:round(b/w,1) for a,b in v.items()})
PY
find $d -type f -delete
