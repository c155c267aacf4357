cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for pass in a b; do
  d=gpurun_out/pmc8_$pass; rm -rf $d $d.log
  if [ $pass = a ]; then ctrs="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES"
  else ctrs="SQ_WAVES SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_WAIT_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; fi
  timeout -s KILL 120 rocprofv3 --pmc $ctrs -d $d --output-format csv -- python3 bench.py --workload sc16_8000 --steps 1 --warmup 1 --no-cpu > $d.log 2>&1 || { echo "pmc $pass FAILED"; tail -3 $d.log; exit 1; }
  python3 - $d <<'PY'
import sys,glob,csv,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob(sys.argv[1]+'/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:40]
        if 'd8_' not in k: continue
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in acc.items():
    print(k, {a:round(b/1e6,2) for a,b in v.items()})
PY
  find $d -type f -delete
done
