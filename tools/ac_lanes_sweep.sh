for n in 2 4 6 8 12; do timeout -k 10 200 python bench.py --workload sc16_2400ac --steps 5 --warmup 2 --no-cpu --batch-streams $n 2>/dev/null > /tmp/ac.json || exit 1; python - $n <<'PY'
import json,sys
d=json.loads(open("/tmp/ac.json").read().strip().splitlines()[-1]); print("lanes",sys.argv[1],"step",d["ms_per_step"],"value",d["value"],"msgs",d["msgs"])
PY
done
