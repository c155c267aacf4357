import csv, glob, collections, sys
tag = sys.argv[1]; waves = float(sys.argv[2]) if len(sys.argv) > 2 else 131072
f = glob.glob('gpurun_out/pmc_%s/*/*counter_collection.csv' % tag)[0]
rows = [r for r in csv.DictReader(open(f)) if 'scan2400' in r['Kernel_Name']]
by = collections.defaultdict(dict)
for r in rows: by[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
ids = sorted(by)
prev = None
for k, i in enumerate(ids):
    if k % 7 == 3:
        d = {c: v / waves for c, v in by[i].items()}
        print("stop", (k // 7 + 1) % 4, " ".join("%s=%.0f" % (c.replace("SQ_", ""), v) for c, v in sorted(d.items())))
print(open('gpurun_out/pmc_%s.log' % tag).read())
