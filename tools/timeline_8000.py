import csv, glob, sys
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48], r.get("Stream_Id", r.get("Queue_Id", "?"))))
for f in glob.glob(d + "/**/*_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", ""), r.get("Stream_Id", "?")))
rows.sort()
trigs = [i for i, r in enumerate(rows) if "d8_trig_kernel" in r[2]]
i0 = trigs[-2]; i1 = trigs[-1]
t0 = rows[i0][0]
for s, e, n, q in rows[i0:i1]:
    if "modes::" in n or "COPY" in n: print("%9.1f %8.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
