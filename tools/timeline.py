"""Prints one step's timeline from a rocprofv3 --kernel-trace --memory-copy-trace run: every kernel / copy with its
start relative to the step's first scan kernel, duration and stream, so gaps between dependent launches show."""
import csv, glob, sys
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Stream_Id", r.get("Queue_Id", "?"))))
for f in glob.glob(d + "/**/*_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") , r.get("Stream_Id", "?")))
rows.sort()
scans = [i for i, r in enumerate(rows) if r[2].startswith("void modes::scan2") or "scan2400_kernel" in r[2] or "scan2000_kernel" in r[2]]
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 1
step = int(sys.argv[3]) if len(sys.argv) > 3 else 2
i0 = scans[step * nr]
i1 = scans[(step + 1) * nr] if (step + 1) * nr < len(scans) else len(rows)
t0 = rows[i0][0]
for s, e, n, q in rows[i0:i1]:
    print("%9.1f %8.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
