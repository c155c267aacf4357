"""Prints a slice of the kernel / copy timeline of a rocprofv3 --kernel-trace --memory-copy-trace run:
   python3 tools/timeline_any.py <dir> <start_ms> <length_ms>   (times relative to the last third of the trace)"""
import csv, glob, sys
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:44], r.get("Stream_Id", r.get("Queue_Id", "?"))))
for f in glob.glob(d + "/**/*_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", ""), r.get("Stream_Id", "?")))
rows = [r for r in rows if "modes::" in r[2] or "COPY" in r[2]]
rows.sort()
t_end = rows[-1][1]
t0 = t_end - int(float(sys.argv[2]) * 1e6)
t1 = t0 + int(float(sys.argv[3]) * 1e6)
busy = 0
for s, e, n, q in rows:
    if s >= t0 and s < t1:
        print("%9.1f %8.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
