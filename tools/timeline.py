"""Kernel timeline of the last pipeline step from a rocprofv3 --kernel-trace csv.  Usage: timeline.py dir [first-kernel-substring]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
key = sys.argv[2] if len(sys.argv) > 2 else 'scan2400'
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if key in r['Kernel_Name']]
i0 = idx[-1]
t0 = int(rows[i0]['Start_Timestamp'])
prev_end = t0
for r in rows[i0:i0 + 48]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if (s - prev_end) > 5e6: break
    name = r['Kernel_Name'].split('(')[0][-44:]
    print("%9.1f us  gap %7.1f  dur %8.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, name))
    prev_end = e
