"""Mean counter value per dispatch per kernel from a rocprofv3 --pmc counter_collection.csv.  Usage: pmc_summary.py file.csv [kernel-substring]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if sub in k:
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k[:90], "VGPR", rows[0].get("VGPR_Count"), )
    for c, v in cs.items():
        print("   %-24s n=%-3d mean %.4g" % (c, len(v), sum(v) / len(v)))
