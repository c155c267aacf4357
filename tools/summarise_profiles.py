"""gpurun_out/profiles_raw (tools/collect_profiles.sh) -> profiles/r01_*: kernel stats of this library's kernels per
workload, and the scan kernel's HBM traffic per launch from FETCH_SIZE / WRITE_SIZE (separate passes)."""
import csv, glob, json, os, collections
raw, out = "gpurun_out/profiles_raw", "profiles"
os.makedirs(out, exist_ok=True)
for old in glob.glob(out + "/r01_*"):
    if "pmc_scan_stages" not in old:                 # written by tools/stage_pmc_json.py
        os.remove(old)
for d in sorted(glob.glob(raw + "/stats_*")):
    wl = os.path.basename(d)[6:]
    f = glob.glob(d + "/*/*kernel_stats.csv")[0]
    rows = list(csv.reader(open(f)))
    keep = [rows[0]] + [r for r in rows[1:] if "modes::" in r[0]]
    csv.writer(open("%s/r01_kernel_stats_%s.csv" % (out, wl), "w")).writerows(keep)
    b = json.load(open("%s/bench_%s.json" % (raw, wl)))
    json.dump(b, open("%s/r01_bench_under_rocprof_%s.json" % (out, wl), "w"), indent=1)
hbm = {}
for wl in ("uc8_2400", "sc16_2400"):
    res = {}
    for ctr in ("fetch", "write"):
        f = glob.glob("%s/%s_%s/*/*counter_collection.csv" % (raw, ctr, wl))[0]
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "modes::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        res[ctr.upper() + "_SIZE_mean_KB"] = {k: round(sum(v) / len(v), 1) for k, v in acc.items()}
    scan = [k for k in res["FETCH_SIZE_mean_KB"] if "scan2400" in k][0]
    fetch, write = res["FETCH_SIZE_mean_KB"][scan], res["WRITE_SIZE_mean_KB"][scan]
    res["scan2400_traffic_bytes_per_launch"] = int((2 * fetch + write) * 1024)
    res["note"] = ("FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM: gfx950 reports half the bytes of wide coalesced reads); "
                   "WRITE_SIZE as reported; KB units; separate --pmc passes")
    hbm[wl] = res
json.dump(hbm, open(out + "/r01_pmc_hbm.json", "w"), indent=1)
for wl, r in hbm.items():
    print(wl, "scan traffic per launch: %.3f GB" % (r["scan2400_traffic_bytes_per_launch"] / 1e9))
