"""gpurun_out/profiles_raw (tools/collect_profiles.sh) -> profiles/<round>_*: kernel stats of this library's kernels per
workload with the bench line of the same run, and the dominant kernel's HBM traffic per step from FETCH_SIZE / WRITE_SIZE
(separate passes).  Fails when a workload's kernel-stats CSV and its bench line disagree by more than 10 % on the
dominant kernel's time per step: then they are not from one run (or the live HIP-event timing is off)."""
import collections
import csv
import glob
import json
import os
import sys

raw, out = "gpurun_out/profiles_raw", "profiles"
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"                    # the round the files are named after
RUNS = 4                                                  # bench.py --steps 3 --warmup 1
os.makedirs(out, exist_ok=True)
bad = []
for d in sorted(glob.glob(raw + "/stats_*")):
    wl = os.path.basename(d)[6:]
    files = glob.glob(d + "/*/*kernel_stats.csv")
    if len(files) != 1:
        sys.exit("%s: expected one kernel_stats.csv, found %d" % (d, len(files)))
    rows = list(csv.reader(open(files[0])))
    keep = [rows[0]] + [r for r in rows[1:] if "modes::" in r[0]]
    b = json.load(open("%s/bench_%s.json" % (raw, wl)))
    kern = b["roofline"]["kernel"]
    kerns = kern.split(" + ")                              # (8 MS/s: the trigger and the window kernel together)
    total_ns = sum(float(r[2]) for r in keep[1:] if any(k in r[0] for k in kerns))
    alone = b["roofline"].get("kernel_ms_alone")
    if isinstance(alone, float):                          # bench.py: one more launch of the kernel, alone, behind the timed steps
        total_ns -= alone * 1e6
    per_step_ms = total_ns / RUNS / 1e6 / max(1, b["config"].get("recordings_per_gpu", 1))      # bench.py: kernel_ms is per recording
    live = b["roofline"]["kernel_ms"]
    rel = abs(per_step_ms - live) / live if live else 1.0
    print("%-14s %-18s csv %.4f ms/step  bench %.4f ms  (%.1f %%)" % (wl, kern, per_step_ms, live, 100 * rel))
    # engines running side by side on their own streams (sc16_2400ac): a launch's events also bracket time it waited
    # for a CU behind the other streams' kernels, the profiler's duration does not
    if rel > (0.10 if b["config"].get("concurrent_engines", 1) == 1 else 0.25):
        bad.append(wl)
    csv.writer(open("%s/%s_kernel_stats_%s.csv" % (out, tag, wl), "w")).writerows(keep)
    json.dump(b, open("%s/%s_bench_under_rocprof_%s.json" % (out, tag, wl), "w"), indent=1)
if bad:
    sys.exit("kernel stats and bench line differ by more than 10 %% for: %s" % ", ".join(bad))

hbm = {}
try:                                   # a partial refresh (collect_profiles.sh all <workload>) keeps the other workloads' entries
    hbm = {k: v for k, v in json.load(open("%s/%s_pmc_hbm.json" % (out, tag))).items() if k != "note"}
except (OSError, ValueError):
    pass
for f in sorted(glob.glob(raw + "/FETCH_SIZE_*.json")):
    wl = os.path.basename(f)[len("FETCH_SIZE_"):-5]
    res = {}
    try:
        b = json.loads(open(f).read().strip().splitlines()[-1])
    except (ValueError, IndexError):
        continue
    kern = b["roofline"]["kernel"]
    kerns = kern.split(" + ")
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob("%s/%s_%s/*/*counter_collection.csv" % (raw, ctr, wl))
        if len(files) != 1:
            sys.exit("%s %s: expected one counter_collection.csv, found %d" % (ctr, wl, len(files)))
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            if "modes::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        res[ctr + "_KB_per_step"] = {k: round(sum(v) / RUNS, 1) for k, v in acc.items()}
        res[ctr + "_launches_per_step"] = {k: round(len(v) / RUNS, 2) for k, v in acc.items()}
    # bench.py decodes the recording once more behind the timed steps, scanned in one launch (roofline.kernel_ms_alone)
    extra = 1 if isinstance(b["roofline"].get("kernel_ms_alone"), float) else 0
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        for k in res[ctr + "_KB_per_step"]:
            if extra:                                     # every kernel of the decode ran once more
                res[ctr + "_KB_per_step"][k] = round(res[ctr + "_KB_per_step"][k] * RUNS / (RUNS + extra), 1)
                if any(q in k for q in kerns):
                    res[ctr + "_launches_per_step"][k] = round((res[ctr + "_launches_per_step"][k] * RUNS - extra) / RUNS, 2)
                else:
                    res[ctr + "_launches_per_step"][k] = round(res[ctr + "_launches_per_step"][k] * RUNS / (RUNS + extra), 2)
    fetch = sum(v for k, v in res["FETCH_SIZE_KB_per_step"].items() if any(q in k for q in kerns))
    write = sum(v for k, v in res["WRITE_SIZE_KB_per_step"].items() if any(q in k for q in kerns))
    res["kernel"] = kern
    res["traffic_bytes_per_step"] = int((2 * fetch + write) * 1024)
    res["all_kernels_traffic_bytes_per_step"] = int((2 * sum(res["FETCH_SIZE_KB_per_step"].values()) + sum(res["WRITE_SIZE_KB_per_step"].values())) * 1024)
    hbm[wl] = res
if hbm:
    hbm["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes without tracing options, on bench.py --steps 3 --warmup 1; "
                   "KB summed over the launches of a kernel and divided by the 4 steps run.  traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: "
                   "FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM: gfx950 reports half the bytes of wide coalesced reads), WRITE_SIZE as reported.")
    json.dump(hbm, open("%s/%s_pmc_hbm.json" % (out, tag), "w"), indent=1)
    for wl, r in hbm.items():
        if wl != "note":
            print("%-14s %-18s traffic per step %.3f GB (all kernels %.3f GB)" % (wl, r["kernel"], r["traffic_bytes_per_step"] / 1e9, r["all_kernels_traffic_bytes_per_step"] / 1e9))
