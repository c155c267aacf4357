"""gpurun_out/pmc_<tag> (tools/stage_pmc.sh) -> per-strip instruction counts of the scan kernel after each stage, as JSON."""
import csv, glob, collections, json, sys
out = {}
for tag, waves in (("sc16", 131072), ("uc8", 262144)):
    f = glob.glob('gpurun_out/pmc_%s/*/*counter_collection.csv' % tag)[0]
    rows = [r for r in csv.DictReader(open(f)) if 'scan2400' in r['Kernel_Name']]
    by = collections.defaultdict(dict)
    for r in rows:
        by[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
    ids = sorted(by)
    res = {}
    for k, i in enumerate(ids):
        if k % 7 == 3:                                     # one timed launch of each debug stop (2 + 5 launches per stop)
            stop = (k // 7 + 1) % 4
            res["through_stage_%s" % "ABCD"[stop - 1 if stop else 3]] = {c.replace("SQ_", ""): round(v / waves, 1) for c, v in sorted(by[i].items())}
    ms = {}
    for line in open('gpurun_out/pmc_%s.log' % tag):
        p = line.split()
        if len(p) >= 5 and p[1] == "stop":
            ms["through_stage_%s" % "ABCD"[int(p[2]) - 1 if int(p[2]) else 3]] = float(p[4])
    out[tag] = {"strips_per_launch": waves, "positions_per_strip": 2048, "per_strip": res, "ms_under_profiler_1GiB": ms}
out["note"] = ("rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY "
               "SQ_INSTS_VMEM_RD on tools/stage_time.py (1 GiB recordings, scan kernel stopped after stage A / B / C / run whole); "
               "values are per wavefront = per 2048-position strip. A wave64 VALU instruction occupies its SIMD for 4 cycles: "
               "INSTS_VALU x 4 x strips per SIMD is the kernel's floor in cycles.")
json.dump(out, open('profiles/r01_pmc_scan_stages.json', 'w'), indent=1)
for tag in ("sc16", "uc8"):
    print(tag, {k: v.get("INSTS_VALU") for k, v in out[tag]["per_strip"].items()}, out[tag]["ms_under_profiler_1GiB"])
