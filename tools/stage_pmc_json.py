"""gpurun_out/pmc_<fmt>_<pass> (tools/stage_pmc.sh) -> profiles/r03_pmc_scan_stages.json: per-strip counters of the scan
kernel stopped after each stage."""
import collections
import csv
import glob
import json
import sys

out = {}
for tag, strips in (("sc16", 131072), ("uc8", 262144)):
    res = collections.defaultdict(dict)
    ms = {}
    for ps in "ab":
        files = glob.glob('gpurun_out/pmc_%s_%s/*/*counter_collection.csv' % (tag, ps))
        if len(files) != 1:
            sys.exit("pmc_%s_%s: expected one counter_collection.csv, found %d" % (tag, ps, len(files)))
        by = collections.defaultdict(dict)
        for r in csv.DictReader(open(files[0])):
            by[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
        ids = sorted(by)
        if len(ids) != 28:
            sys.exit("pmc_%s_%s: %d scan launches, expected 4 stops x (2 + 5)" % (tag, ps, len(ids)))
        for k, i in enumerate(ids):
            if k % 7 == 3:                                     # one timed launch of each debug stop (2 + 5 launches per stop)
                stop = (k // 7 + 1) % 4
                res["through_stage_%s" % "ABCD"[stop - 1 if stop else 3]].update({c.replace("SQ_", ""): round(v / strips, 1) for c, v in sorted(by[i].items())})
        if ps == "a":
            for line in open('gpurun_out/pmc_%s_a.log' % tag):
                p = line.split()
                if len(p) >= 5 and p[1] == "stop":
                    ms["through_stage_%s" % "ABCD"[int(p[2]) - 1 if int(p[2]) else 3]] = float(p[4])
    for k, v in res.items():
        if v.get("ACTIVE_INST_VALU") and v.get("THREAD_CYCLES_VALU"):
            # THREAD_CYCLES_VALU counts lanes x cycles, ACTIVE_INST_VALU cycles a SIMD spent in VALU instructions
            v["valu_lanes_busy_frac"] = round(v["THREAD_CYCLES_VALU"] / (64.0 * v["ACTIVE_INST_VALU"]), 3)
    out[tag] = {"strips_per_launch": strips, "positions_per_strip": 2048, "strips_per_wave": 2 if tag == "uc8" else 1,
                "per_strip": res, "ms_under_profiler_1GiB": ms}
out["note"] = ("rocprofv3 --pmc, two passes (SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU "
               "SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD / SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVES) on "
               "tools/stage_time.py (1 GiB recordings, scan kernel stopped after stage A / B / C / run whole); values are per "
               "2048-position strip (the byte-table instances scan two strips per wave).  A wave64 VALU instruction occupies its SIMD "
               "for 4 cycles: INSTS_VALU x 4 x strips per SIMD is the kernel's floor in cycles.  valu_lanes_busy_frac = "
               "THREAD_CYCLES_VALU / (64 x ACTIVE_INST_VALU).")
json.dump(out, open('profiles/r03_pmc_scan_stages.json', 'w'), indent=1)
for tag in ("sc16", "uc8"):
    print(tag, {k: (v.get("INSTS_VALU"), v.get("valu_lanes_busy_frac")) for k, v in out[tag]["per_strip"].items()}, out[tag]["ms_under_profiler_1GiB"])
