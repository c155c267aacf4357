#!/usr/bin/env python3
"""bench.py -- throughput of the raw-IQ hot path on MI355X (BASELINE.json).

One "step" = one pass of the whole hot path (IQ resident in HBM -> ordered message list in host memory)
over one synthetic recording (or, for the Mode A/C batch, over the rank's recordings back to back).
Prints ONE JSON line on rank 0.

N > 1 (one rank per GPU): ONE recording of N x --gib is cut into N shards of whole 256 KiB blocks (dump1090_amd/shard.py).
Every rank holds its shard plus one block in front of it, forms its own overlap magnitudes, and the entry states (ICAO
cache; at 8 MS/s demod_8000's carried state) are handed from shard to shard until they settle -- all inside the timed
step.  Per-GPU work is fixed as N grows ("scaling": "weak").  The float converters with the DC block on and the
Mode A/C batch run as independent recordings per rank.

Workloads (BASELINE.json `configs`):
  uc8_2400      configs[1]  synthetic 2.4 MS/s uc8, 4 GiB, DF11/DF17 at mixed SNR        (default, the headline)
  uc8_2400_lut              the headline's kernel instance on an input it decodes: a uc8 buffer whose image under the
                            reference's as-written byte table (one byte per output sample) is the same
                            frame train
  sc16_2400                 the same signal as sc16 (the converter that decodes: see DESIGN.md 7)
  sc16_2000     configs[2]  2.0 MS/s demod_2000, 1- and 2-bit CRC repair, dense traffic
  sc16_8000     configs[3]  8 MS/s demod_8000, this rank's 4 GiB shard of the 32 GiB capture
  sc16_2400ac   configs[4]  demod_2400 + demod_2400_AC on a batch of 256 MiB recordings (64 / N per rank)
  uc8_2400_dc               configs[1]'s signal with a DC offset through convert_uc8_generic, DC block ON -- what the
                            reference's shipped dump1090.cfg selects (DC-filter = true); the uc8 path that decodes
  sc16_2400_dc              the same through convert_sc16_generic
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import dump1090_amd as da                     # noqa: E402
from dump1090_amd import synth_torch          # noqa: E402

HBM_PEAK_GBS = 8000.0                         # MI355X_MICROARCH.md: 8 TB/s spec
PMC_HBM_FILES = ("r03_pmc_hbm.json", "r02_pmc_hbm.json")      # newest first (profiles/README.md)
VALU_ISSUE_CYCLES = 4.0                       # what a wave64 VALU instruction of this kernel's mix holds a SIMD for (see valu_issue below)
PMC_STAGE_FILES = ("r03_pmc_scan_stages.json", "r02_pmc_scan_stages.json")

WORKLOADS = {
    #               rate      fmt     engine kwargs                       traffic              kernel
    "uc8_2400":    (2400000, "uc8",  {"fix_bits": 1},                    {"frames_per_sec": 1000.0}, "scan2400_kernel"),
    "uc8_2400_lut": (2400000, "uc8lut", {"fix_bits": 1},                 {"frames_per_sec": 1000.0, "min_amp_db": -26.0}, "scan2400_kernel"),
    "sc16_2400":   (2400000, "sc16", {"fix_bits": 1},                    {"frames_per_sec": 1000.0}, "scan2400_kernel"),
    "sc16_2000":   (2000000, "sc16", {"fix_bits": 2},                    {"frames_per_sec": 5000.0}, "scan2000_kernel"),
    "sc16_8000":   (8000000, "sc16", {"fix_bits": 1},                    {"frames_per_sec": 1000.0}, "d8_window_kernel"),
    "sc16_2400ac": (2400000, "sc16", {"fix_bits": 1, "mode_ac": True},   {"frames_per_sec": 1000.0}, "scan2400_kernel"),
    "uc8_2400_dc": (2400000, "uc8",  {"fix_bits": 1, "filter_dc": True, "compute_power": True},
                    {"frames_per_sec": 1000.0, "dc_offset": (0.011, -0.017)}, "dc_chain_kernel"),
    "sc16_2400_dc": (2400000, "sc16", {"fix_bits": 1, "filter_dc": True, "compute_power": True},
                     {"frames_per_sec": 1000.0, "dc_offset": (0.011, -0.017)}, "dc_chain_kernel"),
}
CONV_ROW = {("uc8", False): 0, ("uc8", True): 1, ("sc16", False): 3, ("sc16", True): 4, ("uc8", "dc"): 2, ("sc16", "dc"): 4}


_CPU_WORKER = """
# One shard of ONE stream on one host core (test infrastructure: the oracle).  Protocol on stdin/stdout, binary: the parent
# sends the ICAO cache the shard is entered with (4096 bytes), the worker decodes and answers with the cache it leaves
# (4096 bytes) and its message count (8 bytes); an empty read ends it.
import sys, numpy as np
sys.path.insert(0, %r)
from oracle import orc
path, rate, fmt, fix_bits, power, also_ac, rank, world = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])
raw = np.load(path, mmap_mode="r")
bps = 2 if fmt == "uc8" else 4
B = 262144 // bps
nblk = raw.nbytes // 262144
base, extra = divmod(nblk, world)
b0 = rank * base + min(rank, extra)
nb = base + (1 if rank < extra else 0)
mine = np.ascontiguousarray(raw[b0 * 262144 // raw.itemsize: (b0 + nb) * 262144 // raw.itemsize])
ov = int((8 + 112 + 16) * 1e-6 * rate)
ifmt = orc.INPUT_UC8 if fmt == "uc8" else orc.INPUT_SC16
tail = None
if b0:      # the magnitudes in front of the shard: the converter over the block before it (a pure function of the samples)
    prev = np.ascontiguousarray(raw[(b0 - 1) * 262144 // raw.itemsize: b0 * 262144 // raw.itemsize])
    tail = orc.convert(orc.convert_select(ifmt, False, bool(power)), prev, B, sample_rate=float(rate))[0][B - ov:]
out, inp = sys.stdout.buffer, sys.stdin.buffer
out.write(b"ready\\n"); out.flush()
while True:
    blob = inp.read(4096)
    if len(blob) < 4096:
        break
    o = orc.Oracle(rate, fix_bits=fix_bits, measure_noise=bool(power))
    o.icao_import(np.frombuffer(blob, dtype=np.uint32))
    o.stream_start(b0 * B, tail)
    if nb:
        o.run_stream(mine, ifmt, rate // 1000, compute_power=bool(power), also_ac=bool(also_ac))
    out.write(o.icao_export().tobytes() + np.array([len(o.messages())], dtype=np.int64).tobytes()); out.flush()
""" % ROOT


def cpu_all_cores(host, rate, fmt, bps, fix_bits, power, also_ac, single_msgs):
    """The oracle on ONE stream over all host cores: the sample is cut into one shard of whole blocks per core, every
    shard is decoded by its own process, and the ICAO caches are handed from shard to shard until no entry moves
    (dump1090_amd.shard.next_entries -- what the GPU ranks do).  Wall time covers every round.  Fresh interpreter
    processes (no torch, no HIP): nothing of this process's GPU state is forked."""
    import subprocess
    import tempfile
    from dump1090_amd import shard as shard_mod
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "16")), 16)
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        path = os.path.join(td, "sample.npy")
        np.save(path, host)
        argv = [sys.executable, "-c", _CPU_WORKER, path, str(rate), fmt, str(fix_bits), str(int(power)), str(int(also_ac))]
        procs = [subprocess.Popen(argv + [str(r), str(cores)], stdin=subprocess.PIPE, stdout=subprocess.PIPE) for r in range(cores)]
        try:
            for pr in procs:
                assert pr.stdout.readline().strip() == b"ready"
            zero = np.zeros(1024, dtype=np.uint32)
            entered = [zero] * cores
            todo = list(range(cores))
            exits, counts = [None] * cores, [0] * cores
            rounds = decodes = 0
            t0 = time.perf_counter()
            while True:
                for r in todo:
                    procs[r].stdin.write(entered[r].tobytes()); procs[r].stdin.flush()
                for r in todo:
                    blob = procs[r].stdout.read(4096 + 8)
                    exits[r] = np.frombuffer(blob[:4096], dtype=np.uint32).copy()
                    counts[r] = int(np.frombuffer(blob[4096:], dtype=np.int64)[0])
                rounds += 1
                decodes += len(todo)
                if all(np.array_equal(exits[j - 1], entered[j]) for j in range(1, cores)) or rounds > cores + 1:
                    break
                nxt = shard_mod.next_entries(entered, exits)
                todo = [r for r in range(cores) if not np.array_equal(nxt[r], entered[r])]
                entered = nxt
            adt = time.perf_counter() - t0
        finally:
            for pr in procs:
                pr.stdin.close()
            for pr in procs:
                pr.wait()
    return {"value": round(host.nbytes // bps / adt / 1e6, 2), "unit": "Msamples/s", "cores": cores, "rounds": rounds,
            "shard_decodes": decodes, "msgs": int(sum(counts)), "msgs_equal_single_thread": int(sum(counts)) == int(single_msgs),
            "sample": "the same %d MiB as ONE stream cut into %d shards of whole blocks, one oracle process per core, ICAO caches "
                      "handed over to a fixed point (all rounds timed)" % (host.nbytes >> 20, cores)}


def _same_messages(g, c):
    """Engine messages (dump1090_amd.MESSAGE_DTYPE) against the oracle's: every integer field and the message bytes."""
    if len(g) != len(c):
        return False
    if len(g) == 0:
        return True
    ok = (np.array_equal(g["timestamp"], c["timestamp"]) and np.array_equal(g["crc"], c["crc"])
          and np.array_equal(g["error_bits"].astype(np.int64), c["error_bits"].astype(np.int64))
          and np.array_equal(g["msg_type"].astype(np.int64), c["msg_type"].astype(np.int64))
          and np.array_equal(g["msg_bits"].astype(np.int64), c["msg_bits"].astype(np.int64)))
    mask = np.where(c["msg_type"] == 18, 0xFFFFFF, 0xFFFFFFFF).astype(np.uint32)      # DF18's flag bit reads stale bytes
    ok = ok and np.array_equal(g["addr"] & mask, c["addr"] & mask)
    nb = c["msg_bits"] // 8
    for k in (2, 7, 14):
        sel = nb == k
        ok = ok and np.array_equal(g["msg"][sel][:, :k], c["msg"][sel][:, :k])
    ok = ok and bool(np.allclose(g["sig_level"], c["sig_level"], rtol=1e-5, atol=0))
    return bool(ok)


def run_workload(name, args, world, rank, local, dev, want_cpu):
    rate, fmt, ekw, traffic, kernel = WORKLOADS[name]
    import torch.distributed as dist
    gen_fmt = fmt
    fmt = "uc8" if fmt == "uc8lut" else fmt
    bps = 2 if fmt == "uc8" else 4
    dc = bool(ekw.get("filter_dc"))
    power = args.converter == "power" or dc
    if name == "sc16_2400ac":
        nrec = max(1, 64 // world)
        nsamples = (256 << 20) // bps
    else:
        nrec = 1
        nsamples = int(args.gib * (1 << 30)) // bps
    fmt_id = da.INPUT_UC8 if fmt == "uc8" else da.INPUT_SC16
    sharded = world > 1 and nrec == 1 and not dc
    t0 = time.time()
    recs = []
    lead = 0
    if sharded:
        # this rank's part of ONE recording of world x nsamples: its shard and the block in front of it
        from dump1090_amd import shard as shard_mod
        B = 262144 // bps
        plan = shard_mod.plan_shards(world * nsamples, B, world)
        first, count = plan[rank]
        chunk = 1 << 24
        lead = B if first else 0
        g0 = (first - lead) // chunk * chunk
        iq_all, info = synth_torch.capture_gpu(world * nsamples, float(rate), gen_fmt, seed=1, device=dev, chunk=chunk,
                                               sample_range=(g0, -(-(first + count) // chunk) * chunk), **traffic)
        iq = iq_all[2 * (first - lead - g0): 2 * (first + count - g0)]
        info = dict(info, pos=info["pos"][(info["pos"] >= first) & (info["pos"] < first + count)])
        recs.append((iq, info))
    else:
        for r in range(nrec):
            iq, info = synth_torch.capture_gpu(nsamples, float(rate), gen_fmt, seed=1 + rank * 64 + r, device=dev, **traffic)
            recs.append((iq, info))
    torch.cuda.synchronize()
    gen_s = time.time() - t0
    if os.environ.get("MODES_BENCH_PROGRESS"):
        print("[rank %d] %s: input formed in %.1f s" % (rank, name, gen_s), file=sys.stderr, flush=True)
    nbytes = recs[0][0].numel() * recs[0][0].element_size() - lead * bps
    eng = da.Engine(rate, fmt_id, device=local, **dict(ekw, compute_power=power))
    stream = torch.cuda.current_stream().cuda_stream
    # a batch of independent recordings (configs[4]) is decoded by several engines at once, each on its own HIP stream
    # and host thread: one engine's host-side steps (counts read back, ordered Mode A/C pass) overlap the others' kernels
    nlanes = min(args.batch_streams, nrec) if nrec > 1 else 1
    lanes = [(eng, stream)]
    for _ in range(nlanes - 1):
        st_ = torch.cuda.Stream(device=dev)
        lanes.append((da.Engine(rate, fmt_id, device=local, **dict(ekw, compute_power=power)), st_.cuda_stream))
        lanes[-1] += (st_,)                  # keep the stream object alive
    pool = None
    if nlanes > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=nlanes)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sh = grp = None
    hand_rounds = [0]
    if sharded:
        iq = recs[0][0]
        sh = shard_mod.EngineShard(eng, iq.data_ptr(), iq.numel() * iq.element_size(), lead, first, count, rate, bps, stream=stream)
        grp = shard_mod.TorchGroup(args.coll_dev)

    def run_sharded():
        sh.reset()
        rounds = shard_mod.decode_engine_rounds(sh, rank, world, grp.all_gather, max_rounds=2 * world + 2)
        hand_rounds[0] = rounds
        a, b = eng.kernel_ms()
        return eng.num_messages(), a, b

    def run_lane(li):
        e, st = lanes[li][0], lanes[li][1]
        nm, kms, k2 = 0, 0.0, 0.0
        for iq, _ in recs[li::nlanes]:
            e.set_icao(None)                  # every recording is decoded from a cold ICAO cache
            if dc:
                e.set_dc_state(0.0, 0.0)      # ... and from convert_init()'s filter state
            e.run_device(iq.data_ptr(), nbytes, stream=st)
            nm += e.num_messages()
            a, b = e.kernel_ms()
            if dc:                            # the chain kernel dominates there; the scan kernel is reported second
                a, b = e.dc_info()["chain_ms"], a
            kms += a
            k2 += b
        return nm, kms, k2

    def one_step():
        if sharded:
            return run_sharded()
        if pool is None:
            return run_lane(0)
        parts = list(pool.map(run_lane, range(nlanes)))
        return tuple(sum(x[i] for x in parts) for i in range(3))

    for _ in range(args.warmup):
        one_step()
    barrier()
    if os.environ.get("MODES_BENCH_PROGRESS"):
        print("[rank %d] %s: warm-up done" % (rank, name), file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    kms = k2 = 0.0
    nm = 0
    for _ in range(args.steps):
        nm, a, b = one_step()
        kms += a
        k2 += b
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=args.coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    last_eng = lanes[(nrec - 1) % nlanes][0]          # the engine that decoded recs[-1] last
    msgs = last_eng.messages()
    stats = last_eng.stats()
    nmsg = torch.tensor([nm], device=args.coll_dev, dtype=torch.int64)
    if world > 1:
        dist.all_reduce(nmsg)
    step_s = dt / args.steps
    value = world * nrec * nsamples / step_s / 1e6               # Msamples/s, whole job
    shard_info = None
    if sharded:
        # outside the timed region: the shards' lists gathered on rank 0 (rank order = time order) and checked
        t0 = time.perf_counter()
        cdev = args.coll_dev
        raw = torch.from_numpy(msgs.view(np.uint8).reshape(-1).copy()).to(cdev) if len(msgs) else torch.zeros(0, dtype=torch.uint8, device=cdev)
        sizes = grp.all_gather(np.array([raw.numel()], dtype=np.int64))
        cap_b = max(int(x[0]) for x in sizes)
        pad = torch.zeros(max(cap_b, 1), dtype=torch.uint8, device=cdev)
        pad[: raw.numel()] = raw
        allb = torch.empty(world * pad.numel(), dtype=torch.uint8, device=cdev)
        dist.all_gather_into_tensor(allb, pad)
        merged_n, ordered = 0, True
        if rank == 0:
            host = allb.cpu().numpy().reshape(world, -1)
            parts = [np.frombuffer(host[r, : int(sizes[r][0])].tobytes(), dtype=da.MESSAGE_DTYPE) for r in range(world)]
            merged = shard_mod.merge(parts)
            merged_n = int(len(merged))
            ordered = bool(np.all(np.diff(merged["timestamp"].astype(np.int64)) >= -2700))      # skip-ahead keeps messages apart
        shard_info = {"recording_gib": round(world * nsamples * bps / (1 << 30), 2), "shards": world, "handover_rounds": hand_rounds[0],
                      "merged_msgs_on_rank0": merged_n, "merged_in_time_order": ordered,
                      "gather_to_rank0_ms": round((time.perf_counter() - t0) * 1e3, 2),
                      "note": "timed step = scan + hand-over rounds + each shard's ordered list in its rank's host memory; "
                              "the gather of the lists to rank 0 is measured here, outside it"}

    # dominant kernel: HIP events on the launch stream around it, averaged over the timed steps
    kernel_ms = kms / args.steps / nrec                           # per recording (a long shard's scan is up to four launches: their sum)
    if rate == 8000000:
        # demod_8000 is two kernels that each read every sample (trigger planes, then the match windows): the pipeline's
        # roofline figure is the algorithmic bytes over BOTH (the window kernel alone would flatter it)
        kernel = "d8_trig_kernel + d8_window_kernel"
        kernel_ms += k2 / args.steps / nrec
    # algorithmic bytes per sample: the IQ bytes the selected converter has to read
    alg_bps = 1 if (fmt == "uc8" and not power) else bps
    achieved = nsamples * alg_bps / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    # HBM bytes per step of the dominant kernel from PMC counters.  The counters cannot be read inside this process
    # (rocprofv3 collects them in passes of their own): the number is the one of the committed pass over THIS command
    # (profiles/README.md), valid only for the same recording -- same size, same seed, one rank -- and null otherwise.
    traffic, traffic_note = args.traffic, "--traffic"
    if traffic is None:
        if args.gib != 4.0 or world != 1 or nrec != 1:
            traffic_note = "null: the committed PMC pass covers the 4 GiB single-GPU recording of seed 1, not this run"
        else:
            traffic_note = "null: no committed PMC pass for this workload"
            for src in PMC_HBM_FILES:
                try:
                    with open(os.path.join(ROOT, "profiles", src)) as f:
                        traffic = json.load(f).get(name, {}).get("traffic_bytes_per_step")
                except (OSError, ValueError):
                    traffic = None
                if traffic is not None:
                    traffic_note = "profiles/%s: (2 x FETCH_SIZE + WRITE_SIZE) x 1024, separate --pmc passes over this command" % src
                    break
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_note, "kernel": kernel,
                "kernel_ms": round(kernel_ms, 4), "alg_bytes_per_sample": alg_bps}
    if kernel == "scan2400_kernel" and not dc and nrec == 1:
        # what binds the scan kernel is VALU issue, not HBM: instruction counts per strip from the committed
        # PMC pass (profiles/README.md) priced at 4 cycles per wave64 instruction and 1024 SIMDs at 2.4 GHz
        for src in PMC_STAGE_FILES:
            try:
                with open(os.path.join(ROOT, "profiles", src)) as f:
                    pm = json.load(f)[fmt]["per_strip"]["through_stage_D"]
            except (OSError, KeyError, ValueError):
                continue
            # measured issue cost (profiles/r03_ubench_valu_rate*.txt): 4.2-4.5 cycles per wave64 instruction and SIMD at
            # 4-8 waves per SIMD for the kinds this kernel is made of (VOP3, compares, selects, DPP, dot, 24-bit
            # multiplies); only plain 32-bit add / and / or / xor / shift-right / move issue in 2.4-2.9
            floor_ms = pm["INSTS_VALU"] * VALU_ISSUE_CYCLES * (nsamples / 2048 / 1024) / 2.4e9 * 1e3
            roofline["valu_issue"] = {"insts_per_2048_positions": pm["INSTS_VALU"], "cycles_per_inst": VALU_ISSUE_CYCLES,
                                      "floor_ms": round(floor_ms, 4),
                                      "frac_of_kernel_ms": round(floor_ms / kernel_ms, 3) if kernel_ms > 0 else None,
                                      "source": "profiles/" + src}
            break
    if kernel in ("scan2400_kernel", "scan2000_kernel") and not dc and nrec == 1 and not sharded:
        # A message-heavy shard is scanned in ranges of blocks whose launches share the GPU with the ordered pass and the
        # read-back of the range before (kernel_ms above: their sum, as timed inside the step).  The same kernel alone:
        # one more decode of the recording behind the timed region, scanned in one launch (MODES_RANGES=1):
        try:
            saved = os.environ.get("MODES_RANGES")
            os.environ["MODES_RANGES"] = "1"
            try:
                eng.set_icao(None)
                eng.run_device(recs[0][0].data_ptr(), nbytes, stream=stream)
                alone = eng.kernel_ms()[0]
            finally:
                if saved is None:
                    del os.environ["MODES_RANGES"]
                else:
                    os.environ["MODES_RANGES"] = saved
            roofline["kernel_ms_alone"] = round(alone, 4)
            roofline["achieved_alone"] = round(nsamples * alg_bps / (alone * 1e-3) / 1e9, 1)
        except Exception as ex:                    # a report, never a reason to lose the bench line
            roofline["kernel_ms_alone"] = repr(ex)
    if k2:
        roofline["second_kernel"] = {"kernel": "scan2400_kernel" if dc else "d8_trig_kernel", "kernel_ms": round(k2 / args.steps / nrec, 4),
                                     "note": "included in kernel_ms" if rate == 8000000 else "runs behind the chain, not in kernel_ms"}
    if dc:
        roofline["dc_chain"] = last_eng.dc_info()

    cpu = None
    if want_cpu:
        from oracle import orc
        iq0 = recs[-1][0]
        # bounded sample of the same recording: whole 256 KiB blocks from the start.  The as-written uc8 converter is a
        # table look-up (about 180 Msamples/s on one core), so the budget covers the headline's WHOLE recording and the
        # comparison is over the full message list; the other converters run at 35-45 Msamples/s and get a prefix.
        est = 170e6 if (fmt == "uc8" and not power) else 35e6
        nblk = max(8, int(args.cpu_seconds * est / (262144 // bps)))
        nblk = min(nblk, nbytes // 262144)
        whole = nblk * 262144 == nbytes
        host = iq0[: nblk * 262144 // iq0.element_size()].cpu().numpy()
        o = orc.Oracle(rate, fix_bits=ekw.get("fix_bits", 1), measure_noise=power)
        t0 = time.perf_counter()
        o.run_stream(host, orc.INPUT_UC8 if fmt == "uc8" else orc.INPUT_SC16, rate // 1000, compute_power=power,
                     filter_dc=dc, also_ac=bool(ekw.get("mode_ac")))
        cdt = time.perf_counter() - t0
        cpu_msgs = o.messages()
        # parity: the engine's messages of those blocks must be the oracle's, field by field.  (demod_8000 messages near the
        # cut belong to windows the shorter stream never completes: compare all but the last block there.)
        lim = nblk - (1 if (rate == 8000000 and not whole) else 0)
        g, c = msgs[msgs["block"] < lim], cpu_msgs[cpu_msgs["block"] < lim]
        parity = _same_messages(g, c)
        cpu = {"value": round(nblk * (262144 // bps) / cdt / 1e6, 2), "unit": "Msamples/s", "cores": 1, "kind": "port",
               "sample": "%s %d blocks (%d MiB) of the same recording, oracle/ C restatement, 1 thread"
                         % ("all" if whole else "first", nblk, nblk // 4), "msgs": int(len(cpu_msgs)),
               "prefix_parity": parity, "prefix_parity_msgs": int(len(c)),
               "parity_fields": "timestamp, msg bytes, msg_bits, DF, addr, crc, error_bits, sig_level (rtol 1e-5)"}
        if whole:                                    # the statistics of the whole run can be compared too
            ost = o.stats()
            cpu["stats_parity"] = bool(all(int(stats[k]) == int(ost[k]) for k in
                                           ("valid_preamble", "demodulated", "CRC_good", "CRC_bad", "CRC_single_bit_fix")))
            cpu["valid_preamble"] = int(ost["valid_preamble"])

    if cpu is not None and args.cpu_all_cores and not dc and rate != 8000000:
        try:
            cpu["all_cores"] = cpu_all_cores(host, rate, fmt, bps, ekw.get("fix_bits", 1), power, bool(ekw.get("mode_ac")), len(cpu_msgs))
        except Exception as ex:                    # the baseline is a report, never a reason to lose the bench line
            cpu["all_cores"] = {"error": repr(ex)}

    conv_name = da.CONVERTER_NAMES[CONV_ROW[(fmt, "dc" if dc else power)]] + (" (DC block on)" if dc else "")
    what = ("a uc8 buffer whose image under the as-written byte table is the frame train" if gen_fmt == "uc8lut"
            else "%s IQ" % fmt)
    return {
        "value": round(value, 1), "ms_per_step": round(step_s * 1e3, 3),
        "config": {"workload": ("%s: synthetic %.1f MS/s %s, %d x %.2f GiB per GPU, injected DF11/DF17 frames at mixed SNR"
                                % (name, rate / 1e6, what, nrec, nbytes / (1 << 30)))
                               + (", ONE %.0f GiB recording sharded over %d GPUs" % (world * nbytes / (1 << 30), world) if sharded else ""),
                   "converter": conv_name, "demod": "demod_%d%s" % (rate // 1000, "+demod_2400_AC" if ekw.get("mode_ac") else ""),
                   "fix_bits": ekw.get("fix_bits", 1), "frames_injected_per_gpu": int(sum(len(i["pos"]) for _, i in recs)),
                   "concurrent_engines": nlanes, "recordings_per_gpu": nrec},
        "msgs": int(nmsg.item()), "msgs_per_s": round(int(nmsg.item()) / step_s, 1),
        "valid_preamble": stats["valid_preamble"], "candidates": stats["candidates"],
        "roofline": roofline, "cpu_baseline": cpu, "gen_seconds": round(gen_s, 1), "shard": shard_info}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gib", type=float, default=4.0, help="IQ bytes per GPU, GiB")
    ap.add_argument("--workload", default="uc8_2400", choices=sorted(WORKLOADS))
    ap.add_argument("--converter", default="reference", choices=["reference", "power"],
                    help="reference = convert_init(fmt, DC off, power off); power = compute_power=1")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget (rank 0, N=1)")
    ap.add_argument("--batch-streams", type=int, default=4, help="engines (HIP streams) that share a batch of recordings")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-cpu-all-cores", dest="cpu_all_cores", action="store_false",
                    help="skip the all-host-cores leg of the CPU baseline (independent streams, one per core)")
    ap.add_argument("--no-also", action="store_true", help="skip the companion lines of the default workload")
    ap.add_argument("--traffic", type=float, default=None, help="HBM bytes per launch of the roofline kernel from PMC counters")
    args = ap.parse_args()

    if os.environ.get("MODES_BENCH_WATCHDOG"):       # seconds: where every thread stands, should the run take that long
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["MODES_BENCH_WATCHDOG"]), exit=False, file=sys.stderr)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # MODES_BENCH_REHEARSE=1: all ranks on GPU 0 with gloo collectives, to rehearse the N > 1 path on a one-GPU box
    rehearse = bool(os.environ.get("MODES_BENCH_REHEARSE"))
    if rehearse:
        local = 0
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    args.coll_dev = "cpu" if rehearse else dev

    res = run_workload(args.workload, args, world, rank, local, dev, want_cpu=(rank == 0 and world == 1 and not args.no_cpu))
    also = None
    if args.workload == "uc8_2400" and not args.no_also:
        # The reference-selected uc8 converter reads one byte per sample and decodes nothing from IQ.
        # Companion lines: the same kernel instance on an input it does decode (uc8_2400_lut), the uc8 path that
        # decodes real IQ and that the reference's shipped dump1090.cfg selects (uc8_2400_dc), and the same signal as sc16.
        also = {}
        # ... and BASELINE.json's other configs: configs[2] (sc16_2000), configs[3] (sc16_8000: one 4 GiB shard of the
        # capture) and configs[4] (sc16_2400ac: the batch of 64 recordings), each with its bounded CPU prefix and parity.
        for wl in ("uc8_2400_lut", "uc8_2400_dc", "sc16_2400", "sc16_2000", "sc16_8000", "sc16_2400ac"):
            args2 = argparse.Namespace(**vars(args))
            args2.cpu_seconds = 4.0 if wl in ("uc8_2400_lut", "uc8_2400_dc", "sc16_2400") else 3.0
            args2.cpu_all_cores = False
            args2.steps, args2.warmup = min(args.steps, 5), min(args.warmup, 2)
            torch.cuda.empty_cache()
            r2 = run_workload(wl, args2, world, rank, local, dev, want_cpu=(rank == 0 and world == 1 and not args.no_cpu))
            also[wl] = {k: r2[k] for k in ("value", "ms_per_step", "config", "msgs", "msgs_per_s", "roofline", "cpu_baseline")}
            also[wl]["steps"] = args2.steps

    if rank == 0:
        out = {"metric": "Msamples/s decoded (+ msgs/s) at 2.4 MS/s uc8", "value": res["value"], "unit": "Msamples/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16", "data": "synthetic"}
        out.update({k: res[k] for k in ("config", "msgs", "msgs_per_s", "valid_preamble", "candidates", "roofline",
                                        "cpu_baseline", "gen_seconds")})
        if res.get("shard"):
            out["shard"] = res["shard"]
        if also:
            out["also"] = also
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
