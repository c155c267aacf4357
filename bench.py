#!/usr/bin/env python3
"""bench.py -- headline benchmark of the 2.4 MS/s path (BASELINE.json configs[1]).

One "step" = one pass of the whole hot path (IQ resident in HBM -> ordered message list on the
host) over one synthetic recording.  Prints ONE JSON line on rank 0.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--gib", type=float, default=4.0, help="IQ bytes per GPU, GiB")
    ap.add_argument("--format", default="uc8", choices=["uc8", "sc16"])
    ap.add_argument("--converter", default="reference", choices=["reference", "power"],
                    help="reference = convert_init(fmt, DC off, power off); power = compute_power=1")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget (rank 0, N=1)")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local

    bps = 2 if args.format == "uc8" else 4
    nsamples = int(args.gib * (1 << 30)) // bps
    fmt = da.INPUT_UC8 if args.format == "uc8" else da.INPUT_SC16
    t0 = time.time()
    iq, info = synth_torch.capture_gpu(nsamples, 2.4e6, args.format, seed=1 + rank, device=dev)
    torch.cuda.synchronize()
    gen_s = time.time() - t0
    nbytes = iq.numel() * iq.element_size()

    eng = da.Engine(2400000, fmt, compute_power=(args.converter == "power"), fix_bits=1, device=local)
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.set_icao(None)                    # every step decodes the recording from a cold ICAO cache
        eng.run_device(iq.data_ptr(), nbytes, stream=stream)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.set_icao(None)
        eng.run_device(iq.data_ptr(), nbytes, stream=stream)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    msgs = eng.messages()
    stats = eng.stats()
    nmsg = torch.tensor([len(msgs)], device=dev, dtype=torch.int64)
    if world > 1:
        dist.all_reduce(nmsg)
    ms_per_step = dt / args.steps * 1e3
    value = world * nsamples / (dt / args.steps) / 1e6            # Msamples/s, whole job

    # dominant kernel: the fused scan kernel, HIP events on the launch stream
    scan_ms = eng.time_scan(iq.data_ptr(), nbytes, reps=5, stream=stream)
    # algorithmic bytes per sample: the IQ bytes the selected converter has to read
    alg_bps = 1 if (args.format == "uc8" and args.converter == "reference") else bps
    achieved = nsamples * alg_bps / (scan_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "scan2400_kernel",
                "kernel_ms": round(scan_ms, 4), "alg_bytes_per_sample": alg_bps}

    cpu = None
    if rank == 0 and not args.no_cpu:
        from oracle import orc
        # bounded sample of the same recording: whole 256 KiB blocks from the start
        nblk = max(8, int(args.cpu_seconds * 35e6 / (262144 // bps)))
        nblk = min(nblk, nbytes // 262144)
        host = iq[: nblk * 262144 // iq.element_size()].cpu().numpy()
        o = orc.Oracle(2400000, fix_bits=1, measure_noise=(args.converter == "power"))
        t0 = time.perf_counter()
        o.run_stream(host, orc.INPUT_UC8 if args.format == "uc8" else orc.INPUT_SC16, 2400,
                     compute_power=(args.converter == "power"))
        cdt = time.perf_counter() - t0
        cpu_msgs = o.messages()
        # prefix parity: the engine's messages of those blocks must be the oracle's
        g = msgs[msgs["block"] < nblk] if world == 1 else None
        parity = None
        if g is not None:
            parity = bool(len(g) == len(cpu_msgs) and np.array_equal(g["timestamp"], cpu_msgs["timestamp"]))
        cpu = {"value": round(nblk * (262144 // bps) / cdt / 1e6, 2), "unit": "Msamples/s", "cores": 1, "kind": "port",
               "sample": "first %d blocks (%d MiB) of the same recording, oracle/ C restatement, 1 thread"
                         % (nblk, nblk // 4), "msgs": int(len(cpu_msgs)), "prefix_parity": parity}

    if rank == 0:
        conv_name = da.CONVERTER_NAMES[{"uc8": {"reference": 0, "power": 1}, "sc16": {"reference": 3, "power": 4}}
                                       [args.format][args.converter]]
        print(json.dumps({
            "metric": "Msamples/s decoded (+ msgs/s) at 2.4 MS/s uc8", "value": round(value, 1),
            "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16", "data": "synthetic",
            "config": {"workload": "synthetic 2.4 MS/s %s IQ, %.2f GiB per GPU, injected DF11/DF17 frames at mixed SNR"
                                   % (args.format, args.gib), "converter": conv_name, "demod": "demod_2400",
                       "fix_bits": 1, "frames_injected_per_gpu": int(len(info["pos"]))},
            "msgs": int(nmsg.item()), "msgs_per_s": round(int(nmsg.item()) / (dt / args.steps), 1),
            "valid_preamble": stats["valid_preamble"], "candidates": stats["candidates"],
            "roofline": roofline, "cpu_baseline": cpu, "gen_seconds": round(gen_s, 1)}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
