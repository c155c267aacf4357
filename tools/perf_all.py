"""Whole-pipeline timings of every demodulator on one synthetic recording each (profiling aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dump1090_amd as da
from dump1090_amd import synth_torch

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
cases = [("2400", 2400000, "sc16", {}), ("2400", 2400000, "uc8", {}), ("2000", 2000000, "sc16", {"fix_bits": 2}),
         ("2000pe", 2000000, "sc16", {"fix_bits": 2, "phase_enhance": True}),
         ("8000", 8000000, "sc16", {}), ("2400+AC", 2400000, "sc16", {"mode_ac": True})]
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
for name, rate, fmt, kw in cases:
    if only and name not in only:
        continue
    n = int(gib * (1 << 30)) // (2 if fmt == "uc8" else 4)
    iq, _ = synth_torch.capture_gpu(n, float(rate), fmt, seed=1, device="cuda:0")
    nb = iq.numel() * iq.element_size()
    e = da.Engine(rate, da.INPUT_UC8 if fmt == "uc8" else da.INPUT_SC16, **kw)
    e.run_device(iq.data_ptr(), nb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        e.set_icao(None)
        e.run_device(iq.data_ptr(), nb)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    st = e.stats()
    print("%-8s %-5s %.2f GiB: %8.2f ms/pass %9.1f Msamples/s  msgs %d  vp %d cand %d rounds %d" %
          (name, fmt, gib, dt * 1e3, n / dt / 1e6, len(e.messages()), st["valid_preamble"], st["candidates"], st["replay_rounds"]), flush=True)
    del e, iq
    torch.cuda.empty_cache()
