"""Small profiling target: a few whole-pipeline runs and scan-only launches on one synthetic recording.
Usage: python3 tools/prof_scan.py {uc8|sc16} [GiB]   (run under rocprofv3 ... -- python3 tools/prof_scan.py sc16)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dump1090_amd as da
from dump1090_amd import synth_torch

fmt = sys.argv[1] if len(sys.argv) > 1 else "sc16"
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
n = int(gib * (1 << 30)) // (2 if fmt == "uc8" else 4)
iq, _ = synth_torch.capture_gpu(n, 2.4e6, fmt, seed=1, device="cuda:0")
nb = iq.numel() * iq.element_size()
e = da.Engine(2400000, da.INPUT_UC8 if fmt == "uc8" else da.INPUT_SC16)
for _ in range(3):
    e.set_icao(None)
    e.run_device(iq.data_ptr(), nb)
ms = e.time_scan(iq.data_ptr(), nb, reps=3)
print(fmt, "samples", n, "scan ms %.4f" % ms, "msgs", len(e.messages()), e.stats())
