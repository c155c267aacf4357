"""PCIe-inclusive rate of the bulk engine (never bench.py's `value`): modes_engine_run_host on a recording in pageable
and in pinned host memory.  usage (GPU box): h2d_rate.py [gib] [fmt]"""
import sys, time, torch
sys.path.insert(0, '.')
import dump1090_amd as da
from dump1090_amd import synth_torch
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
fmt = sys.argv[2] if len(sys.argv) > 2 else 'sc16'
n = int(gib * (1 << 30)) // (2 if fmt == 'uc8' else 4)
iq, _ = synth_torch.capture_gpu(n, 2.4e6, fmt, seed=1, device='cuda:0')
e = da.Engine(2400000, da.INPUT_UC8 if fmt == 'uc8' else da.INPUT_SC16)
for name, host in (("pageable", iq.cpu()), ("pinned", iq.cpu().pin_memory())):
    a = host.numpy()
    for _ in range(2):
        e.set_icao(None)
        t0 = time.perf_counter()
        e.run_host(a)
        dt = time.perf_counter() - t0
    print("%s %s host memory: %.1f ms for %.2f GiB = %.1f Msamples/s (%.1f GB/s incl. decode), %d messages"
          % (fmt, name, dt * 1e3, gib, n / dt / 1e6, a.nbytes / dt / 1e9, e.num_messages()))
