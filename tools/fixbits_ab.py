"""dev aid (GPU box): what the error-table searches cost the scan kernels.  The same recording scanned with the 2-bit
tables (fix_bits = 2: 1326 / 3831 entries, 64 Kbit filter in front), the 1-bit tables and none (fix_bits = 0: no search
at all): the difference bounds what holding the tables in LDS instead of global memory / L2 could gain."""
import sys, torch
sys.path.insert(0, '.')
import dump1090_amd as da
from dump1090_amd import synth_torch
for rate, fps, kw in ((2000000, 5000.0, {}), (2400000, 1000.0, {})):
    n = (1 << 30) // 4 * 2
    iq, info = synth_torch.capture_gpu(n, float(rate), 'sc16', seed=1, device='cuda:0', frames_per_sec=fps)
    nb = iq.numel() * iq.element_size()
    for fb in (2, 1, 0):
        e = da.Engine(rate, da.INPUT_SC16, fix_bits=fb)
        e.time_scan(iq.data_ptr(), nb, reps=2)
        ms = e.time_scan(iq.data_ptr(), nb, reps=8)
        e.run_device(iq.data_ptr(), nb)
        print("rate %d fix_bits %d scan kernel %.3f ms per 2 GiB, %d messages, %d candidates" % (rate, fb, ms, e.num_messages(), e.stats()["candidates"]))
