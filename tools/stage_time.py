import os, sys, torch
sys.path.insert(0, '.')
import dump1090_amd as da
from dump1090_amd import synth_torch
fmt = sys.argv[1]
dc = fmt.endswith('_dc')                                       # uc8_dc: the scan over the magnitudes the DC chain leaves (2 B/sample)
if dc: fmt = fmt[:-3]
n = (1<<30)//(2 if fmt=='uc8' else 4)
iq, info = synth_torch.capture_gpu(n, 2.4e6, fmt, seed=1, device='cuda:0', **({'dc_offset': (0.011, -0.017)} if dc else {}))
e = da.Engine(2400000, da.INPUT_UC8 if fmt=='uc8' else da.INPUT_SC16, filter_dc=dc)
nb = iq.numel()*iq.element_size()
if dc: e.run_device(iq.data_ptr(), nb)
noload = len(sys.argv) > 2 and sys.argv[2] == 'noload'        # one more run: everything but the loads (uc8's raw-byte tile)
for stop in (1,2,3,0) + ((9,) if noload and fmt == 'uc8' else ()):
    os.environ['MODES_DEBUG_STOP'] = str(stop)
    e.time_scan(iq.data_ptr(), nb, reps=2)
    ms = e.time_scan(iq.data_ptr(), nb, reps=5)
    print(fmt, "stop", stop, "ms %.3f" % ms, "Gsamples/s %.1f" % (n/ms/1e6))
