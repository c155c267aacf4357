"""dev aid (GPU box): chain kernel time of the filter_dc bulk path on the bench recording and on the same without
frames (noise + DC offset only: where the float filter dithers on a rounding boundary).  usage: dc_sweep.py [gib]"""
import sys, torch
sys.path.insert(0, '.')
import dump1090_amd as da
from dump1090_amd import synth_torch
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
n = int(gib * (1 << 30)) // 2
for fps in (1000.0, 0.01):
    iq, info = synth_torch.capture_gpu(n, 2.4e6, 'uc8', seed=1, device='cuda:0', dc_offset=(0.011, -0.017), frames_per_sec=fps)
    e = da.Engine(2400000, da.INPUT_UC8, filter_dc=True, compute_power=True)
    for _ in range(2):
        e.set_icao(None); e.set_dc_state(0.0, 0.0)
        e.run_device(iq.data_ptr(), iq.numel())
        torch.cuda.synchronize()
    i = e.dc_info()
    print("fps %6.0f msgs %7d chain_ms %.2f guess_rounds %d passes %d prefix_reruns %d rerun %d final z %r" % (fps, e.num_messages(), i["chain_ms"], i["guess_rounds"], i["plain_rounds"], i['prefix_reruns'], i['rerun_chunks'], e.get_dc_state() if hasattr(e, 'get_dc_state') else None))
    del iq, e
