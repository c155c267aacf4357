"""Run the filter_dc bulk path once on a synthetic uc8 recording (GPU box); prints the chain kernel's numbers.
usage: dc_time.py [gib] [fmt] [seed] [compute_power 0|1]"""
import sys, torch
sys.path.insert(0, '.')
import dump1090_amd as da
from dump1090_amd import synth_torch
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
fmt = sys.argv[2] if len(sys.argv) > 2 else 'uc8'
n = int(gib * (1 << 30)) // (2 if fmt == 'uc8' else 4)
iq, info = synth_torch.capture_gpu(n, 2.4e6, fmt, seed=int(sys.argv[3]) if len(sys.argv) > 3 else 1, device='cuda:0', dc_offset=(0.011, -0.017))
e = da.Engine(2400000, da.INPUT_UC8 if fmt == 'uc8' else da.INPUT_SC16, filter_dc=True, compute_power=(int(sys.argv[4]) != 0 if len(sys.argv) > 4 else True))
for _ in range(2):
    e.set_icao(None); e.set_dc_state(0.0, 0.0)
    e.run_device(iq.data_ptr(), iq.numel() * iq.element_size())
    torch.cuda.synchronize()
print(fmt, n, e.num_messages(), e.dc_info())
