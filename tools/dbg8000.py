import sys, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import dump1090_amd as da
from dump1090_amd import synth
from oracle import orc
from test_gpu_dropin import blocks_of
iq, _ = synth.capture(65536 * 6 + 4321, 8.0e6, "sc16", seed=35, frames_per_sec=2500, bitflip1=0.2, ap_fraction=0.3)
o = orc.Oracle(8000000, fix_bits=1); o.run_stream(iq, orc.INPUT_SC16, 8000); exp = o.messages()
L = da.lib(); L.modes_hip_bind(None); L.crc_init(1); da.take_messages()
assert L.demod_8000_alloc() != 0
for data, ov, ts in blocks_of(iq, "sc16", rate=8000000):
    mb = da.make_mag_buf(data, ov, sample_timestamp=ts); L.demod_8000(C.byref(mb))
got = da.take_messages()
e = da.Engine(8000000, da.INPUT_SC16, fix_bits=1); e.run_host(iq); bulk = e.messages()
print(len(exp), len(got), len(bulk))
gs, es = set(got["timestamp"].tolist()), set(exp["timestamp"].tolist())
for x in exp:
    if int(x["timestamp"]) not in gs: print("missing", x["timestamp"], x["block"], x["j"], x["msg_type"], x["score"], bytes(x["msg"]).hex())
for x in got:
    if int(x["timestamp"]) not in es: print("extra", x["timestamp"], x["block"], x["j"].astype(np.int32), x["msg_type"], x["score"], bytes(x["msg"]).hex())
