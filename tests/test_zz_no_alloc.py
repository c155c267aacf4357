"""modes_convert_device_noalloc() against modes_convert_device(): same kernels, scratch in a fixed per-GPU array.
Two calls in a row per converter, the second on a smaller input (the array keeps the first call's contents)."""
import ctypes as C

import numpy as np
import pytest
import torch

import dump1090_amd as da
from oracle import orc

pytestmark = pytest.mark.gpu

# (converter id, DC block on, bytes per sample the converter reads)
CASES = [(0, False, 1), (1, False, 1), (3, False, 4), (2, False, 2), (2, True, 2), (4, False, 4), (4, True, 4), (5, True, 4)]


def _input(conv, n, seed):
    rng = np.random.default_rng(seed)
    if conv <= 2:
        a = np.clip(np.round(rng.normal(127.5, 30.0, size=2 * n)), 0, 255).astype(np.uint8)
    else:
        scale = 600.0 if conv == 5 else 9000.0
        a = np.clip(np.round(rng.normal(50.0, scale, size=2 * n)), -32768, 32767).astype(np.int16)
        a[:8] = [-32768, -32768, 32767, -32768, 0, 0, -1, 1]
    return a


def _state(dc, rate=2400000.0):
    st = da.ConvertState()
    if dc:
        st.DC_B = float(np.float32(np.exp(-2.0 * np.pi / rate)))
        st.DC_A = float(np.float32(1.0 - np.float64(np.float32(st.DC_B))))
    else:
        st.DC_A, st.DC_B = 0.0, 1.0
    st.z1_I, st.z1_Q = (0.01, -0.02) if dc else (0.0, 0.0)
    return st


@pytest.mark.parametrize("conv,dc,_bytes", CASES)
def test_noalloc_matches_allocating_entry_point(conv, dc, _bytes):
    for n, seed in ((200_003, 5), (50_001, 6)):                 # the second call is smaller: stale scratch must not leak
        iq = _input(conv, n, seed)
        d_iq = torch.from_numpy(iq).cuda()
        ref_mag = torch.zeros(n, dtype=torch.int16, device="cuda")
        st = _state(dc)
        ref_power = da.convert_device(conv, d_iq.data_ptr(), ref_mag.data_ptr(), n, state=st)
        # the allocating entry point itself against the oracle's converter (same start state)
        ost = orc.ConvertState(DC_A=_state(dc).DC_A, DC_B=_state(dc).DC_B, z1_I=_state(dc).z1_I, z1_Q=_state(dc).z1_Q)
        omag, opower, ost = orc.convert(conv, iq, n, state=ost)
        assert np.array_equal(ref_mag.cpu().numpy().view(np.uint16), omag)
        assert ref_power == opower and np.float32(st.z1_I) == np.float32(ost.z1_I) and np.float32(st.z1_Q) == np.float32(ost.z1_Q)
        got_mag = torch.full((n + 64,), -1, dtype=torch.int16, device="cuda")
        res = torch.zeros(2, dtype=torch.float64, device="cuda")
        st0 = _state(dc)
        da.convert_device_noalloc(conv, d_iq.data_ptr(), got_mag.data_ptr(), n, st0.DC_A, st0.DC_B, st0.z1_I, st0.z1_Q,
                                  d_result_ptr=res.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(got_mag[:n], ref_mag)
        assert bool((got_mag[n:] == -1).all())                  # nothing written past nsamples
        r = res.cpu().numpy().view(da.CONVERT_RESULT_DTYPE)[0]
        assert float(r["out_power"]) == ref_power               # bit-equal: same kernels, same order of additions
        assert np.float32(r["z1_I"]) == np.float32(st.z1_I) and np.float32(r["z1_Q"]) == np.float32(st.z1_Q)
        if conv in (1, 2, 4, 5):
            assert ref_power > 0.0


def test_noalloc_rejects_sizes_beyond_its_scratch():
    n = da.NOALLOC_MAX_SAMPLES + 1
    d_iq = torch.zeros(4 * n, dtype=torch.uint8, device="cuda")
    d_mag = torch.zeros(n, dtype=torch.int16, device="cuda")
    with pytest.raises(da.ModesError, match="MODES_NOALLOC_MAX_SAMPLES"):
        da.convert_device_noalloc(3, d_iq.data_ptr(), d_mag.data_ptr(), n)
    da.convert_device_noalloc(3, d_iq.data_ptr(), d_mag.data_ptr(), da.NOALLOC_MAX_SAMPLES)      # the maximum itself is fine
    torch.cuda.synchronize()
    assert int(d_mag.abs().max()) == 0


def test_noalloc_under_stream_capture():
    """The point of the entry point: it can be recorded into a HIP graph (after one set-up call outside capture)."""
    n = 30_000
    iq = _input(4, n, 9)
    d_iq = torch.from_numpy(iq).cuda()
    st = _state(True)
    ref_mag = torch.zeros(n, dtype=torch.int16, device="cuda")
    ref_power = da.convert_device(4, d_iq.data_ptr(), ref_mag.data_ptr(), n, state=st)
    got_mag = torch.zeros(n, dtype=torch.int16, device="cuda")
    res = torch.zeros(2, dtype=torch.float64, device="cuda")
    st0 = _state(True)
    da.convert_device_noalloc(4, d_iq.data_ptr(), got_mag.data_ptr(), n, st0.DC_A, st0.DC_B, st0.z1_I, st0.z1_Q, res.data_ptr())
    torch.cuda.synchronize()
    got_mag.zero_(); res.zero_()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=s):
        da.convert_device_noalloc(4, d_iq.data_ptr(), got_mag.data_ptr(), n, st0.DC_A, st0.DC_B, st0.z1_I, st0.z1_Q,
                                  res.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(got_mag.abs().max()) == 0                        # captured, not run
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(got_mag, ref_mag)
    assert float(res.cpu().numpy().view(da.CONVERT_RESULT_DTYPE)[0]["out_power"]) == ref_power
