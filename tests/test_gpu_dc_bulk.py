"""GPU parity: the float converters with the DC block ON through the bulk engine (filter_dc=1), against the oracle.

The reference's shipped configuration (`DC-filter = true`, dump1090.cfg:101) selects convert_uc8_generic /
convert_sc16_generic (ref: src/convert.c:301-308, 112-200): the one-pole filter runs through the whole stream, so
the magnitudes of every sample -- not only the messages -- must equal the sequential result bit for bit."""
import numpy as np
import pytest

import dump1090_amd as da
from dump1090_amd import synth
from oracle import orc
from util import assert_same_messages, oracle_run

pytestmark = pytest.mark.gpu
FMT = {"uc8": da.INPUT_UC8, "sc16": da.INPUT_SC16, "sc16q11": da.INPUT_SC16Q11}
CONV = {"uc8": 2, "sc16": 4, "sc16q11": 5}
OV = {2000000: 272, 2400000: 326, 8000000: 1088}


def oracle_mags(iq, fmt, rate):
    n = iq.size // 2
    mag, _, st = orc.convert(CONV[fmt], iq, n, sample_rate=float(rate), filter_dc=True)
    return mag, (st.z1_I, st.z1_Q)


def check_run(iq, fmt, rate, warm=0, expect_min_msgs=1, fix_bits=1):
    e = da.Engine(rate, FMT[fmt], filter_dc=True, compute_power=True, fix_bits=fix_bits)
    if warm:
        e.set_dc_lead(0, warm)
    e.run_host(iq)
    n = iq.size // 2
    ov = OV[rate]
    got = e.dc_magnitudes(0, ov + n)
    exp, zexp = oracle_mags(iq, fmt, rate)
    assert not got[:ov].any()                                   # the stream's first block is preceded by zeros
    bad = np.flatnonzero(got[ov:] != exp)
    assert bad.size == 0, "first differing sample %d of %d (%d differ)" % (bad[0], n, bad.size)
    assert e.get_dc_state() == zexp
    emsg, est = oracle_run(iq, fmt, rate=rate, filter_dc=True, compute_power=True, fix_bits=fix_bits)
    assert len(emsg) >= expect_min_msgs
    assert_same_messages(e.messages(), emsg)
    return e


def test_modes1_2000_generic_dc(modes1):
    """The reference's own recording at its native rate: 259 messages."""
    e = check_run(modes1, "uc8", 2000000, expect_min_msgs=259)
    assert e.num_messages() == 259


def test_modes1_2400_generic_dc(modes1):
    e = da.Engine(2400000, da.INPUT_UC8, filter_dc=True, compute_power=True)
    e.run_host(modes1)
    _, est = oracle_run(modes1, "uc8", filter_dc=True, compute_power=True)
    assert e.stats()["valid_preamble"] == est["valid_preamble"] == 2956


@pytest.mark.parametrize("fmt", ["uc8", "sc16", "sc16q11"])
def test_synthetic_with_dc_offset(fmt):
    iq, _ = synth.capture(1_500_000, 2.4e6, fmt, seed=31, frames_per_sec=1500, dc_offset=0.021 - 0.013j,
                          min_amp_db=-20 if fmt == "sc16q11" else -32)
    check_run(iq, fmt, 2400000, expect_min_msgs=200)


@pytest.mark.parametrize("warm", [4096, 8192, 262144])
def test_short_warmup_is_repaired(warm):
    """With a warm-up far too short for the brackets to close, every chunk is re-run from its predecessor's exact
    end state: the result must not depend on the warm-up length."""
    iq, _ = synth.capture(700_000, 2.4e6, "uc8", seed=32, frames_per_sec=1500, dc_offset=-0.01 + 0.03j)
    e = check_run(iq, "uc8", 2400000, warm=warm, expect_min_msgs=100)
    info = e.dc_info()
    assert info["chunks"] >= 8 and info["warm_samples"] == warm
    repaired = info["rerun_chunks"] + info["prefix_reruns"]
    assert repaired >= info["chunks"] - 1 - (warm + 65536 + info["chunk_samples"] - 1) // info["chunk_samples"]


@pytest.mark.parametrize("nsamples", [0, 1, 255, 256, 257, 4099, 65536, 131072 + 331])
def test_ragged_lengths(nsamples):
    iq, _ = synth.capture(max(nsamples, 1), 2.4e6, "uc8", seed=33, frames_per_sec=3000, min_amp_db=-12, dc_offset=0.02)
    iq = iq[: 2 * nsamples]
    check_run(iq, "uc8", 2400000, expect_min_msgs=0)


def test_state_carries_between_runs():
    """Two runs on consecutive pieces of one stream = one run (convert_state persists between calls)."""
    iq, _ = synth.capture(131072 * 5, 2.4e6, "uc8", seed=34, frames_per_sec=1500, dc_offset=0.015 + 0.02j)
    exp, zexp = oracle_mags(iq, "uc8", 2400000)
    cut = 131072 * 2
    e = da.Engine(2400000, da.INPUT_UC8, filter_dc=True, compute_power=True)
    e.run_host(iq[: 2 * cut])
    a = e.dc_magnitudes(326, cut)
    e.run_host(iq[2 * cut:], first_sample=cut)
    b = e.dc_magnitudes(326, iq.size // 2 - cut)
    assert np.array_equal(np.concatenate([a, b]), exp)
    assert e.get_dc_state() == zexp


def test_demod_8000_behind_dc_converter():
    iq, _ = synth.capture(8_000_00, 8e6, "sc16", seed=35, frames_per_sec=2500, dc_offset=0.01 - 0.02j)
    check_run(iq, "sc16", 8000000, expect_min_msgs=50)


def test_long_recording_brackets_close():
    """40 M samples at 2.4 MS/s: chunks beyond the warm-up length start from a bracket around the filter's value in
    real arithmetic and must have closed on the reference's exact state by their first sample, or soon behind it
    (then only those first samples are stepped again)."""
    import torch
    from dump1090_amd import synth_torch
    n = 40_000_000
    iq_t, _ = synth_torch.capture_gpu(n, 2.4e6, "uc8", seed=36, frames_per_sec=800.0, dc_offset=(0.011, -0.017))
    iq = iq_t.cpu().numpy()
    e = da.Engine(2400000, da.INPUT_UC8, filter_dc=True, compute_power=True)
    e.run_device(iq_t.data_ptr(), iq.nbytes)
    torch.cuda.synchronize()
    info = e.dc_info()
    exp, zexp = oracle_mags(iq, "uc8", 2400000)
    got = e.dc_magnitudes(326, n)
    bad = np.flatnonzero(got != exp)
    assert bad.size == 0, "first differing sample %d (%d differ); %r" % (bad[0], bad.size, info)
    assert e.get_dc_state() == zexp
    # nearly all chunks are right as first stepped; the few whose bracket closed late are repaired
    assert info["chunks"] >= 128 and info["rerun_chunks"] + info["prefix_reruns"] <= info["chunks"] // 10, info
    o = orc.Oracle(2400000, fix_bits=1, measure_noise=True)
    o.run_stream(iq, orc.INPUT_UC8, 2400, filter_dc=True, compute_power=True)
    emsg, est = o.messages(), o.stats()
    assert len(emsg) > 5000
    assert_same_messages(e.messages(), emsg)
    # mag_buf.mean_power of every block: the converter's float running sum in sample order (ref: src/convert.c:146,153-154),
    # bit for bit, and with it Modes.stat.noise_power_* of the reference's shipped configuration (DC-filter = true,
    # measure-noise = true; ref: src/demod-2400.c:524-530)
    bp = e.block_power()
    assert len(bp) == len(o.last_block_power) == -(-n // 131072) and np.array_equal(bp, o.last_block_power)
    ps = e.power_stats()
    assert ps["noise_valid"] == 1 and ps["noise_power_count"] == est["noise_power_count"]
    np.testing.assert_allclose(ps["noise_power_sum"], est["noise_power_sum"], rtol=1e-9)
    np.testing.assert_allclose(ps["signal_power_sum"], est["signal_power_sum"], rtol=1e-9)

    # a shard decoded on its own: 14 M samples of lead-in settle the filter, the lead-in also yields the overlap
    first = 153 * 131072
    lead = 14_000_128
    s = da.Engine(2400000, da.INPUT_UC8, filter_dc=True, compute_power=True)
    s.set_dc_lead(lead)
    s.run_device(iq_t.data_ptr() + 2 * first, iq.nbytes - 2 * first, first_sample=first)
    got = s.dc_magnitudes(0, 326 + n - first)
    assert np.array_equal(got, exp[first - 326:])
    assert s.get_dc_state() == zexp
    assert np.array_equal(s.block_power(), o.last_block_power[153:])            # ... and the shard's blocks their powers

    # ... and a lead-in that is too short is refused, not guessed
    s.set_dc_lead(65536)
    with pytest.raises(da.ModesError):
        s.run_device(iq_t.data_ptr() + 2 * first, iq.nbytes - 2 * first, first_sample=first)


def test_mode_ac_behind_the_dc_converter():
    """demod_2400 + demod_2400_AC per block with the DC block on: the Mode A/C pass reads the magnitudes and the block
    powers the DC chain left (its threshold comes from mag_buf.mean_power, ref: src/demod-2400AC.c:34-35)."""
    from test_gpu_parity_2400ac import mode_a_stream
    iq16, _ = mode_a_stream(65536 * 5 + 999, 8, mode_s=0.5)
    # the same signal as uc8 with a DC offset (what the DC block is for)
    x = iq16.astype(np.float64) / 32767.0
    x[0::2] += 0.02
    x[1::2] -= 0.015
    iq = np.clip(np.round(x * 127.5 + 127.5), 0, 255).astype(np.uint8)
    for power in (False, True):
        o = orc.Oracle(2400000, fix_bits=1, mode_ac=False, measure_noise=power)
        o.run_stream(iq, orc.INPUT_UC8, 2400, filter_dc=True, compute_power=power, also_ac=True)
        exp, est = o.messages(), o.stats()
        e = da.Engine(2400000, da.INPUT_UC8, filter_dc=True, compute_power=power, mode_ac=True)
        e.run_host(iq)
        got, gst = e.messages(), e.stats()
        assert (exp["msg_type"] == 17).sum() > 50
        assert_same_messages(got, exp)
        for k in ("valid_preamble", "demod_modeac", "demodulated", "CRC_good"):
            assert gst[k] == est[k], k
        assert np.array_equal(e.block_power(), o.last_block_power)


def _quiet_uc8(n, seed, sigma, offs, bursts=True):
    """Noise of `sigma` LSB around a constant offset (where z1 settles on a point at which RN (z * B) changes its leak and
    crosses it every few samples: the rounds' first guess misses, lanes are proven a few at a time), with a few bursts."""
    rng = np.random.default_rng(seed)
    v = 127.5 + np.array(offs)[None, :] + sigma * rng.standard_normal((n, 2))
    if bursts:
        for p in rng.integers(0, n - 300, size=max(1, n // 400000)):
            v[p:p + 240:2] += 60.0
    return np.clip(np.rint(v), 0, 255).astype(np.uint8).reshape(-1)


@pytest.mark.parametrize("sigma,offs", [(1.5, (1.4, -2.17)), (0.7, (1.4, -2.17)), (0.3, (-3.2, 0.6)), (0.0, (0.5, 0.5))])
def test_noise_on_a_constant_offset(sigma, offs):
    """The regime the exact first guess of a round does not cover: every magnitude, the final filter state
    and the block powers must still be the sequential ones, whatever number of passes and partial rounds it takes."""
    n = 3_000_000
    iq = _quiet_uc8(n, 41, sigma, offs)
    e = check_run(iq, "uc8", 2400000, expect_min_msgs=0)
    assert e.dc_info()["chunks"] >= 32
    o = orc.Oracle(2400000, fix_bits=1, measure_noise=True)
    o.run_stream(iq, orc.INPUT_UC8, 2400, filter_dc=True, compute_power=True)
    est, gst = o.stats(), e.power_stats()
    assert gst["noise_valid"] == 1 and gst["noise_power_count"] == est["noise_power_count"]
    np.testing.assert_allclose(gst["noise_power_sum"], est["noise_power_sum"], rtol=1e-9)     # (a sum of per-block doubles: the blocks' values are bit-equal)


def test_silence_then_signal_sc16():
    """sc16 with stretches of exact zeros (z1 decays through the binades) between noise: ragged chunk ends included."""
    rng = np.random.default_rng(43)
    n = 2_500_003
    v = np.zeros((n, 2))
    for a in range(0, n, 500000):
        v[a:a + 230000] = 40.0 * rng.standard_normal((min(230000, n - a), 2)) + np.array([310.0, -120.0])
    iq = np.clip(np.rint(v), -32768, 32767).astype(np.int16).reshape(-1)
    check_run(iq, "sc16", 2400000, expect_min_msgs=0)
