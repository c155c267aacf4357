"""GPU parity: the HIP engine behind the C ABI against the CPU oracle, 2.0 MS/s path (demod_2000,
BASELINE.json configs[2]: 1- and 2-bit CRC repair, dense traffic).  Bit-exact on every integer/byte field."""
import numpy as np
import pytest

import dump1090_amd as da
from dump1090_amd import synth
from util import assert_same_messages, oracle_run

pytestmark = pytest.mark.gpu
FMT = {"uc8": da.INPUT_UC8, "sc16": da.INPUT_SC16, "sc16q11": da.INPUT_SC16Q11}
STAT_KEYS = ("valid_preamble", "demodulated", "CRC_good", "CRC_bad", "CRC_single_bit_fix", "CRC_two_bits_fix",
             "demod_rejected_unknown", "out_of_phase", "demod_modeac")


def engine_run(iq, fmt, **kw):
    """Device accept pass (when the configuration has one) and the host walk must agree."""
    h = da.Engine(2000000, FMT[fmt], host_replay=True, **kw)
    h.run_host(iq)
    got_h, st_h = h.messages(), h.stats()
    if kw.get("phase_enhance"):
        return got_h, st_h
    e = da.Engine(2000000, FMT[fmt], **kw)
    e.run_host(iq)
    got, st = e.messages(), e.stats()
    assert got.tobytes() == got_h.tobytes()
    for k in st:
        if k not in ("replay_rounds", "candidates"):
            assert st[k] == st_h[k], k
    assert np.array_equal(e.get_icao(), h.get_icao())
    return got, st


def check(iq, fmt, min_msgs, **kw):
    exp, est = oracle_run(iq, fmt, rate=2000000, **kw)
    got, gst = engine_run(iq, fmt, **kw)
    assert len(exp) >= min_msgs, len(exp)
    assert_same_messages(got, exp)
    for k in STAT_KEYS:
        assert gst[k] == est[k], (k, gst[k], est[k])
    return got


@pytest.mark.parametrize("fix_bits", [0, 1, 2])
def test_sc16_dense_traffic_matches_oracle(fix_bits):
    iq, _ = synth.capture(2_000_000, 2.0e6, "sc16", seed=30 + fix_bits, frames_per_sec=5000, bitflip1=0.2, bitflip2=0.05,
                          ap_fraction=0.25)
    check(iq, "sc16", 1500, fix_bits=fix_bits)


def test_modes1_bin_lut_as_written(modes1):
    """modes1.bin at its native 2.0 MS/s through the reference-selected byte-LUT converter: 0 messages, 110
    valid preambles."""
    got = check(modes1, "uc8", 0)
    assert len(got) == 0


def test_weak_signals_with_ties():
    """Small magnitudes make equal sample pairs (encoding errors) common: the error budget, the DF-guess
    repair and the squelch all get exercised."""
    iq, _ = synth.capture(1_500_000, 2.0e6, "sc16", seed=41, frames_per_sec=4000, noise_sigma=0.0004,
                          min_amp_db=-60, max_amp_db=-44, ap_fraction=0.3, n_icao=6)
    check(iq, "sc16", 50, fix_bits=2)


def test_sc16_generic_power_path():
    iq, _ = synth.capture(700_000, 2.0e6, "sc16", seed=8, frames_per_sec=3000)
    check(iq, "sc16", 300, compute_power=True)


def test_sc16q11():
    iq, _ = synth.capture(600_000, 2.0e6, "sc16q11", seed=9, frames_per_sec=3000, min_amp_db=-20)
    check(iq, "sc16q11", 100)


@pytest.mark.parametrize("nsamples", [0, 1, 271, 272, 273, 2047, 2048, 2049, 65536, 65537, 131072 + 17])
def test_ragged_lengths(nsamples):
    iq, _ = synth.capture(max(nsamples, 1), 2.0e6, "sc16", seed=3, frames_per_sec=6000, min_amp_db=-12)
    iq = iq[: 2 * nsamples]
    check(iq, "sc16", 0)


def test_messages_straddling_block_boundaries():
    rng = np.random.default_rng(17)
    offs = (-280, -272, -260, -240, -200, -135, -100, -20, -1, 0, 1, 40)
    n = 65536 * (len(offs) + 1) + 1000
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.005
    for b, d in enumerate(offs, start=1):
        for q in (0, -600):
            msg = synth.make_df17(0x4840D6 + b, bytes(rng.integers(0, 256, 7, dtype=np.uint8)))
            env = synth.envelope(msg, 2.0e6, rng.random() * 0.2)
            p = b * 65536 + d + q
            x[p:p + len(env)] += 0.3 * np.exp(2j * np.pi * rng.random()) * env
    iq = synth.quantise(x, "sc16")
    check(iq, "sc16", 2 * len(offs) - 4)


@pytest.mark.parametrize("seed", [51, 52])
def test_phase_enhance_retries(seed):
    """Modes.phase_enhance: phase-corrected retry at a failed preamble and behind a repaired message."""
    iq, _ = synth.capture(2_000_000, 2.0e6, "sc16", seed=seed, frames_per_sec=4000, bitflip1=0.3, bitflip2=0.1,
                          ap_fraction=0.25, overlap_fraction=0.3)
    exp, est = oracle_run(iq, "sc16", rate=2000000, fix_bits=2, phase_enhance=True)
    got, gst = engine_run(iq, "sc16", fix_bits=2, phase_enhance=True)
    assert (exp["phase"] == 1).sum() > 10 and est["out_of_phase"] > 100
    assert_same_messages(got, exp)
    for k in STAT_KEYS:
        assert gst[k] == est[k], (k, gst[k], est[k])


def mode_a_capture(n, seed, with_mode_s=True):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.004
    pos, codes = 500, []
    while pos < n - 1000:
        if with_mode_s and rng.random() < 0.4:
            msg = synth.make_df17(int(rng.integers(0x400000, 0x400008)), bytes(rng.integers(0, 256, 7, dtype=np.uint8)))
            env = synth.envelope(msg, 2.0e6, rng.random())
        else:
            code = int(rng.integers(0, 8)) << 12 | int(rng.integers(0, 8)) << 8 | int(rng.integers(0, 8)) << 4 | int(rng.integers(0, 8))
            env = synth.envelope_mode_a(code, 2.0e6, rng.random(), spi=rng.random() < 0.2)
            codes.append(code)
        amp = 10 ** (rng.uniform(-14, -3) / 20)
        x[pos:pos + len(env)] += amp * np.exp(2j * np.pi * rng.random()) * env
        pos += len(env) + int(rng.integers(60, 700))
    return synth.quantise(x, "sc16"), codes


def test_mode_ac_replies():
    """Modes.mode_AC: detect_mode_A runs ahead of the Mode S preamble test at every position."""
    iq, codes = mode_a_capture(1_000_000, 61)
    exp, est = oracle_run(iq, "sc16", rate=2000000, mode_ac=True)
    got, gst = engine_run(iq, "sc16", mode_ac=True)
    assert est["demod_modeac"] > 0.3 * len(codes) and (exp["msg_type"] == 17).sum() > 50
    assert_same_messages(got, exp)
    for k in STAT_KEYS:
        assert gst[k] == est[k], (k, gst[k], est[k])


def test_mode_ac_with_phase_enhance():
    iq, _ = mode_a_capture(600_000, 62)
    exp, est = oracle_run(iq, "sc16", rate=2000000, mode_ac=True, phase_enhance=True, fix_bits=2)
    got, gst = engine_run(iq, "sc16", mode_ac=True, phase_enhance=True, fix_bits=2)
    assert_same_messages(got, exp)
    for k in STAT_KEYS:
        assert gst[k] == est[k], (k, gst[k], est[k])
