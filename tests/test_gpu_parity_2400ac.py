"""GPU parity of demod_2400_AC (Mode A/C replies at 2.4 MS/s, BASELINE.json configs[4]) against the CPU oracle.

The demodulator's threshold comes from mag_buf.mean_power, which this fork's converters leave at 0 (integer
converters) or at a block *sum* (float converters): the reference's own `--infile` run therefore finds no reply
(SURVEY.md App. A.5).  The drop-in call is checked with hand-set mean_power values that make it decode; the bulk
engine is checked to agree with the oracle on the stream the reference would produce."""
import ctypes as C

import numpy as np
import pytest

import dump1090_amd as da
from dump1090_amd import synth
from oracle import orc
from util import assert_same_messages, oracle_run

pytestmark = pytest.mark.gpu


def mode_a_stream(n, seed, rate=2.4e6, mode_s=0.0):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.003
    pos, codes = 400, []
    while pos < n - 600:
        if rng.random() < mode_s:
            env = synth.envelope(synth.make_df17(0x4B1234, bytes(rng.integers(0, 256, 7, dtype=np.uint8))), rate, rng.random())
        else:
            code = int(rng.integers(0, 8)) << 12 | int(rng.integers(0, 8)) << 8 | int(rng.integers(0, 8)) << 4 | int(rng.integers(0, 8))
            env = synth.envelope_mode_a(code, rate, rng.random(), spi=rng.random() < 0.2)
            codes.append(code)
        x[pos:pos + len(env)] += 10 ** (rng.uniform(-14, -3) / 20) * np.exp(2j * np.pi * rng.random()) * env
        pos += len(env) + int(rng.integers(20, 500))
    return synth.quantise(x, "sc16"), codes


@pytest.mark.parametrize("mean_power", [0.0, 1e-6, 1e-4, 1e-3, 0.01, 7.5, 1e9])
def test_drop_in_demod_2400_AC_matches_oracle(mean_power):
    n = 300_000
    iq, codes = mode_a_stream(n, 5)
    cid = orc.convert_select(orc.INPUT_SC16, False, False)
    mag, _, _ = orc.convert(cid, iq, n)
    L = da.lib()
    L.modes_hip_bind(None)
    da.take_messages()
    o = orc.Oracle(2400000)
    tail = np.zeros(326, dtype=np.uint16)
    for k, a in enumerate(range(0, n, 65536)):
        data = np.concatenate([tail, mag[a:a + 65536]])
        tail = data[len(data) - 326:].copy()
        ts = int(a * 12e6 / 2.4e6)
        o.demod("2400AC", data, 326, sample_timestamp=ts, mean_power=mean_power)
        mb = da.make_mag_buf(data, 326, sample_timestamp=ts, mean_power=mean_power)
        L.demod_2400_AC(C.byref(mb))
    exp = o.messages()
    got = da.take_messages()
    if 1e-6 <= mean_power <= 1e-3:
        assert len(exp) > 300
    got["block"] = 0
    exp["block"] = 0
    assert_same_messages(got, exp)


def test_mode_ac_rescans_when_a_strip_runs_out_of_slots(monkeypatch):
    """MODES_STRIP_SLOTS=1: both the bulk A/C pass and the drop-in demod_2400_AC scan again with more slots."""
    monkeypatch.setenv("MODES_STRIP_SLOTS", "1")
    test_bulk_mode_s_then_mode_ac_per_block(False)
    test_drop_in_demod_2400_AC_matches_oracle(1e-4)


@pytest.mark.parametrize("power", [False, True])
def test_bulk_mode_s_then_mode_ac_per_block(power):
    """demod_2400 then demod_2400_AC on every block, as BASELINE.json configs[4] asks."""
    iq, _ = mode_a_stream(65536 * 3 + 999, 6, mode_s=0.5)
    exp, est = oracle_run(iq, "sc16", also_ac=True, compute_power=power)
    e = da.Engine(2400000, da.INPUT_SC16, mode_ac=True, compute_power=power)
    e.run_host(iq)
    got, gst = e.messages(), e.stats()
    assert (exp["msg_type"] == 17).sum() > 100
    assert_same_messages(got, exp)
    for k in ("valid_preamble", "demod_modeac", "demodulated", "CRC_good"):
        assert gst[k] == est[k], k


def test_block_power_matches_converter(modes1):
    """mean_power of every block as the float converter reports it (sequential float sum)."""
    iq, _ = synth.capture(65536 * 2 + 4321, 2.4e6, "sc16", seed=4, frames_per_sec=2000)
    o = orc.Oracle(2400000)
    o.run_stream(iq, orc.INPUT_SC16, 2400, compute_power=True)
    e = da.Engine(2400000, da.INPUT_SC16, mode_ac=True, compute_power=True)
    e.run_host(iq)
    got = e.block_power()
    assert np.array_equal(got, o.last_block_power)
