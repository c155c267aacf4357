"""GPU parity: the HIP engine behind the C ABI against the CPU oracle, 2.4 MS/s path.
Bit-exact on every integer/byte field; sig_level within 1e-5 relative (north_star)."""
import numpy as np
import pytest

import dump1090_amd as da
from dump1090_amd import synth
from oracle import orc
from util import assert_same_messages, oracle_run

pytestmark = pytest.mark.gpu
FMT = {"uc8": da.INPUT_UC8, "sc16": da.INPUT_SC16, "sc16q11": da.INPUT_SC16Q11}


def assert_power_stats(e, est, noise):
    """Modes.stat.signal_power_* / noise_power_* (ref: src/demod-2400.c:498-505,524-530)."""
    ps = e.power_stats()
    assert ps["signal_power_count"] == est["signal_power_count"] and ps["strong_signal_count"] == est["strong_signal_count"]
    np.testing.assert_allclose(ps["signal_power_sum"], est["signal_power_sum"], rtol=1e-9)
    np.testing.assert_allclose(ps["peak_signal_power"], est["peak_signal_power"], rtol=1e-12)
    assert ps["noise_valid"] == int(noise)
    if noise:
        assert ps["noise_power_count"] == est["noise_power_count"]
        np.testing.assert_allclose(ps["noise_power_sum"], est["noise_power_sum"], rtol=1e-9)


def engine_run(iq, fmt, **kw):
    """Runs the engine twice -- ordered accept pass on the device (default) and on the host -- and checks
    that both give the same messages and statistics before handing them to the oracle comparison."""
    e = da.Engine(2400000, FMT[fmt], **kw)
    e.run_host(iq)
    got, st = e.messages(), e.stats()
    engine_run.last = e
    h = da.Engine(2400000, FMT[fmt], host_replay=True, **kw)
    h.run_host(iq)
    got_h, st_h = h.messages(), h.stats()
    assert got.tobytes() == got_h.tobytes()
    for k in st:
        if k != "replay_rounds":
            assert st[k] == st_h[k], k
    assert np.array_equal(e.get_icao(), h.get_icao())
    return got, st


@pytest.mark.parametrize("fix_bits", [0, 1, 2])
def test_sc16_synthetic_matches_oracle(fix_bits):
    iq, frames = synth.capture(2_400_000, 2.4e6, "sc16", seed=10 + fix_bits, frames_per_sec=1500, bitflip1=0.2,
                               bitflip2=0.1, ap_fraction=0.25)
    exp, est = oracle_run(iq, "sc16", fix_bits=fix_bits)
    got, gst = engine_run(iq, "sc16", fix_bits=fix_bits)
    assert len(exp) > 500
    assert_same_messages(got, exp)
    for k in ("valid_preamble", "demodulated", "CRC_good", "CRC_bad", "CRC_single_bit_fix", "CRC_two_bits_fix",
              "out_of_phase", "demod_rejected_unknown"):
        assert gst[k] == est[k], k


def test_sc16_phase_enhance_matches_oracle():
    iq, _ = synth.capture(1_200_000, 2.4e6, "sc16", seed=21, frames_per_sec=2000, bitflip1=0.2, ap_fraction=0.2)
    exp, est = oracle_run(iq, "sc16", fix_bits=2, phase_enhance=True)
    got, gst = engine_run(iq, "sc16", fix_bits=2, phase_enhance=True)
    assert len(exp) > 300
    assert_same_messages(got, exp)
    assert gst["valid_preamble"] == est["valid_preamble"] and gst["out_of_phase"] == est["out_of_phase"]


@pytest.mark.parametrize("seed", [31, 32])
def test_dense_traffic_skip_ahead(seed):
    """Frames nearly back to back: the scan kernel drops candidates inside a message the reference certainly accepts
    (its skip-ahead, ref: src/demod-2400.c:514) before slicing them; messages and statistics must not notice."""
    iq, _ = synth.capture(1_500_000, 2.4e6, "sc16", seed=seed, frames_per_sec=6500, ap_fraction=0.3, bitflip1=0.1, n_icao=6)
    exp, est = oracle_run(iq, "sc16", fix_bits=1)
    got, gst = engine_run(iq, "sc16", fix_bits=1)
    assert len(exp) > 1500
    assert_same_messages(got, exp)
    for k in ("valid_preamble", "demodulated", "CRC_good", "CRC_bad", "CRC_single_bit_fix", "out_of_phase", "demod_rejected_unknown"):
        assert gst[k] == est[k], k


def test_uc8_lut_as_written_matches_oracle(modes1):
    """BASELINE config 1: modes1.bin at 2.4 MS/s through the reference-selected converter."""
    exp, est = oracle_run(modes1, "uc8")
    got, gst = engine_run(modes1, "uc8")
    assert len(exp) == 0 and len(got) == 0
    assert gst["valid_preamble"] == est["valid_preamble"] == 2016


@pytest.mark.parametrize("compute_power", [False, True])
def test_uc8_lut_image_messages_match_oracle(compute_power):
    """The headline instantiation (scan2400_kernel<kConvUc8Lut>: convert_uc8_nodc_nopower / _power AS WRITTEN, one
    byte per output sample, ref: src/convert.c:19-40) on an input it decodes: a byte stream whose image under the
    256-entry table is a frame train (synth.capture_lut_image), eight blocks, frames across every block seam --
    which for this converter is also the jump over the never-read second half of the block."""
    iq, frames = synth.capture_lut_image(131072 * 8 + 7001, 2.4e6, seed=41, frames_per_sec=4500, min_amp_db=-22,
                                         bitflip1=0.2, ap_fraction=0.2)
    exp, est = oracle_run(iq, "uc8", fix_bits=1, compute_power=compute_power)
    got, gst = engine_run(iq, "uc8", fix_bits=1, compute_power=compute_power)
    assert len(exp) > 1000
    assert_same_messages(got, exp)
    for k in ("valid_preamble", "demodulated", "CRC_good", "CRC_bad", "CRC_single_bit_fix", "out_of_phase", "demod_rejected_unknown"):
        assert gst[k] == est[k], k
    # ground truth of the generator, not only the oracle: most injected frames come back with their bytes
    sent = {f["clean"] for f in frames}
    back = sum(1 for g in got if bytes(g["msg"][: int(g["msg_bits"]) // 8]) in sent)
    assert back > 0.7 * len(frames)
    assert_power_stats(engine_run.last, est, noise=compute_power)


@pytest.mark.parametrize("shift,variant", [(0, None), (1, None), (8, None), (0, "MODES_RANK_TILE_GATHER"), (0, "MODES_NO_RANK_TILE")])
def test_uc8_lut_image_at_any_address_and_with_every_tile(shift, variant, monkeypatch):
    """The byte-table scan fills its tile by LDS DMA when the input allows it (16-byte aligned) and through registers
    (rank tile, magnitude tile) when it does not: same messages, same statistics, wherever the buffer lies."""
    import torch
    iq, _ = synth.capture_lut_image(131072 * 5 + 3333, 2.4e6, seed=43, frames_per_sec=4000, min_amp_db=-22, bitflip1=0.2, ap_fraction=0.2)
    exp, est = oracle_run(iq, "uc8", fix_bits=1)
    assert len(exp) > 500
    if variant:
        monkeypatch.setenv(variant, "1")
    buf = torch.zeros(iq.size + 64, dtype=torch.uint8, device="cuda:0")
    buf[shift: shift + iq.size] = torch.from_numpy(iq).to("cuda:0")
    assert buf.data_ptr() % 16 == 0
    e = da.Engine(2400000, da.INPUT_UC8, fix_bits=1)
    e.run_device(buf.data_ptr() + shift, iq.size)
    assert_same_messages(e.messages(), exp)
    gst = e.stats()
    for k in ("valid_preamble", "demodulated", "CRC_good", "CRC_bad", "CRC_single_bit_fix", "out_of_phase", "demod_rejected_unknown"):
        assert gst[k] == est[k], k


def test_uc8_lut_image_frames_on_block_seams():
    """Frames placed around every 131072-sample block edge of the as-written uc8 converter's output."""
    rng = np.random.default_rng(43)
    offs = (-330, -300, -268, -200, -135, -100, -20, -1, 0, 1, 40)
    B = synth.LUT_BLOCK_SAMPLES
    n = B * (len(offs) + 1) + 1000
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.01
    for b, d in enumerate(offs, start=1):
        for q, make in ((0, synth.make_df17), (-600, synth.make_df17)):
            msg = make(0x4840D6 + b, bytes(rng.integers(0, 256, 7, dtype=np.uint8)))
            env = synth.envelope(msg, 2.4e6, rng.random())
            p = b * B + d + q
            x[p:p + len(env)] += 0.4 * env
    iq = synth.lut_image_bytes(np.clip(np.round(np.abs(x) * 128.0), 0, 128).astype(np.uint8), 43)
    exp, _ = oracle_run(iq, "uc8")
    got, _ = engine_run(iq, "uc8")
    assert len(exp) >= 2 * len(offs) - 2
    assert_same_messages(got, exp)


def test_sc16_generic_power_path_matches_oracle():
    """compute_power selects convert_sc16_generic (float path, DC block off)."""
    iq, _ = synth.capture(800_000, 2.4e6, "sc16", seed=5, frames_per_sec=1500)
    exp, est = oracle_run(iq, "sc16", compute_power=True)
    got, gst = engine_run(iq, "sc16", compute_power=True)
    assert len(exp) > 200
    assert_same_messages(got, exp)
    assert gst["valid_preamble"] == est["valid_preamble"]
    assert_power_stats(engine_run.last, est, noise=True)


def test_sc16q11_matches_oracle():
    iq, _ = synth.capture(600_000, 2.4e6, "sc16q11", seed=6, frames_per_sec=1500, min_amp_db=-20)
    exp, est = oracle_run(iq, "sc16q11")
    got, gst = engine_run(iq, "sc16q11")
    assert len(exp) > 100
    assert_same_messages(got, exp)


@pytest.mark.parametrize("nsamples", [0, 1, 300, 4095, 4096, 4097, 65536, 65537, 131072 + 17])
def test_ragged_lengths(nsamples):
    iq, _ = synth.capture(max(nsamples, 1), 2.4e6, "sc16", seed=3, frames_per_sec=3000, min_amp_db=-12)
    iq = iq[: 2 * nsamples]
    exp, est = oracle_run(iq, "sc16")
    got, gst = engine_run(iq, "sc16")
    assert_same_messages(got, exp)
    assert gst["valid_preamble"] == est["valid_preamble"]


def test_messages_straddling_block_boundaries():
    """Frames placed across the 65536-sample block edges: skip-ahead resets per block, overlap carries."""
    rng = np.random.default_rng(7)
    offs = (-330, -300, -268, -200, -135, -100, -20, -1, 0, 1, 40)
    n = 65536 * (len(offs) + 1) + 1000
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.005
    for b, d in enumerate(offs, start=1):
        for q, make in ((0, synth.make_df17), (-600, synth.make_df17)):
            msg = make(0x4840D6 + b, bytes(rng.integers(0, 256, 7, dtype=np.uint8)))
            env = synth.envelope(msg, 2.4e6, rng.random())
            p = b * 65536 + d + q
            x[p:p + len(env)] += 0.3 * np.exp(2j * np.pi * rng.random()) * env
    iq = synth.quantise(x, "sc16")
    exp, _ = oracle_run(iq, "sc16")
    got, _ = engine_run(iq, "sc16")
    assert len(exp) >= 2 * len(offs) - 2
    assert_same_messages(got, exp)


def test_icao_cache_chaining_across_runs():
    """A shard run with the cache a previous shard left must equal the oracle's single pass."""
    iq, _ = synth.capture(65536 * 12, 2.4e6, "sc16", seed=11, frames_per_sec=2000, ap_fraction=0.5, n_icao=4)
    exp, _ = oracle_run(iq, "sc16")
    e = da.Engine(2400000, da.INPUT_SC16)
    e.run_host(iq)
    assert_same_messages(e.messages(), exp)
    assert (exp["msg_type"] == 4).sum() + (exp["msg_type"] == 5).sum() + (exp["msg_type"] == 0).sum() > 20


def test_strip_overflow_is_rescanned(monkeypatch):
    """Two candidate slots per strip overflow on any real signal: the engine scans again with more slots until
    everything fits, and the result is the oracle's."""
    iq, _ = synth.capture(600_000, 2.4e6, "sc16", seed=12, frames_per_sec=3000, bitflip1=0.1, ap_fraction=0.2)
    exp, est = oracle_run(iq, "sc16")
    monkeypatch.setenv("MODES_STRIP_SLOTS", "2")
    for host_replay in (False, True):
        e = da.Engine(2400000, da.INPUT_SC16, host_replay=host_replay)
        e.run_host(iq)
        assert e.stats()["scan_retries"] >= 1
        assert_same_messages(e.messages(), exp)
        assert e.stats()["valid_preamble"] == est["valid_preamble"]


def test_rank_tile_falls_back_on_a_halo_the_table_cannot_produce():
    """The byte-table scan keeps ranks of the table's values in LDS (kConvUc8LutR8).  The overlap in front of a shard is
    the one input that does not come through the table: with a halo formed by the converter the rank tile is exact; a
    halo holding any other magnitude makes the engine scan again with the magnitude tile -- same result either way."""
    import os
    import torch
    n = 131072 * 5
    iq, _ = synth.capture_lut_image(n, 2.4e6, seed=61, frames_per_sec=3000, min_amp_db=-22, ap_fraction=0.3, n_icao=6)
    d = torch.from_numpy(iq).to("cuda:0")
    first = 131072 * 2

    def shard_with(halo, plain):
        if plain:
            os.environ["MODES_NO_RANK_TILE"] = "1"
        try:
            e = da.Engine(2400000, da.INPUT_UC8)
            e.run_device(d.data_ptr() + 2 * first, 2 * (n - first), first_sample=first, halo_ptr=halo.data_ptr())
            torch.cuda.synchronize()
            return e
        finally:
            os.environ.pop("MODES_NO_RANK_TILE", None)
    lead = da.Engine(2400000, da.INPUT_UC8)
    lead.run_device(d.data_ptr(), 2 * first)
    ov = 326
    good = torch.empty(ov + 8, dtype=torch.int16, device="cuda:0")
    lead.halo_at(d.data_ptr(), 2 * n, first, good.data_ptr())
    torch.cuda.synchronize()
    a, b = shard_with(good, False), shard_with(good, True)
    assert a.stats()["scan_retries"] == 0 and len(a.messages()) > 200
    assert a.messages().tobytes() == b.messages().tobytes()
    for k in ("valid_preamble", "candidates", "demodulated"):
        assert a.stats()[k] == b.stats()[k], k
    odd = good.clone()
    odd[ov - 40:ov] += 7                                         # 360 * k + 7: no byte maps to that
    a, b = shard_with(odd, False), shard_with(odd, True)
    assert a.stats()["scan_retries"] == 1 and b.stats()["scan_retries"] == 0
    assert a.messages().tobytes() == b.messages().tobytes()
    for k in ("valid_preamble", "candidates", "demodulated"):
        assert a.stats()[k] == b.stats()[k], k
