"""GPU parity: the HIP engine behind the C ABI against the CPU oracle, 8 MS/s path (demod_8000, BASELINE.json
configs[3]).  Bit-exact on every integer/byte field; sig_level within 1e-5 relative."""
import numpy as np
import pytest

import dump1090_amd as da
from dump1090_amd import synth
from util import assert_same_messages, oracle_run

pytestmark = pytest.mark.gpu
FMT = {"uc8": da.INPUT_UC8, "sc16": da.INPUT_SC16, "sc16q11": da.INPUT_SC16Q11}
STAT_KEYS = ("valid_preamble", "demodulated", "CRC_good", "CRC_bad", "CRC_single_bit_fix", "CRC_two_bits_fix",
             "demod_rejected_unknown")


def check(iq, fmt, min_msgs, **kw):
    exp, est = oracle_run(iq, fmt, rate=8000000, **kw)
    e = da.Engine(8000000, FMT[fmt], **kw)
    e.run_host(iq)
    got, gst = e.messages(), e.stats()
    assert len(exp) >= min_msgs, len(exp)
    assert_same_messages(got, exp)
    for k in STAT_KEYS:
        assert gst[k] == est[k], (k, gst[k], est[k])
    return got


@pytest.mark.parametrize("fix_bits", [0, 1, 2])
def test_sc16_synthetic_matches_oracle(fix_bits):
    iq, _ = synth.capture(4_000_000, 8.0e6, "sc16", seed=70 + fix_bits, frames_per_sec=2000, bitflip1=0.2, bitflip2=0.05,
                          ap_fraction=0.25)
    check(iq, "sc16", 500, fix_bits=fix_bits)


def test_quiet_input_needs_the_running_average():
    """Noise of a few LSB keeps max(phase) in the band where the trigger depends on phase_av."""
    iq, _ = synth.capture(1_500_000, 8.0e6, "sc16", seed=81, frames_per_sec=1500, noise_sigma=0.0003,
                          min_amp_db=-50, max_amp_db=-30, ap_fraction=0.2)
    check(iq, "sc16", 100)


def test_silence_then_signal():
    iq, _ = synth.capture(1_200_000, 8.0e6, "sc16", seed=82, frames_per_sec=2000)
    iq[: 2 * 300_000] = 0
    iq[2 * 700_000: 2 * 800_000] = 0
    check(iq, "sc16", 50)


def test_sc16_generic_power_path():
    iq, _ = synth.capture(1_000_000, 8.0e6, "sc16", seed=83, frames_per_sec=2000)
    check(iq, "sc16", 100, compute_power=True)


@pytest.mark.parametrize("nsamples", [0, 7, 8, 543, 544, 1400, 2047, 65536, 65537, 65536 * 2 + 4321, 65536 * 8 + 700, 65536 * 16 + 1300])
def test_ragged_lengths(nsamples):
    iq, _ = synth.capture(max(nsamples, 1), 8.0e6, "sc16", seed=84, frames_per_sec=4000, min_amp_db=-12)
    iq = iq[: 2 * nsamples]
    check(iq, "sc16", 0)


def test_messages_straddling_block_boundaries():
    rng = np.random.default_rng(27)
    offs = (-1400, -1100, -960, -900, -700, -480, -300, -100, -8, 0, 30)
    n = 65536 * (len(offs) + 1) + 3000
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.005
    for b, d in enumerate(offs, start=1):
        msg = synth.make_df17(0x4840D6 + b, bytes(rng.integers(0, 256, 7, dtype=np.uint8)))
        env = synth.envelope(msg, 8.0e6, rng.random())
        p = b * 65536 + d
        x[p:p + len(env)] += 0.3 * np.exp(2j * np.pi * rng.random()) * env
    iq = synth.quantise(x, "sc16")
    check(iq, "sc16", len(offs) - 3)


def test_windows_far_apart_take_a_tile_each():
    """Bursts in silence: the four windows a workgroup of d8_window_kernel takes at a time lie further apart than its
    shared tile is long, so every wave stages its own quarter (the other tests' windows tile the stream)."""
    rng = np.random.default_rng(91)
    nb = 120
    x = np.zeros(4000 + 9000 * nb, dtype=np.complex128)
    for b in range(nb):
        p = 4000 + 9000 * b + int(rng.integers(0, 64))
        x[p - 600:p + 1700] = (rng.standard_normal(2300) + 1j * rng.standard_normal(2300)) * 0.004
        msg = synth.make_df17(0x3C6000 + b, bytes(rng.integers(0, 256, 7, dtype=np.uint8)))
        env = synth.envelope(msg, 8.0e6, rng.random())
        x[p:p + len(env)] += 0.25 * np.exp(2j * np.pi * rng.random()) * env
    iq = synth.quantise(x, "sc16")
    check(iq, "sc16", nb // 2)
    check(synth.quantise(x, "uc8"), "uc8", 0)


def test_chain_through_more_than_one_group_of_chunks():
    """The window chain is resolved over groups of 64 chunks of 65536 steps: 40 M samples are 77 chunks, two groups."""
    iq, _ = synth.capture(2_500_000, 8.0e6, "sc16", seed=93, frames_per_sec=1500, ap_fraction=0.2)
    check(np.tile(iq, 16), "sc16", 1000)


def test_long_recording_runs_in_chunk_ranges():
    """Sixteen chunks and more (8.4 M samples) make the engine walk the windows in four ranges, each range's ordered
    pass on the host overlapping the next range's kernels: the message list must not notice the seams."""
    import torch
    from dump1090_amd import synth_torch
    n = 17 * 65536 * 8 + 12345
    d, _ = synth_torch.capture_gpu(n, 8.0e6, "sc16", seed=91, frames_per_sec=2500, device="cuda:0", n_icao=8, n_bank=512)
    torch.cuda.synchronize()
    iq = d.cpu().numpy().view(np.int16)
    exp, est = oracle_run(iq, "sc16", rate=8000000, fix_bits=1)
    e = da.Engine(8000000, da.INPUT_SC16, fix_bits=1)
    e.run_device(d.data_ptr(), d.numel() * d.element_size())
    got, gst = e.messages(), e.stats()
    assert len(exp) > 1000
    assert_same_messages(got, exp)
    for k in STAT_KEYS:
        assert gst[k] == est[k], (k, gst[k], est[k])


def test_planes_range_by_range_equal_planes_first(monkeypatch):
    """Sixteen chunks and more: the trigger planes, the chain and the windows are formed range by range (the host walks
    range r while the device forms the planes of range r + 2).  Ranges that are not whole groups of 64 chunks, three to
    seven of them, against the run that forms all planes first -- and against the oracle."""
    iq, _ = synth.capture(2_300_000, 8.0e6, "sc16", seed=95, frames_per_sec=2000, ap_fraction=0.25, n_icao=8)
    iq = np.tile(iq, 4)                                           # 9.2 M samples: 17.5 chunks
    exp, est = oracle_run(iq, "sc16", rate=8000000)
    assert len(exp) > 1000
    for ranges, plain in ((None, True), (None, False), (3, False), (5, False), (7, False), (17, False)):
        if ranges is None:
            monkeypatch.delenv("MODES_RANGES", raising=False)
        else:
            monkeypatch.setenv("MODES_RANGES", str(ranges))
        if plain:
            monkeypatch.setenv("MODES_D8_NO_PIPELINE", "1")
        else:
            monkeypatch.delenv("MODES_D8_NO_PIPELINE", raising=False)
        e = da.Engine(8000000, da.INPUT_SC16)
        e.run_host(iq)
        assert_same_messages(e.messages(), exp)
        for k in STAT_KEYS:
            assert e.stats()[k] == est[k], (ranges, plain, k)
        assert e.stats()["scan_retries"] == 0


def test_quiet_stretch_in_a_later_range_ends_the_bet():
    """Forming the planes range by range bets that no step behind the first range needs the running average.  A
    near-silent stretch in the second half of a long recording loses it: the run starts again with all planes first
    (stats.scan_retries says so) and the list is the oracle's."""
    loud, _ = synth.capture(4_700_000, 8.0e6, "sc16", seed=96, frames_per_sec=2000, ap_fraction=0.2, n_icao=8)
    quiet, _ = synth.capture(4_600_000, 8.0e6, "sc16", seed=97, frames_per_sec=1500, noise_sigma=0.0003,
                             min_amp_db=-50, max_amp_db=-30, ap_fraction=0.2, n_icao=8)
    iq = np.concatenate([loud, quiet])                            # 17.7 chunks, four ranges
    exp, est = oracle_run(iq, "sc16", rate=8000000)
    assert len(exp) > 500
    e = da.Engine(8000000, da.INPUT_SC16)
    e.run_host(iq)
    assert e.stats()["scan_retries"] == 1
    assert_same_messages(e.messages(), exp)
    for k in STAT_KEYS:
        assert e.stats()[k] == est[k], k
    # the quiet stretch first: its steps are the first range's, the bet holds
    iq2 = np.concatenate([quiet[: 2 * 2_000_000], loud, loud[: 2 * 2_500_000]])
    exp2, est2 = oracle_run(iq2, "sc16", rate=8000000)
    e2 = da.Engine(8000000, da.INPUT_SC16)
    e2.run_host(iq2)
    assert e2.stats()["scan_retries"] == 0
    assert_same_messages(e2.messages(), exp2)
    for k in STAT_KEYS:
        assert e2.stats()[k] == est2[k], k


def test_range_that_outgrows_its_pinned_share_is_gathered_again():
    """The packed windows / records of a range go straight to pinned host arrays sized from the last run (or a default
    per chunk); a range that does not fit is gathered again into arrays of its own size (MODES_D8_BUDGET: test hook)."""
    import os
    iq, _ = synth.capture(17 * 65536 * 8 // 2 + 999, 8.0e6, "sc16", seed=92, frames_per_sec=2500, n_icao=8)
    iq = np.concatenate([iq, iq])                                 # 17+ chunks: four ranges
    exp, est = oracle_run(iq, "sc16", rate=8000000)
    assert len(exp) > 1000
    os.environ["MODES_D8_BUDGET"] = "1"
    try:
        e = da.Engine(8000000, da.INPUT_SC16)
        e.run_host(iq)
    finally:
        del os.environ["MODES_D8_BUDGET"]
    assert e.stats()["scan_retries"] >= 1
    assert_same_messages(e.messages(), exp)
    for k in STAT_KEYS:
        assert e.stats()[k] == est[k], k
    # a second run on the same engine sizes the arrays from the first: nothing is gathered twice
    e.set_icao(np.zeros(1024, dtype=np.uint32))
    e.run_host(iq)
    assert e.stats()["scan_retries"] == 0
    assert_same_messages(e.messages(), exp)


def test_round_over_the_same_windows_stops_where_it_meets_the_round_before(monkeypatch):
    """A hand-over round of a sharded decode runs over the same windows with another entry cache.  It walks the ranges of
    chunks until its cache is the one the round before had at the same boundary, and takes the rest from that round:
    same messages, statistics and final cache as the complete walk, in fewer ranges (stats.replay_rounds)."""
    import torch
    from dump1090_amd import synth_torch
    n = 17 * 65536 * 8 + 4321                                     # 17+ chunks: four ranges
    d, _ = synth_torch.capture_gpu(n, 8.0e6, "sc16", seed=93, frames_per_sec=2500, device="cuda:0", n_icao=8, n_bank=512)
    torch.cuda.synchronize()
    nbytes = d.numel() * d.element_size()
    # an entry cache as a shard before this one would leave it: some of the aircraft known, some not
    first = da.Engine(8000000, da.INPUT_SC16, fix_bits=1)
    first.run_device(d.data_ptr(), nbytes // 8 // 16 * 16)
    entry = first.get_icao()
    assert np.count_nonzero(entry) >= 3

    def rounds(early):
        if early:
            monkeypatch.delenv("MODES_D8_NO_EARLY_STOP", raising=False)
        else:
            monkeypatch.setenv("MODES_D8_NO_EARLY_STOP", "1")
        e = da.Engine(8000000, da.INPUT_SC16, fix_bits=1)
        e.set_icao(None)
        e.run_device(d.data_ptr(), nbytes)                        # round one: cold cache
        cold = e.messages().copy()
        out = []
        for cache in (entry, entry, None):                        # a changed entry, the same again, back to cold
            e.set_icao(cache)
            e.d8_reuse_planes()
            e.run_device(d.data_ptr(), nbytes)
            out.append((e.messages().copy(), dict(e.stats()), e.get_icao().copy()))
        return cold, out

    cold_a, full = rounds(False)
    cold_b, fast = rounds(True)
    assert_same_messages(cold_a, cold_b)
    assert len(cold_a) > 1000
    assert len(full[0][0]) != len(cold_a) or not np.array_equal(full[0][0]["score"], cold_a["score"])   # the entry cache matters
    for (m0, s0, c0), (m1, s1, c1) in zip(full, fast):
        assert_same_messages(m1, m0)
        np.testing.assert_array_equal(m1["sig_level"], m0["sig_level"])
        np.testing.assert_array_equal(c1, c0)
        for k in STAT_KEYS:
            assert s1[k] == s0[k], k
    assert all(s["replay_rounds"] == 0 for _, s, _ in full)
    assert 1 <= fast[0][1]["replay_rounds"] <= 3                  # met the cold round inside the four ranges
    assert fast[1][1]["replay_rounds"] == 1                       # the same entry again: met behind the first range
    # and the whole thing is what a fresh engine makes of that entry cache
    ref = da.Engine(8000000, da.INPUT_SC16, fix_bits=1)
    ref.set_icao(entry)
    ref.run_device(d.data_ptr(), nbytes)
    assert_same_messages(fast[0][0], ref.messages())
