"""GPU parity: a long shard decoded in ranges of blocks (scan of range r + 1 overlapping the ordered pass, emit and
read-back of range r; engine.cpp scan_and_accept_in_ranges) gives the single pass's list, statistics and ICAO cache,
and both equal the oracle's (ref: src/infile.c:116-197 is one loop over blocks; ranges are an execution detail)."""
import os

import numpy as np
import pytest

import dump1090_amd as da
from dump1090_amd import synth
from util import assert_same_messages, oracle_run

pytestmark = pytest.mark.gpu
FMT = {"uc8": da.INPUT_UC8, "sc16": da.INPUT_SC16}
STAT_KEYS = ("valid_preamble", "demodulated", "CRC_good", "CRC_bad", "CRC_single_bit_fix", "CRC_two_bits_fix",
             "out_of_phase", "demod_rejected_unknown", "candidates")


@pytest.fixture
def ranges_env():
    saved = {k: os.environ.get(k) for k in ("MODES_RANGES", "MODES_STRIP_SLOTS")}

    def set_env(ranges=None, slots=None):
        for k, v in (("MODES_RANGES", ranges), ("MODES_STRIP_SLOTS", slots)):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
    yield set_env
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def run(iq, fmt, rate, ranges, set_env, slots=None, **kw):
    set_env(ranges, slots)
    e = da.Engine(rate, FMT[fmt], **kw)
    e.run_host(iq)
    set_env()
    return e


@pytest.mark.parametrize("fmt,rate,nblocks", [("sc16", 2400000, 23), ("uc8", 2400000, 11), ("sc16", 2000000, 17), ("uc8", 2000000, 9)])
def test_ranges_equal_single_pass_and_oracle(fmt, rate, nblocks, ranges_env):
    block = 262144 // (2 if fmt == "uc8" else 4)
    n = nblocks * block - 4321                                   # a ragged last block
    if fmt == "uc8":        # the as-written byte table reads one byte per sample: a buffer whose image is a capture
        iq, _ = synth.capture_lut_image(n, float(rate), seed=71 + nblocks, frames_per_sec=2500, bitflip1=0.15, ap_fraction=0.3)
    else:
        iq, _ = synth.capture(n, float(rate), fmt, seed=71 + nblocks, frames_per_sec=2500, bitflip1=0.15, ap_fraction=0.3)
    exp, est = oracle_run(iq, fmt, rate=rate)
    assert len(exp) > 300
    one = run(iq, fmt, rate, 1, ranges_env)
    assert_same_messages(one.messages(), exp)
    for r in (2, 3, 5, nblocks):
        e = run(iq, fmt, rate, r, ranges_env)
        assert e.messages().tobytes() == one.messages().tobytes(), r
        st, so = e.stats(), one.stats()
        for k in STAT_KEYS:
            assert st[k] == so[k], (r, k)
        for k in STAT_KEYS[:-1]:
            assert st[k] == est[k], (r, k)
        assert np.array_equal(e.get_icao(), one.get_icao())


def test_ranges_with_overflowing_strips_rescan(ranges_env):
    """A strip that overflows its slots inside one range: the whole run is scanned again with more slots."""
    iq, _ = synth.capture(12 * 65536, 2.4e6, "sc16", seed=77, frames_per_sec=3000, ap_fraction=0.2)
    one = run(iq, "sc16", 2400000, 1, ranges_env)
    e = run(iq, "sc16", 2400000, 4, ranges_env, slots=1)
    assert e.stats()["scan_retries"] >= 1
    assert e.messages().tobytes() == one.messages().tobytes() and len(one.messages()) > 200
    assert np.array_equal(e.get_icao(), one.get_icao())


def test_rerun_accept_behind_a_run_in_ranges(ranges_env):
    """The hand-over rounds of a sharded decode repeat the ordered pass over the whole shard, whatever way it was scanned."""
    iq, _ = synth.capture(10 * 65536, 2.4e6, "sc16", seed=78, frames_per_sec=2500, ap_fraction=0.4)
    ranges_env(3)
    e = da.Engine(2400000, da.INPUT_SC16)
    e.run_host(iq)
    first = e.messages().copy()
    entry = np.zeros(1024, dtype=np.uint32)
    e.set_icao(entry)
    e.rerun_accept()
    assert e.messages().tobytes() == first.tobytes()
    ranges_env()


def test_list_outgrows_the_buffers_mid_run(ranges_env):
    """The message buffers are sized from the ranges decoded so far: traffic that sets in late makes them grow while
    earlier ranges' messages are already in them."""
    quiet, _ = synth.capture(12 * 65536, 2.4e6, "sc16", seed=79, frames_per_sec=40)
    busy, _ = synth.capture(32 * 65536, 2.4e6, "sc16", seed=80, frames_per_sec=6000, ap_fraction=0.1)
    iq = np.concatenate([quiet, busy])
    exp, _ = oracle_run(iq, "sc16")
    assert len(exp) > 2000
    e = run(iq, "sc16", 2400000, 7, ranges_env)
    assert_same_messages(e.messages(), exp)
    e2 = run(iq, "sc16", 2400000, 1, ranges_env)
    assert e.messages().tobytes() == e2.messages().tobytes()
