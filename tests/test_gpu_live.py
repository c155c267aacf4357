"""Live-path shim: blocks handed over the way a receiver's callback does (ref: src/dump1090.c:1243-1282)
must decode to what the reference's loop gives for the same blocks: the oracle's converter + demodulator called block by
block with the overlap of fifo_enqueue (ref: src/fifo.c:195-211), including dropped blocks and the zeroed overlap
behind them."""
import numpy as np
import pytest

import dump1090_amd as da
from dump1090_amd import synth
from oracle import orc
from util import assert_same_messages, oracle_run

pytestmark = pytest.mark.gpu
DEMOD = {2400000: 2400, 2000000: 2000, 8000000: 8000}


def oracle_blocks(blocks, rate, fmt="sc16", filter_dc=False):
    """blocks: list of (raw bytes array, discontinuous).  What rx_callback + fifo + demod_func do with them."""
    o = orc.Oracle(rate, fix_bits=1)
    bps = 2 if fmt == "uc8" else 4
    ov = int((8 + 112 + 16) * 1e-6 * rate)
    cid = orc.convert_select(orc.INPUT_UC8 if fmt == "uc8" else orc.INPUT_SC16, filter_dc, filter_dc)
    state = None
    tail = np.zeros(ov, dtype=np.uint16)
    counter = 0
    for raw, disc in blocks:
        n = raw.nbytes // bps
        mag, power, state = orc.convert(cid, raw, n, sample_rate=float(rate), filter_dc=filter_dc, state=state)
        head = np.zeros(ov, dtype=np.uint16) if disc else tail
        data = np.concatenate([head, mag])
        tail = data[len(data) - ov:].copy()
        o.demod(DEMOD[rate], data, ov, sample_timestamp=int(counter * 12e6 / rate), mean_power=power)
        counter += n
    return o.messages()


def drive(live, blocks):
    for raw in blocks:
        live.flush()                                   # a consumer that keeps up: no drops
        assert live.rx(raw) == 1
    live.flush()
    return live.take(1 << 20)


@pytest.mark.parametrize("rate", [2400000, 2000000, 8000000])
def test_receiver_blocks_match_file_mode(rate):
    """256 KiB blocks, as the RTL-SDR / SDRplay callbacks deliver them: the same messages as --infile on the same bytes."""
    iq, _ = synth.capture(65536 * 6 + 3000, float(rate), "sc16", seed=81, frames_per_sec=2500, ap_fraction=0.3, bitflip1=0.2)
    exp, _ = oracle_run(iq, "sc16", rate=rate)
    raw = iq.view(np.uint8)
    live = da.Live(rate, da.INPUT_SC16, nbuffers=3)
    got = drive(live, [raw[a:a + 262144] for a in range(0, raw.size, 262144)])
    assert len(exp) > (60 if rate == 8000000 else 100)
    assert_same_messages(got, exp)
    st = live.stats()
    assert st["blocks"] == st["blocks_decoded"] == 7 and st["dropped_blocks"] == 0 and st["messages"] == len(exp)
    for b in range(7):                                 # mm.sys_timestamp_msg runs forward inside a block (ms)
        assert np.all(np.diff(got["reserved"][got["block"] == b].view(np.int32).astype(np.int64)) >= 0)
    live.close()


@pytest.mark.parametrize("rate,fmt,dc", [(2400000, "sc16", False), (2400000, "uc8", True), (8000000, "sc16", False)])
def test_arbitrary_block_lengths(rate, fmt, dc):
    """rtl_tcp / SDRconnect hand over whatever the socket read (ref: src/net_io.c:3738-3743): blocks of any length,
    sample counter and overlap carried from one to the next; with the DC filter its state runs through them too."""
    n = 700_000
    iq, _ = synth.capture(n, float(rate), fmt, seed=82, frames_per_sec=2500, ap_fraction=0.3, dc_offset=0.02 - 0.01j if dc else 0.0)
    raw = iq.view(np.uint8)
    bps = 2 if fmt == "uc8" else 4
    rng = np.random.default_rng(5)
    cuts, a = [0], 0
    while a < raw.size:
        a = min(raw.size, a + int(rng.integers(2000, 65536)) * bps)
        cuts.append(a)
    blocks = [raw[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    exp = oracle_blocks([(b, False) for b in blocks], rate, fmt, filter_dc=dc)
    live = da.Live(rate, da.INPUT_UC8 if fmt == "uc8" else da.INPUT_SC16, nbuffers=4, filter_dc=dc, compute_power=dc)
    got = drive(live, blocks)
    assert len(exp) > (40 if rate == 8000000 else 80)
    exp["block"] = got["block"] if len(got) == len(exp) else exp["block"]
    assert_same_messages(got, exp)
    live.close()


def test_dropped_blocks_and_discontinuity():
    """Every buffer busy (a stalled consumer): blocks are dropped and counted, the sample counter does not advance for
    them, and the first block accepted afterwards starts from a zeroed overlap (MAGBUF_DISCONTINUOUS)."""
    rate = 2400000
    iq, _ = synth.capture(65536 * 7, float(rate), "sc16", seed=83, frames_per_sec=3000, ap_fraction=0.2)
    raw = iq.view(np.uint8)
    blocks = [raw[a:a + 262144] for a in range(0, raw.size, 262144)]
    live = da.Live(rate, da.INPUT_SC16, nbuffers=2)
    assert live.rx(blocks[0]) == 1
    live.flush()
    live.hold(True)
    results = [live.rx(b) for b in blocks[1:5]]        # two fit the ring, two are dropped
    assert results == [1, 1, 0, 0]
    with pytest.raises(da.ModesError):                 # on hold the queue is left alone: flush says so instead of waiting for ever
        live.flush()
    live.hold(False)
    live.flush()
    assert live.rx(blocks[5]) == 1                     # discontinuous: behind the drop
    live.flush()
    assert live.rx(blocks[6]) == 1
    live.flush()
    got = live.take(1 << 20)
    st = live.stats()
    assert st["dropped_blocks"] == 2 and st["dropped_samples"] == 2 * 65536 and st["discontinuities"] == 1 and st["blocks"] == 5
    exp = oracle_blocks([(blocks[0], False), (blocks[1], False), (blocks[2], False), (blocks[5], True), (blocks[6], False)], rate)
    assert len(exp) > 100
    assert sorted(set(got["block"].tolist())) == [0, 1, 2, 3, 4]       # `block` counts the accepted blocks
    exp["block"] = got["block"] if len(got) == len(exp) else exp["block"]   # (the oracle's per-call entry does not number them)
    assert_same_messages(got, exp)
    live.close()


def test_oversized_block_is_refused_at_open():
    with pytest.raises(da.ModesError):
        da.Live(2400000, da.INPUT_SC16, max_block_bytes=262144 + 4)
