"""BASELINE.json's full sizes (4 GiB per GPU): the oracle needs minutes there, so the engine is checked through
properties that do not depend on the size of the recording:

* every message it reports satisfies the Mode S parity it was accepted on (computed here, independently, from the
  message bytes: a squitter's 24-bit remainder is 0 / the interrogator id, an Address/Parity reply's is an address
  the ICAO cache had learnt from a squitter before it);
* the list is in stream order (block, then position) with timestamps that never run backwards;
* every strong, clean frame the generator injected comes back, bytes and time;
* the same recording decoded in one range of blocks, in several ranges, and as two shards chained through the tail
  halo and the ICAO cache gives byte-identical lists (seams at arbitrary places of a long recording);
* decoding twice changes nothing.

The small-size parity tests against the oracle are in test_gpu_parity_*.py; bench.py compares the whole headline
recording with the oracle on the CPU."""
import os

import numpy as np
import pytest
import torch

import dump1090_amd as da
from dump1090_amd import shard, synth, synth_torch

pytestmark = pytest.mark.gpu

GIB = float(os.environ.get("MODES_FULL_SIZE_GIB", "4"))


def crc24_rows(msgs, nbytes):
    """Mode S remainder of the first `nbytes` bytes of every row (the parity field included): 0 for a clean squitter."""
    tab = np.zeros(256, dtype=np.uint32)
    for b in range(256):
        r = b << 16
        for _ in range(8):
            r = ((r << 1) ^ 0xFFF409) if r & 0x800000 else (r << 1)
        tab[b] = r & 0xFFFFFF
    rem = np.zeros(len(msgs), dtype=np.uint32)
    for k in range(nbytes - 3):
        rem = ((rem << 8) & 0xFFFFFF) ^ tab[((rem >> 16) ^ msgs[:, k]) & 0xFF]
    par = (msgs[:, nbytes - 3].astype(np.uint32) << 16) | (msgs[:, nbytes - 2].astype(np.uint32) << 8) | msgs[:, nbytes - 1]
    return rem ^ par


def check_parity_of_every_message(m):
    df = m["msg_type"].astype(np.int64)
    nb = (m["msg_bits"] // 8).astype(np.int64)
    assert set(np.unique(nb)) <= {7, 14}
    syn = np.zeros(len(m), dtype=np.uint32)
    for k in (7, 14):
        sel = nb == k
        syn[sel] = crc24_rows(m["msg"][sel], k)
    # (the list holds the message as repaired: ref: src/dump1090.c:3315-3347 fixes the bits before the message is used)
    es = np.isin(df, (17, 18))
    assert es.sum() > 1000 and np.all(syn[es] == 0)
    acq = df == 11
    assert acq.sum() > 1000 and np.all((syn[acq] & 0xFFFF80) == 0)
    # Address/Parity: the remainder IS the address, and the reference only accepts it when the cache knows it
    # (ref: src/dump1090.c:3355-3364): it must have been announced by a squitter earlier in the list
    ap = np.isin(df, (0, 4, 5, 16, 20, 21, 24, 25, 26, 27, 28, 29, 30, 31))    # (24..31: Comm-D, ref: src/dump1090.c:3300-3308)
    if ap.any():
        first_seen = {}
        sq = np.flatnonzero(es | acq)
        addr_sq = m["addr"][sq] & 0xFFFFFF
        for i, a in zip(sq[::-1], addr_sq[::-1]):
            first_seen[int(a)] = int(i)
        for i in np.flatnonzero(ap)[:: max(1, int(ap.sum()) // 5000)]:
            assert int(syn[i]) in first_seen and first_seen[int(syn[i])] < i
    assert np.all(es | acq | ap)


def check_stream_order(m, strict=True):
    blk = m["block"].astype(np.int64)
    assert np.all(np.diff(blk) >= 0)
    ts = m["timestamp"].astype(np.int64)
    same = np.diff(blk) == 0
    # inside a block positions only grow (demod_8000's match windows overlap by half: the message picked from a window may
    # start in front of the one picked from the window before, ref: src/demod-8000.c:231-420)
    assert not strict or np.all(np.diff(ts)[same] > 0)
    assert np.all(np.diff(ts) > -int(12e6 * 400e-6))  # across a block seam at most the overlap goes back


def check_injected_frames_come_back(m, info, rate, nsamples):
    """Strong frames with nothing of another frame within a frame's length: all but a handful must be in the list."""
    pos, which, amp = info["pos"], info["which"], info["amp"]
    L = int(130e-6 * rate)
    clear = np.ones(len(pos), dtype=bool)
    gap = np.diff(pos)
    clear[1:] &= gap > 2 * L
    clear[:-1] &= gap > 2 * L
    clear &= (amp > 10 ** (-14 / 20.0)) & (pos > 4 * L) & (pos < nsamples - 4 * L)
    idx = np.flatnonzero(clear)
    idx = idx[:: max(1, len(idx) // 20000)]
    assert len(idx) > 5000
    ts = m["timestamp"].astype(np.int64)
    order = np.argsort(ts, kind="stable")
    ts_sorted = ts[order]
    found = 0
    for i in idx:
        want = info["msgs"][which[i]]
        t = int(pos[i] * 12e6 / rate)
        a, b = np.searchsorted(ts_sorted, t - 600), np.searchsorted(ts_sorted, t + 2400)
        for k in order[a:b]:
            if bytes(m["msg"][k][: len(want)]) == want:
                found += 1
                break
    print("injected strong frames found: %d of %d" % (found, len(idx)))
    # (demod_2000 slices two samples per bit without a phase search: frames that start half a sample off are lost to it)
    assert found >= (0.85 if rate == 2000000 else 0.98) * len(idx), (found, len(idx))


def decode(e, d, nbytes, ranges=None):
    old = os.environ.get("MODES_RANGES")
    if ranges is not None:
        os.environ["MODES_RANGES"] = str(ranges)
    try:
        e.set_icao(None)
        e.run_device(d.data_ptr(), nbytes)
        return e.messages().copy()
    finally:
        if ranges is not None:
            if old is None:
                del os.environ["MODES_RANGES"]
            else:
                os.environ["MODES_RANGES"] = old


@pytest.mark.parametrize("name,rate,fmt,kw", [("uc8_2400_lut", 2400000, "uc8lut", {}), ("sc16_2400", 2400000, "sc16", {}),
                                              ("sc16_2000", 2000000, "sc16", {"fix_bits": 2}), ("sc16_8000", 8000000, "sc16", {})])
def test_full_size_properties(name, rate, fmt, kw):
    bps = 2 if fmt == "uc8lut" else 4
    n = int(GIB * (1 << 30)) // bps
    d, info = synth_torch.capture_gpu(n, float(rate), fmt, seed=1, device="cuda:0", frames_per_sec=3000.0 if rate != 8000000 else 1000.0)
    torch.cuda.synchronize()
    nbytes = d.numel() * d.element_size()
    e = da.Engine(rate, da.INPUT_UC8 if fmt == "uc8lut" else da.INPUT_SC16, **kw)
    one = decode(e, d, nbytes, ranges=1)
    assert len(one) > 50000
    check_parity_of_every_message(one)
    check_stream_order(one, strict=rate != 8000000)
    check_injected_frames_come_back(one, info, rate, n)
    # the same list whatever way the shard is cut into ranges, and again on a second decode
    for r in (None, 5):
        again = decode(e, d, nbytes, ranges=r)
        assert again.tobytes() == one.tobytes(), r
    if rate == 8000000:
        return
    # two shards chained like consecutive bin_read calls on one stream: tail halo and ICAO cache carried over
    B = 262144 // bps
    plan = shard.plan_shards(n, B, 2)
    e2 = da.Engine(rate, da.INPUT_UC8 if fmt == "uc8lut" else da.INPUT_SC16, **kw)
    parts, halo = [], None
    for first, count in plan:
        e2.run_device(d.data_ptr() + bps * first, bps * count, first_sample=first, halo_ptr=halo)
        parts.append(e2.messages().copy())
        halo = e2.tail_halo()
    assert shard.merge(parts).tobytes() == one.tobytes()


def test_full_size_dc_stream_in_two_calls_equals_one():
    """The DC-block converter over 4 GiB of uc8: the chain's chunks, warm-ups and repairs at full scale.  Two calls on
    consecutive halves of the stream (filter state, overlap and ICAO cache carried by the engine) give the list and
    the final filter state of one call over the whole."""
    n = int(GIB * (1 << 30)) // 2
    d, info = synth_torch.capture_gpu(n, 2.4e6, "uc8", seed=1, device="cuda:0", frames_per_sec=3000.0, dc_offset=(0.006, -0.004))
    torch.cuda.synchronize()
    e = da.Engine(2400000, da.INPUT_UC8, filter_dc=True)
    e.set_dc_state(0.0, 0.0)
    e.run_device(d.data_ptr(), 2 * n)
    whole, z_whole = e.messages().copy(), e.get_dc_state()
    assert len(whole) > 50000 and e.dc_info()["rerun_chunks"] <= 2
    check_parity_of_every_message(whole)
    check_injected_frames_come_back(whole, info, 2.4e6, n)
    e2 = da.Engine(2400000, da.INPUT_UC8, filter_dc=True)
    e2.set_dc_state(0.0, 0.0)
    half = (n // 2) // 131072 * 131072
    e2.run_device(d.data_ptr(), 2 * half)
    a = e2.messages().copy()
    e2.run_device(d.data_ptr() + 2 * half, 2 * (n - half), first_sample=half, halo_ptr=e2.tail_halo())
    b = e2.messages().copy()
    assert shard.merge([a, b]).tobytes() == whole.tobytes()
    assert e2.get_dc_state() == z_whole


@pytest.mark.parametrize("name,rate,fmt", [("uc8_2400_lut", 2400000, "uc8lut"), ("sc16_8000", 8000000, "sc16")])
def test_last_shard_of_eight_ranks(name, rate, fmt):
    """What rank 7 of `bench.py --gpus 8` holds: the last eighth of ONE recording of 8 x 4 GiB, generated on its own and
    decoded at its place in the stream, 15 G (uc8) / 7.5 G (sc16) samples in.  The strong injected frames of that
    eighth come back with their bytes at their stream times, every message passes its parity, the list is in order."""
    bps = 2 if fmt == "uc8lut" else 4
    world, rank = 8, 7
    n = int(GIB * (1 << 30)) // bps
    B = 262144 // bps
    first, count = shard.plan_shards(world * n, B, world)[rank]
    chunk = 1 << 24
    g0 = first // chunk * chunk
    d, info = synth_torch.capture_gpu(world * n, float(rate), fmt, seed=1, device="cuda:0", chunk=chunk,
                                      sample_range=(g0, -(-(first + count) // chunk) * chunk),
                                      frames_per_sec=3000.0 if rate != 8000000 else 1000.0)
    torch.cuda.synchronize()
    sel = (info["pos"] >= first) & (info["pos"] < first + count)
    info = dict(info, pos=info["pos"][sel], which=info["which"][sel], amp=info["amp"][sel])
    e = da.Engine(rate, da.INPUT_UC8 if fmt == "uc8lut" else da.INPUT_SC16)
    if rate == 8000000:
        # the state a shard behind the first is entered with: here that of a stream that was silent so far
        quiet = torch.zeros(2 * 65536, dtype=torch.int16, device="cuda:0")
        e.run_device(quiet.data_ptr(), quiet.numel() * 2)
        e.set_d8_state(e.get_d8_state(want_acc=True))
    e.run_device(d.data_ptr() + bps * (first - g0), bps * count, first_sample=first)
    m = e.messages().copy()
    assert len(m) > 5000
    assert int(m["timestamp"].min()) >= int(first * 12e6 / rate) - 2400
    check_parity_of_every_message(m)
    check_stream_order(m, strict=rate != 8000000)
    # (positions count from the start of the whole recording; the first frames of the shard, whose neighbours in front
    # of it are not in the list, are left out)
    keep = info["pos"] > first + 4000
    check_injected_frames_come_back(m, dict(info, pos=info["pos"][keep], which=info["which"][keep], amp=info["amp"][keep]), rate, first + count)
