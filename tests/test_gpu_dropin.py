"""GPU parity of the reference's own entry points (Part 1 of include/modes_hip.h): the six converters behind
convert_init(), and demod_2400() fed one mag_buf at a time -- once with the library's built-in accept rules,
once with the callbacks bound to a stand-in for dump1090.c (the oracle's score/decode)."""
import ctypes as C

import numpy as np
import pytest

import dump1090_amd as da
from dump1090_amd import synth
from oracle import orc
from util import assert_same_messages

pytestmark = pytest.mark.gpu


def conv_pair(fmt_id, iq, nsamples, dc, pw, rate=2400000.0, chunks=(1.0,)):
    """Run the same sequence of converter calls through convert_init()'s function and through the oracle."""
    c = da.Converter(fmt_id, rate, dc, pw)
    cid = orc.convert_select(fmt_id, dc, pw)
    st = None
    bps = 1 if cid < 2 else (2 if fmt_id == da.INPUT_UC8 else 4)
    pos = 0
    raw = iq.view(np.uint8)
    for frac in chunks:
        n = int(nsamples * frac)
        part = raw[pos * (2 if fmt_id == da.INPUT_UC8 else 4):]
        got, gp = c(part, n)
        exp, ep, st = orc.convert(cid, part, n, rate, dc, st)
        assert np.array_equal(got, exp), (c.name, int((got != exp).sum()))
        if cid in (2, 4, 5):
            assert gp == ep, (c.name, gp, ep)            # float running sum, same order of additions
            assert (c.state.contents.z1_I, c.state.contents.z1_Q) == (st.z1_I, st.z1_Q)
        else:
            assert gp == ep
        pos += n
    c.close()


def test_uc8_converters_all_byte_pairs(modes1):
    pairs = np.stack(np.meshgrid(np.arange(256), np.arange(256), indexing="ij"), -1).reshape(-1, 2).astype(np.uint8)
    for dc, pw in ((False, False), (False, True), (True, True)):
        conv_pair(da.INPUT_UC8, pairs.reshape(-1), 65536, dc, pw)
    conv_pair(da.INPUT_UC8, modes1, len(modes1) // 2, True, True, rate=2000000.0, chunks=(0.3, 0.5, 0.2))
    conv_pair(da.INPUT_UC8, modes1, 131072, False, False)


def test_sc16_converters():
    rng = np.random.default_rng(1)
    iq = rng.integers(-32768, 32768, size=2 * 200000, dtype=np.int16)
    iq[:8] = [-32768, -32768, 32767, 32767, -32768, 0, 0, -32768]
    for dc, pw in ((False, False), (False, True), (True, False), (True, True)):
        conv_pair(da.INPUT_SC16, iq, 200000, dc, pw, chunks=(0.5, 0.5))
    small = (rng.standard_normal(2 * 131072) * 300).astype(np.int16)
    conv_pair(da.INPUT_SC16Q11, small, 131072, True, True, chunks=(0.25, 0.75))
    conv_pair(da.INPUT_SC16Q11, small, 131072, False, False)
    conv_pair(da.INPUT_SC16, iq, 0, False, False)          # empty call


def blocks_of(iq, fmt, rate=2400000):
    """The mag_bufs bin_read would hand to the demodulator (oracle converter, ref: src/infile.c:139-185)."""
    bps = 2 if fmt == "uc8" else 4
    ov = int((8 + 112 + 16) * 1e-6 * rate)
    B = 262144 // bps
    cid = orc.convert_select(orc.INPUT_UC8 if fmt == "uc8" else orc.INPUT_SC16, False, False)
    raw = iq.view(np.uint8)
    tail = np.zeros(ov, dtype=np.uint16)
    out, counter, off, eof = [], 0, 0, False
    while not eof:
        chunk = raw[off: off + 262144]
        eof = len(chunk) < 262144
        n = len(chunk) // bps
        mag, _, _ = orc.convert(cid, chunk, n) if n else (np.zeros(0, np.uint16), 0, None)
        data = np.concatenate([tail, mag])
        tail = data[len(data) - ov:].copy()
        out.append((data, ov, int(counter * 12e6 / rate)))
        counter += n
        off += 262144
    return out


def test_demod_2400_per_block_builtin_rules():
    iq, _ = synth.capture(65536 * 3 + 5000, 2.4e6, "sc16", seed=31, frames_per_sec=2500, bitflip1=0.2, ap_fraction=0.3)
    o = orc.Oracle(2400000, fix_bits=1)
    o.run_stream(iq, orc.INPUT_SC16, 2400)
    exp = o.messages()
    L = da.lib()
    L.modes_hip_bind(None)
    L.crc_init(1)
    da.take_messages()
    base = None
    for data, ov, ts in blocks_of(iq, "sc16"):
        mb = da.make_mag_buf(data, ov, sample_timestamp=ts)
        L.demod_2400(C.byref(mb))
    got = da.take_messages()
    if len(got):
        got["block"] = (got["block"].astype(np.int64) - int(got["block"][0]) + int(exp["block"][0])).astype(np.uint32)
    assert len(exp) > 100
    assert_same_messages(got, exp)


def test_demod_2400_drop_in_rescans_when_a_strip_runs_out_of_slots(monkeypatch):
    """Two candidate slots per strip overflow on any real signal: the drop-in scans again with more until everything
    fits (as the bulk engine does); the block is not lost and the result is the oracle's."""
    monkeypatch.setenv("MODES_STRIP_SLOTS", "2")
    test_demod_2400_per_block_builtin_rules()
    test_demod_2000_per_block_builtin_rules()


def test_demod_2400_per_block_with_host_callbacks():
    """dump1090.c's side of the boundary played by the oracle: score / decode / user_message callbacks."""
    iq, _ = synth.capture(65536 * 2 + 777, 2.4e6, "sc16", seed=32, frames_per_sec=2500, bitflip1=0.3, bitflip2=0.1,
                          ap_fraction=0.3)
    o = orc.Oracle(2400000, fix_bits=2)
    o.run_stream(iq, orc.INPUT_SC16, 2400)
    exp = o.messages()

    host = orc.Oracle(2400000, fix_bits=2)          # the "application" state: ICAO cache, CRC tables
    OL = orc.lib()
    seen = []
    hb = da.HostBinding()
    hb.mm_size = orc.MSG_DTYPE.itemsize
    hb.off_timestamp_msg, hb.off_sys_timestamp_msg = 0, 8
    hb.off_sig_level = orc.MSG_DTYPE.fields["sig_level"][1]
    hb.off_score = orc.MSG_DTYPE.fields["pad2"][1]          # decode wipes mm.score in the reference; parked on padding
    hb.off_error_bits = orc.MSG_DTYPE.fields["error_bits"][1]
    hb.off_AC_flags = orc.MSG_DTYPE.fields["pad"][1] - 2 + 2  # two spare bytes are not enough for a u32: use pad2 too
    hb.off_AC_flags = orc.MSG_DTYPE.fields["pad2"][1]
    hb.off_addr = orc.MSG_DTYPE.fields["addr"][1]
    hb.modeS_message_score = da.SCORE_CB(lambda msg, bits: OL.orc_message_score(host.h, msg, bits))
    hb.decode_mode_S_message = da.DECODE_CB(lambda mm, msg: OL.orc_decode_message(host.h, msg, mm))
    hb.modeS_user_message = da.USER_CB(lambda mm: seen.append(bytes((C.c_char * hb.mm_size).from_address(mm))))
    hb.modeS_message_len_by_type = da.LEN_CB(lambda t: 112 if t in (16, 17, 19, 20, 21) else 56)
    hb.receiveclock_ms_elapsed = da.CLOCK_CB(lambda a, b: (b - a) // 12000)
    hb.fix_bits = 2
    L = da.lib()
    L.crc_exit()
    assert L.modes_hip_bind(C.byref(hb)) == 0
    try:
        for data, ov, ts in blocks_of(iq, "sc16"):
            mb = da.make_mag_buf(data, ov, sample_timestamp=ts)
            L.demod_2400(C.byref(mb))
    finally:
        L.modes_hip_bind(None)
    got = np.frombuffer(b"".join(seen), dtype=orc.MSG_DTYPE)
    assert len(exp) > 40 and len(got) == len(exp)
    for f in ("timestamp", "sys_timestamp", "msg_bits", "msg_type", "error_bits", "addr", "crc", "sig_level"):
        assert np.array_equal(got[f], exp[f]), f
    for g, x in zip(got, exp):
        n = int(x["msg_bits"]) // 8
        assert bytes(g["msg"][:n]) == bytes(x["msg"][:n])


def test_demod_2000_per_block_builtin_rules():
    iq, _ = synth.capture(65536 * 3 + 5000, 2.0e6, "sc16", seed=33, frames_per_sec=4000, bitflip1=0.2, ap_fraction=0.3)
    o = orc.Oracle(2000000, fix_bits=1)
    o.run_stream(iq, orc.INPUT_SC16, 2000)
    exp = o.messages()
    L = da.lib()
    L.modes_hip_bind(None)
    L.crc_init(1)
    da.take_messages()
    for data, ov, ts in blocks_of(iq, "sc16", rate=2000000):
        mb = da.make_mag_buf(data, ov, sample_timestamp=ts)
        L.demod_2000(C.byref(mb))
    got = da.take_messages()
    if len(got):
        got["block"] = (got["block"].astype(np.int64) - int(got["block"][0]) + int(exp["block"][0])).astype(np.uint32)
    assert len(exp) > 200
    assert_same_messages(got, exp)


def test_demod_2000_with_host_callbacks_and_phase_enhance():
    """dump1090.c's side played by the oracle; Modes.phase_enhance handed over by the binding."""
    iq, _ = synth.capture(65536 * 2 + 777, 2.0e6, "sc16", seed=34, frames_per_sec=4000, bitflip1=0.3, bitflip2=0.1,
                          ap_fraction=0.3, overlap_fraction=0.3)
    o = orc.Oracle(2000000, fix_bits=2, phase_enhance=True)
    o.run_stream(iq, orc.INPUT_SC16, 2000)
    exp = o.messages()

    host = orc.Oracle(2000000, fix_bits=2)
    OL = orc.lib()
    seen = []
    hb = da.HostBinding()
    hb.mm_size = orc.MSG_DTYPE.itemsize
    hb.off_timestamp_msg, hb.off_sys_timestamp_msg = 0, 8
    hb.off_sig_level = orc.MSG_DTYPE.fields["sig_level"][1]
    hb.off_score = orc.MSG_DTYPE.fields["pad2"][1]
    hb.off_error_bits = orc.MSG_DTYPE.fields["error_bits"][1]
    hb.off_AC_flags = orc.MSG_DTYPE.fields["pad2"][1]
    hb.off_addr = orc.MSG_DTYPE.fields["addr"][1]
    hb.decode_mode_S_message = da.DECODE_CB(lambda mm, msg: OL.orc_decode_message(host.h, msg, mm))
    hb.modeS_user_message = da.USER_CB(lambda mm: seen.append(bytes((C.c_char * hb.mm_size).from_address(mm))))
    hb.receiveclock_ms_elapsed = da.CLOCK_CB(lambda a, b: (b - a) // 12000)
    hb.fix_bits = 2
    hb.phase_enhance = 1
    hb.mode_AC = 0
    L = da.lib()
    L.crc_exit()
    assert L.modes_hip_bind(C.byref(hb)) == 0
    try:
        for data, ov, ts in blocks_of(iq, "sc16", rate=2000000):
            mb = da.make_mag_buf(data, ov, sample_timestamp=ts)
            L.demod_2000(C.byref(mb))
    finally:
        L.modes_hip_bind(None)
    got = np.frombuffer(b"".join(seen), dtype=orc.MSG_DTYPE)
    assert len(exp) > 50 and len(got) == len(exp)
    for f in ("timestamp", "sys_timestamp", "msg_bits", "msg_type", "error_bits", "addr", "crc"):
        assert np.array_equal(got[f], exp[f]), f
    for g, x in zip(got, exp):
        n = int(x["msg_bits"]) // 8
        assert bytes(g["msg"][:n]) == bytes(x["msg"][:n])


def _elapsed(a, b):
    """receiveclock_ms_elapsed (ref: src/misc.c:1259-1262): signed difference, C division."""
    d = (b - a) & (2 ** 64 - 1)
    d = d - 2 ** 64 if d >= 2 ** 63 else d
    return (abs(d) // 12000) * (1 if d >= 0 else -1)


def _run_blocks_8000(L, iq):
    assert L.demod_8000_alloc() != 0
    for data, ov, ts in blocks_of(iq, "sc16", rate=8000000):
        mb = da.make_mag_buf(data, ov, sample_timestamp=ts)
        L.demod_8000(C.byref(mb))
    L.demod_8000_free()


@pytest.mark.parametrize("quiet", [False, True])
def test_demod_8000_per_block_builtin_rules(quiet):
    """demod_8000 keeps its sliding sums, running average, open window and 1928 differences between calls."""
    kw = dict(noise_sigma=0.0003, min_amp_db=-50, max_amp_db=-30) if quiet else {}
    iq, _ = synth.capture(65536 * 6 + 4321, 8.0e6, "sc16", seed=35, frames_per_sec=2500, bitflip1=0.2, ap_fraction=0.3, **kw)
    o = orc.Oracle(8000000, fix_bits=1)
    o.run_stream(iq, orc.INPUT_SC16, 8000)
    exp = o.messages()
    L = da.lib()
    L.modes_hip_bind(None)
    L.crc_init(1)
    da.take_messages()
    _run_blocks_8000(L, iq)
    got = da.take_messages()
    if len(got):
        got["block"] = (got["block"].astype(np.int64) - int(got["block"][0]) + int(exp["block"][0])).astype(np.uint32)
    assert len(exp) > 40
    assert_same_messages(got, exp)


def test_demod_8000_with_host_callbacks():
    iq, _ = synth.capture(65536 * 4 + 999, 8.0e6, "sc16", seed=36, frames_per_sec=2500, bitflip1=0.3, bitflip2=0.1,
                          ap_fraction=0.3)
    o = orc.Oracle(8000000, fix_bits=2)
    o.run_stream(iq, orc.INPUT_SC16, 8000)
    exp = o.messages()
    host = orc.Oracle(8000000, fix_bits=2)
    OL = orc.lib()
    seen = []
    hb = da.HostBinding()
    hb.mm_size = orc.MSG_DTYPE.itemsize
    hb.off_timestamp_msg, hb.off_sys_timestamp_msg = 0, 8
    hb.off_sig_level = orc.MSG_DTYPE.fields["sig_level"][1]
    hb.off_score = orc.MSG_DTYPE.fields["pad2"][1]
    hb.off_error_bits = orc.MSG_DTYPE.fields["error_bits"][1]
    hb.off_AC_flags = orc.MSG_DTYPE.fields["pad2"][1]
    hb.off_addr = orc.MSG_DTYPE.fields["addr"][1]
    hb.modeS_message_score = da.SCORE_CB(lambda msg, bits: OL.orc_message_score(host.h, msg, bits))
    hb.decode_mode_S_message = da.DECODE_CB(lambda mm, msg: OL.orc_decode_message(host.h, msg, mm))
    hb.modeS_user_message = da.USER_CB(lambda mm: seen.append(bytes((C.c_char * hb.mm_size).from_address(mm))))
    hb.receiveclock_ms_elapsed = da.CLOCK_CB(_elapsed)
    hb.fix_bits = 2
    L = da.lib()
    L.crc_exit()
    assert L.modes_hip_bind(C.byref(hb)) == 0
    try:
        _run_blocks_8000(L, iq)
    finally:
        L.modes_hip_bind(None)
    got = np.frombuffer(b"".join(seen), dtype=orc.MSG_DTYPE)
    assert len(exp) > 30 and len(got) == len(exp)
    for f in ("timestamp", "sys_timestamp", "msg_bits", "msg_type", "error_bits", "addr", "crc", "sig_level"):
        assert np.array_equal(got[f], exp[f]), f
    for g, x in zip(got, exp):
        n = int(x["msg_bits"]) // 8
        assert bytes(g["msg"][:n]) == bytes(x["msg"][:n])
