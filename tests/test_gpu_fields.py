"""Batched field decode (ref: src/dump1090.c:2580-2675,1901-2249) on the device against the oracle's
sequential restatement: every member of the fixed-layout record, bit for bit (heading / velocity: 1e-12 relative)."""
import numpy as np
import pytest

import dump1090_amd as da
from dump1090_amd import synth
from oracle import orc
from util import oracle_run

pytestmark = pytest.mark.gpu
INT_FIELDS = ("AC_flags", "addr", "category", "addrtype", "reliable", "unit", "capa", "IID", "ME_type", "ME_subtype", "cf", "nuc_p",
              "EW_dir", "NS_dir", "flight_status", "identity", "altitude", "altitude_HAE", "HAE_delta", "raw_latitude",
              "raw_longitude", "EW_velocity", "NS_velocity", "vert_rate", "flight", "ME", "MD")


def assert_same_fields(got, exp):
    assert len(got) == len(exp)
    for k in INT_FIELDS:
        assert np.array_equal(got[k], exp[k]), k
    np.testing.assert_allclose(got["heading"], exp["heading"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(got["velocity"], exp["velocity"], rtol=1e-12, atol=0)


def random_messages(n, seed):
    """Messages of every downlink format the accept rules let through, with random payloads; DF18 with every CF,
    DF17/18 with every ME type and subtype; callsigns both clean and with junk characters; Mode A/C replies."""
    rng = np.random.default_rng(seed)
    m = np.zeros(n, dtype=da.MESSAGE_DTYPE)
    dfs = np.array([0, 4, 5, 11, 16, 17, 18, 20, 21, 24, 32])
    m["msg"] = rng.integers(0, 256, size=(n, 14), dtype=np.uint8)
    df = rng.choice(dfs, size=n, p=[.06, .06, .06, .06, .06, .3, .2, .07, .07, .03, .03])
    m["msg_type"] = df
    first = (np.where(df == 32, 0, df) << 3 | rng.integers(0, 8, n)).astype(np.uint8)
    real = df != 32
    m["msg"][real, 0] = first[real]
    # steer ME types so that every case is frequent, zero fields now and then
    es = (df == 17) | (df == 18)
    me = rng.integers(0, 32, n)
    m["msg"][es, 4] = ((me[es] << 3) | rng.integers(0, 8, es.sum())).astype(np.uint8)
    zero = rng.random(n) < 0.15
    m["msg"][zero, 5:11] = 0
    clean_cs = rng.random(n) < 0.5
    letters = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789    ", dtype=np.uint8)
    for i in np.flatnonzero(clean_cs & ~zero):
        txt = bytes(rng.choice(letters, 8))
        v = 0
        for ch in txt:
            v = (v << 6) | (32 if ch == 32 else ch - 64 if ch >= 65 else ch)      # AIS: A-Z = 1..26, 0-9 = 48..57, space = 32
        m["msg"][i, 5:11] = np.frombuffer(v.to_bytes(6, "big"), dtype=np.uint8)
    bds20 = ((df == 20) | (df == 21)) & (rng.random(n) < 0.5)
    m["msg"][bds20, 4] = 0x20
    m["msg_bits"] = np.where(df == 32, 16, np.where(np.isin(df, (16, 17, 20, 21)), 112, 56))
    m["crc"] = np.where(rng.random(n) < 0.5, 0, rng.integers(0, 1 << 24, n)).astype(np.uint32)
    m["error_bits"] = np.where(m["crc"] == 0, 0, rng.integers(0, 3, n))
    return m


def test_every_format_and_me_type_matches_oracle():
    m = random_messages(60000, 7)
    got = da.decode_fields_host(m)
    exp = orc.decode_fields(m)
    assert_same_fields(got, exp)
    # the cases are all there
    es = np.isin(m["msg_type"], (17, 18))
    assert len(np.unique(exp["ME_type"][es])) == 32 and len(np.unique(exp["cf"][m["msg_type"] == 18])) == 8
    assert (exp["AC_flags"] & 0x20).sum() > 1000 and (exp["flight"] != b"").sum() > 1000           # callsigns
    assert (exp["altitude"] == -9999).sum() > 10 and (exp["AC_flags"] & 1).sum() > 1000              # Gillham and 25 ft altitudes
    assert (exp["heading"] > 0).sum() > 500 and (exp["vert_rate"] != 0).sum() > 200
    assert (exp["addrtype"] == 8).sum() > 100


def test_decoded_recording_fields():
    """End to end: a decoded synthetic recording, fields of its message list (device-resident path) against the oracle
    run on the oracle's own messages."""
    iq, _ = synth.capture(1_200_000, 2.4e6, "sc16", seed=71, frames_per_sec=2500, ap_fraction=0.4, bitflip1=0.2)
    exp_msgs, _ = oracle_run(iq, "sc16")
    e = da.Engine(2400000, da.INPUT_SC16, fix_bits=1)
    e.run_host(iq)
    got = e.decode_fields()
    exp = orc.decode_fields(exp_msgs)
    assert len(got) == len(exp) > 800
    assert_same_fields(got, exp)
    assert np.array_equal(got["addr"], e.messages()["addr"])       # the record's address is the message's
    h = da.Engine(2400000, da.INPUT_SC16, fix_bits=1, host_replay=True)   # messages held on the host: uploaded first
    h.run_host(iq)
    assert h.decode_fields().tobytes() == got.tobytes()


def test_modes1_callsigns_and_altitudes(modes1):
    """The reference's own recording (2.0 MS/s, DC filter on: 259 messages): identification messages give the
    callsigns an operator would see in the reference's interactive view."""
    e = da.Engine(2000000, da.INPUT_UC8, filter_dc=True, compute_power=True)
    e.run_host(modes1)
    f, m = e.decode_fields(), e.messages()
    assert len(f) == 259
    exp = orc.decode_fields(m)
    assert_same_fields(f, exp)
    # README.md:289-305: "*20000f1f684a6c;" is "Altitude : 23375 feet", "*5d4d20237a55a6;" capability 5 (airborne)
    df4 = f[[da.msg_hex(x) == "20000f1f684a6c" for x in m]]
    assert len(df4) >= 1 and (df4["altitude"] == 23375).all() and (df4["unit"] == 1).all() and (df4["flight_status"] == 0).all()
    df11 = f[[da.msg_hex(x) == "5d4d20237a55a6" for x in m]]
    assert len(df11) >= 1 and (df11["capa"] == 5).all() and (df11["addr"] == 0x4D2023).all()
    ident = f[(m["msg_type"] == 17) & (f["ME_type"] >= 1) & (f["ME_type"] <= 4)]
    assert len(ident) > 0 and all(x.strip().isalnum() for x in ident["flight"])
    alt = f[(f["AC_flags"] & 1) != 0]["altitude"]
    assert len(alt) > 50 and alt.min() >= -1000 and alt.max() < 50000
