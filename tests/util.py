"""Helpers shared by the GPU parity tests: run the oracle and the HIP engine on the same input
and compare the message streams field by field."""
import numpy as np

from oracle import orc

FMT = {"uc8": orc.INPUT_UC8, "sc16": orc.INPUT_SC16, "sc16q11": orc.INPUT_SC16Q11}


def oracle_run(iq, fmt, rate=2400000, fix_bits=1, phase_enhance=False, filter_dc=False, compute_power=False,
               also_ac=False, mode_ac=False):
    o = orc.Oracle(rate, fix_bits=fix_bits, phase_enhance=phase_enhance, mode_ac=mode_ac,
                   measure_noise=compute_power)
    o.run_stream(iq, FMT[fmt], rate // 1000, filter_dc=filter_dc, compute_power=compute_power, also_ac=also_ac)
    return o.messages(), o.stats()


def assert_same_messages(got, exp, sig_rtol=1e-5):
    """got: dump1090_amd.MESSAGE_DTYPE array, exp: oracle MSG_DTYPE array."""
    assert len(got) == len(exp), "message count %d != %d" % (len(got), len(exp))
    if len(exp) == 0:
        return
    assert np.array_equal(got["timestamp"], exp["timestamp"])
    assert np.array_equal(got["msg_bits"].astype(np.int64), exp["msg_bits"].astype(np.int64))
    assert np.array_equal(got["msg_type"].astype(np.int64), exp["msg_type"].astype(np.int64))
    assert np.array_equal(got["error_bits"].astype(np.int64), exp["error_bits"].astype(np.int64))
    # a Mode A/C message (type 32) leaves mm.CRC as the previous decode_mode_S_message call set it: not a field of it
    ms = exp["msg_type"] != 32
    assert np.array_equal(got["crc"][ms], exp["crc"][ms])
    assert np.array_equal(got["score"].astype(np.int64), exp["score"].astype(np.int64))
    assert np.array_equal(got["phase"].astype(np.int64), exp["phase"].astype(np.int64))
    assert np.array_equal(got["block"].astype(np.int64), exp["block"].astype(np.int64))
    assert np.array_equal(got["j"].astype(np.uint32).view(np.int32), exp["j"].astype(np.int32))   # demod_8000 positions can be < 0
    # DF18 is handled as a 56-bit frame by the reference, so bytes 7..13 that feed the non-ICAO flag are
    # whatever the buffer held: compare the 24-bit address there.
    mask = np.where(exp["msg_type"] == 18, 0xFFFFFF, 0xFFFFFFFF).astype(np.uint32)
    assert np.array_equal(got["addr"] & mask, exp["addr"] & mask)
    nb = exp["msg_bits"] // 8
    for k in (2, 7, 14):
        sel = nb == k
        assert np.array_equal(got["msg"][sel][:, :k], exp["msg"][sel][:, :k])
    np.testing.assert_allclose(got["sig_level"], exp["sig_level"], rtol=sig_rtol, atol=0)
