"""GPU: RAW text of the accepted messages, formatted on the device, against the oracle's messages put
through the reference's two formats: raw_out_send() '*<lower-case hex>;\\n' (ref: src/raw-sbs.c:86-97) and the --raw
stdout line '*<UPPER-CASE HEX>;\\n' (ref: src/dump1090.c:3086-3097).  The README excerpt of modes1.bin is the
reference-held vector (ref: README.md:289-305)."""
import numpy as np
import pytest

import dump1090_amd as da
from dump1090_amd import synth
from oracle import orc
from util import oracle_run

pytestmark = pytest.mark.gpu


def expected(msgs, upper):
    out = []
    for m in msgs:
        h = bytes(m["msg"][: int(m["msg_bits"]) // 8]).hex()
        out.append("*%s;\n" % (h.upper() if upper else h))
    return "".join(out).encode()


@pytest.mark.parametrize("host_replay", [False, True])
def test_raw_text_matches_reference_format(host_replay):
    iq, _ = synth.capture(2_400_000, 2.4e6, "sc16", seed=41, frames_per_sec=3000, bitflip1=0.2, ap_fraction=0.2)
    exp, _ = oracle_run(iq, "sc16")
    e = da.Engine(2400000, da.INPUT_SC16, host_replay=host_replay)
    e.run_host(iq)
    assert len(exp) > 1024                       # more than one scan chunk of messages
    assert e.format_raw(False) == expected(exp, False)
    assert e.format_raw(True) == expected(exp, True)


def test_raw_text_mode_a_and_empty():
    iq, _ = synth.capture(400_000, 2.0e6, "sc16", seed=42, frames_per_sec=2000)
    exp, _ = oracle_run(iq, "sc16", rate=2000000, mode_ac=True)
    e = da.Engine(2000000, da.INPUT_SC16, mode_ac=True)
    e.run_host(iq)
    assert e.format_raw() == expected(exp, False)
    e.run_host(iq[:0])
    assert e.format_raw() == b""


def test_modes1_readme_lines(modes1):
    """README.md:291-304 shows `*5d4d20237a55a6;` and `*20000f1f684a6c;` among the --raw output of modes1.bin."""
    e = da.Engine(2000000, da.INPUT_UC8, filter_dc=True, compute_power=True)
    e.run_host(modes1)
    text = e.format_raw(False)
    assert text.count(b"\n") == 259
    assert b"*5d4d20237a55a6;\n" in text and b"*20000f1f684a6c;\n" in text
