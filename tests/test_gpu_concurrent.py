"""Engines are independent objects: several of them, each on its own HIP stream and host thread, decode a batch of
recordings at once (BASELINE.json configs[4], what bench.py's batch workload does).  Every recording must come out
exactly as it does alone, and as the CPU oracle gives it."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
import torch

import dump1090_amd as da
from dump1090_amd import synth
from util import assert_same_messages, oracle_run

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode_ac", [False, True])
def test_four_engines_on_four_streams(mode_ac):
    nrec, nlanes = 8, 4
    recs = [synth.capture(400_000 + 4096 * r, 2.4e6, "sc16", seed=50 + r, frames_per_sec=2500, bitflip1=0.1, ap_fraction=0.2)[0]
            for r in range(nrec)]
    dev = [torch.from_numpy(iq.view(np.int16)).to("cuda:0") for iq in recs]
    torch.cuda.synchronize()

    alone = []
    e = da.Engine(2400000, da.INPUT_SC16, fix_bits=1, mode_ac=mode_ac)
    for d in dev:
        e.set_icao(None)
        e.run_device(d.data_ptr(), d.numel() * 2)
        alone.append((e.messages().copy(), e.stats()))

    streams = [torch.cuda.Stream(device="cuda:0") for _ in range(nlanes)]
    engines = [da.Engine(2400000, da.INPUT_SC16, fix_bits=1, mode_ac=mode_ac) for _ in range(nlanes)]
    out = [None] * nrec

    def lane(li):
        for _ in range(3):                       # a few passes, so the lanes really overlap
            for r in range(li, nrec, nlanes):
                engines[li].set_icao(None)
                engines[li].run_device(dev[r].data_ptr(), dev[r].numel() * 2, stream=streams[li].cuda_stream)
                out[r] = (engines[li].messages().copy(), engines[li].stats())

    with ThreadPoolExecutor(max_workers=nlanes) as pool:
        list(pool.map(lane, range(nlanes)))

    for r in range(nrec):
        got, gst = out[r]
        ref, rst = alone[r]
        assert got.tobytes() == ref.tobytes(), "recording %d differs from its solo run" % r
        for k in gst:
            if k != "replay_rounds":
                assert gst[k] == rst[k], (r, k)
    exp, _ = oracle_run(recs[3], "sc16", fix_bits=1, also_ac=mode_ac)
    assert len(exp) > 200
    assert_same_messages(out[3][0], exp)
