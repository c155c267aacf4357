"""Sharded decode over the HIP engine: the shards of one recording decoded separately (as N ranks would) with the
halo each shard forms for itself and the ICAO hand-over of dump1090_amd/shard.py must give the single-pass result."""
import numpy as np
import pytest
import torch

import dump1090_amd as da
from dump1090_amd import shard, synth
from util import assert_same_messages, oracle_run

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rate,world", [(2400000, 3), (2000000, 2)])
def test_shards_with_icao_handover_match_single_pass(rate, world):
    iq, _ = synth.capture(65536 * 9 + 4321, float(rate), "sc16", seed=92, frames_per_sec=2500, ap_fraction=0.5, n_icao=6,
                          bitflip1=0.2)
    exp, _ = oracle_run(iq, "sc16", rate=rate)
    d = torch.from_numpy(iq.view(np.int16)).to("cuda:0")
    plan = shard.plan_shards(len(iq) // 2, 65536, world)
    engines = [da.Engine(rate, da.INPUT_SC16, fix_bits=1) for _ in range(world)]
    ov = int((8 + 112 + 16) * 1e-6 * rate)
    halos = [torch.zeros(ov, dtype=torch.int16, device="cuda:0") for _ in range(world)]
    results = [None] * world
    entries = [np.zeros(1024, dtype=np.uint32) for _ in range(world)]
    rounds = 0
    while True:                                       # what decode_sharded does on every rank, all ranks in one process
        exits = []
        for r, (first, count) in enumerate(plan):
            e = engines[r]
            e.set_icao(entries[r])
            halo = None
            if first:
                e.halo_at(d.data_ptr(), d.numel() * 2, first, halos[r].data_ptr())
                halo = halos[r].data_ptr()
            e.run_device(d.data_ptr() + 4 * first, 4 * count, first_sample=first, halo_ptr=halo)
            results[r] = e.messages()
            exits.append(e.get_icao())
        rounds += 1
        new = [entries[0]] + exits[:-1]
        if all(np.array_equal(a, b) for a, b in zip(new, entries)):
            break
        entries = new
    assert rounds <= world + 1
    got = shard.merge(results)
    assert (np.isin(exp["msg_type"], (0, 4, 5, 16, 20, 21))).sum() > 100
    assert_same_messages(got, exp)


def test_tail_halo_chains_shards_sequentially():
    iq, _ = synth.capture(65536 * 5 + 99, 2.4e6, "sc16", seed=93, frames_per_sec=2500, ap_fraction=0.4, n_icao=6)
    exp, _ = oracle_run(iq, "sc16")
    d = torch.from_numpy(iq.view(np.int16)).to("cuda:0")
    e = da.Engine(2400000, da.INPUT_SC16, fix_bits=1)
    parts, halo = [], None
    for first, count in shard.plan_shards(len(iq) // 2, 65536, 3):
        e.run_device(d.data_ptr() + 4 * first, 4 * count, first_sample=first, halo_ptr=halo)
        parts.append(e.messages())
        halo = e.tail_halo()                          # same engine: its ICAO cache carries over by itself
    assert_same_messages(shard.merge(parts), exp)
