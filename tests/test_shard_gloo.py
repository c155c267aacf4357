"""The N > 1 path on CPU: two gloo ranks decode the two halves of one recording (shard plan, halo, ICAO hand-over to a
fixed point, merge) and must reproduce the single-stream result.  The per-shard decoder here is the oracle (test
infrastructure); on the GPU box tests/test_gpu_shard.py runs the same driver over the HIP engine."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dump1090_amd import shard, synth      # noqa: E402
from oracle import orc                      # noqa: E402


def test_plan_covers_whole_blocks():
    for n, B, w in ((10 * 65536 + 5, 65536, 4), (3 * 65536, 65536, 8), (1, 65536, 2), (0, 65536, 2), (65536 * 7, 65536, 2)):
        plan = shard.plan_shards(n, B, w)
        assert len(plan) == w and sum(c for _, c in plan) == n
        pos = 0
        for first, cnt in plan:
            assert first == pos and first % B == 0 or cnt == 0
            pos += cnt


def test_next_entries_composes_caches_over_many_shards():
    """shard.next_entries: an address written in shard 0 reaches every later shard's entry in ONE step, and a slot a
    shard overwrote itself keeps that shard's value for the shards behind it."""
    world = 8
    entered = [np.zeros(1024, dtype=np.uint32) for _ in range(world)]
    exits = [e.copy() for e in entered]
    exits[0][5] = 0xAAAAAA
    exits[3][5] = 0xBBBBBB                     # collides in slot 5 (ref: src/dump1090.c:1455-1461: overwritten)
    exits[2][77] = 0xCCCCCC
    nxt = shard.next_entries(entered, exits)
    assert [int(x[5]) for x in nxt] == [0, 0xAAAAAA, 0xAAAAAA, 0xAAAAAA, 0xBBBBBB, 0xBBBBBB, 0xBBBBBB, 0xBBBBBB]
    assert [int(x[77]) for x in nxt] == [0, 0, 0, 0xCCCCCC, 0xCCCCCC, 0xCCCCCC, 0xCCCCCC, 0xCCCCCC]
    d = shard.next_entries([{"icao": e, "d8": b"x%d" % i} for i, e in enumerate(entered)],
                           [{"icao": e, "d8": b"y%d" % i} for i, e in enumerate(exits)])
    assert d[4]["d8"] == b"y3" and int(d[7]["icao"][5]) == 0xBBBBBB


def make_capture(world=2):
    if world >= 8:
        # 64 aircraft over 24 blocks, half of the traffic DF0/4/5/16/20/21 (accepted only when an earlier DF11/17 made
        # the address known, ref: src/dump1090.c:2440,3351-3357): most shards need caches from several shards back
        return synth.capture(65536 * 24 + 4321, 2.4e6, "sc16", seed=93, frames_per_sec=3000, ap_fraction=0.5, n_icao=64, bitflip1=0.2)[0]
    return synth.capture(65536 * 9 + 4321, 2.4e6, "sc16", seed=91, frames_per_sec=2500, ap_fraction=0.5, n_icao=6, bitflip1=0.2)[0]


def oracle_shard_decoder(iq, first, count, rate=2400000):
    ov = int((8 + 112 + 16) * 1e-6 * rate)

    def decode(entry_icao):
        o = orc.Oracle(rate, fix_bits=1)
        o.icao_import(entry_icao)
        tail = None
        if first:                                   # the magnitudes in front of the shard: a pure function of the samples
            c = orc.convert_select(orc.INPUT_SC16, False, False)
            tail, _, _ = orc.convert(c, iq[2 * (first - ov): 2 * first], ov)
        o.stream_start(first, tail)
        if count:
            o.run_stream(iq[2 * first: 2 * (first + count)], orc.INPUT_SC16, rate // 1000)
        return o.messages(), o.icao_export()
    return decode


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    iq = make_capture(world)
    first, count = shard.plan_shards(len(iq) // 2, 65536, world)[rank]

    def all_gather(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out
    msgs, rounds = shard.decode_sharded(oracle_shard_decoder(iq, first, count), rank, world, all_gather)
    gathered = all_gather(msgs)
    if rank == 0:
        q.put((shard.merge(gathered).tobytes(), rounds))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_two_ranks_reproduce_the_single_stream(world):
    iq = make_capture(world)
    o = orc.Oracle(2400000, fix_bits=1)
    o.run_stream(iq, orc.INPUT_SC16, 2400)
    exp = o.messages()
    assert (np.isin(exp["msg_type"], (0, 4, 5, 16, 20, 21))).sum() > 100      # messages that need the cache of earlier shards
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    raw, rounds = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got = np.frombuffer(raw, dtype=orc.MSG_DTYPE)
    assert len(got) == len(exp) and got.tobytes() == exp.tobytes()
    # every rank sees every exit cache and composes them (shard.next_entries): the states settle in two decodes and
    # the exchange that confirms them -- not in one round per shard the oldest address has to travel
    assert 2 <= rounds <= min(world + 1, 3)
    if world >= 8:
        ap = got[np.isin(got["msg_type"], (0, 4, 5, 16, 20, 21))]
        assert len(np.unique(got["addr"])) >= 60 and len(ap) > 400
        # messages whose address was first heard at least two shards earlier: a neighbour-to-neighbour hand-over of the
        # round-one exits would not have let them through in round two
        B = 65536
        per = -(-(len(iq) // 2 // B) // world)
        first_heard = {}
        for m in got:
            if m["msg_type"] in (11, 17, 18) and m["error_bits"] == 0:
                first_heard.setdefault(int(m["addr"]), int(m["block"]) // per)
        far = [m for m in ap if int(m["addr"]) in first_heard and int(m["block"]) // per - first_heard[int(m["addr"])] >= 2]
        assert len(far) > 50
