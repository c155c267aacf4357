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


def _capture_8000(nblocks=7, extra=321, quiet=False, seed=94):
    """sc16 at 8 MS/s: regular traffic plus one frame laid around every block edge, so that shard seams fall inside
    open match windows.  `quiet`: noise of a dozen LSB, where demod_8000's trigger depends on its running average."""
    rng = np.random.default_rng(seed)
    B = 65536
    n = B * nblocks + extra
    x, _ = synth.capture_complex(n, 8e6, seed + 1, frames_per_sec=1200, ap_fraction=0.3, n_icao=6,
                                 noise_sigma=0.0004 if quiet else 0.01, min_amp_db=-40 if quiet else -32,
                                 max_amp_db=-25 if quiet else -4)
    offs = (-1500, -1200, -960, -700, -400, -130, -20, 0, 90)
    for b in range(1, nblocks):
        p = b * B + offs[(b * 4) % len(offs)]
        msg = synth.make_df17(0x4840D6 + b, bytes(rng.integers(0, 256, 7, dtype=np.uint8)))
        env = synth.envelope(msg, 8e6, rng.random())
        x[p - 200:p + len(env) + 200] *= 0.05                      # clear the regular traffic around it
        x[p:p + len(env)] += (0.03 if quiet else 0.3) * np.exp(2j * np.pi * rng.random()) * env
    return synth.quantise(x, "sc16")


@pytest.mark.parametrize("quiet", [False, True])
def test_8000_shards_chain_sequentially(quiet):
    """demod_8000 shards of one recording, each given the state the shard before left (acc, open window, trigger
    history, 1928 differences) + its tail halo: the three shards together = the oracle's single pass.  `quiet`: input
    so weak that triggers depend on the running average, which then has to travel too."""
    iq = _capture_8000(quiet=quiet)
    exp, _ = oracle_run(iq, "sc16", rate=8000000)
    assert len(exp) > 40
    d = torch.from_numpy(iq).to("cuda:0")
    single = da.Engine(8000000, da.INPUT_SC16, fix_bits=1)
    single.run_device(d.data_ptr(), iq.nbytes)
    assert_same_messages(single.messages(), exp)
    e = da.Engine(8000000, da.INPUT_SC16, fix_bits=1)
    parts, halo, state = [], None, None
    seam_windows = 0
    for first, count in shard.plan_shards(len(iq) // 2, 65536, 3):
        if state is not None:
            e.set_d8_state(state)
        e.run_device(d.data_ptr() + 4 * first, 4 * count, first_sample=first, halo_ptr=halo)
        parts.append(e.messages())
        halo = e.tail_halo()
        state = e.get_d8_state(want_acc=True)
        seam_windows += state.pending < 0
        if quiet:
            assert state.amb_steps > 0
    assert_same_messages(shard.merge(parts), exp)
    assert seam_windows >= 1 or quiet                  # a shard ended inside an open match window

    # a shard in the middle of the stream without the state of the one before is refused, not guessed
    first, count = shard.plan_shards(len(iq) // 2, 65536, 3)[1]
    with pytest.raises(da.ModesError):
        e.run_device(d.data_ptr() + 4 * first, 4 * count, first_sample=first)


def _run_ranks(world, make_shard):
    """`world` threads, one EngineShard each, through shard.decode_engine_sharded (what bench.py runs rank per GPU)."""
    import threading
    grp = shard.ThreadGroup(world)
    out = [None] * world
    err = []

    def work(r):
        try:
            sh = make_shard(r)
            msgs, rounds = shard.decode_engine_sharded(sh, r, world, grp.all_gather(r), max_rounds=2 * world + 2)
            out[r] = (msgs, rounds, sh)
        except BaseException as ex:                   # noqa: BLE001
            err.append(ex)
            grp.barrier.abort()
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if err:
        raise err[0]
    return out


@pytest.mark.parametrize("rate,fmt,world", [(2400000, "sc16", 3), (2000000, "sc16", 2), (2400000, "uc8lut", 2), (8000000, "sc16", 3),
                                            (8000000, "quiet", 3), (8000000, "noisy", 4), (2400000, "sc16", 8), (8000000, "sc16", 8)])
def test_engine_shards_reach_the_single_pass_result(rate, fmt, world):
    """The product path of a sharded decode: every rank holds only its shard and one block in front of it, forms its
    own halo, and the entry states (ICAO cache; at 8 MS/s also demod_8000's carried state) are handed over until they
    settle.  From the second round on no scan kernel runs at 2.0 / 2.4 MS/s."""
    if rate == 8000000 and fmt == "noisy":
        # receiver noise everywhere: demod_8000's trigger never pauses, windows follow each other at 113 steps and where
        # they fall depends on where the chain entered the shard
        iq, _ = synth.capture(65536 * 11 + 777, 8e6, "sc16", seed=97, frames_per_sec=1500, ap_fraction=0.3, n_icao=6)
        efmt, bps, ofmt = da.INPUT_SC16, 4, "sc16"
    elif rate == 8000000:
        iq = _capture_8000(nblocks=9 if world < 8 else 25, quiet=fmt == "quiet")
        efmt, bps, ofmt = da.INPUT_SC16, 4, "sc16"
    elif fmt == "uc8lut":
        iq, _ = synth.capture_lut_image(131072 * 7 + 999, 2.4e6, seed=96, frames_per_sec=3000, min_amp_db=-22, ap_fraction=0.5, n_icao=6)
        efmt, bps, ofmt = da.INPUT_UC8, 2, "uc8"
    else:
        # (eight shards: the node's worth of ranks, three blocks each, and an aircraft population that does not fit
        # the first shard -- caches keep changing down the chain of shards)
        iq, _ = synth.capture(65536 * (9 if world < 8 else 25) + 4321, float(rate), "sc16", seed=92, frames_per_sec=2500, ap_fraction=0.5,
                              n_icao=6 if world < 8 else 64, bitflip1=0.2)
        efmt, bps, ofmt = da.INPUT_SC16, 4, "sc16"
    exp, _ = oracle_run(iq, ofmt, rate=rate)
    B = 262144 // bps
    raw = np.ascontiguousarray(iq).view(np.uint8)
    plan = shard.plan_shards(raw.size // bps, B, world)
    held = []

    def make_shard(r):
        first, count = plan[r]
        lead = B if first else 0
        buf = torch.from_numpy(raw[(first - lead) * bps:(first + count) * bps].copy()).to("cuda:0")     # all this rank ever sees
        held.append(buf)
        return shard.EngineShard(da.Engine(rate, efmt, fix_bits=1), buf.data_ptr(), buf.numel(), lead, first, count, rate, bps)
    res = _run_ranks(world, make_shard)
    got = shard.merge([m for m, _, _ in res])
    assert_same_messages(got, exp)
    assert len(exp) > 40
    rounds = res[0][1]
    assert 2 <= rounds <= (min(world + 1, 4) if fmt != "quiet" else 2 * world + 2)
    if world >= 8 and rate != 8000000:
        assert len(np.unique(exp["addr"][np.isin(exp["msg_type"], (11, 17))])) >= 50
    assert len(exp) > (100 if fmt == "noisy" else 40)
    for r, (_, _, sh) in enumerate(res):
        c = sh.e.counters()
        if rate != 8000000:
            assert c["scan_launches"] == 1 and c["accept_passes"] == sh.rounds    # later rounds: the ordered pass only
        elif fmt == "noisy":
            # the window chain was settled before anything was decoded (modes_engine_d8_chain_map): what rank r - 1
            # left behind its shard is what rank r was entered with -- differences, trigger history, open window
            assert sh.entry_states is not None and rounds <= 3
            if r:
                left = da.D8State.from_buffer_copy(res[r - 1][2]._d8_exit(None))
                had = da.D8State.from_buffer_copy(sh.entry_states[r]["d8"])
                assert (left.pending, left.trig_tail) == (had.pending, had.trig_tail)
                assert bytes(left.dprev) == bytes(had.dprev)
        elif fmt == "quiet":
            assert sh.entry_states is None        # near-silent input: the trigger needs the running average, rounds find the states


@pytest.mark.parametrize("kw", [{"phase_enhance": True}, {"mode_ac": True}, {"host_replay": True}])
def test_shards_of_engines_that_cannot_repeat_the_accept_pass_alone(kw):
    """Engines whose ordered pass is on the host (phase_enhance, host_replay) or that run a second demodulator per block
    (mode_ac) refuse modes_engine_rerun_accept: their shards are decoded again in the hand-over rounds -- on every rank,
    so none is left waiting in the collective -- and still reach the single-pass result."""
    rate, world = 2400000, 3
    iq, _ = synth.capture(65536 * 7 + 4321, float(rate), "sc16", seed=98, frames_per_sec=2500, ap_fraction=0.5, n_icao=6, bitflip1=0.2)
    exp, _ = oracle_run(iq, "sc16", rate=rate, phase_enhance=bool(kw.get("phase_enhance")), also_ac=bool(kw.get("mode_ac")))
    raw = np.ascontiguousarray(iq).view(np.uint8)
    plan = shard.plan_shards(raw.size // 4, 65536, world)
    held = []

    def make_shard(r):
        first, count = plan[r]
        lead = 65536 if first else 0
        buf = torch.from_numpy(raw[(first - lead) * 4:(first + count) * 4].copy()).to("cuda:0")
        held.append(buf)
        return shard.EngineShard(da.Engine(rate, da.INPUT_SC16, fix_bits=1, **kw), buf.data_ptr(), buf.numel(), lead, first, count, rate, 4)
    res = _run_ranks(world, make_shard)
    assert_same_messages(shard.merge([m for m, _, _ in res]), exp)
    assert (np.isin(exp["msg_type"], (0, 4, 5, 16, 20, 21))).sum() > 100 and res[0][1] >= 2
    for _, _, sh in res:
        assert not sh.e.can_rerun_accept()
        assert sh.e.counters()["scan_launches"] >= sh.rounds          # a scan per round: nothing was repeated half-way


def test_torch_group_over_rccl_single_rank():
    """The collective path bench.py --gpus N uses (backend "nccl" = RCCL, device tensors, pinned staging) with the one
    rank a one-GPU box allows: states, flags and arrays come back as they went in."""
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        g = shard.TorchGroup("cuda:0")
        cache = np.arange(1024, dtype=np.uint32) * 2654435761 % (1 << 24)
        d8 = bytes(range(256)) * 30 + bytes(g.D8_BYTES - 7680)
        for _ in range(3):                           # the staging buffers are reused
            out = g.all_gather({"icao": cache, "d8": d8})
            assert len(out) == 1 and np.array_equal(out[0]["icao"], cache) and out[0]["d8"] == d8
            assert g.all_gather(True) == [True] and g.all_gather(False) == [False]
            assert np.array_equal(g.all_gather(cache)[0], cache)
        iq, _ = synth.capture(65536 * 3, 2.4e6, "sc16", seed=95, frames_per_sec=2500, n_icao=6)
        d = torch.from_numpy(iq.view(np.int16)).to("cuda:0")
        e = da.Engine(2400000, da.INPUT_SC16, fix_bits=1)
        sh = shard.EngineShard(e, d.data_ptr(), d.numel() * 2, 0, 0, len(iq) // 2, 2400000, 4)
        msgs, rounds = shard.decode_engine_sharded(sh, 0, 1, g.all_gather)
        exp, _ = oracle_run(iq, "sc16")
        assert rounds == 1
        assert_same_messages(msgs, exp)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("rate", [2400000, 2000000, 8000000])
def test_shard_deep_in_a_long_stream(rate):
    """Rank 7 of 8 x 4 GiB starts 15 G samples into the stream: positions and timestamps beyond 2^32 samples.  The same
    shard decoded at stream sample F and at F + 2^34 (whole blocks) gives the same list, every timestamp moved on by
    2^34 samples' worth of the 12 MHz clock."""
    B = 65536
    iq, _ = synth.capture(B * 6 + 777, float(rate), "sc16", seed=99, frames_per_sec=2500, ap_fraction=0.3, n_icao=6)
    d = torch.from_numpy(iq.view(np.int16)).to("cuda:0")
    first, count = 2 * B, len(iq) // 2 - 2 * B
    shift = 1 << 34
    lists = []
    for off in (0, shift):
        e = da.Engine(rate, da.INPUT_SC16, fix_bits=1)
        e.run_device(d.data_ptr(), 4 * first)                 # the stream in front of the shard: cache, halo, 8 MS/s state
        halo = e.tail_halo()
        if rate == 8000000:
            e.set_d8_state(e.get_d8_state(want_acc=True))
        e.run_device(d.data_ptr() + 4 * first, 4 * count, first_sample=first + off, halo_ptr=halo)
        lists.append(e.messages().copy())
    a, b = lists
    assert len(a) > 50 and len(a) == len(b)
    ticks = shift * 12000000 // rate
    assert np.array_equal(b["timestamp"].astype(np.uint64), a["timestamp"].astype(np.uint64) + np.uint64(ticks))
    for k in ("msg", "msg_bits", "msg_type", "addr", "crc", "error_bits", "sig_level"):
        assert np.array_equal(a[k], b[k]), k
