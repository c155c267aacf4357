"""Sharding one long recording over several GPUs (ref: src/infile.c:116-197).

A recording is cut into contiguous ranges of whole 256 KiB blocks, one range per rank.  What a shard needs from the
stream before it:

  (a) the `overlap` magnitudes in front of its first sample -- a pure function of the samples, each rank forms them
      itself from the few samples it holds in front of its shard (`modes_engine_halo_at`);
  (b) the 1024-slot ICAO cache at its first block, which depends on every message before it;
  (c) at 8 MS/s, what demod_8000 carries between calls (ref: src/demod-8000.c:80-89): the last 1928 differences (a pure
      function of the previous shard's samples), the match window still open and the trigger history (which depend on
      where the window chain stood when it entered the previous shard), and the running average (only read by shards
      with near-silent stretches, only formed when one of them asks).

(b) and (c) are resolved the way the blocks inside a shard are: decode with a guess, exchange what the shards end with
(a few KiB each), decode again from the neighbour's exit state, stop when no entry state changes.  Decisions only depend
on earlier positions, so the fixed point is the sequential result; it is reached after at most `world` rounds, in
practice two at any world size (`next_entries`: every rank sees every exit and composes them).  At 2.0 / 2.4 MS/s the scan kernel's records do not depend on the cache, so rounds after the first
repeat only the ordered accept pass (`modes_engine_rerun_accept`).  No data-path collective: only the states and
the (small) message lists travel.
"""
import ctypes

import numpy as np


def C_memmove(dst, src):
    ctypes.memmove(dst, src, ctypes.sizeof(dst))


def plan_shards(nsamples, block_samples, world):
    """[(first_sample, nsamples_of_shard)] per rank: whole blocks, as even as possible; empty shards at the end."""
    nblocks = -(-nsamples // block_samples)
    base, extra = divmod(nblocks, world)
    out, b0 = [], 0
    for r in range(world):
        nb = base + (1 if r < extra else 0)
        first = min(b0 * block_samples, nsamples)
        last = min((b0 + nb) * block_samples, nsamples)
        out.append((first, last - first))
        b0 += nb
    return out


class LocalGroup:
    """Stand-in for a process group when all shards are decoded by one process, one after the other."""
    def __init__(self, world):
        self.world = world


def _same(a, b):
    if isinstance(a, dict):
        return a.keys() == b.keys() and all(_same(a[k], b[k]) for k in a)
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        return np.array_equal(a, b)
    return a == b


def _icao_of(state):
    return state["icao"] if isinstance(state, dict) else state


def _with_icao(state, icao):
    if isinstance(state, dict):
        out = dict(state)
        out["icao"] = icao
        return out
    return icao


def next_entries(entered, exits):
    """The entry states of the next round, for every rank, from what every rank entered with and left with in this one.

    Rank j's entry is rank j - 1's exit.  Handing the exits on one rank per round would carry an address heard in shard
    0 to shard 7 in seven rounds; but every rank sees every exit, and a shard's exit cache is its entry cache with the
    slots it wrote itself replaced (ref: src/dump1090.c:1455-1461: a slot is overwritten, never merged).  So the exit
    rank j - 1 WOULD have left, had it been entered with the new entry, is predicted as: the slots it wrote (where its
    exit differs from what it entered with) from its exit, every other slot passed through from its new entry.  The
    prediction is only a guess for the next round -- the loop still ends only when every rank's entry IS its
    neighbour's exit -- but it is exact whenever a shard's own writes do not depend on its entry cache (only clean
    DF11/17/18 write, ref: src/dump1090.c:2676-2687), which makes two rounds the rule at any world size.  A rank whose
    predecessors are all settled gets its predecessor's exit unchanged, so at most `world` rounds are needed as before.
    Anything else a state carries (demod_8000's "d8") is taken from the neighbour's exit as it is."""
    out = [entered[0]]
    for j in range(1, len(entered)):
        wrote = _icao_of(exits[j - 1]) != _icao_of(entered[j - 1])
        icao = np.where(wrote, _icao_of(exits[j - 1]), _icao_of(out[j - 1])).astype(np.uint32)
        out.append(_with_icao(exits[j - 1], icao))
    return out


def decode_sharded(decode, rank, world, all_gather, initial_icao=None, max_rounds=None, initial_state=None):
    """Runs `decode(entry_state) -> (messages, exit_state)` for this rank's shard until the entry states of all ranks
    are consistent.  A state is the ICAO cache (uint32[1024]) or a dict of named parts (compared part by part).
    `all_gather(obj) -> [obj of rank 0, ...]`.  A rank whose entry state did not change keeps its result.
    One collective per round: every rank sees every exit state, so every rank can tell by itself whether any rank's
    entry state (= its neighbour's exit) still moved.  Ranks behind the first all start from the same guess
    (`initial_state`, or an empty cache).  Returns (messages of this shard, rounds)."""
    guess = initial_state if initial_state is not None else np.zeros(1024, dtype=np.uint32)
    if initial_state is not None:
        entry = initial_state
    else:
        entry = guess if initial_icao is None or rank > 0 else np.asarray(initial_icao, dtype=np.uint32)
    first = guess if (initial_state is not None or initial_icao is None) else np.asarray(initial_icao, dtype=np.uint32)
    entered = [first] + [guess] * (world - 1)   # what every rank decoded from in this round (rank 0's never changes)
    rounds = 0
    msgs = exit_state = None
    fresh = True
    while True:
        if fresh:
            msgs, exit_state = decode(entry)
        rounds += 1
        exits = all_gather(exit_state)
        settled = all(_same(exits[j - 1], entered[j]) for j in range(1, world))
        if settled:
            return msgs, rounds
        if max_rounds is not None and rounds >= max_rounds:
            raise RuntimeError("state hand-over did not settle in %d rounds" % rounds)
        entered = next_entries(entered, exits)
        fresh = not _same(entered[rank], entry)
        entry = entered[rank]


def merge(per_rank_messages):
    """Shards are consecutive in time, so the merged list is their concatenation in rank order."""
    parts = [m for m in per_rank_messages if len(m)]
    return np.concatenate(parts) if parts else per_rank_messages[0][:0]


class ThreadGroup:
    """all_gather between `world` threads of one process (tests, and hosts that drive several GPUs from one process)."""
    def __init__(self, world):
        import threading
        self.world = world
        self.slots = [None] * world
        self.barrier = threading.Barrier(world)

    def all_gather(self, rank):
        def gather(obj):
            self.slots[rank] = obj
            self.barrier.wait()
            out = list(self.slots)
            self.barrier.wait()
            return out
        return gather


class TorchGroup:
    """all_gather over torch.distributed (one process per GPU; backend "nccl" is RCCL on ROCm, "gloo" on CPU).  The
    hand-over states are a few KiB: they travel as one fixed-size byte tensor per rank, flags as one int.  The staging
    tensors (pinned host, device) are kept per message size: a round costs two small copies and one collective."""
    D8_BYTES = 4 + 4 + 8 + 8 + 8 + 4 * 1928

    def __init__(self, device):
        import torch.distributed as dist
        self.dist, self.device = dist, device
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self._bufs = {}

    def _gather_bytes(self, blob):
        import torch
        n = len(blob)
        if n not in self._bufs:
            on_gpu = str(self.device).startswith("cuda")
            h_in = torch.empty(n, dtype=torch.uint8, pin_memory=on_gpu)
            h_out = torch.empty(self.world * n, dtype=torch.uint8, pin_memory=on_gpu)
            d_in = torch.empty(n, dtype=torch.uint8, device=self.device) if on_gpu else h_in
            d_out = torch.empty(self.world * n, dtype=torch.uint8, device=self.device) if on_gpu else h_out
            self._bufs[n] = (h_in, h_out, d_in, d_out, on_gpu)
        h_in, h_out, d_in, d_out, on_gpu = self._bufs[n]
        h_in.numpy()[:] = np.frombuffer(blob, dtype=np.uint8)
        if on_gpu:
            d_in.copy_(h_in, non_blocking=True)
        self.dist.all_gather_into_tensor(d_out, d_in)
        if on_gpu:
            h_out.copy_(d_out, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
        return h_out.numpy().reshape(self.world, -1).copy()

    def all_gather(self, obj):
        if isinstance(obj, (bool, np.bool_)):
            return [bool(r[0]) for r in self._gather_bytes(bytes([int(obj)]))]
        if isinstance(obj, dict):
            blob = np.asarray(obj["icao"], dtype=np.uint32).tobytes() + obj.get("d8", b"")
            rows = self._gather_bytes(blob)
            out = []
            for r in rows:
                st = {"icao": np.frombuffer(r[:4096].tobytes(), dtype=np.uint32).copy()}
                if "d8" in obj:
                    st["d8"] = r[4096:4096 + self.D8_BYTES].tobytes()
                out.append(st)
            return out
        if isinstance(obj, np.ndarray):
            rows = self._gather_bytes(np.ascontiguousarray(obj).tobytes())
            return [np.frombuffer(r.tobytes(), dtype=obj.dtype).reshape(obj.shape).copy() for r in rows]
        raise TypeError("TorchGroup.all_gather: %r" % type(obj))


class EngineShard:
    """One rank's shard of a recording behind a `dump1090_amd.Engine`: `decode` is what `decode_sharded` calls.

    The rank holds bytes [buf_ptr, buf_ptr + buf_nbytes) of the stream in device memory: `lead_samples` samples in
    front of its shard (whole blocks; one block for every rank but the first -- the overlap magnitudes in front of the
    shard are formed from them) and then the shard's `count` samples, which start at stream sample `first_sample`.
    The state handed from shard to shard is {"icao": uint32[1024]} plus, at 8 MS/s, {"d8": bytes of modes_d8_state}."""

    def __init__(self, engine, buf_ptr, buf_nbytes, lead_samples, first_sample, count, sample_rate, bytes_per_sample, stream=None):
        import torch
        self.e, self.first, self.count = engine, first_sample, count
        self.buf_ptr, self.buf_nbytes, self.lead = buf_ptr, buf_nbytes, lead_samples
        self.rate, self.bps, self.stream = sample_rate, bytes_per_sample, stream
        self.ptr = buf_ptr + lead_samples * bytes_per_sample
        self.ov = int((8 + 112 + 16) * 1e-6 * sample_rate)          # ref: src/dump1090.c:734
        block = 262144 // bytes_per_sample
        if lead_samples % block or (first_sample and lead_samples < block):
            raise ValueError("the lead-in must be whole blocks, one at least for a shard behind the stream's start")
        self.halo = torch.zeros(self.ov + 8, dtype=torch.int16, device="cuda") if first_sample else None
        self.reset()

    def reset(self):
        """Forget the last decode (a new pass over the shard scans again)."""
        self.scanned = False
        self.rounds = 0
        self.amb_steps = 0
        self.entry_states = None                                    # set by settle_chain()

    def settle_chain(self, rank, world, all_gather):
        """8 MS/s: where the window chain enters every shard, before anything is decoded (one collective).

        The differences and the trigger history in front of a shard are functions of the samples in front of it: the
        rank forms them from the block it holds there (decoded as a stream of its own).  Only `pending`, the window
        left open by the shard before, depends on all earlier shards, and through 168 possible values: every rank
        reports what it would hand on for each (`modes_engine_d8_chain_map`) and follows the maps from the stream's
        start.  Returns the entry states of all ranks, or None when near-silent input makes the trigger depend on the
        running average (then the states are found by rounds, decode_engine_rounds)."""
        from . import D8State, D8_CHAIN_STATES, ModesError
        e = self.e
        block = 262144 // self.bps
        local = D8State()
        local.pending, local.acc_valid = 1, 1
        ok = True
        if self.first:
            e.set_icao(None)
            e.run_device(self.ptr - block * self.bps, block * self.bps, first_sample=0, stream=self.stream)
            s = e.get_d8_state(want_acc=False, stream=self.stream)
            # the block starts cold (no differences in front of it), which only its first steps feel; what is read here
            # are its last 64 trigger decisions: they must not hang on the running average
            ok = e.d8_last_ambiguous() <= block // 8 - 64
            local.trig_tail = s.trig_tail
            C_memmove(local.dprev, s.dprev)
            e.set_icao(None)
        cmap = np.zeros(D8_CHAIN_STATES, dtype=np.int32)
        if ok and self.count:
            halo = None
            if self.first:
                e.halo_at(self.buf_ptr, self.buf_nbytes, self.lead, self.halo.data_ptr(), stream=self.stream)
                halo = self.halo.data_ptr()
                e.set_d8_state(local)
            try:
                cmap = e.d8_chain_map(self.ptr, self.count * self.bps, first_sample=self.first, halo_ptr=halo, stream=self.stream)
            except ModesError:
                ok = False
        elif ok:                                                     # an empty shard hands on what it gets
            cmap = np.array([1] + [-i for i in range(1, D8_CHAIN_STATES)], dtype=np.int32)
        blob = np.concatenate([np.array([int(ok)], dtype=np.int32), cmap]).view(np.uint8)
        rows = all_gather(np.concatenate([blob, np.frombuffer(bytes(local), dtype=np.uint8)]))
        if not all(int(r[:4].view(np.int32)[0]) for r in rows):
            e.set_d8_state(None)
            return None
        states, pending = [], 1
        for j in range(world):
            st = D8State.from_buffer_copy(rows[j][4 + 4 * D8_CHAIN_STATES:].tobytes())
            st.pending, st.acc, st.acc_valid, st.amb_steps = pending, 0, 1, 0
            states.append({"icao": np.zeros(1024, dtype=np.uint32), "d8": bytes(st)})
            m = rows[j][4:4 + 4 * D8_CHAIN_STATES].view(np.int32)
            pending = int(m[0 if pending == 1 else -pending])
        self.entry_states = states
        return states

    def initial_state(self):
        st = {"icao": np.zeros(1024, dtype=np.uint32)}
        if self.rate == 8000000:
            from . import D8State
            z = D8State()
            z.pending, z.acc_valid = 1, 1                            # a guess for round one: nothing carried in
            st["d8"] = bytes(z)
        return st

    def decode(self, entry):
        from . import D8State
        e = self.e
        self.rounds += 1
        e.set_icao(entry["icao"])
        # (engines that keep their ordered pass on the host -- mode_ac, phase_enhance, host_replay -- cannot repeat the
        # accept pass alone: they decode the shard again, on every rank alike, so no rank is left waiting in a collective)
        if self.rate != 8000000 and self.scanned and self.count and e.can_rerun_accept():
            e.rerun_accept(stream=self.stream)                       # the scan's records do not depend on the cache
        else:
            halo = None
            if self.first:
                # the magnitudes in front of the shard, from the block the rank holds in front of it
                e.halo_at(self.buf_ptr, self.buf_nbytes, self.lead, self.halo.data_ptr(), stream=self.stream)
                halo = self.halo.data_ptr()
            if self.rate == 8000000 and self.first:
                e.set_d8_state(D8State.from_buffer_copy(entry["d8"]))
            if self.rate == 8000000 and (self.scanned or self.entry_states is not None):
                e.d8_reuse_planes()                                  # same samples as the call before: the trigger planes stand
            e.run_device(self.ptr, self.count * self.bps, first_sample=self.first, halo_ptr=halo, stream=self.stream)
            self.scanned = True
        out = {"icao": e.get_icao()}
        if self.rate == 8000000:
            out["d8"] = self._d8_exit(entry)
        return None, out                                             # the list stays in the engine: messages() fetches it

    def messages(self):
        return self.e.messages()

    def _d8_exit(self, entry):
        from . import D8State
        if self.count == 0:
            return entry["d8"]
        s = self.e.get_d8_state(want_acc=False, stream=self.stream)
        self.amb_steps = int(s.amb_steps)
        s.amb_steps = 0                                              # describes the run, not the state handed on
        s.acc, s.acc_valid = 0, 1                                    # acc travels only on request: see exit_with_acc()
        return bytes(s)

    def exit_with_acc(self, exit_state):
        """The same exit state with the running average formed (a walk over every step of the shard)."""
        from . import D8State
        if self.count == 0:
            return exit_state
        s = self.e.get_d8_state(want_acc=True, stream=self.stream)
        s.amb_steps = 0
        out = dict(exit_state)
        out["d8"] = bytes(s)
        return out


def decode_engine_sharded(shard, rank, world, all_gather, max_rounds=None):
    """decode_engine_rounds, then this shard's messages.  Returns (messages of this shard, rounds)."""
    rounds = decode_engine_rounds(shard, rank, world, all_gather, max_rounds)
    return shard.messages(), rounds


def decode_engine_rounds(shard, rank, world, all_gather, max_rounds=None):
    """decode_sharded over an EngineShard; the list stays in the shard's engine.  Returns the number of rounds.

    One extra rule at 8 MS/s: when a shard behind the first holds steps whose trigger depends on demod_8000's running
    average (near-silent input), every shard forms that average and hands it on as well.  The first round does not
    count for this: its entry states are guesses (no differences carried in looks like silence)."""
    if shard.rate != 8000000:
        _, rounds = decode_sharded(shard.decode, rank, world, all_gather, max_rounds=max_rounds, initial_state=shard.initial_state())
        return rounds
    # One collective per round: the exit states travel with a flag in d8.amb_steps (a field that describes a run, not
    # the state handed on): "this shard, behind the first, holds steps that need the running average".
    from . import D8State
    states = shard.settle_chain(rank, world, all_gather)
    if states is None:                               # near-silent input: guesses, and a round per shard if it comes to that
        entry = shard.initial_state()
        entered = [entry] * world
    else:
        entry, entered = states[rank], states
    need_acc = with_acc = False
    fresh = True
    rounds = 0
    while True:
        if fresh:
            _, exit_state = shard.decode(entry)
            with_acc = False
        rounds += 1
        if need_acc and not with_acc:
            exit_state = shard.exit_with_acc(exit_state)
            with_acc = True
        st = D8State.from_buffer_copy(exit_state["d8"])
        st.amb_steps = 1 if (shard.amb_steps and rank > 0 and rounds > 1) else 0
        gathered = all_gather({"icao": exit_state["icao"], "d8": bytes(st)})
        exits, asked = [], False
        for g in gathered:
            st = D8State.from_buffer_copy(g["d8"])
            asked = asked or bool(st.amb_steps)
            st.amb_steps = 0
            exits.append({"icao": g["icao"], "d8": bytes(st)})
        if asked and not need_acc:
            need_acc = True                      # the states just exchanged lack the average: exchange again, with it
            fresh = False
            rounds -= 1                          # ... which decodes nothing and is no round of the budget
            continue
        if all(_same(exits[j - 1], entered[j]) for j in range(1, world)):
            return rounds
        if max_rounds is not None and rounds >= max_rounds:
            raise RuntimeError("state hand-over did not settle in %d rounds" % rounds)
        entered = next_entries(entered, exits)
        fresh = not _same(entered[rank], entry)
        entry = entered[rank]
