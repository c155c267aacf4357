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
practice two.  At 2.0 / 2.4 MS/s the scan kernel's records do not depend on the cache, so rounds after the first
repeat only the ordered accept pass (`modes_engine_rerun_accept`).  No data-path collective: only the states and
the (small) message lists travel.
"""
import numpy as np


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
    entered = [guess] * world                    # what every rank decoded from in this round (rank 0's never changes)
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
        new_entry = entry if rank == 0 else exits[rank - 1]
        fresh = not _same(new_entry, entry)
        entry = new_entry
        entered = [entered[0]] + [exits[j - 1] for j in range(1, world)]


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
        if self.rate != 8000000 and self.scanned and self.count:
            e.rerun_accept(stream=self.stream)                       # the scan's records do not depend on the cache
        else:
            halo = None
            if self.first:
                # the magnitudes in front of the shard, from the block the rank holds in front of it
                e.halo_at(self.buf_ptr, self.buf_nbytes, self.lead, self.halo.data_ptr(), stream=self.stream)
                halo = self.halo.data_ptr()
            if self.rate == 8000000 and self.first:
                e.set_d8_state(D8State.from_buffer_copy(entry["d8"]))
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
    entry = shard.initial_state()
    entered = [entry] * world
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
        if max_rounds is not None and rounds >= max_rounds + 1:
            raise RuntimeError("state hand-over did not settle in %d rounds" % rounds)
        if asked and not need_acc:
            need_acc = True                      # the states just exchanged lack the average: exchange again, with it
            fresh = False
            continue
        if all(_same(exits[j - 1], entered[j]) for j in range(1, world)):
            return rounds
        new_entry = entry if rank == 0 else exits[rank - 1]
        fresh = not _same(new_entry, entry)
        entry = new_entry
        entered = [entered[0]] + [exits[j - 1] for j in range(1, world)]
