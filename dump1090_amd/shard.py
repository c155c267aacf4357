"""Sharding one long recording over several GPUs (ref: src/infile.c:116-197).

A recording is cut into contiguous ranges of whole 256 KiB blocks, one range per rank.  What a shard needs from the
stream before it: (a) the `overlap` magnitudes in front of its first sample -- a pure function of the samples, each
rank forms them itself -- and (b) the 1024-slot ICAO cache at its first block, which depends on every message before
it.  (b) is resolved the way the blocks inside a shard are: decode with a guess, exchange the caches the shards end
with (4 KiB each), decode again with the neighbour's exit cache, stop when no entry cache changes.  Decisions only depend
on earlier positions, so the fixed point is the sequential result; it is reached after at most `world` rounds, in
practice two.  No data-path collective: only the caches and the (small) message lists travel.
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


def decode_sharded(decode, rank, world, all_gather, initial_icao=None, max_rounds=None):
    """Runs `decode(entry_icao) -> (messages, exit_icao)` for this rank's shard until the entry caches of all ranks
    are consistent.  `all_gather(obj) -> [obj of rank 0, ...]`.  Returns (messages of this shard, rounds)."""
    entry = np.zeros(1024, dtype=np.uint32) if initial_icao is None or rank > 0 else np.asarray(initial_icao, dtype=np.uint32)
    rounds = 0
    while True:
        msgs, exit_icao = decode(entry)
        rounds += 1
        exits = all_gather(np.asarray(exit_icao, dtype=np.uint32))
        new_entry = entry if rank == 0 else np.asarray(exits[rank - 1], dtype=np.uint32)
        changed = bool(np.any(new_entry != entry))
        if not any(all_gather(changed)):
            return msgs, rounds
        if max_rounds is not None and rounds >= max_rounds:
            raise RuntimeError("ICAO hand-over did not settle in %d rounds" % rounds)
        entry = new_entry


def merge(per_rank_messages):
    """Shards are consecutive in time, so the merged list is their concatenation in rank order."""
    parts = [m for m in per_rank_messages if len(m)]
    return np.concatenate(parts) if parts else per_rank_messages[0][:0]
