"""(period, source) units over ranks.

Units are independent (reference CalSurfG.f90:1144-1456: the loop body only appends to the output
arrays), and the only exchange is an all-gather(v) of the receiver times so that every rank ends with the
full vector in the reference's (period, source, receiver) order.  Two partitions:

* `unit_range`: one contiguous slice of the unit list per rank -- whole periods when the rank count divides
  the period count (the drop-in level's engines split the reference's loop nest this way);
* `source_shard` (round 3, bench.py): whole SOURCES per rank, every period of a source on the same rank, so
  that the engine can solve the periods of a source side by side (bundles, csrc/bundle_kernel.hip).  A
  rank's units are then not contiguous in the reference's order; `gather_order` gives the positions of its
  receiver times in the full vector and `all_gather_ordered` puts the gathered pieces there.
"""
import numpy as np


def unit_range(total_units, world, rank):
    """half-open slice [lo, hi) of the unit list owned by `rank`"""
    return (total_units * rank) // world, (total_units * (rank + 1)) // world


def ray_counts(nrec, world):
    """receiver-time count of every rank's slice, given the per-unit receiver counts"""
    nrec = np.asarray(nrec)
    out = []
    for r in range(world):
        lo, hi = unit_range(len(nrec), world, r)
        out.append(int(nrec[lo:hi].sum()))
    return out


def all_gather_times(dist, mine, counts, device=None):
    """all-gather(v) of this rank's receiver times (1-D float32 torch tensor) -> full vector on every
    rank.  One collective: slices are padded to the largest count and trimmed afterwards, which works
    for the nccl (= RCCL) backend with device tensors and for gloo with CPU tensors alike."""
    import torch
    world = len(counts)
    if len(set(counts)) == 1:
        full = torch.empty(sum(counts), dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(full, mine.contiguous())
        return full
    cmax = max(counts)
    padded = torch.zeros(cmax, dtype=mine.dtype, device=mine.device)
    padded[:mine.numel()] = mine
    buf = torch.empty(world * cmax, dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(buf, padded)
    return torch.cat([buf[r * cmax:r * cmax + counts[r]] for r in range(world)])


def source_shard(nsrc, nper, world, rank):
    """unit indices (reference order: period outer, source inner) of the sources [nsrc*rank/world, nsrc*(rank+1)/world), all periods"""
    s_lo, s_hi = (nsrc * rank) // world, (nsrc * (rank + 1)) // world
    return (np.arange(nper, dtype=np.int64)[:, None] * nsrc + np.arange(s_lo, s_hi, dtype=np.int64)[None, :]).reshape(-1)


def ray_positions(nrec, units):
    """positions in the full receiver-time vector of the rays of `units` (in that order)"""
    nrec = np.asarray(nrec, np.int64)
    first = np.concatenate([[0], np.cumsum(nrec)])
    units = np.asarray(units, np.int64)
    if units.size == 0:
        return np.zeros(0, np.int64)
    return np.concatenate([np.arange(first[u], first[u + 1], dtype=np.int64) for u in units])


def gather_order(nrec, nsrc, nper, world):
    """(counts, order): receiver-time count of every rank under `source_shard`, and the position in the full vector of every entry of
    the rank-major concatenation an all-gather returns"""
    pos = [ray_positions(nrec, source_shard(nsrc, nper, world, r)) for r in range(world)]
    return [int(p.size) for p in pos], np.concatenate(pos) if pos else np.zeros(0, np.int64)


def all_gather_ordered(dist, mine, counts, order):
    """all-gather(v) of the ranks' receiver times, then into the reference's order: full[order[k]] = gathered[k].
    `order`: torch int64 tensor on mine's device (from gather_order)"""
    import torch
    gathered = all_gather_times(dist, mine, counts)
    full = torch.empty_like(gathered)
    full[order] = gathered
    return full
