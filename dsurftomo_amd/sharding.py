"""(period, source) units over ranks.

Units are independent (reference CalSurfG.f90:1144-1456: the loop body only appends to the output
arrays), so a rank takes one contiguous slice of the unit list -- whole periods when the rank
count divides the period count, which keeps one velocity map hot per rank -- and the only
exchange is an all-gather(v) of the receiver times so that every rank ends with the full vector
in the reference's (period, source, receiver) order.
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
