"""N > 1 path on CPU: two gloo ranks shard the unit list the way bench.py / the engine host do and
complete the receiver-time vector with an all-gather(v).  The travel times themselves are faked by
a deterministic function of the ray index (no GPU here); what is under test is the partition,
ordering and the collective."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dsurftomo_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nrec, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nrec = np.asarray(nrec)
    first = np.concatenate([[0], np.cumsum(nrec)])
    lo, hi = sharding.unit_range(len(nrec), world, rank)
    mine = torch.arange(int(first[lo]), int(first[hi]), dtype=torch.float32) * 0.5 + 1.0     # "times" of my rays
    full = sharding.all_gather_times(dist, mine, sharding.ray_counts(nrec, world))
    q.put((rank, full.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nrec", [[4] * 16, [3, 0, 5, 1, 2, 7, 4]])
def test_two_ranks_assemble_the_reference_order(nrec):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nrec, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = np.arange(sum(nrec), dtype=np.float32) * 0.5 + 1.0
    for r in range(world):
        assert np.array_equal(got[r], expect)


def test_unit_ranges_tile_the_list():
    for total in (16000, 17, 1):
        for world in (1, 2, 3, 8):
            edges = [sharding.unit_range(total, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
    # whole periods per rank when the rank count divides the period count
    lo, hi = sharding.unit_range(16 * 1000, 8, 3)
    assert lo % 1000 == 0 and hi % 1000 == 0


def _worker_sources(rank, world, port, nsrc, nper, nrec, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nrec = np.asarray(nrec)
    mine_units = sharding.source_shard(nsrc, nper, world, rank)
    pos = sharding.ray_positions(nrec, mine_units)
    mine = torch.from_numpy(pos.astype(np.float32) * 0.5 + 1.0)                # "times" of my rays, in my plan's order
    counts, order = sharding.gather_order(nrec, nsrc, nper, world)
    full = sharding.all_gather_ordered(dist, mine, counts, torch.from_numpy(order))
    q.put((rank, full.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nsrc,nper,nrec", [(6, 4, None), (5, 3, [2, 0, 3, 1, 4, 2, 2, 5, 1, 0, 3, 2, 1, 1, 6])])
def test_two_ranks_sharded_by_sources_assemble_the_reference_order(nsrc, nper, nrec):
    """bench.py's partition since round 3: whole sources per rank (all their periods: the engine bundles them); the all-gather's
    rank-major pieces go to their places in the reference's (period, source, receiver) order"""
    world = 2
    if nrec is None:
        nrec = [3] * (nsrc * nper)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sources, args=(r, world, port, nsrc, nper, nrec, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = np.arange(sum(nrec), dtype=np.float32) * 0.5 + 1.0
    for r in range(world):
        assert np.array_equal(got[r], expect)


def test_source_shards_tile_the_units():
    for nsrc, nper in ((1000, 16), (7, 3), (1, 5)):
        for world in (1, 2, 3, 8):
            parts = [sharding.source_shard(nsrc, nper, world, r) for r in range(world)]
            allu = np.sort(np.concatenate(parts))
            assert np.array_equal(allu, np.arange(nsrc * nper))
            for part in parts:                      # every period of a source on one rank
                srcs = set((part % nsrc).tolist())
                assert part.size == len(srcs) * nper
