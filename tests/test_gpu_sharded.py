"""N > 1 path through the real engine: two ranks (gloo rendezvous, both on device 0 -- the GPU box has one GPU) take the
contiguous unit slices bench.py / sharding.unit_range give them (the loop nest CalSurfG.f90:1144-1145 cut in two),
solve them with their own engine, and complete the receiver-time vector with the all-gather.  The gathered vector must
equal the one-rank result bit for bit, on both ranks.  Also: `python bench.py --gpus 2` typed as is (no launcher)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NX, NSRC, NPER, NREC = 35, 14, 2, 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from dsurftomo_amd import sharding
    from dsurftomo_amd.engine import Engine
    dist.init_process_group("gloo", rank=rank, world_size=world)
    u = synth.units(NX, NSRC, NPER, NREC)
    u["nrec"] = u["nrec"].copy()
    lo, hi = sharding.unit_range(NSRC * NPER, world, rank)
    e = Engine(0)
    e.set_memory_budget(8 << 30)
    pv = np.stack([synth.medium(NX, "smooth", p) for p in range(NPER)])
    e.set_maps(NX, NX, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
    e.plan(u["map_index"][lo:hi], u["scx"][lo:hi], u["scz"][lo:hi], u["nrec"][lo:hi], u["rcx"][lo * NREC:hi * NREC], u["rcz"][lo * NREC:hi * NREC])
    # the rank's slice stays on the device (dsa_solve_device), as on a multi-GPU node where it goes straight into the RCCL all-gather;
    # here the ranks share GPU 0, RCCL cannot join two ranks of one device, so the collective itself runs over gloo on a host copy
    buf = torch.empty(e.ndata, dtype=torch.float32, device="cuda:0")
    e.solve_device(buf.data_ptr())
    torch.cuda.synchronize()
    mine = buf.cpu().numpy()
    assert np.array_equal(mine.view(np.uint32), e.solve().view(np.uint32)), "dsa_solve_device differs from dsa_solve"
    e.close()
    full = sharding.all_gather_times(dist, torch.from_numpy(mine), sharding.ray_counts(u["nrec"], world))
    q.put((rank, full.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_through_the_engine_match_one_rank(engine, world):
    u = synth.units(NX, NSRC, NPER, NREC)
    pv = np.stack([synth.medium(NX, "smooth", p) for p in range(NPER)])
    engine.set_maps(NX, NX, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
    one = engine.traveltimes(**u)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in range(world):
        assert got[r].size == one.size
        assert np.array_equal(got[r].view(np.uint32), one.view(np.uint32)), "rank %d: gathered vector differs from the one-rank result" % r


def test_bench_gpus_2_as_typed():
    """`python bench.py --gpus 2` without a launcher: the parent starts the two ranks (before touching the GPU) and rank 0
    prints one JSON line for the whole job"""
    env = dict(os.environ, DSA_MAX_CHUNK="2048")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 1 and rec["value"] > 0 and rec["config"]["units_per_step"] == 16000


def test_bench_collective_path_over_rccl_with_one_rank():
    """the N-rank code path of bench.py on the hardware that is there: one rank, RCCL (`nccl` backend) process group, the rank's
    receiver times left in HBM by dsa_solve_device, all_gather_into_tensor on the device tensor, host copy after the collective"""
    env = dict(os.environ, DSA_BENCH_FORCE_DIST="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and rec["config"]["units_per_step"] == 16000
