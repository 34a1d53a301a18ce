"""World-size-2 CPU tests (gloo) of the data-parallel layer: image sharding, per-rank seeds,
bucketed gradient all-reduce (including parameters without a gradient on some rank), and the
max-over-ranks timing reduction bench.py uses."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from wssdl_bus_amd.distributed import DistContext
    ctx = DistContext(backend="gloo", bucket_bytes=1 << 12)      # tiny buckets: several collectives
    out = {}
    out["shard"] = ctx.shard_images(7)
    out["seed"] = ctx.seed(3)
    torch.manual_seed(0)                                          # same init on every rank
    net = torch.nn.Sequential(torch.nn.Linear(40, 64), torch.nn.ReLU(), torch.nn.Linear(64, 8),
                              torch.nn.Linear(8, 3))
    params = list(net.parameters())
    torch.manual_seed(100 + rank)                                 # different data per rank
    x = torch.randn(5, 40)
    net[:3](x).sum().backward()                                   # last layer gets NO gradient
    if rank == 0:                                                 # ... except on rank 0
        (net[3](torch.ones(2, 8)).sum()).backward()
    local = [None if p.grad is None else p.grad.clone() for p in params]
    ctx.allreduce_gradients(params)
    out["local"] = local
    out["reduced"] = [p.grad.clone() for p in params]
    out["tmax"] = ctx.max_over_ranks(1.0 + rank)
    out["tsum"] = ctx.sum_over_ranks(1.0 + rank)
    ctx.barrier()
    q.put((rank, out))
    ctx.shutdown()


@pytest.mark.timeout(180)
def test_data_parallel_gloo_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    # image sharding: disjoint, covering, contiguous
    assert res[0]["shard"] == [0, 1, 2, 3] and res[1]["shard"] == [4, 5, 6]
    assert (res[0]["seed"], res[1]["seed"]) == (3, 4)
    # all-reduce = mean over ranks, zeros for ranks without a gradient
    for k in range(len(res[0]["reduced"])):
        g = [res[r]["local"][k] for r in range(world)]
        g = [torch.zeros_like(res[0]["reduced"][k]) if x is None else x for x in g]
        want = (g[0] + g[1]) / world
        for r in range(world):
            assert torch.allclose(res[r]["reduced"][k], want, rtol=1e-6, atol=1e-7)
    assert res[0]["tmax"] == res[1]["tmax"] == 2.0
    assert res[0]["tsum"] == res[1]["tsum"] == 3.0


def test_single_process_context_is_a_noop():
    sys.path.insert(0, ROOT)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    from wssdl_bus_amd.distributed import DistContext
    ctx = DistContext()
    assert not ctx.enabled and ctx.shard_images(3) == [0, 1, 2]
    p = torch.nn.Parameter(torch.ones(3))
    ctx.allreduce_gradients([p])
    assert p.grad is None and ctx.max_over_ranks(2.5) == 2.5
