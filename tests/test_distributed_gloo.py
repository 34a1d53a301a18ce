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
    # the same gradients through the overlapped (hook-driven) path: two steps, the second one a
    # backward that leaves the first layers without gradients
    torch.manual_seed(0)
    net2 = torch.nn.Sequential(torch.nn.Linear(40, 64), torch.nn.ReLU(), torch.nn.Linear(64, 8),
                               torch.nn.Linear(8, 3))
    p2 = list(net2.parameters())
    ctx.bucket_bytes = 256
    ov = ctx.overlap(p2)
    assert len(ov.buckets) >= 2, [len(b) for b in ov.buckets]
    ov.start_timing()                                             # the scaling diagnostics bench.py reports
    net2[:3](x).sum().backward()
    if rank == 0:
        (net2[3](torch.ones(2, 8)).sum()).backward()
    ov.finish()
    out["overlap"] = [p.grad.clone() for p in p2]
    for p in p2:
        p.grad = None
    # Adam with existing moments on every parameter (one earlier step with all gradients)
    opt = torch.optim.Adam(p2, lr=0.1, eps=0.1)
    for p in p2:
        p.grad = torch.ones_like(p)
    opt.step()
    opt.zero_grad(set_to_none=True)
    before = [p.detach().clone() for p in p2]
    (net2[3](torch.full((2, 8), float(rank + 1))).sum()).backward()      # only the last layer
    ov.finish()
    out["overlap2"] = [None if p.grad is None else p.grad.clone() for p in p2]
    opt.step()
    out["adam_moved"] = [bool((p.detach() != b).any()) for p, b in zip(p2, before)]
    st = ov.stats()
    out["stats"] = dict(st)
    out["gathered"] = ctx.gather_floats(10.0 + rank)
    ov.remove()
    out["tmax"] = ctx.max_over_ranks(1.0 + rank)
    out["tsum"] = ctx.sum_over_ranks(1.0 + rank)
    ctx.barrier()
    # plain numpy through the queue (torch tensors are shared by fd and the worker may exit first)
    for k in ("local", "reduced", "overlap", "overlap2"):
        out[k] = [None if t is None else t.detach().numpy().copy() for t in out[k]]
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
    for r in res:
        for k in ("local", "reduced", "overlap", "overlap2"):
            res[r][k] = [None if a is None else torch.from_numpy(a) for a in res[r][k]]
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
    for k in range(len(res[0]["reduced"])):
        for r in range(world):
            assert torch.allclose(res[r]["overlap"][k], res[r]["reduced"][k], rtol=1e-6, atol=1e-7)
    # second step: the parameters without a gradient on ANY rank end with grad None again (so
    # Adam, which holds moments for them, leaves them alone -- the single-GPU / reference
    # behaviour), the last layer gets the mean over ranks
    for r in range(world):
        assert all(g is None for g in res[r]["overlap2"][:4])
        assert res[r]["adam_moved"] == [False] * 4 + [True, True]
        assert torch.allclose(res[r]["overlap2"][4], torch.full((3, 8), 2 * 1.5))      # mean of 2*1 and 2*2
        assert torch.allclose(res[r]["overlap2"][5], torch.full((3,), 2.0))
    for r in range(world):
        st = res[r]["stats"]
        assert st["finishes"] == 2 and st["buckets"] >= 2 and len(st["bucket_bytes"]) == st["buckets"]
        # every parameter's bytes + one flag per parameter, as sent
        assert st["allreduce_bytes_per_finish"] == (40 * 64 + 64 + 64 * 8 + 8 + 8 * 3 + 3 + 6) * 4
        assert st["exposed_wait_ms_per_finish"] >= 0.0 and st["exposed_wait_ms_max"] >= st["exposed_wait_ms_per_finish"]
        # step 2 produced gradients for the last layer only: the other buckets were launched by finish()
        assert 0 < st["late_buckets_per_finish"] <= st["buckets"]
        assert res[r]["gathered"] == [10.0, 11.0]
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
