"""-m gpu: the data-parallel path on the REAL train step.

(a) one rank with WSSDL_FORCE_DIST=1 over RCCL ("nccl" backend): the bucketed, backward-overlapped
    all-reduce path must leave the step unchanged -- as close to the plain step as a repeat of the
    plain step is to itself (bit-for-bit when the convolutions are deterministic).
(b) two ranks sharing the one GPU over gloo (RCCL refuses two ranks on one device): after a
    combined step and after an alternating iteration every rank holds identical parameters, equal
    to Adam applied to the MEAN of the two ranks' gradients (computed serially in-process).

Children are started fresh (multiprocessing 'spawn'); nothing re-execs a process that touched the GPU."""
import os
import socket
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(rank, world, port, backend, force):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.pop("WSSDL_FORCE_DIST", None)
    os.environ.pop("WSSDL_DIST_BACKEND", None)
    if force:
        os.environ["WSSDL_FORCE_DIST"] = "1"
    if backend:
        os.environ["WSSDL_DIST_BACKEND"] = backend
    import numpy as np
    import torch
    from wssdl_bus_amd.fast_rcnn.config import cfg
    from wssdl_bus_amd.networks.factory_bus import get_network
    torch.cuda.set_device(0)
    cfg.TRAIN.IMS_PER_BATCH, cfg.TRAIN.WS_IMS_PER_BATCH = 1, 1
    cfg.SAMPLING_RNG = "reference"            # host numpy stream: re-seedable, so passes can be replayed
    torch.manual_seed(0)                      # identical initial weights on every rank
    return np, torch, cfg, get_network


def _params(net):
    return [p for p in net.parameters() if p.requires_grad]


def _flat(torch, ps):
    return torch.cat([p.detach().reshape(-1) for p in ps])


def _load(net, state):
    net.load_state_dict(state)
    for p in net.parameters():
        p.grad = None


def _worker_force_dist(port, q):
    try:
        np, torch, cfg, get_network = _setup(0, 1, port, None, force=False)
        import copy
        from wssdl_bus_amd import synthetic
        from wssdl_bus_amd.distributed import DistContext
        from wssdl_bus_amd.fast_rcnn.train_bus import SolverWrapper
        net = get_network("Resnet_train", 18).cuda().to(memory_format=torch.channels_last)
        net.train()
        state0 = copy.deepcopy(net.state_dict())
        blobs = synthetic.make_batch(1, 1, 320, 480, seed=21)
        outs = []
        for kind in ("plain", "plain", "plain", "plain", "dist"):
            _load(net, state0)
            ctx = None
            if kind == "dist":
                os.environ["WSSDL_FORCE_DIST"] = "1"
                ctx = DistContext(bucket_bytes=4 << 20)           # several buckets on ResNet-18
                assert ctx.enabled and ctx.backend == "nccl" and ctx.world_size == 1
            solver = SolverWrapper(net, dist_ctx=ctx)
            cfg.DEVICE_RNG_SEED = 3
            np.random.seed(7)
            losses = solver.train_step_joint(blobs)
            torch.cuda.synchronize()
            outs.append((_flat(torch, _params(net)).clone(), float(losses["loss"])))
            if kind == "dist":
                assert len(solver.overlap.buckets) >= 2
                solver.overlap.remove()
                ctx.shutdown()
        plain, b = outs[:-1], outs[-1]
        a = plain[0]
        moved = float((a[0] - _flat(torch, [state0[k] for k, _ in net.named_parameters()])).abs().max())
        # the plain step's own run-to-run noise (MIOpen's weight gradients use atomics): the largest difference
        # among four repeats, and the data-parallel step against each of them
        repeat = max(float((x[0] - y[0]).abs().max()) for i, x in enumerate(plain) for y in plain[i + 1:])
        dist_d = min(float((x[0] - b[0]).abs().max()) for x in plain)
        q.put(dict(ok=True, repeat=repeat, dist=dist_d, moved=moved, loss=(a[1], plain[1][1], b[1])))
    except Exception as e:                                   # surface the child's failure in the parent
        import traceback
        q.put(dict(ok=False, err=traceback.format_exc() + repr(e)))


@pytest.mark.timeout(600)
def test_force_dist_single_rank_rccl_step_equals_plain_step():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_force_dist, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=540)
    p.join(timeout=60)
    assert res["ok"], res.get("err")
    assert res["moved"] > 0                                   # the step did change the parameters
    # the RCCL path may differ from the plain step by no more than the plain step differs from
    # its own repeat (0 when the backward is deterministic -> bit-for-bit)
    if res["repeat"] == 0.0:
        assert res["dist"] == 0.0, res                        # deterministic backward: bit-for-bit
    else:
        # `repeat` is the largest difference among FOUR plain repeats (six pairs), `dist` the distance of the RCCL
        # step to the nearest of them: a lost or unscaled bucket shows up at the scale of `moved`, orders above
        # (an absolute floor beside it: four noisy repeats can by chance lie closer together than the fifth run does)
        assert res["dist"] <= max(2 * res["repeat"], 1e-3 * res["moved"]), res
    assert abs(res["loss"][0] - res["loss"][2]) <= 4 * abs(res["loss"][0] - res["loss"][1]) + 1e-6


def _worker_two_ranks(rank, world, port, q):
    try:
        np, torch, cfg, get_network = _setup(rank, world, port, "gloo", force=False)
        import copy
        import torch.distributed as dist
        from wssdl_bus_amd import synthetic
        from wssdl_bus_amd.distributed import DistContext
        from wssdl_bus_amd.fast_rcnn.train_bus import SolverWrapper
        out = {}
        ctx = DistContext(bucket_bytes=4 << 20)
        assert ctx.enabled and ctx.backend == "gloo" and ctx.world_size == 2

        def expected_step(net, state, backward, blobs_of_rank, seeds, lr):
            """Adam on the mean over ranks of the gradients `backward` leaves, from `state`."""
            grads = []
            for r in range(world):
                _load(net, state)
                np.random.seed(seeds[r])
                backward(blobs_of_rank[r])
                grads.append([None if p.grad is None else p.grad.detach().clone() for p in _params(net)])
            _load(net, state)
            ps = _params(net)
            opt = torch.optim.Adam(ps, lr=lr, eps=0.1)
            for i, p in enumerate(ps):
                gs = [g[i] for g in grads]
                if all(g is None for g in gs):
                    continue                                   # no gradient on any rank: skipped
                p.grad = sum((torch.zeros_like(p) if g is None else g) for g in gs) / world
            opt.step()
            torch.cuda.synchronize()
            new_state = copy.deepcopy(net.state_dict())
            return new_state, [g is not None for g in grads[0]]

        def identical_across_ranks(ps):
            f = _flat(torch, ps)
            hi, lo = f.clone(), f.clone()
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            return bool(torch.equal(hi, lo))

        # ------------------------------------------------ combined step (train.py)
        net = get_network("Resnet_train", 18).cuda().to(memory_format=torch.channels_last)
        net.train()
        state0 = copy.deepcopy(net.state_dict())
        blobs = [synthetic.make_batch(1, 1, 320, 480, seed=100 + r) for r in range(world)]
        plain = SolverWrapper(net)
        exp_state, _ = expected_step(net, state0, plain.joint_backward, blobs, [50, 51], plain.lr)
        expect = _flat(torch, [exp_state[k] for k, _ in net.named_parameters()]).clone()
        # the same expectation once more: what the non-deterministic convolution gradients alone
        # move (the yardstick for the comparison below)
        # (three repeats, the largest difference: one repeat is a single sample of that noise and has come out 10 x
        # below the next one)
        noise = 0.0
        for _ in range(3):
            exp_again, _ = expected_step(net, state0, plain.joint_backward, blobs, [50, 51], plain.lr)
            noise = max(noise, float((expect - _flat(torch, [exp_again[k] for k, _ in net.named_parameters()])).abs().max()))
        out["joint_noise"] = noise
        _load(net, state0)
        solver = SolverWrapper(net, dist_ctx=ctx)
        np.random.seed(50 + rank)
        solver.train_step_joint(blobs[rank])
        torch.cuda.synchronize()
        got = _flat(torch, _params(net))
        out["joint_identical"] = identical_across_ranks(_params(net))
        out["joint_err"] = float((got - expect).abs().max())
        out["joint_moved"] = float((got - _flat(torch, [state0[k] for k, _ in net.named_parameters()])).abs().max())
        worst = []
        for (name, p_) in net.named_parameters():
            d_ = (p_.detach() - exp_state[name]).abs()
            worst.append((float(d_.max()), name, float((exp_state[name] - state0[name]).abs().max()),
                          float((exp_again[name] - exp_state[name]).abs().max())))
        out["joint_worst"] = sorted(worst, reverse=True)[:4]
        solver.overlap.remove()

        # ------------------------------------------------ alternating iteration (train_alter.py)
        torch.manual_seed(0)
        net2 = get_network("Resnet_train_alter", 18).cuda().to(memory_format=torch.channels_last)
        net2.train()
        s0 = copy.deepcopy(net2.state_dict())
        blobs_s = [synthetic.make_batch(1, 0, 320, 480, seed=200 + r) for r in range(world)]
        blobs_w = [synthetic.make_batch(0, 1, 320, 480, seed=300 + r) for r in range(world)]
        plain2 = SolverWrapper(net2)
        s1, _ = expected_step(net2, s0, plain2.supervised_backward, blobs_s, [60, 61], plain2.lr)
        s2, had = expected_step(net2, s1, plain2.weak_backward, blobs_w, [70, 71], plain2.lr)
        names = [k for k, _ in net2.named_parameters()]
        out["weak_nograd_params"] = int(sum(1 for h in had if not h))
        # parameters without a weak gradient must come out of the weak step untouched
        out["weak_nograd_static"] = all(bool(torch.equal(s1[n], s2[n])) for n, h in zip(names, had) if not h)
        expect2 = _flat(torch, [s2[k] for k in names]).clone()
        noise2 = 0.0
        for _ in range(3):
            s1b, _ = expected_step(net2, s0, plain2.supervised_backward, blobs_s, [60, 61], plain2.lr)
            s2b, _ = expected_step(net2, s1b, plain2.weak_backward, blobs_w, [70, 71], plain2.lr)
            noise2 = max(noise2, float((expect2 - _flat(torch, [s2b[k] for k in names])).abs().max()))
        out["alter_noise"] = noise2
        _load(net2, s0)
        solver2 = SolverWrapper(net2, dist_ctx=ctx)
        orig_apply = solver2._apply
        calls = []

        def apply_spy(optimizer=None, count_step=True):
            if not calls:
                np.random.seed(70 + rank)        # the weak half draws after the first apply
            calls.append(count_step)
            return orig_apply(optimizer, count_step)
        solver2._apply = apply_spy
        np.random.seed(60 + rank)
        solver2.train_step_alter(blobs_s[rank], blobs_w[rank])
        torch.cuda.synchronize()
        got2 = _flat(torch, _params(net2))
        out["alter_identical"] = identical_across_ranks(_params(net2))
        out["alter_err"] = float((got2 - expect2).abs().max())
        out["alter_step"] = solver2.global_step
        out["alter_calls"] = calls
        solver2.overlap.remove()
        ctx.barrier()
        out["ok"] = True
        q.put((rank, out))
        ctx.shutdown()
    except Exception as e:
        import traceback
        q.put((rank, dict(ok=False, err=traceback.format_exc() + repr(e))))


@pytest.mark.timeout(900)
def test_two_ranks_on_one_gpu_real_steps_match_mean_gradient_update():
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_two_ranks, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=800) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        assert res[r]["ok"], res[r].get("err")
    for r in range(world):
        o = res[r]
        assert o["joint_identical"] and o["alter_identical"], o
        assert o["joint_moved"] > 1e-5
        # Adam(eps=0.1, lr=5e-4) on the mean gradient: the data-parallel result may differ from
        # the serial expectation by no more than a few times what a REPEAT of the serial
        # computation differs from itself (MIOpen's weight gradients are not deterministic),
        # and in any case by less than 2 % of the largest possible Adam step (lr)
        # (the yardstick is the largest of three repeats)
        assert o["joint_err"] <= max(10 * o["joint_noise"], 2e-6), o
        assert o["alter_err"] <= max(10 * o["alter_noise"], 2e-6), o
        assert o["joint_err"] <= 1e-5 and o["alter_err"] <= 1e-5, o
        assert o["alter_step"] == 1 and o["alter_calls"] == [False, True]
        assert o["weak_nograd_params"] > 0 and o["weak_nograd_static"]


@pytest.mark.timeout(900)
def test_bench_py_under_torch_distributed_run_two_ranks():
    """bench.py ITSELF under the launcher the scaling run uses (`python -m torch.distributed.run
    --nproc-per-node 2 ... bench.py --gpus 2`), two ranks sharing the one GPU over gloo
    (WSSDL_DIST_BACKEND; RCCL refuses two ranks on one device).  The launcher starts fresh children
    before anything touches the GPU; this process only waits for it.  Checks the contract of the one
    JSON line: n_gpus, whole-job value = ranks x images / max-rank time, roofline present, no
    cpu_baseline at N > 1."""
    import json
    import subprocess
    env = dict(os.environ, WSSDL_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "WSSDL_FORCE_DIST", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", "resnet18_sup_b2", "--no-cpu-baseline", "--roofline-iters", "3"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=840)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-2000:] + p.stderr.decode()[-4000:]
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]                      # rank 0 alone prints
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1 and r["scaling"] == "weak"
    assert r["config"]["workload"] == "resnet18_sup_b2" and r["config"]["images_per_gpu"] == 2
    assert "gloo" in r["config"]["parallelism"] and "dp2" in r["config"]["parallelism"]
    # whole-job aggregate: 2 ranks x 2 images per step / the (max over ranks) step time
    assert abs(r["value"] - 2 * 2 / (r["ms_per_step"] * 1e-3)) <= 0.01 * r["value"]
    assert "cpu_baseline" not in r
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["kernel"] in ("roi_pool_forward", "roi_pool_backward")
    assert 0 < rf["frac"] < 1 and rf["frac"] == rf["frac_moved"] and rf["frac_8d"] > rf["frac_moved"]
    assert r["final_loss"] is not None and r["final_loss"] == r["final_loss"]        # not NaN
    assert r["n_ranks_seen"] == 2                            # both ranks took part in a collective
    # the diagnostics one SCALE line needs to explain a miss of the scaling target
    sd = r["scaling_diag"]
    assert len(sd["ms_per_step_per_rank"]) == 2 and all(v > 0 for v in sd["ms_per_step_per_rank"])
    assert max(sd["ms_per_step_per_rank"]) <= r["ms_per_step"] * 1.001
    assert len(sd["exposed_allreduce_ms_per_optimizer_step_per_rank"]) == 2
    assert all(0 <= v < r["ms_per_step"] for v in sd["exposed_allreduce_ms_per_optimizer_step_per_rank"])
    assert sd["buckets"] >= 1 and len(sd["bucket_bytes"]) == sd["buckets"] and sd["backend"] == "gloo"
    assert sd["allreduce_bytes_per_optimizer_step"] == sum(sd["bucket_bytes"]) > 40e6       # ResNet-18 variant, f32 gradients
    assert sd["optimizer_steps_per_step"] == 1.0 and sd["late_buckets_per_optimizer_step"] is not None


@pytest.mark.timeout(900)
def test_bare_bench_py_gpus_2_starts_its_own_launcher():
    """`python3 bench.py --gpus 2` with no launcher around it (the shape of the driver's N = 1 command):
    bench.py starts `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 ...` as a child
    before it imports torch, relays rank 0's one JSON line and the exit code."""
    import json
    import subprocess
    env = dict(os.environ, WSSDL_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "WSSDL_FORCE_DIST", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", "resnet18_sup_b2", "--no-cpu-baseline", "--roofline-iters", "3"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=840)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-2000:] + p.stderr.decode()[-4000:]
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["n_ranks_seen"] == 2 and r["steps"] == 2 and r["scaling"] == "weak"
    assert abs(r["value"] - 2 * 2 / (r["ms_per_step"] * 1e-3)) <= 0.01 * r["value"]
