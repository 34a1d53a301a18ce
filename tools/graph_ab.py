#!/usr/bin/env python3
"""hipGraph A/B of the hot-path chain at the default workload's scale (round-5 review, weak item 4).

The chain -- anchor targets, proposal layer (padded blob), proposal targets (device sampler), RoI pool forward,
backward lists, RoI pool backward -- is ~30 launches, most of them latency-bound.  cfg.PADDED_ROIS makes it free of host
syncs, hence capturable (tests/test_gpu_padded.py).  This tool times it three ways on 4 supervised + 4 weak images,
38 x 63 x 1024: eager (compacting default, one read-back), eager padded, and the padded chain replayed from ONE captured
graph; the GPU time of a pass is measured with events around 30 passes each.

    python3 tools/graph_ab.py [--iters 30]      -> JSON lines
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from kernel_bench import synth_rpn  # noqa: E402
from wssdl_bus_amd import _lib  # noqa: E402
from wssdl_bus_amd.fast_rcnn.config import cfg  # noqa: E402
from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op  # noqa: E402
from wssdl_bus_amd.rpn_msr import proposal_target_layer_tf_bus as ptl  # noqa: E402
from wssdl_bus_amd.rpn_msr import anchor_target_layer_tf_bus as atl  # noqa: E402
from wssdl_bus_amd.rpn_msr.anchor_target_layer_tf_bus import anchor_target_layer_joint  # noqa: E402
from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import proposal_layer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
args = ap.parse_args()
S, WS, H, W, C = 4, 4, 38, 63, 1024
N = S + WS
cfg.TRAIN.IMS_PER_BATCH, cfg.TRAIN.WS_IMS_PER_BATCH = S, WS
cfg.SAMPLING_RNG = "device"
prob, pred = synth_rpn(N, H, W, 9, 3)
info = torch.tensor([[600, 1000, 1.0, 1.0]] * N, device="cuda")
gt = torch.zeros((N, 20, 5), device="cuda")
for i in range(S):
    gt[i, 0] = torch.tensor([100.0 + 20 * i, 80.0, 380.0 + 20 * i, 300.0, 1.0])
    gt[i, 1] = torch.tensor([500.0, 60.0 + 10 * i, 900.0, 420.0, 0.0])
ng = torch.tensor([2] * S + [0] * WS, dtype=torch.int32, device="cuda")
score = torch.zeros((N, H, W, 18), device="cuda")
feat = torch.relu(torch.randn((N, H, W, C), device="cuda", generator=torch.Generator("cuda").manual_seed(4)))
data = torch.zeros((N, 600, 1000, 3), device="cuda")


def chain():
    ptl._device_calls[0] = 41              # both device samplers draw from a per-call counter: pinned, so that the eager
    atl._device_calls[0] = 17              # and the captured passes sample alike
    at = anchor_target_layer_joint(score, gt, ng, info, data, True, [16, ], [8, 16, 32], "SNUBH")
    rois = proposal_layer(prob, pred, info, True, False)
    out = ptl.proposal_target_layer_joint(rois, gt, ng, 3, True)
    r = out[0].contiguous()
    top, arg8 = op.roi_pool_compact(feat, r, 7, 7, 1.0 / 16)
    plan = op.prepare_backward(tuple(feat.shape), r, 7, 7, 1.0 / 16)
    g = op.roi_pool_grad_compact(tuple(feat.shape), r, arg8, top, 7, 7, 1.0 / 16, plan=plan, segments=plan.segments)
    return at[0], r, top, g


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import time
    t0 = time.perf_counter()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters, (time.perf_counter() - t0) * 1e3 / iters


def launches(fn):
    _lib.timeline.reset(True)
    fn()
    torch.cuda.synchronize()
    n = sum(d["calls"] for d in _lib.timeline.summary().values())
    _lib.timeline.reset(False)
    return n


cfg.ROI_POOL_ANNOUNCE_BWD_FORM = False
res = {}
cfg.PADDED_ROIS = False
res["eager_compacting"] = timed(chain, args.iters)
print("# eager compacting done", flush=True)
cfg.PADDED_ROIS = True
res["eager_padded"] = timed(chain, args.iters)
print("# eager padded done", flush=True)
ops = launches(chain)
print("# counted", ops, flush=True)
graph = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    chain()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
with torch.cuda.graph(graph):
    captured = chain()
print("# captured", flush=True)
graph.replay()
torch.cuda.synchronize()
print("# one replay done", flush=True)
res["graph_replay_padded"] = timed(graph.replay, args.iters)
print("# replays done", flush=True)
eager = chain()
torch.cuda.synchronize()
graph.replay()
torch.cuda.synchronize()
same = all(torch.equal(a, b) for a, b in zip(eager, captured))
print(json.dumps(dict(chain="anchor targets + proposal layer + proposal targets + RoI pool fwd + lists + RoI pool bwd",
                      images=N, rois=int(captured[1].shape[0]), library_calls_per_pass=ops, outputs_equal=same,
                      gpu_ms_per_pass={k: round(v[0], 4) for k, v in res.items()},
                      host_ms_per_pass={k: round(v[1], 4) for k, v in res.items()})))
