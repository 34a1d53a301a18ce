#!/usr/bin/env python3
"""Latent out-of-bounds accesses made visible: the hot-path chain stage by stage with torch's caching allocator OFF (every
tensor its own hipMalloc: an access past the end of a buffer meets an unmapped page instead of a neighbour's bytes) and
blocking launches, so that a fault names its stage.
    PYTORCH_NO_CUDA_MEMORY_CACHING=1 HIP_LAUNCH_BLOCKING=1 python3 tools/probes/oob_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from kernel_bench import synth_rpn  # noqa: E402
from wssdl_bus_amd import _lib  # noqa: E402
from wssdl_bus_amd.fast_rcnn.config import cfg  # noqa: E402
from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op  # noqa: E402
from wssdl_bus_amd.rpn_msr import proposal_target_layer_tf_bus as ptl  # noqa: E402
from wssdl_bus_amd.rpn_msr.anchor_target_layer_tf_bus import anchor_target_layer_joint  # noqa: E402
from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import proposal_layer  # noqa: E402


def say(*a):
    print(*a, flush=True)


cfg.SAMPLING_RNG = "device"
cfg.ROI_POOL_ANNOUNCE_BWD_FORM = False
say("caching off:", os.environ.get("PYTORCH_NO_CUDA_MEMORY_CACHING"), "blocking:", os.environ.get("HIP_LAUNCH_BLOCKING"))
for (S, WS, H, W, C, blocks) in ((4, 4, 38, 63, 1024, -1), (0, 2, 38, 63, 1024, 1), (1, 2, 37, 62, 512, 1), (2, 0, 38, 63, 256, -1)):
    N = S + WS
    cfg.TRAIN.IMS_PER_BATCH, cfg.TRAIN.WS_IMS_PER_BATCH = S, WS
    _lib.set_tuning("roi_fwd_blocks", blocks)
    prob, pred = synth_rpn(N, H, W, 9, 3)
    info = torch.tensor([[H * 16.0 - 8, W * 16.0 - 8, 1.0, 1.0]] * N, device="cuda")
    gt = torch.zeros((N, 20, 5), device="cuda")
    for i in range(S):
        gt[i, 0] = torch.tensor([100.0 + 20 * i, 80.0, 380.0 + 20 * i, 300.0, 1.0])
        gt[i, 1] = torch.tensor([500.0, 60.0 + 10 * i, 900.0, 420.0, 0.0])
    ng = torch.tensor([2] * S + [0] * WS, dtype=torch.int32, device="cuda")
    score = torch.zeros((N, H, W, 18), device="cuda")
    feat = torch.relu(torch.randn((N, H, W, C), device="cuda", generator=torch.Generator("cuda").manual_seed(4)))
    for padded in (False, True):
        for rep in range(2):
            cfg.PADDED_ROIS = padded
            say("shape", (S, WS, H, W, C), "blocks", blocks, "padded", padded, "rep", rep)
            if S:
                at = anchor_target_layer_joint(score, gt, ng, info, None, True, [16, ], [8, 16, 32], "SNUBH")
                torch.cuda.synchronize()
                say("  anchor targets ok")
            rois = proposal_layer(prob, pred, info, True, False)
            torch.cuda.synchronize()
            say("  proposal layer ok", tuple(rois.shape))
            if S:
                out = ptl.proposal_target_layer_joint(rois, gt, ng, 3, True)
                r = out[0].contiguous()
            else:
                r = rois.contiguous()
            torch.cuda.synchronize()
            say("  proposal targets ok", tuple(r.shape))
            top, arg8 = op.roi_pool_compact(feat, r, 7, 7, 1.0 / 16)
            torch.cuda.synchronize()
            say("  RoI pool forward ok")
            plan = op.prepare_backward(tuple(feat.shape), r, 7, 7, 1.0 / 16)
            torch.cuda.synchronize()
            say("  backward lists ok:", plan.variant[:40])
            g = op.roi_pool_grad_compact(tuple(feat.shape), r, arg8, top, 7, 7, 1.0 / 16, plan=plan, segments=plan.segments)
            torch.cuda.synchronize()
            say("  RoI pool backward ok")
            op.check_flags()
            del top, arg8, g, plan, r, rois
_lib.set_tuning("roi_fwd_blocks", -1)
say("all stages ok")
