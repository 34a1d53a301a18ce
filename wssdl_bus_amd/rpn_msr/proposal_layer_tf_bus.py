"""proposal_layer on the GPU.

Reference: code/lib/rpn_msr/proposal_layer_tf_bus.py:19-148.  Same call
signature; inputs NHWC.  All images of the batch are processed by one
``wssdl_proposal_layer`` call (decode -> rank/top-K -> NMS mask -> sweep), then one
small device-to-host copy of the per-image counts sizes the output blob.
"""
import numpy as np
import torch

from .. import _lib
from ..fast_rcnn.config import cfg
from .generate_anchors import generate_anchors

DEBUG = False


def _rpn_params(is_training):
    key = "TRAIN" if is_training else "TEST"                       # :42
    c = cfg[key]
    return c.RPN_PRE_NMS_TOP_N, c.RPN_POST_NMS_TOP_N, c.RPN_NMS_THRESH, c.RPN_MIN_SIZE


def proposal_layer_padded(rpn_cls_prob_reshape, rpn_bbox_pred, im_info, is_training,
                          _feat_stride=[16, ], anchor_scales=[8, 16, 32], debug=False,
                          from_logits=False):
    """Device-resident form: returns (rois_padded [N, post_nms_topN, 5], counts [N] i32)
    without any host synchronisation (+ decoded / sorted_index / sorted_count when
    `debug`)."""
    prob = _lib.to_device(rpn_cls_prob_reshape, torch.float32)
    pred = _lib.to_device(rpn_bbox_pred, torch.float32, prob.device)
    info = _lib.to_device(im_info, torch.float32, prob.device)
    if prob.dim() != 4 or pred.dim() != 4 or info.dim() != 2:
        raise ValueError("expected rpn_cls_prob_reshape/rpn_bbox_pred [N,H,W,C] and im_info [N,k]")
    pre, post, thresh, min_size = _rpn_params(bool(is_training))
    if cfg.USE_GPU_NMS:
        # the reference's proposal layer calls nms_wrapper.nms (proposal_layer_tf_bus.py:139), which cfg.USE_GPU_NMS sends
        # to the CUDA kernel's `>` / f32-threshold rule
        from ..fast_rcnn.nms_wrapper import gpu_rule_threshold
        thresh = gpu_rule_threshold(thresh)
    anchors = generate_anchors(scales=np.array(anchor_scales))
    A = anchors.shape[0]
    N, H, W = prob.shape[:3]
    if prob.shape[3] != 2 * A or pred.shape[3] != 4 * A or tuple(pred.shape[:3]) != (N, H, W):
        raise ValueError("channel counts do not match %d anchors" % A)
    if info.shape[0] != N:
        raise ValueError("im_info rows must match the batch size")   # :40 loops im_info.shape[0]
    M = H * W * A
    topn = pre if 0 < pre < M else M
    pitch = post if post > 0 else topn
    stride = int(np.asarray(_feat_stride).ravel()[0])
    L = _lib.lib()
    dev = prob.device
    with torch.cuda.device(dev):
        ws = torch.empty((L.wssdl_proposal_workspace_bytes(N, H, W, A, pre),), dtype=torch.uint8,
                         device=dev)
        rois = torch.empty((N, pitch, 5), dtype=torch.float32, device=dev)
        counts = torch.empty((N,), dtype=torch.int32, device=dev)
        dec = sidx = scnt = None
        if debug:
            dec = torch.empty((N, M, 4), dtype=torch.float32, device=dev)
            sidx = torch.empty((N, topn), dtype=torch.int32, device=dev)
            scnt = torch.empty((N,), dtype=torch.int32, device=dev)
        with _lib.timed("proposal_layer", dict(N=N, H=H, W=W, A=A, pre=topn, post=pitch)):
          fn = L.wssdl_proposal_layer_from_logits if from_logits else L.wssdl_proposal_layer
          _lib.check(fn(
            _lib.ptr(prob), _lib.ptr(pred), _lib.ptr(info), info.shape[1], N, H, W,
            _lib.host_ptr(anchors), A, stride, int(pre), int(post), float(thresh),
            float(min_size), _lib.ptr(rois), _lib.ptr(counts), _lib.ptr(dec), _lib.ptr(sidx),
            _lib.ptr(scnt), _lib.ptr(ws), ws.numel(), _lib.stream()), "wssdl_proposal_layer")
    if debug:
        return rois, counts, dec, sidx, scnt
    return rois, counts


def compact_rois(rois_padded, counts, counts_host=None):
    """[N, P, 5] + counts -> the reference's contiguous blob [sum counts, 5]."""
    N, P = rois_padded.shape[:2]
    if counts_host is None:
        counts_host = counts.cpu().numpy()
    if (np.asarray(counts_host) < 0).any():
        # WSSDL_NMS_TIMED_OUT: the sweep of that image gave up waiting for the fused launch's mask blocks
        raise _lib.HipCallError("proposal layer: NMS sweep timed out for image(s) %s (roi count -1): the GPU was held "
                                "by another process or kernel; the proposals are incomplete"
                                % np.nonzero(np.asarray(counts_host) < 0)[0].tolist())
    total = int(counts_host.sum())
    out = torch.empty((total, 5), dtype=torch.float32, device=rois_padded.device)
    with torch.cuda.device(rois_padded.device):
        _lib.check(_lib.lib().wssdl_proposal_compact(_lib.ptr(rois_padded), _lib.ptr(counts), N, P,
                                                     _lib.ptr(out), total, _lib.stream()),
                   "wssdl_proposal_compact")
    # rows are grouped by image: let the proposal-target layer reuse the counts instead of
    # reading the batch column back
    out._wssdl_counts = tuple(int(c) for c in counts_host)
    return out


def padded_blob(rois_padded, counts):
    """cfg.PADDED_ROIS: the fixed-shape form of the blob, [N * post_nms_topN, 5] with batch index
    -1 on the rows an image does not use -- no device->host copy.  RoI pooling treats such rows as
    empty, the proposal-target layer never samples them, the MIL selection never matches them."""
    N, P = rois_padded.shape[:2]
    # a negative count (WSSDL_NMS_TIMED_OUT) leaves the image without live rows here (a step with no proposals for
    # that image, not a corrupt one) and raises the deferred flag: the next poll_flags() (SolverWrapper, before every
    # optimiser step) switches the process to the two-launch NMS and warns once; check_flags() (tests, bench.py after
    # its timed region) raises
    from ..roi_pooling_layer import roi_pooling_op as _rp
    _rp.note_roi_counts(counts)
    live = torch.arange(P, device=rois_padded.device).view(1, P) < counts.view(N, 1).to(torch.int64)
    out = rois_padded.clone()
    out[:, :, 0] = torch.where(live, out[:, :, 0], torch.full_like(out[:, :, 0], -1.0))
    out = out.reshape(N * P, 5)
    out._wssdl_pitch = P
    out._wssdl_counts_dev = counts
    return out


_timeout_warned = [False]


def note_nms_timeout(where):
    """A sweep of the fused mask + sweep launch gave up waiting (roi count -1).  The two-launch form has no waits between
    workgroups and gives identical results (tests/test_gpu_edges.py::test_fused_mask_sweep_launch_equals_two_launches),
    so the run carries on with it: warn once, never die where recomputing is possible."""
    if not _timeout_warned[0]:
        _timeout_warned[0] = True
        import warnings
        warnings.warn("proposal layer: an NMS sweep of the fused launch timed out (%s) -- the GPU was held by another "
                      "process or kernel; falling back to the two-launch NMS (same results)" % where, RuntimeWarning)


# After a time-out the per-call path stays on the two-launch NMS for this many calls before it tries the fused launch
# again: a GPU that is shared with another process stays shared, and every fused call would stall for the whole wait
# ("nms_wait_us") before it recovers.  Same results either way.
NMS_TIMEOUT_COOLDOWN_CALLS = 1000
_cooldown = [0]


def _run(scores, rpn_bbox_pred, im_info, is_training, _feat_stride, anchor_scales, from_logits):
    as_np = _lib.wants_numpy(scores, rpn_bbox_pred, im_info)
    if _cooldown[0] > 0 and not (cfg.PADDED_ROIS and not as_np):
        _cooldown[0] -= 1
        with _lib.tuned(nms_fused=0):
            rois, counts = proposal_layer_padded(scores, rpn_bbox_pred, im_info, is_training, _feat_stride, anchor_scales,
                                                 from_logits=from_logits)
    else:
        rois, counts = proposal_layer_padded(scores, rpn_bbox_pred, im_info, is_training, _feat_stride, anchor_scales,
                                             from_logits=from_logits)
    if cfg.PADDED_ROIS and not as_np:
        return padded_blob(rois, counts)
    counts_host = counts.cpu().numpy()
    if (counts_host < 0).any():
        # WSSDL_NMS_TIMED_OUT: recompute THIS call with mask and sweep as two launches (no cross-workgroup waits) and
        # keep the following calls on that form for a cool-down
        note_nms_timeout("image(s) %s" % np.nonzero(counts_host < 0)[0].tolist())
        _cooldown[0] = int(NMS_TIMEOUT_COOLDOWN_CALLS)
        with _lib.tuned(nms_fused=0):
            rois, counts = proposal_layer_padded(scores, rpn_bbox_pred, im_info, is_training, _feat_stride,
                                                 anchor_scales, from_logits=from_logits)
        counts_host = counts.cpu().numpy()
    blob = compact_rois(rois, counts, counts_host)
    return blob.cpu().numpy() if as_np else blob


def proposal_layer(rpn_cls_prob_reshape, rpn_bbox_pred, im_info, is_training, is_ws,
                   _feat_stride=[16, ], anchor_scales=[8, 16, 32]):
    """Same contract as the reference: returns the rois blob [sum R_i, 5] f32 with
    rows (batch_idx, x1, y1, x2, y2).  `is_ws` is accepted and unused, as in the
    reference.  numpy in -> numpy out; GPU tensors in -> GPU tensor out."""
    return _run(rpn_cls_prob_reshape, rpn_bbox_pred, im_info, is_training, _feat_stride, anchor_scales, False)


def proposal_layer_from_score(rpn_cls_score, rpn_bbox_pred, im_info, is_training, is_ws=False,
                              _feat_stride=[16, ], anchor_scales=[8, 16, 32]):
    """f2: proposal_layer fed with the raw ``rpn_cls_score`` [N,H,W,2A]: the reference's
    reshape_layer(2) -> softmax -> reshape_layer(2A) chain (Resnet_train_bus.py:76-81) is fused
    into the decode kernel.  Same output contract as proposal_layer."""
    return _run(rpn_cls_score, rpn_bbox_pred, im_info, is_training, _feat_stride, anchor_scales, True)
