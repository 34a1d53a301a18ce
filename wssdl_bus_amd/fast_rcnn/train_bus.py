"""Multi-task loss and train steps (combined and alternating) on PyTorch-ROCm.

Reference: code/lib/fast_rcnn/train_bus.py:184-270 (alternating losses), :603-678 (combined
losses), :286-301 / :694-705 (optimisers and the per-variable sum of the two gradient sets),
:334-394 (one alternating iteration = a supervised step then a weak step).  Only what drives
the hot path's backward is restated here; data loading, LR
schedules, snapshots and logging are out of scope.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from ..mil import core as mil_core
from ..roi_pooling_layer import roi_pooling_op
from .config import cfg


# ----------------------------------------------------------------- losses ---

def rpn_cls_loss(rpn_cls_score_reshape, rpn_labels):
    """train_bus.py:186-192 / :605-610: CE over anchors whose label != -1.
    rpn_cls_score_reshape [N, A*H, W, 2]; rpn_labels [N, 1, A*H, W] int."""
    score = rpn_cls_score_reshape.reshape(-1, 2)
    label = rpn_labels.reshape(-1).to(torch.int64)
    # mean CE over the anchors whose label != -1 (tf.gather of the kept rows in the reference);
    # ignore_index does the same without materialising the kept rows (no host sync)
    return F.cross_entropy(score, label, ignore_index=-1)


def rpn_box_loss(rpn_bbox_pred, rpn_data, n_images=None):
    """train_bus.py:203-210 / :613-620.  The 'smooth L1' switches at |d| < 1 while using the
    sigma = 3 pieces (0.5*(3*in_w*d)^2 and |d| - 0.5/9): the reference's formula, kept as is.
    rpn_bbox_pred [N,H,W,4A]; rpn_data = (labels, targets, inside_w, outside_w) [N,4A,H,W]."""
    tg, inw, outw = (t.permute(0, 2, 3, 1) for t in rpn_data[1:4])
    pred = rpn_bbox_pred
    if n_images is not None:                       # combined mode slices the supervised images
        pred, tg, inw, outw = pred[:n_images], tg[:n_images], inw[:n_images], outw[:n_images]
    d = pred - tg
    sign = (d.abs() < 1).to(pred.dtype)
    per = outw * (0.5 * (inw * d * 3) ** 2 * sign + (d.abs() - 0.5 / 9.0) * (sign - 1).abs())
    return per.sum(dim=(1, 2)).mean() * 10


def rcnn_cls_loss(cls_score, labels):
    """train_bus.py:218 / :623-630: CE over the first len(labels) rows."""
    label = labels.reshape(-1).to(torch.int64)
    # label -1 = padding row of a fixed-shape RoI list (an image short of candidates): not a row of
    # the reference's blob, so it is left out of the mean
    return F.cross_entropy(cls_score[:label.numel()], label, ignore_index=-1)


def rcnn_box_loss(bbox_pred, roi_data):
    """train_bus.py:231-235 / :641-647: plain L1, mean over rows of the weighted row sums."""
    tg, inw, outw = roi_data[2], roi_data[3], roi_data[4]
    pred = bbox_pred[:tg.shape[0]]
    per_row = (outw * (inw * (pred - tg).abs())).sum(dim=1)
    # mean over the rows of the reference's blob: padding rows (label -1, zero weights) do not count
    n_rows = (roi_data[1].reshape(-1) >= 0).sum().clamp_min(1)
    return per_row.sum() / n_rows


def mil_loss(cls_score_ws, batch_inds, mil_label, n_bags, global_step, funcs, counts_host=None):
    """train_bus.py:239-260 / :650-671: bag logits -> CE weighted by the class prior
    [0, WS_MAL_PCT, 1-WS_MAL_PCT] and by 1 - 0.99*0.9^floor(step/2000) (or a constant).
    On the GPU the bag selection is the HIP op (no host round trip); on CPU tensors (tests) the
    host-side restatement of mil/core.py runs."""
    if cfg.TRAIN.WS_LOSS_USE_ADAPTIVE_SCALE_FACTOR:
        scale = 1.0 - 0.99 * (0.9 ** (int(global_step) // 2000))      # exponential_decay, staircase
    else:
        scale = cfg.TRAIN.WS_LOSS_SCALE_FACTOR
    # (the fused MIL op takes 3..8 classes, the reference has 3; anything else: selection op + torch CE)
    if cls_score_ws.is_cuda and cfg.get('FUSED_LOSS', True) and 3 <= cls_score_ws.shape[1] <= 8:
        # selection + weighted CE + mean as one device op with its own backward (f1)
        prior = [0.0, float(cfg.TRAIN.WS_MAL_PCT), 1.0 - float(cfg.TRAIN.WS_MAL_PCT)]
        return mil_core.mil_loss_device(cls_score_ws, batch_inds, 0.0, mil_label, n_bags, funcs, prior, scale)
    valid = None
    if cls_score_ws.is_cuda:
        bag_logits, _, valid = mil_core.get_bag_logit_device(cls_score_ws, batch_inds, 0.0, mil_label,
                                                             n_bags, funcs, return_valid=True)
    else:
        bag_logits, _ = mil_core.get_bag_logit(cls_score_ws, batch_inds, 3, mil_label, n_bags,
                                               funcs, counts_host)
    label = mil_label.reshape(-1).to(torch.int64)
    w = _class_prior(bag_logits)[label]
    if valid is not None:                          # an empty bag (no proposals) carries no loss
        w = w * valid.to(w.dtype)
    ce = F.cross_entropy(bag_logits, label, reduction='none')
    return (scale * (w * ce)).mean()


_prior_cache = {}


def _class_prior(like):
    """[0, WS_MAL_PCT, 1 - WS_MAL_PCT] on like's device (train_bus.py:252,664), built once: a tensor
    made from a Python list is a blocking host->device copy."""
    key = (float(cfg.TRAIN.WS_MAL_PCT), str(like.device), like.dtype)
    t = _prior_cache.get(key)
    if t is None:
        t = torch.tensor([0.0, key[0], 1 - key[0]], dtype=like.dtype, device=like.device)
        _prior_cache[key] = t
    return t


def l2_weight_decay(params):
    """train_bus.py:268-270: sum(l2_loss(w)) * WEIGHT_DECAY over '*weights' variables."""
    if not params:
        return 0.0
    return torch.stack([(p * p).sum() for p in params]).sum() * (0.5 * cfg.TRAIN.WEIGHT_DECAY)


def supervised_loss(layers, params, n_sup=None):
    # (the device op takes 2 <= K <= 32 classes; wider heads take the torch chain)
    if cfg.get('FUSED_LOSS', True) and layers['cls_score'].is_cuda and 'rpn_cls_score' in layers \
            and 2 <= layers['cls_score'].shape[1] <= 32:
        # the four terms and their gradients as one device op (csrc/loss.hip)
        from .loss_op import multi_task_loss, TERMS
        terms = multi_task_loss(layers['rpn_cls_score'], layers['rpn_bbox_pred'], layers['cls_score'],
                                layers['bbox_pred'], layers['rpn-data'], layers['roi-data'], n_sup)
        l = {name: terms[i] for i, name in enumerate(TERMS)}
    else:
        l = dict(
            rpn_cross_entropy=rpn_cls_loss(layers['rpn_cls_score_reshape'], layers['rpn-data'][0]),
            rpn_loss_box=rpn_box_loss(layers['rpn_bbox_pred'], layers['rpn-data'], n_sup),
            cross_entropy=rcnn_cls_loss(layers['cls_score'], layers['roi-data'][1]),
            loss_box=rcnn_box_loss(layers['bbox_pred'], layers['roi-data']),
        )
    l['weight_decay'] = l2_weight_decay(params)
    l['loss'] = (l['cross_entropy'] + l['loss_box'] + l['rpn_cross_entropy'] + l['rpn_loss_box']
                 + l['weight_decay'])
    return l


# ------------------------------------------------------------ train steps ---

class SolverWrapper(object):
    """The part of the reference's SolverWrapper (train_bus.py:97-131) that a step needs:
    the network, Adam(eps=0.1) (:286-289,:694-695), the global step, and -- new in this
    implementation -- gradient averaging across data-parallel ranks."""

    def __init__(self, network, lr=None, dist_ctx=None):
        self.net = network
        self.lr = cfg.TRAIN.get('LEARNING_RATE', 0.0005) if lr is None else lr
        self.params = [p for p in network.parameters() if p.requires_grad]
        # combined mode: ONE Adam applies the summed gradients and counts the global step
        # (:694-705).  Alternating mode: the supervised op is Adam.minimize(loss) WITHOUT a
        # global step, the weak op a SECOND Adam whose apply_gradients counts it (:286-301);
        # the two keep their own moments and bias-correction powers.
        self.optimizer = self._adam()
        self.optimizer_ws = None                  # created by the first alternating iteration
        self.global_step = 0
        self.dist = dist_ctx
        if dist_ctx is not None and dist_ctx.enabled:
            # seed per rank = seed + rank, for the device samplers too; a single
            # process keeps the DEVICE_RNG_SEED its caller set
            if getattr(dist_ctx, "_device_seed_base", None) is None:
                dist_ctx._device_seed_base = int(cfg.DEVICE_RNG_SEED)      # a second solver on this context
            cfg.DEVICE_RNG_SEED = dist_ctx.seed(dist_ctx._device_seed_base)   # does not add the rank again
        # data parallel: bucketed gradient all-reduce overlapped with backward
        self.overlap = dist_ctx.overlap(self.params) if (dist_ctx is not None and dist_ctx.enabled) else None

    def _adam(self):
        # plumbing: on the GPU ask for the multi-tensor ("fused") implementation -- the per-parameter
        # loop is ~7 small launches for each of ~160 parameters per step
        fused = bool(self.params) and all(p.is_cuda for p in self.params) and os.environ.get("WSSDL_ADAM_FUSED", "1") != "0"
        return torch.optim.Adam(self.params, lr=self.lr, eps=0.1, fused=True) if fused \
            else torch.optim.Adam(self.params, lr=self.lr, eps=0.1)

    def _apply(self, optimizer=None, count_step=True):
        if self.overlap is not None:
            self.overlap.finish()
        elif self.dist is not None:
            self.dist.allreduce_gradients(self.params)
        roi_pooling_op.poll_flags()               # deferred error flags of the RoI-pool pair: no read-back
        (optimizer or self.optimizer).step()      # parameters whose grad is None are skipped,
        self.optimizer.zero_grad(set_to_none=True)  # like apply_gradients with a None gradient
        if count_step:
            self.global_step += 1

    def check_flags(self):
        """Synchronising look at the deferred device flags (RoI-pool overflow, NMS time-out).  The poll in
        _apply runs one step late -- by the time it raises, the optimiser has already applied the step before
        it, so the parameters of that step are tainted -- and it never sees the last step: call this before
        saving parameters (the reference's snapshot, train_bus.py:66-101) and at the end of training."""
        roi_pooling_op.check_flags()

    def joint_backward(self, blobs):
        """Forward + backward of one combined mini-batch (train_bus.py:732-764): supervised images
        first, weak images after; the supervised and MIL gradients are summed per variable
        (:701-705), which is the gradient of (loss + mil_cross_entropy).  Gradients are left in
        .grad (and, under data parallelism, their all-reduce is in flight)."""
        n_s = int(cfg.TRAIN.IMS_PER_BATCH)
        n_ws = int(cfg.TRAIN.WS_IMS_PER_BATCH)
        layers = self.net(blobs['data'], blobs['im_info'], blobs['gt_boxes'], blobs['num_gt_boxes'],
                          is_training=True, is_ws=False)
        losses = supervised_loss(layers, self.net.weight_decay_params(), n_s)
        n_valid = layers['roi-data'][1].numel()                       # len(label), :624-628
        cls_ws = layers['cls_score'][n_valid:]
        batch_inds = layers['roi-data'][0][n_valid:, 0] - n_s         # :653
        mil_label = blobs['im_info'][n_s:, 3].to(torch.int32)         # image-level labels, :654
        funcs = [mil_core.get_mal_max_logit, mil_core.get_mal_max_logit]     # :655
        losses['mil_cross_entropy'] = mil_loss(cls_ws, batch_inds, mil_label, n_ws,
                                               self.global_step, funcs)
        (losses['loss'] + losses['mil_cross_entropy']).backward()
        return losses

    def train_step_joint(self, blobs):
        """One combined optimiser step: joint_backward, then Adam on the (averaged) gradients."""
        losses = self.joint_backward(blobs)
        self._apply()
        return losses

    def supervised_backward(self, blobs_s):
        """The supervised half of an alternating iteration (train_bus.py:334-360)."""
        layers = self.net(blobs_s['data'], blobs_s['im_info'], blobs_s['gt_boxes'],
                          blobs_s['num_gt_boxes'], is_training=True, is_ws=False)
        losses = supervised_loss(layers, self.net.weight_decay_params())
        losses['loss'].backward()
        return losses

    def weak_backward(self, blobs_ws):
        """The weak half (train_bus.py:362-394): the MIL loss alone, with is_ws=True."""
        layers = self.net(blobs_ws['data'], blobs_ws['im_info'], blobs_ws['gt_boxes'],
                          blobs_ws['num_gt_boxes'], is_training=True, is_ws=True)
        batch_inds = layers['roi-data'][0][:, 0]                      # :239
        mil_label = blobs_ws['im_info'][:, 3].to(torch.int32)         # :240
        funcs = [mil_core.get_mass_max_logit, mil_core.get_mal_max_logit]    # :241
        mil = mil_loss(layers['cls_score'], batch_inds, mil_label, blobs_ws['data'].shape[0],
                       self.global_step, funcs)
        mil.backward()
        return mil

    def train_step_alter(self, blobs_s, blobs_ws):
        """One alternating iteration (train_bus.py:334-394): a supervised step on `loss`, then
        a weak step on the MIL loss alone with is_ws=True."""
        losses = self.supervised_backward(blobs_s)
        self._apply(self.optimizer, count_step=False)              # train_op_s: no global_step
        losses['mil_cross_entropy'] = self.weak_backward(blobs_ws)
        if self.optimizer_ws is None:
            self.optimizer_ws = self._adam()
        self._apply(self.optimizer_ws, count_step=True)            # train_op_ws counts the step
        return losses
