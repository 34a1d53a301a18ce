"""Multiple-instance (image-level) bag logits for the weakly supervised loss.

Reference: code/lib/mil/core.py:11-96.  Instances (RoIs) arrive grouped by image in
batch order; for each bag (image) one instance row is selected and its full logit row
becomes the bag logit.  Tiny host-side tensor ops in PyTorch.
"""
import torch


def get_mal_max_logit(cur_logits):
    """Row of the instance with the largest malignant (class 2) logit (core.py:60-69)."""
    return cur_logits[torch.argmax(cur_logits[:, 2])].unsqueeze(0)


def get_ben_max_logit(cur_logits):
    """Row of the instance with the largest benign (class 1) logit (core.py:49-57)."""
    return cur_logits[torch.argmax(cur_logits[:, 1])].unsqueeze(0)


def get_mass_max_logit(cur_logits):
    """Row of the instance with the smallest background (class 0) logit (core.py:88-96)."""
    return cur_logits[torch.argmin(cur_logits[:, 0])].unsqueeze(0)


def get_disc_max_logit(cur_logits):
    """Row of the instance with the largest non-background logit (core.py:78-86)."""
    return cur_logits[torch.argmax(cur_logits[:, 1:].max(dim=1).values)].unsqueeze(0)


def get_bag_logit(instance_logits, batch_inds, num_classes, bag_labels, batch_size, funcs,
                  counts_host=None):
    """core.py:11-46.  instance_logits [R, num_classes]; batch_inds [R] or [R,1] bag index
    of every instance (0-based, grouped and ascending); bag_labels [batch_size] int.
    funcs[0] is used for bags labelled 1, funcs[1] otherwise (:40-42).
    Returns (bag_logits [batch_size, num_classes], scale_factors [batch_size]).
    `counts_host` (instances per bag) avoids a device->host copy when the caller knows it."""
    if counts_host is None:
        b = batch_inds.reshape(-1).to(torch.int64)
        counts_host = torch.bincount(b, minlength=batch_size)[:batch_size].cpu().tolist()
    labels_host = bag_labels.reshape(-1).cpu().tolist() if isinstance(bag_labels, torch.Tensor) \
        else [int(x) for x in bag_labels]
    rows, scales = [], []
    start = 0
    for i in range(batch_size):
        n = int(counts_host[i])
        cur = instance_logits[start:start + n]
        start += n
        f = funcs[0] if int(labels_host[i]) == 1 else funcs[1]
        row = f(cur)
        rows.append(row)
        scales.append(torch.softmax(row, dim=1)[0, int(labels_host[i])])
    return torch.cat(rows, dim=0), torch.stack(scales)


# ------------------------------------------------------------------ device op (f1) ---
_SELECTORS = {get_mal_max_logit: 0, get_ben_max_logit: 1, get_mass_max_logit: 2}


def get_bag_logit_device(instance_logits, bag_column, bag_offset, bag_labels, batch_size, funcs,
                         return_valid=False):
    """get_bag_logit as ONE kernel launch + one gather, with no host round trip: the HIP op
    ``wssdl_mil_select`` picks each bag's instance row, ``index_select`` (differentiable) gathers
    the logit rows.  `bag_column` is the column holding each instance's bag index plus
    `bag_offset` (a strided view such as ``rois[n_valid:, 0]`` is fine).  Returns
    (bag_logits [batch_size, K], scale_factors [batch_size]) like get_bag_logit.

    A bag without instances (a weak image whose proposals were all filtered out) makes the
    reference's tf.arg_max fail on the host.  Here the op reports row -1 for it; the gather then
    uses row 0 and the bag's logits are zeroed, and with `return_valid` the [batch_size] bool
    mask of the non-empty bags is returned as a third value so that the loss can give such a
    bag zero weight -- all without a device->host copy.  No instances at all raises ValueError."""
    from .. import _lib
    logits = instance_logits.contiguous()
    R, K = logits.shape
    if R == 0 and batch_size > 0:
        raise ValueError("get_bag_logit: no instances for %d bag(s)" % batch_size)
    col = bag_column if bag_column.dtype == torch.float32 else bag_column.to(torch.float32)
    stride = col.stride(0) if col.dim() == 1 and R > 1 else 1
    if col.dim() != 1 or (R > 1 and stride < 1):
        col = col.reshape(-1).contiguous()
        stride = 1
    labels = bag_labels.reshape(-1).to(torch.int32).contiguous()
    rows = torch.empty((batch_size,), dtype=torch.int32, device=logits.device)
    with torch.cuda.device(logits.device):
        _lib.check(_lib.lib().wssdl_mil_select(
            _lib.ptr(logits), R, K, _lib.ptr(col), int(stride), float(bag_offset), _lib.ptr(labels),
            int(batch_size), _SELECTORS[funcs[0]], _SELECTORS[funcs[1]], _lib.ptr(rows), None,
            _lib.stream()), "wssdl_mil_select")
    valid = rows >= 0
    bag_logits = instance_logits.index_select(0, rows.clamp_min(0).to(torch.int64))
    bag_logits = bag_logits * valid.unsqueeze(1).to(bag_logits.dtype)
    scale = torch.softmax(bag_logits, dim=1).gather(1, labels.to(torch.int64).unsqueeze(1)).squeeze(1)
    if return_valid:
        return bag_logits, scale, valid
    return bag_logits, scale


class _MilLoss(torch.autograd.Function):
    """The MIL term as one device op (csrc/mil.hip): selection + weighted CE + mean, and its backward."""

    @staticmethod
    def forward(ctx, instance_logits, bag_column, bag_offset, bag_labels, n_bags, sel1, sel_other, class_weights, scale):
        import numpy as np
        from .. import _lib
        logits = instance_logits.contiguous()
        R, K = logits.shape
        col = bag_column if bag_column.dtype == torch.float32 else bag_column.to(torch.float32)
        stride = col.stride(0) if col.dim() == 1 and R > 1 else 1
        if col.dim() != 1 or (R > 1 and stride < 1):
            col = col.reshape(-1).contiguous()
            stride = 1
        labels = bag_labels.reshape(-1).to(torch.int32).contiguous()
        dev = logits.device
        rows = torch.empty((n_bags,), dtype=torch.int32, device=dev)
        bag_loss = torch.empty((n_bags,), dtype=torch.float32, device=dev)
        loss = torch.empty((1,), dtype=torch.float32, device=dev)
        cw = np.ascontiguousarray(np.asarray(class_weights, np.float32))
        with torch.cuda.device(dev), _lib.timed("mil_loss", dict(R=R, bags=int(n_bags))):
            _lib.check(_lib.lib().wssdl_mil_loss_forward(
                _lib.ptr(logits), R, K, _lib.ptr(col), int(stride), float(bag_offset), _lib.ptr(labels), int(n_bags),
                int(sel1), int(sel_other), _lib.host_ptr(cw), float(scale), _lib.ptr(loss), _lib.ptr(rows),
                _lib.ptr(bag_loss), _lib.stream()), "wssdl_mil_loss_forward")
        ctx.save_for_backward(logits, col, labels, rows)
        ctx.misc = (R, K, int(stride), float(bag_offset), int(n_bags), cw, float(scale))
        return loss[0]

    @staticmethod
    def backward(ctx, grad_loss):
        from .. import _lib
        logits, col, labels, rows = ctx.saved_tensors
        R, K, stride, bag_offset, n_bags, cw, scale = ctx.misc
        g = torch.empty_like(logits)
        gl = grad_loss.reshape(1).to(torch.float32).contiguous()
        with torch.cuda.device(logits.device), _lib.timed("mil_loss_backward", dict(R=R, bags=n_bags)):
            _lib.check(_lib.lib().wssdl_mil_loss_backward(
                _lib.ptr(logits), R, K, _lib.ptr(col), stride, bag_offset, _lib.ptr(labels), n_bags, _lib.ptr(rows),
                _lib.host_ptr(cw), scale, _lib.ptr(gl), _lib.ptr(g), _lib.stream()), "wssdl_mil_loss_backward")
        return (g,) + (None,) * 8


def mil_loss_device(instance_logits, bag_column, bag_offset, bag_labels, n_bags, funcs, class_weights, scale):
    """train_bus.py:239-260 / :650-671 on the device: mean over the bags of
    scale * class_weights[label] * CE(selected instance's logits, label); an empty bag contributes 0.
    No instances at all raises ValueError like get_bag_logit_device."""
    if instance_logits.shape[0] == 0 and n_bags > 0:
        raise ValueError("get_bag_logit: no instances for %d bag(s)" % n_bags)
    return _MilLoss.apply(instance_logits, bag_column, bag_offset, bag_labels, int(n_bags), _SELECTORS[funcs[0]],
                          _SELECTORS[funcs[1]], class_weights, scale)
