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
