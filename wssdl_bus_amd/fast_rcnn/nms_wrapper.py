"""NMS dispatcher (reference: code/lib/fast_rcnn/nms_wrapper.py:13-21)."""
from ..nms.hip_nms import hip_nms


def nms(dets, thresh, force_cpu=False):
    """Same call as the reference's dispatcher.  `force_cpu` is accepted for
    signature parity and ignored: the HIP kernel implements the cpu_nms rule, and
    this package has no CPU path."""
    if dets.shape[0] == 0:
        return []
    return hip_nms(dets, thresh)
