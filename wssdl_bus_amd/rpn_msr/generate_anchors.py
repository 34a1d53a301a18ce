"""generate_anchors (reference: code/lib/rpn_msr/generate_anchors.py:37-97).

The nine base anchors are computed by the host-side C entry point
``wssdl_generate_anchors_host``; the shifted anchor grid is rebuilt inside the HIP
kernels from these values and never materialised on the hot path.
"""
import numpy as np

from .. import _lib


def generate_anchors(base_size=16, ratios=[0.5, 1, 2], scales=2 ** np.arange(3, 6)):
    """Anchor windows for every (ratio, scale) around a (0, 0, base-1, base-1) box.
    Returns [len(ratios)*len(scales), 4] float64, ratio-major like the reference."""
    return _lib.generate_anchors_host(base_size, ratios, scales)


def shifted_anchors(height, width, feat_stride, anchors):
    """All K*A shifted anchors, [K*A, 4] f64 on the GPU, row (h*W+w)*A+a
    (anchor_target_layer_tf_bus.py:59-73).  Debug / API-parity helper."""
    import torch
    base = np.ascontiguousarray(anchors, dtype=np.float64)
    A = base.shape[0]
    out = torch.empty((height * width * A, 4), dtype=torch.float64, device="cuda")
    stride = int(np.asarray(feat_stride).ravel()[0])
    _lib.check(_lib.lib().wssdl_shifted_anchors(_lib.host_ptr(base), A, height, width, stride,
                                                _lib.ptr(out), _lib.stream()),
               "wssdl_shifted_anchors")
    return out
