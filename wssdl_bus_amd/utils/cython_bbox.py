"""bbox_overlaps on the GPU (reference: code/lib/utils/bbox.pyx:15-55, imported there as
``utils.cython_bbox.bbox_overlaps``).  f64, bit-identical to the Cython kernel."""
import torch

from .. import _lib


def _run(fn_name, boxes, query_boxes):
    as_np = _lib.wants_numpy(boxes, query_boxes)
    b = _lib.to_device(boxes, torch.float64)
    q = _lib.to_device(query_boxes, torch.float64, b.device)
    if b.dim() != 2 or q.dim() != 2 or b.shape[1] < 4 or q.shape[1] < 4:
        raise ValueError("boxes and query_boxes must be 2-D with >= 4 columns")
    out = torch.zeros((b.shape[0], q.shape[0]), dtype=torch.float64, device=b.device)
    fn = getattr(_lib.lib(), fn_name)
    with torch.cuda.device(b.device):
        _lib.check(fn(_lib.ptr(b), b.shape[0], b.shape[1], _lib.ptr(q), q.shape[0], q.shape[1],
                      _lib.ptr(out), _lib.stream()), fn_name)
    return out.cpu().numpy() if as_np else out


def bbox_overlaps(boxes, query_boxes):
    """boxes (N, 4) float, query_boxes (K, 4) float -> (N, K) IoU with the +1 pixel
    convention.  numpy in -> numpy out; GPU tensors in -> GPU tensor out."""
    return _run("wssdl_bbox_overlaps", boxes, query_boxes)
