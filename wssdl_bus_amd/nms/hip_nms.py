"""Greedy NMS on the GPU with the semantics of the reference's ``cpu_nms``
(code/lib/nms/cpu_nms.pyx:17-68): f32 box arithmetic, suppression when
``float64(iou) >= thresh``, visiting order = descending score."""
import torch

from .. import _lib


def hip_nms(dets, thresh, max_keep=None, rule="nms"):
    """dets [n,5] (x1,y1,x2,y2,score).  Returns kept indices in score order:
    a python list for numpy input (like cpu_nms), an int64 GPU tensor otherwise.
    rule "nms_new": the containment rule of utils/nms.pyx:70-123 on top of the IoU test."""
    as_np = _lib.wants_numpy(dets)
    d = _lib.to_device(dets, torch.float32)
    n = d.shape[0]
    if n == 0:
        return [] if as_np else torch.zeros((0,), dtype=torch.int64, device=d.device)
    if d.dim() != 2 or d.shape[1] != 5:
        raise ValueError("dets must be [n, 5]")
    mk = n if max_keep is None else min(int(max_keep), n)
    L = _lib.lib()
    with torch.cuda.device(d.device):
        ws = torch.empty((L.wssdl_nms_workspace_bytes(n),), dtype=torch.uint8, device=d.device)
        keep = torch.empty((max(mk, 1),), dtype=torch.int32, device=d.device)
        num = torch.zeros((1,), dtype=torch.int32, device=d.device)
        entry = {"nms": L.wssdl_nms, "nms_new": L.wssdl_nms_new}[rule]
        _lib.check(entry(_lib.ptr(d), n, float(thresh), mk, _lib.ptr(keep), _lib.ptr(num),
                         _lib.ptr(ws), ws.numel(), _lib.stream()), "wssdl_" + rule)
        k = int(num.item())
    keep = keep[:k]
    return [int(i) for i in keep.cpu().numpy()] if as_np else keep.to(torch.int64)
