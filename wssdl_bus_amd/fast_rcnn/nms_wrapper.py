"""NMS dispatcher (reference: code/lib/fast_rcnn/nms_wrapper.py:13-21)."""
import numpy as np

from .config import cfg
from ..nms.hip_nms import hip_nms


def gpu_rule_threshold(thresh):
    """The reference's CUDA NMS (nms/nms_kernel.cu:24-32,71, reached through gpu_nms when cfg.USE_GPU_NMS) suppresses
    box j when ``devIoU(i, j) > nms_overlap_thresh`` with BOTH sides f32; its IoU arithmetic is the f32 arithmetic of
    cpu_nms.pyx:57-64.  The HIP kernel compares ``(double)iou >= t`` (the vendored cpu_nms rule), and for
    an f32 iou  ``iou > (float)thresh``  <=>  ``(double)iou >= nextafter((double)(float)thresh, +inf)``:  the CUDA rule is
    the CPU rule at that threshold.  (At thresh = 0.7 the two rules agree anyway, since (double)0.7f < 0.7; at 0.5 they
    differ on a pair whose IoU is exactly 0.5.)"""
    return float(np.nextafter(np.float64(np.float32(thresh)), np.inf))


def nms(dets, thresh, force_cpu=False):
    """Same call as the reference's dispatcher: the cpu_nms rule, or -- cfg.USE_GPU_NMS and not force_cpu
    (nms_wrapper.py:18-19) -- the CUDA kernel's `>` / f32-threshold rule.  Both run the HIP kernel."""
    if dets.shape[0] == 0:
        return []
    if cfg.USE_GPU_NMS and not force_cpu:
        return hip_nms(dets, gpu_rule_threshold(thresh))
    return hip_nms(dets, thresh)
