"""``utils.cython_nms`` of the reference (code/lib/utils/nms.pyx, imported at fast_rcnn/test_bus.py:10) on the GPU.

``nms`` (:17-68) is the greedy rule of ``cpu_nms.pyx`` in a second file; ``nms_new`` (:70-123) adds the
containment terms ``inter / area_i > 0.95 or inter / area_j > 0.95``.  Both return the kept indices in
score order like the Cython functions (a list for NumPy input)."""
from ..nms.hip_nms import hip_nms


def nms(dets, thresh):
    if dets.shape[0] == 0:
        return []
    return hip_nms(dets, thresh)


def nms_new(dets, thresh):
    if dets.shape[0] == 0:
        return []
    return hip_nms(dets, thresh, rule="nms_new")
