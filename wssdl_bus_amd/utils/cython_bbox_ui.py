"""bbox_overlaps_ui on the GPU (reference: code/lib/utils/bbox_ui.pyx:12-47):
intersection area divided by the area of ``boxes[n]``."""
from .cython_bbox import _run


def bbox_overlaps_ui(boxes, query_boxes):
    return _run("wssdl_bbox_overlaps_ui", boxes, query_boxes)
