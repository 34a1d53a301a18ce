"""wssdl_bus_amd -- MI355X-native implementation of the wssdl_bus detection hot path.

Package layout mirrors the reference's ``code/lib`` so that its call sites map over
unchanged (``from rpn_msr.proposal_layer_tf_bus import proposal_layer`` becomes
``from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import proposal_layer``):

  csrc/ + libwssdl_bus_hip.so   hand-written HIP kernels behind a C ABI (include/wssdl_bus_hip.h)
  _lib.py                       ctypes binding (no fallback: raises if the .so is missing)
  rpn_msr/                      generate_anchors, anchor_target_layer[_ws|_joint],
                                proposal_layer, proposal_target_layer[_joint]
  utils/                        bbox_overlaps, bbox_overlaps_ui
  nms/, fast_rcnn/nms_wrapper   nms
  roi_pooling_layer/            roi_pool / roi_pool_grad (+ torch.autograd.Function)
  fast_rcnn/                    cfg, bbox_transform helpers, multi-task loss, train step
  networks/                     Network layer mirror (network.py:196-291) + PyTorch backbones
"""
__version__ = "0.1"
