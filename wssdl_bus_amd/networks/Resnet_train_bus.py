"""ResNet Faster-R-CNN training network, combined (joint) or alternating wiring.

Reference: code/lib/networks/Resnet_train_bus.py:55-101 (combined mode: ``*_joint`` layers);
the alternating wiring (``anchor_target_layer(..., is_ws)`` / ``proposal_target_layer(..., is_ws)``)
follows VGGnet_train_bus_alter.py:68,88.  Backbone convs are stock PyTorch-ROCm; the
hot-path nodes are the ``Network`` layer methods backed by the HIP library.
"""
import torch
import torch.nn as nn

from ..fast_rcnn.config import cfg

from .backbones import RESNET_DEFS, Conv, ResNetTrunk
from .roi_head import ConvNHWC, ResNetHeadNHWC
from .network import Network

n_classes = 3
_feat_stride = [16, ]
anchor_scales = [8, 16, 32]


def _fc(n_in, n_out, std):
    fc = nn.Linear(n_in, n_out)
    nn.init.trunc_normal_(fc.weight, std=std, a=-2 * std, b=2 * std)
    nn.init.zeros_(fc.bias)
    return fc


class Resnet_train_bus(nn.Module, Network):
    def __init__(self, net_depth, dataset='SNUBH', norm_type='BN', alter=False):
        nn.Module.__init__(self)
        Network.__init__(self)
        self.net_depth, self.dataset, self.norm_type, self.alter = net_depth, dataset, norm_type, alter
        block = RESNET_DEFS[net_depth][1]
        self.trunk = ResNetTrunk(net_depth, norm_type)
        c = self.trunk.out_channels
        A = len(anchor_scales) * 3
        self.rpn_conv = Conv(c, 256 * block.expansion, 3, 1, norm_type)          # 'rpn_conv/3x3'
        self.rpn_cls_score = Conv(256 * block.expansion, A * 2, 1, 1, None, relu=False, padding='VALID')
        self.rpn_bbox_pred = Conv(256 * block.expansion, A * 4, 1, 1, None, relu=False, padding='VALID')
        self.head = ResNetHeadNHWC(net_depth, norm_type)
        self.cls_score = _fc(self.head.out_features, n_classes, 0.01)
        self.bbox_pred = _fc(self.head.out_features, n_classes * 4, 0.001)

    def forward(self, data, im_info, gt_boxes, num_gt_boxes, is_training=True, is_ws=False,
                test_net=False):
        """data [N,H,W,3] NHWC f32 (like the reference's placeholder); returns self.layers."""
        self.layers = {'data': data, 'im_info': im_info, 'gt_boxes': gt_boxes,
                       'num_gt_boxes': num_gt_boxes, 'is_training': is_training, 'is_ws': is_ws}
        x = data.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
        feat = self.trunk(x)                                          # NCHW-shaped, channels_last
        self.layers['group2/relu'] = feat.permute(0, 2, 3, 1)         # NHWC view, no copy
        rpn = self.rpn_conv(feat)
        self.layers['rpn_conv/3x3'] = rpn.permute(0, 2, 3, 1)
        self.layers['rpn_cls_score'] = self.rpn_cls_score(rpn).permute(0, 2, 3, 1).contiguous()
        self.layers['rpn_bbox_pred'] = self.rpn_bbox_pred(rpn).permute(0, 2, 3, 1).contiguous()

        if not test_net:
            if self.alter:
                (self.feed('rpn_cls_score', 'gt_boxes', 'num_gt_boxes', 'im_info', 'data')
                     .anchor_target_layer(_feat_stride, anchor_scales, self.dataset, is_ws, name='rpn-data'))
            else:
                (self.feed('rpn_cls_score', 'gt_boxes', 'num_gt_boxes', 'im_info', 'data')
                     .anchor_target_layer_joint(_feat_stride, anchor_scales, self.dataset, is_training,
                                                name='rpn-data'))
        (self.feed('rpn_cls_score')
             .reshape_layer(2, name='rpn_cls_score_reshape'))
        if cfg.FUSED_RPN_SOFTMAX:
            (self.feed('rpn_cls_score', 'rpn_bbox_pred', 'im_info')
                 .proposal_layer_from_score(_feat_stride, anchor_scales, is_training, is_ws, name='rpn_rois'))
        else:
            (self.feed('rpn_cls_score_reshape')
                 .softmax(name='rpn_cls_prob'))
            (self.feed('rpn_cls_prob')
                 .reshape_layer(len(anchor_scales) * 3 * 2, name='rpn_cls_prob_reshape'))
            (self.feed('rpn_cls_prob_reshape', 'rpn_bbox_pred', 'im_info')
                 .proposal_layer(_feat_stride, anchor_scales, is_training, is_ws, name='rpn_rois'))
        if test_net:
            # Resnet_test_bus.py:60-64 wiring of the test network: proposals feed RoI pooling directly
            self.layers['roi-data'] = self.layers['rpn_rois']
        elif self.alter:
            (self.feed('rpn_rois', 'gt_boxes', 'num_gt_boxes')
                 .proposal_target_layer(n_classes, is_training, is_ws, name='roi-data'))
        else:
            (self.feed('rpn_rois', 'gt_boxes', 'num_gt_boxes')
                 .proposal_target_layer_joint(n_classes, is_training, name='roi-data'))
        (self.feed('group2/relu', 'roi-data')
             .roi_pool(7, 7, 1.0 / 16, name='roi_pool'))
        if (cfg.PADDED_ROIS or cfg.SAMPLING_RNG == 'device') and self.training and not test_net:
            # rows with batch index -1 are dead (padding of the fixed-shape RoI blob, or of a supervised
            # image the device sampler found short of candidates): the head's batch statistics are taken
            # over the live rows only, as if the blob had been compacted
            from . import roi_head
            rois_in = self.layers['roi-data'][0] if isinstance(self.layers['roi-data'], tuple) else self.layers['roi-data']
            roi_head.set_roi_mask((rois_in[:, 0] >= 0).to(torch.float32))
            try:
                gap = self.head(self.layers['roi_pool'])
            finally:
                roi_head.set_roi_mask(None)
        else:
            gap = self.head(self.layers['roi_pool'])                  # [R,7,7,C] NHWC, GEMM head
        self.layers['gap'] = gap
        self.layers['cls_score'] = self.cls_score(gap)
        self.layers['cls_prob'] = torch.softmax(self.layers['cls_score'], dim=-1)
        self.layers['bbox_pred'] = self.bbox_pred(gap)
        return self.layers

    def weight_decay_params(self):
        """Variables named '*weights' in the reference (conv / fc kernels), train_bus.py:268-270."""
        return [m.weight for m in self.modules() if isinstance(m, (nn.Conv2d, nn.Linear, ConvNHWC))
                and m.weight.requires_grad]
