"""VGG-16 Faster-R-CNN training network (reference: code/lib/networks/VGGnet_train_bus.py:43-101
combined wiring, VGGnet_train_bus_alter.py alternating wiring)."""
import torch
import torch.nn as nn

from ..fast_rcnn.config import cfg

from .backbones import Conv, VGGHead, VGGTrunk
from .network import Network
from .Resnet_train_bus import _fc, _feat_stride, anchor_scales, n_classes


class VGGnet_train_bus(nn.Module, Network):
    def __init__(self, dataset='SNUBH', alter=False, keep_prob=0.5):
        nn.Module.__init__(self)
        Network.__init__(self)
        self.dataset, self.alter = dataset, alter
        self.trunk = VGGTrunk()
        A = len(anchor_scales) * 3
        self.rpn_conv = Conv(512, 512, 3, 1, None)
        self.rpn_cls_score = Conv(512, A * 2, 1, 1, None, relu=False, padding='VALID')
        self.rpn_bbox_pred = Conv(512, A * 4, 1, 1, None, relu=False, padding='VALID')
        self.head = VGGHead(keep_prob)
        self.cls_score = _fc(512, n_classes, 0.01)
        self.bbox_pred = _fc(512, n_classes * 4, 0.001)

    def forward(self, data, im_info, gt_boxes, num_gt_boxes, is_training=True, is_ws=False,
                test_net=False):
        self.layers = {'data': data, 'im_info': im_info, 'gt_boxes': gt_boxes,
                       'num_gt_boxes': num_gt_boxes, 'is_training': is_training, 'is_ws': is_ws}
        x = data.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
        feat = self.trunk(x)
        self.layers['conv5_3'] = feat.permute(0, 2, 3, 1)
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
            # VGGnet_test_bus.py wiring of the test network: proposals feed RoI pooling directly
            self.layers['roi-data'] = self.layers['rpn_rois']
        elif self.alter:
            (self.feed('rpn_rois', 'gt_boxes', 'num_gt_boxes')
                 .proposal_target_layer(n_classes, is_training, is_ws, name='roi-data'))
        else:
            (self.feed('rpn_rois', 'gt_boxes', 'num_gt_boxes')
                 .proposal_target_layer_joint(n_classes, is_training, name='roi-data'))
        (self.feed('conv5_3', 'roi-data')
             .roi_pool(7, 7, 1.0 / 16, name='pool_5'))
        pooled = self.layers['pool_5'].permute(0, 3, 1, 2)
        fc7 = self.head(pooled)
        self.layers['drop7'] = fc7
        self.layers['cls_score'] = self.cls_score(fc7)
        self.layers['cls_prob'] = torch.softmax(self.layers['cls_score'], dim=-1)
        self.layers['bbox_pred'] = self.bbox_pred(fc7)
        return self.layers

    def weight_decay_params(self):
        return [m.weight for m in self.modules() if isinstance(m, (nn.Conv2d, nn.Linear))
                and m.weight.requires_grad]
