"""PyTorch-ROCm backbones with the reference's wiring (plumbing: stock conv / BN / pool;
the detection hot path lives in the HIP library).

Reference: code/lib/networks/network.py:100-172 (conv / conv_int), :417-545 (ResNet blocks,
layer_group, normalization), Resnet_train_bus.py:55-63,91-101, VGGnet_train_bus.py:43-101.
Tensors are NCHW-shaped in channels_last memory format, so ``x.permute(0, 2, 3, 1)`` is the
contiguous NHWC view the hot-path ops take, with no copy.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class BatchNormAct2d(nn.BatchNorm2d):
    """nn.BatchNorm2d (same parameters and buffers) whose forward can also apply the ReLU that
    follows it.  On channels_last GPU tensors in training mode the layer runs on the fused row
    batch-norm kernels of the plumbing library ([N,H,W,C] is an [N*H*W, C] row matrix there);
    anywhere else it is the stock module followed by F.relu."""

    def forward(self, x, relu=False):
        if self.training and x.dim() == 4 and x.is_cuda and self.track_running_stats:
            rows = x.permute(0, 2, 3, 1)
            if rows.is_contiguous():
                from . import _plumbing
                from .roi_head import _FusedRowBatchNormFn
                n, h, w, c = rows.shape
                r2 = rows.reshape(-1, c)
                if _plumbing.usable(r2):
                    y, mean, var, _ = _FusedRowBatchNormFn.apply(r2, self.weight, self.bias, self.eps, bool(relu))
                    with torch.no_grad():
                        m = r2.shape[0]
                        mom = self.momentum if self.momentum is not None else 0.1
                        self.running_mean.lerp_(mean, mom)
                        self.running_var.lerp_(var * (m / max(m - 1, 1)), mom)
                        self.num_batches_tracked += 1
                    return y.view(n, h, w, c).permute(0, 3, 1, 2)
        y = super().forward(x)
        return F.relu(y) if relu else y


def _same_pad(size, k, s):
    """TF 'SAME' padding along one axis: (before, after)."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return total // 2, total - total // 2


class Conv(nn.Module):
    """conv + optional BatchNorm + optional ReLU (network.py:100-135).  TF 'SAME' padding
    is reproduced exactly (asymmetric when needed); bias only without normalisation."""

    def __init__(self, c_i, c_o, k, s, norm=None, relu=True, padding="SAME"):
        super().__init__()
        self.k, self.s, self.padding, self.relu = k, s, padding, relu
        self.conv = nn.Conv2d(c_i, c_o, k, s, 0, bias=(norm is None))
        nn.init.trunc_normal_(self.conv.weight, std=0.01, a=-0.02, b=0.02)     # :110
        if self.conv.bias is not None:
            nn.init.zeros_(self.conv.bias)
        self.bn = BatchNormAct2d(c_o, eps=1e-3, momentum=0.01) if norm == "BN" else None

    def forward(self, x):
        if self.padding == "SAME" and self.k > 1:
            pt, pb = _same_pad(x.shape[2], self.k, self.s)
            pl, pr = _same_pad(x.shape[3], self.k, self.s)
            x = F.pad(x, (pl, pr, pt, pb))
        x = self.conv(x)
        if self.bn is not None:
            return self.bn(x, relu=self.relu)
        return F.relu(x) if self.relu else x


class Bottleneck(nn.Module):
    """network.py:475-491: 1x1 -> 3x3 (stride here) -> 1x1(x4), pre-activation variants."""
    expansion = 4

    def __init__(self, c_i, c_o, s, preact, norm):
        super().__init__()
        self.preact = preact
        if preact != "no_preact":
            self.pre_bn = BatchNormAct2d(c_i, eps=1e-3, momentum=0.01) if norm == "BN" else None
        self.conv1 = Conv(c_i, c_o, 1, 1, norm)
        self.conv2 = Conv(c_o, c_o, 3, s, norm)
        self.conv3 = Conv(c_o, c_o * 4, 1, 1, norm, relu=False)
        self.short = Conv(c_i, c_o * 4, 1, s, norm, relu=False) if c_i != c_o * 4 else None

    def forward(self, x):
        ori = x
        if self.preact != "no_preact":
            y = self.pre_bn(x, relu=True) if self.pre_bn is not None else F.relu(x)
            if self.preact == "both_preact":
                ori = y
            x = y
        x = self.conv3(self.conv2(self.conv1(x)))
        return x + (self.short(ori) if self.short is not None else ori)


class BasicBlock(nn.Module):
    """network.py:457-473: 3x3 (stride here) -> 3x3."""
    expansion = 1

    def __init__(self, c_i, c_o, s, preact, norm):
        super().__init__()
        self.preact = preact
        if preact != "no_preact":
            self.pre_bn = BatchNormAct2d(c_i, eps=1e-3, momentum=0.01) if norm == "BN" else None
        self.conv1 = Conv(c_i, c_o, 3, s, norm)
        self.conv2 = Conv(c_o, c_o, 3, 1, norm, relu=False)
        self.short = Conv(c_i, c_o, 1, s, norm, relu=False) if c_i != c_o else None

    def forward(self, x):
        ori = x
        if self.preact != "no_preact":
            y = self.pre_bn(x, relu=True) if self.pre_bn is not None else F.relu(x)
            if self.preact == "both_preact":
                ori = y
            x = y
        x = self.conv2(self.conv1(x))
        return x + (self.short(ori) if self.short is not None else ori)


def layer_group(block, c_i, c_o, count, s, norm, first=False):
    """network.py:493-502."""
    blocks = [block(c_i, c_o, s, "no_preact" if first else "both_preact", norm)]
    for _ in range(1, count):
        blocks.append(block(c_o * block.expansion, c_o, 1, "default", norm))
    return nn.Sequential(*blocks)


RESNET_DEFS = {18: ([2, 2, 2, 2], BasicBlock), 34: ([3, 4, 6, 3], BasicBlock),
               50: ([3, 4, 6, 3], Bottleneck), 101: ([3, 4, 23, 3], Bottleneck)}   # Resnet_train_bus.py:32-37


class ResNetTrunk(nn.Module):
    """conv0 ... group2/relu (Resnet_train_bus.py:55-63): stride-16 feature map."""

    def __init__(self, depth, norm="BN"):
        super().__init__()
        defs, block = RESNET_DEFS[depth]
        e = block.expansion
        self.conv0 = Conv(3, 64, 7, 2, norm)
        self.group0 = layer_group(block, 64, 64, defs[0], 1, norm, first=True)
        self.group1 = layer_group(block, 64 * e, 128, defs[1], 2, norm)
        self.group2 = layer_group(block, 128 * e, 256, defs[2], 2, norm)
        self.norm = BatchNormAct2d(256 * e, eps=1e-3, momentum=0.01) if norm == "BN" else None
        self.out_channels = 256 * e

    def forward(self, x):
        x = self.conv0(x)
        x = F.max_pool2d(x, 3, 2)                         # 'VALID'
        x = self.group2(self.group1(self.group0(x)))
        return self.norm(x, relu=True) if self.norm is not None else F.relu(x)


class ResNetHead(nn.Module):
    """group3 -> norm -> relu -> global average pool (Resnet_train_bus.py:91-97)."""

    def __init__(self, depth, norm="BN"):
        super().__init__()
        defs, block = RESNET_DEFS[depth]
        e = block.expansion
        self.group3 = layer_group(block, 256 * e, 512, defs[3], 2, norm)
        self.norm = BatchNormAct2d(512 * e, eps=1e-3, momentum=0.01) if norm == "BN" else None
        self.out_features = 512 * e

    def forward(self, x):
        x = self.group3(x)
        x = self.norm(x, relu=True) if self.norm is not None else F.relu(x)
        return x.mean(dim=(2, 3))


class VGGTrunk(nn.Module):
    """conv1_1 ... conv5_3 (VGGnet_train_bus.py:44-61); conv1_x / conv2_x are frozen there."""

    def __init__(self):
        super().__init__()
        cfgs = [(3, 64), (64, 64), "P", (64, 128), (128, 128), "P", (128, 256), (256, 256), (256, 256),
                "P", (256, 512), (512, 512), (512, 512), "P", (512, 512), (512, 512), (512, 512)]
        layers = []
        n_conv = 0
        for c in cfgs:
            if c == "P":
                layers.append(nn.MaxPool2d(2, 2))         # 'VALID'
            else:
                conv = Conv(c[0], c[1], 3, 1, None)
                n_conv += 1
                if n_conv <= 4:                           # trainable=False for conv1_1..conv2_2
                    for p in conv.parameters():
                        p.requires_grad_(False)
                layers.append(conv)
        self.features = nn.Sequential(*layers)
        self.out_channels = 512

    def forward(self, x):
        return self.features(x)


class VGGHead(nn.Module):
    """fc6 -> drop6 -> fc7 -> drop7 (VGGnet_train_bus.py:91-96); fc flattens in (C,H,W)
    order (network.py:336)."""

    def __init__(self, keep_prob=0.5):
        super().__init__()
        self.fc6 = nn.Linear(512 * 7 * 7, 512)
        self.fc7 = nn.Linear(512, 512)
        for fc in (self.fc6, self.fc7):
            nn.init.trunc_normal_(fc.weight, std=0.01, a=-0.02, b=0.02)
            nn.init.zeros_(fc.bias)
        self.drop = nn.Dropout(1.0 - keep_prob)
        self.out_features = 512

    def forward(self, x):
        x = x.reshape(x.shape[0], -1)                     # NCHW-shaped input: (C,H,W) order
        x = self.drop(F.relu(self.fc6(x)))
        return self.drop(F.relu(self.fc7(x)))
