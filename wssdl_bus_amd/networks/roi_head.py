"""Per-RoI ResNet head (group3 ... gap) as NHWC GEMMs.

Reference wiring: code/lib/networks/Resnet_train_bus.py:91-97 (group3 -> norm -> relu -> gap),
blocks network.py:457-491.  The number of RoIs changes from step to step (NMS keeps a
data-dependent count), and MIOpen re-tunes / re-compiles convolution kernels for every new
batch dimension (tens of seconds per new R on this stack), so the head does not use
convolution kernels at all: RoI-pool output is already [R,7,7,C] NHWC, 1x1 convs are
F.linear on [R*h*w, C] rows, 3x3 convs gather their TF-'SAME' patches and run one GEMM, and
normalisation is batch-norm over rows.  GEMMs (rocBLAS/hipBLASLt) are shape-agnostic.
Plumbing, not the product: stock PyTorch ops, except that batch-norm (+ReLU) runs on the
fused kernels of csrc/plumbing/rowbn.hip when that library is built (the activations are
[R*h*w, C] with up to ~4e5 rows: separate elementwise passes over them were a third of the
step).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _plumbing
from .backbones import RESNET_DEFS, _same_pad


class _RowBatchNormFn(torch.autograd.Function):
    """Training-mode batch norm over the rows of [M, C] built from column reductions
    (`var_mean`, `sum`) and fused elementwise ops (stock PyTorch; used on the CPU and when the
    plumbing library is not built).  PyTorch's native channels-last batch-norm kernels take
    12 ms forward+backward on a [136k, 2048] f32 tensor on MI355X; this form about 2.5 ms."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        var, mean = torch.var_mean(x, dim=0, unbiased=False)
        rstd = torch.rsqrt(var + eps)
        scale = rstd * weight
        y = torch.addcmul(bias - mean * scale, x, scale)
        ctx.save_for_backward(x, mean, rstd, weight)
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dmean, _dvar):
        x, mean, rstd, weight = ctx.saved_tensors
        m = x.shape[0]
        sum_dy = dy.sum(0)
        sum_dy_x = (dy * x).sum(0)
        sum_dy_xhat = (sum_dy_x - mean * sum_dy) * rstd
        # dx = w*rstd * (dy - mean(dy) - xhat * mean(dy*xhat)),  xhat = (x - mean) * rstd
        a = weight * rstd
        k1 = a * rstd * sum_dy_xhat / m                  # multiplies (x - mean)
        k0 = a * sum_dy / m - k1 * mean                  # constant per column: a*mean(dy) - k1*mean
        dx = torch.addcmul(-k0, dy, a)
        dx.addcmul_(x, -k1)
        return dx, sum_dy_xhat, sum_dy, None


class _FusedRowBatchNormFn(torch.autograd.Function):
    """The same layer (optionally with its ReLU) on the fused HIP kernels of
    csrc/plumbing/rowbn.hip: 3 passes over the tensor forward, 5 backward, instead of 5 + 14
    with separate elementwise ops; the ReLU mask is recomputed from x in the backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, relu, roi_mask=None):
        y, stats, count = _plumbing.rowbn_forward(x, weight, bias, eps, relu, roi_mask)
        ctx.masked = roi_mask is not None
        if ctx.masked:
            ctx.save_for_backward(x, weight, stats, roi_mask)
        else:
            ctx.save_for_backward(x, weight, stats)
            count = stats[0, :1]                       # placeholder (unused without a mask): no extra launch
        ctx.relu = relu
        mean, var = stats[0], stats[1]
        ctx.mark_non_differentiable(mean, var, count)
        return y, mean, var, count

    @staticmethod
    def backward(ctx, dy, _dmean, _dvar, _dcount):
        if ctx.masked:
            x, weight, stats, roi_mask = ctx.saved_tensors
        else:
            (x, weight, stats), roi_mask = ctx.saved_tensors, None
        dx, dw, db = _plumbing.rowbn_backward(x, dy.contiguous(), weight, stats, ctx.relu, roi_mask)
        return dx, dw, db, None, None, None


# The head can see RoI rows that are not live: the padding rows of the fixed-shape blob
# (cfg.PADDED_ROIS) and those of a supervised image that ran short of candidates under the device
# sampler (cfg.SAMPLING_RNG = 'device': the layer keeps its fixed S*128 rows, batch index -1).
# The networks set this mask ([R] f32, 1 = live) around the head call; batch statistics are
# taken over the live rows only and dead rows are zeroed after every normalisation, so that the
# live rows come out as if the blob had been compacted.  On the GPU this is the masked form of the
# fused kernels (csrc/plumbing/rowbn.hip: dead rows are not even read); elsewhere plain PyTorch ops.
# No host sync either way.
_ROI_MASK = None


def set_roi_mask(mask):
    global _ROI_MASK
    _ROI_MASK = mask


def _masked_row_batch_norm(x, weight, bias, eps, relu, roi_mask):
    M = x.shape[0]
    per = M // roi_mask.shape[0]
    m = roi_mask.repeat_interleave(per).unsqueeze(1)
    n = (roi_mask.sum() * per).clamp_min(1.0)
    mean = (x * m).sum(0) / n
    d = (x - mean) * m
    var = (d * d).sum(0) / n
    y = d * (torch.rsqrt(var + eps) * weight) + bias
    if relu:
        y = F.relu(y)
    return y * m, mean.detach(), var.detach(), n


class RowBatchNorm(nn.Module):
    """BatchNorm over rows ([M, C] input) with the usual running statistics; `relu=True`
    applies the ReLU that follows it in the network inside the same kernels."""

    def __init__(self, num_features, eps=1e-3, momentum=0.01):
        super().__init__()
        self.eps, self.momentum = eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))

    def forward(self, x, relu=False):
        fused = _plumbing.usable(x)
        if not self.training:
            scale = self.weight * torch.rsqrt(self.running_var + self.eps)
            shift = self.bias - self.running_mean * scale
            if fused and not torch.is_grad_enabled():
                return _plumbing.rowbn_apply(x, scale.contiguous(), shift.contiguous(), relu)
            y = torch.addcmul(shift, x, scale)
            return F.relu(y) if relu else y
        if _ROI_MASK is not None:
            if fused and x.shape[0] % _ROI_MASK.shape[0] == 0:
                y, mean, var, n = _FusedRowBatchNormFn.apply(x, self.weight, self.bias, self.eps, bool(relu), _ROI_MASK)
                n = n[0]
            else:
                y, mean, var, n = _masked_row_batch_norm(x, self.weight, self.bias, self.eps, relu, _ROI_MASK)
            with torch.no_grad():
                self.running_mean.lerp_(mean, self.momentum)
                self.running_var.lerp_(var * (n / (n - 1).clamp_min(1.0)), self.momentum)
            return y
        if fused:
            y, mean, var, _ = _FusedRowBatchNormFn.apply(x, self.weight, self.bias, self.eps, bool(relu))
        else:
            y, mean, var = _RowBatchNormFn.apply(x, self.weight, self.bias, self.eps)
            if relu:
                y = F.relu(y)
        with torch.no_grad():
            m = x.shape[0]
            self.running_mean.lerp_(mean, self.momentum)
            self.running_var.lerp_(var * (m / max(m - 1, 1)), self.momentum)
        return y


class ConvNHWC(nn.Module):
    """conv (TF 'SAME') + optional BN + optional ReLU on NHWC tensors via one GEMM.
    weight [c_o, k*k*c_i] with the patch laid out (kh, kw, c_i)."""

    def __init__(self, c_i, c_o, k, s, norm=None, relu=True):
        super().__init__()
        self.c_i, self.c_o, self.k, self.s, self.relu = c_i, c_o, k, s, relu
        self.weight = nn.Parameter(torch.empty(c_o, k * k * c_i))
        nn.init.trunc_normal_(self.weight, std=0.01, a=-0.02, b=0.02)
        self.bias = nn.Parameter(torch.zeros(c_o)) if norm is None else None
        self.bn = RowBatchNorm(c_o) if norm == "BN" else None

    def forward(self, x):
        r, h, w, c = x.shape
        k, s = self.k, self.s
        if k == 1:
            if s > 1:
                x = x[:, ::s, ::s, :]
            oh, ow = x.shape[1], x.shape[2]
            rows = x.reshape(-1, c)
        elif k == 3 and _plumbing.im2col_usable(x):
            pt, pb = _same_pad(h, k, s)
            pl, pr = _same_pad(w, k, s)
            oh, ow = -(-h // s), -(-w // s)
            rows = _plumbing.Im2Col3x3Fn.apply(x, s, oh, ow, pt, pl)
        else:
            pt, pb = _same_pad(h, k, s)
            pl, pr = _same_pad(w, k, s)
            xp = F.pad(x, (0, 0, pl, pr, pt, pb))
            p = xp.unfold(1, k, s).unfold(2, k, s)            # [R, oh, ow, C, kh, kw]
            oh, ow = p.shape[1], p.shape[2]
            rows = p.permute(0, 1, 2, 4, 5, 3).reshape(-1, k * k * c)
        y = F.linear(rows, self.weight, self.bias)
        if self.bn is not None:
            y = self.bn(y, relu=self.relu)
        elif self.relu:
            y = F.relu(y)
        return y.view(r, oh, ow, self.c_o)


def _bn_rows(bn, x, relu=False):
    r, h, w, c = x.shape
    return bn(x.reshape(-1, c), relu=relu).view(r, h, w, c)


class BottleneckNHWC(nn.Module):
    expansion = 4

    def __init__(self, c_i, c_o, s, preact, norm):
        super().__init__()
        self.preact = preact
        self.pre_bn = RowBatchNorm(c_i) if (preact != "no_preact" and norm == "BN") else None
        self.conv1 = ConvNHWC(c_i, c_o, 1, 1, norm)
        self.conv2 = ConvNHWC(c_o, c_o, 3, s, norm)
        self.conv3 = ConvNHWC(c_o, c_o * 4, 1, 1, norm, relu=False)
        self.short = ConvNHWC(c_i, c_o * 4, 1, s, norm, relu=False) if c_i != c_o * 4 else None

    def forward(self, x):
        ori = x
        if self.preact != "no_preact":
            y = _bn_rows(self.pre_bn, x, relu=True) if self.pre_bn is not None else F.relu(x)
            if self.preact == "both_preact":
                ori = y
            x = y
        x = self.conv3(self.conv2(self.conv1(x)))
        return x + (self.short(ori) if self.short is not None else ori)


class BasicBlockNHWC(nn.Module):
    expansion = 1

    def __init__(self, c_i, c_o, s, preact, norm):
        super().__init__()
        self.preact = preact
        self.pre_bn = RowBatchNorm(c_i) if (preact != "no_preact" and norm == "BN") else None
        self.conv1 = ConvNHWC(c_i, c_o, 3, s, norm)
        self.conv2 = ConvNHWC(c_o, c_o, 3, 1, norm, relu=False)
        self.short = ConvNHWC(c_i, c_o, 1, s, norm, relu=False) if c_i != c_o else None

    def forward(self, x):
        ori = x
        if self.preact != "no_preact":
            y = _bn_rows(self.pre_bn, x, relu=True) if self.pre_bn is not None else F.relu(x)
            if self.preact == "both_preact":
                ori = y
            x = y
        x = self.conv2(self.conv1(x))
        return x + (self.short(ori) if self.short is not None else ori)


class ResNetHeadNHWC(nn.Module):
    """[R,7,7,C] NHWC -> [R, 512*expansion]."""

    def __init__(self, depth, norm="BN"):
        super().__init__()
        defs, block = RESNET_DEFS[depth]
        blk = BottleneckNHWC if block.expansion == 4 else BasicBlockNHWC
        e = blk.expansion
        blocks = [blk(256 * e, 512, 2, "both_preact", norm)]
        for _ in range(1, defs[3]):
            blocks.append(blk(512 * e, 512, 1, "default", norm))
        self.group3 = nn.Sequential(*blocks)
        self.norm = RowBatchNorm(512 * e) if norm == "BN" else None
        self.out_features = 512 * e

    def forward(self, x):
        x = self.group3(x)
        x = _bn_rows(self.norm, x, relu=True) if self.norm is not None else F.relu(x)
        return x.mean(dim=(1, 2))
