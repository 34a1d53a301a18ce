"""Host-side checks of the per-RoI head's plumbing that run without a GPU: the stock-PyTorch
fallbacks (used on CPU tensors and when the plumbing library is absent) against torch's own
batch-norm and unfold."""
import numpy as np
import torch
import torch.nn.functional as F


def test_row_batchnorm_fallback_matches_batchnorm1d():
    from wssdl_bus_amd.networks.roi_head import RowBatchNorm
    g = torch.Generator().manual_seed(0)
    x = torch.randn((257, 48), generator=g) * 3 + 1
    dy = torch.randn((257, 48), generator=g)
    for relu in (False, True):
        a = RowBatchNorm(48)
        b = torch.nn.BatchNorm1d(48, eps=a.eps, momentum=a.momentum)
        with torch.no_grad():
            a.weight.uniform_(0.5, 1.5); a.bias.normal_()
            b.weight.copy_(a.weight); b.bias.copy_(a.bias)
        xa = x.clone().requires_grad_(True)
        xb = x.clone().requires_grad_(True)
        ya = a(xa, relu=relu)
        yb = b(xb)
        yb = torch.relu(yb) if relu else yb
        ya.backward(dy)
        yb.backward(dy)
        assert torch.allclose(ya, yb, atol=1e-5, rtol=1e-5)
        assert torch.allclose(xa.grad, xb.grad, atol=1e-5, rtol=1e-4)
        assert torch.allclose(a.weight.grad, b.weight.grad, atol=1e-4, rtol=1e-4)
        assert torch.allclose(a.bias.grad, b.bias.grad, atol=1e-4, rtol=1e-4)
        assert torch.allclose(a.running_mean, b.running_mean, atol=1e-6)
        assert torch.allclose(a.running_var, b.running_var, atol=1e-5)
        a.eval(); b.eval()
        with torch.no_grad():
            ye, yr = a(x, relu=relu), b(x)
            yr = torch.relu(yr) if relu else yr
        assert torch.allclose(ye, yr, atol=1e-5, rtol=1e-5)


def test_conv_nhwc_matches_conv2d():
    """ConvNHWC (GEMM on gathered patches, TF 'SAME' padding) against F.conv2d."""
    from wssdl_bus_amd.networks.backbones import _same_pad
    from wssdl_bus_amd.networks.roi_head import ConvNHWC
    g = torch.Generator().manual_seed(1)
    for (h, w, c_i, c_o, k, s) in ((7, 7, 8, 12, 3, 2), (4, 4, 8, 8, 3, 1), (7, 7, 8, 16, 1, 2), (5, 6, 4, 4, 3, 1)):
        conv = ConvNHWC(c_i, c_o, k, s, norm=None, relu=False)
        x = torch.randn((3, h, w, c_i), generator=g)
        y = conv(x)
        wt = conv.weight.view(c_o, k, k, c_i).permute(0, 3, 1, 2)          # (kh, kw, c_i) patch layout
        xn = x.permute(0, 3, 1, 2)
        if k > 1:
            pt, pb = _same_pad(h, k, s)
            pl, pr = _same_pad(w, k, s)
            xn = F.pad(xn, (pl, pr, pt, pb))
        ref = F.conv2d(xn, wt, conv.bias, stride=s).permute(0, 2, 3, 1)
        assert y.shape == ref.shape
        assert torch.allclose(y, ref, atol=1e-5, rtol=1e-5), (h, w, k, s)


def test_batchnorm_act2d_cpu_is_stock_batchnorm():
    from wssdl_bus_amd.networks.backbones import BatchNormAct2d
    g = torch.Generator().manual_seed(2)
    x = torch.randn((2, 6, 5, 7), generator=g)
    a = BatchNormAct2d(6, eps=1e-3, momentum=0.01)
    b = torch.nn.BatchNorm2d(6, eps=1e-3, momentum=0.01)
    assert sorted(a.state_dict()) == sorted(b.state_dict())        # checkpoint-compatible
    ya, yb = a(x, relu=True), torch.relu(b(x))
    assert torch.equal(ya, yb)
    assert np.isclose(float(a.running_var.sum()), float(b.running_var.sum()))
