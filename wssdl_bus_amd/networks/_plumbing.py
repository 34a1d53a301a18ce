"""ctypes binding of ``libwssdl_plumbing_hip.so``: fused row batch-norm (+ReLU) kernels for the
per-RoI head (``csrc/plumbing/rowbn.hip``).  Plumbing around the hot path, not the drop-in C
ABI; when the library has not been built the head falls back to stock PyTorch ops (slower,
same maths) and says so once."""
import ctypes
import os
import warnings

import torch

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_HERE, "libwssdl_plumbing_hip.so")
_lib = None
_warned = [False]

_vp, _i, _ll, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_size_t


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            if not _warned[0]:
                _warned[0] = True
                warnings.warn("%s not built: the per-RoI head uses stock PyTorch batch-norm ops "
                              "(python -m wssdl_bus_amd.build builds it)" % LIB_PATH)
            return None
        L = ctypes.CDLL(LIB_PATH)
        L.wsplumb_rowbn_workspace_bytes.restype = _sz
        L.wsplumb_rowbn_workspace_bytes.argtypes = [_ll, _i]
        L.wsplumb_rowbn_supported.restype = _i
        L.wsplumb_rowbn_supported.argtypes = [_ll, _i]
        L.wsplumb_rowbn_forward.restype = _i
        L.wsplumb_rowbn_forward.argtypes = [_vp, _ll, _i, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                            _vp, _sz, _vp]
        L.wsplumb_rowbn_apply.restype = _i
        L.wsplumb_rowbn_apply.argtypes = [_vp, _ll, _i, _vp, _vp, _i, _vp, _vp]
        L.wsplumb_rowbn_backward.restype = _i
        L.wsplumb_rowbn_backward.argtypes = [_vp, _vp, _ll, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp,
                                             _vp, _vp, _sz, _vp]
        L.wsplumb_rowbn_forward_masked.restype = _i
        L.wsplumb_rowbn_forward_masked.argtypes = [_vp, _ll, _i, _vp, _vp, _f, _i, _vp, _i, _i, _vp, _vp, _vp, _vp,
                                                   _vp, _vp, _vp, _vp, _sz, _vp]
        L.wsplumb_rowbn_backward_masked.restype = _i
        L.wsplumb_rowbn_backward_masked.argtypes = [_vp, _vp, _ll, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i,
                                                    _vp, _vp, _vp, _vp, _vp, _sz, _vp]
        for name in ("wsplumb_im2col3x3", "wsplumb_col2im3x3"):
            f = getattr(L, name)
            f.restype = _i
            f.argtypes = [_vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]
        _lib = L
    return _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def usable(x):
    """True when the fused kernels can take this [M, C] tensor."""
    if os.environ.get("WSSDL_DISABLE_FUSED_BN"):           # A/B switch for measurements
        return False
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()):
        return False
    L = lib()
    return L is not None and bool(L.wsplumb_rowbn_supported(x.shape[0], x.shape[1]))


def _workspace(L, M, C, dev):
    n = L.wsplumb_rowbn_workspace_bytes(M, C)
    return torch.empty((n,), dtype=torch.uint8, device=dev), n


def rowbn_forward(x, weight, bias, eps, relu, mask=None):
    """mask: [n_rois] f32 on x's device (0 = dead RoI), x = [n_rois * per, C]; returns (y, stats, count)
    where count is None without a mask, else a [1] tensor holding the number of live rows."""
    L = lib()
    M, C = x.shape
    dev = x.device
    y = torch.empty_like(x)
    stats = torch.empty((5, C), dtype=torch.float32, device=dev)   # mean, var, rstd, scale, shift
    count = None
    with torch.cuda.device(dev):
        ws, n = _workspace(L, M, C, dev)
        if mask is None:
            rc = L.wsplumb_rowbn_forward(_p(x), M, C, _p(weight), _p(bias), float(eps), int(relu), _p(y),
                                         _p(stats[0]), _p(stats[1]), _p(stats[2]), _p(stats[3]),
                                         _p(stats[4]), _p(ws), n, _stream())
        else:
            n_rois = mask.shape[0]
            assert M % n_rois == 0 and mask.dtype == torch.float32 and mask.is_contiguous()
            count = torch.empty((1,), dtype=torch.float32, device=dev)
            rc = L.wsplumb_rowbn_forward_masked(_p(x), M, C, _p(weight), _p(bias), float(eps), int(relu), _p(mask),
                                                n_rois, M // n_rois, _p(y), _p(stats[0]), _p(stats[1]), _p(stats[2]),
                                                _p(stats[3]), _p(stats[4]), _p(count), _p(ws), n, _stream())
    if rc:
        raise RuntimeError("wsplumb_rowbn_forward failed (%d)" % rc)
    return y, stats, count


def rowbn_apply(x, scale, shift, relu):
    L = lib()
    M, C = x.shape
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = L.wsplumb_rowbn_apply(_p(x), M, C, _p(scale), _p(shift), int(relu), _p(y), _stream())
    if rc:
        raise RuntimeError("wsplumb_rowbn_apply failed (%d)" % rc)
    return y


def rowbn_backward(x, dy, weight, stats, relu, mask=None):
    L = lib()
    M, C = x.shape
    dev = x.device
    dx = torch.empty_like(x)
    dwb = torch.empty((2, C), dtype=torch.float32, device=dev)
    coef = torch.empty((3, C), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        ws, n = _workspace(L, M, C, dev)
        if mask is None:
            rc = L.wsplumb_rowbn_backward(_p(x), _p(dy), M, C, _p(weight), _p(stats[0]), _p(stats[2]),
                                          _p(stats[3]), _p(stats[4]), int(relu), _p(dx), _p(dwb[0]),
                                          _p(dwb[1]), _p(coef), _p(ws), n, _stream())
        else:
            n_rois = mask.shape[0]
            rc = L.wsplumb_rowbn_backward_masked(_p(x), _p(dy), M, C, _p(weight), _p(stats[0]), _p(stats[2]),
                                                 _p(stats[3]), _p(stats[4]), int(relu), _p(mask), n_rois,
                                                 M // n_rois, _p(dx), _p(dwb[0]), _p(dwb[1]), _p(coef), _p(ws), n,
                                                 _stream())
    if rc:
        raise RuntimeError("wsplumb_rowbn_backward failed (%d)" % rc)
    return dx, dwb[0], dwb[1]


def im2col_usable(x):
    """True when the 3x3 patch kernels can take this [R, h, w, C] tensor."""
    if os.environ.get("WSSDL_DISABLE_FUSED_IM2COL"):       # A/B switch for measurements
        return False
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
            and x.shape[3] % 4 == 0 and x.shape[0] > 0 and lib() is not None)


class Im2Col3x3Fn(torch.autograd.Function):
    """[R, h, w, C] -> [R*oh*ow, 9*C] patches (kh, kw, c), TF 'SAME' padding given by (pt, pl)."""

    @staticmethod
    def forward(ctx, x, stride, oh, ow, pt, pl):
        L = lib()
        r, h, w, c = x.shape
        cols = torch.empty((r * oh * ow, 9 * c), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = L.wsplumb_im2col3x3(_p(x), r, h, w, c, oh, ow, stride, pt, pl, _p(cols), _stream())
        if rc:
            raise RuntimeError("wsplumb_im2col3x3 failed (%d)" % rc)
        ctx.geom = (r, h, w, c, oh, ow, stride, pt, pl)
        return cols

    @staticmethod
    def backward(ctx, dcols):
        L = lib()
        r, h, w, c, oh, ow, stride, pt, pl = ctx.geom
        dcols = dcols.contiguous()
        dx = torch.empty((r, h, w, c), dtype=torch.float32, device=dcols.device)
        with torch.cuda.device(dcols.device):
            rc = L.wsplumb_col2im3x3(_p(dcols), r, h, w, c, oh, ow, stride, pt, pl, _p(dx), _stream())
        if rc:
            raise RuntimeError("wsplumb_col2im3x3 failed (%d)" % rc)
        return dx, None, None, None, None, None
