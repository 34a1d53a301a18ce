"""-m gpu: the training-path RoI pool pair with a 1-byte arg-max (wssdl_roi_pool_forward_compact /
_backward_compact) against the C oracle: top bit-equal, the codes expanded to the reference's i32
argmax bit-equal, bottom_diff bit-equal (same f32 summation order), for both bin roundings, at
every config's feature-map shape and for every tile variant the backward can be built with."""
import os

import numpy as np
import pytest

from oracle import c_oracle
from test_gpu_parity import _random_rois

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    from wssdl_bus_amd import _lib
    _lib.lib()
    return torch


def _rois_for(rs, R, N, H, W):
    im_h, im_w = 16 * H - 8, 16 * W - 8
    rois = _random_rois(rs, R, N, im_h, im_w)
    k = min(10, R // 4)
    rois[:k, 3:] = rois[:k, 1:3] + rs.uniform(0, 60, (k, 2))               # smaller than 7x7 cells
    rois[k] = [0, 0, 0, im_w - 1, im_h - 1]                                 # the whole image
    rois[k + 1] = [N - 1, im_w - 17, im_h - 17, im_w - 1, im_h - 1]         # bottom-right corner
    rois[k + 2] = [0, 8, 8, 24, 24]                                         # .5 cell coordinates
    rois[k + 3] = [N - 1, 200, 100, 100, 50]                                # malformed (end < start)
    return np.ascontiguousarray(rois[rs.permutation(R)])                    # NOT grouped by image


@pytest.mark.parametrize("shape", [(2, 38, 63, 256), (3, 38, 63, 1024), (1, 63, 100, 1024), (1, 37, 62, 512),
                                   (2, 38, 63, 96), (5, 20, 30, 64), (1, 20, 30, 2048), (2, 12, 17, 4096),
                                   (1, 25, 40, 768)])
@pytest.mark.parametrize("mode", ["cuda", "cpu"])
def test_compact_pair_vs_oracle(torch_cuda, shape, mode):
    torch = torch_cuda
    from wssdl_bus_amd import _lib
    from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op
    N, H, W, C = shape
    assert op.compact_supported(H, W, C, 7, 7)
    rs = np.random.RandomState(H * 1000 + C)
    f = np.maximum(rs.normal(size=shape), 0).astype(np.float32)             # ReLU plateaus: ties at 0
    R = 300 if C >= 512 else 500
    rois = _rois_for(rs, R, N, H, W)
    et, ea = c_oracle.roi_pool_forward(f, rois, 7, 7, 1.0 / 16, mode, threads=16)
    ft, rt = torch.from_numpy(f).cuda(), torch.from_numpy(rois).cuda()
    # forward kernels: 0 = default (windows from the table of wssdl_roi_pool_forward_windows when the shape
    # allows, shared bin columns in registers), 1 = one bin row per wave with a store per bin, 2 =
    # 128-channel waves, 3 = one RoI (7 one-row waves) per workgroup, 4 = as 0 with the geometry computed
    # in the kernel, 5 = a whole RoI per wave, 9 = the two-rows-per-wave sliced kernel
    for fwd in (9, 3, 2, 1, 4, 5, 0):
        with _lib.tuned(roi_fwd_variant=fwd):
            top, arg8 = op.roi_pool_compact(ft, rt, 7, 7, 1.0 / 16, rounding=mode)
        assert arg8.dtype == torch.uint8 and tuple(arg8.shape) == (R, 7, 7, C)
        assert np.array_equal(top.cpu().numpy(), et), fwd
        arg = op.expand_argmax(arg8, rt, shape, 7, 7, 1.0 / 16, rounding=mode)
        assert np.array_equal(arg.cpu().numpy(), ea), fwd
    # a launch this small: waves per bin row (round 6; 7 = one wave per bin, the default where C % 256 == 0; 0 = the sliced
    # kernel; + 100 = whatever the launch size), both arg-max layouts
    for split in (0, 4, 7, 104, 107):
        with _lib.tuned(roi_fwd_one_bin=split):
            top_s, arg8_s = op.roi_pool_compact(ft, rt, 7, 7, 1.0 / 16, rounding=mode)
            top_i, arg_i = op.roi_pool(ft, rt, 7, 7, 1.0 / 16, rounding=mode)
        assert torch.equal(top_s, top) and torch.equal(arg8_s, arg8), split
        assert torch.equal(top_i, top) and torch.equal(arg_i, arg), split
    # the i32 pair of the reference contract gives the same tensors
    top_i, arg_i = op.roi_pool(ft, rt, 7, 7, 1.0 / 16, rounding=mode)
    assert torch.equal(top_i, top) and torch.equal(arg_i, arg)
    diff = rs.normal(size=et.shape).astype(np.float32)
    want = c_oracle.roi_pool_backward(diff, ea, rois, f.shape, 7, 7, 1.0 / 16)
    dt = torch.from_numpy(diff).cuda()
    # every plan of the list-driven walk (tile shape x records in flight x channels per lane) ...
    n_plans = _lib.lib().wssdl_roi_pool_backward_plan_count()
    assert n_plans >= 26
    for plan_id in range(n_plans):
        with _lib.tuned(roi_bwd_plan=plan_id):
            plan = op.roi_pool_grad_prepare(shape, rt, 7, 7, 1.0 / 16, rounding=mode)
            assert plan.plan == plan_id
            got = op.roi_pool_grad_compact(shape, rt, arg8, dt, 7, 7, 1.0 / 16, rounding=mode, plan=plan)
        assert np.array_equal(got.cpu().numpy(), want), (shape, mode, "plan", plan_id)
        got = op.roi_pool_grad_compact(shape, rt, arg8, 2 * dt, 7, 7, 1.0 / 16, rounding=mode, plan=plan)
        assert np.array_equal(got.cpu().numpy(), 2 * want)            # a plan serves many backward calls
        assert not op.flags_raised()
    # ... and every shape of the fallback kernel that filters the RoIs itself (no workspace)
    for variant in range(6):
        with _lib.tuned(roi_bwdc_variant=variant):
            got = op.roi_pool_grad_compact(shape, rt, arg8, dt, 7, 7, 1.0 / 16, rounding=mode, use_workspace=False)
        assert np.array_equal(got.cpu().numpy(), want), (shape, mode, "fallback", variant)


def test_compact_other_pooled_sizes_and_unsupported_shapes(torch_cuda):
    torch = torch_cuda
    from wssdl_bus_amd import _lib
    from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op
    L = _lib.lib()
    assert L.wssdl_roi_pool_compact_supported(38, 63, 1024, 7, 7) == 1
    assert L.wssdl_roi_pool_compact_supported(97, 104, 64, 7, 7) == 1
    assert L.wssdl_roi_pool_compact_supported(98, 100, 64, 7, 7) == 0       # windows could exceed 15 rows
    assert L.wssdl_roi_pool_compact_supported(38, 105, 64, 7, 7) == 0       # ... or 16 columns
    assert L.wssdl_roi_pool_compact_supported(38, 63, 70, 7, 7) == 0        # C % 32 != 0
    assert L.wssdl_roi_pool_compact_supported(38, 63, 64, 1, 1) == 0        # one bin = the whole RoI
    rs = np.random.RandomState(2)
    shape = (1, 20, 24, 32)
    f = np.maximum(rs.normal(size=shape), 0).astype(np.float32)
    x1, y1 = rs.uniform(0, 300, 40), rs.uniform(0, 250, 40)
    rois = np.stack([np.zeros(40), x1, y1, x1 + rs.uniform(8, 80, 40), y1 + rs.uniform(8, 70, 40)], 1).astype(np.float32)
    ft, rt = torch.from_numpy(f).cuda(), torch.from_numpy(rois).cuda()
    for ph, pw, scale in ((6, 6, 1.0 / 3), (14, 14, 1.0 / 16), (3, 11, 1.0 / 8), (2, 2, 1.0 / 16)):
        if not op.compact_supported(20, 24, 32, ph, pw):
            continue
        et, ea = c_oracle.roi_pool_forward(f, rois, ph, pw, scale, "cuda")
        top, arg8 = op.roi_pool_compact(ft, rt, ph, pw, scale, rounding="cuda")
        assert np.array_equal(top.cpu().numpy(), et)
        assert np.array_equal(op.expand_argmax(arg8, rt, shape, ph, pw, scale, rounding="cuda").cpu().numpy(), ea)
        d = rs.normal(size=et.shape).astype(np.float32)
        want = c_oracle.roi_pool_backward(d, ea, rois, shape, ph, pw, scale)
        got = op.roi_pool_grad_compact(shape, rt, arg8, torch.from_numpy(d).cuda(), ph, pw, scale, rounding="cuda")
        assert np.array_equal(got.cpu().numpy(), want), (ph, pw)
    # other pooled sizes on the one-wave-per-bin forward (256-channel waves)
    shape2 = (2, 20, 24, 256)
    f2 = np.maximum(rs.normal(size=shape2), 0).astype(np.float32)
    rois2 = rois.copy()
    rois2[::2, 0] = 1
    f2t, r2t = torch.from_numpy(f2).cuda(), torch.from_numpy(rois2).cuda()
    for ph, pw, scale in ((6, 6, 1.0 / 3), (14, 14, 1.0 / 16), (3, 11, 1.0 / 8), (2, 2, 1.0 / 16), (7, 7, 1.0 / 16)):
        if not op.compact_supported(20, 24, 256, ph, pw):
            continue
        et, ea = c_oracle.roi_pool_forward(f2, rois2, ph, pw, scale, "cuda")
        for split in (7, 4, 0):
            with _lib.tuned(roi_fwd_one_bin=split):
                top, arg8 = op.roi_pool_compact(f2t, r2t, ph, pw, scale, rounding="cuda")
                top_i, arg_i = op.roi_pool(f2t, r2t, ph, pw, scale, rounding="cuda")
            assert np.array_equal(top.cpu().numpy(), et), (ph, pw, split)
            assert np.array_equal(op.expand_argmax(arg8, r2t, shape2, ph, pw, scale, rounding="cuda").cpu().numpy(), ea), (ph, pw, split)
            assert np.array_equal(top_i.cpu().numpy(), et) and np.array_equal(arg_i.cpu().numpy(), ea), (ph, pw, split)
    # empty RoI list, and an all-empty gradient
    top, arg8 = op.roi_pool_compact(ft, rt[:0], 7, 7, 1.0 / 16)
    assert top.shape == (0, 7, 7, 32) and arg8.shape == (0, 7, 7, 32)
    g = op.roi_pool_grad_compact(shape, rt[:0], arg8, top, 7, 7, 1.0 / 16)
    assert tuple(g.shape) == shape and not bool(g.any())


def test_window_table_path_on_a_train_sized_list(torch_cuda):
    """R >= 1024: the Python layer builds the window table (wssdl_roi_pool_forward_windows) and the
    pooling kernel reads the RoI geometry from it; smaller lists (the tests above) compute it in the
    kernel or take the sliced kernel.  Same tensors as the oracle, both roundings."""
    torch = torch_cuda
    from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op
    assert op._WINDOW_TABLE_MIN_ROIS <= 1500
    rs = np.random.RandomState(17)
    for shape, mode in (((2, 38, 63, 256), "cuda"), ((3, 37, 62, 512), "cpu")):
        N, H, W, C = shape
        f = np.maximum(rs.normal(size=shape), 0).astype(np.float32)
        rois = _rois_for(rs, 1500, N, H, W)
        et, ea = c_oracle.roi_pool_forward(f, rois, 7, 7, 1.0 / 16, mode, threads=16)
        ft, rt = torch.from_numpy(f).cuda(), torch.from_numpy(rois).cuda()
        assert _lib_windows_bytes(1500, H, W, C) == 1500 * 7 * 32
        top, arg8 = op.roi_pool_compact(ft, rt, 7, 7, 1.0 / 16, rounding=mode)
        assert np.array_equal(top.cpu().numpy(), et)
        arg = op.expand_argmax(arg8, rt, shape, 7, 7, 1.0 / 16, rounding=mode)
        assert np.array_equal(arg.cpu().numpy(), ea)
        assert not op.compact_overflowed()


@pytest.mark.parametrize("shape", [(2, 38, 63, 256), (2, 38, 63, 1024), (1, 63, 100, 1024), (3, 37, 62, 512),
                                   (1, 20, 30, 2048), (4, 12, 17, 512)])
@pytest.mark.parametrize("mode", ["cuda", "cpu"])
def test_block_table_forward_vs_oracle(torch_cuda, shape, mode):
    """The forward for many proposals per image (csrc/roi_pool_blocks.hip: block-maximum tables of the feature map,
    four table reads per covered bin, bin rows walked in (image, first window row) order): top and the expanded
    arg-max bit-equal to the C oracle -- on ReLU data (ties at 0: the first cell in (h, w) order must win across
    overlapping blocks), on a map whose every value is one of three (ties everywhere), with cells the scan never
    takes (NaN, -inf, -FLT_MAX) and with a -0.0 in the map (the kernel then scans cells), sorted and in RoI order."""
    torch = torch_cuda
    from wssdl_bus_amd import _lib
    from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op
    N, H, W, C = shape
    L = _lib.lib()
    R = 1100
    assert L.wssdl_roi_pool_forward_blocks_bytes(R, N, H, W, C, 7, 7) > 0
    rs = np.random.RandomState(H * 100 + C + (mode == "cpu"))
    rois = _rois_for(rs, R, N, H, W)
    rt = torch.from_numpy(rois).cuda()
    maps = {"relu": np.maximum(rs.normal(size=shape), 0).astype(np.float32),
            "three values": rs.randint(0, 3, size=shape).astype(np.float32) - 1.0}
    odd = rs.normal(size=shape).astype(np.float32)
    pick = rs.uniform(size=shape)
    odd[pick < 0.2] = np.nan
    odd[(pick >= 0.2) & (pick < 0.4)] = -np.inf
    odd[(pick >= 0.4) & (pick < 0.6)] = -np.finfo(np.float32).max
    odd[(pick >= 0.6) & (pick < 0.62)] = np.inf
    odd[:, : H // 2, : W // 2] = np.nan                   # whole windows without a cell the scan takes
    maps["cells the scan never takes"] = odd
    mz = maps["relu"].copy()
    mz[mz == 0] = np.where(rs.uniform(size=int((mz == 0).sum())) < 0.5, np.float32(-0.0), np.float32(0.0))
    maps["minus zero"] = mz
    for name, f in maps.items():
        et, ea = c_oracle.roi_pool_forward(f, rois, 7, 7, 1.0 / 16, mode, threads=16)
        ft = torch.from_numpy(f).cuda()
        for sort, pipe in ((1, 1), (0, 1), (1, 2), (1, 4), (0, 7)):           # (sorted bin rows, waves per bin row)
            with _lib.tuned(roi_fwd_blocks=1, roi_fwd_blocks_sort=sort, roi_fwd_blocks_parts=pipe):
                _lib.timeline.reset(True)
                top, arg8 = op.roi_pool_compact(ft, rt, 7, 7, 1.0 / 16, rounding=mode)
                torch.cuda.synchronize()
                assert "roi_pool_forward_blocks_prepare" in _lib.timeline.summary(), "the block-table path did not run"
                _lib.timeline.reset(False)
            got = top.cpu().numpy()
            assert np.array_equal(got.view(np.uint32), et.view(np.uint32)), (name, sort, pipe)      # bit for bit, -0.0 too
            arg = op.expand_argmax(arg8, rt, shape, 7, 7, 1.0 / 16, rounding=mode)
            assert np.array_equal(arg.cpu().numpy(), ea), (name, sort, pipe)
        # and the rows kernel on the same inputs
        with _lib.tuned(roi_fwd_blocks=0):
            top0, arg80 = op.roi_pool_compact(ft, rt, 7, 7, 1.0 / 16, rounding=mode)
        assert torch.equal(top0.view(torch.int32), top.view(torch.int32)) and torch.equal(arg80, arg8), name
    assert not op.flags_raised()


@pytest.mark.parametrize("which", ["resnet50_alter_weak_r4000_large", "vgg16_joint_r4128", "resnet50_alter_weak_r4000"])
def test_block_table_forward_on_the_saved_proposal_sets(torch_cuda, which):
    """Every RoI of the saved proposal sets the block-table forward is meant for (the alternating weak step's and
    VGG-16's own proposals): top and arg-max codes equal to the rows kernel's, which the tests above tie to the oracle,
    and to the C oracle directly on all of them."""
    torch = torch_cuda
    from wssdl_bus_amd import _lib
    from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op
    rois = np.load(os.path.join(os.path.dirname(__file__), "..", "profiles", "roofline_rois_%s.npy" % which))
    H, W, C = (37, 62, 512) if which.startswith("vgg") else (38, 63, 1024)
    N = int(rois[:, 0].max()) + 1
    g = torch.Generator(device="cuda").manual_seed(5)
    f = torch.relu(torch.randn((N, H, W, C), device="cuda", generator=g))
    rt = torch.from_numpy(rois).cuda()
    with _lib.tuned(roi_fwd_blocks=1):
        top, arg8 = op.roi_pool_compact(f, rt, 7, 7, 1.0 / 16)
    with _lib.tuned(roi_fwd_blocks=0):
        top0, arg80 = op.roi_pool_compact(f, rt, 7, 7, 1.0 / 16)
    assert torch.equal(top, top0) and torch.equal(arg8, arg80)
    # the switch of the host layer: 'auto' = the library's rule by launch shape (all three sets qualify), False = never
    from wssdl_bus_amd.fast_rcnn.config import cfg
    assert cfg.ROI_POOL_FWD_BLOCKS == "auto" and _lib.lib().wssdl_roi_pool_forward_blocks_auto(len(rois), N, H, W, C, 7, 7) == 1
    for setting, expect in (("auto", True), (False, False), (True, True)):
        cfg.ROI_POOL_FWD_BLOCKS = setting
        try:
            _lib.timeline.reset(True)
            t2, a2 = op.roi_pool_compact(f, rt, 7, 7, 1.0 / 16)
            torch.cuda.synchronize()
            assert ("roi_pool_forward_blocks_prepare" in _lib.timeline.summary()) == expect, setting
            assert torch.equal(t2, top) and torch.equal(a2, arg8)
        finally:
            _lib.timeline.reset(False)
            cfg.ROI_POOL_FWD_BLOCKS = "auto"
    # the padded blob's dead rows (batch index -1, cfg.PADDED_ROIS): zeros and the empty code, like the rows kernel
    dead = rt.clone()
    dead[::3, 0] = -1.0
    dead[1::7] = torch.tensor([-1.0, 0.0, 0.0, 0.0, 0.0], device="cuda")
    with _lib.tuned(roi_fwd_blocks=1):
        top_d, arg8_d = op.roi_pool_compact(f, dead, 7, 7, 1.0 / 16)
    with _lib.tuned(roi_fwd_blocks=0):
        top_d0, arg8_d0 = op.roi_pool_compact(f, dead, 7, 7, 1.0 / 16)
    assert torch.equal(top_d, top_d0) and torch.equal(arg8_d, arg8_d0)
    assert not bool(top_d[::3].any()) and bool((arg8_d[::3] == 255).all())
    live = (dead[:, 0] >= 0)
    assert torch.equal(top_d[live], top[live]) and torch.equal(arg8_d[live], arg8[live])
    # ... and against the C oracle directly, EVERY RoI of the set (image by image: the oracle's outputs for one image at a time)
    checked = 0
    for n in range(N):
        idx = np.nonzero(rois[:, 0] == n)[0]
        sub = rois[idx].copy()
        sub[:, 0] = 0
        et, ea = c_oracle.roi_pool_forward(f[n:n + 1].cpu().numpy(), sub, 7, 7, 1.0 / 16, "cuda", threads=16)
        it = torch.from_numpy(idx).cuda()
        assert np.array_equal(top[it].cpu().numpy(), et), n
        arg = op.expand_argmax(arg8[it].contiguous(), torch.from_numpy(sub).cuda(), (1, H, W, C), 7, 7, 1.0 / 16)
        assert np.array_equal(arg.cpu().numpy(), ea), n
        checked += len(idx)
    assert checked == len(rois)


def _lib_windows_bytes(R, H, W, C):
    from wssdl_bus_amd import _lib
    return _lib.lib().wssdl_roi_pool_forward_windows_bytes(R, H, W, C, 7, 7)


def test_compact_overflow_flag_for_rois_far_outside_the_map(torch_cuda):
    torch = torch_cuda
    from wssdl_bus_amd import _lib
    f = torch.relu(torch.randn((1, 38, 63, 64), device="cuda"))
    top = torch.empty((2, 7, 7, 64), device="cuda")
    arg8 = torch.empty((2, 7, 7, 64), dtype=torch.uint8, device="cuda")
    flag = torch.zeros((1,), dtype=torch.int32, device="cuda")

    def run(rois):
        r = torch.tensor(rois, dtype=torch.float32, device="cuda")
        _lib.check(_lib.lib().wssdl_roi_pool_forward_compact(_lib.ptr(f), 1, 38, 63, 64, _lib.ptr(r), 2, 7, 7,
                                                             1.0 / 16, 0, _lib.ptr(top), _lib.ptr(arg8),
                                                             _lib.ptr(flag), _lib.stream()), "fwd compact")
        return int(flag.item())
    assert run([[0, 0, 0, 1007, 607], [0, -40, -40, 1100, 700]]) == 0       # inside / slightly outside: fine
    assert run([[0, 0, 0, 1007, 607], [0, -9000, -9000, 9000, 9000]]) == 1  # one bin spans the whole map

    # the window-table form raises the same flag from its table kernel, and pools the same values
    f2 = torch.relu(torch.randn((1, 38, 63, 256), device="cuda"))
    L = _lib.lib()
    nwin = L.wssdl_roi_pool_forward_windows_bytes(2, 38, 63, 256, 7, 7)
    assert nwin == 2 * 7 * 32 and L.wssdl_roi_pool_forward_windows_bytes(2, 38, 63, 64, 7, 7) == 0
    assert L.wssdl_roi_pool_forward_windows_bytes(2, 38, 63, 256, 6, 6) == 0
    table = torch.empty((nwin,), dtype=torch.uint8, device="cuda")
    for rois, want in (([[0, 0, 0, 1007, 607], [0, -40, -40, 1100, 700]], 0),
                       ([[0, 0, 0, 1007, 607], [0, -9000, -9000, 9000, 9000]], 1)):
        r = torch.tensor(rois, dtype=torch.float32, device="cuda")
        flag.zero_()
        _lib.check(L.wssdl_roi_pool_forward_windows(_lib.ptr(r), 2, 1, 38, 63, 256, 7, 7, 1.0 / 16, 0, _lib.ptr(table),
                                                    nwin, _lib.ptr(flag), _lib.stream()), "windows")
        assert int(flag.item()) == want
        if want == 0:
            t1 = torch.empty((2, 7, 7, 256), device="cuda")
            a1 = torch.empty((2, 7, 7, 256), dtype=torch.uint8, device="cuda")
            t2, a2 = torch.empty_like(t1), torch.empty_like(a1)
            _lib.check(L.wssdl_roi_pool_forward_compact_windows(_lib.ptr(f2), 1, 38, 63, 256, _lib.ptr(r), 2, 7, 7, 1.0 / 16,
                                                                0, _lib.ptr(table), _lib.ptr(t1), _lib.ptr(a1),
                                                                _lib.stream()), "fwd windows")
            _lib.check(L.wssdl_roi_pool_forward_compact(_lib.ptr(f2), 1, 38, 63, 256, _lib.ptr(r), 2, 7, 7, 1.0 / 16, 0,
                                                        _lib.ptr(t2), _lib.ptr(a2), _lib.ptr(flag), _lib.stream()), "fwd")
            assert torch.equal(t1, t2) and torch.equal(a1, a2)


def test_autograd_uses_compact_path_and_matches_oracle(torch_cuda):
    torch = torch_cuda
    from wssdl_bus_amd.fast_rcnn.config import cfg
    from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op
    rs = np.random.RandomState(8)
    f_np = np.maximum(rs.normal(size=(2, 38, 63, 64)), 0).astype(np.float32)
    rois_np = _rois_for(rs, 60, 2, 38, 63)
    want_t, want_a = c_oracle.roi_pool_forward(f_np, rois_np, 7, 7, 1.0 / 16, "cuda")
    w_np = rs.normal(size=want_t.shape).astype(np.float32)
    want_g = c_oracle.roi_pool_backward(w_np, want_a, rois_np, f_np.shape, 7, 7, 1.0 / 16)
    for compact in (True, False):
        cfg.ROI_POOL_COMPACT_ARGMAX = compact
        try:
            f = torch.from_numpy(f_np).cuda().requires_grad_(True)
            top, saved = op.RoiPoolFunction.apply(f, torch.from_numpy(rois_np).cuda(), 7, 7, 1.0 / 16, None)
            assert saved.dtype == (torch.uint8 if compact else torch.int32)
            top2, arg = op.roi_pool_autograd(f, torch.from_numpy(rois_np).cuda(), 7, 7, 1.0 / 16)
            assert arg.dtype == torch.int32 and np.array_equal(arg.cpu().numpy(), want_a)
            assert np.array_equal(top2.detach().cpu().numpy(), want_t)
            (top2 * torch.from_numpy(w_np).cuda()).sum().backward()
            assert np.array_equal(f.grad.cpu().numpy(), want_g)
        finally:
            cfg.ROI_POOL_COMPACT_ARGMAX = True
    assert not op.compact_overflowed()


@pytest.mark.parametrize("shape,R,P", [((3, 38, 63, 1024), 1500, 7), ((2, 37, 62, 512), 2600, 7),
                                       ((1, 20, 30, 128), 5000, 7), ((2, 38, 63, 256), 5000, 6)])
@pytest.mark.parametrize("mode", ["cuda", "cpu"])
def test_i32_contract_forward_on_the_wave_uniform_kernel(torch_cuda, shape, R, P, mode):
    """wssdl_roi_pool_forward (the reference op's own two outputs: f32 top + i32 flat-index argmax) runs the
    round-3 forward (one wave per bin row, scalar windows, shared bin columns) with an i32 store for train-sized
    RoI lists: top and argmax bit-equal to the C oracle, RoIs reaching far outside the map included (no window
    limit on this path), and N = -1 ("batch size unknown", the declared ROIPoolForwardLaucher signature)."""
    torch = torch_cuda
    from wssdl_bus_amd import _lib
    from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op
    N, H, W, C = shape
    rs = np.random.RandomState(C + R)
    f = np.maximum(rs.normal(size=shape), 0).astype(np.float32)
    rois = _rois_for(rs, R, N, H, W)
    rois[5] = [0, -4000, -3000, 16 * W + 4000, 16 * H + 3000]      # windows far larger than 15 x 16 cells
    rois[6] = [N - 1, -500, 40, 200, 16 * H + 900]
    et, ea = c_oracle.roi_pool_forward(f, rois, P, P, 1.0 / 16, mode, threads=16)
    ft, rt = torch.from_numpy(f).cuda(), torch.from_numpy(rois).cuda()
    top, arg = op.roi_pool(ft, rt, P, P, 1.0 / 16, rounding=mode)
    assert arg.dtype == torch.int32
    assert np.array_equal(top.cpu().numpy(), et) and np.array_equal(arg.cpu().numpy(), ea)
    # RoiPoolGrad with the i32 arg-max on the list-driven walk (wssdl_roi_pool_backward_ws): the reference's
    # summation order, bit for bit; without a workspace the tile-owner kernel gives the same bits
    w_np = rs.normal(size=et.shape).astype(np.float32)
    want_g = c_oracle.roi_pool_backward(w_np, ea, rois, shape, P, P, 1.0 / 16)
    wt = torch.from_numpy(w_np).cuda()
    g = op.roi_pool_grad(ft, rt, arg, wt, P, P, 1.0 / 16)
    assert np.array_equal(g.cpu().numpy(), want_g)
    g0 = torch.empty_like(ft)
    _lib.check(_lib.lib().wssdl_roi_pool_backward(_lib.ptr(wt), _lib.ptr(arg), _lib.ptr(rt), R, N, H, W, C, P, P, 1.0 / 16,
                                                  _lib.ptr(g0), _lib.stream()), "wssdl_roi_pool_backward")
    assert torch.equal(g0, g)
    # N = -1 (WSSDL_ROI_BATCH_UNKNOWN): no range check of the batch index (every index here is in range, so the result
    # is the same); N = 0 with RoIs to pool is refused (round 5: it used to mean "unknown" as well)
    top0 = torch.empty_like(top)
    arg0 = torch.empty_like(arg)
    _lib.check(_lib.lib().wssdl_roi_pool_forward(_lib.ptr(ft), -1, H, W, C, _lib.ptr(rt), R, P, P, 1.0 / 16,
                                                 {"cuda": 0, "cpu": 1}[mode], _lib.ptr(top0), _lib.ptr(arg0), _lib.stream()),
               "wssdl_roi_pool_forward(N=-1)")
    assert torch.equal(top0, top) and torch.equal(arg0, arg)
    assert _lib.lib().wssdl_roi_pool_forward(_lib.ptr(ft), 0, H, W, C, _lib.ptr(rt), R, P, P, 1.0 / 16,
                                             {"cuda": 0, "cpu": 1}[mode], _lib.ptr(top0), _lib.ptr(arg0),
                                             _lib.stream()) == _lib.ERR_INVALID_ARGUMENT
    # a tuned backward plan the i32 form is not built for must not fail the op (advisor, round 4): it takes its own choice
    with _lib.tuned(roi_bwd_plan=7):
        assert torch.equal(op.roi_pool_grad(ft, rt, arg, wt, P, P, 1.0 / 16), g)
    # with N given, an index >= N is an empty RoI (zeros, -1); with N = -1 it would be read (not tested: out of bounds)
    bad = rois.copy()
    bad[7, 0] = N + 3
    bad[8, 0] = -1
    tb, ab = op.roi_pool(ft, torch.from_numpy(bad).cuda(), P, P, 1.0 / 16, rounding=mode)
    assert not tb[7].any() and not tb[8].any() and bool((ab[7] == -1).all()) and bool((ab[8] == -1).all())
    keep = np.ones(R, bool)
    keep[[7, 8]] = False
    assert np.array_equal(tb.cpu().numpy()[keep], et[keep]) and np.array_equal(ab.cpu().numpy()[keep], ea[keep])


def test_split_walk_is_deterministic_and_within_1e6_of_the_oracle(torch_cuda):
    """wssdl_roi_pool_backward_compact_split (few images, many RoIs per image: the chain-bound launches of the
    reference's default 1 + 2 batch and of the alternating mode): a tile's slot stream cut into K segments
    walked by K waves, partial tiles added in segment order.  K = 1 is the exact walk (bit-equal to the C
    oracle); K > 1 associates each element's f32 sum differently: the same bits on every run, every element within
    1e-6 of the oracle relative to its own sum of |terms|, the tensor within north_star's 1e-5 of its scale.  The library suggests K by launch shape; the autograd pair follows cfg.ROI_POOL_BWD_SPLIT."""
    torch = torch_cuda
    from wssdl_bus_amd import _lib
    from wssdl_bus_amd.fast_rcnn.config import cfg
    from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op
    L = _lib.lib()
    assert L.wssdl_roi_pool_backward_split_segments(4128, 3, 37, 62, 512) == 8        # VGG-16, 1 + 2 images
    assert L.wssdl_roi_pool_backward_split_segments(2000, 1, 38, 63, 1024) == 8       # one weak image
    assert L.wssdl_roi_pool_backward_split_segments(4000, 2, 38, 63, 1024) == 4       # alternating weak step: 8x8 tiles' fewer bytes
    assert L.wssdl_roi_pool_backward_split_segments(4128, 3, 38, 63, 1024) == 4       # 1 + 2 images on a ResNet
    assert L.wssdl_roi_pool_backward_split_segments(8000, 4, 38, 63, 1024) == 1       # four images x 1024 channels: exact walk
    assert L.wssdl_roi_pool_backward_split_segments(4000, 4, 38, 63, 512) == 1
    assert L.wssdl_roi_pool_backward_split_segments(8512, 8, 38, 63, 1024) == 1       # the default workload: exact walk
    assert L.wssdl_roi_pool_backward_split_segments(256, 2, 38, 63, 256) == 1
    assert L.wssdl_roi_pool_backward_split_scratch_bytes(3, 37, 62, 512, 4) == 3 * 3 * 37 * 62 * 512 * 4
    rs = np.random.RandomState(21)
    N, H, W, C = 2, 37, 62, 128
    f_np = np.maximum(rs.normal(size=(N, H, W, C)), 0).astype(np.float32)
    # heavily overlapping RoIs around the image centre (what an untrained RPN proposes): long slot chains
    R = 2400
    ctr = rs.normal([500, 300], [60, 40], size=(R, 2))
    wh = np.exp(rs.normal(np.log(300), 0.25, size=(R, 2)))
    rois_np = np.concatenate([(np.arange(R) % N)[:, None], np.clip(ctr - wh / 2, 0, None),
                              np.minimum(ctr + wh / 2, [991, 591])], axis=1).astype(np.float32)
    rois_np = rois_np[np.argsort(rois_np[:, 0], kind="stable")]
    want_t, want_a = c_oracle.roi_pool_forward(f_np, rois_np, 7, 7, 1.0 / 16, "cuda", threads=16)
    w_np = rs.normal(size=want_t.shape).astype(np.float32)
    want_g = c_oracle.roi_pool_backward(w_np, want_a, rois_np, f_np.shape, 7, 7, 1.0 / 16)
    # per element: the sum of |terms| (the scale its rounding errors live on)
    mag = c_oracle.roi_pool_backward(np.abs(w_np), want_a, rois_np, f_np.shape, 7, 7, 1.0 / 16)
    ft, rt, wt = torch.from_numpy(f_np).cuda(), torch.from_numpy(rois_np).cuda(), torch.from_numpy(w_np).cuda()
    top, arg8 = op.roi_pool_compact(ft, rt, 7, 7, 1.0 / 16)
    assert np.array_equal(top.cpu().numpy(), want_t)
    scale = float(np.abs(want_g).max())
    for plan_id in (-1, 11, 23, 21):
        with _lib.tuned(roi_bwd_plan=plan_id):
            plan = op.roi_pool_grad_prepare((N, H, W, C), rt, 7, 7, 1.0 / 16)
        exact = op.roi_pool_grad_compact((N, H, W, C), rt, arg8, wt, 7, 7, 1.0 / 16, plan=plan, segments=1).cpu().numpy()
        assert np.array_equal(exact, want_g)
        for K in (2, 4, 7, 16):
            a = op.roi_pool_grad_compact((N, H, W, C), rt, arg8, wt, 7, 7, 1.0 / 16, plan=plan, segments=K)
            b = op.roi_pool_grad_compact((N, H, W, C), rt, arg8, wt, 7, 7, 1.0 / 16, plan=plan, segments=K)
            assert torch.equal(a, b)                                                    # deterministic
            a = a.cpu().numpy()
            # a re-associated f32 sum of n terms differs by at most ~n * 2^-24 * sum |terms| (in practice ~sqrt(n)):
            # every element within 1e-6 of the oracle relative to ITS OWN sum of magnitudes, and the whole
            # tensor within north_star's 1e-5 of its scale
            assert np.all(np.abs(a - want_g) <= 1e-6 * mag + 1e-30), (plan_id, K, float((np.abs(a - want_g) / (mag + 1e-30)).max()))
            assert np.abs(a - want_g).max() <= 1e-5 * scale, (plan_id, K)
    assert not op.flags_raised()
    # the autograd pair: 'auto' takes the library's suggestion (8 here: 2 images x 1200 RoIs x 128 channels), 0 the exact walk
    assert cfg.ROI_POOL_BWD_SPLIT == "auto" and op.split_segments((N, H, W, C), R) == 8
    for split, exact_bits in (("auto", False), (0, True)):
        cfg.ROI_POOL_BWD_SPLIT = split
        try:
            f = ft.clone().requires_grad_(True)
            t, _ = op.RoiPoolFunction.apply(f, rt, 7, 7, 1.0 / 16, None)
            (t * wt).sum().backward()
            g = f.grad.cpu().numpy()
            assert np.all(np.abs(g - want_g) <= 1e-6 * mag + 1e-30) and np.abs(g - want_g).max() <= 1e-5 * scale
            assert np.array_equal(g, want_g) or not exact_bits
        finally:
            cfg.ROI_POOL_BWD_SPLIT = "auto"


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 38, 63, 256), (3, 38, 63, 1024), (1, 63, 100, 1024), (1, 37, 62, 512),
                                   (2, 38, 63, 96), (5, 20, 30, 64), (1, 20, 30, 2048), (2, 12, 17, 4096),
                                   (1, 25, 40, 768)])
@pytest.mark.parametrize("mode", ["cuda", "cpu"])
def test_owner_walk_vs_oracle(torch_cuda, shape, mode):
    """wssdl_roi_pool_backward_compact_owner (round 5): every bin listed by ONE tile (the tile of its window's first
    cell; a window larger than the tile's region continues in the next tile), halos merged in a fixed order.  On the
    shapes of test_compact_pair_vs_oracle, every owner plan:
      * integer-valued top_diff (every partial sum exact in f32): bit-equal to the oracle -- each (bin, cell) pair
        is applied exactly once, with the reference's in_roi / candidate tests (roi_pooling_op_gpu.cu.cc:141-186);
      * real-valued top_diff: the same bits on every run, every element within 1e-6 of the oracle relative to its
        own sum of |terms|, the tensor within north_star's 1e-5 of its scale."""
    torch = torch_cuda
    from wssdl_bus_amd import _lib
    from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op
    N, H, W, C = shape
    rs = np.random.RandomState(H * 1000 + C + 7)
    f = np.maximum(rs.normal(size=shape), 0).astype(np.float32)
    R = 300 if C >= 512 else 500
    rois = _rois_for(rs, R, N, H, W)
    # many small RoIs as well (windows of 1-3 cells: the owned case) next to _rois_for's large ones (chains)
    small = rois[: R // 2].copy()
    small[:, 3:] = small[:, 1:3] + rs.uniform(8, 260, (len(small), 2))
    rois = np.ascontiguousarray(np.concatenate([rois, small.astype(np.float32)])[rs.permutation(R + len(small))])
    R = len(rois)
    et, ea = c_oracle.roi_pool_forward(f, rois, 7, 7, 1.0 / 16, mode, threads=16)
    ft, rt = torch.from_numpy(f).cuda(), torch.from_numpy(rois).cuda()
    top, arg8 = op.roi_pool_compact(ft, rt, 7, 7, 1.0 / 16, rounding=mode)
    assert np.array_equal(top.cpu().numpy(), et)
    ints = rs.randint(-8, 9, size=et.shape).astype(np.float32)
    real = rs.normal(size=et.shape).astype(np.float32)
    want_i = c_oracle.roi_pool_backward(ints, ea, rois, f.shape, 7, 7, 1.0 / 16)
    want_r = c_oracle.roi_pool_backward(real, ea, rois, f.shape, 7, 7, 1.0 / 16)
    mag = c_oracle.roi_pool_backward(np.abs(real), ea, rois, f.shape, 7, 7, 1.0 / 16)
    scale = float(np.abs(want_r).max())
    it, rl = torch.from_numpy(ints).cuda(), torch.from_numpy(real).cuda()
    n_plans = _lib.lib().wssdl_roi_pool_backward_owner_plan_count()
    assert n_plans >= 4
    for owner in range(n_plans):
        plan = op.roi_pool_grad_prepare_owner(shape, rt, 7, 7, 1.0 / 16, owner, rounding=mode)
        got = op.roi_pool_grad_compact(shape, rt, arg8, it, 7, 7, 1.0 / 16, rounding=mode, plan=plan).cpu().numpy()
        assert np.array_equal(got, want_i), (shape, mode, owner, int((got != want_i).sum()))
        a = op.roi_pool_grad_compact(shape, rt, arg8, rl, 7, 7, 1.0 / 16, rounding=mode, plan=plan)
        b = op.roi_pool_grad_compact(shape, rt, arg8, rl, 7, 7, 1.0 / 16, rounding=mode, plan=plan)
        assert torch.equal(a, b), (shape, mode, owner)
        a = a.cpu().numpy()
        assert np.all(np.abs(a - want_r) <= 1e-6 * mag + 1e-30), (shape, mode, owner)
        assert np.abs(a - want_r).max() <= 1e-5 * scale, (shape, mode, owner)
        assert not op.flags_raised()
        # the same lists walked by several waves per tile stream (round 6, ..._compact_owner_split): every (bin, cell) pair
        # still applied exactly once, every region cell of every segment written, the merge pass sums them in a fixed order
        if C % 4 == 0:
            for nseg in ((2, 3) if owner in (0, 8, 9) else (2,)):
                plan.owner_segments = nseg
                got = op.roi_pool_grad_compact(shape, rt, arg8, it, 7, 7, 1.0 / 16, rounding=mode, plan=plan).cpu().numpy()
                assert np.array_equal(got, want_i), (shape, mode, owner, nseg, int((got != want_i).sum()))
                a = op.roi_pool_grad_compact(shape, rt, arg8, rl, 7, 7, 1.0 / 16, rounding=mode, plan=plan)
                b = op.roi_pool_grad_compact(shape, rt, arg8, rl, 7, 7, 1.0 / 16, rounding=mode, plan=plan)
                assert torch.equal(a, b), (shape, mode, owner, nseg)
                a = a.cpu().numpy()
                assert np.all(np.abs(a - want_r) <= 1e-6 * mag + 1e-30), (shape, mode, owner, nseg)
                assert np.abs(a - want_r).max() <= 1e-5 * scale, (shape, mode, owner, nseg)
            plan.owner_segments = 1
            assert not op.flags_raised()


@pytest.mark.parametrize("shape", [(3, 38, 63, 1024), (2, 25, 40, 256)])
def test_owner_form_with_the_reference_ops_i32_argmax(torch_cuda, shape):
    """wssdl_roi_pool_backward_owner_i32: the bin-owner walk reading the reference op's own arg-max layout (i32 flat index,
    roi_pooling_op_gpu.cu.cc:71-79), list building + walk + merge in one call (opt-in; the declared contract stays
    the exact walk).  Integer-valued top_diff: bit-equal to the oracle; real-valued: repeatable, within 1e-5 of scale."""
    torch = torch_cuda
    from wssdl_bus_amd import _lib
    from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op
    N, H, W, C = shape
    rs = np.random.RandomState(C + 11)
    f = np.maximum(rs.normal(size=shape), 0).astype(np.float32)
    rois = _rois_for(rs, 600, N, H, W)
    et, ea = c_oracle.roi_pool_forward(f, rois, 7, 7, 1.0 / 16, "cuda", threads=16)
    ft, rt = torch.from_numpy(f).cuda(), torch.from_numpy(rois).cuda()
    top, arg = op.roi_pool(ft, rt, 7, 7, 1.0 / 16)
    assert np.array_equal(arg.cpu().numpy(), ea)
    L = _lib.lib()
    R = rois.shape[0]
    for owner in (0, 1):
        nws = L.wssdl_roi_pool_backward_workspace_bytes(R, N, H, W, 7, 7)
        nscr = L.wssdl_roi_pool_backward_owner_scratch_bytes(N, H, W, C, owner)
        ws = torch.empty((nws,), dtype=torch.uint8, device="cuda")
        scr = torch.empty((nscr,), dtype=torch.uint8, device="cuda")

        def run(d):
            out = torch.empty(shape, dtype=torch.float32, device="cuda")
            _lib.check(L.wssdl_roi_pool_backward_owner_i32(_lib.ptr(d), _lib.ptr(arg), _lib.ptr(rt), R, N, H, W, C, 7, 7, 1.0 / 16,
                                                           _lib.ptr(out), _lib.ptr(ws), nws, owner, _lib.ptr(scr), nscr,
                                                           _lib.stream()), "wssdl_roi_pool_backward_owner_i32")
            return out
        ints = rs.randint(-8, 9, size=et.shape).astype(np.float32)
        assert np.array_equal(run(torch.from_numpy(ints).cuda()).cpu().numpy(),
                              c_oracle.roi_pool_backward(ints, ea, rois, f.shape, 7, 7, 1.0 / 16)), owner
        real = rs.normal(size=et.shape).astype(np.float32)
        want = c_oracle.roi_pool_backward(real, ea, rois, f.shape, 7, 7, 1.0 / 16)
        rl = torch.from_numpy(real).cuda()
        a, b = run(rl), run(rl)
        assert torch.equal(a, b)
        assert np.abs(a.cpu().numpy() - want).max() <= 1e-5 * np.abs(want).max()
    assert L.wssdl_roi_pool_backward_owner_i32(_lib.ptr(rl), _lib.ptr(arg), _lib.ptr(rt), R, N, H, W, C, 7, 7, 1.0 / 16,
                                               _lib.ptr(a), _lib.ptr(ws), nws, 2, _lib.ptr(scr), nscr, _lib.stream()) != 0


def test_owner_rule_respects_the_pooled_size(torch_cuda):
    """A train-sized launch with 14 x 14 bins: the 1-byte pair takes it, the list-driven backward (walk and owner forms:
    pooled sizes up to 8) does not.  The rule must say -1 for it -- with cfg.ROI_POOL_BWD_OWNER = 'auto' the autograd
    pair used to hand owner plan 8 to wssdl_roi_pool_backward_owner_prepare, which refused the launch (round-5 advice)
    -- and the autograd pair must run the fallback kernel and agree with the i32 pair of the reference contract."""
    torch = torch_cuda
    from wssdl_bus_amd import _lib
    from wssdl_bus_amd.fast_rcnn.config import cfg
    from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op
    L = _lib.lib()
    N, H, W, C, R = 2, 20, 30, 1024, 1536
    assert cfg.ROI_POOL_BWD_OWNER == "auto" and not cfg.ROI_POOL_BWD_EXACT
    assert L.wssdl_roi_pool_backward_owner_plan(R, N, H, W, C) == 8                      # the 7 x 7 rule by shape alone
    assert L.wssdl_roi_pool_backward_owner_plan_for(R, N, H, W, C, 7, 7) == 8
    assert L.wssdl_roi_pool_backward_owner_plan_for(R, N, H, W, C, 14, 14) == -1
    assert L.wssdl_roi_pool_backward_owner_plan_for(R, N, H, W, C, 8, 8) == 8
    assert L.wssdl_roi_pool_backward_owner_plan_for(6_000_000, 2, 38, 63, 1024, 7, 7) == -1    # R * 49 * C beyond the walk's offsets
    assert op.compact_supported(H, W, C, 14, 14) and op.owner_plan((N, H, W, C), R, 14, 14) == -1
    rs = np.random.RandomState(8)
    f_np = np.maximum(rs.normal(size=(N, H, W, C)), 0).astype(np.float32)
    rois_np = _rois_for(rs, R, N, H, W)
    rois_np = rois_np[np.argsort(rois_np[:, 0], kind="stable")]
    rt = torch.from_numpy(rois_np).cuda()
    f = torch.from_numpy(f_np).cuda().requires_grad_(True)
    top, arg8 = op.RoiPoolFunction.apply(f, rt, 14, 14, 1.0 / 16, None)
    assert arg8.dtype == torch.uint8
    w = torch.randn(top.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
    (top * w).sum().backward()
    top_i, arg_i = op.roi_pool(f.detach(), rt, 14, 14, 1.0 / 16)
    assert torch.equal(top.detach(), top_i)
    assert torch.equal(op.expand_argmax(arg8, rt, (N, H, W, C), 14, 14, 1.0 / 16), arg_i)
    want = op.roi_pool_grad(f.detach(), rt, arg_i, w, 14, 14, 1.0 / 16)
    assert torch.equal(f.grad, want)                    # both are the reference's summation order
    assert not op.flags_raised()


def test_owner_rule_and_autograd(torch_cuda):
    """Which launches the library sends to the bin-owner form (wssdl_roi_pool_backward_owner_plan; measured in
    tools/owner_ab.sh) and the autograd pair under cfg.ROI_POOL_BWD_OWNER / _EXACT."""
    torch = torch_cuda
    from wssdl_bus_amd import _lib
    from wssdl_bus_amd.fast_rcnn.config import cfg
    from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op
    L = _lib.lib()
    assert L.wssdl_roi_pool_backward_owner_plan(8512, 8, 38, 63, 1024) == 8      # the default workload
    assert L.wssdl_roi_pool_backward_owner_plan(4000, 2, 38, 63, 1024) == 8      # alternating weak step
    assert L.wssdl_roi_pool_backward_owner_plan(4128, 3, 38, 63, 1024) == 8      # the reference's default 1 + 2 batch
    # round 6: from 512 to 2047 (image, channel) pairs the owner form with TWO waves per tile stream
    assert L.wssdl_roi_pool_backward_owner_plan(4128, 3, 37, 62, 512) == 8       # VGG-16 (1536 pairs)
    assert L.wssdl_roi_pool_backward_owner_plan(2000, 1, 38, 63, 1024) == 8      # one weak image x 1024
    assert L.wssdl_roi_pool_backward_owner_plan(4000, 2, 38, 63, 256) == 9       # ResNet-18's weak step (512 pairs)
    assert L.wssdl_roi_pool_backward_owner_plan(2000, 1, 38, 63, 256) == -1      # 256 pairs: the split form
    assert L.wssdl_roi_pool_backward_owner_segments(4128, 3, 37, 62, 512) == 2
    assert L.wssdl_roi_pool_backward_owner_segments(4000, 2, 38, 63, 256) == 2
    assert L.wssdl_roi_pool_backward_owner_segments(4000, 2, 38, 63, 1024) == 1  # 2048 pairs and more: one wave per stream
    assert L.wssdl_roi_pool_backward_owner_segments(8512, 8, 38, 63, 1024) == 1
    assert L.wssdl_roi_pool_backward_owner_split_scratch_bytes(3, 37, 62, 512, 8, 2) == 2 * 3 * 10 * 13 * 6 * 7 * 512 * 4
    assert L.wssdl_roi_pool_backward_owner_plan(1024, 8, 38, 63, 1024) == -1     # supervised-only step: the exact walk
    assert L.wssdl_roi_pool_backward_owner_plan(300, 1, 63, 100, 1024) == -1
    assert L.wssdl_roi_pool_backward_owner_plan(8512, 8, 38, 63, 96) == -1       # C % 128 != 0
    nscr = L.wssdl_roi_pool_backward_owner_scratch_bytes(8, 38, 63, 1024, 0)
    assert nscr == 8 * 10 * 13 * 6 * 7 * 1024 * 4                                 # tiles of 4 x 5 cells, regions of 6 x 7
    rs = np.random.RandomState(33)
    N, H, W, C = 2, 38, 63, 1024
    R = 2400
    f_np = np.maximum(rs.normal(size=(N, H, W, C)), 0).astype(np.float32)
    ctr = rs.normal([500, 300], [200, 120], size=(R, 2))
    wh = np.exp(rs.normal(np.log(200), 0.5, size=(R, 2)))
    rois_np = np.concatenate([(np.arange(R) % N)[:, None], np.clip(ctr - wh / 2, 0, None),
                              np.minimum(ctr + wh / 2, [991, 591])], axis=1).astype(np.float32)
    rois_np = rois_np[np.argsort(rois_np[:, 0], kind="stable")]
    want_t, want_a = c_oracle.roi_pool_forward(f_np, rois_np, 7, 7, 1.0 / 16, "cuda", threads=16)
    w_np = rs.normal(size=want_t.shape).astype(np.float32)
    want_g = c_oracle.roi_pool_backward(w_np, want_a, rois_np, f_np.shape, 7, 7, 1.0 / 16)
    mag = c_oracle.roi_pool_backward(np.abs(w_np), want_a, rois_np, f_np.shape, 7, 7, 1.0 / 16)
    scale = float(np.abs(want_g).max())
    ft, rt, wt = torch.from_numpy(f_np).cuda(), torch.from_numpy(rois_np).cuda(), torch.from_numpy(w_np).cuda()
    assert cfg.ROI_POOL_BWD_OWNER == "auto" and not cfg.ROI_POOL_BWD_EXACT and op.owner_plan((N, H, W, C), R) == 8
    saved = (cfg.ROI_POOL_BWD_OWNER, cfg.ROI_POOL_BWD_EXACT)
    try:
        for owner, exact, bits in (("auto", False, False), (0, False, False), ("auto", True, True), (-1, False, False)):
            cfg.ROI_POOL_BWD_OWNER, cfg.ROI_POOL_BWD_EXACT = owner, exact
            grads = []
            for _ in range(2):
                f = ft.clone().requires_grad_(True)
                t, _ = op.RoiPoolFunction.apply(f, rt, 7, 7, 1.0 / 16, None)
                assert np.array_equal(t.detach().cpu().numpy(), want_t)
                (t * wt).sum().backward()
                grads.append(f.grad)
            assert torch.equal(grads[0], grads[1])
            g = grads[0].cpu().numpy()
            assert np.all(np.abs(g - want_g) <= 1e-6 * mag + 1e-30) and np.abs(g - want_g).max() <= 1e-5 * scale
            assert np.array_equal(g, want_g) or not bits
    finally:
        cfg.ROI_POOL_BWD_OWNER, cfg.ROI_POOL_BWD_EXACT = saved
    assert not op.flags_raised()


def test_compact_full_size_properties(torch_cuda):
    """BASELINE config 3 size (R = 4*128 + 4*2000 = 8512, C = 1024, 8 images 38x63): the compact pair
    equals the i32 pair bit for bit (which test_gpu_parity.py checks against the oracle by
    sampling), plus a sampled oracle comparison of its own."""
    torch = torch_cuda
    from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op
    rs = np.random.RandomState(3)
    N, H, W, C, R = 8, 38, 63, 1024, 8512
    f = torch.relu(torch.randn((N, H, W, C), device="cuda", generator=torch.Generator("cuda").manual_seed(3)))
    rois_np = _random_rois(rs, R, N, 600, 1000)
    rois_np = rois_np[np.argsort(rois_np[:, 0], kind="stable")]
    rois = torch.from_numpy(rois_np).cuda()
    top_i, arg_i = op.roi_pool(f, rois, 7, 7, 1.0 / 16)
    top, arg8 = op.roi_pool_compact(f, rois, 7, 7, 1.0 / 16)
    assert torch.equal(top, top_i)
    assert torch.equal(op.expand_argmax(arg8, rois, (N, H, W, C), 7, 7, 1.0 / 16), arg_i)
    d = torch.randn(top.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(4))
    g_i = op.roi_pool_grad(f, rois, arg_i, d, 7, 7, 1.0 / 16)
    g = op.roi_pool_grad_compact((N, H, W, C), rois, arg8, d, 7, 7, 1.0 / 16)
    assert torch.equal(g, g_i)
    assert torch.equal(op.roi_pool_grad_compact((N, H, W, C), rois, arg8, d, 7, 7, 1.0 / 16), g)   # deterministic
    img0 = np.where(rois_np[:, 0] == 0)[0][:200]
    et, ea = c_oracle.roi_pool_forward(f[:1].cpu().numpy(), rois_np[img0], 7, 7, 1.0 / 16, "cuda", threads=16)
    want = c_oracle.roi_pool_backward(d[img0].cpu().numpy(), ea, rois_np[img0], (1, H, W, C), 7, 7, 1.0 / 16)
    got = op.roi_pool_grad_compact((1, H, W, C), rois[img0], arg8[img0].contiguous(), d[img0].contiguous(), 7, 7, 1.0 / 16)
    assert np.array_equal(got.cpu().numpy(), want)
    assert not op.compact_overflowed()


def roofline_set_parity(torch, images=tuple(range(8)), plans=(None,), owners=()):
    """The RoI-pool pair on the FIXED set bench.py's roofline is quoted on
    (profiles/roofline_rois_r8512.npy: 4 x 128 sampled rows + 4 x 2000 topped-up proposals), run at
    full size, against the C oracle image by image: top, expanded arg-max and bottom_diff bit for
    bit (roi_pooling_op_gpu.cu.cc:20-85,114-190) for the exact walk's `plans`; the bin-owner form's `owners` (what the
    bench line times since round 5) within 1e-6 of every element's own sum of |terms| and 1e-5 of the tensor's scale,
    and bit-repeatable.  All 8512 RoIs by default.  Also used by tools/roofline_leg.py --check."""
    from wssdl_bus_amd import _lib
    from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rois_np = np.load(os.path.join(root, "profiles", "roofline_rois_r8512.npy"))
    N, H, W, C = int(rois_np[:, 0].max()) + 1, 38, 63, 1024
    R = rois_np.shape[0]
    assert (N, R) == (8, 8512)
    gen = torch.Generator("cuda").manual_seed(3)
    f = torch.relu(torch.randn((N, H, W, C), device="cuda", generator=gen))
    rois = torch.from_numpy(rois_np).cuda()
    top, arg8 = op.roi_pool_compact(f, rois, 7, 7, 1.0 / 16)
    arg = op.expand_argmax(arg8, rois, (N, H, W, C), 7, 7, 1.0 / 16)
    d = torch.randn(top.shape, device="cuda", generator=gen)
    grads = []
    for p in plans:
        with _lib.tuned(roi_bwd_plan=-1 if p is None else p):
            plan = op.roi_pool_grad_prepare((N, H, W, C), rois, 7, 7, 1.0 / 16)
            grads.append((plan, op.roi_pool_grad_compact((N, H, W, C), rois, arg8, d, 7, 7, 1.0 / 16, plan=plan)))
    own = []
    for o in owners:
        plan = op.roi_pool_grad_prepare_owner((N, H, W, C), rois, 7, 7, 1.0 / 16, o)
        a = op.roi_pool_grad_compact((N, H, W, C), rois, arg8, d, 7, 7, 1.0 / 16, plan=plan)
        b = op.roi_pool_grad_compact((N, H, W, C), rois, arg8, d, 7, 7, 1.0 / 16, plan=plan)
        assert torch.equal(a, b), ("owner plan not repeatable", o)
        own.append((o, a))
    assert not op.flags_raised()          # no window overflow, lists within their workspace
    # the prepare's record count against the bound the workspace was sized with
    off = _lib.lib().wssdl_roi_pool_backward_status_offset(R, N, H, W, 7, 7)
    status = grads[0][0].workspace[off:off + 16].view(torch.int32).cpu().numpy()
    assert status[1] == 0 and 0 < status[0] * 64 < grads[0][0].nbytes
    checked = {}
    for n in images:
        idx = np.where(rois_np[:, 0] == n)[0]
        sub = rois_np[idx].copy()
        sub[:, 0] = 0
        fn = f[n:n + 1].cpu().numpy()
        et, ea = c_oracle.roi_pool_forward(fn, sub, 7, 7, 1.0 / 16, "cuda", threads=16)
        sl = slice(int(idx[0]), int(idx[-1]) + 1)
        assert np.array_equal(idx, np.arange(sl.start, sl.stop))            # grouped by image
        assert np.array_equal(top[sl].cpu().numpy(), et), ("top", n)
        assert np.array_equal(arg[sl].cpu().numpy(), ea), ("argmax", n)
        want = c_oracle.roi_pool_backward(d[sl].cpu().numpy(), ea, sub, (1, H, W, C), 7, 7, 1.0 / 16)
        for plan, g in grads:
            assert np.array_equal(g[n].cpu().numpy(), want[0]), ("bottom_diff", n, plan.plan)
        if own:
            mag = c_oracle.roi_pool_backward(np.abs(d[sl].cpu().numpy()), ea, sub, (1, H, W, C), 7, 7, 1.0 / 16)[0]
            scale = float(np.abs(want[0]).max())
            for o, g in own:
                got = g[n].cpu().numpy()
                assert np.all(np.abs(got - want[0]) <= 1e-6 * mag + 1e-30), ("owner", o, n)
                assert np.abs(got - want[0]).max() <= 1e-5 * scale, ("owner", o, n)
        checked[n] = len(idx)
    return checked, [p.plan for p, _ in grads]


def test_roofline_roi_set_matches_oracle(torch_cuda):
    """Parity ON the timed workload, in full (round 5: all eight images = all 8512 RoIs of the fixed roofline set):
    top, expanded arg-max and the exact walk's bottom_diff bit for bit -- the plan the library picks for the exact
    walk, 6x6 tiles with two and three records in flight, a 64-channel plan -- and the bin-owner form the bench line
    times (owner plan 8 = plan 0 with the lean decode; plans 0 and 1) at its tolerance."""
    from wssdl_bus_amd import _lib
    assert _lib.lib().wssdl_roi_pool_backward_owner_plan(8512, 8, 38, 63, 1024) == 8
    checked, plans = roofline_set_parity(torch_cuda, images=tuple(range(8)), plans=(None, 11, 13, 23), owners=(8, 0, 1))
    assert checked == {0: 128, 1: 128, 2: 128, 3: 128, 4: 2000, 5: 2000, 6: 2000, 7: 2000}


def test_oversized_roi_flags_and_i32_fallback(torch_cuda):
    """A RoI reaching far outside the map has bin windows the 1-byte code cannot describe: the compact
    forward raises its device flag.  'deferred' checking turns that into an error at the next check;
    'eager' checking re-runs the call on the i32 pair, whose result equals the oracle."""
    torch = torch_cuda
    from wssdl_bus_amd import _lib
    from wssdl_bus_amd.fast_rcnn.config import cfg
    from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op
    rs = np.random.RandomState(11)
    N, H, W, C = 2, 38, 63, 256
    f_np = np.maximum(rs.normal(size=(N, H, W, C)), 0).astype(np.float32)
    rois_np = _rois_for(rs, 64, N, H, W)
    rois_np[5] = [1, -9000.0, -7000.0, 9000.0, 8000.0]            # bins of ~160 x 130 cells, clipped to the map
    et, ea = c_oracle.roi_pool_forward(f_np, rois_np, 7, 7, 1.0 / 16, "cuda", threads=8)
    w_np = rs.normal(size=et.shape).astype(np.float32)
    want = c_oracle.roi_pool_backward(w_np, ea, rois_np, f_np.shape, 7, 7, 1.0 / 16)
    rois, w = torch.from_numpy(rois_np).cuda(), torch.from_numpy(w_np).cuda()
    assert not op.flags_raised()
    old = cfg.ROI_POOL_FLAG_CHECK
    try:
        cfg.ROI_POOL_FLAG_CHECK = "deferred"
        f = torch.from_numpy(f_np).cuda().requires_grad_(True)
        top, _ = op.roi_pool_autograd(f, rois, 7, 7, 1.0 / 16, return_argmax=False)
        with pytest.raises(_lib.HipCallError, match="15 x 16"):
            op.check_flags()
        assert not op.flags_raised()                              # cleared by the check
        cfg.ROI_POOL_FLAG_CHECK = "eager"
        f = torch.from_numpy(f_np).cuda().requires_grad_(True)
        top, arg = op.roi_pool_autograd(f, rois, 7, 7, 1.0 / 16)
        assert arg.dtype == torch.int32
        assert np.array_equal(top.detach().cpu().numpy(), et) and np.array_equal(arg.cpu().numpy(), ea)
        (top * w).sum().backward()
        assert np.array_equal(f.grad.cpu().numpy(), want)
        op.check_flags()                                          # nothing left raised
        # without the far RoI the eager mode stays on the 1-byte pair
        keep = np.ones(len(rois_np), bool)
        keep[5] = False
        f2 = torch.from_numpy(f_np).cuda().requires_grad_(True)
        top2, arg2 = op.RoiPoolFunction.apply(f2, rois[torch.from_numpy(keep).cuda()].contiguous(), 7, 7, 1.0 / 16, None)
        assert arg2.dtype == torch.uint8 and np.array_equal(top2.detach().cpu().numpy(), et[keep])
    finally:
        cfg.ROI_POOL_FLAG_CHECK = old
        op.flags_raised()
