"""-m gpu: the hot path at the shapes of every BASELINE.json config, checked against the oracle.

config 1  VGG-16, 37x62 feature map, C = 512, CPU bin rounding (roi_pooling_op.cc:167-170)
config 2  ResNet-18, 2 supervised 600x1000 images (38x63, C = 256, R = 256)
config 3  ResNet-50 combined mini-batch (covered by test_gpu_parity.py::test_roi_pool_full_size_properties)
config 4  ResNet-50 alternating, weak step (R = 2 * 2000, C = 1024)
config 5  ResNet-101, 1000x1600 (63x100, C = 1024), test-mode RPN, R <= 300
Plus: the five loss terms of a real step against the oracle (a13), the MIL selection op (f1)
and the fused RPN softmax (f2) against their oracle restatements."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import c_oracle, np_oracle as O
from test_gpu_parity import _random_rois

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    from wssdl_bus_amd import _lib
    _lib.lib()
    return torch


@pytest.fixture()
def cfg_guard():
    """Restores the cfg switches the tests below flip."""
    from wssdl_bus_amd.fast_rcnn.config import cfg
    old = (cfg.TRAIN.IMS_PER_BATCH, cfg.TRAIN.WS_IMS_PER_BATCH, cfg.SAMPLING_RNG, cfg.ROI_POOL_ROUNDING,
           cfg.FUSED_RPN_SOFTMAX, cfg.ROI_POOL_BWD_SPLIT, cfg.ROI_POOL_BWD_OWNER, cfg.ROI_POOL_BWD_EXACT)
    yield cfg
    (cfg.TRAIN.IMS_PER_BATCH, cfg.TRAIN.WS_IMS_PER_BATCH, cfg.SAMPLING_RNG, cfg.ROI_POOL_ROUNDING,
     cfg.FUSED_RPN_SOFTMAX, cfg.ROI_POOL_BWD_SPLIT, cfg.ROI_POOL_BWD_OWNER, cfg.ROI_POOL_BWD_EXACT) = old


def _np(t):
    return t.detach().cpu().numpy()


def _check_pool_against_oracle(torch, feat, rois, top, mode, grad_fn=None, top_diff=None, threads=16):
    """feat [N,H,W,C], rois [R,5], top [R,7,7,C] (all GPU): forward bit-equal to the C oracle; when
    grad_fn is given, grad_fn(top_diff) (the product's backward) bit-equal to the oracle's
    ordered scatter of top_diff through the ORACLE's argmax."""
    f_np, r_np = _np(feat), _np(rois)
    et, ea = c_oracle.roi_pool_forward(f_np, r_np, 7, 7, 1.0 / 16, mode, threads=threads)
    assert np.array_equal(_np(top), et)
    if grad_fn is not None:
        want = c_oracle.roi_pool_backward(_np(top_diff), ea, r_np, f_np.shape, 7, 7, 1.0 / 16)
        got = _np(grad_fn(top_diff))
        assert np.array_equal(got, want)
    return et, ea


# ------------------------------------------------- kernels at the config shapes ---

def test_config5_roi_pool_63x100x1024_r300(torch_cuda):
    """63 % 4 != 0 and W = 100: other tile-edge pattern and FastDiv magic than 38x63."""
    torch = torch_cuda
    from wssdl_bus_amd.roi_pooling_layer.roi_pooling_op import roi_pool, roi_pool_grad
    rs = np.random.RandomState(55)
    N, H, W, C, R = 1, 63, 100, 1024, 300
    f = np.maximum(rs.normal(size=(N, H, W, C)), 0).astype(np.float32)
    rois = _random_rois(rs, R, N, 1000, 1600)
    rois[:8, 3:] = rois[:8, 1:3] + rs.uniform(0, 70, (8, 2))               # smaller than 7x7 cells
    rois[8] = [0, 0, 0, 1599, 999]                                          # the whole image
    rois[9] = [0, 1584, 992, 1599, 999]                                     # last cell (62, 99)
    for mode in ("cuda", "cpu"):
        et, ea = c_oracle.roi_pool_forward(f, rois, 7, 7, 1.0 / 16, mode, threads=16)
        top, arg = roi_pool(f, rois, 7, 7, 1.0 / 16, rounding=mode)
        assert np.array_equal(arg, ea) and np.array_equal(top, et)
        diff = rs.normal(size=top.shape).astype(np.float32)
        want = c_oracle.roi_pool_backward(diff, ea, rois, f.shape, 7, 7, 1.0 / 16)
        assert np.array_equal(roi_pool_grad(f, rois, arg, diff, 7, 7, 1.0 / 16), want), mode
    # properties on the GPU tensors: argmax addresses the pooled value; mass conservation
    ft, rt = torch.from_numpy(f).cuda(), torch.from_numpy(rois).cuda()
    top, arg = roi_pool(ft, rt, 7, 7, 1.0 / 16)
    nz = arg >= 0
    assert torch.equal(ft.reshape(-1)[arg[nz].long()], top[nz])
    d = torch.randn_like(top)
    g = roi_pool_grad(ft, rt, arg, d, 7, 7, 1.0 / 16)
    routed = torch.where(nz, d, torch.zeros_like(d)).double().sum().item()
    assert abs(g.double().sum().item() - routed) < 1e-2 * max(1.0, abs(routed)) + 1.0


def test_config1_roi_pool_37x62x512_cpu_rounding(torch_cuda):
    """VGG-16's conv5_3 shape with the proposals of the reference's own run (golden) and the
    CPU op's bin rounding, which BASELINE config 1 names."""
    from wssdl_bus_amd.roi_pooling_layer.roi_pooling_op import roi_pool, roi_pool_grad
    g = load_golden("proposal_layer")
    rois = np.ascontiguousarray(g["vgg_37x62_train/rois"])                  # [2000, 5] from the reference
    rs = np.random.RandomState(51)
    f = np.maximum(rs.normal(size=(1, 37, 62, 512)), 0).astype(np.float32)
    for mode in ("cpu", "cuda"):
        et, ea = c_oracle.roi_pool_forward(f, rois, 7, 7, 1.0 / 16, mode, threads=16)
        top, arg = roi_pool(f, rois, 7, 7, 1.0 / 16, rounding=mode)
        assert np.array_equal(arg, ea) and np.array_equal(top, et)
        diff = rs.normal(size=top.shape).astype(np.float32)
        want = c_oracle.roi_pool_backward(diff, ea, rois, f.shape, 7, 7, 1.0 / 16)
        assert np.array_equal(roi_pool_grad(f, rois, arg, diff, 7, 7, 1.0 / 16), want), mode
    assert (ea == -1).any() or True     # cpu rounding leaves empty bins for small RoIs; not required


# ------------------------------------------------------ networks at the config shapes ---

def test_config5_resnet101_test_forward(torch_cuda, cfg_guard):
    """resnet101_1600_test: im_detect on one 1000x1600 image; <= 300 proposals (test-mode RPN),
    RoI-pool output of the net bit-equal to the oracle on the net's own feature map / rois."""
    torch = torch_cuda
    from wssdl_bus_amd import synthetic
    from wssdl_bus_amd.fast_rcnn.test_bus import im_detect
    from wssdl_bus_amd.networks.factory_bus import get_network
    torch.manual_seed(5)
    net = get_network("Resnet_train", 101).cuda().to(memory_format=torch.channels_last)
    blobs = synthetic.make_batch(1, 0, 1000, 1600, seed=5)
    scores, boxes = im_detect(net, blobs["data"], blobs["im_info"])
    L = net.layers
    feat, rois, top = L["group2/relu"], L["rpn_rois"], L["roi_pool"]
    assert tuple(feat.shape) == (1, 63, 100, 1024)
    R = rois.shape[0]
    assert 0 < R <= 300 and tuple(top.shape) == (R, 7, 7, 1024)
    assert scores.shape == (R, 3) and boxes.shape == (R, 12)
    assert bool((rois[:, 0] == 0).all())
    assert bool((rois[:, 1] >= 0).all()) and bool((rois[:, 3] <= 1599).all()) and bool((rois[:, 4] <= 999).all())
    _check_pool_against_oracle(torch, feat.contiguous(), rois, top, "cuda")
    # the proposals themselves: NMS of the oracle on the GPU-decoded candidates (exact)
    from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import proposal_layer_padded
    prob = L["rpn_cls_prob_reshape"].contiguous() if "rpn_cls_prob_reshape" in L else None
    if prob is not None:
        rp, cnt, dec, sidx, scnt = proposal_layer_padded(prob, L["rpn_bbox_pred"], blobs["im_info"], False,
                                                         debug=True)
        n = int(scnt[0])
        order = _np(sidx)[0, :n]
        # a random-init net in eval mode saturates: many EQUAL scores, whose order inside the
        # reference's re-sort (cpu_nms.pyx:25) is NumPy-version dependent.  The
        # candidates already are in the kernel's (documented) order, so hand the oracle strictly
        # decreasing surrogate scores: greedy NMS only depends on the order.
        dets = np.hstack((_np(dec)[0][order], np.arange(n, 0, -1, dtype=np.float32)[:, None])).astype(np.float32)
        keep = np.asarray(O.nms(dets, 0.7)[:300], dtype=np.int64)
        assert int(cnt[0]) == len(keep) == R
        assert np.array_equal(_np(rois)[:, 1:], dets[keep, :4])


def _pool_grad_through_autograd(torch, layers, feat_key, pool_key):
    feat, top = layers[feat_key], layers[pool_key]

    def fn(top_diff):
        return torch.autograd.grad(top, feat, grad_outputs=top_diff, retain_graph=True)[0]
    return fn


def test_config2_resnet18_two_supervised_images(torch_cuda, cfg_guard):
    """configs[1]: ResNet-18, batch 2 fully supervised at 600x1000: R = 2 * 128, C = 256.  RoI-pool
    output and its gradient inside the real autograd graph against the oracle (all RoIs)."""
    torch = torch_cuda
    cfg = cfg_guard
    from wssdl_bus_amd import synthetic
    from wssdl_bus_amd.fast_rcnn.train_bus import supervised_loss
    from wssdl_bus_amd.networks.factory_bus import get_network
    cfg.SAMPLING_RNG = "device"
    torch.manual_seed(2)
    net = get_network("Resnet_train_alter", 18).cuda().to(memory_format=torch.channels_last)
    net.train()
    blobs = synthetic.make_batch(2, 0, 600, 1000, seed=2)
    L = net(blobs["data"], blobs["im_info"], blobs["gt_boxes"], blobs["num_gt_boxes"], is_training=True, is_ws=False)
    feat, rois, top = L["group2/relu"], L["roi-data"][0], L["roi_pool"]
    assert tuple(feat.shape) == (2, 38, 63, 256) and tuple(top.shape) == (256, 7, 7, 256)
    assert int((rois[:, 0] == 0).sum()) == 128 and int((rois[:, 0] == 1).sum()) == 128
    top.retain_grad()
    losses = supervised_loss(L, net.weight_decay_params())
    losses["loss"].backward(retain_graph=True)
    top_diff = top.grad.clone()
    assert float(top_diff.abs().sum()) > 0
    _check_pool_against_oracle(torch, feat.detach().contiguous(), rois, top, "cuda",
                               _pool_grad_through_autograd(torch, L, "group2/relu", "roi_pool"), top_diff)


def test_config4_resnet50_weak_step(torch_cuda, cfg_guard):
    """configs[3], weak half of an alternating iteration: 2 weak images -> R = 2 * (<= 2000) RoIs,
    C = 1024, MIL loss only.  Forward against the oracle on all RoIs; the step's own top gradient
    and a random one through the backward."""
    torch = torch_cuda
    cfg = cfg_guard
    from wssdl_bus_amd import synthetic
    from wssdl_bus_amd.fast_rcnn.train_bus import SolverWrapper
    from wssdl_bus_amd.networks.factory_bus import get_network
    cfg.SAMPLING_RNG = "device"
    # the bit-for-bit comparison is the exact walk's contract; this launch shape (2 images x 1024 channels, >= 1000 RoIs
    # each) would otherwise take the bin-owner form, which is checked against the same oracle at its own tolerance below
    assert cfg.ROI_POOL_BWD_SPLIT == "auto" and cfg.ROI_POOL_BWD_OWNER == "auto" and not cfg.ROI_POOL_BWD_EXACT
    cfg.ROI_POOL_BWD_EXACT = True
    torch.manual_seed(4)
    net = get_network("Resnet_train_alter", 50).cuda().to(memory_format=torch.channels_last)
    net.train()
    solver = SolverWrapper(net)
    blobs = synthetic.make_batch(0, 2, 600, 1000, seed=4)
    L = net(blobs["data"], blobs["im_info"], blobs["gt_boxes"], blobs["num_gt_boxes"], is_training=True, is_ws=True)
    feat, rois, top = L["group2/relu"], L["roi-data"][0], L["roi_pool"]
    R = rois.shape[0]
    assert tuple(feat.shape) == (2, 38, 63, 1024) and 2000 < R <= 4000 and tuple(top.shape) == (R, 7, 7, 1024)
    assert L["roi-data"][1].shape[0] == R and not bool(L["roi-data"][1].any())     # zero labels (:282-295)
    top.retain_grad()
    from wssdl_bus_amd.fast_rcnn.train_bus import mil_loss
    from wssdl_bus_amd.mil import core as mil_core
    mil = mil_loss(L["cls_score"], rois[:, 0], blobs["im_info"][:, 3].to(torch.int32), 2, solver.global_step,
                   [mil_core.get_mass_max_logit, mil_core.get_mal_max_logit])
    mil.backward(retain_graph=True)
    top_diff = top.grad.clone()
    # (the per-RoI head has batch-norm layers in training mode, so although one instance per bag
    # carries the loss every RoI receives a gradient)
    assert float(top_diff.abs().sum()) > 0
    grad_fn = _pool_grad_through_autograd(torch, L, "group2/relu", "roi_pool")
    et, ea = _check_pool_against_oracle(torch, feat.detach().contiguous(), rois, top, "cuda", grad_fn, top_diff)
    dense = torch.randn(top.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(44))
    want = c_oracle.roi_pool_backward(_np(dense), ea, _np(rois), tuple(feat.shape), 7, 7, 1.0 / 16)
    assert np.array_equal(_np(grad_fn(dense)), want)
    # the default setting on the same inputs: the bin-owner walk (owner plan 0; the split form with 4 segments when the
    # owner form is switched off), both deterministic and within north_star's 1e-5 of the gradient's scale of the
    # oracle's ordered sum
    from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op
    cfg.ROI_POOL_BWD_EXACT = False
    assert op.owner_plan(tuple(feat.shape), R) == 8 and op.split_segments(tuple(feat.shape), R) == 4
    for owner_key, want_variant in (("auto", "bin-owner"), (-1, "split walk, 4")):
        cfg.ROI_POOL_BWD_OWNER = owner_key
        assert op.prepare_backward(tuple(feat.shape), rois, 7, 7, 1.0 / 16).variant.startswith(want_variant)
        f2 = feat.detach().contiguous().requires_grad_(True)
        top2, _ = op.RoiPoolFunction.apply(f2, rois, 7, 7, 1.0 / 16, None)
        assert torch.equal(top2, top)
        g_a, = torch.autograd.grad(top2, f2, dense, retain_graph=True)
        g_b, = torch.autograd.grad(top2, f2, dense)
        assert torch.equal(g_a, g_b)
        assert np.abs(_np(g_a) - want).max() <= 1e-5 * np.abs(want).max()
    cfg.ROI_POOL_BWD_OWNER = "auto"
    cfg.ROI_POOL_BWD_EXACT = True
    # the oracle's MIL loss on the step's own logits
    want_mil = O.multi_task_loss_alter_weak({"roi-data": (_np(rois),), "im_info": _np(blobs["im_info"]),
                                             "cls_score": _np(L["cls_score"])}, 2, solver.global_step)
    assert abs(float(mil) - want_mil) <= 1e-5 * max(1.0, abs(want_mil))


def test_config4_resnet50_supervised_step(torch_cuda, cfg_guard):
    """configs[3], supervised half of an alternating iteration at ResNet-50 (round 6: both halves now run as real
    steps): S = 2 supervised images -> R = 128 * S sampled RoIs, C = 1024, the four supervised loss terms.  RoI-pool
    output on every RoI and its gradient inside the real autograd graph -- the step's own top gradient and a random
    one -- bit for bit against the oracle (a launch of this size takes the exact walk: the reference's summation
    order), and the oracle's losses on the step's own head outputs."""
    torch = torch_cuda
    cfg = cfg_guard
    from wssdl_bus_amd import synthetic
    from wssdl_bus_amd.fast_rcnn.train_bus import supervised_loss
    from wssdl_bus_amd.networks.factory_bus import get_network
    from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op
    cfg.SAMPLING_RNG = "device"
    assert cfg.ROI_POOL_BWD_OWNER == "auto" and not cfg.ROI_POOL_BWD_EXACT
    torch.manual_seed(5)
    net = get_network("Resnet_train_alter", 50).cuda().to(memory_format=torch.channels_last)
    net.train()
    S = 2
    blobs = synthetic.make_batch(S, 0, 600, 1000, seed=5)
    L = net(blobs["data"], blobs["im_info"], blobs["gt_boxes"], blobs["num_gt_boxes"], is_training=True, is_ws=False)
    feat, rois, top = L["group2/relu"], L["roi-data"][0], L["roi_pool"]
    R = rois.shape[0]
    assert tuple(feat.shape) == (S, 38, 63, 1024) and R == 128 * S and tuple(top.shape) == (R, 7, 7, 1024)
    assert all(int((rois[:, 0] == i).sum()) == 128 for i in range(S))
    # the forms this launch takes by default: rows kernel forward, exact walk backward
    assert op.owner_plan(tuple(feat.shape), R) == -1 and op.split_segments(tuple(feat.shape), R) == 1
    assert op.prepare_backward(tuple(feat.shape), rois, 7, 7, 1.0 / 16).variant.startswith("exact walk")
    top.retain_grad()
    losses = supervised_loss(L, net.weight_decay_params())
    losses["loss"].backward(retain_graph=True)
    top_diff = top.grad.clone()
    assert float(top_diff.abs().sum()) > 0
    grad_fn = _pool_grad_through_autograd(torch, L, "group2/relu", "roi_pool")
    et, ea = _check_pool_against_oracle(torch, feat.detach().contiguous(), rois, top, "cuda", grad_fn, top_diff)
    dense = torch.randn(top.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(55))
    want = c_oracle.roi_pool_backward(_np(dense), ea, _np(rois), tuple(feat.shape), 7, 7, 1.0 / 16)
    assert np.array_equal(_np(grad_fn(dense)), want)
    # the four supervised terms (train_bus.py:184-236) from the oracle on this step's own tensors
    rd = tuple(_np(t) for t in L["roi-data"])
    rpn_data = tuple(_np(t) for t in L["rpn-data"])
    want_l = dict(rpn_cross_entropy=O.loss_rpn_cross_entropy(_np(L["rpn_cls_score_reshape"]), rpn_data[0]),
                  rpn_loss_box=O.loss_rpn_box(_np(L["rpn_bbox_pred"]), rpn_data),
                  cross_entropy=O.loss_rcnn_cross_entropy(_np(L["cls_score"]), rd[1]),
                  loss_box=O.loss_rcnn_box(_np(L["bbox_pred"]), rd[2], rd[3], rd[4]))
    for k, want_v in want_l.items():
        assert abs(float(losses[k].detach()) - want_v) <= 1e-5 * max(1.0, abs(want_v)), k


def test_config1_vgg16_forward(torch_cuda, cfg_guard):
    """configs[0] wiring (VGGnet_train_bus.py:43-101) on the GPU: conv5_3 is 37x62x512; pool_5 with
    the CPU op's rounding against the oracle; the combined layer outputs have the reference's
    shapes (1 supervised + 2 weak images)."""
    torch = torch_cuda
    cfg = cfg_guard
    from wssdl_bus_amd import synthetic
    from wssdl_bus_amd.networks.factory_bus import get_network
    cfg.TRAIN.IMS_PER_BATCH, cfg.TRAIN.WS_IMS_PER_BATCH = 1, 2
    cfg.SAMPLING_RNG = "device"
    cfg.ROI_POOL_ROUNDING = "cpu"
    torch.manual_seed(1)
    net = get_network("VGGnet_train").cuda().to(memory_format=torch.channels_last)
    net.train()
    blobs = synthetic.make_batch(1, 2, 600, 1000, seed=1)
    with torch.no_grad():
        L = net(blobs["data"], blobs["im_info"], blobs["gt_boxes"], blobs["num_gt_boxes"], is_training=True,
                is_ws=False)
    feat, rois, top = L["conv5_3"], L["roi-data"][0], L["pool_5"]
    assert tuple(feat.shape) == (3, 37, 62, 512)
    assert tuple(L["rpn-data"][0].shape) == (3, 1, 9 * 37, 62) and L["rpn-data"][0].dtype == torch.int32
    assert bool((L["rpn-data"][0][1:] == -1).all())                        # weak images: all-ignore (:613-626)
    n_valid = L["roi-data"][1].shape[0]
    assert n_valid == 128 and rois.shape[0] > 128 and bool((rois[128:, 0] >= 1).all())
    assert tuple(L["cls_score"].shape) == (rois.shape[0], 3) and tuple(L["bbox_pred"].shape) == (rois.shape[0], 12)
    _check_pool_against_oracle(torch, feat.contiguous(), rois, top, "cpu")


# ----------------------------------------------- a13: the losses of a real step ---

def test_real_step_losses_match_oracle(torch_cuda, cfg_guard):
    """One combined mini-batch (2 supervised + 2 weak images) through the mirrored network:
    the five loss terms the solver reports against the oracle's f64 evaluation of the
    reference formulas on the step's own layer outputs (incl. the row slicing
    train_bus.py:626-628,652-654), at 1e-5."""
    torch = torch_cuda
    cfg = cfg_guard
    from wssdl_bus_amd import synthetic
    from wssdl_bus_amd.fast_rcnn.train_bus import SolverWrapper
    from wssdl_bus_amd.networks.factory_bus import get_network
    cfg.TRAIN.IMS_PER_BATCH, cfg.TRAIN.WS_IMS_PER_BATCH = 2, 2
    cfg.SAMPLING_RNG = "device"
    torch.manual_seed(13)
    net = get_network("Resnet_train", 18).cuda().to(memory_format=torch.channels_last)
    net.train()
    solver = SolverWrapper(net)
    solver.global_step = 4100                                              # MIL scale 1 - 0.99 * 0.81
    blobs = synthetic.make_batch(2, 2, 320, 480, seed=13)
    losses = solver.joint_backward(blobs)
    L = net.layers
    layers = {k: (tuple(_np(x) for x in L[k]) if isinstance(L[k], tuple) else _np(L[k]))
              for k in ("rpn_cls_score_reshape", "rpn-data", "rpn_bbox_pred", "cls_score", "bbox_pred",
                        "roi-data", "im_info")}
    n_valid = layers["roi-data"][1].size
    assert 0 < n_valid <= 2 * 128 and layers["cls_score"].shape[0] > n_valid
    want = O.multi_task_loss_combined(layers, 2, 2, 4100, [_np(w) for w in net.weight_decay_params()])
    for k in ("rpn_cross_entropy", "rpn_loss_box", "cross_entropy", "loss_box", "mil_cross_entropy",
              "weight_decay", "loss"):
        assert abs(float(losses[k]) - want[k]) <= 1e-5 * max(1.0, abs(want[k])), (k, float(losses[k]), want[k])
    assert want["mil_cross_entropy"] > 0 and want["rpn_loss_box"] > 0


# ------------------------------- configs[2]: the step the driver times, layer by layer ---

def test_config3_resnet50_joint_4_plus_4_step_layer_by_layer(torch_cuda, cfg_guard):
    """BASELINE configs[2] at full scale, exactly as bench.py builds its default workload (`resnet50_joint_b8`:
    ResNet-50, 4 supervised + 4 weak images of 600 x 1000, device samplers, fused RPN softmax and loss, compact blob):
    ONE forward + backward of the combined mini-batch, then every hot-path layer of THAT step against the oracle on
    the step's own inputs (round 5; the smaller test_real_step_losses_match_oracle stays):
      a5   anchor labels before sub-sampling == oracle, bit for bit; the device sub-sampler's draw by invariants
           (anchor_target_layer_tf_bus.py:202-217 quotas, subset of the pre-labels, weak images all-ignore :613-626)
      a6/7 decoded boxes within exp rounding; candidate order a valid descending sort of the oracle's f64 softmax
      a8/9 kept rows == the oracle's NMS (cpu_nms.pyx:17-68) on the GPU-decoded candidates, exact; == the layer's rois
      a10  proposal-target rows by invariants (candidates of their image, quotas, overlap bands, labels, targets at
           4 ulp, weights :228-280,187-226); the weak images' rois appended whole (:162-182)
      a11  RoI-pool top and arg-max == oracle on ALL of the step's RoIs, bit for bit
      a12  bottom_diff: the exact walk == oracle bit for bit; the default (bin-owner) form within 1e-5 of the scale
      a13  the five losses at 1e-5."""
    torch = torch_cuda
    cfg = cfg_guard
    from wssdl_bus_amd import _lib, synthetic
    from wssdl_bus_amd.fast_rcnn.train_bus import SolverWrapper, mil_loss, supervised_loss
    from wssdl_bus_amd.mil import core as mil_core
    from wssdl_bus_amd.networks.factory_bus import get_network
    from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op
    from wssdl_bus_amd.rpn_msr.anchor_target_layer_tf_bus import anchor_target_layer_joint
    from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import proposal_layer_padded
    n_s, n_ws, im_h, im_w = 4, 4, 600, 1000
    cfg.TRAIN.IMS_PER_BATCH, cfg.TRAIN.WS_IMS_PER_BATCH = n_s, n_ws
    cfg.SAMPLING_RNG = "device"
    cfg.FUSED_RPN_SOFTMAX = True
    assert cfg.get("FUSED_LOSS", True) and not cfg.PADDED_ROIS and cfg.ROI_POOL_BWD_OWNER == "auto"
    seed = int(cfg.RNG_SEED)
    np.random.seed(seed)
    torch.manual_seed(seed)
    net = get_network("Resnet_train", 50).cuda().to(memory_format=torch.channels_last)
    net.train()
    solver = SolverWrapper(net)
    blobs = synthetic.make_batch(n_s, n_ws, im_h, im_w, seed)
    # joint_backward (train_bus.py:732-764 mirror) with the pooled tensor's gradient retained
    L = net(blobs["data"], blobs["im_info"], blobs["gt_boxes"], blobs["num_gt_boxes"], is_training=True, is_ws=False)
    top = L["roi_pool"]
    top.retain_grad()
    losses = supervised_loss(L, net.weight_decay_params(), n_s)
    n_valid = L["roi-data"][1].numel()
    losses["mil_cross_entropy"] = mil_loss(L["cls_score"][n_valid:], L["roi-data"][0][n_valid:, 0] - n_s,
                                           blobs["im_info"][n_s:, 3].to(torch.int32), n_ws, solver.global_step,
                                           [mil_core.get_mal_max_logit, mil_core.get_mal_max_logit])
    (losses["loss"] + losses["mil_cross_entropy"]).backward(retain_graph=True)
    op.check_flags()
    feat = L["group2/relu"]
    N, H, W, C = feat.shape
    assert (N, H, W, C) == (8, 38, 63, 1024)
    gt, ng, info = _np(blobs["gt_boxes"]), _np(blobs["num_gt_boxes"]), _np(blobs["im_info"])
    A = 9

    # ---- a5: anchor targets ------------------------------------------------------------------------------------
    score_shape = np.zeros((N, H, W, 2 * A), np.float32)
    pre_want = O.anchor_target_layer_joint(score_shape, gt, ng, info, None, True, (16,), (8, 16, 32), "SNUBH",
                                           rng=np.random.RandomState(0),
                                           cfg=dict(RPN_BATCHSIZE=10 ** 9, IMS_PER_BATCH=n_s, WS_IMS_PER_BATCH=n_ws))[0]
    old_bs = cfg.TRAIN.RPN_BATCHSIZE
    cfg.TRAIN.RPN_BATCHSIZE = 10 ** 9                          # nothing to draw: the labels before sub-sampling
    try:
        pre_got = anchor_target_layer_joint(L["rpn_cls_score"], blobs["gt_boxes"], blobs["num_gt_boxes"], blobs["im_info"],
                                            None, True, [16, ], [8, 16, 32], "SNUBH")[0]
    finally:
        cfg.TRAIN.RPN_BATCHSIZE = old_bs
    pre_got = _np(pre_got)
    assert pre_got.shape == pre_want.shape == (N, 1, A * H, W)
    assert np.array_equal(pre_got, pre_want)                                    # bit-identical anchor labels
    lab = _np(L["rpn-data"][0])
    assert lab.shape == pre_want.shape and np.all(lab[n_s:] == -1)              # weak images: all-ignore
    for i in range(n_s):
        a, b = lab[i].reshape(-1), pre_want[i].reshape(-1)
        assert np.all((a == b) | (a == -1))                                     # sub-sampling only disables
        n_fg_pre, n_bg_pre = int((b == 1).sum()), int((b == 0).sum())
        n_fg, n_bg = int((a == 1).sum()), int((a == 0).sum())
        assert n_fg == min(n_fg_pre, 128) and n_bg == min(n_bg_pre, 256 - n_fg)  # :202-217
        inw, outw = _np(L["rpn-data"][2])[i], _np(L["rpn-data"][3])[i]
        fgm = (lab[i].reshape(A, H, W) == 1)
        used = (lab[i].reshape(A, H, W) >= 0)
        for j in range(4):
            assert np.array_equal(inw.reshape(A, 4, H, W)[:, j] == 1, fgm)        # inside weights on fg (:228)
            assert np.array_equal(outw.reshape(A, 4, H, W)[:, j] > 0, used)       # 1 / #(labels >= 0) on fg and bg (:233-244)
        assert np.allclose(outw[outw > 0], 1.0 / (n_fg + n_bg), rtol=1e-6)

    # ---- a6 - a9: proposals ------------------------------------------------------------------------------------
    rp, cnt, dec, sidx, scnt = proposal_layer_padded(L["rpn_cls_score"].detach(), L["rpn_bbox_pred"].detach(),
                                                     blobs["im_info"], True, debug=True, from_logits=True)
    rp, cnt, dec, sidx, scnt = (_np(t) for t in (rp, cnt, dec, sidx, scnt))
    p64 = O.rpn_cls_prob_reshape(_np(L["rpn_cls_score"]))
    pred = _np(L["rpn_bbox_pred"])
    anchors = O.shifted_anchors(H, W, 16, O.generate_anchors(scales=[8, 16, 32]))
    tol = 2.0 ** -21
    rois_layer = _np(L["rpn_rois"])
    off = 0
    for i in range(N):
        st = O.proposal_stages_one_image(p64[i].astype(np.float32), pred[i], info[i], anchors, A, 12000, 2000, 0.7, 16)
        assert np.allclose(dec[i], st["decoded"], rtol=2e-6, atol=2e-4)
        n = int(scnt[i])
        order = sidx[i, :n]
        assert n == len(st["order"]) == 12000
        s64 = p64[i].reshape(-1, 2 * A)[:, A:].reshape(-1)
        so = s64[order]
        assert np.all(so[:-1] >= so[1:] - tol * np.maximum(so[:-1], 1e-3))       # a valid descending sort
        diff = np.setxor1d(order, st["order"])
        if diff.size:
            edge = s64[st["order"][-1]]
            assert np.all(np.abs(s64[diff] - edge) <= tol * max(edge, 1e-3))     # only boundary ties may differ
        assert diff.size <= 12
        # greedy NMS only depends on the order: strictly decreasing surrogate scores (an untrained net has ties)
        dets = np.hstack((dec[i][order], np.arange(n, 0, -1, dtype=np.float32)[:, None])).astype(np.float32)
        keep = np.asarray(O.nms(dets, 0.7)[:2000], dtype=np.int64)
        c = int(cnt[i])
        assert c == len(keep) and c > 0
        assert np.array_equal(rp[i, :c, 1:], dets[keep, :4])
        assert np.array_equal(rois_layer[off:off + c, 1:], dets[keep, :4]) and np.all(rois_layer[off:off + c, 0] == i)
        off += c
    assert off == rois_layer.shape[0]

    # ---- a10: proposal targets (device sampler: invariants) -------------------------------------------------------
    rois = _np(L["roi-data"][0])
    labels, tgt, inw, outw = (_np(L["roi-data"][k]) for k in (1, 2, 3, 4))
    assert n_valid == n_s * 128 and labels.shape == (n_valid, 1) and tgt.shape == (n_valid, 12)
    n_weak = int((rois_layer[:, 0] >= n_s).sum())
    assert rois.shape[0] == n_valid + n_weak
    assert np.array_equal(rois[n_valid:], rois_layer[rois_layer[:, 0] >= n_s])   # the weak images' rois, whole, in order
    from test_gpu_parity import ulp_diff_f32
    for i in range(n_s):
        rows = slice(i * 128, (i + 1) * 128)
        live = rois[rows, 0] >= 0
        assert np.all(rois[rows][live, 0] == i) and np.all(rois[rows][~live] == [-1, 0, 0, 0, 0])
        pos = gt[i, :ng[i]]
        pos = pos[pos[:, 4] > 0]
        cand = np.vstack((rois_layer[rois_layer[:, 0] == i][:, 1:], pos[:, :4]))
        # every sampled row is a candidate of its image, none more often than it occurs
        import collections
        have = collections.Counter(map(bytes, np.ascontiguousarray(cand)))
        took = collections.Counter(map(bytes, np.ascontiguousarray(rois[rows][live, 1:])))
        assert all(have[k] >= v for k, v in took.items())
        ov = O.bbox_overlaps(rois[rows][live, 1:].astype(np.float64), pos[:, :4].astype(np.float64))
        mx, am = ov.max(axis=1), ov.argmax(axis=1)
        lab_i = labels[rows][live, 0]
        fgm = lab_i > 0
        assert fgm.sum() <= 32 and live.sum() <= 128
        assert np.all(mx[fgm] >= 0.5) and np.all(mx[~fgm] < 0.5) and np.all(mx >= 0)
        assert np.array_equal(lab_i[fgm], pos[am[fgm], 4])
        n_fg_avail = int((O.bbox_overlaps(cand.astype(np.float64), pos[:, :4].astype(np.float64)).max(axis=1) >= 0.5).sum())
        assert fgm.sum() == min(32, n_fg_avail)
        want_t = O.bbox_transform(rois[rows][live, 1:][fgm], pos[am[fgm], :4])
        t_i, iw_i, ow_i = tgt[rows][live], inw[rows][live], outw[rows][live]
        for r_, (k, w) in enumerate(zip(lab_i[fgm].astype(int), want_t.astype(np.float32))):
            row = np.where(fgm)[0][r_]
            assert ulp_diff_f32(t_i[row, 4 * k:4 * k + 4], w).max() <= 4
            e = np.zeros(12, np.float32)
            e[4 * k:4 * k + 4] = 1
            assert np.array_equal(iw_i[row], e) and np.array_equal(ow_i[row], e)
        assert not t_i[~fgm].any() and not iw_i[~fgm].any() and not ow_i[~fgm].any()

    # ---- a11 / a12: RoI pooling on ALL of the step's RoIs ---------------------------------------------------------
    R = rois.shape[0]
    assert tuple(top.shape) == (R, 7, 7, C) and 4000 < R <= n_valid + 8000
    f_np = _np(feat.detach().contiguous())
    livem = rois[:, 0] >= 0
    et, ea = c_oracle.roi_pool_forward(f_np, rois[livem], 7, 7, 1.0 / 16, "cuda", threads=16)
    top_np = _np(top)
    assert np.array_equal(top_np[livem], et)                                    # bit for bit, every RoI
    assert not top_np[~livem].any()
    rt = L["roi-data"][0]
    arg8 = op.roi_pool_compact(feat.detach().contiguous(), rt, 7, 7, 1.0 / 16)[1]
    arg = _np(op.expand_argmax(arg8, rt, (N, H, W, C), 7, 7, 1.0 / 16))
    assert np.array_equal(arg[livem], ea) and np.all(arg[~livem] == -1)
    top_diff = top.grad
    assert float(top_diff.abs().sum()) > 0
    td = _np(top_diff)
    want_g = c_oracle.roi_pool_backward(td[livem], ea, rois[livem], f_np.shape, 7, 7, 1.0 / 16)
    scale = float(np.abs(want_g).max())
    # the default form of this launch (what the step ran and the bench times): the bin-owner walk
    assert op.owner_plan((N, H, W, C), R) == 8
    g_default = torch.autograd.grad(top, feat, top_diff, retain_graph=True)[0]
    g_again = torch.autograd.grad(top, feat, top_diff, retain_graph=True)[0]
    assert torch.equal(g_default, g_again)
    assert np.abs(_np(g_default) - want_g).max() <= 1e-5 * scale
    # the exact walk on the same inputs: the reference's summation order, bit for bit
    cfg.ROI_POOL_BWD_EXACT = True
    f2 = feat.detach().contiguous().requires_grad_(True)
    top2, _ = op.RoiPoolFunction.apply(f2, rt, 7, 7, 1.0 / 16, None)
    assert torch.equal(top2, top)
    g_exact = torch.autograd.grad(top2, f2, top_diff)[0]
    assert np.array_equal(_np(g_exact), want_g)
    cfg.ROI_POOL_BWD_EXACT = False

    # ---- a13: the five losses --------------------------------------------------------------------------------------
    layers = {k: (tuple(_np(x) for x in L[k]) if isinstance(L[k], tuple) else _np(L[k]))
              for k in ("rpn_cls_score_reshape", "rpn-data", "rpn_bbox_pred", "cls_score", "bbox_pred", "roi-data", "im_info")}
    want = O.multi_task_loss_combined(layers, n_s, n_ws, solver.global_step, [_np(w) for w in net.weight_decay_params()])
    for k in ("rpn_cross_entropy", "rpn_loss_box", "cross_entropy", "loss_box", "mil_cross_entropy", "weight_decay", "loss"):
        assert abs(float(losses[k]) - want[k]) <= 1e-5 * max(1.0, abs(want[k])), (k, float(losses[k]), want[k])
    op.check_flags()


# ------------------------------------------------------------------ f1: MIL op ---

def test_mil_select_matches_oracle(torch_cuda):
    torch = torch_cuda
    from wssdl_bus_amd.mil import core as M
    rs = np.random.RandomState(31)
    counts = [1500, 1, 2000, 37]
    R = sum(counts)
    logits = rs.normal(size=(R, 3)).astype(np.float32)
    logits[7] = logits[3]                                    # duplicate rows: first extremum wins
    logits[1501 + 40] = logits[1501 + 12]
    logits[1501:1501 + 2000, 2] = np.round(logits[1501:1501 + 2000, 2], 1)      # many exact ties
    logits[1501:1501 + 2000, 0] = np.round(logits[1501:1501 + 2000, 0], 1)
    bag = np.repeat(np.arange(4), counts).astype(np.float32)
    lt = torch.from_numpy(logits).cuda()
    col = torch.from_numpy(bag + 2.0).cuda()                 # batch column with IMS_PER_BATCH = 2 in front
    pairs = {"mal": (M.get_mal_max_logit, O.mil_mal_max), "ben": (M.get_ben_max_logit, O.mil_ben_max),
             "mass": (M.get_mass_max_logit, O.mil_mass_max)}
    for labels in ([1, 2, 1, 2], [2, 1, 2, 1], [1, 1, 1, 1]):
        lab = torch.tensor(labels, dtype=torch.int32, device="cuda")
        for a, b in (("mal", "mal"), ("mass", "mal"), ("ben", "mass")):      # :655, :241, a third wiring
            want, wscale = O.mil_get_bag_logit(logits, bag, 3, np.asarray(labels), 4, [pairs[a][1], pairs[b][1]])
            got, gscale = M.get_bag_logit_device(lt, col, 2.0, lab, 4, [pairs[a][0], pairs[b][0]])
            assert np.array_equal(_np(got), want), (labels, a, b)
            assert np.allclose(_np(gscale), wscale, rtol=1e-6, atol=1e-7)


def test_mil_empty_bag_is_masked_not_fatal(torch_cuda):
    """A weak image without proposals: row -1 from the op; the host gathers row 0, zeroes the bag
    and gives it zero weight -- no device-side assert, no host sync."""
    torch = torch_cuda
    from wssdl_bus_amd import _lib
    from wssdl_bus_amd.fast_rcnn.train_bus import mil_loss
    from wssdl_bus_amd.mil import core as M
    logits = torch.tensor([[0.0, 0.0, 1.0], [0.0, 0.0, 2.0], [1.0, 5.0, 0.5]], device="cuda", requires_grad=True)
    col = torch.tensor([0.0, 0.0, 2.0], device="cuda")                      # bag 1 has no instance
    lab = torch.tensor([2, 2, 1], dtype=torch.int32, device="cuda")
    rows = torch.empty((3,), dtype=torch.int32, device="cuda")
    cnt = torch.empty((3,), dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().wssdl_mil_select(_lib.ptr(logits.detach()), 3, 3, _lib.ptr(col), 1, 0.0, _lib.ptr(lab), 3,
                                           0, 0, _lib.ptr(rows), _lib.ptr(cnt), _lib.stream()), "mil_select")
    assert rows.tolist() == [1, -1, 2] and cnt.tolist() == [2, 0, 1]
    bag, scale, valid = M.get_bag_logit_device(logits, col, 0.0, lab, 3, [M.get_mal_max_logit, M.get_mal_max_logit],
                                               return_valid=True)
    assert valid.tolist() == [True, False, True] and bag[1].tolist() == [0.0, 0.0, 0.0]
    loss = mil_loss(logits, col, lab, 3, 0, [M.get_mal_max_logit, M.get_mal_max_logit])
    # oracle on the two non-empty bags, mean still over 3 bags
    w0 = O.loss_mil(_np(logits)[[0, 1]], np.zeros(2), np.array([2]), 1, 0, [O.mil_mal_max, O.mil_mal_max])
    w2 = O.loss_mil(_np(logits)[[2]], np.zeros(1), np.array([1]), 1, 0, [O.mil_mal_max, O.mil_mal_max])
    assert abs(float(loss) - (w0 + w2) / 3) < 1e-6
    loss.backward()
    assert torch.isfinite(logits.grad).all()
    with pytest.raises(ValueError):
        M.get_bag_logit_device(logits[:0], col[:0], 0.0, lab, 3, [M.get_mal_max_logit, M.get_mal_max_logit])


# ------------------------------------------------- f2: fused RPN softmax vs oracle ---

@pytest.mark.parametrize("case", ["res_38x63_train", "res_63x100_test", "vgg_37x62_train"])
def test_fused_rpn_softmax_matches_f64_oracle(torch_cuda, case):
    """wssdl_proposal_layer_from_logits on logits whose softmax is the golden run's probability map:
    (1) the candidate order it produces is a valid descending sort of the ORACLE's f64 softmax
    (network.py:283-291,398-404) up to a few ulp of f32; (2) its top-N set equals the oracle's up to
    such boundary ties; (3) its final rois equal the reference's own output rows (exp-ulp
    tolerance, >= 99 % of rows)."""
    torch = torch_cuda
    from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import proposal_layer_from_score, proposal_layer_padded
    g = load_golden("proposal_layer")
    prob, pred, info = g[case + "/prob"], g[case + "/pred"], g[case + "/im_info"]
    train = bool(g[case + "/is_training"])
    N, H, W, A2 = prob.shape
    A = A2 // 2
    rs = np.random.RandomState(17)
    # logits with softmax == prob (up to rounding): (log p_bg + s, log p_fg + s), s a random shift per anchor
    shift = rs.normal(0, 2, size=(N, H, W, A)).astype(np.float32)
    logits = np.log(np.maximum(prob, 1e-30)).astype(np.float32)
    logits = logits + np.concatenate([shift, shift], axis=-1)
    p64 = O.rpn_cls_prob_reshape(logits)                                    # f64 oracle of the chain
    pre, post = (12000, 2000) if train else (6000, 300)
    rp, cnt, dec, sidx, scnt = proposal_layer_padded(logits, pred, info, train, debug=True, from_logits=True)
    anchors = O.shifted_anchors(H, W, 16, O.generate_anchors(scales=[8, 16, 32]))
    tol = 2.0 ** -21                                                         # a few ulp of f32 (exp, sum, divide)
    for i in range(N):
        st = O.proposal_stages_one_image(p64[i].astype(np.float32), pred[i], info[i], anchors, A, pre, post, 0.7, 16)
        s64 = p64[i].reshape(-1, A2)[:, A:].reshape(-1)
        n = int(scnt[i])
        order = _np(sidx)[i, :n]
        assert n == len(st["order"])
        so = s64[order]
        assert np.all(so[:-1] >= so[1:] - tol * np.maximum(so[:-1], 1e-3)), "not a descending sort of the f64 softmax"
        diff = np.setxor1d(order, st["order"])
        if diff.size:                                                        # only boundary ties may differ
            edge = s64[st["order"][-1]]
            assert np.all(np.abs(s64[diff] - edge) <= tol * max(edge, 1e-3))
        assert diff.size <= max(4, n // 1000)
    got = proposal_layer_from_score(logits, pred, info, train, False)
    ref = g[case + "/rois"]
    m = min(len(ref), len(got))
    assert abs(len(ref) - len(got)) <= max(2, len(ref) // 100)
    assert np.all(np.abs(ref[:m] - got[:m]) <= 1e-3, axis=1).mean() >= 0.99
