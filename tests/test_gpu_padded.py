"""-m gpu: the sync-free (cfg.PADDED_ROIS) form of the hot path: fixed-shape RoI blob with dead rows
(batch index -1), no device->host copy between the RPN outputs and the loss, capturable in a
hipGraph.  Checked against the compacting default path, which the other tests pin to the oracle."""
import numpy as np
import pytest

from conftest import load_golden

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
    from wssdl_bus_amd.fast_rcnn.config import cfg
    old = (cfg.TRAIN.IMS_PER_BATCH, cfg.TRAIN.WS_IMS_PER_BATCH, cfg.SAMPLING_RNG, cfg.PADDED_ROIS,
           cfg.FUSED_RPN_SOFTMAX)
    yield cfg
    (cfg.TRAIN.IMS_PER_BATCH, cfg.TRAIN.WS_IMS_PER_BATCH, cfg.SAMPLING_RNG, cfg.PADDED_ROIS,
     cfg.FUSED_RPN_SOFTMAX) = old


def _rpn_inputs(torch, case="res_38x63_train"):
    g = load_golden("proposal_layer")
    return (torch.from_numpy(g[case + "/prob"]).cuda(), torch.from_numpy(g[case + "/pred"]).cuda(),
            torch.from_numpy(g[case + "/im_info"]).cuda())


def test_padded_blob_live_rows_equal_compact_blob(torch_cuda, cfg_guard):
    torch = torch_cuda
    cfg = cfg_guard
    from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import proposal_layer
    prob, pred, info = _rpn_inputs(torch)
    cfg.PADDED_ROIS = False
    compact = proposal_layer(prob, pred, info, True, False)
    cfg.PADDED_ROIS = True
    padded = proposal_layer(prob, pred, info, True, False)
    N = prob.shape[0]
    assert tuple(padded.shape) == (N * 2000, 5) and padded._wssdl_pitch == 2000
    live = padded[:, 0] >= 0
    assert torch.equal(padded[live], compact)
    assert bool((padded[~live][:, 1:] == 0).all()) and bool((padded[~live][:, 0] == -1).all())
    for i in range(N):                     # live rows lead each image's slot
        col = padded[i * 2000:(i + 1) * 2000, 0]
        n = int((col == i).sum())
        assert bool((col[:n] == i).all()) and bool((col[n:] == -1).all())


def test_roi_pool_ignores_dead_rows(torch_cuda):
    torch = torch_cuda
    from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op
    from test_gpu_parity import _random_rois
    rs = np.random.RandomState(5)
    N, H, W, C = 2, 38, 63, 256
    f = torch.relu(torch.randn((N, H, W, C), device="cuda", generator=torch.Generator("cuda").manual_seed(1)))
    live_np = _random_rois(rs, 300, N, 600, 1000)
    live_np = live_np[np.argsort(live_np[:, 0], kind="stable")]
    n0 = int((live_np[:, 0] == 0).sum())
    dead = np.zeros((57, 5), np.float32)
    dead[:, 0] = -1
    padded_np = np.concatenate([live_np[:n0], dead[:20], live_np[n0:], dead[20:]])
    is_live = padded_np[:, 0] >= 0
    live, padded = torch.from_numpy(live_np).cuda(), torch.from_numpy(padded_np).cuda()
    top_c, a_c = op.roi_pool_compact(f, live, 7, 7, 1.0 / 16)
    top_p, a_p = op.roi_pool_compact(f, padded, 7, 7, 1.0 / 16)
    m = torch.from_numpy(is_live).cuda()
    assert torch.equal(top_p[m], top_c) and torch.equal(a_p[m], a_c)
    assert not bool(top_p[~m].any()) and bool((a_p[~m] == 255).all())
    d_c = torch.randn(top_c.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(2))
    d_p = torch.randn(top_p.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(3))
    d_p[m] = d_c                           # whatever gradient arrives for dead rows must not matter
    g_c = op.roi_pool_grad_compact((N, H, W, C), live, a_c, d_c, 7, 7, 1.0 / 16)
    g_p = op.roi_pool_grad_compact((N, H, W, C), padded, a_p, d_p, 7, 7, 1.0 / 16)
    assert torch.equal(g_p, g_c)
    g_f = op.roi_pool_grad_compact((N, H, W, C), padded, a_p, d_p, 7, 7, 1.0 / 16, use_workspace=False)
    assert torch.equal(g_f, g_c)
    # the i32 pair of the reference contract treats a negative batch index the same way
    top_i, a_i = op.roi_pool(f, padded, 7, 7, 1.0 / 16)
    assert not bool(top_i[~m].any()) and bool((a_i[~m] == -1).all())
    assert torch.equal(op.roi_pool_grad(f, padded, a_i, d_p, 7, 7, 1.0 / 16), g_c)


def test_head_masked_batch_norm_matches_compact_rows(torch_cuda):
    torch = torch_cuda
    from wssdl_bus_amd.networks import roi_head
    torch.manual_seed(0)
    head = roi_head.ResNetHeadNHWC(18).cuda()
    head.train()
    R, Rp = 48, 70
    x_live = torch.relu(torch.randn((R, 7, 7, 256), device="cuda"))
    mask = torch.zeros(Rp, device="cuda")
    idx = torch.randperm(Rp, device="cuda")[:R].sort().values
    mask[idx] = 1.0
    x_pad = torch.zeros((Rp, 7, 7, 256), device="cuda")
    x_pad[idx] = x_live
    xa = x_live.clone().requires_grad_(True)
    xb = x_pad.clone().requires_grad_(True)
    ya = head(xa)                      # (training mode: the running statistics do not enter the output)
    roi_head.set_roi_mask(mask)
    try:
        yb = head(xb)
    finally:
        roi_head.set_roi_mask(None)
    assert torch.allclose(yb[idx], ya, rtol=2e-3, atol=2e-4)
    w = torch.randn_like(ya)
    (ya * w).sum().backward()
    wb = torch.randn_like(yb)
    wb[idx] = w
    (yb * wb * mask.unsqueeze(1)).sum().backward()
    assert torch.allclose(xb.grad[idx], xa.grad, rtol=5e-3, atol=1e-5)
    assert not bool(xb.grad[mask == 0].any())


def test_padded_step_has_no_host_sync_and_matches_compact_losses(torch_cuda, cfg_guard):
    """One combined step in padded mode under torch's sync debugger (any .cpu() / .item() between
    the backbone output and the backward raises), and the same step's losses against the
    compacting mode on identical sampled rows (sampling switched to exhaustive: every RoI of the
    supervised image is a candidate and the quota covers them)."""
    torch = torch_cuda
    cfg = cfg_guard
    from wssdl_bus_amd import synthetic
    from wssdl_bus_amd.fast_rcnn.train_bus import SolverWrapper
    from wssdl_bus_amd.networks.factory_bus import get_network
    cfg.TRAIN.IMS_PER_BATCH, cfg.TRAIN.WS_IMS_PER_BATCH = 1, 2
    cfg.SAMPLING_RNG = "device"
    cfg.FUSED_RPN_SOFTMAX = True
    torch.manual_seed(9)
    net = get_network("Resnet_train", 18).cuda().to(memory_format=torch.channels_last)
    net.train()
    solver = SolverWrapper(net)
    blobs = synthetic.make_batch(1, 2, 320, 480, seed=9)
    cfg.PADDED_ROIS = True
    solver.joint_backward(blobs)            # warm-up: allocator, caches (not under the debugger)
    net.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        losses = solver.joint_backward(blobs)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    L = net.layers
    rois = L["roi-data"][0]
    assert rois.shape[0] == 128 + 2 * 2000                      # fixed shape
    assert int((rois[128:, 0] == -1).sum()) > 0                 # ... with dead rows
    assert bool(torch.isfinite(losses["loss"])) and bool(torch.isfinite(losses["mil_cross_entropy"]))
    for p in net.parameters():
        assert p.grad is None or bool(torch.isfinite(p.grad).all())
    # the weak rows' live part equals what the compacting mode makes of the same RPN outputs
    from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import proposal_layer_from_score
    cfg.PADDED_ROIS = False
    with torch.no_grad():
        compact = proposal_layer_from_score(L["rpn_cls_score"], L["rpn_bbox_pred"], blobs["im_info"], True)
    live = rois[128:][rois[128:, 0] >= 0]
    assert torch.equal(live, compact[compact[:, 0] >= 1])


@pytest.mark.parametrize("fwd_blocks", [0, 1])
def test_hot_path_chain_is_graph_capturable(torch_cuda, cfg_guard, fwd_blocks):
    """proposal layer -> padded blob -> proposal targets (device sampling) -> RoI pool forward ->
    backward prepare -> RoI pool backward, captured into ONE hipGraph and replayed: identical
    outputs to the eager calls (include/wssdl_bus_hip.h promises capturable entry points).
    fwd_blocks = 1: the forward takes the block-table form (round 6: tables + bin-row order + pooling, no memset,
    no host step), same outputs."""
    torch = torch_cuda
    cfg = cfg_guard
    from wssdl_bus_amd import _lib
    _lib.set_tuning("roi_fwd_blocks", fwd_blocks)
    try:
        _hot_path_chain_capture(torch, cfg, fwd_blocks)
    finally:
        _lib.set_tuning("roi_fwd_blocks", -1)


def _hot_path_chain_capture(torch, cfg, fwd_blocks):
    from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op
    from wssdl_bus_amd.rpn_msr import proposal_target_layer_tf_bus as ptl
    from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import proposal_layer
    cfg.TRAIN.IMS_PER_BATCH, cfg.TRAIN.WS_IMS_PER_BATCH = 1, 1
    cfg.SAMPLING_RNG = "device"
    cfg.PADDED_ROIS = True
    prob, pred, info = _rpn_inputs(torch)
    N, H, W = prob.shape[:3]
    gt = torch.zeros((N, 20, 5), device="cuda")
    gt[0, 0] = torch.tensor([100.0, 80.0, 380.0, 300.0, 1.0])
    gt[0, 1] = torch.tensor([500.0, 60.0, 900.0, 420.0, 0.0])
    ng = torch.tensor([2, 0], dtype=torch.int32, device="cuda")
    feat = torch.relu(torch.randn((N, H, W, 256), device="cuda", generator=torch.Generator("cuda").manual_seed(4)))
    diff_seed = torch.randn((128 + 2000, 7, 7, 256), device="cuda", generator=torch.Generator("cuda").manual_seed(5))

    def chain():
        ptl._device_calls[0] = 41                            # same sampler seed on every run
        rois = proposal_layer(prob, pred, info, True, False)
        out = ptl.proposal_target_layer_joint(rois, gt, ng, 3, True)
        r = out[0].contiguous()
        top, arg8 = op.roi_pool_compact(feat, r, 7, 7, 1.0 / 16)
        plan = op.roi_pool_grad_prepare(tuple(feat.shape), r, 7, 7, 1.0 / 16)
        g = op.roi_pool_grad_compact(tuple(feat.shape), r, arg8, diff_seed * top, 7, 7, 1.0 / 16, plan=plan)
        return r, out[1], top, g
    eager = [t.clone() for t in chain()]                       # also warms caches up
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        chain()                                                # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        captured = chain()
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(eager, captured):
        assert torch.equal(a, b)
    assert tuple(captured[0].shape) == (128 + 2000, 5)
    if fwd_blocks:
        from wssdl_bus_amd import _lib
        _lib.timeline.reset(True)
        chain()
        torch.cuda.synchronize()
        assert "roi_pool_forward_blocks_prepare" in _lib.timeline.summary()
        _lib.timeline.reset(False)


@pytest.mark.parametrize("C,per,relu", [(2048, 16, True), (512, 49, True), (1024, 49, False), (64, 16, True)])
def test_fused_masked_row_batch_norm_equals_compacted(torch_cuda, C, per, relu):
    """csrc/plumbing/rowbn.hip with a live-row mask: the live rows (forward, dx) and the parameter
    gradients equal the unmasked kernels run on the compacted rows; dead rows come out as zeros in
    both directions whatever they hold; the live-row count is right."""
    torch = torch_cuda
    from wssdl_bus_amd.networks import _plumbing
    g = torch.Generator("cuda").manual_seed(C + per)
    R = 301
    mask = (torch.rand((R,), device="cuda", generator=g) > 0.3).to(torch.float32)
    mask[0], mask[-1] = 0.0, 1.0
    x = torch.randn((R * per, C), device="cuda", generator=g)
    x[(mask == 0).repeat_interleave(per)] = 1e6                   # junk in dead rows must not matter
    dy = torch.randn((R * per, C), device="cuda", generator=g)
    w = torch.rand((C,), device="cuda", generator=g) + 0.5
    b = torch.randn((C,), device="cuda", generator=g)
    if not _plumbing.usable(x):
        pytest.skip("fused row batch-norm not usable for this shape")
    live = (mask != 0).repeat_interleave(per)
    y, stats, count = _plumbing.rowbn_forward(x, w, b, 1e-3, relu, mask)
    yc, stats_c, _ = _plumbing.rowbn_forward(x[live].contiguous(), w, b, 1e-3, relu)
    assert float(count) == float(live.sum())
    assert torch.equal(stats[:2], stats_c[:2]) or torch.allclose(stats, stats_c, rtol=1e-6, atol=1e-7)
    assert torch.allclose(y[live], yc, rtol=1e-6, atol=1e-6)
    assert float(y[~live].abs().max()) == 0.0
    dx, dw, db = _plumbing.rowbn_backward(x, dy, w, stats, relu, mask)
    dxc, dwc, dbc = _plumbing.rowbn_backward(x[live].contiguous(), dy[live].contiguous(), w, stats_c, relu)
    assert torch.allclose(dx[live], dxc, rtol=1e-5, atol=1e-6)
    assert float(dx[~live].abs().max()) == 0.0
    assert torch.allclose(dw, dwc, rtol=1e-5, atol=1e-4) and torch.allclose(db, dbc, rtol=1e-5, atol=1e-4)
    # an all-live mask is the unmasked layer
    ones = torch.ones((R,), device="cuda")
    y1, s1, c1 = _plumbing.rowbn_forward(x, w, b, 1e-3, relu, ones)
    y0, s0, _ = _plumbing.rowbn_forward(x, w, b, 1e-3, relu)
    assert torch.equal(y1, y0) and torch.equal(s1, s0) and float(c1) == R * per


def test_short_supervised_image_does_not_leak_into_head_batch_norm(torch_cuda, cfg_guard):
    """Device sampler, DEFAULT (compacting) mode: a supervised image that runs short of candidates keeps
    its fixed 128 rows, the missing ones as padding rows (-1, 0, 0, 0, 0).  Those rows must not enter the
    per-RoI head's batch statistics: the head's outputs for the live rows equal a run on the live rows
    alone, and the running statistics move identically."""
    torch = torch_cuda
    cfg = cfg_guard
    import copy
    from wssdl_bus_amd.networks import roi_head
    from wssdl_bus_amd.networks.roi_head import ResNetHeadNHWC
    cfg.SAMPLING_RNG = "device"
    cfg.PADDED_ROIS = False
    head = ResNetHeadNHWC(18).cuda().train()
    head2 = copy.deepcopy(head)
    g = torch.Generator("cuda").manual_seed(9)
    R, dead = 96, 17
    pooled = torch.relu(torch.randn((R, 7, 7, 256), device="cuda", generator=g))
    mask = torch.ones((R,), device="cuda")
    mask[R - dead:] = 0.0
    pooled[R - dead:] = 0.0                                       # what RoI pooling writes for batch index -1
    roi_head.set_roi_mask(mask)
    try:
        out = head(pooled)
    finally:
        roi_head.set_roi_mask(None)
    ref = head2(pooled[:R - dead].contiguous())
    assert torch.allclose(out[:R - dead], ref, rtol=1e-4, atol=1e-5)
    for (k, a), (_, b) in zip(head.state_dict().items(), head2.state_dict().items()):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5), k
    # without the mask the dead rows shift the statistics (this is what the mask is for)
    head3 = copy.deepcopy(head2)
    wrong = head3(pooled)
    assert not torch.allclose(wrong[:R - dead], ref, rtol=1e-3, atol=1e-4)
    # and the network installs the mask whenever the device sampler is on
    import inspect
    from wssdl_bus_amd.networks import Resnet_train_bus
    assert "SAMPLING_RNG == 'device'" in inspect.getsource(Resnet_train_bus)
