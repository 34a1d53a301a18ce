"""-m gpu: the fused multi-task loss op (csrc/loss.hip) against the oracle's
f64 restatement of train_bus.py:186-235 / :605-647 and against the chain of torch ops it replaces
(values and gradients)."""
import numpy as np
import pytest

from oracle import np_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    from wssdl_bus_amd import _lib
    _lib.lib()
    return torch


def _inputs(torch, rs, N, H, W, A, n_rows, rows_total, K=3, weak_from=None, pad_rows=0):
    """Layer tensors of one step: scores / predictions (leaf tensors), rpn-data, roi-data."""
    dev = "cuda"
    rpn_cls = torch.tensor(rs.normal(0, 2, (N, H, W, 2 * A)), dtype=torch.float32, device=dev, requires_grad=True)
    rpn_box = torch.tensor(rs.normal(0, 0.7, (N, H, W, 4 * A)), dtype=torch.float32, device=dev, requires_grad=True)
    labels = rs.choice([-1, 0, 1], size=(N, 1, A * H, W), p=[0.8, 0.15, 0.05]).astype(np.int32)
    tg = rs.normal(0, 0.7, (N, 4 * A, H, W)).astype(np.float32)
    fg = (labels.reshape(N, A, H, W) == 1)
    inw = np.repeat(fg, 4, axis=1).astype(np.float32)              # channel a*4 + j
    n_ex = max(int((labels >= 0).sum()), 1)
    outw = np.repeat(labels.reshape(N, A, H, W) >= 0, 4, axis=1).astype(np.float32) / n_ex
    # a few large differences so that both branches of the |d| < 1 switch are taken on live elements
    tg[inw > 0] += rs.choice([0.0, 3.0, -2.5], size=int((inw > 0).sum())).astype(np.float32)
    if weak_from is not None:                                       # weak images: all-ignore labels, zero weights
        labels[weak_from:] = -1
        inw[weak_from:] = 0
        outw[weak_from:] = 0
    cls = torch.tensor(rs.normal(0, 1.5, (rows_total, K)), dtype=torch.float32, device=dev, requires_grad=True)
    box = torch.tensor(rs.normal(0, 0.5, (rows_total, 4 * K)), dtype=torch.float32, device=dev, requires_grad=True)
    lab = rs.randint(0, K, size=(n_rows, 1)).astype(np.int32)
    if pad_rows:
        lab[-pad_rows:] = -1
    rtg = rs.normal(0, 0.5, (n_rows, 4 * K)).astype(np.float32)
    rinw = np.zeros((n_rows, 4 * K), np.float32)
    for r in range(n_rows):
        if lab[r, 0] > 0:
            rinw[r, 4 * lab[r, 0]:4 * lab[r, 0] + 4] = 1.0
    routw = (rinw > 0).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    rpn_data = (t(labels), t(tg), t(inw), t(outw))
    roi_data = (torch.zeros((n_rows, 5), device=dev), t(lab), t(rtg), t(rinw), t(routw))
    return rpn_cls, rpn_box, cls, box, rpn_data, roi_data


def _oracle_terms(rpn_cls, rpn_box, cls, box, rpn_data, roi_data, n_box):
    npf = lambda x: x.detach().cpu().numpy()
    reshape = O.reshape_layer(npf(rpn_cls), 2)
    lab = npf(roi_data[1]).reshape(-1)
    live = lab >= 0                                                  # padding rows are not rows of the reference's blob
    rd = [npf(x) for x in roi_data]
    return np.array([
        O.loss_rpn_cross_entropy(reshape, npf(rpn_data[0])),
        O.loss_rpn_box(npf(rpn_box), [npf(x) for x in rpn_data], n_box),
        O.loss_rcnn_cross_entropy(npf(cls)[:lab.size][live], lab[live]),
        O.loss_rcnn_box(npf(box)[:lab.size][live], rd[2][live], rd[3][live], rd[4][live]),
    ])


@pytest.mark.parametrize("case", ["joint_38x63", "sup_63x100", "small_padded", "vgg_37x62"])
def test_fused_loss_values_and_gradients(torch_cuda, case):
    torch = torch_cuda
    from wssdl_bus_amd.fast_rcnn import train_bus as TB
    from wssdl_bus_amd.fast_rcnn.loss_op import multi_task_loss
    rs = np.random.RandomState(len(case))
    if case == "joint_38x63":        # BASELINE configs[2]: 4 supervised + 4 weak images, 512 supervised rows of 8512
        N, H, W, A, n_rows, rows_total, n_box, weak_from, pad = 8, 38, 63, 9, 512, 8512, 4, 4, 0
    elif case == "sup_63x100":       # 1000x1600 map, supervised only
        N, H, W, A, n_rows, rows_total, n_box, weak_from, pad = 1, 63, 100, 9, 128, 128, None, None, 0
    elif case == "small_padded":     # fixed-shape RoI list with padding rows (label -1)
        N, H, W, A, n_rows, rows_total, n_box, weak_from, pad = 2, 5, 7, 3, 40, 61, 2, None, 9
    else:
        N, H, W, A, n_rows, rows_total, n_box, weak_from, pad = 3, 37, 62, 9, 128, 4128, 1, 1, 0
    rpn_cls, rpn_box, cls, box, rpn_data, roi_data = _inputs(torch, rs, N, H, W, A, n_rows, rows_total,
                                                             weak_from=weak_from, pad_rows=pad)
    terms = multi_task_loss(rpn_cls, rpn_box, cls, box, rpn_data, roi_data, n_box)
    want = _oracle_terms(rpn_cls, rpn_box, cls, box, rpn_data, roi_data, n_box)
    got = terms.detach().cpu().numpy().astype(np.float64)
    assert np.allclose(got, want, rtol=1e-5, atol=1e-7), (got, want)          # north_star tolerance: 1e-5

    # gradients: the chain of torch ops the op replaces (autograd), with unequal upstream weights
    wts = torch.tensor([0.7, 1.3, 2.0, 0.4], device="cuda")
    (terms * wts).sum().backward()
    fused = [x.grad.clone() for x in (rpn_cls, rpn_box, cls, box)]
    for x in (rpn_cls, rpn_box, cls, box):
        x.grad = None
    n, h, w, c = rpn_cls.shape
    reshaped = rpn_cls.permute(0, 3, 1, 2).reshape(n, 2, A * h, w).permute(0, 2, 3, 1)     # network.py:283-291, d = 2
    ref = torch.stack([TB.rpn_cls_loss(reshaped, rpn_data[0]), TB.rpn_box_loss(rpn_box, rpn_data, n_box),
                       TB.rcnn_cls_loss(cls, roi_data[1]), TB.rcnn_box_loss(box, roi_data)])
    assert torch.allclose(ref, terms.detach(), rtol=2e-5, atol=1e-7)
    (ref * wts).sum().backward()
    for name, f, x in zip(("rpn_cls_score", "rpn_bbox_pred", "cls_score", "bbox_pred"), fused, (rpn_cls, rpn_box, cls, box)):
        g = x.grad
        scale = float(g.abs().max().clamp_min(1e-30))
        assert float((f - g).abs().max()) <= 2e-5 * scale + 1e-12, name
        assert torch.equal(f == 0, g == 0) or float((f - g).abs().max()) <= 1e-9, name    # same support (zeros stay zeros)


def test_fused_loss_is_the_step_default_and_matches_unfused(torch_cuda):
    """supervised_loss goes through the op for GPU layers (cfg.FUSED_LOSS) and gives the same terms as
    the torch chain; a step without any labelled anchor yields NaN for the RPN CE like the mean of an
    empty gather (train_bus.py:186-192)."""
    torch = torch_cuda
    from wssdl_bus_amd.fast_rcnn import train_bus as TB
    from wssdl_bus_amd.fast_rcnn.config import cfg
    rs = np.random.RandomState(4)
    rpn_cls, rpn_box, cls, box, rpn_data, roi_data = _inputs(torch, rs, 2, 6, 9, 9, 64, 64)
    n, h, w, c = rpn_cls.shape
    layers = {"rpn_cls_score": rpn_cls, "rpn_bbox_pred": rpn_box, "cls_score": cls, "bbox_pred": box,
              "rpn-data": rpn_data, "roi-data": roi_data,
              "rpn_cls_score_reshape": rpn_cls.permute(0, 3, 1, 2).reshape(n, 2, 9 * h, w).permute(0, 2, 3, 1)}
    old = cfg.get("FUSED_LOSS", True)
    try:
        cfg.FUSED_LOSS = True
        a = TB.supervised_loss(layers, [], None)
        cfg.FUSED_LOSS = False
        b = TB.supervised_loss(layers, [], None)
    finally:
        cfg.FUSED_LOSS = old
    for k in ("rpn_cross_entropy", "rpn_loss_box", "cross_entropy", "loss_box", "loss"):
        assert torch.allclose(a[k], b[k], rtol=2e-5, atol=1e-7), k
    # the fused terms are slices of the op's output, the unfused ones come from F.cross_entropy
    assert "MultiTaskLoss" in type(a["rpn_cross_entropy"].grad_fn.next_functions[0][0]).__name__
    assert "MultiTaskLoss" not in type(b["rpn_cross_entropy"].grad_fn).__name__
    # no labelled anchor at all
    ignore = (torch.full_like(rpn_data[0], -1),) + tuple(rpn_data[1:])
    from wssdl_bus_amd.fast_rcnn.loss_op import multi_task_loss
    t = multi_task_loss(rpn_cls, rpn_box, cls, box, ignore, roi_data, None)
    assert torch.isnan(t[0]) and torch.isfinite(t[1:]).all()
    t[1:].sum().backward()                                           # gradients of the other terms stay finite
    assert torch.isfinite(rpn_box.grad).all() and float(rpn_cls.grad.abs().max()) == 0.0


# ------------------------------------------------------------------ the MIL term (f1) ---
@pytest.mark.parametrize("mode", ["combined", "alter"])
def test_fused_mil_loss_values_and_gradients(torch_cuda, mode):
    """wssdl_mil_loss_forward / _backward against the oracle's loss_mil (train_bus.py:239-260 /
    :650-671, f64) and against autograd through the selection op + torch CE it replaces: ties in the
    selected column (first instance wins), a single-instance bag, an empty bag, both bag labels (the
    alternating mode switches the selector on label 1, :241), the bag index read from a strided
    column of the rois blob with an offset (:653), and both forms of the scale factor."""
    torch = torch_cuda
    from wssdl_bus_amd.fast_rcnn import train_bus as TB
    from wssdl_bus_amd.fast_rcnn.config import cfg
    from wssdl_bus_amd.mil import core as M
    rs = np.random.RandomState(7 if mode == "combined" else 8)
    counts = [40, 1, 0, 25, 2000]                              # bag 2 is empty
    n_bags = len(counts)
    bag_labels = np.array([1, 2, 1, 2, 1], np.int32)
    R = sum(counts)
    logits_np = rs.normal(0, 2, (R, 3)).astype(np.float32)
    logits_np[3:9, 2] = logits_np[:40, 2].max() + 1.0          # six-way tie of the malignant logit in bag 0
    logits_np[45:50, 0] = logits_np[41:66, 0].min() - 1.0      # tie of the smallest background logit in bag 3
    offset = 4.0 if mode == "combined" else 0.0                 # batch column minus IMS_PER_BATCH (:653)
    rois = np.zeros((R, 5), np.float32)
    rois[:, 0] = np.repeat(np.arange(n_bags), counts) + offset
    funcs_t = [M.get_mal_max_logit, M.get_mal_max_logit] if mode == "combined" else [M.get_mass_max_logit, M.get_mal_max_logit]
    funcs_o = [O.mil_mal_max, O.mil_mal_max] if mode == "combined" else [O.mil_mass_max, O.mil_mal_max]
    old = (cfg.TRAIN.WS_LOSS_USE_ADAPTIVE_SCALE_FACTOR, cfg.get("FUSED_LOSS", True))
    try:
        for adaptive, step in ((True, 4100), (True, 0), (False, 10)):
            cfg.TRAIN.WS_LOSS_USE_ADAPTIVE_SCALE_FACTOR = adaptive
            # the oracle has no notion of an empty bag (tf.arg_max would fail): evaluate it on the
            # non-empty bags and rescale the mean to all bags
            keep = [b for b in range(n_bags) if counts[b] > 0]
            sel_rows = np.concatenate([np.nonzero(rois[:, 0] - offset == b)[0] for b in keep])
            inds = np.repeat(np.arange(len(keep)), [counts[b] for b in keep])
            want = O.loss_mil(logits_np[sel_rows], inds, bag_labels[keep], len(keep), step, funcs_o,
                              dict(WS_LOSS_USE_ADAPTIVE_SCALE_FACTOR=adaptive,
                                   WS_LOSS_SCALE_FACTOR=cfg.TRAIN.WS_LOSS_SCALE_FACTOR,
                                   WS_MAL_PCT=cfg.TRAIN.WS_MAL_PCT)) * len(keep) / n_bags
            res = []
            for fused in (True, False):
                cfg.FUSED_LOSS = fused
                x = torch.tensor(logits_np, device="cuda", requires_grad=True)
                col = torch.from_numpy(rois).cuda()[:, 0] - offset          # strided view, like rois[n_valid:, 0] - n_s
                lab = torch.from_numpy(bag_labels).cuda()
                loss = TB.mil_loss(x, col, lab, n_bags, step, funcs_t)
                (loss * 1.7).backward()
                res.append((float(loss.detach()), x.grad.clone()))
            assert abs(res[0][0] - want) <= 1e-5 * max(abs(want), 1e-3), (mode, adaptive, step, res[0][0], want)
            assert abs(res[0][0] - res[1][0]) <= 2e-6 * max(abs(want), 1e-3)
            g_f, g_t = res[0][1], res[1][1]
            assert float((g_f - g_t).abs().max()) <= 2e-6 * float(g_t.abs().max().clamp_min(1e-12))
            assert int((g_f != 0).any(dim=1).sum()) <= n_bags - 1            # one row per non-empty bag at most
            assert torch.equal((g_f != 0).any(dim=1), (g_t != 0).any(dim=1))
    finally:
        cfg.TRAIN.WS_LOSS_USE_ADAPTIVE_SCALE_FACTOR, cfg.FUSED_LOSS = old


def test_wide_heads_take_the_torch_chain_instead_of_failing(torch_cuda):
    """The fused loss ops serve 2..32 classes (multi-task) and 3..8 (MIL); a wider head must fall back to
    the chain of torch ops, not raise INVALID_ARGUMENT."""
    torch = torch_cuda
    from wssdl_bus_amd.fast_rcnn import train_bus
    from wssdl_bus_amd.fast_rcnn.config import cfg
    from wssdl_bus_amd.mil import core as mil_core
    rs = np.random.RandomState(40)
    N, H, W, A, K = 1, 9, 11, 9, 40
    rpn_cls, rpn_box, cls, box, rpn_data, roi_data = _inputs(torch, rs, N, H, W, A, 24, 24, K=K)
    layers = {"rpn_cls_score": rpn_cls, "rpn_bbox_pred": rpn_box, "cls_score": cls, "bbox_pred": box,
              "rpn-data": rpn_data, "roi-data": roi_data,
              "rpn_cls_score_reshape": rpn_cls.reshape(N, H, W, 2, A).permute(0, 4, 1, 2, 3).reshape(N, A * H, W, 2)}
    assert cfg.get("FUSED_LOSS", True)
    l = train_bus.supervised_loss(layers, [])
    want = _oracle_terms(rpn_cls, rpn_box, cls, box, rpn_data, roi_data, None)
    got = np.array([float(l[k]) for k in ("rpn_cross_entropy", "rpn_loss_box", "cross_entropy", "loss_box")])
    assert np.allclose(got, want, rtol=1e-5, atol=1e-6)
    l["loss"].backward()
    assert cls.grad is not None and bool(torch.isfinite(cls.grad).all())
    # MIL with 12 classes: selection op + torch CE
    logits = torch.tensor(rs.normal(size=(30, 12)), dtype=torch.float32, device="cuda", requires_grad=True)
    inds = torch.tensor(rs.randint(0, 2, size=30).astype(np.float32), device="cuda")
    lab = torch.tensor([1, 2], dtype=torch.int32, device="cuda")
    m = train_bus.mil_loss(logits, inds, lab, 2, 0, [mil_core.get_mal_max_logit, mil_core.get_mal_max_logit])
    m.backward()
    assert bool(torch.isfinite(m)) and bool(torch.isfinite(logits.grad).all())
