"""Edge cases and size-independent properties of the HIP path on a real MI355X (-m gpu):
empty / degenerate inputs, duplicate boxes, batch independence, determinism, and one
end-to-end train step of the mirrored network."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import c_oracle, np_oracle as O
from test_gpu_parity import ulp_diff_f32

pytestmark = pytest.mark.gpu

STRIDE = [16, ]
SCALES = [8, 16, 32]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    from wssdl_bus_amd import _lib
    _lib.lib()
    return torch


def test_roi_pool_empty_and_degenerate(torch_cuda):
    from wssdl_bus_amd.roi_pooling_layer.roi_pooling_op import roi_pool, roi_pool_grad
    rs = np.random.RandomState(0)
    f = rs.normal(size=(2, 10, 12, 64)).astype(np.float32)
    top, arg = roi_pool(f, np.zeros((0, 5), np.float32), 7, 7, 1.0 / 16)
    assert top.shape == (0, 7, 7, 64) and arg.shape == (0, 7, 7, 64)
    g = roi_pool_grad(f, np.zeros((0, 5), np.float32), arg, top, 7, 7, 1.0 / 16)
    assert g.shape == f.shape and not g.any()                       # fully written, all zero
    rois = np.array([[0, 500, 500, 600, 600],                       # entirely outside the 12x10 map
                     [1, 0, 0, 0, 0],                               # single cell
                     [0, 100, 60, 40, 20],                          # malformed: end < start
                     [1, 0, 0, 191, 159]], np.float32)              # whole map
    for mode in ("cuda", "cpu"):
        et, ea = c_oracle.roi_pool_forward(f, rois, 7, 7, 1.0 / 16, mode)
        top, arg = roi_pool(f, rois, 7, 7, 1.0 / 16, rounding=mode)
        assert np.array_equal(top, et) and np.array_equal(arg, ea)
        assert (arg[0] == -1).all() and (top[0] == 0).all()
        d = rs.normal(size=top.shape).astype(np.float32)
        want = c_oracle.roi_pool_backward(d, ea, rois, f.shape, 7, 7, 1.0 / 16, literal=True)
        assert np.array_equal(roi_pool_grad(f, rois, arg, d, 7, 7, 1.0 / 16), want)
    with pytest.raises(ValueError):
        roi_pool(f[0], rois, 7, 7, 1.0 / 16)                        # "data must be 4-dimensional"
    with pytest.raises(ValueError):
        roi_pool(f, rois[:, :4], 7, 7, 1.0 / 16)


def test_roi_pool_other_pooled_sizes(torch_cuda):
    from wssdl_bus_amd.roi_pooling_layer.roi_pooling_op import roi_pool, roi_pool_grad
    rs = np.random.RandomState(2)
    f = np.maximum(rs.normal(size=(1, 20, 24, 32)), 0).astype(np.float32)
    x1 = rs.uniform(0, 300, 40)
    y1 = rs.uniform(0, 250, 40)
    rois = np.stack([np.zeros(40), x1, y1, x1 + rs.uniform(8, 80, 40), y1 + rs.uniform(8, 70, 40)], 1).astype(np.float32)
    for ph, pw, scale in ((6, 6, 1.0 / 3), (1, 1, 1.0 / 16), (14, 14, 1.0 / 16), (3, 11, 1.0 / 8)):
        et, ea = c_oracle.roi_pool_forward(f, rois, ph, pw, scale, "cuda")
        top, arg = roi_pool(f, rois, ph, pw, scale)
        assert np.array_equal(top, et) and np.array_equal(arg, ea), (ph, pw)
        d = rs.normal(size=top.shape).astype(np.float32)
        want = c_oracle.roi_pool_backward(d, ea, rois, f.shape, ph, pw, scale)
        assert np.array_equal(roi_pool_grad(f, rois, arg, d, ph, pw, scale), want), (ph, pw)   # incl. >8 bins per tile


def test_nms_duplicates_and_idempotence(torch_cuda):
    from wssdl_bus_amd.nms.hip_nms import hip_nms
    d = np.tile(np.array([[10, 10, 50, 60]], np.float32), (200, 1))
    dets = np.hstack((d, (np.arange(200)[:, None] / 200.0).astype(np.float32)))
    assert hip_nms(dets, 0.7) == [199]                              # identical boxes: only the best survives
    assert hip_nms(dets, 1.5) == list(range(199, -1, -1))           # threshold above 1: nothing suppressed
    g = load_golden("nms")
    dets = g["n6000/dets"]
    keep = hip_nms(dets, 0.7)
    assert keep == g["n6000/keep_07"].tolist()
    kept = dets[keep]
    assert hip_nms(kept, 0.7) == list(range(len(keep)))             # NMS of its own output keeps everything
    ties = dets[:300].copy()
    ties[:, 4] = 0.5                                                # documented tie rule: higher index first
    k = hip_nms(ties, 0.7)
    order = np.arange(299, -1, -1)
    assert k == [int(order[i]) for i in c_oracle_keep(ties[order], 0.7)]


def c_oracle_keep(dets_sorted, th):
    # oracle NMS visiting the rows in the given order (scores made strictly decreasing)
    d = dets_sorted.copy()
    d[:, 4] = np.linspace(1.0, 0.5, len(d), dtype=np.float32)
    return O.nms(d, th)


def test_proposal_layer_batch_independence_and_properties(torch_cuda):
    torch = torch_cuda
    from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import proposal_layer, proposal_layer_padded
    g = load_golden("proposal_layer")
    prob, pred, info = g["res_38x63_train/prob"], g["res_38x63_train/pred"], g["res_38x63_train/im_info"]
    both = proposal_layer(prob, pred, info, True, False, STRIDE, SCALES)
    for i in range(2):
        single = proposal_layer(prob[i:i + 1], pred[i:i + 1], info[i:i + 1], True, False, STRIDE, SCALES)
        sel = both[both[:, 0] == i]
        assert np.array_equal(sel[:, 1:], single[:, 1:])           # an image's rois do not depend on its batch
    # BASELINE config 3 size: 8 images in one call; invariants of the result
    gen = torch.Generator("cuda").manual_seed(5)
    N, H, W, A = 8, 38, 63, 9
    p = torch.softmax(torch.randn((N, H, W, A, 2), device="cuda", generator=gen), -1)
    prob8 = torch.cat((p[..., 0], p[..., 1]), -1).contiguous()
    pred8 = 0.2 * torch.randn((N, H, W, 4 * A), device="cuda", generator=gen)
    info8 = torch.tensor([[600, 1000, 1.0, 1]] * N, device="cuda")
    rois, counts, dec, sidx, scnt = proposal_layer_padded(prob8, pred8, info8, True, STRIDE, SCALES, debug=True)
    rois2, counts2 = proposal_layer_padded(prob8, pred8, info8, True, STRIDE, SCALES)
    assert torch.equal(rois, rois2) and torch.equal(counts, counts2)      # deterministic
    c = counts.cpu().numpy()
    assert (c <= 2000).all() and (c > 0).all()
    fg = prob8[..., A:].reshape(N, -1)
    for i in range(N):
        r = rois[i, :c[i]].cpu().numpy()
        assert (r[:, 0] == i).all()
        assert (r[:, 1] >= 0).all() and (r[:, 3] <= 999).all() and (r[:, 2] >= 0).all() and (r[:, 4] <= 599).all()
        assert ((r[:, 3] - r[:, 1] + 1) >= 16).all() and ((r[:, 4] - r[:, 2] + 1) >= 16).all()
        n = int(scnt[i])
        s = fg[i][sidx[i, :n].long()].cpu().numpy()
        assert (np.diff(s) <= 0).all()                               # candidates in descending score order
        assert not rois[i, c[i]:].any()


def test_anchor_target_no_positive_gt_is_all_ignore(torch_cuda):
    # the reference raises here (argmax over an empty axis); this implementation returns
    # all-ignore labels and zero targets instead of launching with an invalid index
    from wssdl_bus_amd.fast_rcnn.config import cfg
    from wssdl_bus_amd.rpn_msr.anchor_target_layer_tf_bus import anchor_target_layer
    gt = np.zeros((1, 20, 5), np.float32)
    gt[0, 0] = [100, 100, 300, 300, 0]                                # background box only
    cfg.SAMPLING_RNG = "reference"
    lab, tg, inw, outw = anchor_target_layer(np.zeros((1, 38, 63, 18), np.float32), gt, np.array([1], np.int32),
                                             np.array([[600, 1000, 1, 1]], np.float32), None, STRIDE, SCALES,
                                             "SNUBH", rng=np.random.RandomState(0))
    assert (lab != 1).all() and not tg.any() and not inw.any()


def test_network_train_step_end_to_end(torch_cuda):
    torch = torch_cuda
    from wssdl_bus_amd import synthetic
    from wssdl_bus_amd.fast_rcnn.config import cfg
    from wssdl_bus_amd.fast_rcnn.train_bus import SolverWrapper
    from wssdl_bus_amd.networks.factory_bus import get_network
    cfg.TRAIN.IMS_PER_BATCH, cfg.TRAIN.WS_IMS_PER_BATCH = 1, 1
    cfg.SAMPLING_RNG = "reference"
    np.random.seed(3)
    torch.manual_seed(3)
    try:
        net = get_network("Resnet_train", 18).cuda().to(memory_format=torch.channels_last)
        blobs = synthetic.make_batch(1, 1, 320, 480, seed=3)
        solver = SolverWrapper(net)
        losses = solver.train_step_joint(blobs)
        L = net.layers
        assert L["rpn-data"][0].dtype == torch.int32 and L["rpn-data"][0].shape == (2, 1, 9 * 20, 30)
        assert L["roi-data"][1].dtype == torch.int32
        n_sup = L["roi-data"][1].shape[0]
        assert L["roi-data"][0].shape[0] == L["cls_score"].shape[0] >= n_sup
        assert (L["roi-data"][0][n_sup:, 0] == 1).all()             # weak image's rois follow the sampled rows
        for k in ("loss", "mil_cross_entropy", "rpn_cross_entropy", "rpn_loss_box", "cross_entropy", "loss_box"):
            assert torch.isfinite(losses[k]).all(), k
        before = float(losses["loss"])
        for _ in range(3):
            losses = solver.train_step_joint(blobs)
        assert torch.isfinite(losses["loss"]) and solver.global_step == 4
        # alternating network: supervised + weak step
        net2 = get_network("Resnet_train_alter", 18).cuda().to(memory_format=torch.channels_last)
        s2 = SolverWrapper(net2)
        out = s2.train_step_alter(synthetic.make_batch(1, 0, 320, 480, seed=4), synthetic.make_batch(0, 2, 320, 480, seed=5))
        # the supervised op carries no global step, the weak op counts it (train_bus.py:286-301)
        assert torch.isfinite(out["loss"]) and torch.isfinite(out["mil_cross_entropy"]) and s2.global_step == 1
        assert s2.optimizer_ws is not None and s2.optimizer_ws is not s2.optimizer
        # `loss` is the sum of its parts (train_bus.py:262-270); the parts themselves are compared with
        # the oracle in test_gpu_configs.py::test_real_step_losses_match_oracle
        parts = sum(float(out[k]) for k in ("cross_entropy", "loss_box", "rpn_cross_entropy", "rpn_loss_box",
                                              "weight_decay"))
        assert abs(float(out["loss"]) - parts) <= 1e-5 * max(1.0, abs(parts)) and before > 0
    finally:
        cfg.TRAIN.IMS_PER_BATCH, cfg.TRAIN.WS_IMS_PER_BATCH = 1, 2


def test_post_detection_nms_matches_oracle(torch_cuda):
    """f3: per-class NMS + max_per_image cap (test_bus.py:360-401) against the NumPy formulas."""
    torch = torch_cuda
    from wssdl_bus_amd.fast_rcnn.bbox_transform import bbox_transform_inv
    from wssdl_bus_amd.fast_rcnn.test_bus import _clip_boxes, postprocess_detections
    rs = np.random.RandomState(7)
    R, K = 300, 3
    ctr = rs.uniform(50, 700, size=(R, 2)) * [1.0, 0.6]
    wh = rs.uniform(30, 200, size=(R, 2))
    boxes = np.hstack((ctr - wh / 2, ctr + wh / 2)).astype(np.float32)
    deltas = rs.normal(0, 0.1, size=(R, 4 * K)).astype(np.float32)
    logits = rs.normal(0, 2, size=(R, K)).astype(np.float32)
    e = np.exp(logits - logits.max(1, keepdims=True))
    scores = (e / e.sum(1, keepdims=True)).astype(np.float32)
    pred = _clip_boxes(bbox_transform_inv(torch.from_numpy(boxes).cuda(), torch.from_numpy(deltas).cuda()), (480, 760))
    # oracle decode (K classes): same formulas per class
    want_pred = np.zeros_like(deltas)
    for j in range(K):
        want_pred[:, 4 * j:4 * j + 4] = O.bbox_transform_inv(boxes.astype(np.float64), deltas[:, 4 * j:4 * j + 4])
    want_pred[:, 0::4] = np.maximum(want_pred[:, 0::4], 0)
    want_pred[:, 1::4] = np.maximum(want_pred[:, 1::4], 0)
    want_pred[:, 2::4] = np.minimum(want_pred[:, 2::4], 759)
    want_pred[:, 3::4] = np.minimum(want_pred[:, 3::4], 479)
    assert np.allclose(pred.cpu().numpy(), want_pred, rtol=1e-5, atol=1e-3)
    got = postprocess_detections(torch.from_numpy(scores).cuda(), pred, K, thresh=0.05, max_per_image=40)
    p = pred.cpu().numpy()
    want = {}
    for j in range(1, K):
        inds = np.where(scores[:, j] > 0.05)[0]
        d = np.hstack((p[inds, 4 * j:4 * j + 4], scores[inds, j:j + 1])).astype(np.float32)
        want[j] = d[O.nms(d, 0.3)]
    alls = np.hstack([want[j][:, 4] for j in range(1, K)])
    if len(alls) > 40:
        th = np.sort(alls)[-40]
        for j in range(1, K):
            want[j] = want[j][want[j][:, 4] >= th]
    for j in range(1, K):
        assert np.array_equal(got[j].cpu().numpy(), want[j]), j
    assert sum(len(want[j]) for j in want) <= 40 + 2


def test_im_detect_runs(torch_cuda):
    torch = torch_cuda
    from wssdl_bus_amd import synthetic
    from wssdl_bus_amd.fast_rcnn.test_bus import im_detect, postprocess_detections
    from wssdl_bus_amd.networks.factory_bus import get_network
    torch.manual_seed(1)
    net = get_network("Resnet_train", 18).cuda().to(memory_format=torch.channels_last)
    blobs = synthetic.make_batch(1, 0, 320, 480, seed=9)
    scores, boxes = im_detect(net, blobs["data"], blobs["im_info"])
    assert scores.shape[0] == boxes.shape[0] <= 300 and scores.shape[1] == 3 and boxes.shape[1] == 12
    assert torch.allclose(scores.sum(1), torch.ones_like(scores[:, 0]), atol=1e-5)
    dets = postprocess_detections(scores, boxes, 3, thresh=0.0, max_per_image=50)
    assert sum(d.shape[0] for d in dets.values()) <= 52


def test_mil_select_device_matches_host_restatement(torch_cuda):
    """f1: the HIP bag-selection op against the host restatement of mil/core.py:11-96."""
    torch = torch_cuda
    from wssdl_bus_amd.mil import core as M
    g = torch.Generator("cuda").manual_seed(3)
    counts = [2000, 1, 1737, 300]
    R = sum(counts)
    logits = torch.randn((R, 3), device="cuda", generator=g).requires_grad_(True)
    logits.data[5] = logits.data[2]                       # duplicated rows: the first extremum must win
    logits.data[2050] = logits.data[2040]
    rois = torch.zeros((R, 5), device="cuda")
    rois[:, 0] = torch.repeat_interleave(torch.arange(4, device="cuda"), torch.tensor(counts, device="cuda")) + 3.0
    labels = torch.tensor([1, 2, 1, 2], dtype=torch.int32, device="cuda")
    for funcs in ([M.get_mal_max_logit, M.get_mal_max_logit], [M.get_mass_max_logit, M.get_mal_max_logit],
                  [M.get_ben_max_logit, M.get_mass_max_logit]):
        want, wscale = M.get_bag_logit(logits, rois[:, 0] - 3, 3, labels, 4, funcs)
        got, gscale = M.get_bag_logit_device(logits, rois[:, 0], 3.0, labels, 4, funcs)   # strided column
        assert torch.equal(got, want) and torch.allclose(gscale, wscale)
    got.sum().backward()                                  # differentiable gather
    assert logits.grad is not None and int((logits.grad != 0).any(dim=1).sum()) == 4


def test_proposal_layer_from_logits_matches_softmax_path(torch_cuda):
    """f2: the fused reshape->softmax->reshape + proposal layer against the unfused chain."""
    torch = torch_cuda
    from wssdl_bus_amd.networks.network import Network
    from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import proposal_layer, proposal_layer_from_score
    gen = torch.Generator("cuda").manual_seed(11)
    N, H, W, A = 2, 38, 63, 9
    score = torch.randn((N, H, W, 2 * A), device="cuda", generator=gen)
    pred = 0.2 * torch.randn((N, H, W, 4 * A), device="cuda", generator=gen)
    info = torch.tensor([[600, 1000, 1.0, 1], [584, 1000, 2.0, 2]], device="cuda")
    net = Network({'rpn_cls_score': score})
    (net.feed('rpn_cls_score').reshape_layer(2, name='rpn_cls_score_reshape').softmax(name='rpn_cls_prob')
        .reshape_layer(2 * A, name='rpn_cls_prob_reshape'))
    ref = proposal_layer(net.get_output('rpn_cls_prob_reshape').contiguous(), pred, info, True, False)
    got = proposal_layer_from_score(score, pred, info, True, False)
    assert abs(ref.shape[0] - got.shape[0]) <= 4
    m = min(ref.shape[0], got.shape[0])
    same = (ref[:m] - got[:m]).abs().amax(dim=1) <= 1e-3
    assert same.float().mean().item() >= 0.99          # softmax differs by an ulp between the two paths


def test_nms_large_n_uses_global_kept_list(torch_cuda):
    """n > ~15k with max_keep = n: the sweep's kept list lives in global scratch, not LDS."""
    from wssdl_bus_amd.nms.hip_nms import hip_nms
    rs = np.random.RandomState(12)
    n = 20000
    c = rs.uniform(0, 3000, size=(n, 2))
    wh = rs.uniform(10, 80, size=(n, 2))
    d = np.hstack((c, c + wh, rs.permutation(n)[:, None] / float(n))).astype(np.float32)
    assert hip_nms(d, 0.5) == O.nms(d, 0.5)


@pytest.mark.gpu
def test_nms_pipelined_sweep_at_its_size_limits(torch_cuda):
    """The role-pipelined sweep takes up to 320 chunks (a helper lane per chunk of the column summaries, five
    summary words per scribe lane) and 2240 kept boxes: sizes just inside and just outside both limits (outside,
    the general sweep runs), and candidate counts that are not a multiple of the 16-word row pitch."""
    from wssdl_bus_amd.nms.hip_nms import hip_nms
    rs = np.random.RandomState(13)
    for n, keep in ((20480, 1500), (20479, 2240), (20481, 1500), (16385, 2241), (4100, 2000)):
        c = rs.uniform(0, 3000, size=(n, 2))
        wh = rs.uniform(10, 90, size=(n, 2))
        d = np.hstack((c, c + wh, rs.permutation(n)[:, None] / float(n))).astype(np.float32)
        assert hip_nms(d, 0.5, max_keep=keep) == O.nms(d, 0.5)[:keep], (n, keep)


# ---------------------------------------------------------------- device RoI sampling ---
@pytest.mark.gpu
def test_proposal_target_device_sampling(torch_cuda):
    """cfg.SAMPLING_RNG='device' (wssdl_roi_sample_device): not the reference's random stream,
    so the check is structural -- quotas, thresholds, no duplicates, reproducibility -- and the
    labels / targets of the drawn rows are compared with the oracle's formulas."""
    import torch
    np_oracle = O
    from wssdl_bus_amd.fast_rcnn.config import cfg
    from wssdl_bus_amd.rpn_msr import proposal_target_layer_tf_bus as ptl
    g = load_golden("proposal_target")
    rois, gt, ng = g["rois_in"], g["gt_boxes"], g["num_gt"]
    n_img = gt.shape[0]
    dev = torch.device("cuda", 0)
    rois_d = torch.from_numpy(rois).to(dev)
    gt_d, ng_d = torch.from_numpy(gt).to(dev), torch.from_numpy(ng.astype(np.int32)).to(dev)
    old = cfg.SAMPLING_RNG, cfg.DEVICE_RNG_SEED
    cfg.SAMPLING_RNG = "device"
    try:
        runs = []
        for rep in range(3):
            cfg.DEVICE_RNG_SEED = 11 if rep < 2 else 12
            ptl._device_calls[0] = 0
            o = ptl.proposal_target_layer(rois_d, gt_d, ng_d, 3, True, False)
            runs.append([t.cpu().numpy() for t in o])
        a, b, c = runs
        for x, y in zip(a, b):
            assert np.array_equal(x, y)                      # same seed -> same rows
        assert not np.array_equal(a[0], c[0])                # another seed -> another draw
        out_rois, labels, tg, inw, outw = a
        rpi = int(cfg.TRAIN.BATCH_SIZE)
        fg_rpi = int(np.round(cfg.TRAIN.FG_FRACTION * rpi))
        row = 0
        for i in range(n_img):
            npos = int(np.sum(gt[i, :ng[i], 4] != 0))
            cand = np.vstack([rois[rois[:, 0] == i],
                              np.hstack([np.full((npos, 1), i, np.float32), gt[i, :npos, :4]])])
            ov = np_oracle.c_oracle.bbox_overlaps(cand[:, 1:5].astype(np.float64),
                                                  gt[i, :npos, :4].astype(np.float64))
            mo, am = ov.max(axis=1), ov.argmax(axis=1)
            n_fg_have = int(np.sum(mo >= cfg.TRAIN.FG_THRESH))
            n_bg_have = int(np.sum((mo < cfg.TRAIN.BG_THRESH_HI) & (mo >= cfg.TRAIN.BG_THRESH_LO)))
            n_fg = min(fg_rpi, n_fg_have)
            n_bg = min(rpi - n_fg, n_bg_have)
            blk = out_rois[row:row + n_fg + n_bg]
            assert np.all(blk[:, 0] == i)
            # every drawn row is a candidate of this image, none twice
            key = {tuple(r) for r in cand.tolist()}
            assert all(tuple(r) in key for r in blk.tolist())
            lookup = {}
            for j, r in enumerate(cand.tolist()):
                lookup.setdefault(tuple(r), j)
            idx = np.array([lookup[tuple(r)] for r in blk.tolist()])
            # duplicates among candidates share coordinates, hence overlap and targets
            assert np.all(mo[idx[:n_fg]] >= cfg.TRAIN.FG_THRESH)
            assert np.all((mo[idx[n_fg:]] < cfg.TRAIN.BG_THRESH_HI) & (mo[idx[n_fg:]] >= cfg.TRAIN.BG_THRESH_LO))
            lab = labels[row:row + n_fg + n_bg, 0]
            assert np.array_equal(lab[:n_fg], gt[i, am[idx[:n_fg]], 4])
            assert np.all(lab[n_fg:] == 0)
            # targets of the fg rows: bbox_transform in f32, expanded at 4*cls
            t = np_oracle.bbox_transform(blk[:n_fg, 1:5], gt[i, am[idx[:n_fg]], :4]).astype(np.float32)
            for q in range(n_fg):
                cls = int(lab[q])
                e = np.zeros(12, np.float32)
                e[4 * cls:4 * cls + 4] = t[q]
                assert ulp_diff_f32(tg[row + q], e).max() <= 4
                w = np.zeros(12, np.float32)
                w[4 * cls:4 * cls + 4] = 1
                assert np.array_equal(inw[row + q], w) and np.array_equal(outw[row + q], w)
            assert not tg[row + n_fg:row + n_fg + n_bg].any()
            row += n_fg + n_bg
        assert row == out_rois.shape[0]
        # uniformity smoke: over many seeds every fg candidate of image 0 is drawn sometimes
        cfg.DEVICE_RNG_SEED = 5
        seen = set()
        for rep in range(40):
            o = ptl.proposal_target_layer(rois_d, gt_d, ng_d, 3, True, False)
            r0 = o[0].cpu().numpy()
            l0 = o[1].cpu().numpy()[:, 0]
            seen |= {tuple(r) for r in r0[(r0[:, 0] == 0) & (l0 > 0)].tolist()}
        i = 0
        npos = int(np.sum(gt[i, :ng[i], 4] != 0))
        cand = np.vstack([rois[rois[:, 0] == i],
                          np.hstack([np.full((npos, 1), i, np.float32), gt[i, :npos, :4]])])
        mo = np_oracle.c_oracle.bbox_overlaps(cand[:, 1:5].astype(np.float64),
                                              gt[i, :npos, :4].astype(np.float64)).max(axis=1)
        fg_all = {tuple(r) for r in cand[mo >= cfg.TRAIN.FG_THRESH].tolist()}
        if len(fg_all) > fg_rpi:
            assert len(seen) > fg_rpi                        # not always the same subset
        assert seen <= fg_all
    finally:
        cfg.SAMPLING_RNG, cfg.DEVICE_RNG_SEED = old


@pytest.mark.gpu
def test_proposal_target_device_sampling_with_interleaved_images(torch_cuda):
    """The device sampler walks the span of the candidate list that holds its image's rows (one stretch when the
    proposal blob is ordered by image); with the images' rows INTERLEAVED the spans overlap and cover nearly the
    whole list -- quotas, thresholds, membership and the absence of duplicates must not change."""
    import torch
    from wssdl_bus_amd.fast_rcnn.config import cfg
    from wssdl_bus_amd.rpn_msr import proposal_target_layer_tf_bus as ptl
    g = load_golden("proposal_target")
    rois, gt, ng = g["rois_in"], g["gt_boxes"], g["num_gt"]
    rois = rois[np.random.RandomState(3).permutation(rois.shape[0])]
    n_img = gt.shape[0]
    dev = torch.device("cuda", 0)
    old = cfg.SAMPLING_RNG, cfg.DEVICE_RNG_SEED
    cfg.SAMPLING_RNG, cfg.DEVICE_RNG_SEED = "device", 21
    try:
        o = ptl.proposal_target_layer(torch.from_numpy(rois).to(dev), torch.from_numpy(gt).to(dev),
                                      torch.from_numpy(ng.astype(np.int32)).to(dev), 3, True, False)
        out_rois, labels = o[0].cpu().numpy(), o[1].cpu().numpy()[:, 0]
        rpi = int(cfg.TRAIN.BATCH_SIZE)
        fg_rpi = int(np.round(cfg.TRAIN.FG_FRACTION * rpi))
        row = 0
        for i in range(n_img):
            npos = int(np.sum(gt[i, :ng[i], 4] != 0))
            cand = np.vstack([rois[rois[:, 0] == i],
                              np.hstack([np.full((npos, 1), i, np.float32), gt[i, :npos, :4]])])
            mo = O.c_oracle.bbox_overlaps(cand[:, 1:5].astype(np.float64), gt[i, :npos, :4].astype(np.float64)).max(axis=1)
            n_fg = min(fg_rpi, int(np.sum(mo >= cfg.TRAIN.FG_THRESH)))
            n_bg = min(rpi - n_fg, int(np.sum((mo < cfg.TRAIN.BG_THRESH_HI) & (mo >= cfg.TRAIN.BG_THRESH_LO))))
            blk = out_rois[row:row + n_fg + n_bg]
            assert np.all(blk[:, 0] == i)
            lookup = {}
            for j, r in enumerate(cand.tolist()):
                lookup.setdefault(tuple(r), []).append(j)
            idx = []
            for r in blk.tolist():
                assert tuple(r) in lookup
                idx.append(lookup[tuple(r)][0])
            idx = np.array(idx)
            assert np.all(mo[idx[:n_fg]] >= cfg.TRAIN.FG_THRESH) and np.all(labels[row:row + n_fg] > 0)
            assert np.all((mo[idx[n_fg:]] < cfg.TRAIN.BG_THRESH_HI) & (mo[idx[n_fg:]] >= cfg.TRAIN.BG_THRESH_LO))
            # no candidate drawn more often than it occurs in the list
            from collections import Counter
            drawn = Counter(tuple(r) for r in blk.tolist())
            assert all(drawn[k] <= len(lookup[k]) for k in drawn)
            row += n_fg + n_bg
        assert row == out_rois.shape[0]
    finally:
        cfg.SAMPLING_RNG, cfg.DEVICE_RNG_SEED = old


# ----------------------------------------------------------- NMS: suppression chains ---
def test_nms_suppression_chains(torch_cuda):
    """Worst cases for the sweep's fixed-point resolver: inside a 64-box chunk every box
    suppresses the next one but not the one after (the kept set alternates and each round of
    K <- cand & ~ballot(T & K) only settles one more position), chains that run across chunk
    borders, and dense clusters where one box suppresses a whole chunk."""
    from wssdl_bus_amd.nms.hip_nms import hip_nms
    # chain: unit-height boxes of width 100 shifted by 15 px: IoU(i, i+1) = 86/116 = 0.741 (> 0.7),
    # IoU(i, i+2) = 71/131 = 0.54
    for n in (64, 65, 200, 1000):
        x1 = 15.0 * np.arange(n)
        d = np.stack([x1, np.zeros(n), x1 + 100.0, np.full(n, 50.0), 1.0 - np.arange(n) / float(n + 1)],
                     axis=1).astype(np.float32)
        k = hip_nms(d, 0.7)
        assert k == O.nms(d, 0.7), n
        assert k == list(range(0, n, 2))                       # the alternating pattern
        # the same boxes in reversed score order, and with a max_keep cut
        d2 = d.copy()
        d2[:, 4] = d[::-1, 4]
        assert hip_nms(d2, 0.7) == O.nms(d2, 0.7)
        assert hip_nms(d, 0.7, max_keep=10) == O.nms(d, 0.7)[:10]
    # dense clusters: 40 clusters of 100 near-identical boxes, random scores
    rs = np.random.RandomState(5)
    centers = rs.uniform(100, 900, size=(40, 2))
    c = np.repeat(centers, 100, axis=0) + rs.uniform(-3, 3, size=(4000, 2))
    wh = 120.0 + rs.uniform(-4, 4, size=(4000, 2))
    d = np.hstack((c - wh / 2, c + wh / 2, rs.permutation(4000)[:, None] / 4000.0)).astype(np.float32)
    for th in (0.5, 0.7, 0.9):
        assert hip_nms(d, th) == O.nms(d, th), th
    # 12000 boxes in a tight field: long runs of chunks with few survivors
    n = 12000
    c = rs.uniform(0, 120, size=(n, 2)) * [1.0, 0.6]
    wh = np.exp(rs.normal(4.5, 0.4, size=(n, 2)))
    d = np.hstack((c - wh / 2, c + wh / 2, rs.permutation(n)[:, None] / float(n))).astype(np.float32)
    assert hip_nms(d, 0.7, max_keep=2000) == O.nms(d, 0.7)[:2000]


# ------------------------------------------------ plumbing: fused row batch-norm (+ReLU) ---
def test_fused_row_batchnorm_matches_torch_ops(torch_cuda):
    """csrc/plumbing/rowbn.hip against the stock-PyTorch formulation of the same layer
    (float tolerance: different summation order, f64 column sums)."""
    import torch
    from wssdl_bus_amd.networks import _plumbing
    from wssdl_bus_amd.networks.roi_head import RowBatchNorm, _RowBatchNormFn
    assert _plumbing.lib() is not None
    g = torch.Generator(device="cuda").manual_seed(0)
    for M, C in ((1000, 512), (4097, 2048), (333, 64), (50000, 1024), (7, 256)):
        x = (torch.randn((M, C), device="cuda", generator=g) * 2.0 + 0.5)
        w = torch.rand((C,), device="cuda", generator=g) + 0.5
        b = torch.randn((C,), device="cuda", generator=g)
        dy = torch.randn((M, C), device="cuda", generator=g)
        for relu in (False, True):
            assert _plumbing.usable(x)
            bn = RowBatchNorm(C).cuda()
            with torch.no_grad():
                bn.weight.copy_(w)
                bn.bias.copy_(b)
            xa = x.clone().requires_grad_(True)
            ya = bn(xa, relu=relu)
            ya.backward(dy)
            xb = x.clone().requires_grad_(True)
            wb, bb = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            yb, mean, var = _RowBatchNormFn.apply(xb, wb, bb, bn.eps)
            if relu:
                yb = torch.relu(yb)
            yb.backward(dy)
            def close(a, e, tol):
                return float((a - e).abs().max()) <= tol * (1.0 + float(e.abs().max()))
            assert close(ya, yb, 1e-5), (M, C, relu)
            assert close(xa.grad, xb.grad, 2e-4), (M, C, relu, float((xa.grad - xb.grad).abs().max()))
            assert close(bn.weight.grad, wb.grad, 2e-4), (M, C, relu)
            assert close(bn.bias.grad, bb.grad, 2e-4), (M, C, relu)
            assert close(bn.running_mean, 0.01 * mean, 1e-5)
            # inference statistics path
            bn.eval()
            with torch.no_grad():
                ye = bn(x, relu=relu)
                scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
                ref = x * scale + (bn.bias - bn.running_mean * scale)
                ref = torch.relu(ref) if relu else ref
            assert close(ye, ref, 1e-5)
    # unsupported width falls back to the stock ops
    assert not _plumbing.usable(torch.zeros((10, 96), device="cuda"))


def test_fused_batchnorm2d_channels_last_matches_stock(torch_cuda):
    import torch
    from wssdl_bus_amd.networks.backbones import BatchNormAct2d
    g = torch.Generator(device="cuda").manual_seed(1)
    for shape in ((2, 64, 37, 50), (3, 256, 10, 17)):
        x = torch.randn(shape, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(shape, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
        for relu in (False, True):
            a = BatchNormAct2d(shape[1], eps=1e-3, momentum=0.01).cuda()
            b = torch.nn.BatchNorm2d(shape[1], eps=1e-3, momentum=0.01).cuda()
            with torch.no_grad():
                a.weight.uniform_(0.5, 1.5); a.bias.normal_()
                b.weight.copy_(a.weight); b.bias.copy_(a.bias)
            xa = x.clone().requires_grad_(True)
            xb = x.clone().requires_grad_(True)
            ya = a(xa, relu=relu)
            yb = b(xb)
            yb = torch.relu(yb) if relu else yb
            assert ya.shape == yb.shape and ya.is_contiguous(memory_format=torch.channels_last)
            ya.backward(dy)
            yb.backward(dy)
            for u, v, tol in ((ya, yb, 1e-5), (xa.grad, xb.grad, 2e-4), (a.weight.grad, b.weight.grad, 2e-4),
                              (a.bias.grad, b.bias.grad, 2e-4), (a.running_mean, b.running_mean, 1e-5),
                              (a.running_var, b.running_var, 1e-5)):
                assert float((u - v).abs().max()) <= tol * (1.0 + float(v.abs().max())), (shape, relu)
            assert int(a.num_batches_tracked) == 1


def test_proposal_target_device_sampling_short_image(torch_cuda):
    """Fewer background candidates than the quota: the reference returns fewer than rois_per_image
    rows for that image; the device path keeps the shape fixed (no read-back of the counts) and fills
    the image's slot with padding rows (-1,0,0,0,0), label -1, zero weights, which RoI pooling, the
    losses and the MIL selection ignore."""
    import torch
    from wssdl_bus_amd.fast_rcnn.config import cfg
    from wssdl_bus_amd.rpn_msr import proposal_target_layer_tf_bus as ptl
    dev = torch.device("cuda", 0)
    gt = np.zeros((2, 20, 5), np.float32)
    gt[0, 0] = [100, 100, 300, 300, 1]
    gt[1, 0] = [50, 60, 250, 220, 2]
    ng = np.array([1, 1], np.int32)
    rs = np.random.RandomState(0)
    # image 0: 200 rois all close to its gt box (all fg, no bg); image 1: 40 fg-like + 500 far away
    j0 = np.hstack([np.zeros((200, 1)), np.array([100, 100, 300, 300]) + rs.uniform(-6, 6, (200, 4))])
    j1 = np.hstack([np.ones((40, 1)), np.array([50, 60, 250, 220]) + rs.uniform(-6, 6, (40, 4))])
    far = rs.uniform(400, 900, (500, 2))
    f1 = np.hstack([np.ones((500, 1)), far, far + rs.uniform(20, 80, (500, 2))])
    rois = np.vstack([j0, j1, f1]).astype(np.float32)
    old = cfg.SAMPLING_RNG
    cfg.SAMPLING_RNG = "device"
    try:
        o = ptl.proposal_target_layer(torch.from_numpy(rois).to(dev), torch.from_numpy(gt).to(dev),
                                      torch.from_numpy(ng).to(dev), 3, True, False)
        out_rois, labels = o[0].cpu().numpy(), o[1].cpu().numpy()[:, 0]
        fg_rpi = int(np.round(cfg.TRAIN.FG_FRACTION * cfg.TRAIN.BATCH_SIZE))
        rpi = int(cfg.TRAIN.BATCH_SIZE)
        assert out_rois.shape == (2 * rpi, 5) and o[2].shape == (2 * rpi, 12)
        n0 = int(np.sum(out_rois[:, 0] == 0))
        n1 = int(np.sum(out_rois[:, 0] == 1))
        assert n0 == fg_rpi                      # all candidates of image 0 are fg: 32 fg, 0 bg
        assert n1 == rpi                         # image 1: 32 fg + 96 bg
        assert np.all(out_rois[:n0, 0] == 0) and np.all(labels[:n0] == 1)
        pad = slice(n0, rpi)                     # the rest of image 0's slot is padding
        assert np.all(out_rois[pad, 0] == -1) and not out_rois[pad, 1:].any() and np.all(labels[pad] == -1)
        assert not o[2].cpu().numpy()[pad].any() and not o[3].cpu().numpy()[pad].any() and not o[4].cpu().numpy()[pad].any()
        assert np.all(labels[rpi:rpi + fg_rpi] == 2) and np.all(labels[rpi + fg_rpi:] == 0)
        # the losses leave the padding rows out: CE over 32 + 128 rows, box loss averaged over them
        from wssdl_bus_amd.fast_rcnn.train_bus import rcnn_box_loss, rcnn_cls_loss
        cls = torch.randn((2 * rpi, 3), device=dev)
        live = labels >= 0
        want = torch.nn.functional.cross_entropy(cls[torch.from_numpy(live).to(dev)],
                                                 torch.from_numpy(labels[live]).long().to(dev))
        assert abs(float(rcnn_cls_loss(cls, o[1])) - float(want)) < 1e-6
        bp = torch.randn((2 * rpi, 12), device=dev)
        per = (o[4] * (o[3] * (bp - o[2]).abs())).sum(1)
        assert abs(float(rcnn_box_loss(bp, o)) - float(per.sum() / int(live.sum()))) < 1e-6
    finally:
        cfg.SAMPLING_RNG = old


def test_im2col3x3_matches_unfold(torch_cuda):
    """csrc/plumbing/im2col.hip against the stock pad -> unfold -> permute route (exact: copies
    forward; the adjoint adds at most 9 terms in a fixed order, compared with a tolerance)."""
    import torch
    import torch.nn.functional as F
    from wssdl_bus_amd.networks import _plumbing
    from wssdl_bus_amd.networks.backbones import _same_pad
    g = torch.Generator(device="cuda").manual_seed(2)
    for (r, h, w, c, s) in ((5, 7, 7, 64, 2), (3, 4, 4, 512, 1), (2, 5, 6, 32, 1), (4, 7, 7, 16, 1), (1, 1, 1, 8, 1)):
        x = torch.randn((r, h, w, c), device="cuda", generator=g)
        pt, pb = _same_pad(h, 3, s)
        pl, pr = _same_pad(w, 3, s)
        oh, ow = -(-h // s), -(-w // s)
        xa = x.clone().requires_grad_(True)
        assert _plumbing.im2col_usable(xa)
        ca = _plumbing.Im2Col3x3Fn.apply(xa, s, oh, ow, pt, pl)
        xb = x.clone().requires_grad_(True)
        p = F.pad(xb, (0, 0, pl, pr, pt, pb)).unfold(1, 3, s).unfold(2, 3, s)
        cb = p.permute(0, 1, 2, 4, 5, 3).reshape(-1, 9 * c)
        assert ca.shape == cb.shape and torch.equal(ca, cb), (r, h, w, c, s)
        d = torch.randn(ca.shape, device="cuda", generator=g)
        ca.backward(d)
        cb.backward(d)
        assert float((xa.grad - xb.grad).abs().max()) <= 1e-5 * (1.0 + float(xb.grad.abs().max()))


def test_proposal_layer_two_pass_nms_equals_one_pass_and_oracle(torch_cuda):
    """The proposal layer's NMS runs as a probe over the first candidates plus, for the images the
    probe cannot settle, a completion pass (nms.hip: launch_nms_two_pass).  (a) Heavy suppression:
    zero deltas make the proposals the anchors themselves, neighbours suppress each other, far
    fewer than 2000 survive among the first 8192 candidates -> the completion pass runs; boxes are
    exact (exp(0) = 1), so the rows must equal the oracle's.  (b) On the golden inputs both forms
    give identical blobs.  (c) Mixed batch: one image settled by the probe, one not."""
    import os
    torch = torch_cuda
    from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import proposal_layer
    rs = np.random.RandomState(23)
    N, H, W, A = 2, 38, 63, 9
    info = np.array([[600, 1000, 1.0, 1], [600, 1000, 1.0, 2]], np.float32)
    prob = np.zeros((N, H, W, 2 * A), np.float32)
    prob[..., A:] = rs.permutation(N * H * W * A).reshape(N, H, W, A).astype(np.float32) / (N * H * W * A) * 0.9 + 0.05
    prob[..., :A] = 1 - prob[..., A:]
    pred = np.zeros((N, H, W, 4 * A), np.float32)
    pred[1] = rs.normal(0, 0.3, size=(H, W, 4 * A)).astype(np.float32)          # image 1: ordinary proposals

    from wssdl_bus_amd.fast_rcnn.config import cfg
    old_thresh = cfg.TRAIN.RPN_NMS_THRESH
    cfg.TRAIN.RPN_NMS_THRESH = 0.3             # anchors of neighbouring cells suppress each other

    from wssdl_bus_amd import _lib

    def run(one_pass):
        with _lib.tuned(nms_one_pass=int(one_pass)):
            return proposal_layer(prob, pred, info, True, False)
    try:
        two, one = run(False), run(True)
    finally:
        cfg.TRAIN.RPN_NMS_THRESH = old_thresh
    assert np.array_equal(two, one)
    n0 = int((two[:, 0] == 0).sum())
    assert 0 < n0 < 2000                       # image 0 needed every candidate (completion pass)
    want = O.proposal_layer(prob[:1], pred[:1], info[:1], True, False, STRIDE, SCALES,
                            cfg=dict(TRAIN_RPN_NMS_THRESH=0.3))
    assert np.array_equal(two[:n0], want)
    g = load_golden("proposal_layer")
    for case in ("res_38x63_train", "res_63x100_test"):
        prob_g, pred_g, info_g = g[case + "/prob"], g[case + "/pred"], g[case + "/im_info"]
        train = bool(g[case + "/is_training"])
        with _lib.tuned(nms_one_pass=1):
            a = proposal_layer(prob_g, pred_g, info_g, train, False)
        assert np.array_equal(proposal_layer(prob_g, pred_g, info_g, train, False), a)


# ------------------------------------------------ NMS: the geometric prefilter's corners ---
def test_nms_prefilter_thresholds_and_degenerate_boxes(torch_cuda):
    """The mask kernel only runs the exact overlap rule (cpu_nms.pyx:43-66) on pairs whose centres
    are close enough to reach the threshold at all (0.25 <= thresh < 1).  Cases that sit on that
    filter's edges: thresholds at and around the range limits, pairs whose overlap equals the
    threshold, boxes with non-positive width or height (negative 'areas': the reference still
    computes with them), exact duplicates, boxes far from the origin (centre rounding), and the
    same workspace re-used with a different threshold (words of the earlier call that are zero
    now are not rewritten and must not be read)."""
    from wssdl_bus_amd.nms.hip_nms import hip_nms
    rs = np.random.RandomState(11)
    n = 3000
    c = rs.uniform(0, 600, size=(n, 2))
    wh = np.exp(rs.uniform(np.log(4), np.log(400), size=(n, 2)))
    b = np.hstack((c - wh / 2, c + wh / 2))
    # degenerate rows: swapped corners, zero and negative extents
    bad = rs.choice(n, 200, replace=False)
    b[bad[:80], 2] = b[bad[:80], 0] - rs.uniform(0, 30, 80)            # x2 < x1
    b[bad[80:140], 3] = b[bad[80:140], 1] - 1.0                        # height exactly 0
    b[bad[140:], 2:] = b[bad[140:], :2] - rs.uniform(1, 50, (60, 2))   # both negative
    # duplicates and near-duplicates of other rows
    b[100:200] = b[0:100]
    b[200:300] = b[0:100] + rs.uniform(-1, 1, (100, 4))
    d = np.hstack((b, rs.permutation(n)[:, None] / float(n))).astype(np.float32)
    for th in (0.95, 0.3, 0.25, 0.2499, 0.5, 0.7, 0.999, 1.0, 1.0001, 0.05):
        assert hip_nms(d, th) == O.nms(d, th), th
    # far from the origin: centres near 1e5 px, where f32 spacing is ~0.008 px
    d2 = d.copy()
    d2[:, :4] += np.float32(1e5)
    for th in (0.3, 0.7):
        assert hip_nms(d2, th) == O.nms(d2, th), th
    # pairs whose overlap is exactly the threshold: width-w boxes shifted so that inter/union = 1/2, 1/3, 3/4
    rows = []
    for w, s in ((30, 10), (40, 20), (70, 10), (9, 3), (64, 32), (100, 25)):
        rows += [[0, 0, w - 1, 9], [s, 0, s + w - 1, 9]]
    e = np.array(rows, np.float32)
    e[:, [1, 3]] += 20.0 * (np.arange(len(e)) // 2)[:, None]
    e = np.hstack((e, np.linspace(1, 0.5, len(e), dtype=np.float32)[:, None]))
    for th in (0.5, 1.0 / 3.0, 0.75, 0.6, 0.25, float(np.float32(0.5)), float(np.float32(1.0 / 3.0))):
        assert hip_nms(e, th) == O.nms(e, th), th


def test_nms_prefilter_box_families_vs_oracle(torch_cuda):
    """Whole keep-lists against the oracle (cpu_nms.pyx:17-68) for box populations that stress the
    centre-distance prefilter in different ways: integer coordinates (overlaps that equal simple
    thresholds exactly), boxes clipped to the image border (shared edges, extreme aspect ratios),
    tiny boxes next to image-sized ones, and a proposal-like mix at 12000 candidates."""
    from wssdl_bus_amd.nms.hip_nms import hip_nms
    rs = np.random.RandomState(21)

    def with_scores(b):
        n = len(b)
        return np.hstack((b, rs.permutation(n)[:, None] / float(n))).astype(np.float32)

    # integer boxes on a coarse lattice: many pairs with IoU exactly 1/2, 1/3, 2/3, 3/4 ...
    n = 4000
    x1 = rs.randint(0, 40, n) * 8.0
    y1 = rs.randint(0, 30, n) * 8.0
    w = rs.choice([16, 24, 32, 48, 64], n).astype(np.float64)
    h = rs.choice([16, 24, 32, 48, 64], n).astype(np.float64)
    lattice = with_scores(np.stack([x1, y1, x1 + w - 1, y1 + h - 1], axis=1))
    for th in (0.5, 1.0 / 3.0, 2.0 / 3.0, 0.75, 0.7, 0.25):
        assert hip_nms(lattice, th) == O.nms(lattice, th), ("lattice", th)
    # clipped to a 1000 x 600 image: edges pile up on the border, some boxes become slivers
    c = rs.uniform(-100, 1100, size=(5000, 2)) * [1.0, 0.6]
    wh = np.exp(rs.uniform(np.log(8), np.log(900), size=(5000, 2)))
    b = np.hstack((c - wh / 2, c + wh / 2))
    b[:, [0, 2]] = np.clip(b[:, [0, 2]], 0, 999)
    b[:, [1, 3]] = np.clip(b[:, [1, 3]], 0, 599)
    clipped = with_scores(b)
    for th in (0.7, 0.3, 0.5):
        assert hip_nms(clipped, th) == O.nms(clipped, th), ("clipped", th)
    # tiny boxes inside image-sized ones
    tiny = np.hstack((rs.uniform(0, 990, (3000, 1)), rs.uniform(0, 590, (3000, 1))))
    tiny = np.hstack((tiny, tiny + rs.uniform(0, 6, (3000, 2))))
    huge = np.hstack((rs.uniform(0, 30, (500, 2)), rs.uniform(960, 999, (500, 1)), rs.uniform(560, 599, (500, 1))))
    mixed = with_scores(np.vstack((tiny, huge)))
    for th in (0.7, 0.3, 0.9):
        assert hip_nms(mixed, th) == O.nms(mixed, th), ("mixed", th)
    # proposal-like population at the training size, with the training cut
    n = 12000
    c = rs.uniform(0, 1000, size=(n, 2)) * [1.0, 0.6]
    s = np.exp(rs.normal(np.log(150), 0.7, size=(n, 1)))
    ar = np.exp(rs.normal(0, 0.5, size=(n, 1)))
    wh = np.hstack((s * np.sqrt(ar), s / np.sqrt(ar)))
    b = np.hstack((c - wh / 2, c + wh / 2))
    b[:, [0, 2]] = np.clip(b[:, [0, 2]], 0, 999)
    b[:, [1, 3]] = np.clip(b[:, [1, 3]], 0, 599)
    prop = with_scores(b)
    want = O.nms(prop, 0.7)
    assert hip_nms(prop, 0.7) == want
    assert hip_nms(prop, 0.7, max_keep=2000) == want[:2000]


def test_fused_mask_sweep_launch_equals_two_launches(torch_cuda):
    """One-pass NMS of the proposal layer as ONE launch (sweep workgroups trailing the mask workgroups
    row block by row block, nms.hip: nms_mask_sweep_fused_kernel) against the two launches it replaces:
    identical blobs on the golden train inputs, on 8 images of synthetic RPN maps (heavy and light
    suppression, an image with no candidates at all) and with the heavy-suppression maps of the two-pass
    test; the golden rows themselves are pinned by test_proposal_layer_golden with the fused default."""
    torch = torch_cuda
    from wssdl_bus_amd import _lib
    from wssdl_bus_amd.fast_rcnn.config import cfg
    from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import proposal_layer
    assert _lib.get_tuning("nms_fused") == 1

    def both(prob, pred, info, train=True):
        with _lib.tuned(nms_fused=1):
            a = proposal_layer(prob, pred, info, train, False)
        with _lib.tuned(nms_fused=0):
            b = proposal_layer(prob, pred, info, train, False)
        assert np.array_equal(a, b)
        return a

    g = load_golden("proposal_layer")
    both(g["res_38x63_train/prob"], g["res_38x63_train/pred"], g["res_38x63_train/im_info"])
    rs = np.random.RandomState(5)
    N, H, W, A = 8, 38, 63, 9
    info = np.tile(np.array([[600, 1000, 1.0, 1]], np.float32), (N, 1))
    logits = rs.normal(size=(N, H, W, A, 2)).astype(np.float32)
    e = np.exp(logits - logits.max(-1, keepdims=True))
    p = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    prob = np.concatenate((p[..., 0], p[..., 1]), axis=-1)
    pred = rs.normal(0, 0.2, size=(N, H, W, 4 * A)).astype(np.float32)
    pred[1] = 0.0                                   # the anchors themselves: neighbours suppress each other
    pred[2] *= 4.0                                  # wild boxes, mostly clipped to the image border
    info[3, :2] = [8, 8]                            # every box filtered out (min size): no candidates
    old = cfg.TRAIN.RPN_NMS_THRESH
    seen = set()
    try:
        for t in (0.7, 0.3):
            cfg.TRAIN.RPN_NMS_THRESH = t
            out = both(prob, pred, info)
            counts = [int((out[:, 0] == i).sum()) for i in range(N)]
            assert counts[3] == 0 and all(0 < c <= 2000 for i, c in enumerate(counts) if i != 3)
            seen.update(counts)
            for _ in range(3):                      # repeatable (the waits never change the result)
                with _lib.tuned(nms_fused=1):
                    assert np.array_equal(proposal_layer(prob, pred, info, True, False), out)
    finally:
        cfg.TRAIN.RPN_NMS_THRESH = old
    assert 2000 in seen and any(0 < c < 2000 for c in seen)      # sweeps that stop early and sweeps that walk every chunk


def test_fused_nms_time_out_is_reported_not_swallowed(torch_cuda):
    """The sweep of the fused launch gives up when the mask blocks it waits for make no progress (a GPU held by
    another process): the image then reports roi count -1 (WSSDL_NMS_TIMED_OUT), never a silent 0.  Forced with
    the fault injector `nms_fused_fault` (image 0's segment counts are withheld, wait shortened to 2 ms): the
    blob-building path RECOMPUTES the call with the two-launch NMS (identical results) and warns once, the sync-free
    padded path raises its deferred flag (check_flags raises; poll_flags -- the training loop -- switches the process
    to the two-launch form and carries on), the other images are intact."""
    torch = torch_cuda
    from wssdl_bus_amd import _lib
    from wssdl_bus_amd.fast_rcnn.config import cfg
    from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as rp
    from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import padded_blob, proposal_layer, proposal_layer_padded
    rs = np.random.RandomState(6)
    N, H, W, A = 3, 38, 63, 9
    info = np.tile(np.array([[600, 1000, 1.0, 1]], np.float32), (N, 1))
    logits = rs.normal(size=(N, H, W, A, 2)).astype(np.float32)
    e = np.exp(logits - logits.max(-1, keepdims=True))
    p = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    prob = np.concatenate((p[..., 0], p[..., 1]), axis=-1)
    pred = rs.normal(0, 0.2, size=(N, H, W, 4 * A)).astype(np.float32)
    good = proposal_layer(prob, pred, info, True, False)
    rp.check_flags()
    with _lib.tuned(nms_fused=1, nms_fused_fault=2000):
        rois, counts = proposal_layer_padded(torch.from_numpy(prob).cuda(), torch.from_numpy(pred).cuda(),
                                             torch.from_numpy(info).cuda(), True)
        c = counts.cpu().numpy()
        assert c[0] == -1 and c[1] > 0 and c[2] > 0, c
        for i in (1, 2):            # the other images' sweeps were not disturbed
            assert np.array_equal(rois[i, :c[i], 1:].cpu().numpy(), good[good[:, 0] == i][:, 1:])
        # the reference-shaped call recomputes with mask and sweep as two launches: the two-launch path's rois exactly
        import warnings
        from wssdl_bus_amd.rpn_msr import proposal_layer_tf_bus as plt
        plt._timeout_warned[0] = False
        plt._cooldown[0] = 0
        import time
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            with _lib.tuned(nms_fused_fault=200000):                            # a stall one can time: 0.2 s
                t0 = time.perf_counter()
                again = proposal_layer(prob, pred, info, True, False)
                t1 = time.perf_counter()
                # sticky: the following calls stay on two launches for a cool-down -- ONE stall, not one per call
                assert plt._cooldown[0] == plt.NMS_TIMEOUT_COOLDOWN_CALLS
                again2 = proposal_layer(prob, pred, info, True, False)
                again3 = proposal_layer(prob, pred, info, True, False)
                t2 = time.perf_counter()
            assert t1 - t0 >= 0.2 and t2 - t1 < 0.15, (t1 - t0, t2 - t1)
            assert plt._cooldown[0] == plt.NMS_TIMEOUT_COOLDOWN_CALLS - 2
        assert np.array_equal(again, good) and np.array_equal(again2, good) and np.array_equal(again3, good)
        assert sum("timed out" in str(w.message) for w in caught) == 1          # logged once
        # after the cool-down the fused launch is tried again (and, the fault still injected, times out again)
        plt._cooldown[0] = 1
        assert np.array_equal(proposal_layer(prob, pred, info, True, False), good) and plt._cooldown[0] == 0
        assert np.array_equal(proposal_layer(prob, pred, info, True, False), good)
        assert plt._cooldown[0] == plt.NMS_TIMEOUT_COOLDOWN_CALLS
        plt._cooldown[0] = 0
        assert _lib.get_tuning("nms_wait_us") == 50000
        with _lib.tuned(nms_fused=0):
            assert np.array_equal(proposal_layer(prob, pred, info, True, False), again)
        blob = padded_blob(rois, counts)                       # the form that never reads the counts back
        assert int((blob[:2000, 0] >= 0).sum()) == 0            # image 0: no live rows
        with pytest.raises(_lib.HipCallError, match="NMS sweep"):
            rp.check_flags()
        # the training loop's poll: no raise, the process carries on with the two-launch form
        padded_blob(rois, counts)
        rp.poll_flags()
        torch.cuda.synchronize()
        assert _lib.get_tuning("nms_fused") == 1
        rp.poll_flags()
        assert _lib.get_tuning("nms_fused") == 0
        _lib.set_tuning("nms_fused", 1)
    # the step the poll let through ran without image 0's proposals: counted, and named by the next synchronising check
    # (what the train loop runs before a snapshot)
    assert rp.tainted_steps() == 1
    with pytest.raises(_lib.HipCallError, match="proposals missing"):
        rp.check_flags()
    assert rp.tainted_steps() == 0
    rp.check_flags()                                            # reported once, then clean
    assert np.array_equal(proposal_layer(prob, pred, info, True, False), good)


def test_deferred_flags_poll_raises_one_step_late_and_only_once(torch_cuda):
    """roi_pooling_op.poll_flags (what SolverWrapper._apply calls before every optimiser step): no read-back -- a poll
    starts an async copy of the flags, the NEXT poll looks at it -- so a raised flag surfaces one poll later, names its
    cause, clears exactly the bits it reported and leaves later ones up."""
    torch = torch_cuda
    from wssdl_bus_amd import _lib
    from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as rp
    rp.check_flags()
    rp.poll_flags()
    torch.cuda.synchronize()
    rp.poll_flags()                                                   # nothing up: no raise
    rp.note_roi_counts(torch.tensor([5, -1, 7], dtype=torch.int32, device="cuda"))
    rp.poll_flags()                                                   # starts the copy that holds the flag
    torch.cuda.synchronize()
    rp._flags(torch.device("cuda", torch.cuda.current_device())).flags[0:1].fill_(1)      # a second flag, after the copy
    # the poll does not raise for an NMS time-out: it switches the process to the two-launch NMS (round 5; a run should
    # not die where recomputing is possible) and clears exactly that bit
    saved_fused = _lib.get_tuning("nms_fused")
    _lib.set_tuning("nms_fused", 1)
    rp.poll_flags()
    assert _lib.get_tuning("nms_fused") == 0
    _lib.set_tuning("nms_fused", saved_fused)
    with pytest.raises(_lib.HipCallError, match="15 x 16"):            # the later flag was not wiped by the first report
        rp.check_flags()
    rp.poll_flags()
    torch.cuda.synchronize()
    rp.poll_flags()                                                   # the device flags are clean again
    # ... but the step the poll let through is remembered until a synchronising check has named it (round 6)
    assert rp.tainted_steps() == 1
    with pytest.raises(_lib.HipCallError, match="proposals missing"):
        rp.check_flags()
    rp.check_flags()


def test_post_detections_device_op_edges(torch_cuda):
    """wssdl_post_detections (f3 as one C-ABI call) against the step-by-step form (which
    test_post_detection_nms_matches_oracle pins to the oracle) and against the oracle directly: more
    classes, a class with no row above the score threshold, no cap, a cap whose threshold score is tied
    (`>=` keeps the ties), a single row, no rows."""
    torch = torch_cuda
    from wssdl_bus_amd.fast_rcnn.config import cfg
    from wssdl_bus_amd.fast_rcnn.test_bus import post_detections_device, postprocess_detections
    rs = np.random.RandomState(12)

    def oracle(scores, boxes, K, thresh, cap):
        want = {}
        for j in range(1, K):
            inds = np.where(scores[:, j] > thresh)[0]
            d = np.hstack((boxes[inds, 4 * j:4 * j + 4], scores[inds, j:j + 1])).astype(np.float32)
            want[j] = d[O.nms(d, cfg.TEST.NMS)] if len(d) else d
        alls = np.hstack([want[j][:, 4] for j in range(1, K)]) if K > 1 else np.zeros(0)
        if cap > 0 and len(alls) > cap:
            th = np.sort(alls)[-cap]
            for j in range(1, K):
                want[j] = want[j][want[j][:, 4] >= th]
        return want

    def case(R, K, thresh, cap, tie=False, dead_class=None):
        ctr = rs.uniform(50, 900, size=(R, 1, 2)) * [1.0, 0.6] + rs.normal(0, 6, size=(R, K, 2))
        wh = rs.uniform(30, 220, size=(R, K, 2))
        boxes = np.concatenate((ctr - wh / 2, ctr + wh / 2), axis=2).reshape(R, 4 * K).astype(np.float32)
        scores = rs.uniform(0, 1, size=(R, K)).astype(np.float32)
        if tie and R > 8:
            scores[:8, 1] = np.float32(0.625)                     # eight equal scores around the cap's threshold
            scores[8:, 1] = np.minimum(scores[8:, 1], np.float32(0.6))
            scores[:, 2:] = np.minimum(scores[:, 2:], np.float32(0.5)) if K > 2 else scores[:, 2:]
            boxes[:8, 4:8] = boxes[:8, 4:8] + np.arange(8, dtype=np.float32)[:, None] * 400     # far apart: none suppressed
        if dead_class is not None:
            scores[:, dead_class] = np.float32(0.01)
        st, bt = torch.from_numpy(scores).cuda(), torch.from_numpy(boxes).cuda()
        want = oracle(scores, boxes, K, thresh, cap)
        got = postprocess_detections(st, bt, K, thresh=thresh, max_per_image=cap)
        old = cfg.TEST.FUSED_POST_DETECTIONS
        try:
            cfg.TEST.FUSED_POST_DETECTIONS = False
            slow = postprocess_detections(st, bt, K, thresh=thresh, max_per_image=cap)
        finally:
            cfg.TEST.FUSED_POST_DETECTIONS = old
        def canon(a):                     # rows of EQUAL score come in an order the reference leaves to argsort
            a = np.asarray(a)
            return a[np.lexsort((a[:, 0], a[:, 1], -a[:, 4]))] if tie and len(a) else a
        for j in range(1, K):
            assert np.array_equal(canon(got[j].cpu().numpy()), canon(want[j])), (R, K, cap, j)
            assert np.array_equal(canon(slow[j].cpu().numpy()), canon(want[j])), (R, K, cap, j, "step-by-step")
            g = got[j].cpu().numpy()
            assert np.all(g[:-1, 4] >= g[1:, 4])                  # descending scores either way
        return want

    case(300, 3, 0.05, 300)
    case(300, 3, 0.05, 0)                                  # no cap
    w = case(300, 6, 0.3, 25, dead_class=4)
    assert len(w[4]) == 0 and sum(len(v) for v in w.values()) >= 25
    w = case(120, 3, 0.05, 5, tie=True)                    # cap 5 inside eight tied scores: all eight stay
    assert len(w[1]) == 8 and len(w[2]) == 0
    case(1, 3, 0.05, 10)
    case(700, 2, 0.5, 50)
    # no rows at all: zero counts, nothing launched
    dets, counts = post_detections_device(torch.zeros((0, 3), device="cuda"), torch.zeros((0, 12), device="cuda"), 3)
    assert counts.cpu().tolist() == [0, 0]


def test_order_sort_equals_hand_written_ranking(torch_cuda):
    """The candidates' order by sorted runs + cross ranks (order_sort.hip, the default) and by the hand-written
    select + sample sort (nms.hip): identical blobs and identical sorted candidate
    lists -- train and test mode, 1-8 images, duplicated scores (ties go to the higher index either way), images
    with few or no candidates, scores that follow the anchor index (a run's keys then all fall into one gap of
    its neighbour's: the galloping search), a map with fewer anchors than one run."""
    torch = torch_cuda
    from wssdl_bus_amd import _lib
    from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import proposal_layer_padded
    assert _lib.get_tuning("topk_sort") == 1
    rs = np.random.RandomState(31)
    # (8, 1 and 12 images of this size take the rank kernel with 4, 2 and 8 own keys per thread)
    for N, (H, W), train in ((8, (38, 63), True), (1, (63, 100), False), (3, (37, 62), True), (2, (12, 17), False),
                             (2, (5, 7), True), (12, (38, 63), True)):
        A = 9
        prob = rs.uniform(0.01, 0.99, size=(N, H, W, 2 * A)).astype(np.float32)
        prob[0, :, :, A:] = np.round(prob[0, :, :, A:], 2)                   # image 0: ~100 distinct scores, many ties
        if N >= 8:                                                            # scores rising / falling with the index
            ramp = np.linspace(0.02, 0.98, H * W * A, dtype=np.float32).reshape(H, W, A)
            prob[1, :, :, A:] = ramp
            prob[2, :, :, A:] = ramp[::-1, ::-1, ::-1]
            prob[3, :, :, A:] = np.where(ramp > 0.5, 0.75, 0.25)              # two plateaus
        pred = rs.normal(0, 0.25, size=(N, H, W, 4 * A)).astype(np.float32)
        info = np.tile(np.array([[16 * H - 8, 16 * W - 8, 1.0, 1]], np.float32), (N, 1))
        if N >= 3:
            info[1, :2] = [40, 40]                                            # hardly any box passes the size filter
            info[2, :2] = [8, 8]                                              # none does
        args = [torch.from_numpy(a).cuda() for a in (prob, pred, info)]
        outs = []
        for mode in (1, 0):
            with _lib.tuned(topk_sort=mode):
                outs.append([t.clone() for t in proposal_layer_padded(*args, train, debug=True)])
        for other in outs[1:]:
            for a, b in zip(outs[0], other):
                assert torch.equal(a, b), (N, H, W, train)
        counts = outs[0][1].cpu().tolist()
        if N >= 3:
            assert counts[2] == 0 and 0 < counts[0]
