"""a13: the PyTorch restatement of the multi-task loss against an independent NumPy (f64)
evaluation of the reference's formulas (fast_rcnn/train_bus.py:184-270, :603-678, mil/core.py).
CPU only; the HIP library is not touched."""
import numpy as np
import torch

from wssdl_bus_amd.fast_rcnn import train_bus as T
from wssdl_bus_amd.fast_rcnn.config import cfg
from wssdl_bus_amd.mil import core as mil_core


def softmax_ce(logits, labels):
    z = logits - logits.max(1, keepdims=True)
    logp = z - np.log(np.exp(z).sum(1, keepdims=True))
    return -logp[np.arange(len(labels)), labels]


def test_rpn_losses_match_numpy():
    rs = np.random.RandomState(0)
    N, H, W, A = 2, 5, 6, 9
    score = rs.normal(size=(N, H, W, 2 * A)).astype(np.float32)
    # reshape_layer(2): out[n, a*H+h, w, c] = in[n,h,w,c*A+a]   (network.py:283-291)
    resh = score.reshape(N, H, W, 2, A).transpose(0, 4, 1, 2, 3).reshape(N, A * H, W, 2)
    labels = rs.randint(-1, 2, size=(N, 1, A * H, W)).astype(np.int32)
    got = T.rpn_cls_loss(torch.from_numpy(resh), torch.from_numpy(labels)).item()
    keep = labels.reshape(-1) != -1
    want = softmax_ce(resh.reshape(-1, 2).astype(np.float64)[keep], labels.reshape(-1)[keep]).mean()
    assert abs(got - want) < 1e-5
    pred = rs.normal(0, 1.2, size=(N, H, W, 4 * A)).astype(np.float32)
    tg = rs.normal(0, 1.0, size=(N, 4 * A, H, W)).astype(np.float32)
    inw = (rs.rand(N, 4 * A, H, W) > 0.7).astype(np.float32)
    outw = (rs.rand(N, 4 * A, H, W) > 0.5).astype(np.float32) / 256.0
    data = (torch.from_numpy(labels), torch.from_numpy(tg), torch.from_numpy(inw), torch.from_numpy(outw))
    for n_img in (None, 1):
        got = T.rpn_box_loss(torch.from_numpy(pred), data, n_img).item()
        p = pred[:n_img].astype(np.float64)
        t, i, o = (x.transpose(0, 2, 3, 1)[:n_img].astype(np.float64) for x in (tg, inw, outw))
        d = p - t
        s = (np.abs(d) < 1).astype(np.float64)            # threshold 1 with the sigma=3 pieces: the reference's quirk
        per = o * (0.5 * (i * d * 3) ** 2 * s + (np.abs(d) - 0.5 / 9.0) * np.abs(s - 1))
        want = per.sum(axis=(1, 2)).mean() * 10
        assert abs(got - want) < 1e-4 * max(1.0, abs(want))


def test_rcnn_and_mil_losses_match_numpy():
    rs = np.random.RandomState(1)
    R, K = 40, 3
    cls = rs.normal(size=(R + 25, K)).astype(np.float32)          # 40 supervised rows + 25 weak rows
    labels = rs.randint(0, K, size=(R, 1)).astype(np.int32)
    got = T.rcnn_cls_loss(torch.from_numpy(cls), torch.from_numpy(labels)).item()
    want = softmax_ce(cls[:R].astype(np.float64), labels.reshape(-1)).mean()
    assert abs(got - want) < 1e-5
    bp = rs.normal(size=(R + 25, 4 * K)).astype(np.float32)
    tg = rs.normal(size=(R, 4 * K)).astype(np.float32)
    inw = (rs.rand(R, 4 * K) > 0.6).astype(np.float32)
    outw = (inw > 0).astype(np.float32)
    roi_data = (None, torch.from_numpy(labels), torch.from_numpy(tg), torch.from_numpy(inw), torch.from_numpy(outw))
    got = T.rcnn_box_loss(torch.from_numpy(bp), roi_data).item()
    want = (outw * inw * np.abs(bp[:R] - tg)).astype(np.float64).sum(1).mean()
    assert abs(got - want) < 1e-5
    # MIL, combined mode: both bag labels use the max-malignant instance (train_bus.py:655)
    ws = cls[R:]
    batch_inds = np.array([0] * 10 + [1] * 15, np.float32)
    mil_label = np.array([1, 2], np.int32)
    for step in (0, 2500):
        got = T.mil_loss(torch.from_numpy(ws), torch.from_numpy(batch_inds), torch.from_numpy(mil_label), 2, step,
                         [mil_core.get_mal_max_logit, mil_core.get_mal_max_logit]).item()
        rows = np.stack([ws[:10][np.argmax(ws[:10, 2])], ws[10:][np.argmax(ws[10:, 2])]]).astype(np.float64)
        ce = softmax_ce(rows, mil_label)
        w = np.array([0.0, cfg.TRAIN.WS_MAL_PCT, 1 - cfg.TRAIN.WS_MAL_PCT])[mil_label]
        scale = 1.0 - 0.99 * 0.9 ** (step // 2000)
        want = (scale * w * ce).mean()
        assert abs(got - want) < 1e-6
    # alternating mode: label 1 -> instance with the smallest background logit (core.py:88-96)
    got = T.mil_loss(torch.from_numpy(ws), torch.from_numpy(batch_inds), torch.from_numpy(mil_label), 2, 0,
                     [mil_core.get_mass_max_logit, mil_core.get_mal_max_logit]).item()
    rows = np.stack([ws[:10][np.argmin(ws[:10, 0])], ws[10:][np.argmax(ws[10:, 2])]]).astype(np.float64)
    want = ((1.0 - 0.99) * np.array([cfg.TRAIN.WS_MAL_PCT, 1 - cfg.TRAIN.WS_MAL_PCT]) * softmax_ce(rows, mil_label)).mean()
    assert abs(got - want) < 1e-6


def test_weight_decay_and_bag_logit_shapes():
    p = [torch.nn.Parameter(torch.full((3, 2), 2.0)), torch.nn.Parameter(torch.ones(4))]
    wd = T.l2_weight_decay(p).item()
    assert abs(wd - (0.5 * (6 * 4.0 + 4 * 1.0) * cfg.TRAIN.WEIGHT_DECAY)) < 1e-9     # sum(l2_loss(w)) * WEIGHT_DECAY
    logits = torch.arange(18, dtype=torch.float32).reshape(6, 3)
    bag, scale = mil_core.get_bag_logit(logits, torch.tensor([0, 0, 0, 1, 1, 1]), 3, torch.tensor([2, 1]), 2,
                                        [mil_core.get_mass_max_logit, mil_core.get_mal_max_logit])
    assert bag.shape == (2, 3) and scale.shape == (2,)
    assert torch.equal(bag[0], logits[2]) and torch.equal(bag[1], logits[3])         # label 2 -> mal max; label 1 -> bg min


def test_reshape_layer_index_map():
    from wssdl_bus_amd.networks.network import Network
    net = Network()
    N, H, W, A = 1, 3, 4, 9
    x = torch.arange(N * H * W * 2 * A, dtype=torch.float32).reshape(N, H, W, 2 * A)
    net.layers['s'] = x
    y = net.feed('s').reshape_layer(2, name='rpn_cls_score_reshape').get_output('rpn_cls_score_reshape')
    assert y.shape == (N, A * H, W, 2)
    for (h, w, a, c) in ((0, 0, 0, 0), (2, 3, 8, 1), (1, 2, 4, 0)):
        assert y[0, a * H + h, w, c] == x[0, h, w, c * A + a]
    z = net.feed('rpn_cls_score_reshape').reshape_layer(2 * A, name='rpn_cls_prob_reshape').get_output('rpn_cls_prob_reshape')
    assert torch.equal(z, x)                                                          # inverse map


def test_gpu_nms_rule_as_a_threshold_of_the_cpu_rule():
    """fast_rcnn/nms_wrapper.gpu_rule_threshold: for every f32 iou, iou > (float)thresh (nms_kernel.cu:71) is
    (double)iou >= gpu_rule_threshold(thresh) -- the compare the HIP kernel makes."""
    import numpy as np
    from wssdl_bus_amd.fast_rcnn.nms_wrapper import gpu_rule_threshold
    rs = np.random.RandomState(0)
    for th in (0.7, 0.5, 0.3, 0.05, 0.95, float(np.float32(0.7)), 1.0 / 3):
        t32 = np.float32(th)
        near = np.array([t32, np.nextafter(t32, np.float32(1)), np.nextafter(t32, np.float32(0))], np.float32)
        iou = np.concatenate([near, rs.uniform(0, 1, 20000).astype(np.float32),
                              (t32 + rs.uniform(-1e-6, 1e-6, 20000)).astype(np.float32)])
        assert np.array_equal(iou > t32, iou.astype(np.float64) >= gpu_rule_threshold(th)), th
    assert gpu_rule_threshold(0.7) < 0.7 and gpu_rule_threshold(0.5) > 0.5
