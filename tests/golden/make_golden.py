#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own code (imported from
/root/reference through oracle/ref_python_stage.py) on seeded inputs.

Run in the build container only:  python tests/golden/make_golden.py
The .npz files hold inputs and the reference's outputs -- data only.  No
reference source is stored.  Re-running must reproduce the committed files
bit-for-bit (legacy numpy RandomState streams are frozen).

GT box layouts come from the five sample annotations the reference ships
(SNUBH_BUS/Annotations/*.xml: class, xmin, ymin, xmax, ymax, image w/h, BIRADS
diag), loaded the way datasets/bus.py:176-223 does (0-based, uint16) and scaled
as roi_data_layer/minibatch_bus.py does (boxes * im_scale, float32).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_python_stage as stage  # noqa: E402

# (width, height, diag, [(cls, xmin, ymin, xmax, ymax), ...]) -- positives first
SAMPLE_XML = {
    "FILE01182": (498, 291, 0, [(1, 147, 24, 319, 117), (0, 70, 148, 447, 281)]),
    "FILE01654": (777, 535, 0, [(1, 322, 110, 410, 197), (0, 22, 219, 747, 510),
                                (0, 21, 13, 296, 160)]),
    "FILE02539": (738, 578, 1, [(2, 89, 19, 642, 347), (0, 307, 368, 722, 556)]),
    "FILE04254": (738, 594, 1, [(2, 305, 107, 364, 173), (0, 74, 354, 708, 563),
                                (0, 13, 5, 720, 100), (0, 20, 117, 261, 325)]),
    "FILE04591": (738, 578, 1, [(2, 226, 206, 610, 410), (0, 32, 438, 363, 566),
                                (0, 17, 8, 215, 174)]),
}
MAX_GT = 20  # config.py:92


def sample_gt(name, target=600, max_size=1000):
    """gt_boxes [20,5] f32, num_gt, im_info [4] f32 for one sample annotation."""
    w, h, diag, objs = SAMPLE_XML[name]
    scale = float(target) / min(w, h)
    if np.round(scale * max(w, h)) > max_size:
        scale = float(max_size) / max(w, h)
    boxes = np.array([[o[1] - 1, o[2] - 1, o[3] - 1, o[4] - 1] for o in objs], dtype=np.uint16)
    gt = np.zeros((MAX_GT, 5), dtype=np.float32)
    gt[:len(objs), :4] = boxes.astype(np.float32) * np.float32(scale)
    gt[:len(objs), 4] = [o[0] for o in objs]
    im_info = np.array([np.round(h * scale), np.round(w * scale), scale, diag + 1], dtype=np.float32)
    return gt, len(objs), im_info


def pad_gt(rows):
    gt = np.zeros((MAX_GT, 5), dtype=np.float32)
    rows = np.asarray(rows, dtype=np.float32).reshape(-1, 5)
    gt[:rows.shape[0]] = rows
    return gt, rows.shape[0]


def synth_gt_sets(im_h, im_w):
    """Edge-case GT sets."""
    sets = {}
    # GT that overlaps no inside anchor -> column max 0 -> every zero-overlap
    # anchor becomes fg (anchor_target_layer_tf_bus.py:446-449)
    sets["outside_quirk"] = pad_gt([[im_w + 50, im_h + 50, im_w + 90, im_h + 95, 1],
                                    [100, 100, 260, 300, 2]])
    # GT exactly on an anchor: base anchor 4 (-56,-56,71,71) shifted by (16*10, 16*12)
    sets["on_anchor"] = pad_gt([[-56 + 160, -56 + 192, 71 + 160, 71 + 192, 1],
                                [300, 40, 700, 330, 0]])
    # 20 GT boxes: 8 positives then 12 background boxes
    rs = np.random.RandomState(11)
    rows = []
    for k in range(20):
        bw, bh = rs.randint(60, 400), rs.randint(60, 350)
        x1, y1 = rs.randint(0, im_w - bw), rs.randint(0, im_h - bh)
        rows.append([x1 + rs.rand(), y1 + rs.rand(), x1 + bw + rs.rand(), y1 + bh + rs.rand(),
                     (1 + k % 2) if k < 8 else 0])
    sets["twenty"] = pad_gt(rows)
    # many anchors above 0.7 with a large positive box -> fg sub-sampling kicks in
    sets["big_pos"] = pad_gt([[40, 30, im_w - 60, im_h - 40, 2], [5, 5, 120, 90, 0]])
    # positives only (exist_neg False)
    sets["pos_only"] = pad_gt([[200, 150, 420, 330, 1], [600, 100, 760, 420, 2]])
    return sets


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %-34s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024.0))


def labels_i8(x):
    assert np.all((x == -1) | (x == 0) | (x == 1))
    return x.astype(np.int8)


def main():
    R = stage.load()
    cfg = R.cfg
    scales = [8, 16, 32]
    stride = [16, ]

    # ---- a1/a2 anchors -------------------------------------------------------
    save("anchors",
         a_8_16_32=R.generate_anchors(scales=np.array([8, 16, 32])),
         a_4_8_16_32=R.generate_anchors(scales=np.array([4, 8, 16, 32])),
         a_default=R.generate_anchors())

    # ---- a3/a4 IoU -----------------------------------------------------------
    rs = np.random.RandomState(3)
    xy = rs.uniform(0, 900, size=(700, 2))
    wh = rs.uniform(1, 400, size=(700, 2))
    boxes = np.hstack((xy, xy + wh))
    boxes[:50] = np.round(boxes[:50])
    q = boxes[rs.choice(700, 17, replace=False)].copy()
    q[:5] += rs.uniform(-30, 30, size=(5, 4))
    q[5] = boxes[5]                     # identical box -> IoU 1
    q[6] = [2000, 2000, 2100, 2100]     # disjoint from everything
    q[7] = [boxes[7, 2], boxes[7, 3], boxes[7, 2] + 10, boxes[7, 3] + 10]  # touches in 1 px
    save("bbox_overlaps", boxes=boxes, query=q,
         iou=R.bbox_overlaps(np.ascontiguousarray(boxes), np.ascontiguousarray(q)),
         ui=R.bbox_overlaps_ui(np.ascontiguousarray(boxes), np.ascontiguousarray(q)))

    # ---- a5 anchor targets ---------------------------------------------------
    shapes = {"vgg_37x62": (37, 62, 600, 1000), "res_38x63": (38, 63, 600, 1000),
              "res_63x100": (63, 100, 1000, 1600)}
    for sname, (H, W, im_h, im_w) in shapes.items():
        cases = {}
        target, max_size = (600, 1000) if im_h == 600 else (1000, 1600)
        for xml in SAMPLE_XML:
            gt, n, info = sample_gt(xml, target, max_size)
            cases[xml] = (gt, n, info, "SNUBH")
        for k, (gt, n) in synth_gt_sets(im_h, im_w).items():
            cases[k] = (gt, n, np.array([im_h, im_w, 1.0, 1], np.float32), "SNUBH")
        gt, n = synth_gt_sets(im_h, im_w)["twenty"]
        cases["twenty_fg"] = (gt, n, np.array([im_h, im_w, 1.0, 1], np.float32), "SNUBH_FG")
        gt, n = synth_gt_sets(im_h, im_w)["pos_only"]
        cases["pos_only_udiat"] = (gt, n, np.array([im_h, im_w, 1.0, 1], np.float32), "UDIAT")
        if sname != "res_38x63":      # keep the big fixtures to one full set
            cases = {k: cases[k] for k in ("FILE04254", "outside_quirk", "big_pos", "twenty")}
        out = {}
        for cname, (gt, n, info, dataset) in cases.items():
            score = np.zeros((1, H, W, 18), np.float32)
            gtb = gt[None]
            ng = np.array([n], np.int32)
            ii = info[None]
            # pre-subsample labels: run with a batch size no image can exceed
            cfg.TRAIN.RPN_BATCHSIZE = 10 ** 9
            pre = R.anchor_target_layer(score, gtb, ng, ii, None, stride, scales, dataset)
            cfg.TRAIN.RPN_BATCHSIZE = 256
            seed = 3 + len(out)
            np.random.seed(seed)
            fin = R.anchor_target_layer(score, gtb, ng, ii, None, stride, scales, dataset)
            out[cname + "/gt_boxes"] = gt
            out[cname + "/num_gt"] = ng
            out[cname + "/im_info"] = info
            out[cname + "/dataset"] = np.array(dataset)
            out[cname + "/seed"] = np.array(seed)
            out[cname + "/labels_pre"] = labels_i8(pre[0])
            out[cname + "/targets_pre"] = pre[1]
            out[cname + "/labels"] = labels_i8(fin[0])
            out[cname + "/targets"] = fin[1]
            out[cname + "/inside_w"] = fin[2]
            out[cname + "/outside_w"] = fin[3]
        save("anchor_target_" + sname, H=np.array(H), W=np.array(W), **out)

    # joint (combined) and ws variants, batch of 1 supervised + 2 weak images
    H, W = 38, 63
    gt0, n0, info0 = sample_gt("FILE04591")
    gtb = np.stack([gt0, np.zeros_like(gt0), np.zeros_like(gt0)])
    ng = np.array([n0, 0, 0], np.int32)
    ii = np.stack([info0, np.array([600, 1000, 1, 2], np.float32), np.array([600, 1000, 1, 1], np.float32)])
    score = np.zeros((3, H, W, 18), np.float32)
    np.random.seed(7)
    jt = R.anchor_target_layer_joint(score, gtb, ng, ii, None, True, stride, scales, "SNUBH")
    np.random.seed(7)
    jf = R.anchor_target_layer_joint(score[:1], gtb[:1], ng[:1], ii[:1], None, False, stride, scales, "SNUBH")
    ws = R.anchor_target_layer_ws(score[1:], gtb[1:], ng[1:], ii[1:], None, stride, scales)
    save("anchor_target_joint", gt_boxes=gtb, num_gt=ng, im_info=ii, seed=np.array(7),
         train_labels=labels_i8(jt[0]), train_targets=jt[1], train_inside=jt[2], train_outside=jt[3],
         test_labels=labels_i8(jf[0]), test_targets=jf[1], test_inside=jf[2], test_outside=jf[3],
         ws_labels=labels_i8(ws[0]), ws_targets_sum=np.array(float(np.abs(ws[1]).sum())),
         ws_shape=np.array(ws[1].shape))

    # ---- a8 NMS --------------------------------------------------------------
    out = {}
    for n in (1, 2, 64, 65, 300, 6000, 12000):
        rs = np.random.RandomState(100 + n)
        # clustered boxes so that suppression chains are long
        nc = max(1, n // 12)
        centers = rs.uniform(0, 1000, size=(nc, 2)) * [1.0, 0.6]
        cid = rs.randint(0, nc, size=n)
        ctr = centers[cid] + rs.normal(0, 12, size=(n, 2))
        wh = np.exp(rs.normal(4.5, 0.6, size=(n, 2)))
        b = np.hstack((ctr - wh / 2, ctr + wh / 2))
        b[:, 0::2] = np.clip(b[:, 0::2], 0, 999)
        b[:, 1::2] = np.clip(b[:, 1::2], 0, 599)
        sc = rs.permutation(n).astype(np.float64) / n + rs.uniform(0, 0.1 / n, size=n)
        dets = np.hstack((b, sc[:, None])).astype(np.float32)
        if n >= 64:
            dets[5, :4] = dets[3, :4]          # exact duplicate boxes
            dets[9, :4] = dets[3, :4]
            # threshold-boundary pair: IoU exactly 0.7 in f32 arithmetic
            # (areas 100 and 70 nested -> 70/100)
            dets[11, :4] = [10, 10, 19, 19]
            dets[12, :4] = [10, 10, 19, 16]
        assert len(np.unique(dets[:, 4])) == n
        order = dets[:, 4].argsort()[::-1]
        dets = dets[order]                      # proposal layer hands NMS sorted dets
        for th in (0.7, 0.3):
            keep = np.asarray(R.cpu_nms(dets, th), dtype=np.int32)
            out["n%d/keep_%02d" % (n, int(th * 10))] = keep
        out["n%d/dets" % n] = dets
    # unsorted input: cpu_nms sorts internally (cpu_nms.pyx:25)
    rs = np.random.RandomState(5)
    dets = out["n300/dets"][rs.permutation(300)]
    out["unsorted/dets"] = dets
    out["unsorted/keep_07"] = np.asarray(R.cpu_nms(dets, 0.7), dtype=np.int32)
    save("nms", **out)

    # ---- f3: the test path's NMS, utils/nms.pyx (`nms` :17-68, `nms_new` :70-123) --------------
    # the same det sets as above at the test threshold 0.3 (config.py:238) and at 0.7, plus a set built
    # for nms_new's containment terms: small boxes inside large ones (low IoU, inter / area_small = 1),
    # pairs whose inter / area sits on either side of 0.95, and the f32 neighbours of 0.95 itself
    # (100 x 100 against 95 x 100: 0.95 in exact arithmetic; 19 of 20 columns)
    out2 = {}
    for n in (1, 2, 64, 65, 300, 6000, 12000):
        dets = out["n%d/dets" % n]
        for th in (0.3, 0.7):
            out2["n%d/nms_%02d" % (n, int(th * 10))] = np.asarray(R.utils_nms(dets, th), dtype=np.int32)
            out2["n%d/nms_new_%02d" % (n, int(th * 10))] = np.asarray(R.utils_nms_new(dets, th), dtype=np.int32)
    rs = np.random.RandomState(77)
    big = np.hstack((rs.uniform(0, 600, size=(40, 2)), np.zeros((40, 2))))
    big[:, 2:] = big[:, :2] + rs.uniform(120, 380, size=(40, 2))
    rows = [big]
    for k in range(40):
        x1, y1, x2, y2 = big[k]
        w, h = x2 - x1 + 1, y2 - y1 + 1
        inner = []
        for f in (0.2, 0.5, 0.9, 0.94, 0.96):                 # nested at the corner: inter / area_big = f * 1
            inner.append([x1, y1, x1 + f * w - 1, y2])
        for f in (0.02, 0.05, 0.5):                           # small box sticking out by a fraction f of its width
            sw = 0.2 * w
            inner.append([x2 - (1 - f) * sw, y1 + 5, x2 + f * sw, y1 + 5 + 0.2 * h])
        rows.append(np.array(inner))
    exact = np.array([[0, 700, 99, 799], [0, 700, 94, 799], [200, 700, 299, 799], [200, 700, 294, 798],
                      [400, 700, 419, 719], [400, 700, 418, 719], [400, 700, 419, 718]], dtype=np.float64)
    b = np.vstack(rows + [exact])
    n = b.shape[0]
    sc = rs.permutation(n).astype(np.float64) / n + rs.uniform(0, 0.1 / n, size=n)
    dets = np.hstack((b, sc[:, None])).astype(np.float32)
    assert len(np.unique(dets[:, 4])) == n
    out2["contain/dets"] = dets
    for th in (0.3, 0.5, 0.7):
        out2["contain/nms_%02d" % int(th * 10)] = np.asarray(R.utils_nms(dets, th), dtype=np.int32)
        out2["contain/nms_new_%02d" % int(th * 10)] = np.asarray(R.utils_nms_new(dets, th), dtype=np.int32)
    assert out2["contain/nms_03"].tolist() != out2["contain/nms_new_03"].tolist()
    for n in (1, 2, 64, 65, 300, 6000, 12000):                # `nms` is cpu_nms's rule in a second file
        for th in ("03", "07"):
            assert out2["n%d/nms_%s" % (n, th)].tolist() == out["n%d/keep_%s" % (n, th)].tolist()
    save("nms_utils", **out2)

    # ---- a6/a7/a9 proposal layer ---------------------------------------------
    def rpn_inputs(H, W, N, seed):
        rs = np.random.RandomState(seed)
        logits = rs.normal(0, 1, size=(N, H, W, 9, 2)).astype(np.float32)
        e = np.exp(logits - logits.max(-1, keepdims=True))
        p = (e / e.sum(-1, keepdims=True)).astype(np.float32)
        prob = np.concatenate((p[..., 0], p[..., 1]), axis=-1)      # [N,H,W,18]: bg then fg
        fg = prob[..., 9:]
        # make fg scores pairwise distinct per image (sort ties are unspecified)
        for i in range(N):
            flat = fg[i].reshape(-1)
            u, idx, cnt = np.unique(flat, return_inverse=True, return_counts=True)
            while cnt.max() > 1:
                dup = np.where(cnt[idx] > 1)[0]
                flat[dup] = np.nextafter(flat[dup], np.float32(1), dtype=np.float32) \
                    + (rs.rand(len(dup)) * 1e-6).astype(np.float32)
                u, idx, cnt = np.unique(flat, return_inverse=True, return_counts=True)
            fg[i] = flat.reshape(H, W, 9)
        prob[..., 9:] = fg
        pred = rs.normal(0, 0.2, size=(N, H, W, 36)).astype(np.float32)
        return prob, pred

    pl = {}
    for name, (H, W, N, info, train) in {
        "res_38x63_train": (38, 63, 2, [[600, 1000, 1.0, 1], [584, 1000, 2.008, 2]], True),
        "res_38x63_test": (38, 63, 1, [[600, 1000, 1.0, 1]], False),
        "vgg_37x62_train": (37, 62, 1, [[600, 1000, 1.0, 2]], True),
        "res_63x100_test": (63, 100, 1, [[1000, 1600, 1.0, 1]], False),
    }.items():
        prob, pred = rpn_inputs(H, W, N, seed=3 + H + W + int(train))
        ii = np.array(info, np.float32)
        rois = R.proposal_layer(prob, pred, ii, train, False, stride, scales)
        pl[name + "/prob"] = prob
        pl[name + "/pred"] = pred
        pl[name + "/im_info"] = ii
        pl[name + "/is_training"] = np.array(train)
        pl[name + "/rois"] = rois
    save("proposal_layer", **pl)

    # ---- a10 proposal target --------------------------------------------------
    rois = pl["res_38x63_train/rois"]
    gt0, n0, info0 = sample_gt("FILE04254")
    gt1, n1, info1 = sample_gt("FILE02539")
    gtb = np.stack([gt0, gt1])
    ng = np.array([n0, n1], np.int32)
    pt = dict(rois_in=rois, gt_boxes=gtb, num_gt=ng)
    for tag, fn in (
        ("alt_train", lambda: R.proposal_target_layer(rois, gtb, ng, 3, True, False)),
        ("alt_ws", lambda: R.proposal_target_layer(rois, gtb, ng, 3, True, True)),
        ("alt_test", lambda: R.proposal_target_layer(rois, gtb, ng, 3, False, False)),
    ):
        np.random.seed(13)
        o = fn()
        for k, nm in enumerate(("rois", "labels", "targets", "inside", "outside")):
            pt["%s/%s" % (tag, nm)] = o[k]
    # joint: IMS_PER_BATCH=1 supervised image + the second image's rois as weak
    cfg.TRAIN.IMS_PER_BATCH = 1
    cfg.TRAIN.WS_IMS_PER_BATCH = 1
    for tag, tr in (("joint_train", True), ("joint_test", False)):
        np.random.seed(17)
        o = R.proposal_target_layer_joint(rois, gtb, ng, 3, tr)
        for k, nm in enumerate(("rois", "labels", "targets", "inside", "outside")):
            pt["%s/%s" % (tag, nm)] = o[k]
    cfg.TRAIN.WS_IMS_PER_BATCH = 2
    pt["seed_alt"] = np.array(13)
    pt["seed_joint"] = np.array(17)
    save("proposal_target", **pt)

    # ---- a10 with cfg.TRAIN.BBOX_NORMALIZE_TARGETS_PRECOMPUTED (proposal_target_layer_tf_bus.py:221-224) -------
    # the reference's own means / stds (config.py:182-183: (0,0,0,0) / (0.1,0.1,0.2,0.2)), same inputs and seeds as above
    ptn = dict(rois_in=rois, gt_boxes=gtb, num_gt=ng,
               means=np.asarray(cfg.TRAIN.BBOX_NORMALIZE_MEANS, np.float64), stds=np.asarray(cfg.TRAIN.BBOX_NORMALIZE_STDS, np.float64))
    assert not cfg.TRAIN.BBOX_NORMALIZE_TARGETS_PRECOMPUTED
    cfg.TRAIN.BBOX_NORMALIZE_TARGETS_PRECOMPUTED = True
    np.random.seed(13)
    o = R.proposal_target_layer(rois, gtb, ng, 3, True, False)
    for k, nm in enumerate(("rois", "labels", "targets", "inside", "outside")):
        ptn["alt_train/%s" % nm] = o[k]
    cfg.TRAIN.IMS_PER_BATCH = 1
    cfg.TRAIN.WS_IMS_PER_BATCH = 1
    np.random.seed(17)
    o = R.proposal_target_layer_joint(rois, gtb, ng, 3, True)
    for k, nm in enumerate(("rois", "labels", "targets", "inside", "outside")):
        ptn["joint_train/%s" % nm] = o[k]
    cfg.TRAIN.WS_IMS_PER_BATCH = 2
    cfg.TRAIN.BBOX_NORMALIZE_TARGETS_PRECOMPUTED = False
    ptn["seed_alt"] = np.array(13)
    ptn["seed_joint"] = np.array(17)
    assert np.array_equal(ptn["alt_train/rois"], pt["alt_train/rois"]) and not np.array_equal(ptn["alt_train/targets"], pt["alt_train/targets"])
    save("proposal_target_norm", **ptn)

    # ---- box transforms -------------------------------------------------------
    rs = np.random.RandomState(21)
    ex = rs.uniform(0, 500, size=(64, 2))
    ex = np.hstack((ex, ex + rs.uniform(5, 300, size=(64, 2))))
    g = rs.uniform(0, 500, size=(64, 2))
    g = np.hstack((g, g + rs.uniform(5, 300, size=(64, 2)))).astype(np.float32)
    d = rs.normal(0, 0.3, size=(64, 4)).astype(np.float32)
    save("bbox_transform", ex=ex, gt=g, deltas=d,
         t_f64_f32=R.bbox_transform(ex, g), t_f32_f32=R.bbox_transform(ex.astype(np.float32), g),
         inv=R.bbox_transform_inv(ex, d),
         clipped=R.clip_boxes(R.bbox_transform_inv(ex, d), np.array([400, 450], np.float32)))


if __name__ == "__main__":
    if not stage.reference_present():
        sys.exit("reference tree not present: golden vectors can only be generated in the build container")
    main()
