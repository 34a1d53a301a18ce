"""Pin the CPU oracle (oracle/) against golden vectors produced by the reference's
own code (tests/golden/make_golden.py) and, where oracle/_ref is present, against
the reference's own Cython kernels directly.  CPU only."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import c_oracle, np_oracle as O, ref_kernels

STRIDE = [16, ]
SCALES = [8, 16, 32]


def groups(npz):
    names = sorted({k.split("/")[0] for k in npz.files if "/" in k})
    return names


# ---------------------------------------------------------------- anchors ---

def test_generate_anchors_golden():
    g = load_golden("anchors")
    assert np.array_equal(O.generate_anchors(scales=[8, 16, 32]), g["a_8_16_32"])
    assert np.array_equal(O.generate_anchors(scales=[4, 8, 16, 32]), g["a_4_8_16_32"])
    assert np.array_equal(O.generate_anchors(scales=2 ** np.arange(3, 6)), g["a_default"])
    # live-code values, not the Matlab comment block (generate_anchors.py:10-35)
    assert O.generate_anchors()[0].tolist() == [-84.0, -40.0, 99.0, 55.0]


def test_shifted_anchor_counts():
    base = O.generate_anchors()
    for (H, W, im_h, im_w, total, inside) in ((37, 62, 600, 1000, 20646, 8151),
                                              (38, 63, 600, 1000, 21546, 8151),
                                              (63, 100, 1000, 1600, 56700, 32276)):
        a = O.shifted_anchors(H, W, 16, base)
        assert a.shape == (total, 4)
        ins = (a[:, 0] >= 0) & (a[:, 1] >= 0) & (a[:, 2] < im_w) & (a[:, 3] < im_h)
        assert ins.sum() == inside


# --------------------------------------------------------------------- IoU ---

def test_bbox_overlaps_golden():
    g = load_golden("bbox_overlaps")
    assert np.array_equal(O.bbox_overlaps(g["boxes"], g["query"]), g["iou"])
    assert np.array_equal(O.bbox_overlaps_ui(g["boxes"], g["query"]), g["ui"])
    assert g["iou"][5, 5] == 1.0 and np.all(g["iou"][:, 6] == 0)


@pytest.mark.skipif(not ref_kernels.available(), reason="oracle/_ref not built")
def test_bbox_overlaps_vs_reference_cython():
    rs = np.random.RandomState(0)
    for _ in range(5):
        xy = rs.uniform(-50, 900, size=(3000, 2))
        b = np.hstack((xy, xy + rs.uniform(0, 500, size=(3000, 2))))
        q = b[rs.choice(3000, 20, replace=False)] + rs.uniform(-5, 5, size=(20, 4))
        assert np.array_equal(O.bbox_overlaps(b, q), ref_kernels.bbox_overlaps(b, q))
        assert np.array_equal(O.bbox_overlaps_ui(b, q), ref_kernels.bbox_overlaps_ui(b, q))
    assert O.bbox_overlaps(np.zeros((0, 4)), q).shape == (0, 20)
    assert O.bbox_overlaps(b, np.zeros((0, 4))).shape == (3000, 0)


# --------------------------------------------------------------------- NMS ---

def test_nms_golden():
    g = load_golden("nms")
    for name in groups(g):
        dets = g[name + "/dets"]
        for key in ("keep_07", "keep_03"):
            if name + "/" + key in g.files:
                th = 0.7 if key.endswith("07") else 0.3
                assert O.nms(dets, th) == g[name + "/" + key].tolist(), (name, key)
    assert O.nms(np.zeros((0, 5), np.float32), 0.7) == []     # nms_wrapper.py:16-17


def test_nms_threshold_is_double_compare():
    # IoU of these two boxes is 70/100 -> 0.7f in f32; (double)0.7f < 0.7, so the
    # reference (PyFloat compare, cpu_nms.c:2495) keeps both at thresh 0.7.
    d = np.array([[10, 10, 19, 19, 0.9], [10, 10, 19, 16, 0.8]], dtype=np.float32)
    assert O.nms(d, 0.7) == [0, 1]
    assert O.nms(d, float(np.float32(0.7))) == [0]
    if ref_kernels.available():
        assert ref_kernels.cpu_nms(d, 0.7) == [0, 1]
        assert ref_kernels.cpu_nms(d, float(np.float32(0.7))) == [0]


@pytest.mark.skipif(not ref_kernels.available(), reason="oracle/_ref not built")
def test_nms_vs_reference_cython_random():
    rs = np.random.RandomState(1)
    for n in (1, 7, 200, 1500):
        c = rs.uniform(0, 300, size=(n, 2))
        wh = rs.uniform(5, 120, size=(n, 2))
        d = np.hstack((c, c + wh, rs.permutation(n)[:, None] / float(n))).astype(np.float32)
        for th in (0.3, 0.5, 0.7):
            assert O.nms(d, th) == [int(k) for k in ref_kernels.cpu_nms(d, th)]


def test_utils_nms_and_nms_new_golden():
    """f3: the test path calls utils/nms.pyx (`nms` :17-68, fast_rcnn/test_bus.py:366), not cpu_nms.pyx; the
    fixtures hold that file's own outputs (`nms` and the containment variant `nms_new` :70-123)."""
    g = load_golden("nms_utils")
    for name in groups(g):
        dets = g[name + "/dets"] if name + "/dets" in g.files else load_golden("nms")[name + "/dets"]
        for key in [k.split("/")[1] for k in g.files if k.startswith(name + "/") and "dets" not in k]:
            th = int(key[-2:]) / 10.0
            got = O.nms_new(dets, th) if key.startswith("nms_new") else O.nms(dets, th)
            assert got == g[name + "/" + key].tolist(), (name, key)
    assert O.nms_new(np.zeros((0, 5), np.float32), 0.3) == []
    # the containment terms bite: a small box inside a large one has a low IoU and inter / area_small = 1
    d = np.array([[0, 0, 199, 199, 0.9], [50, 50, 69, 69, 0.8], [0, 12, 199, 211, 0.7]], dtype=np.float32)
    assert O.nms(d, 0.99) == [0, 1, 2] and O.nms_new(d, 0.99) == [0, 2]      # box 2: 188/200 = 0.94 of either area
    d[2, 1::2] = [9, 208]                                                      # 191/200 = 0.955 > 0.95
    assert O.nms_new(d, 0.99) == [0]


@pytest.mark.skipif(getattr(ref_kernels, "nms_new", None) is None, reason="oracle/_ref/cython_nms.so not built")
def test_utils_nms_vs_reference_cython_random():
    rs = np.random.RandomState(2)
    for n in (1, 7, 200, 1500):
        c = rs.uniform(0, 300, size=(n, 2))
        wh = np.exp(rs.uniform(np.log(4), np.log(200), size=(n, 2)))          # nested boxes are common
        d = np.hstack((c, c + wh, rs.permutation(n)[:, None] / float(n))).astype(np.float32)
        for th in (0.3, 0.5, 0.7):
            assert O.nms(d, th) == [int(k) for k in ref_kernels.nms(d, th)]
            assert O.nms_new(d, th) == [int(k) for k in ref_kernels.nms_new(d, th)]


# ------------------------------------------------------------ anchor target ---

@pytest.mark.parametrize("shape", ["vgg_37x62", "res_38x63", "res_63x100"])
def test_anchor_target_golden(shape):
    g = load_golden("anchor_target_" + shape)
    H, W = int(g["H"]), int(g["W"])
    score = np.zeros((1, H, W, 18), np.float32)
    for name in groups(g):
        gt = g[name + "/gt_boxes"][None]
        ng = g[name + "/num_gt"]
        ii = g[name + "/im_info"][None]
        ds = str(g[name + "/dataset"])
        pre = O.anchor_target_layer(score, gt, ng, ii, None, STRIDE, SCALES, ds,
                                    cfg=dict(RPN_BATCHSIZE=10 ** 9))
        assert np.array_equal(pre[0].astype(np.int8), g[name + "/labels_pre"]), name
        assert np.array_equal(pre[1], g[name + "/targets_pre"]), name
        rng = np.random.RandomState(int(g[name + "/seed"]))
        fin = O.anchor_target_layer(score, gt, ng, ii, None, STRIDE, SCALES, ds, rng=rng)
        assert np.array_equal(fin[0].astype(np.int8), g[name + "/labels"]), name
        assert np.array_equal(fin[1], g[name + "/targets"]), name
        assert np.array_equal(fin[2], g[name + "/inside_w"]), name
        assert np.array_equal(fin[3], g[name + "/outside_w"]), name
        lab = g[name + "/labels"]
        assert (lab == 1).sum() <= 128 and (lab >= 0).sum() <= 256


def test_anchor_target_zero_overlap_quirk():
    g = load_golden("anchor_target_res_38x63")
    pre = g["outside_quirk/labels_pre"]
    # a GT box that overlaps no inside anchor turns every zero-overlap inside
    # anchor into fg (anchor_target_layer_tf_bus.py:446-449): thousands of 1s
    assert (pre == 1).sum() > 5000


def test_anchor_target_joint_and_ws_golden():
    g = load_golden("anchor_target_joint")
    gt, ng, ii = g["gt_boxes"], g["num_gt"], g["im_info"]
    score = np.zeros((3, 38, 63, 18), np.float32)
    rng = np.random.RandomState(int(g["seed"]))
    jt = O.anchor_target_layer_joint(score, gt, ng, ii, None, True, STRIDE, SCALES, "SNUBH", rng=rng)
    assert jt[0].shape == (3, 1, 342, 63)
    assert np.array_equal(jt[0].astype(np.int8), g["train_labels"])
    for k, nm in ((1, "train_targets"), (2, "train_inside"), (3, "train_outside")):
        assert np.array_equal(jt[k], g[nm])
    rng = np.random.RandomState(int(g["seed"]))
    jf = O.anchor_target_layer_joint(score[:1], gt[:1], ng[:1], ii[:1], None, False, STRIDE,
                                     SCALES, "SNUBH", rng=rng)
    assert np.array_equal(jf[0].astype(np.int8), g["test_labels"])
    assert np.array_equal(jf[1], g["test_targets"])
    ws = O.anchor_target_layer_ws(score[1:], None, None, None, None, STRIDE, SCALES)
    assert np.array_equal(ws[0].astype(np.int8), g["ws_labels"])
    assert tuple(g["ws_shape"]) == ws[1].shape and not ws[1].any() and not ws[3].any()


# ----------------------------------------------------------- proposal layer ---

@pytest.mark.parametrize("case", ["res_38x63_train", "res_38x63_test", "vgg_37x62_train",
                                  "res_63x100_test"])
def test_proposal_layer_golden(case):
    g = load_golden("proposal_layer")
    rois = O.proposal_layer(g[case + "/prob"], g[case + "/pred"], g[case + "/im_info"],
                            bool(g[case + "/is_training"]), False, STRIDE, SCALES)
    assert rois.dtype == np.float32
    assert np.array_equal(rois, g[case + "/rois"])


def test_box_transforms_golden():
    g = load_golden("bbox_transform")
    assert np.array_equal(O.bbox_transform(g["ex"], g["gt"]), g["t_f64_f32"])
    assert np.array_equal(O.bbox_transform(g["ex"].astype(np.float32), g["gt"]), g["t_f32_f32"])
    inv = O.bbox_transform_inv(g["ex"], g["deltas"])
    assert np.array_equal(inv, g["inv"])
    assert np.array_equal(O.clip_boxes(inv, np.array([400, 450], np.float32)), g["clipped"])
    assert O.bbox_transform_inv(np.zeros((0, 4)), np.zeros((0, 4), np.float32)).shape == (0, 4)


# ---------------------------------------------------------- proposal target ---

def test_proposal_target_golden():
    g = load_golden("proposal_target")
    rois, gt, ng = g["rois_in"], g["gt_boxes"], g["num_gt"]
    names = ("rois", "labels", "targets", "inside", "outside")
    for tag, args in (("alt_train", (True, False)), ("alt_ws", (True, True)),
                      ("alt_test", (False, False))):
        rng = np.random.RandomState(int(g["seed_alt"]))
        o = O.proposal_target_layer(rois, gt, ng, 3, args[0], args[1], rng=rng)
        for k, nm in enumerate(names):
            assert np.array_equal(o[k], g["%s/%s" % (tag, nm)]), (tag, nm)
    for tag, tr in (("joint_train", True), ("joint_test", False)):
        rng = np.random.RandomState(int(g["seed_joint"]))
        o = O.proposal_target_layer_joint(rois, gt, ng, 3, tr, rng=rng,
                                          cfg=dict(IMS_PER_BATCH=1, WS_IMS_PER_BATCH=1))
        for k, nm in enumerate(names):
            assert np.array_equal(o[k], g["%s/%s" % (tag, nm)]), (tag, nm)
    # joint/train: 128 sampled rows for the supervised image + every roi of the weak one
    assert g["joint_train/labels"].shape[0] == 128
    assert g["joint_train/rois"].shape[0] == 128 + int((rois[:, 0] == 1).sum())


def test_proposal_target_golden_with_precomputed_normalisation():
    """cfg.TRAIN.BBOX_NORMALIZE_TARGETS_PRECOMPUTED = True (proposal_target_layer_tf_bus.py:221-224) with the
    reference's own means / stds (config.py:182-183): the reference's outputs, bit for bit."""
    g = load_golden("proposal_target_norm")
    rois, gt, ng = g["rois_in"], g["gt_boxes"], g["num_gt"]
    names = ("rois", "labels", "targets", "inside", "outside")
    over = dict(BBOX_NORMALIZE_TARGETS_PRECOMPUTED=True, BBOX_NORMALIZE_MEANS=tuple(g["means"]),
                BBOX_NORMALIZE_STDS=tuple(g["stds"]))
    o = O.proposal_target_layer(rois, gt, ng, 3, True, False, rng=np.random.RandomState(int(g["seed_alt"])), cfg=over)
    for k, nm in enumerate(names):
        assert np.array_equal(o[k], g["alt_train/%s" % nm]), nm
    o = O.proposal_target_layer_joint(rois, gt, ng, 3, True, rng=np.random.RandomState(int(g["seed_joint"])),
                                      cfg=dict(over, IMS_PER_BATCH=1, WS_IMS_PER_BATCH=1))
    for k, nm in enumerate(names):
        assert np.array_equal(o[k], g["joint_train/%s" % nm]), nm
    # the switch matters: the plain fixtures' targets differ, and by exactly the scaling (means are zero)
    p = load_golden("proposal_target")
    assert not np.array_equal(p["alt_train/targets"], g["alt_train/targets"])
    fg = p["alt_train/targets"] != 0
    scaled = (p["alt_train/targets"].astype(np.float64).reshape(-1, 3, 4) / g["stds"]).reshape(p["alt_train/targets"].shape)
    assert np.array_equal(scaled.astype(np.float32)[fg], g["alt_train/targets"][fg])


def test_oracle_roundtrip_c_binding_shapes():
    top, arg = c_oracle.roi_pool_forward(np.zeros((1, 4, 4, 2), np.float32),
                                         np.zeros((0, 5), np.float32), 7, 7, 1.0 / 16)
    assert top.shape == (0, 7, 7, 2) and arg.dtype == np.int32
