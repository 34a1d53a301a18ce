"""Known-answer pins for the RoI-pool oracle.

The reference's RoiPool op cannot be built here (TensorFlow headers absent) and
its only test asserts nothing, so the oracle for this op is "parity unpinned"
by the reference.  These hand-computed cases pin it instead; expected values
are derived in the comments from roi_pooling_op_gpu.cu.cc:36-84 (mode 'cuda')
and roi_pooling_op.cc:152-194 (mode 'cpu').  CPU only."""
import numpy as np
import pytest

from oracle import c_oracle


def ramp(N, H, W, C=1):
    """feature[n,h,w,c] = 10000*n + 100*h + w + 0.25*c: strictly increasing in (h,w)."""
    n, h, w, c = np.meshgrid(np.arange(N), np.arange(H), np.arange(W), np.arange(C), indexing="ij")
    return (10000.0 * n + 100.0 * h + w + 0.25 * c).astype(np.float32)


def fwd(bottom, rois, ph, pw, scale, mode):
    return c_oracle.roi_pool_forward(bottom, np.asarray(rois, np.float32), ph, pw, scale, mode)


def test_one_by_one_roi():
    f = ramp(1, 4, 4)
    # x1=y1=x2=y2=16, scale 1/16 -> start=end=1: a 1x1 RoI on cell (1,1)=101, flat idx 5
    top, arg = fwd(f, [[0, 16, 16, 16, 16]], 7, 7, 1.0 / 16, "cuda")
    # cuda rounding: floor(ph/7)=0, ceil((ph+1)/7)=1 for every bin -> all 49 see the cell
    assert np.all(top == 101.0) and np.all(arg == 5)
    top, arg = fwd(f, [[0, 16, 16, 16, 16]], 7, 7, 1.0 / 16, "cpu")
    # cpu rounding: hend=(int)((ph+1)*f32(1/7)) is 0 for ph<6 and 1 for ph=6
    exp = np.zeros((7, 7), np.float32)
    exp[6, 6] = 101.0
    assert np.array_equal(top[0, :, :, 0], exp)
    assert arg[0, 6, 6, 0] == 5 and (arg[0, :, :, 0] == -1).sum() == 48


def test_full_map_two_by_two():
    f = ramp(1, 4, 4)
    for mode in ("cuda", "cpu"):
        top, arg = fwd(f, [[0, 0, 0, 48, 48]], 2, 2, 1.0 / 16, mode)
        assert top[0, :, :, 0].tolist() == [[101.0, 103.0], [301.0, 303.0]]
        assert arg[0, :, :, 0].tolist() == [[5, 7], [13, 15]]


def test_round_half_away_from_zero():
    f = ramp(1, 4, 4)
    # 8/16 = 0.5 -> 1 and 24/16 = 1.5 -> 2 under C round(); half-to-even would give 0 and 2
    top, arg = fwd(f, [[0, 8, 8, 24, 24]], 1, 1, 1.0 / 16, "cuda")
    assert top.ravel().tolist() == [202.0] and arg.ravel().tolist() == [10]
    # window is rows/cols 1..2 only: the (1,1) corner is where a constant map's argmax lands
    top, arg = fwd(np.ones((1, 4, 4, 1), np.float32), [[0, 8, 8, 24, 24]], 1, 1, 1.0 / 16, "cuda")
    assert arg.ravel().tolist() == [5]      # first maximum in (h,w) scan order, strict '>'


def test_border_clip_and_empty_bins():
    f = ramp(1, 4, 4)
    # end = round(160/16) = 10 -> 11x11 RoI on a 4x4 map, 2x2 bins of size 5.5:
    # bin 0 = [0,6) clipped to [0,4); bin 1 = [5,11) clipped to [4,4) = empty -> 0 / -1
    top, arg = fwd(f, [[0, 0, 0, 160, 160]], 2, 2, 1.0 / 16, "cuda")
    assert top[0, :, :, 0].tolist() == [[303.0, 0.0], [0.0, 0.0]]
    assert arg[0, :, :, 0].tolist() == [[15, -1], [-1, -1]]


def test_reference_test_script_case():
    # roi_pooling_op_test.py:18,23: rois [[0,10,10,20,20],[31,30,30,40,40]], 6x6, scale 1/3
    f = ramp(32, 100, 100)
    rois = [[0, 10, 10, 20, 20], [31, 30, 30, 40, 40]]
    top, arg = fwd(f, rois, 6, 6, 1.0 / 3, "cuda")
    # roi 0: start=round(3.33)=3, end=round(6.67)=7 -> 5x5, bin=5/6.
    # hend=ceil((ph+1)*5/6) = 1,2,3,4,5,5 -> last row of each bin is 3+hend-1
    last = np.array([3, 4, 5, 6, 7, 7])
    exp0 = 100.0 * last[:, None] + last[None, :]
    assert np.array_equal(top[0, :, :, 0], exp0.astype(np.float32))
    assert np.array_equal(arg[0, :, :, 0], (last[:, None] * 100 + last[None, :]))
    # roi 1: start=10, end=round(13.33)=13 -> 4x4, bin=2/3; hend=ceil((ph+1)*2/3)=1,2,2,3,4,4
    last = 10 + np.array([1, 2, 2, 3, 4, 4]) - 1
    exp1 = 310000.0 + 100.0 * last[:, None] + last[None, :]
    assert np.array_equal(top[1, :, :, 0], exp1.astype(np.float32))
    # cpu rounding on roi 0: hend=(int)((ph+1)*f32(5/6)) = 0,1,2,3,4,5 -> bin 0 empty
    top, arg = fwd(f, rois, 6, 6, 1.0 / 3, "cpu")
    last = np.array([3, 4, 5, 6, 7])
    assert np.all(top[0, 0, :, 0] == 0) and np.all(top[0, :, 0, 0] == 0)
    assert np.all(arg[0, 0, :, 0] == -1)
    assert np.array_equal(top[0, 1:, 1:, 0], (100.0 * last[:, None] + last[None, :]).astype(np.float32))


def test_channels_and_batch_index():
    f = ramp(3, 6, 5, C=4)
    top, arg = fwd(f, [[2, 0, 0, 64, 80]], 1, 1, 1.0 / 16, "cuda")
    # whole 6x5 map of image 2: max at (5,4); flat index (5*5+4)*4+c, value 20504+0.25c
    assert top.ravel().tolist() == [20504.0, 20504.25, 20504.5, 20504.75]
    assert arg.ravel().tolist() == [116, 117, 118, 119]


def test_backward_two_overlapping_rois_known_answer():
    f = ramp(1, 4, 4)
    rois = np.array([[0, 0, 0, 48, 48], [0, 0, 0, 48, 48], [0, 16, 16, 48, 48]], np.float32)
    top, arg = fwd(f, rois, 2, 2, 1.0 / 16, "cuda")
    diff = np.arange(1, 13, dtype=np.float32).reshape(3, 2, 2, 1)
    for literal in (True, False):
        g = c_oracle.roi_pool_backward(diff, arg, rois, f.shape, 2, 2, 1.0 / 16, literal=literal)
        exp = np.zeros((4, 4), np.float32)
        # rois 0,1: bins -> cells (1,1),(1,3),(3,1),(3,3) with diffs 1..4 and 5..8
        exp[1, 1] = 1 + 5
        exp[1, 3] = 2 + 6
        exp[3, 1] = 3 + 7
        exp[3, 3] = 4 + 8
        # roi 2 = rows/cols 1..3 (3x3), bin 1.5: bins [0,2),[1,3) -> cells (2,2),(2,3),(3,2),(3,3)
        exp[2, 2] += 9
        exp[2, 3] += 10
        exp[3, 2] += 11
        exp[3, 3] += 12
        assert np.array_equal(g[0, :, :, 0], exp)


@pytest.mark.parametrize("mode", ["cuda", "cpu"])
def test_backward_scatter_equals_literal_gather(mode):
    rs = np.random.RandomState(4)
    N, H, W, C = 2, 9, 13, 5
    f = rs.normal(size=(N, H, W, C)).astype(np.float32)
    f[f < 0] = 0                                    # post-ReLU plateau -> ties
    R = 60
    x1 = rs.uniform(0, 180, R)
    y1 = rs.uniform(0, 120, R)
    rois = np.stack([rs.randint(0, N, R), x1, y1, x1 + rs.uniform(0, 120, R) ** 1.2,
                     y1 + rs.uniform(0, 90, R) ** 1.2], axis=1).astype(np.float32)
    rois[:6, 3:] = rois[:6, 1:3] + rs.uniform(0, 40, (6, 2))      # RoIs smaller than 7x7 cells
    top, arg = c_oracle.roi_pool_forward(f, rois, 7, 7, 1.0 / 16, mode)
    # properties: output is the max of its window; argmax names an equal value or is -1
    flat = f.reshape(N, -1)
    nz = arg >= 0
    b = np.broadcast_to(rois[:, 0].astype(int)[:, None, None, None], arg.shape)
    assert np.array_equal(flat[b[nz], arg[nz]], top[nz])
    assert np.all(top[~nz] == 0)
    diff = rs.normal(size=top.shape).astype(np.float32)
    g_lit = c_oracle.roi_pool_backward(diff, arg, rois, f.shape, 7, 7, 1.0 / 16, literal=True)
    g_sc = c_oracle.roi_pool_backward(diff, arg, rois, f.shape, 7, 7, 1.0 / 16, literal=False)
    assert np.array_equal(g_lit, g_sc)
    # every routed top_diff is summed exactly once: total mass is conserved (f64)
    routed = np.where(nz, diff, 0).astype(np.float64).sum()
    assert abs(g_sc.astype(np.float64).sum() - routed) < 1e-3


def test_threaded_forward_matches_single_thread():
    rs = np.random.RandomState(9)
    f = rs.normal(size=(2, 12, 15, 8)).astype(np.float32)
    R = 40
    x1 = rs.uniform(0, 150, R)
    y1 = rs.uniform(0, 100, R)
    rois = np.stack([rs.randint(0, 2, R), x1, y1, x1 + rs.uniform(0, 90, R),
                     y1 + rs.uniform(0, 90, R)], axis=1).astype(np.float32)
    a = c_oracle.roi_pool_forward(f, rois, 7, 7, 1.0 / 16, "cuda", threads=1)
    b = c_oracle.roi_pool_forward(f, rois, 7, 7, 1.0 / 16, "cuda", threads=4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # backward: channel ranges on host threads == the single-threaded ordered scatter, bit for bit
    d = rs.normal(size=a[0].shape).astype(np.float32)
    g1 = c_oracle.roi_pool_backward(d, a[1], rois, f.shape, 7, 7, 1.0 / 16)
    g4 = c_oracle.roi_pool_backward(d, a[1], rois, f.shape, 7, 7, 1.0 / 16, threads=4)
    assert np.array_equal(g1, g4)
