#!/usr/bin/env python3
"""Random cases for the DEVICE samplers (cfg.SAMPLING_RNG = 'device': wssdl_anchor_subsample_device, wssdl_roi_sample_device -- a
counter-based device RNG, not NumPy's stream, so the checks are structural): the sub-sampled anchor labels are a subset of the
oracle's pre-sub-sampling labels with the reference's quotas (:202-217) and weights; the sampled RoIs are candidates of their
image, none more often than it occurs, within the fg / bg overlap bands and quotas (_sample_rois :228-280), with the oracle's labels,
regression targets (4 ulp) and weights; the same seed gives the same draw.
    python3 tools/sampler_fuzz.py [--cases 40] [--seed 0]"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from oracle import c_oracle, np_oracle as O  # noqa: E402
from wssdl_bus_amd.fast_rcnn.config import cfg  # noqa: E402
from wssdl_bus_amd.rpn_msr import proposal_target_layer_tf_bus as ptl  # noqa: E402
from wssdl_bus_amd.rpn_msr.anchor_target_layer_tf_bus import anchor_target_layer  # noqa: E402

MAX_GT = 20


def ulp(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, np.int64(-2 ** 31) - a, a)
    b = np.where(b < 0, np.int64(-2 ** 31) - b, b)
    return np.abs(a - b)


ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=40)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
rs = np.random.RandomState(args.seed)
old = cfg.SAMPLING_RNG, cfg.DEVICE_RNG_SEED, cfg.TRAIN.RPN_BATCHSIZE
bad = 0
try:
    for k in range(args.cases):
        why = []
        H, W = int(rs.randint(14, 64)), int(rs.randint(14, 101))
        im_h, im_w = H * 16 - int(rs.randint(0, 16)), W * 16 - int(rs.randint(0, 16))
        n = int(rs.randint(1, MAX_GT + 1))
        gt = np.zeros((MAX_GT, 5), np.float32)
        for j in range(n):
            bw, bh = float(np.exp(rs.normal(np.log(170), 0.7))), float(np.exp(rs.normal(np.log(150), 0.7)))
            x1, y1 = rs.uniform(0, 0.8) * im_w, rs.uniform(0, 0.8) * im_h
            gt[j] = [x1, y1, x1 + bw, y1 + bh, int(rs.choice([0, 1, 1, 2]))]
        if not (gt[:n, 4] > 0).any():
            gt[0, 4] = 1
        ng = np.array([n], np.int32)
        ii = np.array([[im_h, im_w, 1.0, 1]], np.float32)
        score = np.zeros((1, H, W, 18), np.float32)
        # ---- anchors: pre-sub-sampling labels from the oracle (a batch size nothing exceeds), then the device draw
        O_pre = O.anchor_target_layer(score, gt[None], ng, ii, None, (16,), (8, 16, 32), "SNUBH", rng=np.random.RandomState(0),
                                      cfg=dict(RPN_BATCHSIZE=10 ** 9))[0].astype(np.int8).reshape(-1)
        cfg.SAMPLING_RNG = "device"
        cfg.DEVICE_RNG_SEED = 100 + k
        outs = [anchor_target_layer(torch.from_numpy(score).cuda(), torch.from_numpy(gt[None]).cuda(), torch.from_numpy(ng).cuda(),
                                    torch.from_numpy(ii).cuda(), None, [16], [8, 16, 32], "SNUBH") for _ in range(2)]
        lab = outs[0][0].cpu().numpy().astype(np.int8).reshape(-1)
        n_fg_pre, n_bg_pre = int((O_pre == 1).sum()), int((O_pre == 0).sum())
        n_fg, n_bg = int((lab == 1).sum()), int((lab == 0).sum())
        if n_fg != min(n_fg_pre, 128) or n_bg != min(n_bg_pre, 256 - n_fg):
            why.append("anchor quotas %d/%d of %d/%d" % (n_fg, n_bg, n_fg_pre, n_bg_pre))
        if not (np.all(O_pre[lab == 1] == 1) and np.all(O_pre[lab == 0] == 0)):
            why.append("anchor labels not a subset of the pre-sub-sampling labels")
        w = outs[0][3].cpu().numpy()
        if (w > 0).any() and not np.allclose(w[w > 0], 1.0 / max(n_fg + n_bg, 1)):
            why.append("anchor outside weights")
        # ---- RoIs
        Ni = int(rs.randint(1, 4))
        gts = np.zeros((Ni, MAX_GT, 5), np.float32)
        ngs = np.zeros((Ni,), np.int32)
        rl = []
        for i in range(Ni):
            m = int(rs.randint(1, n + 1))
            gts[i, :m] = gt[rs.permutation(n)[:m]]
            if not (gts[i, :m, 4] > 0).any():
                gts[i, 0, 4] = 1
            # positives first, as the data layer hands them over (the layer appends gt[:npos])
            order = np.argsort(-(gts[i, :m, 4] > 0).astype(np.int32), kind="stable")
            gts[i, :m] = gts[i, :m][order]
            ngs[i] = m
            R = int(rs.randint(1, 600))
            src = gts[i, rs.randint(0, m, R), :4]
            jit = src + rs.normal(0, 25, size=(R, 4)) * (rs.uniform(size=(R, 1)) < 0.8)
            far = rs.uniform(0, 1, size=(R, 4)) * [im_w, im_h, im_w, im_h]
            b = np.where(rs.uniform(size=(R, 1)) < 0.6, jit, np.hstack((np.minimum(far[:, :2], far[:, 2:]), np.maximum(far[:, :2], far[:, 2:]))))
            if R > 4:
                b[1] = b[0]                                                    # duplicated candidates
            rl.append(np.hstack((np.full((R, 1), i), b)).astype(np.float32))
        rois = np.concatenate(rl)
        dev = torch.device("cuda", 0)
        args_d = (torch.from_numpy(rois).to(dev), torch.from_numpy(gts).to(dev), torch.from_numpy(ngs).to(dev))
        runs = []
        for rep in range(2):
            cfg.DEVICE_RNG_SEED = 500 + k
            ptl._device_calls[0] = 0
            runs.append([t.cpu().numpy() for t in ptl.proposal_target_layer(*args_d, 3, True, False)])
        if not all(np.array_equal(x, y) for x, y in zip(*runs)):
            why.append("roi draw not reproducible")
        out_rois, labels, tg, inw, outw = runs[0]
        rpi = int(cfg.TRAIN.BATCH_SIZE)
        fg_rpi = int(np.round(cfg.TRAIN.FG_FRACTION * rpi))
        row = 0
        for i in range(Ni):
            npos = int(np.sum(gts[i, :ngs[i], 4] != 0))
            cand = np.vstack([rois[rois[:, 0] == i], np.hstack([np.full((npos, 1), i, np.float32), gts[i, :npos, :4]])])
            ov = c_oracle.bbox_overlaps(cand[:, 1:5].astype(np.float64), gts[i, :npos, :4].astype(np.float64))
            mo, am = ov.max(axis=1), ov.argmax(axis=1)
            n_fg = min(fg_rpi, int(np.sum(mo >= cfg.TRAIN.FG_THRESH)))
            n_bg = min(rpi - n_fg, int(np.sum((mo < cfg.TRAIN.BG_THRESH_HI) & (mo >= cfg.TRAIN.BG_THRESH_LO))))
            # the device path keeps the shape fixed: every image owns rois_per_image rows, the ones beyond its quotas are padding
            # (-1, 0, 0, 0, 0), label -1, zero targets and weights (the reference returns fewer rows for such an image)
            row = i * rpi
            blk = out_rois[row:row + n_fg + n_bg]
            pad = slice(row + n_fg + n_bg, row + rpi)
            if blk.shape[0] != n_fg + n_bg or not np.all(blk[:, 0] == i):
                why.append("roi block of image %d: quotas %d + %d" % (i, n_fg, n_bg))
                break
            if not np.all(out_rois[pad, 0] == -1) or out_rois[pad, 1:].any() or not np.all(labels[pad, 0] == -1) or \
                    tg[pad].any() or inw[pad].any() or outw[pad].any():
                why.append("padding rows of image %d" % i)
            have = collections.Counter(tuple(r) for r in cand.tolist())
            drawn = collections.Counter(tuple(r) for r in blk.tolist())
            if any(drawn[t] > have.get(t, 0) for t in drawn):
                why.append("a row drawn more often than it occurs among the candidates (image %d)" % i)
            lookup = {}
            for j, r in enumerate(cand.tolist()):
                lookup.setdefault(tuple(r), j)
            idx = np.array([lookup.get(tuple(r), -1) for r in blk.tolist()], dtype=np.int64)
            if (idx < 0).any():
                why.append("a drawn row is no candidate (image %d)" % i)
                break
            if not np.all(mo[idx[:n_fg]] >= cfg.TRAIN.FG_THRESH) or \
                    not np.all((mo[idx[n_fg:]] < cfg.TRAIN.BG_THRESH_HI) & (mo[idx[n_fg:]] >= cfg.TRAIN.BG_THRESH_LO)):
                why.append("overlap bands (image %d)" % i)
            lab_i = labels[row:row + n_fg + n_bg, 0]
            if not np.array_equal(lab_i[:n_fg], gts[i, am[idx[:n_fg]], 4]) or not np.all(lab_i[n_fg:] == 0):
                why.append("labels (image %d)" % i)
            t = O.bbox_transform(blk[:n_fg, 1:5], gts[i, am[idx[:n_fg]], :4]).astype(np.float32)
            for q in range(n_fg):
                cls = int(lab_i[q])
                e = np.zeros(12, np.float32)
                e[4 * cls:4 * cls + 4] = t[q]
                wv = np.zeros(12, np.float32)
                wv[4 * cls:4 * cls + 4] = 1
                if ulp(tg[row + q], e).max() > 4 or not np.array_equal(inw[row + q], wv) or not np.array_equal(outw[row + q], wv):
                    why.append("targets / weights of a fg row (image %d)" % i)
                    break
            if tg[row + n_fg:row + n_fg + n_bg].any():
                why.append("bg rows carry targets (image %d)" % i)
        if not why and out_rois.shape[0] != Ni * rpi:
            why.append("row count %d against %d" % (out_rois.shape[0], Ni * rpi))
        if why:
            bad += 1
            print("MISMATCH case %d map %dx%d gt %d images %d: %s" % (k, H, W, n, Ni, "; ".join(why[:3])), flush=True)
        if (k + 1) % 10 == 0:
            print("case %d ok so far (%d mismatches)" % (k + 1, bad), flush=True)
finally:
    cfg.SAMPLING_RNG, cfg.DEVICE_RNG_SEED, cfg.TRAIN.RPN_BATCHSIZE = old
print("cases %d mismatches %d" % (args.cases, bad))
sys.exit(1 if bad else 0)
