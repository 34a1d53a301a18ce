#!/usr/bin/env python3
"""CPU model of the bin-owner RoI-pool backward on the fixed roofline RoI set (no GPU needed): for a tile SH x SW and a
region RH x RW (tile + halo), how often a bin is listed (`f`: 1.0 = every top_diff / code row read exactly once; the exact
walk's 6x6 tiles: 1.58), the halo cells per tile cell, and the HBM bytes per launch over the bytes the launch must move
(walk reads f x bins, writes tile + halo cells; the merge reads the receiving cells + the halos and writes the receiving
cells).  The listing rule is the one of csrc/roi_pool_walk.hip: axis_entry_own (a window is listed by the tile of its
first line; a window that does not fit the region continues in the tile of its first uncovered line).
EXPERIMENTS.md, round 5, quotes these numbers; measured traffic: owner plan 2 (8x8 / 6x6) 1.198, plan 8 (6x7 / 4x5) 1.227.

    python3 tools/owner_model.py            -> one line per shape
"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W, C, N = 38, 63, 1024, 8


def windows(roi):
    """forward windows, CUDA rounding (roi_pooling_op_gpu.cu.cc:40-64)"""
    x1, y1, x2, y2 = [int(np.floor(v * 0.0625 + 0.5)) for v in roi[1:]]
    rw, rh = max(x2 - x1 + 1, 1), max(y2 - y1 + 1, 1)
    bw, bh = np.float32(rw) / np.float32(7), np.float32(rh) / np.float32(7)
    hs = [min(max(int(np.floor(np.float32(p) * bh)) + y1, 0), H) for p in range(7)]
    he = [min(max(int(np.ceil(np.float32(p + 1) * bh)) + y1, 0), H) for p in range(7)]
    ws = [min(max(int(np.floor(np.float32(p) * bw)) + x1, 0), W) for p in range(7)]
    we = [min(max(int(np.ceil(np.float32(p + 1) * bw)) + x1, 0), W) for p in range(7)]
    return hs, he, ws, we


def chain_pieces(s, e, tile, region):
    """[(tile index, first line, one past the last line)] of the window [s, e): the chain of axis_entry_own -- the tile of
    the first uncovered line lists the lines from there to the end of ITS region, and so on"""
    out, u = [], s
    while u < e:
        t = u // tile
        end = min(e, t * tile + region)
        out.append((t, u, end))
        u = t * tile + region
    return out


def listings(s, e, tile, region):
    """tiles (per axis) that list the window [s, e); an empty window is listed nowhere"""
    return len(chain_pieces(s, e, tile, region))


def model(rois, wins, sh, sw, rh, rw):
    slots = 0
    for hs, he, ws, we in wins:
        slots += sum(listings(hs[p], he[p], sh, rh) for p in range(7)) * sum(listings(ws[p], we[p], sw, rw) for p in range(7))
    bins = sum(sum(1 for p in range(7) if he[p] > hs[p]) * sum(1 for p in range(7) if we[p] > ws[p]) for hs, he, ws, we in wins)
    f = slots / max(bins, 1)
    halo = (rh * rw - sh * sw) / float(sh * sw)
    recv = 1.0 - (sh - (rh - sh)) * (sw - (rw - sw)) / float(sh * sw) if halo > 0 else 0.0
    base = len(rois) * 49 * C * 5
    fmap = N * H * W * C * 4
    traffic = base * f + fmap * (1 + halo) + (fmap * (halo + 2 * recv) if halo > 0 else 0)
    return dict(tile=(sh, sw), region=(rh, rw), f=round(f, 3), halo_cells_per_tile_cell=round(halo, 2),
                traffic_over_moved=round(traffic / (base + fmap), 3), lds_kib_128ch=round((rh * rw + 1) * 0.5, 1))


def main():
    rois = np.load(os.path.join(ROOT, "profiles", "roofline_rois_r8512.npy"))
    wins = [windows(x) for x in rois]
    for shape in [(6, 6, 6, 6), (4, 4, 6, 6), (4, 5, 6, 7), (5, 5, 7, 7), (6, 6, 8, 8), (5, 6, 7, 8), (4, 4, 5, 5), (8, 8, 10, 10)]:
        print(model(rois, wins, *shape))


if __name__ == "__main__":
    main()
