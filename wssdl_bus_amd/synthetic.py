"""Seeded synthetic inputs shaped like the reference's blobs.

  data        [N, im_h, im_w, 3] f32: one speckle-like grayscale plane (Rayleigh noise times a
              smooth mask, clipped to [0,1]) replicated x3 like roi_data_layer/minibatch_bus.py:270,
              normalised with the ResNet mean/std (config.py:284-287)
  im_info     [N, 4] f32 (height, width, scale, birads_diag+1)   minibatch_bus.py:46
  gt_boxes    [N, MAX_GT_PER_IMAGE, 5] f32 (x1,y1,x2,y2,cls), positives first; 1-2 positive
              boxes (class 1/2) and 1-3 background boxes (class 0) per supervised image,
              sizes 60-550 px; weak images have no boxes
  num_gt_boxes [N] i32
There is no network access for datasets: everything here is generated, and bench.py says so.
"""
import numpy as np
import torch

PIXEL_MEAN = 68.274 / 255.0
PIXEL_STD = 52.802 / 255.0


def make_gt(rs, im_h, im_w, max_gt=20):
    gt = np.zeros((max_gt, 5), dtype=np.float32)
    n_pos = rs.randint(1, 3)
    n_bg = rs.randint(1, 4)
    k = 0
    for j in range(n_pos + n_bg):
        bw = rs.randint(60, min(550, im_w - 20))
        bh = rs.randint(60, min(550, im_h - 20))
        x1 = rs.uniform(0, im_w - bw - 1)
        y1 = rs.uniform(0, im_h - bh - 1)
        cls = rs.randint(1, 3) if j < n_pos else 0
        gt[k] = [x1, y1, x1 + bw, y1 + bh, cls]
        k += 1
    return gt, k


def make_batch(n_sup, n_ws, im_h=600, im_w=1000, seed=3, device="cuda", max_gt=20):
    """Blobs for n_sup supervised images followed by n_ws weak images."""
    rs = np.random.RandomState(seed)
    n = n_sup + n_ws
    g = torch.Generator(device="cpu").manual_seed(seed)
    # Rayleigh(sigma=0.25) = sigma * sqrt(-2 ln U); smooth mask = low-res noise upsampled
    u = torch.rand((n, 1, im_h, im_w), generator=g).clamp_min(1e-7)
    ray = 0.25 * torch.sqrt(-2.0 * torch.log(u))
    low = torch.rand((n, 1, max(im_h // 50, 2), max(im_w // 50, 2)), generator=g)
    mask = torch.nn.functional.interpolate(low, size=(im_h, im_w), mode="bilinear", align_corners=False)
    plane = (ray * (0.5 + mask)).clamp(0, 1)
    plane = (plane - PIXEL_MEAN) / PIXEL_STD
    data = plane.permute(0, 2, 3, 1).expand(n, im_h, im_w, 3).contiguous()
    gt = np.zeros((n, max_gt, 5), dtype=np.float32)
    ng = np.zeros((n,), dtype=np.int32)
    info = np.zeros((n, 4), dtype=np.float32)
    for i in range(n):
        info[i] = [im_h, im_w, 1.0, rs.randint(1, 3)]
        if i < n_sup:
            gt[i], ng[i] = make_gt(rs, im_h, im_w, max_gt)
    return dict(data=data.to(device), im_info=torch.from_numpy(info).to(device),
                gt_boxes=torch.from_numpy(gt).to(device), num_gt_boxes=torch.from_numpy(ng).to(device))
