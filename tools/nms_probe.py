import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from kernel_bench import synth_rpn
from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import proposal_layer_padded
from wssdl_bus_amd.nms.hip_nms import hip_nms
N,H,W=8,38,63
info = torch.tensor([[600, 1000, 1.0, 1.0]] * N, device="cuda")
prob, pred = synth_rpn(N, H, W, 9, 3)
rp, cnt, dec, sidx, scnt = proposal_layer_padded(prob, pred, info, True, debug=True)
print("counts", cnt.tolist(), "sorted", scnt.tolist())
for i in (0, 5):
    n = int(scnt[i]); order = sidx[i,:n].long()
    boxes = dec[i][order]
    dets = torch.cat([boxes, torch.arange(n,0,-1,device="cuda").float().unsqueeze(1)],1).contiguous()
    keep = hip_nms(dets, 0.7)
    keep = np.asarray(keep.cpu() if hasattr(keep, "cpu") else keep)
    print("image", i, "n", n, "kept total", len(keep), "index of 2000th kept:", keep[1999] if len(keep)>=2000 else None,
          "kept among first 2048/4096/8192:", (keep<2048).sum(), (keep<4096).sum(), (keep<8192).sum())
# the real network at init
from wssdl_bus_amd import synthetic
from wssdl_bus_amd.networks.factory_bus import get_network
from wssdl_bus_amd.fast_rcnn.config import cfg
cfg.TRAIN.IMS_PER_BATCH, cfg.TRAIN.WS_IMS_PER_BATCH = 4, 4; cfg.SAMPLING_RNG="device"
torch.manual_seed(3)
net = get_network("Resnet_train", 50).cuda().to(memory_format=torch.channels_last); net.train()
blobs = synthetic.make_batch(4, 4, 600, 1000, 3)
with torch.no_grad():
    L = net(blobs["data"], blobs["im_info"], blobs["gt_boxes"], blobs["num_gt_boxes"], is_training=True, is_ws=False)
rp, cnt, dec, sidx, scnt = proposal_layer_padded(L["rpn_cls_score"], L["rpn_bbox_pred"], blobs["im_info"], True, debug=True, from_logits=True)
print("net counts", cnt.tolist())
for i in (0, 5):
    n = int(scnt[i]); order = sidx[i,:n].long()
    dets = torch.cat([dec[i][order], torch.arange(n,0,-1,device="cuda").float().unsqueeze(1)],1).contiguous()
    k_ = hip_nms(dets, 0.7); keep = np.asarray(k_.cpu() if hasattr(k_, "cpu") else k_)
    print("net image", i, "n", n, "kept total", len(keep), "index of 2000th kept:", keep[1999] if len(keep)>=2000 else None,
          "kept among first 2048/4096/8192:", (keep<2048).sum(), (keep<4096).sum(), (keep<8192).sum())
