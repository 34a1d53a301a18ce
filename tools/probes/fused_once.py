import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import proposal_layer_padded
rs = np.random.RandomState(5)
N, H, W, A = 8, 38, 63, 9
prob = torch.from_numpy(rs.uniform(0.01, 0.99, size=(N, H, W, 2 * A)).astype(np.float32)).cuda()
pred = torch.from_numpy(rs.normal(0, 0.5, size=(N, H, W, 4 * A)).astype(np.float32)).cuda()
info = torch.from_numpy(np.tile(np.array([[16 * H - 8, 16 * W - 8, 1.0, 1]], np.float32), (N, 1))).cuda()
out = proposal_layer_padded(prob, pred, info, True)
torch.cuda.synchronize()
print(out[1].cpu().tolist())
