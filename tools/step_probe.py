import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from wssdl_bus_amd import synthetic
from wssdl_bus_amd.fast_rcnn.config import cfg
from wssdl_bus_amd.fast_rcnn import train_bus
from wssdl_bus_amd.networks.factory_bus import get_network
n_s, n_ws = int(sys.argv[1]), int(sys.argv[2])
depth = int(sys.argv[3])
cfg.TRAIN.IMS_PER_BATCH, cfg.TRAIN.WS_IMS_PER_BATCH = n_s, n_ws
cfg.SAMPLING_RNG = "device"
net = get_network("Resnet_train", depth).cuda().to(memory_format=torch.channels_last)
blobs = synthetic.make_batch(n_s, n_ws)
def sync():
    torch.cuda.synchronize(); return time.perf_counter()
# hook timings on submodules
marks = []
def mk(name):
    def pre(m, i): marks.append((name + ":pre", sync()))
    def post(m, i, o): marks.append((name + ":post", sync()))
    return pre, post
for name in ("trunk", "rpn_conv", "head"):
    pre, post = mk(name)
    getattr(net, name).register_forward_pre_hook(pre)
    getattr(net, name).register_forward_hook(post)
solver = train_bus.SolverWrapper(net)
for it in range(3):
    marks.clear()
    t0 = sync()
    layers = net(blobs['data'], blobs['im_info'], blobs['gt_boxes'], blobs['num_gt_boxes'], True, False)
    t1 = sync()
    losses = train_bus.supervised_loss(layers, net.weight_decay_params(), n_s)
    t2 = sync()
    losses['loss'].backward()
    t3 = sync()
    solver._apply()
    t4 = sync()
    print("iter", it, "R", layers['roi-data'][0].shape[0], "fwd %.1f loss %.1f bwd %.1f opt %.1f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3), flush=True)
    prev = t0
    for n, t in marks:
        print("    %-16s +%.1f ms" % (n, (t - prev) * 1e3)); prev = t
    print("    end              +%.1f ms" % ((t1 - prev) * 1e3), flush=True)
