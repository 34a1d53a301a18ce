import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wssdl_bus_amd.networks.backbones import ResNetTrunk, ResNetHead

def t(fn, n=2):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

cl = sys.argv[1] == "cl"
R = int(sys.argv[2])
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 50
mf = torch.channels_last if cl else torch.contiguous_format
head = ResNetHead(depth).cuda().to(memory_format=mf)
C = 1024 if depth >= 50 else 256
x = torch.randn(R, C, 7, 7, device="cuda").contiguous(memory_format=mf).requires_grad_(True)
def fwdbwd():
    y = head(x); y.sum().backward()
print("head depth", depth, "cl" if cl else "nchw", "R", R, "fwd+bwd ms", t(fwdbwd), flush=True)
if len(sys.argv) > 4:
    trunk = ResNetTrunk(depth).cuda().to(memory_format=mf)
    im = torch.randn(int(sys.argv[4]), 3, 600, 1000, device="cuda").contiguous(memory_format=mf)
    def tf():
        y = trunk(im); y.sum().backward()
    print("trunk images", sys.argv[4], "fwd+bwd ms", t(tf), flush=True)
