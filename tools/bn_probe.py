import torch, time
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
for M, C in ((8512*16, 2048), (8512*49, 512), (8512*16, 512)):
    x = torch.randn(M, C, device="cuda"); g = torch.randn(M, C, device="cuda")
    bn = torch.nn.BatchNorm1d(C).cuda()
    xr = x.clone().requires_grad_(True)
    def bn_fb():
        y = bn(xr); y.backward(g)
    ones = torch.ones(1, M, device="cuda")
    print(M, C, "bytes GB", M*C*4/1e9)
    print("  bn fwd+bwd      %.2f ms" % t(bn_fb))
    print("  bn fwd only     %.2f ms" % t(lambda: bn(x)))
    print("  var_mean dim0   %.2f ms" % t(lambda: torch.var_mean(x, dim=0, unbiased=False)))
    print("  sum dim0        %.2f ms" % t(lambda: x.sum(0)))
    print("  ones @ x        %.2f ms" % t(lambda: ones @ x))
    print("  (x*x).sum(0)    %.2f ms" % t(lambda: (x*x).sum(0)))
    print("  x*g sum0        %.2f ms" % t(lambda: (x*g).sum(0)))
    print("  elementwise a*x+b %.2f ms" % t(lambda: torch.addcmul(x[0], x, g[0])))
    print("  relu            %.2f ms" % t(lambda: torch.relu(x)))
