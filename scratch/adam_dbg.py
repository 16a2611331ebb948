import math, torch, sys
sys.path.insert(0,'.')
from gps_slam_amd import gsplat_ops as ops
dev='cuda:0'
gen = torch.Generator(device="cpu").manual_seed(1)
shapes = [(50000, 3), (50000, 15, 3), (50000, 1)]
lrs = [1.6e-4 * 1.1 * 3.3, 1.25e-4, 5e-2]
b1, b2, eps = 0.9, 0.999, 1e-15
P = [torch.randn(s, generator=gen).to(dev) for s in shapes]
Pe = [p.clone() for p in P]
M = [torch.zeros_like(p) for p in P]; V = [torch.zeros_like(p) for p in P]
Me = [torch.zeros_like(p) for p in P]; Ve = [torch.zeros_like(p) for p in P]
for step in range(1, 4):
    G = [torch.randn(s, generator=gen).to(dev) * 1e-3 for s in shapes]
    for p, g, m, v, lr in zip(Pe, G, Me, Ve, lrs):
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-(lr / bc1))
    ops.adam_step(P, G, M, V, lrs, step, (b1, b2), eps)
    for name, A, B in (('p',P,Pe),('m',M,Me),('v',V,Ve)):
        print(step, name, [int((a!=b).sum()) for a,b in zip(A,B)], [float((a-b).abs().max()) for a,b in zip(A,B)])
