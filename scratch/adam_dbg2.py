import math, torch, sys
sys.path.insert(0,'.')
from gps_slam_amd import gsplat_ops as ops
dev='cuda:0'
gen = torch.Generator(device="cpu").manual_seed(1)
shapes = [(50000, 3), (50000, 3), (50000, 4), (50000, 3), (50000, 15, 3), (50000, 1), (10, 3, 4)]
lrs = [1.6e-4 * 1.1 * 3.3, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2, 1e-3]
b1, b2, eps = 0.9, 0.999, 1e-15
P = [torch.randn(s, generator=gen).to(dev) for s in shapes]
Pe = [p.clone() for p in P]
M = [torch.zeros_like(p) for p in P]; V = [torch.zeros_like(p) for p in P]
Me = [torch.zeros_like(p) for p in P]; Ve = [torch.zeros_like(p) for p in P]
for step in range(1, 6):
    G = [torch.randn(s, generator=gen).to(dev) * (0.0 if (step == 3 and k % 2) else 1e-3) for k, s in enumerate(shapes)]
    for p, g, m, v, lr in zip(Pe, G, Me, Ve, lrs):
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-(lr / bc1))
    ops.adam_step(P, G, M, V, lrs, step, (b1, b2), eps)
    for name, A, B in (('p',P,Pe),('m',M,Me),('v',V,Ve)):
        bad=[int((a!=b).sum()) for a,b in zip(A,B)]
        if any(bad): 
            print(step, name, bad)
            for a,b in zip(A,B):
                i=(a!=b).flatten().nonzero()
                if len(i): 
                    j=int(i[0]); print('   ', a.flatten()[j].item(), b.flatten()[j].item(), G[[id(x) for x in A].index(id(a))].flatten()[j].item() if name!='x' else '')
print('done')
