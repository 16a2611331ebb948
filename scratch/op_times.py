import torch, time
d='cuda:0'
g=torch.Generator(device=d).manual_seed(0)
def tm(name, fn, n=5):
    fn(); torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print('%-30s %.3f ms'%(name,1000*(time.perf_counter()-t)/n))
n=30000
x=torch.rand(480,640,3,device=d); m=(torch.rand(480,640,1,device=d)>0.9).expand(480,640,3)
tm('rand(gen)', lambda: torch.rand(n,device=d,generator=g))
r=torch.rand(n,device=d)
tm('argsort', lambda: torch.argsort(r))
tm('sort', lambda: torch.sort(r))
tm('masked_select', lambda: torch.masked_select(x,m))
tm('nonzero', lambda: torch.nonzero(m[...,0]))
tm('randperm cuda gen', lambda: torch.randperm(n,device=d,generator=g))
tm('randperm cuda nogen', lambda: torch.randperm(n,device=d))
cg=torch.Generator().manual_seed(0)
tm('randperm cpu->gpu', lambda: torch.randperm(n,generator=cg)[:n//4].to(d))
mask=torch.rand(200000,device=d)>0.1
tm('nonzero 200k', lambda: torch.nonzero(~mask))
tm('mask.sum item', lambda: int(mask.sum()))
