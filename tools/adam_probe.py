import torch, numpy as np
dev='cuda:0'
g=torch.Generator().manual_seed(0)
m=torch.randn(1<<20,generator=g).to(dev)*1e-3; gr=torch.randn(1<<20,generator=g).to(dev)*1e-3
b1=0.9
t=m.clone().mul_(b1).add_(gr,alpha=1-b1)
m64,g64=m.double(),gr.double()
mb=(m*torch.tensor(b1,dtype=torch.float32,device=dev))  # rounded product
fma=(g64*float(np.float32(1-b1))+mb.double()).float()
nofma=((gr*float(np.float32(1-b1)))+mb)
print('add_: fma match',(t==fma).float().mean().item(),'nofma match',(t==nofma).float().mean().item())
v=torch.rand(1<<20,generator=g).to(dev)*1e-6
b2=0.999
t=v.clone().mul_(b2).addcmul_(gr,gr,value=1-b2)
vb=v*torch.tensor(b2,dtype=torch.float32,device=dev)
om=float(np.float32(1-b2))
c1=((g64*om).float().double()*g64+vb.double()).float()   # fma(om*g, g, vb)
c2=((gr*om)*gr+vb)                                        # no fma
c3=((g64*g64).float().double()*om+vb.double()).float()   # fma(om, g*g, vb)
c4=((gr*gr)*om+vb)
for n,c in (('fma(om*g,g,vb)',c1),('(om*g)*g+vb',c2),('fma(om,g*g,vb)',c3),('(g*g)*om+vb',c4)):
    print('addcmul_',n,(t==c).float().mean().item())
p=torch.randn(1<<20,generator=g).to(dev); den=torch.rand(1<<20,generator=g).to(dev)+1e-3
ss=-1.234e-3
t=p.clone().addcdiv_(m,den,value=ss)
ssf=float(np.float32(ss))
d1=(p+(ssf*m)/den); d2=(p.double()+ ((m/den).double()*ssf)).float(); d3=p+ssf*(m/den)
d4=(p.double()+((m*ssf).float().double()/den.double()).float().double()).float()
for n,c in (('p+(ss*m)/den',d1),('fma(ss,m/den,p)',d2),('p+ss*(m/den)',d3)):
    print('addcdiv_',n,(t==c).float().mean().item())
sq=(v.sqrt()/0.0316).add_(1e-15)
e1=(v.sqrt()/float(np.float32(0.0316)))+float(np.float32(1e-15))
e2=(v.sqrt()*float(np.float32(1/0.0316)))+float(np.float32(1e-15))
print('denom div',(sq==e1).float().mean().item(),'mul-recip',(sq==e2).float().mean().item())
