import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch.nn.functional as TF
from vfd_gan_amd import functional as F, _lib
dev=torch.device('cuda',0)
torch.manual_seed(0)
for (xs,cout,k,p,name) in [((2,3,6,20,20),32,(3,3,3),(1,1,1),'cin8'), ((2,32,4,24,24),64,(3,3,3),(1,1,1),'halo64'), ((8,64,1,28,28),128,(1,4,4),(0,1,1),'igemm128'), ((8,128,1,14,14),256,(1,4,4),(0,1,1),'igemm256')]:
    x=(torch.rand(xs)*2-1).bfloat16().float(); w=((torch.rand((cout,xs[1])+k)*2-1)*0.2).bfloat16().float(); b=torch.rand(cout)
    s=(1,2,2) if name.startswith('igemm') else 1
    pre=TF.conv3d(x.double(),w.double(),b.double(),s,p)
    sums=F.new_stats_buffer(cout,dev)
    yc=F.conv(F.to_cl(x.to(dev),torch.bfloat16), torch.nn.Parameter(w.to(dev)), torch.nn.Parameter(b.to(dev)), s, p, 0, False, 0, 0.0, stats=sums)
    torch.cuda.synchronize()
    cp=F.cpad(cout)
    f=sums.view(F.STATS_REPLICAS,2,cp).sum(0).cpu()
    s1=pre.sum(dim=(0,2,3,4)); s2=(pre*pre).sum(dim=(0,2,3,4))
    print(name, 'sum relerr %.3e  sq relerr %.3e'%(float((f[0,:cout]-s1).abs().max()/s1.abs().max()), float((f[1,:cout]-s2).abs().max()/s2.abs().max())), sums.dtype)
