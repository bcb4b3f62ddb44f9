"""Find which op breaks hipGraph capture: runs each candidate in its own process."""
import subprocess
import sys

CASES = {
    "adam": "p=[torch.nn.Parameter(torch.randn(100,device=d))]; o=O.Adam(p,lr=1e-3); fn=lambda: (o.zero_grad(), o.step())",
    "tocl": "x=torch.randn(4,3,16,16,device=d); fn=lambda: F.to_cl(x)",
    "conv": "x=F.to_cl(torch.randn(4,8,16,16,device=d)); w=torch.nn.Parameter(torch.randn(16,8,4,4,device=d)); fn=lambda: F.conv(x,w,None,2,1)",
    "conv_stats": "x=F.to_cl(torch.randn(4,8,16,16,device=d)); w=torch.nn.Parameter(torch.randn(16,8,4,4,device=d)); fn=lambda: F.conv(x,w,None,2,1,stats=F.new_stats_buffer(16,d))",
    "conv_bwd": "x=F.to_cl(torch.randn(4,8,16,16,device=d).requires_grad_()); w=torch.nn.Parameter(torch.randn(16,8,4,4,device=d)); fn=lambda: F.conv(x,w,None,2,1).t.float().sum().backward()",
    "bn": "x=F.to_cl(torch.randn(4,8,16,16,device=d)); g=torch.nn.Parameter(torch.ones(8,device=d)); b=torch.nn.Parameter(torch.zeros(8,device=d)); rm=torch.zeros(8,device=d); rv=torch.ones(8,device=d); fn=lambda: F.bn_act(x,g,b,rm,rv,1e-5,0.1,1,0.2)",
    "loss": "x=F.to_cl(torch.rand(4,8,16,16,device=d)); y=F.to_cl(torch.rand(4,8,16,16,device=d)); fn=lambda: F.l1_loss(x,y)",
    "splitk": "x=F.to_cl(torch.randn(512,512,7,7,device=d)); w=torch.nn.Parameter(torch.randn(1,512,7,7,device=d)); fn=lambda: F.conv(x,w,None,1,0)",
    "torch_bwd": "x=torch.randn(100,device=d,requires_grad=True); fn=lambda: (x*2).sum().backward()",
    "tocl_bwd": "x=torch.randn(4,8,16,16,device=d,requires_grad=True); fn=lambda: F.to_cl(x).t.float().sum().backward()",
    "conv_dgrad_only": "x=F.to_cl(torch.randn(4,8,16,16,device=d).requires_grad_()); w=torch.randn(16,8,4,4,device=d); fn=lambda: F.conv(x,w,None,2,1).t.float().sum().backward()",
    "conv_wgrad_only": "x=F.to_cl(torch.randn(4,8,16,16,device=d)); w=torch.nn.Parameter(torch.randn(16,8,4,4,device=d)); fn=lambda: F.conv(x,w,None,2,1).t.float().sum().backward()",
    "convT_fwd": "x=F.to_cl(torch.randn(4,8,16,16,device=d)); w=torch.nn.Parameter(torch.randn(8,16,4,4,device=d)); fn=lambda: F.conv(x,w,None,2,1,0,True)",
    "convT_fwd_big": "x=F.to_cl(torch.randn(4,256,16,16,device=d)); w=torch.nn.Parameter(torch.randn(256,256,4,4,device=d)); fn=lambda: F.conv(x,w,None,2,1,0,True)",
    "conv_fwd_big": "x=F.to_cl(torch.randn(4,256,16,16,device=d)); w=torch.nn.Parameter(torch.randn(256,256,4,4,device=d)); fn=lambda: F.conv(x,w,None,2,1)",
    "conv_dgrad_s1": "x=F.to_cl(torch.randn(4,8,16,16,device=d).requires_grad_()); w=torch.randn(16,8,3,3,device=d); fn=lambda: F.conv(x,w,None,1,1).t.float().sum().backward()",
    "dgrad_leaf": "xt=torch.randn(4,1,16,16,8,device=d).bfloat16().requires_grad_(); x=F.ClTensor(xt,8,2); w=torch.randn(16,8,3,3,device=d); fn=lambda: F.conv(x,w,None,1,1).t.float().sum().backward()",
    "dgrad_leaf_gradset": "xt=torch.randn(4,1,16,16,8,device=d).bfloat16().requires_grad_(); xt.grad=torch.zeros_like(xt); x=F.ClTensor(xt,8,2); w=torch.randn(16,8,3,3,device=d); fn=lambda: F.conv(x,w,None,1,1).t.float().sum().backward()",
    "dgrad_autograd_grad": "xt=torch.randn(4,1,16,16,8,device=d).bfloat16().requires_grad_(); x=F.ClTensor(xt,8,2); w=torch.randn(16,8,3,3,device=d); fn=lambda: torch.autograd.grad(F.conv(x,w,None,1,1).t.float().sum(), xt)",
    "torch_only": "x=torch.randn(100,device=d); fn=lambda: (x*2).sum()",
}
TEMPLATE = """
import faulthandler, torch
faulthandler.enable()
from vfd_gan_amd import functional as F, optim as O
d=torch.device('cuda',0)
F.set_compute_dtype(torch.bfloat16)
{setup}
s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    fn(); fn()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
gr=torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    out=fn()
torch.cuda.synchronize()
gr.replay(); gr.replay(); torch.cuda.synchronize()
print('OK')
"""
for name, setup in CASES.items():
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    r = subprocess.run([sys.executable, "-c", TEMPLATE.format(setup=setup)], capture_output=True, text=True, timeout=300)
    tail = [t for t in (r.stdout + r.stderr).strip().splitlines() if "amdgpu.ids" not in t][-3:]
    print("%-12s rc=%d %s" % (name, r.returncode, " | ".join(t[:110] for t in tail)), flush=True)
