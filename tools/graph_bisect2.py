"""Bisect hipGraph capture of the ganomaly step phases (each in its own process)."""
import subprocess
import sys

PHASES = {
    "fwd_g": "m.forward_g()",
    "fwd_gd": "m.forward_g(); m.forward_d()",
    "bwd_g": "m.forward_g(); m.forward_d(); m.optimizer_g.zero_grad(); m.backward_g()",
    "bwd_g_step": "m.forward_g(); m.forward_d(); m.optimizer_g.zero_grad(); m.backward_g(); m.optimizer_g.step()",
    "bwd_d": "m.forward_g(); m.forward_d(); m.optimizer_d.zero_grad(); m.backward_d()",
    "full": "m.optimize_params(check_collapse=False)",
}
TEMPLATE = """
import faulthandler, torch, types, tempfile
faulthandler.enable()
from vfd_gan_amd import functional as F
from vfd_gan_amd.models import ganomaly as HG
from vfd_gan_amd.lib.data import synthetic_batch
F.set_compute_dtype(torch.{dt})
args = types.SimpleNamespace(batchsize=2, nfr=4, isize=32, ich=3, lr=2e-4, beta1=0.5, w_adv=1, w_con=50, freq=10**9, ep=1, model='ganomaly', result_root=tempfile.mkdtemp(), gpu=[0])
m = HG.Ganomaly(args, None, opt=HG.make_opt(isize=32, ngf=16))
m.set_input(synthetic_batch(2, 4, 32, 3, seed=1))
def fn():
    {body}
s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    fn(); fn()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
gr=torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    fn()
torch.cuda.synchronize()
gr.replay(); gr.replay(); torch.cuda.synchronize()
print('OK')
"""
for dt in ("float32", "bfloat16"):
    for name, body in PHASES.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        r = subprocess.run([sys.executable, "-c", TEMPLATE.format(body=body, dt=dt)], capture_output=True, text=True, timeout=300)
        tail = [t for t in (r.stdout + r.stderr).strip().splitlines() if "amdgpu.ids" not in t and "SAVE PATH" not in t and t.strip()][-2:]
        print("%-8s %-12s rc=%d %s" % (dt, name, r.returncode, " | ".join(t[:120] for t in tail)), flush=True)
