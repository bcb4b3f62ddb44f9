import os, sys, types, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from vfd_gan_amd import functional as F, trainer
from vfd_gan_amd.trainer import build_model
from vfd_gan_amd.lib.data import synthetic_batch
from test_trainer_resume import _args, _step
which = sys.argv[1]
tmp = tempfile.mkdtemp()
F.set_compute_dtype(torch.float32); F.dropout_manual_seed(1234); torch.manual_seed(11); torch.cuda.manual_seed(11)
a = _args(os.path.join(tmp, "r1"), which)
m1 = trainer.main(a)
ck = m1.last_checkpoint
print("rng after save", torch.cuda.get_rng_state()[:16].tolist(), F.dropout_state())
batch = synthetic_batch(a.batchsize, a.nfr, a.isize, 3, seed=999)
torch.manual_seed(777); torch.cuda.manual_seed(777); F.dropout_manual_seed(4321)
m2 = build_model(_args(os.path.join(tmp, "r2"), which, resume=ck, ep=2), None)
print("rng after load", torch.cuda.get_rng_state()[:16].tolist(), F.dropout_state())
for (k, v), (_, r) in zip(list(m1.netg.state_dict().items()) + list(m1.netd.state_dict().items()), list(m2.netg.state_dict().items()) + list(m2.netd.state_dict().items())):
    if not torch.equal(v, r): print("param differs before step", k)
o1, o2 = m1._optimizers(), m2._optimizers()
for a_, b_ in zip(o1, o2):
    print("adam equal:", torch.equal(a_.exp_avg, b_.exp_avg), torch.equal(a_.exp_avg_sq, b_.exp_avg_sq), int(a_._step_dev), int(b_._step_dev), a_.param_groups[0]["lr"], b_.param_groups[0]["lr"], a_._bc_dev.tolist(), b_._bc_dev.tolist())
st = torch.cuda.get_rng_state()
m1.set_input(batch); _step(m1); e1 = m1.errors()
torch.cuda.set_rng_state(st)
F.set_dropout_state({"seed": 1234, "step": 2}, m2.device)
m2.set_input(batch); _step(m2); e2 = m2.errors()
for k in e1: print(k, e1[k], e2[k])
