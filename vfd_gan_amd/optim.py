"""Flat-arena Adam: the optimiser of the GAN step (``optim.Adam(net.parameters(), lr, betas=(beta1, 0.999))``,
reference models/mygannet.py:270-273, models/anogan.py:139-140, models/ganomaly.py:455-456) as ONE HIP launch.

All parameters of a net are re-homed into one contiguous float32 arena (each ``param.data`` becomes a view),
gradients into a second arena (``param.grad`` views, so autograd accumulates in place), moments into two more.
That is the MI355X-first layout: one streaming kernel over 4 arenas per step instead of 4 kernels per tensor, and
gradient buckets for the RCCL reducer are plain slices of the gradient arena.
"""
import torch

from . import _lib
from . import functional as F
from ._lib import check, load, stream

_ALIGN = 64  # floats (256 B): every parameter slice starts on a cache-line boundary


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("the reference uses plain Adam (no weight decay, no amsgrad)")
        params = [p for p in params]
        if not params:
            raise ValueError("optimizer got an empty parameter list")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._params = [p for g in self.param_groups for p in g["params"]]
        dev = self._params[0].device
        _lib.require_device(self._params[0])
        offs, total = [], 0
        for p in self._params:
            if p.dtype != torch.float32:
                raise TypeError("master parameters must be float32")
            offs.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self._offsets, self._total = offs, total
        self._epoch = [0]
        self.param_arena = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad_arena = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(self._params, offs):
                n = p.numel()
                view = self.param_arena[o:o + n].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad_arena[o:o + n].view(p.shape)
                p._vfd_direct_grad = True    # backward kernels accumulate straight into the arena (functional._direct_grad)
                p._vfd_epoch = self._epoch   # bumped by step(): invalidates THIS optimiser's packed filter copies only
        self._step = 0
        self._step_dev = torch.zeros(1, dtype=torch.int32, device=dev)      # device-side counter (graph replay)
        self._bc_dev = torch.zeros(2, dtype=torch.float32, device=dev)
        self.grad_scale = 1.0  # set to 1/world_size when gradients are summed over ranks
        F.invalidate_weight_cache()

    # gradient slices for the data-parallel reducer: (offset, numel) per parameter, in registration order
    def slices(self):
        return [(o, p.numel()) for p, o in zip(self._params, self._offsets)]

    def zero_grad(self, set_to_none=False):
        # the arena views must stay attached, so gradients are zeroed, never dropped
        F.zero_(self.grad_arena)
        for p, o in zip(self._params, self._offsets):
            if p.grad is None or p.grad.data_ptr() != self.grad_arena.data_ptr() + 4 * o:
                p.grad = self.grad_arena[o:o + p.numel()].view(p.shape)

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closure")
        F.join_side_stream()         # filter gradients issued on the side stream (functional._SIDE) land in the arena first
        g = self.param_groups[0]
        self._step += 1      # host mirror; the kernel uses the device counter so a captured step can be replayed
        b1, b2 = g["betas"]
        check(load().vfd_adam_step_dev(self.param_arena.data_ptr(), self.grad_arena.data_ptr(), self.exp_avg.data_ptr(),
                                       self.exp_avg_sq.data_ptr(), self._total, float(g["lr"]), float(b1), float(b2),
                                       float(g["eps"]), self._step_dev.data_ptr(), self._bc_dev.data_ptr(),
                                       float(self.grad_scale), stream()), "adam_step_dev")
        self._epoch[0] += 1          # the kernel wrote through raw pointers: no torch version bump
        F.repack_owned(self._epoch)  # ... and every packed (bf16, K-major) copy of these filters follows in one launch

    # ---- resume support (the reference saves no optimiser state) -------------------------------
    def state_dict(self):
        return {"step": int(self._step_dev.item()), "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        self._step = int(sd["step"])
        self._step_dev.fill_(self._step)
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for g, s in zip(self.param_groups, sd.get("param_groups", [])):
            g.update(s)
