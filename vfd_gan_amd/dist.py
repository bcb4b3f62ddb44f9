"""Data parallelism for the GAN step: one process per GPU, gradients summed with RCCL over xGMI.

Replaces the reference's single-process ``torch.nn.DataParallel`` (models/mygannet.py:232-237,
models/anogan.py:126-131).  Numerical contract: mean-loss gradient over the GLOBAL batch,
BatchNorm statistics per replica (unsynchronised), replicas start from identical weights (broadcast from rank 0).

The reducer works on the flat gradient arena of :class:`vfd_gan_amd.optim.Adam`: parameters are grouped into a
few large contiguous buckets (xGMI is point-to-point, ~153 GB/s per link: few large collectives beat many small
ones); a bucket's all-reduce is launched from autograd's post-accumulate hooks as soon as its last gradient has
landed, so it overlaps the rest of the backward pass; ``finish()`` joins before the optimiser step.  The 1/world
factor is folded into the Adam kernel (``optimizer.grad_scale``), not applied as a separate pass.
"""
import os

import torch
import torch.distributed as tdist


def is_initialized():
    return tdist.is_available() and tdist.is_initialized()


def rank():
    return tdist.get_rank() if is_initialized() else 0


def world_size():
    return tdist.get_world_size() if is_initialized() else 1


def _single_forced():
    # VFD_DIST_SINGLE=1: a ONE-process group is created anyway and every collective of the data-parallel path is issued
    # (all-reduce of one rank = identity).  This is how the RCCL calls themselves — communicator set-up, asynchronous
    # handles from autograd hooks, wait() stream semantics next to the side stream and between replayed graphs — are
    # exercised on a one-GPU box (tests/test_ddp_gpu.py::test_single_rank_rccl_*).
    return bool(int(os.environ.get("VFD_DIST_SINGLE", "0") or 0))


def collectives_on():
    """True when the data-parallel machinery (broadcasts, gradient reductions, phase graphs) is active."""
    return world_size() > 1 or (is_initialized() and _single_forced())


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun); no-op for a single process.
    backend defaults to "nccl" (= RCCL on ROCm) when a GPU is present, else "gloo"."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if is_initialized() or (ws <= 1 and not _single_forced()):
        return rank(), world_size()
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    use_gpu = torch.cuda.is_available()
    if backend is None:
        backend = os.environ.get("VFD_DIST_BACKEND") or ("nccl" if use_gpu else "gloo")
    if use_gpu:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    tdist.init_process_group(backend=backend, rank=int(os.environ["RANK"]), world_size=ws)
    return rank(), world_size()


def barrier():
    """Process-group barrier bound to this rank's device (RCCL wants the device named; gloo does not take one)."""
    if collectives_on():
        if tdist.get_backend() == "nccl" and torch.cuda.is_available():
            tdist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            tdist.barrier()


def broadcast_module(module, src=0):
    """Make every replica start from rank `src`'s parameters and buffers (DataParallel replicates each forward)."""
    if not collectives_on():
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            tdist.broadcast(t.data, src=src)


def _join_side():
    from . import functional as F      # filter gradients issued on the side stream must have landed (functional._SIDE)
    F.join_side_stream()


def make_buckets(slices, bucket_elems):
    """Group consecutive (offset, numel) parameter slices — walked in REVERSE registration order, which is roughly
    the order gradients become ready — into contiguous [lo, hi) ranges of about `bucket_elems` elements.
    Returns (buckets, owner) with buckets = [(lo, hi, [param indices])], owner[i] = bucket of parameter i."""
    buckets, owner = [], [None] * len(slices)
    cur, lo, hi = [], None, None
    for i in range(len(slices) - 1, -1, -1):
        o, n = slices[i]
        if lo is None:
            lo, hi = o, o + n
        lo, hi = min(lo, o), max(hi, o + n)
        cur.append(i)
        if hi - lo >= bucket_elems:
            buckets.append((lo, hi, cur))
            cur, lo, hi = [], None, None
    if cur:
        buckets.append((lo, hi, cur))
    for b, (_, _, idxs) in enumerate(buckets):
        for i in idxs:
            owner[i] = b
    return buckets, owner


class GradReducer:
    """Bucketed, backward-overlapped gradient all-reduce over a flat gradient arena."""

    def __init__(self, params, grad_arena, slices, bucket_mb=32.0):
        self.params = list(params)
        self.arena = grad_arena
        self.world = world_size()
        self.collective = collectives_on()
        self.buckets, self.owner = make_buckets(slices, int(bucket_mb * (1 << 20) / 4))
        self.pending = [0] * len(self.buckets)
        self.launched = [False] * len(self.buckets)
        self.handles = []
        self.passes = 1
        self.enabled = True      # toggled by the step (a frozen net's gradients are not reduced)
        self.suspended = False   # set while a hipGraph capture is open: no collective may be issued then
        self._hooks = []
        for i, p in enumerate(self.params):
            # fires once per backward after the parameter's LAST gradient contribution — also when the backward kernels
            # accumulated in place and handed autograd None (functional._direct_grad)
            self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        self.reset()

    @classmethod
    def for_optimizer(cls, opt, bucket_mb=32.0):
        r = cls(opt._params, opt.grad_arena, opt.slices(), bucket_mb)
        opt.grad_scale = 1.0 / r.world
        return r

    def arm(self, passes=1):
        """Re-arm for the next optimiser step.  `passes` = number of backward passes that deposit gradients into this
        net before finish() (AnoGAN's discriminator: two, reference models/anogan.py:233-241): a bucket's all-reduce
        is launched when its LAST parameter has received its LAST contribution, i.e. after `passes` hook firings per
        parameter, so the sum of all passes is reduced and no later backward writes into a bucket RCCL is reading."""
        self.passes = passes
        for b, (_, _, idxs) in enumerate(self.buckets):
            self.pending[b] = passes * sum(1 for i in idxs if self.params[i].requires_grad)
            self.launched[b] = False
        self.handles = []

    def reset(self):
        self.arm(1)

    def _launch(self, b):
        _join_side()
        if self.launched[b]:
            return
        self.launched[b] = True
        if self.collective:
            lo, hi, _ = self.buckets[b]
            self.handles.append(tdist.all_reduce(self.arena[lo:hi], op=tdist.ReduceOp.SUM, async_op=True))

    def _make_hook(self, i):
        def hook(_param):
            if not self.enabled or self.suspended:
                return
            b = self.owner[i]
            self.pending[b] -= 1
            if self.pending[b] == 0:
                self._launch(b)
        return hook

    def reduce_async(self):
        """Issue every bucket's all-reduce now, asynchronously (graph mode: the backward that produced the gradients is a
        graph replay, no autograd hook fires).  RCCL's stream waits for the work already queued on the current stream
        and the current stream does NOT wait for RCCL: whatever is launched next (the other net's backward graph) runs
        beside the collective.  join() makes the current stream wait."""
        _join_side()
        for b in range(len(self.buckets)):
            self.launched[b] = False
            self._launch(b)

    def join(self):
        for h in self.handles:
            h.wait()
        self.reset()

    def reduce_all(self):
        """All buckets now, and wait (blocking form of reduce_async + join)."""
        _join_side()
        self.reduce_async()
        self.join()

    def finish(self):
        """Launch whatever has not been launched (parameters that received no gradient), wait for all buckets
        (the current stream waits; the host does not block with RCCL), and re-arm for the next backward."""
        _join_side()
        for b in range(len(self.buckets)):
            self._launch(b)
        self.join()
