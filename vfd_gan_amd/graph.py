"""hipGraph capture of a whole training step.

Every kernel of the step is launched from Python through ctypes (~500 launches + the autograd tape per ganomaly
step); once the kernels got fast that host work (~25 ms) exceeded the GPU time (~20 ms).  All entry points of the
C ABI are capture-safe by construction (no allocation, no synchronisation, launches on the current stream; the Adam
step counter lives on the device), so the step — forward, losses, both backward passes, both Adam updates, filter
re-packing — is captured ONCE into a hipGraph (torch.cuda.CUDAGraph supplies the capture stream and the private
memory pool) and replayed per step.  Host-side decisions of the reference step stay outside the graph: loading the
next batch into the static input buffer, and ganomaly's ``err_d.item() < 1e-5 -> reinit_d()`` check.
"""
import gc

import torch

from . import functional as F


class GraphedStep:
    """Captured training step.  world_size == 1: one graph for the whole step.  Data parallel: the model's
    ``step_program()`` — a list of ("graph", fn) / ("reduce", reducer) / ("join", reducer) — is executed per step:
    every "graph" entry is captured once into its own hipGraph and replayed; collectives are never captured, they are
    issued between the graphs (asynchronously: "reduce" returns at once, the next graph runs beside the collective,
    "join" makes the stream wait for it)."""

    def __init__(self, model, warmup=2):
        self.model = model
        self.warmup = warmup
        self.program = None      # [(kind, fn / reducer, graph or None)]

    def _eager_step(self):
        m = self.model
        if "check_collapse" in m.optimize_params.__code__.co_varnames:
            m.optimize_params(check_collapse=False)
        else:
            m.optimize_params()

    def capture(self):
        """Run `warmup` eager steps on a side stream (allocator / autograd warm-up), then capture one step.
        Capturing only RECORDS the step (its kernels do not run): training state advances by `warmup` steps."""
        from . import dist as vdist
        if self.warmup < 1:
            # capturing a backward pass whose leaf-gradient streams were never initialised off the default stream makes
            # autograd insert a cross-stream wait on the (non-capturing) default stream: the capture is invalidated
            # and hipStreamEndCapture faults instead of returning an error (round 1, gpurun_out/t7.log)
            raise ValueError("at least one warm-up step on the side stream is required before capture")
        if F._TIMER[0] is not None:
            raise RuntimeError("kernel timing events cannot be recorded inside a graph capture")
        m = self.model
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(self.warmup):
                self._eager_step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if vdist.collectives_on():
            if not hasattr(m, "step_program"):
                raise NotImplementedError("%s has no step_program() for data-parallel graph capture" % type(m).__name__)
            program = m.step_program()
        else:
            program = [("graph", self._eager_step)]
        # Packed filter copies (functional._packed_filter) live in buffers that are allocated once and re-packed in place,
        # and every optimiser re-packs all copies of its filters in one launch right after its update (optim.Adam.step ->
        # functional.repack_owned), which is captured with the step: a replay therefore always reads current weights at
        # stable addresses.  (Round-2 history: copies used to be re-allocated on a miss, and a copy that was still valid
        # at a net's first use inside the capture was read at its pre-capture address for ever.)
        reducers = {id(r): r for kind, r in program if kind != "graph"}.values()
        for r in reducers:
            r.suspended = True         # no collective may be issued while a capture is open
        self.program = []
        pool = None
        # Cyclic garbage must not be collected while a capture is open: a collected CUDAGraph / private-pool block of an earlier
        # GraphedStep frees device memory from its destructor, which is not permitted on a capturing thread and aborts the
        # process (round 3, tests/test_ganomaly_step.py after other graph tests; torch >= 2.9 no longer collects on entry to
        # torch.cuda.graph unless torch.compiler.config.force_cudagraph_gc is set).  Collect now, hold the collector until done.
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            for kind, obj in program:
                if kind == "graph":
                    g = torch.cuda.CUDAGraph()
                    # capture_error_mode "thread_local": in the default "global" mode EVERY thread's capture-unsafe call
                    # fails while the capture is open, and RCCL's watchdog thread polls the events of earlier collectives
                    # (hipEventQuery) whenever it likes — found by the single-rank RCCL test (mygan, graph mode: "operation
                    # not permitted when stream is capturing" raised inside ProcessGroupNCCL's watchdog, process aborted)
                    # stream=s: the capture runs on the stream the warm-up steps ran on, so that autograd's leaf-gradient
                    # accumulation (AccumulateGrad remembers the stream of a gradient's first accumulation) needs no
                    # cross-stream wait inside the capture (the "AccumulateGrad stream mismatch ... may break CUDA graph
                    # capture" warning of rounds 1-2, and the mechanism behind round 1's capture_end fault)
                    with torch.cuda.graph(g, pool=pool, stream=s, capture_error_mode="thread_local"):
                        obj()
                        F.join_side_stream()      # a side stream forked inside the capture must rejoin before it ends
                    pool = g.pool()        # later phases read tensors the earlier ones allocated: share one pool
                    self.program.append((kind, obj, g))
                else:
                    self.program.append((kind, obj, None))
        finally:
            if gc_was_on:
                gc.enable()
            for r in reducers:
                r.suspended = False
                r.reset()
        torch.cuda.synchronize()
        return self

    @property
    def graphs(self):
        return [g for _, _, g in self.program if g is not None]

    def load_input(self, batch):
        """Copy a new (input, real, gt, lb) batch into the buffers the captured graph reads."""
        m = self.model
        old = {k: getattr(m, k, None) for k in ("x", "real_cl", "input_cl", "gt_cl", "gt_flow", "pre_flow")}
        m.set_input(batch)
        for k, o in old.items():
            n = getattr(m, k, None)
            if o is not None and n is not None and n is not o:
                o.t.copy_(n.t)
                setattr(m, k, o)

    def replay(self):
        for kind, obj, g in self.program:
            if kind == "graph":
                g.replay()
            elif kind == "reduce":
                obj.reduce_async()
            else:
                obj.join()
        m = self.model
        if hasattr(m, "d_collapsed") and m.d_collapsed():
            m.reinit_d()                   # reference models/ganomaly.py:519 (host decision, outside the graph)
