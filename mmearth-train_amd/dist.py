"""Data-parallel step runner: one process per GPU, RCCL (torch.distributed backend "nccl")
gradient all-reduce over xGMI, overlapped with the rest of backward on a side HIP stream.

Reference behaviour reproduced (/root/reference): DistributedDataParallel gradient averaging
(main_pretrain.py:306-310), rank-local GRN statistics, per-rank mask noise, scalar loss
all-reduce for logging (engine_pretrain.py:104 -> helpers.py:393-401). Not reproduced on
purpose: DDP's 25 MB bucketing heuristics and its per-iteration host syncs.

The flat fp32 gradient buffer is cut into three contiguous buckets that become final at known
points of the backward program (parameters are laid out in state-dict order):
    bucket 0  [proj ... end]            ready after heads + decoder + proj backward
    bucket 1  [stages.2, stages.3]      ready after the stage-3 and stage-2 blocks
    bucket 2  [0 ... stages.2)          initial conv, stem, downsample layers, stages 0-1 (last)
Each bucket is all-reduced (SUM) as soon as its segment of the backward program has been
enqueued; averaging (1/world) is folded into the AdamW kernel's grad_scale. With HIP graphs the
forward, every backward segment and the optimizer are separate captured graphs, so a step is
six graph launches plus three collectives.
"""
import os

import torch
import torch.distributed as dist


# ----------------------------------------------------------------------------- process group
def init(backend="nccl", local_rank=0):
    if dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    kw = {}
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        kw["device_id"] = torch.device("cuda", local_rank)
    dist.init_process_group(backend=backend, init_method="env://", **kw)


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def barrier():
    if dist.is_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def max_over_ranks(x: float) -> float:
    if not dist.is_initialized():
        return x
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def mean_scalar(x: float) -> float:
    """helpers.all_reduce_mean (/root/reference/helpers.py:393-401)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return x
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([x], dtype=torch.float32, device=dev)
    dist.all_reduce(t)
    return float(t.item()) / dist.get_world_size()


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()


# ----------------------------------------------------------------------------- bucket planning
def plan_buckets(offsets, n_params):
    """offsets: OrderedDict key -> (offset, numel) in flat order. Returns [(lo, hi)] x 3 in the
    order the buckets become ready during backward."""
    keys = list(offsets.keys())
    first_proj = next(k for k in keys if not k.startswith("encoder."))
    first_s2 = next(k for k in keys if k.startswith("encoder.stages.2."))
    lo_proj = offsets[first_proj][0]
    lo_s2 = offsets[first_s2][0]
    # everything from stages.2 up to the first non-encoder tensor must be stages.2/.3 only
    for k in keys:
        o = offsets[k][0]
        if lo_s2 <= o < lo_proj:
            assert k.startswith("encoder.stages.2.") or k.startswith("encoder.stages.3."), k
    return [(lo_proj, n_params), (lo_s2, lo_proj), (0, lo_s2)]


def split_bwd_segments(bwd_ops):
    """Cut the backward launch list where each bucket becomes final. Returns 3 lists of ops."""
    names = [op[0] for op in bwd_ops]
    i1 = next(i for i, n in enumerate(names) if n.startswith("encoder.stages.3."))
    i2 = next(i for i, n in enumerate(names) if n.startswith("encoder.downsample_layers.1"))
    assert 0 < i1 < i2 < len(names)
    return [bwd_ops[:i1], bwd_ops[i1:i2], bwd_ops[i2:]]


def allreduce_buckets_sync(gflat, buckets):
    """Reference implementation of the exchange (used by the CPU/gloo tests and as the eager path)."""
    for lo, hi in buckets:
        dist.all_reduce(gflat[lo:hi], op=dist.ReduceOp.SUM)


# ----------------------------------------------------------------------------- step runner
class StepRunner:
    """Runs pretraining micro-steps of an Engine: forward, backward (+ overlapped bucketed
    all-reduce when world_size > 1), AdamW. mode "program" records the step once into a native
    launch program (C, one HIP stream per lane) and replays it with one call per piece;
    "hipgraph" captures the same launches into HIP graphs (falls back to eager launches, loudly,
    if capture is not possible; measured slower than "program" because graph branches do not
    overlap as well as real streams); "eager" is the Python loop over the C-ABI calls."""

    def __init__(self, engine, world_size=1, use_graph=True, lr=1e-4, weight_decay=0.05, mode=None):
        self.eng = engine
        self.world = world_size
        self.lr = lr
        self.wd = weight_decay
        self.t = 0
        self.graph_mode = "eager"
        self.buckets = plan_buckets(engine.offsets, engine.n_params) if world_size > 1 else []
        self.segments = split_bwd_segments(engine.bwd_ops) if world_size > 1 else [engine.bwd_ops]
        self.comm_stream = torch.cuda.Stream(device=engine.device) if world_size > 1 else None
        self.loss_buf = torch.zeros(1, dtype=torch.float32, device=engine.device)
        self.graphs = None
        self.prog = None
        # mode: "program" = native launch program replayed from C on one HIP stream per lane (default on GPU),
        #       "hipgraph" = the same launches captured into HIP graphs, "eager" = Python loop over the C-ABI calls
        if mode is None:
            mode = "hipgraph" if use_graph else "eager"
        if mode == "program":
            self.prog, self.spans = engine.record_program(
                engine.step_pieces(self.segments if world_size > 1 else None, weight_decay=weight_decay))
            self.graph_mode = "program"
        elif mode == "hipgraph":
            self._capture()

    # -- program pieces (all enqueue on the current stream)
    def _fwd(self):
        self.eng.forward()

    def _bwd_head(self):
        eng = self.eng
        st = eng._stream()
        eng.gflat.zero_()
        eng.finalize_loss(st, True, 1.0)

    def _bwd_seg(self, i):
        self.eng._run(self.segments[i], self.eng._stream())

    def _opt(self, note=True):
        self.eng.launch_adamw(self.wd, note=note)

    def _capture(self):
        eng = self.eng
        # Captured graphs keep every launch on one stream: with the weight-gradient lane captured as a
        # parallel branch, replays on ROCm 7.2 were no faster than the single-stream graph and the
        # second replay's gradients differed from the eager/program result (1e-2 relative) — the
        # event-ordered branches are only used by the "program" and "eager" drivers.
        eng.single_stream = True
        try:
            side = torch.cuda.Stream(device=eng.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):      # warm-up outside capture (lazy module loads, allocator)
                eng.set_hyper(0.0, 1)
                self._fwd(); self._bwd_head()
                for i in range(len(self.segments)):
                    self._bwd_seg(i)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graphs = []

            def cap(fn):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    fn()
                graphs.append(g)

            if self.world == 1:
                def whole():
                    self._fwd(); self._bwd_head(); self._bwd_seg(0); self._opt(note=False)
                cap(whole)
            else:
                cap(lambda: (self._fwd(), self._bwd_head(), self._bwd_seg(0)))
                for i in range(1, len(self.segments)):
                    cap(lambda i=i: self._bwd_seg(i))
                cap(lambda: self._opt(note=False))
            self.graphs = graphs
            self.graph_mode = "hipgraph"
            torch.cuda.synchronize()
        except Exception as e:  # pragma: no cover - depends on the runtime
            print(f"[StepRunner] HIP graph capture unavailable ({type(e).__name__}: {e}); running eagerly",
                  flush=True)
            self.graphs = None
            self.graph_mode = "eager"
            torch.cuda.synchronize()
        finally:
            eng.single_stream = False

    def _launch_allreduce(self, b, works):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.comm_stream.wait_event(ev)
        lo, hi = self.buckets[b]
        with torch.cuda.stream(self.comm_stream):
            works.append(dist.all_reduce(self.eng.gflat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def step(self):
        eng = self.eng
        self.t += 1
        eng.set_hyper(self.lr, self.t, grad_scale=1.0 / self.world)
        if self.prog is not None:
            return self._step_program()
        if self.world == 1:
            if self.graphs:
                self.graphs[0].replay()
                eng.note_optimizer_launch()
            else:
                self._fwd(); self._bwd_head(); self._bwd_seg(0); self._opt()
            return
        works = []
        nseg = len(self.segments)
        for i in range(nseg):
            if self.graphs:
                self.graphs[i].replay()
            else:
                if i == 0:
                    self._fwd(); self._bwd_head()
                self._bwd_seg(i)
            self._launch_allreduce(i, works)
        # scalar loss all-reduce for logging (engine_pretrain.py:104), no host sync
        self.loss_buf.copy_(eng.total)
        lw = dist.all_reduce(self.loss_buf, op=dist.ReduceOp.SUM, async_op=True)
        for w in works:
            w.wait()                     # current stream waits for the collectives
        lw.wait()
        if self.graphs:
            self.graphs[-1].replay()
            eng.note_optimizer_launch()
        else:
            self._opt()

    def _step_program(self):
        eng = self.eng
        if self.world == 1:
            first, _ = self.spans[0]
            last, cnt = self.spans[-1]
            eng.run_program(self.prog, (first, last + cnt - first))       # the whole step in one call
            eng.note_optimizer_launch()
            return
        works = []
        for i in range(len(self.segments)):
            eng.run_program(self.prog, self.spans[i])
            self._launch_allreduce(i, works)
        self.loss_buf.copy_(eng.total)
        lw = dist.all_reduce(self.loss_buf, op=dist.ReduceOp.SUM, async_op=True)
        for w in works:
            w.wait()
        lw.wait()
        eng.run_program(self.prog, self.spans[-1])
        eng.note_optimizer_launch()

    def mean_loss(self) -> float:
        if self.world == 1:
            return float(self.eng.total.item())
        return float(self.loss_buf.item()) / self.world
