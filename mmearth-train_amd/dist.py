"""Data-parallel step runner: one process per GPU, RCCL (torch.distributed backend "nccl")
gradient all-reduce over xGMI, overlapped with the rest of backward on a side HIP stream.

Reference behaviour reproduced (/root/reference): DistributedDataParallel gradient averaging
(main_pretrain.py:306-310), rank-local GRN statistics, per-rank mask noise, scalar loss
all-reduce for logging (engine_pretrain.py:104 -> helpers.py:393-401). Not reproduced on
purpose: DDP's 25 MB bucketing heuristics and its per-iteration host syncs.

The flat fp32 gradient buffer is cut into four contiguous buckets that become final at known
points of the backward program (parameters are laid out in state-dict order, heads regrouped last):
    bucket 0  [pred_dict ... end]       7.6 MB (atto all_mod)  ready after the head weight gradients - the FIRST thing backward does
    bucket 1  [proj ... pred_dict)      9.2 MB   proj, mask token, shared decoder block: ready after proj.wgrad
    bucket 2  [stages.2, stages.3]     11.7 MB   ready after the stage-3 and stage-2 blocks
    bucket 3  [0 ... stages.2)          1.9 MB   initial conv, stem, downsample layers, stages 0-1 (last)
Each bucket is all-reduced (SUM) on a communication stream as soon as it is final. With the native launch program (default) the
forward and the WHOLE backward are one replay call and the communication stream waits for the program's per-bucket "ready" events
(the last main-lane op of the bucket's segment and the last weight-gradient-lane op so far: mpmae_program_stream_wait), so the main
lane never stops for the weight-gradient lane at a bucket boundary; `overlap="segments"` replays one call per bucket instead (each
joins the side lane first). Averaging (1/world) is folded into the AdamW kernel's grad_scale. `allreduce_dtype=
torch.bfloat16` sends the buckets as bf16 (half the xGMI bytes; the sum then carries bf16 rounding). With HIP
graphs the forward, every backward segment and the optimizer are separate captured graphs.
"""
import ctypes
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
        # eager communicator binding only when there is someone to talk to: on ROCm 7 / torch 2.10 a ONE-rank group created with
        # device_id makes the plain (no-exchange) step 1.0 ms slower (5.38 vs 4.38 ms, tools/exchange_cost_probe.py); the N > 1
        # path - exchange on - runs at 4.55 ms per rank with it and keeps it (no lazy-init stalls at the first collective)
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
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
    """offsets: OrderedDict key -> (offset, numel) in flat order. Returns [(lo, hi)] x 4 in the
    order the buckets become ready during backward."""
    keys = list(offsets.keys())
    first_proj = next(k for k in keys if not k.startswith("encoder."))
    first_s2 = next(k for k in keys if k.startswith("encoder.stages.2."))
    first_pred = next(k for k in keys if k.startswith("pred_dict."))
    lo_proj, lo_s2, lo_pred = offsets[first_proj][0], offsets[first_s2][0], offsets[first_pred][0]
    for k in keys:
        o = offsets[k][0]
        if lo_s2 <= o < lo_proj:      # everything from stages.2 up to the first non-encoder tensor must be stages.2/.3 - or, with the dense
            # encoder (sparse=False), its final norm / classifier head (convnextv2.py:151-152), which never receive a gradient: any bucket will do
            assert k.startswith(("encoder.stages.2.", "encoder.stages.3.", "encoder.norm.", "encoder.head.")), k
        if lo_proj <= o < lo_pred:    # proj, mask token, the shared decoder block
            assert k.startswith(("proj.", "mask_token", "decoder_dict.")), k
        if o >= lo_pred:              # heads, their shared LayerNorm, the uncertainty weights
            assert k.startswith(("pred_dict.", "layer_norm_tmp.", "loss_fn.")), k
    return [(lo_pred, n_params), (lo_proj, lo_pred), (lo_s2, lo_proj), (0, lo_s2)]


def split_bwd_segments(bwd_ops):
    """Cut the backward launch list where each bucket becomes final. Returns 4 lists of ops."""
    names = [op[0] for op in bwd_ops]
    i0 = next(i for i, n in enumerate(names) if n.startswith("decoder_dict."))
    i1 = next(i for i, n in enumerate(names) if n.startswith(("encoder.stages.3.", "encoder.stages.3:")))
    i2 = next(i for i, n in enumerate(names) if n.startswith("encoder.downsample_layers.1"))
    assert 0 < i0 < i1 < i2 < len(names)
    assert not any(n.startswith(("head", "dloss")) for n in names[i0:]), "head gradients must be final before the decoder segment"
    return [bwd_ops[:i0], bwd_ops[i0:i1], bwd_ops[i1:i2], bwd_ops[i2:]]


def allreduce_buckets_sync(gflat, buckets):
    """Reference implementation of the exchange (used by the CPU/gloo tests and as the eager path)."""
    for lo, hi in buckets:
        dist.all_reduce(gflat[lo:hi], op=dist.ReduceOp.SUM)


def _check(err, what):
    if err != 0:
        raise RuntimeError(f"{what}: hipError {err}")


def _c_stream(stream):
    import ctypes
    return ctypes.c_void_p(stream.cuda_stream)


class _EventWork:
    """`.wait()` of a chain that finished on the communication stream: the current stream waits for its event."""

    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


# ----------------------------------------------------------------------------- step runner
def pick_concurrent_stream(engine, prog, tries=8):
    """A new torch stream that runs CONCURRENTLY with the current (main) stream and with the side lanes of `prog`: ROCm balances HIP
    streams over GPU_MAX_HW_QUEUES (default 4) hardware queues and serialises streams that share one - an exchange stream on the main
    lane's queue parks its bucket waits in front of the main lane's kernels (measured 7-8 ms per step instead of 4.6). Candidates are
    probed by mpmae_program_stream_overlaps (include/mpmae_hip.h); rejected ones stay alive until the choice is made, so that the
    runtime's stream-count balancing moves on to another queue."""
    main = torch.cuda.current_stream(engine.device)
    cands = []
    for _ in range(tries):
        s = torch.cuda.Stream(device=engine.device)
        cands.append(s)
        ok = engine.lib.mpmae_program_stream_overlaps(prog, ctypes.c_void_p(main.cuda_stream), ctypes.c_void_p(s.cuda_stream))
        if ok != 0:              # 1 = concurrent; < 0 = the probe itself failed: take the stream as it is
            break
    return cands[-1]


class StepRunner:
    """Runs pretraining micro-steps of an Engine: forward, backward (+ overlapped bucketed
    all-reduce when world_size > 1), AdamW. mode "program" records the step once into a native
    launch program (C, one HIP stream per lane) and replays it with one call per piece;
    "hipgraph" captures the same launches into HIP graphs (falls back to eager launches, loudly,
    if capture is not possible; measured slower than "program" because graph branches do not
    overlap as well as real streams); "eager" is the Python loop over the C-ABI calls."""

    def __init__(self, engine, world_size=1, use_graph=True, lr=1e-4, weight_decay=0.05, mode=None, update_freq=1,
                 allreduce_dtype=None, overlap="events", force_exchange=False):
        self.eng = engine
        self.world = world_size
        # the bucketed gradient exchange runs when there is someone to exchange with - or on request at world size 1 (a process
        # group of one rank: the RCCL path, its streams and the program's "bucket ready" events, exercised on a single GPU)
        self.exchange = world_size > 1 or (bool(force_exchange) and dist.is_initialized())
        self.lr = lr
        self.wd = weight_decay
        self.t = 0                       # optimizer updates done (AdamW bias-correction step)
        self.micro = 0                   # micro-steps done inside the current accumulation window
        # gradient accumulation (main_pretrain.py --update_freq, engine_pretrain.py:87-94): every micro-step
        # back-propagates loss / update_freq into the SAME flat gradient buffer; the buffer is zeroed at the
        # start of a window and all-reduced + applied at its end. (The reference's DDP all-reduces on every
        # micro-step; the sum is the same, the traffic here is 1 / update_freq of it.)
        self.update_freq = int(update_freq)
        if self.update_freq > 1 and (mode == "hipgraph" or (mode is None and use_graph)):
            raise ValueError("update_freq > 1 needs mode 'program' or 'eager'")
        self.graph_mode = "eager"
        self.buckets = plan_buckets(engine.offsets, engine.n_params) if self.exchange else []
        self.segments = split_bwd_segments(engine.bwd_ops) if self.exchange else [engine.bwd_ops]
        self.comm_stream = None             # created behind the program (below): it must not share a hardware queue with a lane
        self.measure_comm_tail = False      # bench / probes: record (backward end, collectives done) event pairs on the main stream
        self.comm_tail_events = []
        self.loss_buf = torch.zeros(1, dtype=torch.float32, device=engine.device)
        # fold_loss: the scalar loss is element n_params of the flat gradient buffer (engine.gflat_ext) and is all-reduced WITH the first
        # bucket (the heads end the buffer; the loss exists before any gradient does) - one collective fewer at the tail of the step
        ext = getattr(engine, "gflat_ext", None)
        opt = getattr(engine, "opt", None)
        self.fold_loss = False
        # optional low-precision exchange: one staging buffer per bucket in the wire dtype
        if (self.exchange and ext is not None and allreduce_dtype in (None, torch.float32) and (opt is None or bool(opt.get("fold_loss", 1)))
                and self.buckets and self.buckets[0][1] == engine.n_params):
            self.fold_loss = True
            self.loss_buf = ext[engine.n_params:engine.n_params + 1]
        self.wire = ([torch.empty(hi - lo, dtype=allreduce_dtype, device=engine.device) for lo, hi in self.buckets]
                     if allreduce_dtype not in (None, torch.float32) else None)
        self.graphs = None
        self.prog = None
        # mode: "program" = native launch program replayed from C on one HIP stream per lane (default on GPU),
        #       "hipgraph" = the same launches captured into HIP graphs, "eager" = Python loop over the C-ABI calls
        if mode is None:
            mode = "hipgraph" if use_graph else "eager"
        if mode == "program":
            self.prog, self.spans = engine.record_program(
                engine.step_pieces(self.segments if self.exchange else None, weight_decay=weight_decay,
                                   loss_scale=1.0 / self.update_freq,
                                   # data parallel: the non-finite guard reads the ALL-REDUCED loss, so every rank skips (or applies) the
                                   # same update; with a rank-local guard one rank skipped while the others applied NaN-poisoned gradients
                                   guard_loss=self.loss_buf if self.exchange else None))
            self.graph_mode = "program"
            # "inputs free" point of the step (Engine.set_inputs_async): exported so that a stream outside the program can wait for it
            self.inputs_free_signal = None
            key = getattr(engine, "_inputs_free_key", None)
            if key and key in engine._program_ids and engine.device.type == "cuda":
                self.inputs_free_signal = engine._program_ids[key]
                _check(engine.lib.mpmae_program_export_signal(self.prog, self.inputs_free_signal), "program_export_signal")
            # overlap "events" (default): the whole backward is ONE replay call and the communication stream waits for the
            # per-bucket "ready" events of the program; "segments": one replay call per bucket, each joining the side lane first
            self.bucket_signals = []
            if self.exchange and overlap == "events" and engine.device.type == "cuda":
                for keys in engine._bucket_keys:
                    sig = [engine._program_ids[k] for k in keys]
                    for i_ in sig:
                        _check(engine.lib.mpmae_program_export_signal(self.prog, i_), "program_export_signal")
                    self.bucket_signals.append(sig)
        elif mode == "hipgraph":
            self._capture()
        if self.exchange and engine.device.type == "cuda" and self.comm_stream is None:
            self.comm_stream = pick_concurrent_stream(engine, self.prog)


    # -- program pieces (all enqueue on the current stream)
    def _fwd(self):
        self.eng.forward()

    def _bwd_head(self, zero=True):
        eng = self.eng
        st = eng._stream()
        if zero:
            eng.gflat.zero_()
        eng.finalize_loss(st, True, 1.0 / self.update_freq)

    def _bwd_seg(self, i):
        self.eng._run(self.segments[i], self.eng._stream())

    def _opt(self, note=True):
        self.eng.launch_adamw(self.wd, note=note, guard_loss=self.loss_buf if self.exchange else None)

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

            if not self.exchange:
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

    def _launch_allreduce(self, b, works, ready=False):
        """ready=True: the communication stream already waits for the bucket (program events); otherwise it waits for the
        current stream's position."""
        lo, hi = self.buckets[b]
        grad = self.eng.gflat[lo:hi]
        fold = self.fold_loss and b == 0
        if fold:                             # the heads' bucket + the loss slot behind it
            grad = self.eng.gflat_ext[lo:hi + 1]
        if self.comm_stream is None:         # host tensors (gloo logic tests): the collective is synchronous
            if fold:
                self.loss_buf.copy_(self.eng.total.reshape(1))
            if self.wire is not None:
                self.wire[b].copy_(self.eng.gflat[lo:hi])
                dist.all_reduce(self.wire[b], op=dist.ReduceOp.SUM)
                self.eng.gflat[lo:hi].copy_(self.wire[b])
            else:
                dist.all_reduce(grad, op=dist.ReduceOp.SUM)
            return
        if not ready:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)
        with torch.cuda.stream(self.comm_stream):
            if self.wire is not None:        # cast -> all-reduce -> cast back, all on the communication stream
                self.wire[b].copy_(self.eng.gflat[lo:hi])
                dist.all_reduce(self.wire[b], op=dist.ReduceOp.SUM)
                self.eng.gflat[lo:hi].copy_(self.wire[b])
                ev2 = torch.cuda.Event()
                ev2.record(self.comm_stream)
                works.append(_EventWork(ev2))
            else:
                if fold:
                    self.loss_buf.copy_(self.eng.total.reshape(1))
                works.append(dist.all_reduce(grad, op=dist.ReduceOp.SUM, async_op=True))

    def step(self):
        """One micro-step. Returns True when it ended with an optimizer update."""
        eng = self.eng
        first = self.micro == 0                       # zero the gradient buffer
        last = self.micro == self.update_freq - 1     # exchange + AdamW
        self.micro = 0 if last else self.micro + 1
        if hasattr(eng, "wait_inputs"):
            eng.wait_inputs()                         # a pending asynchronous input stage (Engine.set_inputs_async)
        if last:
            self.t += 1
            eng.set_hyper(self.lr, self.t, grad_scale=1.0 / self.world)
        if self.prog is not None:
            self._step_program(first, last)
        else:
            self._step_launches(first, last)
        return last

    def _step_launches(self, first, last):
        eng = self.eng
        if self.graphs and not self.exchange:
            self.graphs[0].replay()
            eng.note_optimizer_launch()
            return
        works = []
        nseg = len(self.segments)
        for i in range(nseg):
            if self.graphs:
                self.graphs[i].replay()
            else:
                if i == 0:
                    self._fwd(); self._bwd_head(zero=first)
                self._bwd_seg(i)
            if last and self.exchange:
                self._launch_allreduce(i, works)
        if not last:
            return
        if self.exchange:
            # scalar loss all-reduce for logging (engine_pretrain.py:104), no host sync
            if not self.fold_loss:
                self.loss_buf.copy_(eng.total)
                works.append(dist.all_reduce(self.loss_buf, op=dist.ReduceOp.SUM, async_op=True))
            tail = self.measure_comm_tail and eng.device.type == "cuda"
            if tail:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(torch.cuda.current_stream())
            for w in works:
                w.wait()                     # current stream waits for the collectives
            if tail:
                e1.record(torch.cuda.current_stream())
                self.comm_tail_events.append((e0, e1))
        if self.graphs:
            self.graphs[-1].replay()
            eng.note_optimizer_launch()
        else:
            self._opt()

    def _span(self, i, j):
        """Program range covering pieces i..j (inclusive): [fwd, zero, seg0, seg1, ..., opt]."""
        lo = self.spans[i][0]
        return (lo, self.spans[j][0] + self.spans[j][1] - lo)

    def _step_program(self, first, last):
        eng = self.eng
        nseg = len(self.segments)
        FWD, ZERO, SEG0, OPT = 0, 1, 2, 2 + nseg
        if not self.exchange or not last:
            hi = OPT if last else OPT - 1
            if first:
                eng.run_program(self.prog, self._span(FWD, hi))            # the whole micro-step in one call
            else:
                eng.run_program(self.prog, self._span(FWD, FWD))
                eng.run_program(self.prog, self._span(SEG0, hi))
            if last:
                eng.note_optimizer_launch()
            return
        works = []
        if self.bucket_signals:
            # ONE replay call for forward + the whole backward (no side-lane join between buckets: the main lane never waits for
            # the weight-gradient lane at a bucket boundary); the communication stream waits for each bucket's "ready" events
            if first:
                eng.run_program(self.prog, self._span(FWD, SEG0 + nseg - 1))
            else:
                eng.run_program(self.prog, self._span(FWD, FWD))
                eng.run_program(self.prog, self._span(SEG0, SEG0 + nseg - 1))
            comm = _c_stream(self.comm_stream)
            for i in range(nseg):
                for sig in self.bucket_signals[i]:
                    _check(eng.lib.mpmae_program_stream_wait(self.prog, sig, comm), "program_stream_wait")
                self._launch_allreduce(i, works, ready=True)
        else:
            eng.run_program(self.prog, self._span(FWD, ZERO if first else FWD))
            for i in range(nseg):
                eng.run_program(self.prog, self._span(SEG0 + i, SEG0 + i))
                self._launch_allreduce(i, works)
        if not self.fold_loss:
            self.loss_buf.copy_(eng.total)
            works.append(dist.all_reduce(self.loss_buf, op=dist.ReduceOp.SUM, async_op=True))
        tail = getattr(self, "measure_comm_tail", False) and eng.device.type == "cuda"
        if tail:            # exposed communication = how long the main stream sits between the end of the backward and the last collective
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
        for w in works:
            w.wait()
        if tail:
            e1.record(torch.cuda.current_stream())
            self.comm_tail_events.append((e0, e1))
        eng.run_program(self.prog, self._span(OPT, OPT))
        eng.note_optimizer_launch()

    # -- checkpoint state of the fused optimizer (helpers.save_model / auto_load_model: the "optimizer" entry)
    def state_dict(self):
        eng = self.eng
        return {"fused_adamw": True, "step": self.t, "exp_avg": eng.mflat.detach().cpu().clone(),
                "exp_avg_sq": eng.vflat.detach().cpu().clone(), "param_order": list(eng.offsets.keys()),
                "lr": self.lr, "weight_decay": self.wd, "betas": (0.9, 0.95)}

    def load_state_dict(self, sd):
        eng = self.eng
        if not sd.get("fused_adamw") or list(sd["param_order"]) != list(eng.offsets.keys()):
            raise ValueError("optimizer state does not belong to this model's fused AdamW")
        eng.mflat.copy_(sd["exp_avg"])
        eng.vflat.copy_(sd["exp_avg_sq"])
        self.t = int(sd["step"])
        self.micro = 0

    def comm_tail_ms(self):
        """Mean exposed communication per update step (ms) over the recorded event pairs (measure_comm_tail); host sync. The pairs sit on
        the main stream: e0 behind the last backward kernel, e1 behind the wait for the last bucket's all-reduce."""
        if not self.comm_tail_events:
            return None
        torch.cuda.synchronize()
        v = [a.elapsed_time(b) for a, b in self.comm_tail_events]
        self.comm_tail_events = []
        return sum(v) / len(v)

    def skipped_steps(self) -> int:
        """Updates skipped because the loss was non-finite (device-side guard in mpmae_hp_fetch); host sync."""
        return int(self.eng.hp[5].item())

    def barrier_timeouts(self) -> int:
        """Steps in which a persistent stage kernel's grid barrier timed out (a workgroup never became resident, e.g. another process
        holds CUs): the update was skipped on the device (counted in skipped_steps() too); the activations of that step are invalid."""
        return int(self.eng.hp[6].item())

    def mean_loss(self) -> float:
        if not self.exchange:
            return float(self.eng.total.item())
        return float(self.loss_buf.item()) / self.world
