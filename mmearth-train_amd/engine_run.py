"""Engine, part 6 of 6: execution - eager runs, input staging, forward segments, device meters, the step as program pieces, native launch programs, optimizer launches, results in the reference's shapes."""
import contextlib
import os
import sys
import ctypes as C
import math
from collections import OrderedDict

import torch

from . import _lib
from ._lib import EPI, PRO
from .config import ModelCfg
from .synth import dense_aliases, flat_param_spec, param_view, state_dict_spec
from .engine_common import *  # noqa: F401,F403
from .engine_common import _p, _rup, _ParamDict, _lib  # noqa: F401


class RunMixin:
    # ------------------------------------------------------------------ execution
    def nondefault_options(self):
        """Every switch of this engine's step that is not at its measured-best default: {"engine": {...}, "library": {...}}, both empty on a
        clean run. bench.py prints it in the JSON line (config.options), so that a stray MPMAE_ENGINE_OPTS on a box leaves a trace."""
        eng = {k: v for k, v in self.opt.items() if ENGINE_OPTIONS.get(k) != v}
        if self.opt["det"] and eng.get("ps") == 0:      # (implied by det = 1, not a switch of its own)
            eng.pop("ps")
        lib = _lib.nondefault_options()
        if self.opt["det"] and lib.get("DET") == 1:
            lib.pop("DET")
        return dict(engine=eng, library=lib)

    def _tail_main(self):
        v = int(self.opt["tail_main"])
        if v >= 0:
            return v
        return 0 if any(b.get("wgf") for b in self.blocks if b["stage"] == 0) else 1

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @contextlib.contextmanager
    def _det_scope(self):
        """DET is a process-wide library switch read when a launch is ISSUED or RECORDED: every eager run and every program recording of this
        engine sets it from the engine's own `det` option and puts the previous value back (a det = 0 engine built after a det = 1
        engine must not flip the first one's later eager launches to unordered folds). A developer override MPMAE_ENGINE_OPTS="DET=..." wins."""
        i, want = _lib.OPT["DET"], (1 if self.opt["det"] else 0)
        old = int(self.lib.mpmae_get_option(i))
        if self._det_env or old == want:
            yield
            return
        _lib.check(self.lib.mpmae_set_option(i, want), "set_option DET")
        try:
            yield
        finally:
            self.lib.mpmae_set_option(i, old)

    def _run(self, ops, stream=None):
        with self._det_scope():
            return self._run_ops(ops, stream)

    def _run_ops(self, ops, stream=None):
        """Enqueue a launch program. Lane-1 ops (weight gradients) go to a side HIP stream forked
        from the current stream and ordered by events; the side stream is joined at the end, so a
        program is self-contained (and capturable into one HIP graph with parallel branches)."""
        nl = (max(m["lane"] for _, _, _, m in ops) + 1) if (self.concurrent and ops and not self.single_stream) else 1
        main = torch.cuda.current_stream(self.device) if self.device.type == "cuda" else None
        if nl == 1:
            st = C.c_void_p(main.cuda_stream) if main is not None else stream
            for name, fn, args, _ in ops:
                err = fn(*args, st)
                if err != 0:
                    raise _lib.HipLibraryError(f"{name}: hipError {err}")
            return
        if not hasattr(self, "_side_streams"):
            self._side_streams = []
        while len(self._side_streams) < nl - 1:
            from . import dist as _mdist      # (a stream that does not share the main stream's hardware queue)
            self._side_streams.append(_mdist.pick_concurrent_stream(self, None))
        streams = [main] + self._side_streams[:nl - 1]
        for st in streams[1:]:
            st.wait_stream(main)                   # fork
        handles = [C.c_void_p(st.cuda_stream) for st in streams]
        events = {}
        for name, fn, args, m in ops:
            lane = m["lane"]
            for key in m["wait"]:
                ev = events.get(key)
                if ev is not None:                 # recorded earlier in THIS program (else: already joined)
                    streams[lane].wait_event(ev)
            err = fn(*args, handles[lane])
            if err != 0:
                raise _lib.HipLibraryError(f"{name}: hipError {err}")
            if m["signal"] is not None:
                ev = torch.cuda.Event()
                ev.record(streams[lane])
                events[m["signal"]] = ev
        for st in streams[1:]:
            main.wait_stream(st)                   # join

    def set_inputs(self, imgs_dict, noise, crop=None, raw=None):
        """Copy a batch and the mask noise into the engine's static device buffers (on the current stream). crop = (ty, tx): int32
        device tensors [N] of per-sample window origins - the pixel-wise modalities (larger tiles than img_size, resident on the
        device) are cut at the same window by mpmae_crop straight into the static buffers (fcmae.py:419-434).
        raw: optional dict modality -> preparation of a RAW tile fused into the same pass (mmearth_dataset.py:100-142):
        dict(mean=, std=, nodata=) for a continuous modality stored as fp32 / uint16 / uint8 (-> no-data to NaN, z-score, fp32), or
        dict(lut=int32[256]) for a class map stored as uint8 (-> remapped int64 labels, -1 = no data). noise=None: drawn on the device."""
        S = self.cfg.img_size
        st = self._stream()
        raw = raw or {}
        for k, dst in self.inp.items():
            src = imgs_dict[k]
            ty, tx = (crop if (crop is not None and src.dim() == 4 and src.shape[-1] != S) else (None, None))
            if k in raw:
                src, r = src.contiguous(), raw[k]
                assert src.device == dst.device and src.dim() == 4 and src.shape[:2] == dst.shape[:2] and (ty is not None or src.shape[-1] == S), k
                if "lut" in r:
                    assert src.dtype == torch.uint8 and dst.dtype == torch.int64 and r["lut"].dtype == torch.int32 and r["lut"].numel() == 256
                    _lib.check(self.lib.mpmae_crop_lut(_p(src), _p(dst), src.shape[0], src.shape[-1], S, _p(ty), _p(tx), _p(r["lut"]), st), "crop_lut")
                else:
                    code = {torch.float32: 0, torch.uint16: 1, torch.uint8: 2}[src.dtype]
                    _lib.check(self.lib.mpmae_crop_norm(_p(src), code, _p(dst), src.shape[0], src.shape[1], src.shape[-1], S, _p(ty), _p(tx),
                                                        _p(r["mean"]), _p(r["std"]), float(r.get("nodata", float("nan"))), st), "crop_norm")
            elif ty is not None:
                src = src.contiguous()
                assert src.device == dst.device and src.dtype == dst.dtype and src.shape[:2] == dst.shape[:2], k
                _lib.check(self.lib.mpmae_crop(_p(src), _p(dst), src.element_size(), src.shape[0], src.shape[1], src.shape[-1], S,
                                               _p(ty), _p(tx), st), "crop")
            else:
                dst.copy_(src.reshape(dst.shape), non_blocking=True)
        if noise is None:
            self.noise.normal_()
        else:
            self.noise.copy_(noise, non_blocking=True)
        if self.device.type == "cuda":
            cur = torch.cuda.current_stream(self.device)
            if cur != getattr(self, "_in_stream", None):      # an in-order stage on the caller's stream: a later asynchronous stage must not overtake it
                pe = torch.cuda.Event()
                pe.record(cur)
                self._pre_step_ev = pe

    def input_stage(self, runner=None):
        """Context manager: everything enqueued inside runs on the engine's INPUT STREAM, ordered behind the running step's last reader
        of the static input buffers (the loss-gradient launch at the head of the backward: the program's exported "inputs free" event)
        - so host-to-device copies, crop-window draws, Engine.set_inputs and the mask noise of step k+1 overlap the remaining ~2.5 ms of
        step k's backward with no second set of buffers. The next forward waits for the stage's event (wait_inputs, called by
        StepRunner.step). Work the stage depends on must be issued INSIDE the context (tensors produced on the main stream just
        before it are not ordered against the input stream)."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            if self.device.type != "cuda":
                yield
                return
            if not hasattr(self, "_in_stream"):
                # (not on a hardware queue of the main stream or of a lane: dist.pick_concurrent_stream)
                from . import dist as _mdist
                self._in_stream = _mdist.pick_concurrent_stream(self, getattr(runner, "prog", None) if runner is not None else None)
            main, ins = torch.cuda.current_stream(self.device), self._in_stream
            prog = getattr(runner, "prog", None) if runner is not None else None
            sig = getattr(runner, "inputs_free_signal", None) if runner is not None else None
            if prog is not None and sig:
                # the "inputs free" event of the most recent replay (a no-op before the first one), and never ahead of the main-stream
                # position in front of that replay (an in-order set_inputs of an older batch)
                if getattr(self, "_pre_step_ev", None) is not None:
                    ins.wait_event(self._pre_step_ev)
                _lib.check(self.lib.mpmae_program_stream_wait(prog, sig, C.c_void_p(ins.cuda_stream)), "program_stream_wait")
            else:
                ins.wait_stream(main)
            if getattr(self, "_pre_step_ev", None) is not None:
                ins.wait_event(self._pre_step_ev)
            with torch.cuda.stream(ins):
                yield
                ev = torch.cuda.Event()
                ev.record(ins)
            self._inputs_ready_ev = ev
        return ctx()

    def set_inputs_async(self, imgs_dict, noise=None, crop=None, raw=None, runner=None):
        """set_inputs inside input_stage(). Host tensors are copied to the device on the input stream; device tensors must already be
        complete (resident batches) - produce fresh ones inside `with eng.input_stage(runner):` instead."""
        with self.input_stage(runner):
            if self.device.type == "cuda":
                ins = self._in_stream
                imgs_dict = {k: (v.to(self.device, non_blocking=True) if isinstance(v, torch.Tensor) and not v.is_cuda else v)
                             for k, v in imgs_dict.items()}
                for t in list(imgs_dict.values()) + ([noise] if noise is not None else []) + (list(crop) if crop is not None else []):
                    if isinstance(t, torch.Tensor) and t.is_cuda:
                        t.record_stream(ins)
            self.set_inputs(imgs_dict, noise, crop=crop, raw=raw)

    def wait_inputs(self):
        """Called in front of a forward: the main stream waits for a pending asynchronous input stage; marks the stream position in
        front of the step for the NEXT stage."""
        if self.device.type != "cuda":
            return
        ev = getattr(self, "_inputs_ready_ev", None)
        main = torch.cuda.current_stream(self.device)
        if ev is not None:
            main.wait_event(ev)
            self._inputs_ready_ev = None
        pe = torch.cuda.Event()
        pe.record(main)
        self._pre_step_ev = pe

    # ------------------------------------------------------------------ forward segments (FCMAE.forward_encoder / _decoder / _loss)
    def _segment_bounds(self):
        names = [op[0] for op in self.fwd_ops]
        i_proj = names.index("proj")
        i_loss = next(i for i, n in enumerate(names) if n.startswith("loss:"))
        return dict(encoder=(0, i_proj), decoder=(i_proj, i_loss), loss=(i_loss, len(names)))

    def run_segment(self, which: str):
        """Run one of the three pieces of the forward program: "encoder" (mask, stem, stages -> enc_out rows),
        "decoder" (proj, mask token, decoder block, heads -> predictions), "loss" (12 losses + weighting).
        The later pieces re-stage the weights first (a caller may have changed them since the last encoder run)."""
        lo, hi = self._segment_bounds()[which]
        ops = list(self.fwd_ops[lo:hi])
        if which != "encoder":
            # "prep" (weight staging) and, in fp8 mode, the MX weight quantisers - ON THE MAIN LANE here: in the full program prep runs
            # on the side lane and only the stem GEMM waits for it, so a slice that starts at `proj` would race its own re-staging
            pre = [op for op in self.fwd_ops[:lo] if op[0] == "prep" or op[0].startswith("prep:")]
            ops = [(n_, f_, a_, dict(m_, lane=0, wait=(), signal=None)) for n_, f_, a_, m_ in pre] + ops
        if which == "loss":
            self.loss_acc.zero_()
        elif which == "encoder":
            self.stats.zero_()
            if hasattr(self, "ps_sync"):
                self.ps_sync[:, 2].zero_()      # (as in Engine.forward: an eager encoder pass starts with clean grid-barrier error words)
        self._run(ops, self._stream())
        if which == "loss":
            self.finalize_loss(self._stream(), False, 1.0)

    def set_mask(self, mask):
        """Install a caller-supplied mask [N, L] (0 keep / 1 remove, `keep` zeros per row): the rank kernel is
        stable, so ranking the mask values themselves reproduces exactly this mask and its vis / inv tables."""
        self.noise.copy_(mask.reshape(self.N, self.L).to(torch.float32))
        if self.dense:
            _lib.check(self.lib.mpmae_mask_gen_dense(_p(self.noise), self.N, self.L, self.keep_mask, _p(self.mask), _p(self.inv),
                                                     self._stream()), "mask_gen_dense")
            return
        _lib.check(self.lib.mpmae_mask_gen(_p(self.noise), self.N, self.L, self.keep, _p(self.mask), _p(self.vis),
                                           _p(self.inv), self._stream()), "mask_gen")

    def set_preds(self, preds):
        """Load predictions in the reference's shapes ([N, p*p*C, h, w] / [N, K]) into the head output buffers."""
        N, L = self.N, self.L
        for om in self.cfg.out_mods:
            c, v = self.head_cols[om.name], preds[om.name]
            if om.kind.startswith("pix"):
                self.pred_pix[:, c:c + om.head_out] = v.reshape(N, om.head_out, L).permute(0, 2, 1).reshape(N * L, om.head_out)
            else:
                self.pred_img[:, c:c + om.head_out] = v.reshape(N, om.head_out)

    # ------------------------------------------------------------------ native launch programs
    def _meters(self):
        if not hasattr(self, "_meters_rec"):
            m = _lib.Meters()
            m.losses, m.T = self.losses.data_ptr(), len(self.cfg.out_mods)
            m.weighted = self.weighted.data_ptr() if self.cfg.loss_aggr == "uncertainty" else 0
            m.ring, m.window, m.sums, m.gnorm2 = self.meter_ring.data_ptr(), self.METER_WINDOW, self.meter_sums.data_ptr(), self.gnorm2.data_ptr()
            if hasattr(self, "ps_sync"):       # grid-barrier error words of the persistent stage kernels: a timeout skips the update and is counted in hp[6]
                m.err_words, m.n_err, m.err_stride = self.ps_sync.data_ptr(), int(self._ps_launches), int(self.ps_sync.shape[1])
            self._meters_rec = m
        return C.byref(self._meters_rec)

    def reset_meters(self):
        """New epoch: the reference builds a fresh MetricLogger per epoch (engine_pretrain.py:34)."""
        self.meter_ring.zero_()
        self.meter_sums.zero_()

    def meter_global_averages(self):
        """Per-epoch statistics, synchronised between the ranks with ONE all-reduce of the running sums
        (MetricLogger.synchronize_between_processes, helpers.py:66-77,134-136): dict column -> global average over all ranks' updates."""
        import torch.distributed as tdist
        sums = self.meter_sums.clone()
        T = len(self.cfg.out_mods)
        nb = int(self.gnorm2[0].item())
        sums[2 * T + 1] = sums[2 * T + 1] + torch.sqrt(self.gnorm2[1:1 + nb].sum()) * self.hp[3]      # the last update's norm has not been fetched yet
        if tdist.is_initialized() and tdist.get_world_size() > 1:
            tdist.all_reduce(sums)
        sums = sums.cpu()
        cnt = max(float(sums[-1]), 1.0)
        names = [om.name for om in self.cfg.out_mods]
        cols = [f"loss_{n}" for n in names] + [f"weighted_{n}" for n in names] + ["loss", "grad_norm"]
        return {c: float(sums[i]) / cnt for i, c in enumerate(cols)}

    def read_meters(self):
        """ONE device-to-host copy: dict name -> dict(value, median, avg (both over the last <= 20 updates), global_avg) for every
        per-modality loss, its uncertainty-weighted form, the total loss and the gradient norm (reference SmoothedValue properties)."""
        T, W = len(self.cfg.out_mods), self.METER_WINDOW
        buf = torch.cat([self.meter_ring.reshape(-1), self.meter_sums]).cpu()
        ring, sums = buf[:W * (2 * T + 2)].view(W, 2 * T + 2), buf[W * (2 * T + 2):]
        cnt = int(sums[-1].item())
        names = [om.name for om in self.cfg.out_mods]
        cols = [f"loss_{n}" for n in names] + [f"weighted_{n}" for n in names] + ["loss", "grad_norm"]
        out = {"count": cnt}
        for i, c in enumerate(cols):
            n = cnt - 1 if c == "grad_norm" else cnt          # the norm of the latest update lands with the next fetch
            if n <= 0:
                continue
            k = min(n, W)
            idx = [(n - 1 - j) % W for j in range(k)]
            win = ring[idx, i]
            out[c] = dict(value=float(win[0]), median=float(win.median()), avg=float(win.mean()), global_avg=float(sums[i]) / n)
        return out

    def step_pieces(self, bwd_segments=None, weight_decay=0.05, beta1=0.9, beta2=0.95, eps=1e-8, loss_scale=1.0, guard_loss=None):
        """The whole micro-step as op tuples, grouped into the pieces a data-parallel / gradient-accumulating
        runner issues separately: [forward + loss], [gradient zeroing], [backward segment 0], [segment 1], ...,
        [AdamW]. Consecutive pieces are contiguous in the recorded program, so any run of them is ONE
        mpmae_program_run call (the plain single-GPU step is the whole range)."""
        lib, a = self.lib, self._fin_args
        m0 = dict(lane=0, wait=(), signal=None)

        zs = self.lanes and bool(self.opt["zero_side"])
        zl = dict(lane=1, wait=(), signal=None) if zs else m0

        def fin(dlv):      # the forward finalisation also joins the image-head chain that ran on the side lane
            w = tuple(getattr(self, "_fwd_join_keys", ())) if (zs or not dlv) else ()
            m = dict(lane=0, wait=w + (("grads_zero",) if dlv and zs else ()), signal=None)
            return ("loss.finalize", lib.mpmae_loss_finalize_guarded,
                    (a[0], self.loss_slots, a[1], a[2], float(loss_scale), a[3], a[4], a[5], a[6], a[7] if dlv else None) + tuple(self._err_words()), m)

        segs = bwd_segments if bwd_segments is not None else [self.bwd_ops]
        # zero fills: with `zero_side` they run on the side lane, which is idle in the forward (the main lane's first wait for a side-lane
        # event - the stem GEMM waiting for the weight staging - covers the statistics; the gradient finalisation waits for "grads_zero")
        fwd = [("stats.zero", lib.mpmae_memset_async, (_p(self.stats), 0, self.stats.numel() * 4), dict(zl, signal="stats_zero") if zs else zl)]
        if zs:      # the first statistics producer of the main lane waits for the fill explicitly (program_run drops the wait when an earlier
            # main-lane wait for a later side-lane event already implies it; without prep_side / in fp8 mode nothing else orders them: ADVICE r3)
            for op in self.fwd_ops:
                if op[3]["lane"] == 0 and (op[0].endswith((":ln+pw1", ":pw1")) or ":ps.fwd" in op[0]):
                    if "stats_zero" not in op[3]["wait"]:
                        op[3]["wait"] = tuple(op[3]["wait"]) + ("stats_zero",)
                    break
        # ... and the forward's own finalisation is dropped: the one in front of the backward computes the same losses / total plus
        # d total / d log_vars (a caller that replays ONLY the forward piece reads its losses through Engine.forward instead)
        fwd += list(self.fwd_ops) + ([] if zs else [fin(False)])
        zero = [("grads.zero", lib.mpmae_memset_async, (_p(self.gflat), 0, self.gflat.numel() * 4),
                 dict(zl, signal="grads_zero") if zs else m0)]
        first = [fin(True)] + list(segs[0])
        # AdamW reads every gradient: when the optimizer is replayed in the SAME mpmae_program_run call as the backward (the
        # single-GPU step), the side lanes are only joined at the end of that call, so its first op waits for the last op of
        # every side lane (in-order streams: that implies all of them). Without it the update raced the last weight
        # gradients whenever the main lane got ahead (seen once the scratch rings stopped throttling it).
        last_side = {}
        for sg in segs:
            for op in sg:
                if op[3]["lane"] != 0:
                    last_side[op[3]["lane"]] = op
        joins = []
        for ln, op in sorted(last_side.items()):
            if op[3]["signal"] is None:
                self._evseq += 1
                op[3]["signal"] = f"j{self._evseq}"
            joins.append(op[3]["signal"])
        fetch = ("hp.fetch", lib.mpmae_hp_fetch, (C.c_void_p(self.hp_ring.data_ptr()), self.HP_SLOTS, _p(self.hp_counter),
                                                  _p(self.hp), _p(guard_loss if guard_loss is not None else self.total), self._meters()))
        # (the optimizer cut along the gradient buckets - a bucket's AdamW on the weight-gradient lane as soon as its gradients are final - was built in
        #  round 5, did not move the step (3.653 / 3.656 vs 3.650 / 3.647 ms, profiles/r05/ab_adamw_split.txt) and is removed)
        opt = [fetch + (dict(lane=0, wait=tuple(joins), signal=None),),
               ("adamw", lib.mpmae_adamw, (_p(self.pflat), _p(self.gflat), _p(self.mflat), _p(self.vflat), _p(self.hp),
                                           beta1, beta2, eps, weight_decay, self.n_params, _p(self.decay_mask), _p(self.gnorm2)), m0)]
        # "bucket ready" points for a data-parallel runner that replays the whole backward as ONE call: per segment the keys of its last
        # main-lane op and of the last side-lane op seen so far (in-order lanes: they imply everything before them)
        self._bucket_keys = []
        last_side_key = None
        for sg in segs:
            keys = []
            main_ops = [op for op in sg if op[3]["lane"] == 0]
            side_ops = [op for op in sg if op[3]["lane"] != 0]
            for op in ([main_ops[-1]] if main_ops else []) + ([side_ops[-1]] if side_ops else []):
                if op[3]["signal"] is None:
                    self._evseq += 1
                    op[3]["signal"] = f"b{self._evseq}"
                keys.append(op[3]["signal"])
            if side_ops:
                last_side_key = side_ops[-1][3]["signal"]
            elif last_side_key is not None:
                keys.append(last_side_key)
            self._bucket_keys.append(keys)
        return [fwd, zero, first] + [list(sg) for sg in segs[1:]] + [opt]

    def record_program(self, pieces):
        """Record op tuples into a native launch program (include/mpmae_hip.h, "launch programs").
        Returns (program handle, [(first op, op count) per piece])."""
        lib = self.lib
        prog = C.c_void_p(lib.mpmae_program_create())
        ids, spans, n = {}, [], 0
        det = self._det_scope()
        det.__enter__()
        try:
            for piece in pieces:
                spans.append((n, len(piece)))
                for name, fn, args, meta in piece:
                    waits = [ids.setdefault(k, len(ids) + 1) for k in meta.get("wait", ()) if k]
                    arr = (C.c_int * max(1, len(waits)))(*waits)
                    sig = ids.setdefault(meta["signal"], len(ids) + 1) if meta.get("signal") else 0
                    _lib.check(lib.mpmae_program_begin_op(prog, int(meta.get("lane", 0)), arr, len(waits), sig), "program_begin_op")
                    _lib.check(fn(*args, None), "record " + name)
                    n += 1
        finally:
            err = lib.mpmae_program_end(prog)
            det.__exit__(None, None, None)
        _lib.check(err, "program_end")
        assert lib.mpmae_program_num_ops(prog) == n
        self._programs = getattr(self, "_programs", []) + [prog]
        self._program_ids = ids                       # event key -> signal id of the most recently recorded program
        return prog, spans

    def run_program(self, prog, span):
        _lib.check(self.lib.mpmae_program_run(prog, span[0], span[1], self._stream()), "program_run")

    def forward(self, loss_scale: float = 1.0):
        st = self._stream()
        self.stats.zero_()
        if hasattr(self, "ps_sync"):
            # the grid-barrier error words are consumed (and cleared) by hp_fetch, i.e. by an optimizer step: a forward-only / eval caller would
            # keep reading total = +inf after ONE timeout although its own forwards completed - an eager forward starts clean
            self.ps_sync[:, 2].zero_()
        self._run(self.fwd_ops, st)
        self.finalize_loss(st, False, loss_scale)
        self._loss_scale = float(loss_scale)

    def backward(self, zero_grad: bool = True):
        st = self._stream()
        if zero_grad:
            self.gflat.zero_()
        # the GRN backward statistics are ACCUMULATED into the arena the forward zeroed: a second backward behind the same forward (retain_graph,
        # backward-only replays) must start from zero again, like the gradient buffer (the step programs zero the whole arena once per step)
        for blk in self.blocks + self.decs:
            blk["S01"].zero_()
        # d(total)/d(log_vars) and the per-modality coefficients (second finalize pass adds dlog_vars)
        self.finalize_loss(st, True, self._loss_scale)
        self._run(self.bwd_ops, st)

    def optimizer_step(self, lr: float, weight_decay: float = 0.05, beta1: float = 0.9, beta2: float = 0.95,
                       eps: float = 1e-8, grad_scale: float = 1.0):
        self.step_count += 1
        self.set_hyper(lr, self.step_count, beta1, beta2, grad_scale)
        self.launch_adamw(weight_decay, beta1, beta2, eps)

    def set_hyper(self, lr, t, beta1=0.9, beta2=0.95, grad_scale=1.0):
        """Fill the hyper-parameter record {lr, 1/(1-b1^t), 1/sqrt(1-b2^t), grad_scale} the NEXT optimizer
        launch will fetch (slot = launches so far % HP_SLOTS of the pinned ring)."""
        slot = self._hp_n % self.HP_SLOTS
        ev = self._hp_ev[slot]
        if ev is not None:                      # the launch that last used this slot must have fetched it
            ev.synchronize()
            self._hp_ev[slot] = None
        r = self.hp_ring[slot]
        r[0] = lr
        r[1] = 1.0 / (1.0 - beta1 ** t)
        r[2] = 1.0 / math.sqrt(1.0 - beta2 ** t)
        r[3] = grad_scale

    def note_optimizer_launch(self):
        """Call after enqueueing one optimizer launch (hp_fetch + AdamW), however it was issued."""
        if self.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._hp_ev[self._hp_n % self.HP_SLOTS] = ev
        self._hp_n += 1

    def launch_adamw(self, weight_decay=0.05, beta1=0.9, beta2=0.95, eps=1e-8, note=True, guard_loss=None):
        """guard_loss: the device scalar whose non-finiteness skips the update (default: this rank's loss; a data-parallel runner
        passes the all-reduced loss so that every rank takes the same decision)."""
        st = self._stream()
        _lib.check(self.lib.mpmae_hp_fetch(C.c_void_p(self.hp_ring.data_ptr()), self.HP_SLOTS, _p(self.hp_counter),
                                           _p(self.hp), _p(guard_loss if guard_loss is not None else self.total), self._meters(), st), "hp_fetch")
        err = self.lib.mpmae_adamw(_p(self.pflat), _p(self.gflat), _p(self.mflat), _p(self.vflat), _p(self.hp),
                                   beta1, beta2, eps, weight_decay, self.n_params, _p(self.decay_mask), _p(self.gnorm2), st)
        _lib.check(err, "adamw")
        if note:
            self.note_optimizer_launch()

    def grad_norm(self):
        """Global L2 norm of the flat gradient buffer NOW (a pass of its own: mpmae_sumsq). The training loop does not call this - the
        norm of every update rides in the AdamW launch and is read through read_meters()["grad_norm"]."""
        t = torch.zeros(1, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.mpmae_sumsq(_p(self.gflat), self.n_params, _p(t), self._stream()), "sumsq")
        return t.sqrt()

    # ------------------------------------------------------------------ results (reference shapes)
    def preds(self):
        """dict modality -> prediction in the reference's shapes ([N, p*p*C, h, w] / [N, K])."""
        N, L, g = self.N, self.L, self.grid
        out = OrderedDict()
        for om in self.cfg.out_mods:
            c = self.head_cols[om.name]
            if om.kind.startswith("pix"):
                v = self.pred_pix[:, c:c + om.head_out].reshape(N, L, om.head_out)
                out[om.name] = v.permute(0, 2, 1).reshape(N, om.head_out, g, g)
            else:
                out[om.name] = self.pred_img[:, c:c + om.head_out]
        return out

    def dense_map(self, rows, Cc, stage):
        """Scatter compacted stage rows [M, C] to the reference's dense [N, C, G, G] map (tests)."""
        N, keep, S, g = self.N, self.keep, self.S[stage], self.grid
        x = rows.float().reshape(N, keep, S, S, Cc)
        out = torch.zeros(N, g, S, g, S, Cc, device=rows.device)
        vis = self.vis.view(N, keep).long()
        py, px = vis // g, vis % g
        n_idx = torch.arange(N, device=rows.device)[:, None].expand(N, keep)
        out[n_idx, py, :, px, :, :] = x.permute(0, 1, 2, 3, 4)
        return out.reshape(N, g * S, g * S, Cc).permute(0, 3, 1, 2)
