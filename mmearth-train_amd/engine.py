"""MP-MAE pretraining step on MI355X: forward + losses + backward + AdamW as an explicit
launch sequence of libmpmae_hip.so kernels (no autograd, no MinkowskiEngine, no CPU fallback).

The engine owns (through torch, which is only the allocator / stream / collective provider):
  * one flat fp32 parameter buffer, one flat fp32 gradient buffer, AdamW moments;
  * a compute-type (fp32 or bf16) arena of staged weights in the [N][K] layouts the GEMMs read;
  * all activation workspaces of the step for a fixed per-GPU batch size N;
  * pre-built ctypes argument records, so a step is a flat loop of C-ABI calls that can be
    captured into a HIP graph.

Reference call path reproduced (paths relative to /root/reference):
  FCMAE.forward            models/fcmae.py:414-456
  SparseConvNeXtV2.forward models/convnextv2_sparse.py:191-220 (+ Block :47-56)
  forward_decoder          models/fcmae.py:249-265 (shared decoder Block evaluated once)
  forward_loss             models/fcmae.py:267-412, custom_loss.py:19-30
  backward / optimizer     engine_pretrain.py:87-94, main_pretrain.py:312-320
"""
# The engine is one class cut into six modules (round 6: engine.py was a 2 400-line monolith): this file holds construction - parameters, buffers, weight
# staging - and the mixins hold the op helpers (engine_ops), the block programs (engine_blocks), the forward and backward program builders
# (engine_forward, engine_backward) and execution (engine_run). ENGINE_OPTIONS lives in engine_common and is re-exported here.
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
from .engine_ops import OpsMixin
from .engine_blocks import BlocksMixin
from .engine_forward import ForwardMixin
from .engine_backward import BackwardMixin
from .engine_run import RunMixin


class Engine(OpsMixin, BlocksMixin, ForwardMixin, BackwardMixin, RunMixin):
    def __init__(self, cfg: ModelCfg, batch_size: int, dtype: str = "bf16", device="cuda",
                 track_activity: bool = True, mask_ratio=None, block_mode=None, param_buffers=None, lanes=None, options=None):
        self.lib = _lib.load()
        self.opt = dict(ENGINE_OPTIONS)
        for kv in filter(None, os.environ.get("MPMAE_ENGINE_OPTS", "").split(",")):
            k, v = kv.split("=")
            if k.strip() in _lib.OPT:           # upper-case names are library options (process-wide, mpmae_set_option)
                _lib.check(self.lib.mpmae_set_option(_lib.OPT[k.strip()], int(v)), "set_option " + k)
                if k.strip() == "RSC_PF":      # the engine plans around this one (grn_fold needs the staged GRN prologue)
                    self.opt[k.strip().lower()] = int(v)
            else:
                self.opt[k.strip()] = int(v)
        self.opt.update(options or {})
        unknown = set(self.opt) - set(ENGINE_OPTIONS)
        if unknown:
            raise KeyError(f"unknown engine options {sorted(unknown)}")
        if self.opt["det"]:          # reproducible forward: ordered folds in the library (process-wide switch, read when the program is built), no persistent stage kernel
            self.opt["ps"] = 0
        # DET is a process-wide library switch read while THIS engine's launches are built / recorded: set it from this engine's option
        # every time (an earlier det = 1 engine of the process must not leave later engines on the slower ordered folds);
        # a developer override MPMAE_ENGINE_OPTS="DET=..." (A/B of the shared-row behaviour, -1) is kept
        self._det_env = "DET=" in os.environ.get("MPMAE_ENGINE_OPTS", "")
        if not self._det_env:
            _lib.check(self.lib.mpmae_set_option(_lib.OPT["DET"], 1 if self.opt["det"] else 0), "set_option DET")
        self.cfg = cfg
        self.N = N = int(batch_size)
        # "fp8": the bf16 program with the decoder block's pointwise layers (K % 128 == 0) on the MX-fp8 MFMA path: e4m3
        # operands + E8M0 block scales for pwconv1 / pwconv2 forward and data gradient; weight gradients, statistics,
        # normalisations, losses and the optimizer stay as in bf16 mode (BASELINE configs[4])
        self.fp8 = dtype in ("fp8", "mxfp8")
        self.dt = {"f32": F32, "fp32": F32, "float32": F32, "bf16": BF16, "bfloat16": BF16, "fp8": BF16, "mxfp8": BF16}[dtype]
        self.tdtype = torch.float32 if self.dt == F32 else torch.bfloat16
        self.device = torch.device(device)
        # FCMAE(sparse=False) (fcmae.py:103-111): the dense ConvNeXtV2 encoder computes EVERY patch (rows = all N * L patches in patch order,
        # geometry tables NULL) and the mask only zeroes input pixels, places the mask token and selects the loss patches
        self.dense = not getattr(cfg, "sparse", True)
        self.track_activity = track_activity and not self.dense
        self.block_mode_override = block_mode      # None (policy) | "fused" | "mat"
        self.disable_rs = False                    # tests: force the unfused kernels for the small-C stages
        self.disable_rsc = (not self.opt["rsc"])
        self.down_grouped = bool(self.opt["down_grouped"])
        self.rsc_small = bool(self.opt["rsc_small"])     # fused GRN prologues at C = 40 / 80 too
        # GRN finalisation recomputed in the prologue of the fused kernels (no separate launches on the main lane)
        self.grn_fold = bool(self.opt["grn_fold"]) and bool(self.opt["rsc_pf"])
        self.dz_recompute = bool(self.opt["dzr"])
        # weight gradients on a side HIP stream (lanes=False / MPMAE_LANES=0: single in-order stream)
        self.concurrent = ((self.device.type == "cuda") and bool(self.opt["lanes"])
                           and (lanes is None or bool(lanes)))
        self.single_stream = False                 # set while capturing HIP graphs (see dist.StepRunner._capture)
        self.lanes = self.concurrent and (block_mode or ("mat" if self.dt == BF16 else "fused")) == "mat"
        self._side_readers = {}
        self._evseq = 0
        self._ext_buffers = param_buffers          # optional (pflat, gflat) owned by the caller (FCMAE module)
        self.L = cfg.num_patches
        self.grid = cfg.grid
        self.keep_mask = cfg.len_keep(mask_ratio)                    # patches the mask keeps per sample
        self.keep = self.L if self.dense else self.keep_mask         # patches per sample the encoder has rows for
        self.p = cfg.patch_size
        self.S = [8, 4, 2, 1]
        self.M = [N * self.keep * s * s for s in self.S]
        self.Mfull = N * self.keep * self.p * self.p
        self.D = cfg.decoder_embed_dim
        self._keepalive = []
        self.mx, self._mx_wq = {}, {}
        self._build_params()
        self._alloc()
        self._build_prep()
        self.fwd_ops, self.bwd_ops = [], []
        self._build_forward()
        self._build_backward()
        if self._mx_wq:            # MX copies of the staged weights: right behind the weight staging, on ITS lane (round 6: the side lane - eight 6.5 us
            # launches sat on the main lane in front of the forward); every MX GEMM waits for the last of them ("wq_done": in-order lane)
            wq = []
            for wname, (w, buf) in self._mx_wq.items():
                self._quant(wq, "prep:" + wname + ".quant", w["t"], w["ld"], buf)
            at = next(i for i, op in enumerate(self.fwd_ops) if op[0] == "prep")
            lane = self.fwd_ops[at][3]["lane"]
            for op in wq:
                op[3]["lane"] = lane
            if lane:
                wq[-1][3]["signal"] = "wq_done"
                for op in self.fwd_ops + self.bwd_ops:
                    if op[3]["kind"].startswith("gemm_mx"):
                        op[3]["wait"] = tuple(op[3]["wait"]) + ("wq_done",)
            self.fwd_ops[at + 1:at + 1] = wq
        self.step_count = 0

    # ------------------------------------------------------------------ params
    def _build_params(self):
        spec = flat_param_spec(self.cfg)      # state-dict order with the prediction heads regrouped (see synth.py)
        total = sum(math.prod(s) for _, s, _ in spec)
        dev = self.device
        # gflat_ext: the flat gradient buffer + 4 floats; element `total` is the LOSS SLOT of the data-parallel exchange (dist.StepRunner
        # all-reduces the scalar loss as one more element of the first gradient bucket - the heads, which end the flat buffer - instead
        # of a collective of its own: 70 us per step at the tail)
        if self._ext_buffers is not None:
            self.pflat, self.gflat = self._ext_buffers
            assert self.pflat.numel() == total and self.gflat.numel() == total
            assert self.pflat.device == dev and self.pflat.dtype == torch.float32
            st = self.gflat.untyped_storage()
            self.gflat_ext = None
            if self.gflat.is_contiguous() and st.nbytes() >= (self.gflat.storage_offset() + total + 1) * 4:      # the caller left room behind it
                self.gflat_ext = torch.as_strided(self.gflat, (total + 1,), (1,))
        else:
            self.pflat = torch.zeros(total, dtype=torch.float32, device=dev)
            self.gflat_ext = torch.zeros(total + 4, dtype=torch.float32, device=dev)
            self.gflat = self.gflat_ext[:total]
        self.mflat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.vflat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.decay_mask = torch.zeros(total, dtype=torch.uint8, device=dev)
        self.params, self.grads, self.offsets = _ParamDict(), _ParamDict(), OrderedDict()
        off = 0
        for key, shape, _ in spec:
            n = math.prod(shape)
            self.params[key] = param_view(self.pflat[off:off + n], key, shape)
            self.grads[key] = param_view(self.gflat[off:off + n], key, shape)
            self.offsets[key] = (off, n)
            # timm param_groups_weight_decay: ndim <= 1 or name endswith ".bias" -> no decay. The dense encoder's classifier head
            # (convnextv2.py:151-152) is in the state dict but not in the pretraining graph: its gradient is None in the reference, so
            # the optimizer skips it - here its gradient stays zero and it must not decay either
            if len(shape) > 1 and not key.endswith(".bias") and not key.startswith("encoder.head."):
                self.decay_mask[off:off + n] = 1
            off += n
        if self.dense:
            for akey, key, ashape in dense_aliases(self.cfg):
                o, n = self.offsets[key]
                self.params.alias[akey] = self.pflat[o:o + n].view(ashape)
                self.grads.alias[akey] = self.gflat[o:o + n].view(ashape)
        self.n_params = total
        self.hp = torch.zeros(8, dtype=torch.float32, device=dev)   # {lr, 1/bc1, 1/sqrt(bc2), grad_scale, skip, n_skipped, -, -}
        # hyper-parameter ring: slot t % HP_SLOTS is written by set_hyper for optimizer launch t and read on the
        # device by mpmae_hp_fetch (a replayed step must not read a record the host has already overwritten)
        self.HP_SLOTS = 16
        self.hp_ring = (torch.zeros(self.HP_SLOTS, 4, dtype=torch.float32).pin_memory() if dev.type == "cuda"
                        else torch.zeros(self.HP_SLOTS, 4))
        self.hp_counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self._hp_n = 0                              # optimizer launches enqueued so far
        self._hp_ev = [None] * self.HP_SLOTS
        self.gnorm2 = torch.zeros(1 + 4096, dtype=torch.float32, device=dev)      # {count, per-workgroup sums of g^2} written by mpmae_adamw (<= 4096 workgroups)
        # device-resident meters (reference MetricLogger / SmoothedValue(window_size=20), helpers.py:49-206): written by mpmae_hp_fetch
        T = len(self.cfg.out_mods)
        self.METER_WINDOW = 20
        self.meter_ring = torch.zeros(self.METER_WINDOW, 2 * T + 2, dtype=torch.float32, device=dev)
        self.meter_sums = torch.zeros(2 * T + 3, dtype=torch.float32, device=dev)      # running sums [2T + 2] + count

    def load_state_dict(self, sd):
        """sd: reference-layout state dict (aliases of the shared decoder block are accepted)."""
        first = self.cfg.out_mods[0].name
        for key, t in self.params.items():
            src = sd[key] if key in sd else None
            if src is None:
                raise KeyError(key)
            t.copy_(src.to(torch.float32).reshape(t.shape))

    def state_dict(self):
        """Reference-layout state dict (shared decoder block replicated under every modality)."""
        first = self.cfg.out_mods[0].name
        pre = f"decoder_dict.{first}."
        out = OrderedDict()
        for k, v in self.params.items():
            if not k.startswith(pre):
                out[k] = v
        for m in self.cfg.out_mods:
            for k, v in self.params.items():
                if k.startswith(pre):
                    out[f"decoder_dict.{m.name}." + k[len(pre):]] = v
        return out

    # ------------------------------------------------------------------ buffers
    def _t(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or self.tdtype, device=self.device)

    def _alloc(self):
        cfg, N, L, D = self.cfg, self.N, self.L, self.D
        dims, dev = cfg.dims, self.device
        f32 = torch.float32
        S = cfg.img_size
        # static inputs
        self.inp = OrderedDict()
        self.inp["sentinel2"] = torch.zeros(N, cfg.in_chans, S, S, dtype=f32, device=dev)
        for om in cfg.out_mods:
            if om.name == "sentinel2":
                continue
            if om.kind == "pix_cont":
                self.inp[om.name] = torch.zeros(N, om.chans, S, S, dtype=f32, device=dev)
            elif om.kind == "pix_cat":
                self.inp[om.name] = torch.zeros(N, 1, S, S, dtype=torch.int64, device=dev)
            elif om.kind == "img_cat":
                self.inp[om.name] = torch.zeros(N, om.chans, dtype=torch.int64, device=dev)
            else:
                self.inp[om.name] = torch.zeros(N, om.chans, dtype=f32, device=dev)
        self.noise = torch.zeros(N, L, dtype=f32, device=dev)
        self.mask = torch.zeros(N, L, dtype=f32, device=dev)
        self.vis = torch.zeros(N * self.keep, dtype=torch.int32, device=dev)
        self.inv = torch.zeros(N * L, dtype=torch.int32, device=dev)
        if self.dense:      # every patch is a row, in patch order (the stem's gather walks `vis`; `inv` comes from mpmae_mask_gen_dense)
            self.vis.copy_(torch.arange(L, dtype=torch.int32, device=dev).repeat(N))
        # activity maps
        if self.dense and getattr(cfg, "use_orig_stem", False):
            self.act_full, self.act = None, [None] * 4      # Conv2d k = s = patch / 8 without padding: every point exists (convnextv2.py:97-106)
        elif self.dense:
            # The dense stem's 3x3 convolution is VALID (convnextv2.py:110: S - 2 points per side) and its depthwise k x k stride-k
            # convolution pads k // 2 (:117-121): in pixel-centred terms the outermost ring of convolution outputs does not exist and
            # contributes zero to the depthwise sum. That is exactly an inactive site of the sparse stem kernels: a STATIC activity map.
            S_ = cfg.img_size
            ring = torch.ones(S_, S_, dtype=torch.uint8)
            ring[0, :] = 0; ring[-1, :] = 0; ring[:, 0] = 0; ring[:, -1] = 0
            g_, p_ = self.grid, self.p
            rows = ring.view(g_, p_, g_, p_).permute(0, 2, 1, 3).reshape(-1)          # (patch, iy, ix) = the row order within a sample
            self.act_full = rows.repeat(N).contiguous().to(dev)
            self.act = [None] * 4
        elif self.track_activity:
            self.act_full = torch.ones(self.Mfull, dtype=torch.uint8, device=dev)
            self.act = [self.act_full if self.p == 8 else torch.ones(self.M[0], dtype=torch.uint8, device=dev)]
            for i in range(1, 4):
                self.act.append(torch.ones(self.M[i], dtype=torch.uint8, device=dev))
        else:
            self.act_full = None
            self.act = [None] * 4
        C0 = dims[0]
        # stem
        self.c1 = self._t(self.Mfull, C0)
        self.c1hat = self._t(self.Mfull, C0)
        self.rstd1 = self._t(self.Mfull, dtype=f32)
        self.a1 = self._t(self.Mfull, C0)
        self.s0 = self._t(self.M[0], C0)
        self.s0hat = self._t(self.M[0], C0)
        self.rstd2 = self._t(self.M[0], dtype=f32)
        self.x0 = self._t(self.M[0], C0)
        # statistics arena (zeroed once per step): GRN stats fwd/bwd, loss accumulators
        self._stat_sizes = []
        self.blocks = []           # encoder blocks then the decoder block
        for i in range(4):
            for j in range(cfg.depths[i]):
                if self.dense:      # the dense Block (convnextv2.py:18-55): per-sample GRN over the stage's whole map
                    self.blocks.append(self._alloc_block(f"encoder.stages.{i}.{j}", self.M[i], dims[i], N, i, sparse=False,
                                                         rpg=self.M[i] // N))
                else:
                    self.blocks.append(self._alloc_block(f"encoder.stages.{i}.{j}", self.M[i], dims[i], 1, i, sparse=True))
        # the decoder: nn.Sequential of decoder_depth dense Blocks shared by every modality (fcmae.py:119-121,137,145)
        self.decs = [self._alloc_block(f"decoder_dict.{cfg.out_mods[0].name}.{j}", N * L, D, N, None, sparse=False)
                     for j in range(cfg.decoder_depth)]
        self.dec = self.decs[0]
        self.dec_dx = [self._t(N * L * D) for _ in range(cfg.decoder_depth - 1)]      # data gradients between the decoder blocks
        self.down = []
        for i in range(3):
            self.down.append(dict(xhat=self._t(self.M[i], dims[i]), rstd=self._t(self.M[i], dtype=f32),
                                  out=self._t(self.M[i + 1], dims[i + 1])))
        self.xdec = self._t(N * L, D)
        # heads
        self.Wpix = sum(m.head_out for m in cfg.pix_mods)
        self.Wimg = sum(m.head_out for m in cfg.img_mods)
        self.ldimg = max(8, _rup(self.Wimg, 8))
        self.pred_pix = self._t(N * L, max(self.Wpix, 8))
        self.dpred_pix = self._t(N * L, max(self.Wpix, 8))
        self.pred_img = torch.zeros(N, self.ldimg, dtype=self.tdtype, device=dev)
        self.dpred_img = torch.zeros(N, self.ldimg, dtype=self.tdtype, device=dev)
        self.yhat = self._t(N * L, D)
        self.rstd_y = self._t(N * L, dtype=f32)
        self.yln = self._t(N * L, D)
        self.pooled = self._t(N, D)
        self.dpooled = self._t(N, D)
        T = len(cfg.out_mods)
        self.n_stats = sum(self._stat_sizes)
        # persistent stage kernels: PS_NG accumulator copies of each statistics vector (same arena: zeroed once per step)
        self.PS_NG = 4      # (workgroup n adds into copy n % 4: 1 / 2 / 8 copies measured within noise, profiles/r05/ab_ps_ng.txt)
        ps_blocks = [blk for blk in self.blocks if self._ps_ok(blk["stage"])]
        sw_blocks = [blk for blk in self.blocks if blk["sparse"] and blk["C"] <= int(self.opt["dzr_maxc"]) and bool(self.opt["stats_wgrad"]) and self.dt == BF16]
        self.stats = torch.zeros(self.n_stats + 3 * self.PS_NG * sum(b["H"] for b in ps_blocks), dtype=f32, device=dev)
        off = 0
        for blk in self.blocks + self.decs:
            G, H = blk["G"], blk["H"]
            for nm in ("G2", "S0", "S1"):
                blk[nm] = self.stats[off:off + G * H]
                off += G * H
            blk["S01"] = self.stats[off - 2 * G * H:off]      # the two backward statistics vectors, adjacent (Engine.backward zeroes them)
        for blk in ps_blocks:
            for nm in ("G2", "S0", "S1"):
                blk["ps_" + nm] = self.stats[off:off + self.PS_NG * blk["H"]]
                off += self.PS_NG * blk["H"]
        for blk in sw_blocks:        # candidates for statistics-from-the-weight-gradient (stats_wgrad; decided per block in _block_fwd_mat)
            blk["sw_cand"] = True
        # per-sample {sum, count} partials; with the row-split continuous losses N * grid slots per modality (slot n * grid + row; the kernels that
        # write one partial per sample use the first N slots, the rest stay zero: the finalisation folds all of them in a fixed order)
        self.loss_slots = N
        self.loss_acc = torch.zeros(T, self.loss_slots, 2, dtype=f32, device=dev)
        self.losses = torch.zeros(T, dtype=f32, device=dev)
        self.weighted = torch.zeros(T, dtype=f32, device=dev)
        self.total = torch.zeros(1, dtype=f32, device=dev)
        self.coef = torch.zeros(T, dtype=f32, device=dev)
        npc = max(1, len([m for m in cfg.out_mods if m.kind == "pix_cont"]))
        self.patch_buf = torch.zeros(npc, 4, N * L, dtype=f32, device=dev)
        # backward scratch
        maxMH = max(b["M"] * b["H"] for b in self.blocks + self.decs)
        maxMC = max(max(b["M"] * b["C"] for b in self.blocks + self.decs), self.Mfull * C0)
        self.scr_dz2 = [self._t(maxMH) for _ in range(int(self.opt["dz_ring"]))]   # dz / dh, one per block in turn (side lane reads dh)
        self.scr_dz = self.scr_dz2[0]
        self.scr_dxn = self._t(maxMC)
        self.ring = int(self.opt["ring"])      # depth of the dd / dx rings (2 = ping-pong)
        self.scr_dd2 = [self._t(maxMC) for _ in range(self.ring)]   # dd, one per block in turn (side lane reads it)
        self.scr_dd = self.scr_dd2[0]
        # dx ring: a block's dout is still read by its pw2 weight gradient (side lane) while later blocks run; with
        # a ping-pong the data gradient two blocks on had to wait for it (20-60 us main-lane stalls in the timeline)
        self.scr_dx = [self._t(maxMC) for _ in range(self.ring)]
        self.scr_dxA, self.scr_dxB = self.scr_dx[0], self.scr_dx[1]
        self.dy = self._t(N * L, D)
        # fp32 scratch for the two-stage reductions (per-block / per-split partial slabs)
        self.ws_floats = 32 * 1024 * 1024
        self.ws = torch.empty(self.ws_floats, dtype=f32, device=dev)
        self.ws2 = torch.empty(self.ws_floats, dtype=f32, device=dev)     # side-lane (weight-gradient) scratch
        self.ws3 = torch.empty(self.ws_floats, dtype=f32, device=dev)     # second side lane (depthwise weight gradients)

    def _alloc_block(self, prefix, M, Cc, G, stage, sparse, rpg=None):
        H = 4 * Cc
        f32 = torch.float32
        blk = dict(prefix=prefix, M=M, C=Cc, H=H, G=G, stage=stage, sparse=sparse, rpg=rpg or (M if sparse else self.L),
                   d=self._t(M, Cc), dhat=self._t(M, Cc), rstd=self._t(M, dtype=f32),
                   h=self._t(M, H), out=self._t(M, Cc),
                   Gx=self._t(G * H, dtype=f32), Ainv=self._t(G, dtype=f32),
                   scale=self._t(G * H, dtype=f32), coef=self._t(G * H, dtype=f32))
        self._stat_sizes.append(3 * G * H)
        return blk

    # ------------------------------------------------------------------ weight staging
    def _build_prep(self):
        """Arena of compute-type weight copies in [N][K] layout + the device descriptor table."""
        descs = []
        self.w = {}
        chunks = []

        def add(name, src, rows, cols, sr, sc, ld=None, dst=None, coloff=0):
            """dst[r, coloff + c] = src.flat[r*sr + c*sc]; a new [rows][ld] matrix unless dst is given."""
            if dst is None:
                dst = dict(rows=rows, ld=ld or _rup(cols, 8), off=None)
                chunks.append((name, dst))
                self.w[name] = dst
            descs.append((src, None, dst, rows, cols, sr, sc, coloff))
            return dst

        P = self.params
        cfg, dims, D = self.cfg, self.cfg.dims, self.D
        C0 = dims[0]
        if getattr(cfg, "use_orig_stem", False):      # ME layout [(kw*k + kh)*Cin + cin][C0] -> [C0][k*k*Cin]
            k = P["encoder.stem_orig.0.kernel"]
            add("stem.Wt", k, C0, cfg.stem_k * cfg.stem_k * cfg.in_chans, 1, C0)
        else:
            k = P["encoder.initial_conv.0.kernel"]
            add("stem.Wt", k, C0, 9 * cfg.in_chans, 1, C0)

        def block_weights(prefix, Cc, sparse):
            H = 4 * Cc
            w1 = P[prefix + (".pwconv1.linear.weight" if sparse else ".pwconv1.weight")]
            w2 = P[prefix + (".pwconv2.linear.weight" if sparse else ".pwconv2.weight")]
            add(prefix + ".W1", w1, H, Cc, Cc, 1)
            add(prefix + ".W2", w2, Cc, H, H, 1)
            add(prefix + ".W1T", w1, Cc, H, 1, Cc)
            add(prefix + ".W2T", w2, H, Cc, 1, H)

        for blk in self.blocks:
            block_weights(blk["prefix"], blk["C"], blk["sparse"])
        for i in range(3):
            kk = P[f"encoder.downsample_layers.{i}.1.kernel"]      # (4, C, C')
            Ci, Co = dims[i], dims[i + 1]
            add(f"down{i}.Wt", kk, Co, 4 * Ci, 1, Co)            # [C'][4C]
            add(f"down{i}.W", kk, 4 * Ci, Co, Co, 1)             # [4C][C']
        add("proj.W", P["proj.weight"], D, dims[3], dims[3], 1)
        add("proj.WT", P["proj.weight"], dims[3], D, 1, dims[3])
        for d_ in self.decs:
            block_weights(d_["prefix"], D, False)
        pixT = imgT = None
        coff_p = coff_i = 0
        for om in cfg.out_mods:
            wsrc = P[f"pred_dict.{om.name}.weight"]
            add(f"head.{om.name}.W", wsrc, om.head_out, D, D, 1)
            if om.kind.startswith("pix"):
                if pixT is None:
                    pixT = add("head.pixT", wsrc, D, om.head_out, 1, D, ld=_rup(self.Wpix, 8))
                else:
                    add("head.pixT", wsrc, D, om.head_out, 1, D, dst=pixT, coloff=coff_p)
                coff_p += om.head_out
            else:
                if imgT is None:
                    imgT = add("head.imgT", wsrc, D, om.head_out, 1, D, ld=self.ldimg)
                else:
                    add("head.imgT", wsrc, D, om.head_out, 1, D, dst=imgT, coloff=coff_i)
                coff_i += om.head_out
        # all heads of a family as ONE [W, D] matrix (the flat buffer keeps their weights contiguous in column order)
        self.heads_merged = {}
        for fam, mods in (("pix", cfg.pix_mods), ("img", cfg.img_mods)):
            ws_ = [P[f"pred_dict.{m.name}.weight"] for m in mods]
            bs_ = [P[f"pred_dict.{m.name}.bias"] for m in mods]
            ok = bool(mods) and bool(self.opt["heads_merged"])
            for ts in (ws_, bs_):
                ok = ok and all(a.data_ptr() + a.numel() * 4 == b_.data_ptr() for a, b_ in zip(ts, ts[1:]))
            self.heads_merged[fam] = ok
            if ok:
                add(f"head.{fam}.W", ws_[0], sum(m.head_out for m in mods), D, D, 1)
        # lay the arena out
        total = 0
        for _, dst in chunks:
            dst["off"] = total
            total += _rup(dst["rows"] * dst["ld"], 64)
        self.warena = torch.zeros(total, dtype=self.tdtype, device=self.device)
        esz = self.warena.element_size()
        for _, dst in chunks:
            dst["t"] = self.warena[dst["off"]:dst["off"] + dst["rows"] * dst["ld"]]
        table = (_lib.PrepDesc * len(descs))()
        mx = 0
        for i, (src, _, dst, rows, cols, sr, sc, coloff) in enumerate(descs):
            table[i].src = src.data_ptr()
            table[i].dst = self.warena.data_ptr() + (dst["off"] + coloff) * esz
            table[i].rows, table[i].cols, table[i].sr, table[i].sc = rows, cols, sr, sc
            table[i].dst_ld = dst["ld"]
            mx = max(mx, ((rows + 63) // 64) * ((cols + 63) // 64))      # 64x64 tiles of the largest view
        raw = bytes(table)
        self.prep_table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        self.prep_n, self.prep_max = len(descs), mx
