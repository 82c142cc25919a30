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

F32, BF16 = 0, 1

# Launch-program options (A/B switches of the step's structure): defaults are the measured-best choices on MI355X. Override per
# engine with Engine(options={...}); the developer variable MPMAE_ENGINE_OPTS="name=value,..." is read HERE on the host side
# (the C library itself reads no environment, include/mpmae_hip.h mpmae_set_option).
ENGINE_OPTIONS = dict(
    down_grouped=1,         # LayerNorm writes the grouped operand of the 2x2/2 convolution
    rsc_small=1,            # chunked row-streaming kernels at C = 40 / 80 too
    grn_fold=1,             # GRN finalisation recomputed in the fused kernels' prologues
    rsc_pf=1,               # LDS-staged GRN vectors in the narrow kernels (needed by grn_fold)
    dzr=1,                  # dz never materialised at small C
    lanes=1,                # weight gradients on a side HIP stream
    heads_merged=1,         # one GEMM / weight gradient per head family
    stem_im2col=1,          # 3x3 stem convolution through a materialised im2col
    stem_fused=1,           # fused stem tail (patch 8)
    stem_front=1,           # ... and the 3x3 convolution in front of it in the same launch (stem_front_kernel), which also writes the im2col matrix of the weight gradient
    loss_multi=1,           # one launch per loss kind
    loss_rows=1,            # continuous pixel losses: row-band forward kernel
    loss_rows_bwd=1,        # ... and its gradient twin
    img_side=1,             # image-level head chain on the side lane
    tail_fold_group=1,      # the LayerNorm-gradient fold group that runs in order on the main lane (tail_main) as ONE launch, like the stem kernel's three folds (library: FOLD_GROUP >= 0)
    img_dgrad_side=1,       # round 5: the image-level heads' data-gradient GEMM (eight workgroups, pure latency) on the weight-gradient lane in front of the heads' weight gradients
    prep_side=1,            # weight staging of the forward on the side lane
    prep_late=1,            # the side lane runs activity + poolings FIRST and the weight staging behind them: the first stage-0 kernel (depthwise, fp32 taps) only waits for the poolings, the first staged weight is needed 50 us later
    front_side=1,           # ... followed there by the pixel-activity map and its poolings (main lane: mask -> im2col)
    z_free=1,               # fused blocks: z = GRN(gelu(h)) is not stored by the forward; pwconv2's weight gradient rebuilds it from h in its operand prologue
    z_free_maxc=40,         # ... up to this width (the prologue's GELU costs the weight-gradient lane 17 us per launch; the forward saves 27 us per block at C = 40, 13 at C = 80)
    fold_loss=1,            # data parallel: the scalar loss rides in the first gradient bucket's all-reduce (no collective of its own)
    proj_compact=1,         # proj as a plain NT GEMM on compact rows: the token kernel assembles the decoder input, its backward gathers the visible rows (no scatter / gather GEMM variants)
    zero_side=1,            # the step's zero fills (statistics, flat gradients, padded stem dW) on the side lane, ONE loss finalisation per step
    wgrad_late=1,           # pw2's weight gradient issued behind the block's second fused kernel (one main-lane event per block)
    ps=2,                   # persistent per-sample stage kernels (ps.cuh): bit 1 = (C, S) = (160, 2), bit 0 = (320, 1); one launch per stage. 2 since late round 4: with the decoder / head GEMMs on the vendor route the per-block kernels at stage 3 (2 blocks, M = 3584) measure 3.868-3.873 vs 3.892-3.897 ms for 3 in three interleaved pairs (a tie in round 3); 1: 4.04
    rsc=1,                  # chunked row-streaming kernels (rsc.cuh) at C = 160 / 320
    act_in_stem=1,          # the fused stem kernel writes the pixel-activity bytes itself (it computes them anyway): no activity launch, the first stage-0 op waits for nothing on the side lane, the poolings run behind the stem
    tail_main=-1,           # the LAST depthwise weight gradient of the backward (stage 0, block 0) and its folds in order on the MAIN lane behind the data gradient (1) or on the weight-gradient lane (0); -1 = by the lane's load: 0 where the stage-0 blocks carry their pointwise weight gradients inside the main-lane kernels (wg_fused: the lane has slack - atto 3.435 / 3.451 vs 3.462 / 3.467 ms, profiles/r06/option_sweep.txt), 1 otherwise (round 4: the lane ended 115 us after the main lane, 4.06 -> 4.02 ms; tiny 112/16 in round 6: 14.635 vs 14.70 ms, profiles/r06/tiny_option_sweep.txt); 2: the unfused pointwise pw1 weight gradient too
    dz_ring=16,             # depth of the dz / dh scratch ring (the pw1 weight gradient on the side lane reads dh); >= blocks of the net:
    ring=16,                # / of the dd / dx rings: no main-lane op ever waits for the side lane to release a scratch buffer (3 / 4: +60 us)
    rsn_maxc=192,           # largest C with the GRN application / its backward fused into the NARROW row-streaming kernels (beyond: tiled GEMMs + element-wise kernels; 384 on tiny 112/16: 16.64 vs 15.15 ms)
    dzr_maxc=80,            # largest C recomputing dz
    dw_group=1,             # ... and ONE launch for its depthwise weight gradients (mpmae_dwconv7_wgrad_group), from this stage index on (stage 0 stays per block: its weight gradients are the tail of the backward; 9 = never)
    ln_fold_defer=1,        # the LayerNorm gamma / beta gradient folds of the fused pointwise backward kernels leave the main lane: one mpmae_fold_group per stage on the weight-gradient lane
    grn_group=1,            # dense decoder blocks: GRN statistics + finalisation + application as ONE launch per direction (mpmae_grn_group_fwd / _bwd, rows of a sample in registers between the passes), gamma / beta gradient folds deferred to the side lane
    wgrad_group=1,          # ONE launch (+ one fold) for all pwconv1 / pwconv2 weight gradients of an encoder stage (mpmae_wgrad_group), issued behind the stage's data-gradient chain
    down_fused=1,           # round 6: the LayerNorm in front of a 2x2/2 downsample convolution computed in the epilogue of the stage's LAST fused pwconv2 kernel (MpmaeRsArgs.dn_*: stages 0 -> 1 and 1 -> 2; x-hat, rstd and the grouped affine output leave from the kernel that has the row in registers): no mpmae_ln_fwd_down launch, the stage output itself is never stored or re-read
    wg_fused=1,             # round 6: pwconv1's weight gradient of the stage-0 blocks (C = 40, dz recomputed) INSIDE the fused backward kernel (MpmaeRsArgs.wg_ws: U = dh^T x-hat and db1 per persistent workgroup, folded by mpmae_rs_wgrad_fold with the LayerNorm affine applied by linearity): dh is never stored (100 MB per block), the transpose-read product over dh and xn and its fold leave the weight-gradient lane, the forward does not store xn
    stats_wgrad=1,          # blocks that recompute dz (C <= dzr_maxc): the GRN backward statistics come from pwconv2's weight gradient - T = dout^T gelu(h) on the MAIN lane (mpmae_rs which = 6, csrc/rst.cuh: one read of dout and h at the price of the statistics-only pass it replaces, 33.9 vs 34.7 us at stage 0), then mpmae_grn_stats_from_wgrad -> S0, S1, dW2, db2: the weight-gradient lane loses pwconv2's transpose-read product and fold over the same two tensors (72 + 12 us per stage-0 block), the forward never stores z at those widths. Round 6: 3.471 vs 3.569 ms (profiles/r06/ab_stats_wgrad.txt). 0 = statistics pass + separate weight gradient (round 5's route to the same T - the generic gemm_tn2 kernel on the main lane - measured 3.65 ms, slower than off, and is removed)
    grn_apply_fin=1,        # unfused sparse blocks (C = 320: tiled GEMMs + element-wise GRN passes): the GRN finalisation runs in the prologue of the element-wise pass (mpmae_grn_apply_fin / _bwd_apply_fin) - two launches fewer per block on the main lane
    loss_onepass=1,         # pixel losses in ONE pass (round 5): the forward kernels also write the loss gradient without its per-modality scalar; the scalar is folded into the heads' data-gradient weights (mpmae_head_scale) and weight-gradient fold (rowscale): the dloss:pix_* kernels (69 us of the main lane, a second pass over predictions and targets) leave the step
    det=0,                  # 1 = reproducible forward: no persistent stage kernel (its GRN exchange is float atomics), library option DET = 1 (every fold as one ordered row group - parameter-gradient folds included); 4.53-4.55 vs 3.89-3.90 ms
)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _rup(x, m):
    return (x + m - 1) // m * m


class _ParamDict(OrderedDict):
    """Parameters (or gradients) by state-dict key; `alias` holds engine-internal names of the same storage (dense encoder: the stem and
    downsampling tensors under the sparse encoder's names and layouts, synth.dense_aliases) - looked up, never iterated."""

    def __init__(self):
        super().__init__()
        self.alias = {}

    def __missing__(self, key):
        return self.alias[key]


class Engine:
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

    # ------------------------------------------------------------------ op helpers
    def _ln_bwd_fn(self, *args):
        *a, stream = args
        return self.lib.mpmae_ln_bwd(*a, _p(self.ws), self.ws_floats, stream)

    def _ln_bwd_down_fn(self, *args):
        *a, stream = args
        return self.lib.mpmae_ln_bwd_down(*a, _p(self.ws), self.ws_floats, stream)

    def _ln_bwd_callable(self, Cc, down=False):
        """The LayerNorm-backward entry point of one op. With `ln_fold_defer` (bf16, two lanes) the op gets a slab of its own and a
        host-side fold record: the gamma / beta gradient fold is launched later by the segment's mpmae_fold_group on the side lane."""
        if not (self.lanes and self.dt == BF16 and bool(self.opt["ln_fold_defer"])):
            return self._ln_bwd_down_fn if down else self._ln_bwd_fn
        slab = torch.empty(1024 * 2 * Cc, dtype=torch.float32, device=self.device)      # <= 1024 workgroups (one slab row each)
        fd = _lib.FoldDesc()
        self._keepalive += [slab, fd]
        if not hasattr(self, "_fold_pending"):
            self._fold_pending = []
        self._fold_pending.append(fd)
        fn = self.lib.mpmae_ln_bwd_down_defer if down else self.lib.mpmae_ln_bwd_defer

        def call(*args, _slab=slab, _fd=fd, _fn=fn):
            *a, stream = args
            return _fn(*a, _p(_slab), _slab.numel(), C.addressof(_fd), stream)
        call.__name__ = "mpmae_ln_bwd_down" if down else "mpmae_ln_bwd"
        return call

    def _colstats_fn(self, *args):
        *a, stream = args
        return self.lib.mpmae_colstats(*a, _p(self.ws), self.ws_floats, stream)

    def _op(self, lst, name, fn, *args, kind=None, nbytes=0, flops=0, lane=0, wait=(), signal=None):
        """Append one C-ABI launch; `kind` names the kernel, nbytes/flops are its ALGORITHMIC
        traffic (operands read once + results written once) and work, for the roofline report.
        lane 0 = main dependency chain, lane 1 = side HIP stream (weight gradients); `wait` /
        `signal` are event keys ordering the two lanes (see _run)."""
        kname = kind or fn.__name__
        lst.append((name, fn, args, dict(kind=kname, bytes=int(nbytes), flops=int(flops), lane=lane,
                                         wait=tuple(wait), signal=signal)))

    # -- cross-lane hazard tracking (build time) --------------------------------------------
    def _side_read(self, key, *tensors):
        """A side-lane op (signalling `key` when done) reads these scratch tensors."""
        for t in tensors:
            self._side_readers.setdefault(t.untyped_storage().data_ptr(), []).append(key)

    def _after(self, lst):
        """Event key signalled by the most recent main-lane op of `lst` (its results are ready)."""
        for i in range(len(lst) - 1, -1, -1):
            m = lst[i][3]
            if m["lane"] == 0:
                if m["signal"] is None:
                    self._evseq += 1
                    m["signal"] = f"m{self._evseq}"
                return m["signal"]
        return None

    def _guard(self, lst, *tensors):
        """The op just appended (main lane) overwrites these scratch tensors: make it wait for every
        side-lane op still reading them."""
        keys = self._write_waits(*tensors)
        if keys:
            m = lst[-1][3]
            m["wait"] = tuple(m["wait"]) + tuple(keys)

    def _side_wgrad(self, lst, name, ppro, qpro, reads, **kw):
        """Weight gradient on the side lane: starts once the latest main-lane op has finished,
        and protects the scratch tensors it reads (`reads`) from later main-lane writers."""
        if not self.lanes:
            return self._wgrad(lst, name, ppro, qpro, **kw)
        k = self._after(lst)
        self._evseq += 1
        key = f"s{self._evseq}"
        self._wgrad(lst, name, ppro, qpro, lane=1, wait=(k,) if k else (), signal=key, **kw)
        self._side_read(key, *reads)

    # -- grouped weight gradients (mpmae_wgrad_group): the pointwise weight gradients of a stage's blocks are collected while the
    # stage's data-gradient chain is built and issued as ONE side-lane op behind it (their operands persist: one ring slot per block)
    def _group_ok(self, blk, qpro):
        return (self.lanes and bool(self.opt["wgrad_group"]) and self.dt == BF16 and blk["sparse"] and qpro == "NONE"
                and (blk["C"] % 80 == 0 or blk["C"] % 96 == 0) and blk["H"] == 4 * blk["C"])

    def _group_add(self, lst, name, reads, **kw):
        if not hasattr(self, "_group_pending"):
            self._group_pending = []
        self._group_pending.append((name, list(reads), kw))
        # a ring slot is reused every len(ring) blocks: flush before a later block of the same stage could overwrite an operand
        # (flushing a stage's groups every 2-3 blocks, so that the side lane starts under the stage's own chain: 3.66-3.68 vs 3.648 ms - not kept)
        # (re-measured in round 6 with the lighter weight-gradient lane, which idles ~300 us under the stage-2 chain: every 2 / 3 blocks 3.407-3.420 / 3.412 vs
        #  3.420-3.428 ms - inside the noise again, profiles/r06/ab_group_flush_not_kept.txt)
        if len(self._group_pending) >= 2 * min(_lib.TNG_MAXP // 2, max(1, min(len(self.scr_dz2), len(self.scr_dx)) - 2)):
            self._group_flush(lst)

    def _group_flush(self, lst, name=None):
        pend = getattr(self, "_group_pending", [])
        if not pend:
            return
        self._group_pending = []
        arr = (_lib.WgradArgs * len(pend))()
        nbytes = flops = 0
        for i, (_, _, kw) in enumerate(pend):
            a = arr[i]
            for k, v in kw.items():
                setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else (0 if v is None else v))
            a.rpg = max(int(a.M), 1)
            nbytes += int(a.M) * (int(a.Nn) + int(a.Kk)) * 2 + int(a.Nn) * int(a.Kk) * 4
            flops += 2 * int(a.M) * int(a.Nn) * int(a.Kk)
        self._keepalive.append(arr)
        stage = pend[0][0].split(":")[0].rsplit(".", 1)[0]          # "encoder.stages.2"
        k = self._after(lst)
        self._evseq += 1
        key = f"s{self._evseq}"
        self._op(lst, name or f"{stage}:pw.wgrad[{len(pend)}]", self.lib.mpmae_wgrad_group, self.dt, arr, len(pend), _p(self.ws2), self.ws_floats,
                 kind="wgrad_group", nbytes=nbytes, flops=flops, lane=1, wait=(k,) if k else (), signal=key)
        for _, reads, _ in pend:
            self._side_read(key, *reads)

    def _fold_flush(self, lst, stage, lane=1):
        pend = getattr(self, "_fold_pending", [])
        if not pend:
            return
        self._fold_pending = []
        arr = (_lib.FoldDesc * len(pend))()
        self._keepalive.append(arr)
        srcs = list(pend)

        def fold(stream, _arr=arr, _srcs=srcs):      # the records are filled by the mpmae_rs calls of the stage (recorded / issued before this op)
            for i_, fd in enumerate(_srcs):
                _arr[i_] = fd
            if lane == 0 and grouped:      # the fold group in order on the main lane (the step's exposed tail): ONE launch for its records
                fg = _lib.OPT["FOLD_GROUP"]
                old = self.lib.mpmae_get_option(fg)
                self.lib.mpmae_set_option(fg, 1)
                try:
                    return self.lib.mpmae_fold_group(_arr, len(_srcs), stream)
                finally:
                    self.lib.mpmae_set_option(fg, old)
            return self.lib.mpmae_fold_group(_arr, len(_srcs), stream)
        grouped = bool(self.opt["tail_fold_group"])
        k = self._after(lst) if lane else None
        self._op(lst, f"{stage}:ln.fold[{len(pend)}]", fold, kind="ln_fold_group", lane=lane, wait=(k,) if k else ())

    def _write_waits(self, *tensors):
        """Event keys a main-lane op must wait for before overwriting these scratch tensors."""
        keys = []
        for t in tensors:
            keys += self._side_readers.pop(t.untyped_storage().data_ptr(), [])
        return keys

    def _gemm(self, lst, name, pro, epi, **kw):
        a = _lib.GemmArgs()
        for k, v in kw.items():
            setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else (0 if v is None else v))
        if not kw.get("rpg"):
            a.rpg = max(int(a.M), 1)
        a.ws, a.ws_floats = self.ws.data_ptr(), self.ws_floats
        self._keepalive.append(a)
        esz = 4 if self.dt == F32 else 2
        M_, N_, K_ = int(a.M), int(a.N), int(a.K)
        a_bytes = M_ * K_ * esz * (2 if pro == "GRN_BWD" else 1)
        if pro == "IM2COL3":
            a_bytes = M_ * int(a.Cseg) * 4           # each visible pixel's channels read once
        c_bytes = M_ * N_ * esz * (2 if epi in ("RESID", "DZ_STATS") else 1)
        self._op(lst, name, self.lib.mpmae_gemm, self.dt, PRO[pro], EPI[epi], C.byref(a),
                 kind=f"gemm<{pro},{epi}>", nbytes=a_bytes + c_bytes + N_ * K_ * esz, flops=2 * M_ * N_ * K_)

    def _wgrad(self, lst, name, ppro, qpro, lane=0, wait=(), signal=None, **kw):
        a = _lib.WgradArgs()
        for k, v in kw.items():
            setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else (0 if v is None else v))
        if not kw.get("rpg"):
            a.rpg = max(int(a.M), 1)
        tiles = ((a.Nn + 63) // 64) * ((a.Kk + 63) // 64)
        splits = max(1, min((768 + tiles - 1) // tiles, (a.M + 255) // 256))
        a.ws, a.ws_floats = (self.ws2 if lane == 1 else self.ws).data_ptr(), self.ws_floats
        self._keepalive.append(a)
        esz = 4 if self.dt == F32 else 2
        M_, N_, K_ = int(a.M), int(a.Nn), int(a.Kk)
        p_bytes = M_ * N_ * esz * (2 if ppro == "GRN_BWD" else 1)
        q_bytes = M_ * int(a.Cseg) * 4 if qpro == "IM2COL3" else M_ * K_ * esz
        self._op(lst, name, self.lib.mpmae_wgrad, self.dt, PRO[ppro], PRO[qpro], C.byref(a), splits,
                 kind=f"wgrad<{ppro},{qpro}>", nbytes=p_bytes + q_bytes + N_ * K_ * 4, flops=2 * M_ * N_ * K_,
                 lane=lane, wait=wait, signal=signal)

    # ---- MX-fp8 pointwise path (decoder block) ---------------------------------------------------------------
    def _mx_buf(self, name, rows, K):
        """e4m3 matrix [rows][K] + slab-major block scales [K/128][rows] (include/mpmae_hip.h, mpmae_quant_mx)."""
        if name not in self.mx:
            self.mx[name] = dict(q=torch.empty(rows * K, dtype=torch.uint8, device=self.device),
                                 s=torch.zeros((K // 128) * rows, dtype=torch.int32, device=self.device), rows=rows, K=K)
        return self.mx[name]

    def _quant(self, lst, name, src, ld, buf):
        self._op(lst, name, self.lib.mpmae_quant_mx, _p(src), ld, buf["rows"], buf["K"], _p(buf["q"]), _p(buf["s"]), buf["rows"],
                 kind="quant_mx", nbytes=buf["rows"] * buf["K"] * 3)

    def _gemm_mx(self, lst, name, epi, qa, qb, **kw):
        a = _lib.GemmArgs()
        for k, v in kw.items():
            setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else (0 if v is None else v))
        a.A, a.B = qa["q"].data_ptr(), qb["q"].data_ptr()
        a.lda, a.ldb, a.rpg = qa["K"], qb["K"], max(int(a.M), 1)
        self._keepalive.append(a)
        M_, N_, K_ = int(a.M), int(a.N), int(a.K)
        self._op(lst, name, self.lib.mpmae_gemm_mx, EPI[epi], C.byref(a), _p(qa["s"]), qa["rows"], _p(qb["s"]), qb["rows"],
                 kind=f"gemm_mx<{epi}>", nbytes=M_ * K_ + N_ * K_ + M_ * N_ * 2 * (2 if epi == "RESID" else 1), flops=2 * M_ * N_ * K_)

    def _mx_block(self, blk):
        return self.fp8 and not blk["sparse"] and blk["C"] % 128 == 0

    def _mx_sparse(self, blk):
        """Round 6 (BASELINE config 5, 'fp8 MFMA pointwise path' on the ENCODER): the sparse blocks whose K = H pointwise products are plain tiled GEMMs
        (stage 3 of atto: pwconv2 forward and pwconv1's data gradient, K = 1280, rows masked by the activity bytes) take the MX-fp8 GEMM too: the
        activation operand (z, dh) is quantised by mpmae_quant_mx, the staged weights once per step. The K = C products of those blocks (320: not a
        multiple of the 128-element MX slab) and every block of stages 0-2 run inside the fused row-streaming kernels, whose matrix time is 6 % of
        their duration (profiles/r05/mfma_util.txt) - nothing for a faster MFMA to shorten."""
        return self.fp8 and blk["sparse"] and blk.get("rs_n") is None and blk["H"] % 128 == 0 and blk["C"] % 8 == 0

    def _mx_weight(self, wname):
        """Staged bf16 weight [N][K] -> e4m3 + scales, re-quantised once per step right after weight staging (fwd op list)."""
        w = self.w[wname]
        buf = self._mx_buf("w:" + wname, w["rows"], w["ld"])
        if wname not in self._mx_wq:
            self._mx_wq[wname] = (w, buf)
        return buf

    def _rs_ok(self, blk):
        return self._rsc_ok(blk)

    def _rs_plan(self, blk):
        """(wide, narrow): which row-streaming kernels a block uses. wide: LN+pw1 / pw2.dgrad fused with
        their GRN statistics (which 0/1); narrow: None (tiled GEMMs + element-wise kernels) or "fused" (which 4/5, GRN application
        and its backward in the operand prologue). Measured on MI355X at bs 256: fused wins for C <= 192; at C = 320 (M = 4864 rows,
        76 workgroups) the tiled GEMMs are faster than the narrow row-streaming kernel."""
        if not self._rs_ok(blk):
            return False, None
        return True, ("fused" if blk["C"] <= int(self.opt["rsn_maxc"]) else None)      # (C = 320 / 384: the narrow kernels need 250 VGPRs - tiled GEMMs)

    def _rsc_ok(self, blk):
        """chunked row-streaming kernels (rsc.cuh) with the GRN application / its backward fused in"""
        shapes = (((160, 640), (320, 1280), (192, 768), (384, 1536))      # (atto stages 2-3; tiny stages 1-2, BASELINE config 4)
                  + (((40, 160), (80, 320), (96, 384)) if self.rsc_small else ()))
        return (self.dt == BF16 and blk["sparse"] and not self.disable_rs and not self.disable_rsc
                and (blk["C"], blk["H"]) in shapes)

    def _rs(self, lst, name, which, blk, nbytes, flops, **kw):
        a = _lib.RsArgs()
        for k, v in kw.items():
            setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else (0 if v is None else v))
        a.M, a.C, a.H = blk["M"], blk["C"], blk["H"]
        if "ws" not in kw:
            a.ws, a.ws_floats = self.ws.data_ptr(), self.ws_floats
        self._keepalive.append(a)
        self._op(lst, name, self.lib.mpmae_rs, which, C.byref(a), kind=f"rs<{which}>", nbytes=nbytes, flops=flops)

    def _geom(self, stage):
        g = _lib.Geom()
        if stage is None:      # dense decoder grid
            g.vis, g.inv, g.N, g.keep, g.grid, g.S = 0, 0, self.N, self.L, self.grid, 1
        elif self.dense:       # every patch present: NULL tables, slot = patch
            g.vis, g.inv, g.N, g.keep, g.grid, g.S = 0, 0, self.N, self.L, self.grid, self.S[stage]
        else:
            g.vis, g.inv = self.vis.data_ptr(), self.inv.data_ptr()
            g.N, g.keep, g.grid, g.S = self.N, self.keep, self.grid, self.S[stage]
        return g

    def _dw_tiling(self, stage, Cc):
        S = 1 if stage is None else self.S[stage]
        TP = {8: 1, 4: 2, 2: 4, 1: 7}[S]
        if TP * S > 8:
            TP = 8 // S
        tiles_side = (self.grid + TP - 1) // TP
        CC = Cc if Cc <= 96 else 64
        return TP, tiles_side, CC

    def _dw_weight(self, blk):
        P, G = self.params, self.grads
        pre = blk["prefix"]
        Cc = blk["C"]
        if blk["sparse"]:   # ME kernel (49, C), index (kw*7+kh)*C + c
            return P[pre + ".dwconv.kernel"], G[pre + ".dwconv.kernel"], P[pre + ".dwconv.bias"], G[pre + ".dwconv.bias"], (Cc, 7 * Cc, 1)
        return P[pre + ".dwconv.weight"], G[pre + ".dwconv.weight"], P[pre + ".dwconv.bias"], G[pre + ".dwconv.bias"], (7, 1, 49)

    def _dwconv(self, lst, name, blk, x, out, add, flip, with_bias):
        w, _, b, _, (skh, skw, sc) = self._dw_weight(blk)
        TP, ts, CC = self._dw_tiling(blk["stage"], blk["C"])
        a = _lib.DwArgs()
        a.x, a.out, a.add = x.data_ptr(), out.data_ptr(), (add.data_ptr() if add is not None else 0)
        a.w, a.bias = w.data_ptr(), (b.data_ptr() if with_bias else 0)
        a.s_kh, a.s_kw, a.s_c, a.flip = skh, skw, sc, flip
        a.g = self._geom(blk["stage"])
        a.C, a.CC, a.TP, a.tiles_side = blk["C"], CC, TP, ts
        act = self.act[blk["stage"]] if blk["sparse"] else None
        a.act = act.data_ptr() if act is not None else 0
        self._keepalive.append(a)
        esz = 4 if self.dt == F32 else 2
        mc = blk["M"] * blk["C"]
        self._op(lst, name, self.lib.mpmae_dwconv7_fwd, self.dt, C.byref(a), kind="dwconv7",
                 nbytes=mc * esz * (3 if add is not None else 2), flops=2 * 49 * mc)

    def _block_names(self, blk):
        pre, sp = blk["prefix"], blk["sparse"]
        return dict(
            ln_w=pre + (".norm.ln.weight" if sp else ".norm.weight"),
            ln_b=pre + (".norm.ln.bias" if sp else ".norm.bias"),
            w1=pre + (".pwconv1.linear.weight" if sp else ".pwconv1.weight"),
            b1=pre + (".pwconv1.linear.bias" if sp else ".pwconv1.bias"),
            w2=pre + (".pwconv2.linear.weight" if sp else ".pwconv2.weight"),
            b2=pre + (".pwconv2.linear.bias" if sp else ".pwconv2.bias"),
            gg=pre + ".grn.gamma", gb=pre + ".grn.beta")

    # ------------------------------------------------------------------ forward program
    # ---- block programs -------------------------------------------------------------------
    # "fused": LN-affine / GELU / GRN are applied in the GEMM prologues (fewest bytes; used for the
    #          bandwidth-bound stages with small C).
    # "mat"  : xn, z = GRN(GELU(h)) and dh are materialised by row-wise kernels so that every GEMM is
    #          a plain NT / TN product on the fast bf16 MFMA kernels (compute-shaped layers).
    def _block_mode(self, blk):
        if self.block_mode_override:
            return self.block_mode_override
        return "mat" if self.dt == BF16 else "fused"      # measured on MI355X: mat 13.2 vs fused-small-C 14.5 ms/step

    def _block_fwd(self, lst, blk, x):
        blk["mode"] = self._block_mode(blk)
        return (self._block_fwd_mat if blk["mode"] == "mat" else self._block_fwd_fused)(lst, blk, x)

    def _block_bwd(self, lst, blk, dout, dx):
        return (self._block_bwd_mat if blk["mode"] == "mat" else self._block_bwd_fused)(lst, blk, dout, dx)

    # ---- persistent per-sample stage kernels (csrc/ps.cuh) ----------------------------------
    def _ps_ok(self, stage):
        """One launch for the whole stage: bf16, (C, S) = (160, 2) or (320, 1), every sample's workgroup resident (N <= CUs)."""
        if not self.opt["ps"] or self.dt != BF16 or self.disable_rs or (self.block_mode_override or "mat") != "mat" or self.dense:
            return False
        Cc, S, depth = self.cfg.dims[stage], self.S[stage], self.cfg.depths[stage]
        if (Cc, S) not in ((160, 2), (320, 1)) or depth > _lib.PS_MAXBLK or not (int(self.opt["ps"]) >> (S - 1)) & 1:
            return False
        if self.keep * S * S > (80 if S == 2 else 32) or self.keep * S * S * Cc * 4 >= 65535:
            return False
        cus = torch.cuda.get_device_properties(self.device).multi_processor_count if self.device.type == "cuda" else 256
        return self.N <= cus

    def _stage_fwd_ps(self, lst, stage, blks, x):
        P = self.params
        a = _lib.PsArgs()
        a.x_in, a.g, a.act = x.data_ptr(), self._geom(stage), (self.act[stage].data_ptr() if self.act[stage] is not None else 0)
        a.C, a.nblk, a.eps, a.ng = blks[0]["C"], len(blks), 1e-6, self.PS_NG
        if not hasattr(self, "ps_sync"):
            self.ps_sync = torch.zeros(8, 64, dtype=torch.int32, device=self.device)     # one {arrivals, departures, error, -, debug...} row per launch
            self._ps_launches = 0
        a.sync = self.ps_sync[self._ps_launches].data_ptr()
        self._ps_launches += 1
        nbytes = flops = 0
        for i, blk in enumerate(blks):
            nm, tag = self._block_names(blk), blk["prefix"]
            M, Cc, H = blk["M"], blk["C"], blk["H"]
            blk["mode"], blk["x"] = "mat", x
            if "xn" not in blk:
                blk["xn"] = self._t(M, Cc)
                blk["z"] = self._t(M, H)
            blk["rs"], blk["rs_n"] = self._rs_plan(blk)
            blk["grn_fold"] = (blk["rs_n"] == "fused" and blk["G"] == 1 and self.grn_fold)
            b = a.blk[i]
            b.dw_w, b.dw_b = P[tag + ".dwconv.kernel"].data_ptr(), P[tag + ".dwconv.bias"].data_ptr()
            b.ln_g, b.ln_b = P[nm["ln_w"]].data_ptr(), P[nm["ln_b"]].data_ptr()
            w1, w2 = self.w[tag + ".W1"], self.w[tag + ".W2"]
            b.W1, b.ldw1, b.b1 = w1["t"].data_ptr(), w1["ld"], P[nm["b1"]].data_ptr()
            b.W2, b.ldw2, b.b2 = w2["t"].data_ptr(), w2["ld"], P[nm["b2"]].data_ptr()
            b.grn_g, b.grn_b = P[nm["gg"]].data_ptr(), P[nm["gb"]].data_ptr()
            b.dhat, b.rstd, b.xn, b.h, b.z, b.out = (blk[k].data_ptr() for k in ("dhat", "rstd", "xn", "h", "z", "out"))
            b.G2, b.Gx, b.Ainv, b.scale = (blk[k].data_ptr() for k in ("ps_G2", "Gx", "Ainv", "scale"))
            nbytes += (4 * M * Cc + 2 * M * H) * 2 + 2 * Cc * H * 2       # x-hat, xn, out written + one read of x; h, z written; weights once
            flops += 4 * M * Cc * H + 2 * 49 * M * Cc
            x = blk["out"]
        self._keepalive.append(a)
        self._op(lst, f"encoder.stages.{stage}:ps.fwd[{len(blks)}]", self.lib.mpmae_ps_fwd, C.byref(a), kind="ps_fwd", nbytes=nbytes, flops=flops)
        return x

    def _block_fwd_mat(self, lst, blk, x):
        P, lib, dt = self.params, self.lib, self.dt
        nm = self._block_names(blk)
        M, Cc, H, G = blk["M"], blk["C"], blk["H"], blk["G"]
        act = self.act[blk["stage"]] if blk["sparse"] else None
        rpg = blk["rpg"]
        eps = 1e-6 if blk["sparse"] else 1e-4
        tag = blk["prefix"]
        esz = 4 if dt == F32 else 2
        blk["x"] = x
        if "xn" not in blk:
            blk["xn"] = self._t(M, Cc)
            blk["z"] = self._t(M, H)
        self._dwconv(lst, tag + ":dw", blk, x, blk["d"], None, 0, True)
        rs, rs_n = self._rs_plan(blk)
        blk["rs"], blk["rs_n"] = rs, rs_n
        # pwconv1's weight gradient inside the fused backward kernel (wg_fused): the same conditions as the dz recomputation it rides on, C = 40
        blk["wgf"] = (bool(self.opt["wg_fused"]) and rs and rs_n == "fused" and Cc == 40 and blk["sparse"] and G == 1 and self.grn_fold
                      and Cc <= int(self.opt["dzr_maxc"]) and self.dz_recompute
                      and bool(self.opt["rsc_pf"])
                      # (it lives in the persistent 4-wave backward kernel of rsp.cuh: the library switches that select another kernel switch it off)
                      and lib.mpmae_get_option(_lib.OPT["RSP"]) > 0 and (lib.mpmae_get_option(_lib.OPT["RSP_NARROW"]) & 2)
                      and lib.mpmae_get_option(_lib.OPT["RSP_NWV"]) in (0, 4) and lib.mpmae_get_option(_lib.OPT["RSC_PF"]) > 0)
        if rs:   # LN + pwconv1 + GELU^2 column sums in one row-streaming kernel
            self._rs(lst, tag + ":ln+pw1", 0, blk, ((2 if blk["wgf"] else 3) * M * Cc + M * H) * esz, 2 * M * Cc * H, A=blk["d"],
                     W=self.w[tag + ".W1"]["t"], ldw=self.w[tag + ".W1"]["ld"], bias=P[nm["b1"]], v0=P[nm["ln_w"]],
                     v1=P[nm["ln_b"]], out=blk["h"], xhat=blk["dhat"], xn=None if blk["wgf"] else blk["xn"], rstd=blk["rstd"], act=act,
                     s0=blk["G2"])
        else:
            self._op(lst, tag + ":ln", lib.mpmae_ln_fwd, dt, _p(blk["d"]), _p(blk["dhat"]), _p(blk["rstd"]),
                     _p(blk["xn"]), _p(P[nm["ln_w"]]), _p(P[nm["ln_b"]]), 0, 1e-6, M, Cc, _p(act), kind="ln_fwd",
                     nbytes=3 * M * Cc * esz)
        if rs:
            pass
        elif blk["sparse"]:      # column sums ride in the GEMM epilogue
            self._gemm(lst, tag + ":pw1", "NONE", "GELU_SUMSQ", A=blk["xn"], B=self.w[tag + ".W1"]["t"],
                       bias=P[nm["b1"]], C=blk["h"], M=M, N=H, K=Cc, lda=Cc, ldb=self.w[tag + ".W1"]["ld"], ldc=H,
                       rpg=rpg, s0=blk["G2"], act=act)
        elif self._mx_block(blk):
            qx, qw = self._mx_buf(tag + ".xn", M, Cc), self._mx_weight(tag + ".W1")
            self._quant(lst, tag + ":xn.quant", blk["xn"], Cc, qx)
            self._gemm_mx(lst, tag + ":pw1", "STORE", qx, qw, bias=P[nm["b1"]], C=blk["h"], M=M, N=H, K=Cc, ldc=H, act=act)
            self._op(lst, tag + ":grn.stats", self._colstats_fn, dt, _p(blk["h"]), None, 0, _p(blk["G2"]), None, M, H,
                     rpg, kind="colstats", nbytes=M * H * esz)
        else:
            self._gemm(lst, tag + ":pw1", "NONE", "STORE", A=blk["xn"], B=self.w[tag + ".W1"]["t"], bias=P[nm["b1"]],
                       C=blk["h"], M=M, N=H, K=Cc, lda=Cc, ldb=self.w[tag + ".W1"]["ld"], ldc=H, act=act)
            self._op(lst, tag + ":grn.stats", self._colstats_fn, dt, _p(blk["h"]), None, 0, _p(blk["G2"]), None, M, H,
                     rpg, kind="colstats", nbytes=M * H * esz)
        fold = blk["grn_fold"] = (rs_n == "fused" and G == 1 and self.grn_fold
                                 )
        gg = blk["grn_group"] = (not rs and not blk["sparse"] and bool(self.opt["grn_group"])
                                 and bool(lib.mpmae_grn_group_ok(dt, M, H, rpg)))
        if gg:      # the three GRN launches (statistics just appended, finalisation, application) as one
            assert lst[-1][0] == tag + ":grn.stats"
            lst.pop()
            self._op(lst, tag + ":grn.group", lib.mpmae_grn_group_fwd, dt, _p(blk["h"]), _p(blk["z"]), _p(P[nm["gg"]]),
                     _p(P[nm["gb"]]), eps, M, H, rpg, _p(blk["Gx"]), _p(blk["Ainv"]), _p(blk["scale"]),
                     kind="grn_group_fwd", nbytes=2 * M * H * esz)
        afin = blk["afin"] = (not fold and not gg and rs_n != "fused" and G == 1 and blk["sparse"] and bool(self.opt["grn_apply_fin"])
                              and H % 8 == 0 and H <= 8160)
        if gg or afin:
            pass
        elif not fold:
            self._op(lst, tag + ":grn", lib.mpmae_grn_fwd_finalize, _p(blk["G2"]), _p(P[nm["gg"]]), eps, G, H,
                     _p(blk["Gx"]), _p(blk["Ainv"]), _p(blk["scale"]))
        # z_free: z is never written - pw2's weight gradient (mpmae_wgrad with the GRN prologue on Q = h) rebuilds it slab by slab
        blk["z_free"] = (rs_n == "fused" and G == 1 and bool(self.opt["z_free"])
                         and Cc % 8 == 0 and Cc <= int(self.opt["z_free_maxc"]) and blk["sparse"])
        # statistics from the weight gradient (stats_wgrad): the blocks whose backward recomputes dz; z is then never needed (T = dout^T gelu(h))
        blk["sw"] = (bool(blk.get("sw_cand")) and rs and rs_n == "fused" and Cc <= int(self.opt["dzr_maxc"]) and fold and self.dz_recompute and G == 1
                     and Cc % 8 == 0 and self.lanes)
        if blk["sw"]:
            blk["z_free"] = True
        if rs_n == "fused":   # z = GRN(gelu(h)) computed in the pw2 operand prologue (and stored for pw2.wgrad)
            fin = dict(fin_sum=blk["G2"], fin_gamma=P[nm["gg"]], fin_gx=blk["Gx"], fin_ainv=blk["Ainv"],
                       fin_out=blk["scale"], fin_eps=eps) if fold else {}   # GRN finalisation folded into the prologue
            # (h recomputed from xn in this kernel - 26 MB instead of 105 MB read per stage-0 block - measured as noise in round 2, 5.146 vs 5.152 ms; the
            #  engine route is removed in round 6, the kernel form stays pinned by test_dz_recomputation_matches_the_materialised_path)
            self._rs(lst, tag + ":grn.apply+pw2", 4, blk, (2 * M * H + 2 * M * Cc) * esz, 2 * M * Cc * H, A=blk["h"],
                     W=self.w[tag + ".W2"]["t"], ldw=self.w[tag + ".W2"]["ld"], bias=P[nm["b2"]], v0=blk["scale"],
                     v1=P[nm["gb"]], out=blk["out"], xn=None if blk["z_free"] else blk["z"], R=x, act=act, rpg=0, **fin)
            return blk["out"]
        if afin:
            self._op(lst, tag + ":grn+apply", lib.mpmae_grn_apply_fin, dt, _p(blk["h"]), _p(blk["z"]), _p(blk["G2"]), _p(P[nm["gg"]]), _p(P[nm["gb"]]),
                     eps, M, H, _p(act), _p(blk["Gx"]), _p(blk["Ainv"]), _p(blk["scale"]), kind="grn_apply", nbytes=2 * M * H * esz)
        elif not gg:
            self._op(lst, tag + ":grn.apply", lib.mpmae_grn_apply, dt, _p(blk["h"]), _p(blk["z"]), _p(blk["scale"]),
                     _p(P[nm["gb"]]), M, H, rpg, _p(act), kind="grn_apply", nbytes=2 * M * H * esz)
        if self._mx_block(blk) or self._mx_sparse(blk):
            qz, qw = self._mx_buf(tag + ".z", M, H), self._mx_weight(tag + ".W2")
            self._quant(lst, tag + ":z.quant", blk["z"], H, qz)
            self._gemm_mx(lst, tag + ":pw2", "RESID", qz, qw, bias=P[nm["b2"]], C=blk["out"], R=x, M=M, N=Cc, K=H, ldc=Cc, ldr=Cc, act=act)
        else:
            self._gemm(lst, tag + ":pw2", "NONE", "RESID", A=blk["z"], B=self.w[tag + ".W2"]["t"], bias=P[nm["b2"]],
                       C=blk["out"], R=x, M=M, N=Cc, K=H, lda=H, ldb=self.w[tag + ".W2"]["ld"], ldc=Cc, ldr=Cc, act=act)
        return blk["out"]

    def _block_bwd_mat(self, lst, blk, dout, dx):
        P, Gd, lib, dt = self.params, self.grads, self.lib, self.dt
        nm = self._block_names(blk)
        M, Cc, H, G = blk["M"], blk["C"], blk["H"], blk["G"]
        act = self.act[blk["stage"]] if blk["sparse"] else None
        rpg = blk["rpg"]
        tag = blk["prefix"]
        esz = 4 if dt == F32 else 2
        t = self._bwd_t = getattr(self, "_bwd_t", -1) + 1      # dz / dd alternate per block: the side lane reads them
        dz = self.scr_dz2[t % len(self.scr_dz2)][:M * H]
        dxn = self.scr_dxn[:M * Cc]
        dd = self.scr_dd2[t % len(self.scr_dd2)][:M * Cc]
        w2t, w1t = self.w[tag + ".W2T"], self.w[tag + ".W1T"]
        rs, rs_n = blk.get("rs", False), blk.get("rs_n")
        # HBM-bound stages: dz is never materialised - pw2.dgrad only produces the GRN statistics and the fused
        # pw1.dgrad kernel recomputes dz = dout W2 chunk by chunk (MpmaeRsArgs.dz_*)
        dzr = (rs and rs_n == "fused" and Cc <= int(self.opt["dzr_maxc"]) and blk.get("grn_fold", False)
               and self.dz_recompute)
        sw = bool(blk.get("sw")) and dzr
        if sw:
            # round 6: T = dout^T gelu(h) by the persistent kernel of csrc/rst.cuh at the statistics pass's price (round 5's route - T through the generic
            # gemm_tn2 kernel on the main lane - lost: 3.65 ms). The kernel leaves one slab row [C H | C] per workgroup plus a small one [2 H] with the workgroup's
            # share of S0 / S1: only the small ones are folded here (the fused backward kernel waits for them); the big ones become dW2 / db2 on the
            # weight-gradient lane
            w2s_ = self.w[tag + ".W2"]
            if "t_slab" not in blk:
                cus = torch.cuda.get_device_properties(self.device).multi_processor_count if self.device.type == "cuda" else 256
                blk["t_slab"] = torch.empty(3 * cus * (Cc * H + Cc + 2 * H), dtype=torch.float32, device=self.device)
                blk["t_rows"] = C.c_int(0)
            self._rs(lst, tag + ":pw2.wgrad(T)+stats", 6, blk, (M * Cc + M * H) * esz + Cc * H * 4, 2 * M * Cc * H, A=dout, R=blk["h"],
                     W=w2s_["t"], ldw=w2s_["ld"], s0=blk["S0"], s1=blk["S1"], ws=blk["t_slab"], ws_floats=blk["t_slab"].numel(),
                     wg_rows=C.addressof(blk["t_rows"]))

            def tfold(stream, _b=blk, _c=Cc, _h=H, _g=P[nm["gb"]], _dw=Gd[nm["w2"]], _db=Gd[nm["b2"]]):
                return lib.mpmae_rs_wgrad_fold(_c, _h, _p(_b["t_slab"]), _b["t_rows"].value, _p(_b["scale"]), _p(_g), _p(_dw), _p(_db), stream)
            if self.lanes:
                k = self._after(lst)
                self._evseq += 1
                self._op(lst, tag + ":pw2.wgrad.fold", tfold, kind="rs_wgrad_fold", nbytes=blk["t_slab"].numel() * 4 // 3, lane=1, wait=(k,) if k else (),
                         signal=f"s{self._evseq}")
            else:
                self._op(lst, tag + ":pw2.wgrad.fold", tfold, kind="rs_wgrad_fold", nbytes=blk["t_slab"].numel() * 4 // 3)
        elif rs:
            self._rs(lst, tag + ":pw2.dgrad", 1, blk, (M * Cc + (1 if dzr else 2) * M * H) * esz, 2 * M * Cc * H, A=dout,
                     W=w2t["t"], ldw=w2t["ld"], out=None if dzr else dz, R=blk["h"], s0=blk["S0"], s1=blk["S1"])
        elif blk["sparse"]:
            self._gemm(lst, tag + ":pw2.dgrad", "NONE", "DZ_STATS", A=dout, B=w2t["t"], C=dz, R=blk["h"], M=M, N=H,
                       K=Cc, lda=Cc, ldb=w2t["ld"], ldc=H, ldr=H, rpg=rpg, s0=blk["S0"], s1=blk["S1"])
        elif self._mx_block(blk):
            qd, qw = self._mx_buf(tag + ".dout", M, Cc), self._mx_weight(tag + ".W2T")
            self._quant(lst, tag + ":dout.quant", dout, Cc, qd)
            self._gemm_mx(lst, tag + ":pw2.dgrad", "STORE", qd, qw, C=dz, M=M, N=H, K=Cc, ldc=H)
        else:
            self._gemm(lst, tag + ":pw2.dgrad", "NONE", "STORE", A=dout, B=w2t["t"], C=dz, M=M, N=H, K=Cc, lda=Cc,
                       ldb=w2t["ld"], ldc=H)
        self._guard(lst, dz)
        # pw2's weight gradient only reads dout and z: issued right here it needs an event of its own between the two fused kernels of
        # the main lane; with `wgrad_late` it is issued behind the second one and shares that kernel's event with pw1 / depthwise
        late_w2 = self.lanes and bool(self.opt["wgrad_late"]) and rs and rs_n == "fused"
        late_all = self.lanes and int(self.opt["wgrad_late"]) >= 2 and not rs and rs_n is None      # unfused blocks: all three behind ln.bwd
        w2_args = dict(P=dout, Q=blk["z"], M=M, Nn=Cc, Kk=H, ldp=Cc, ldq=H, dW=Gd[nm["w2"]], sn=H, sk=1, db=Gd[nm["b2"]])
        w2_qpro = "NONE"
        if blk.get("z_free"):
            w2_args.update(Q=blk["h"], qp0=blk["scale"], qp1=P[nm["gb"]])
            w2_qpro = "GRN"
        grouped = self._group_ok(blk, w2_qpro)
        if sw:
            pass                     # (pwconv2's weight gradient is already out: it produced the statistics)
        elif grouped:
            self._group_add(lst, tag + ":pw2.wgrad", [dout], **w2_args)
        elif not late_w2 and not late_all:
            self._side_wgrad(lst, tag + ":pw2.wgrad", "NONE", w2_qpro, [dout], **w2_args)
        gg = blk.get("grn_group", False)
        if gg:
            # statistics + finalisation + dh over dz in one launch; the samples' gamma / beta gradient rows are folded on the side lane
            # by the decoder's fold op (a static record: slab[G][2H] -> dgamma[H], dbeta[H])
            if "grn_slab" not in blk:
                blk["grn_slab"] = torch.empty(G * 2 * H, dtype=torch.float32, device=self.device)
            self._op(lst, tag + ":grn.bgroup", lib.mpmae_grn_group_bwd, dt, _p(dz), _p(blk["h"]), _p(blk["scale"]), _p(blk["Gx"]),
                     _p(blk["Ainv"]), _p(P[nm["gg"]]), M, H, rpg, _p(blk["grn_slab"]), kind="grn_group_bwd", nbytes=3 * M * H * esz)
            delta = (Gd[nm["gb"]].data_ptr() - Gd[nm["gg"]].data_ptr()) // 4
            assert abs(delta) < 2 ** 31
            fd = _lib.FoldDesc(blk["grn_slab"].data_ptr(), G, 2 * H, Gd[nm["gg"]].data_ptr(), H, delta, 1)
            self._keepalive.append(fd)
            if not hasattr(self, "_fold_pending"):
                self._fold_pending = []
            self._fold_pending.append(fd)
        elif not blk["sparse"]:
            self._op(lst, tag + ":grn.bstats", self._colstats_fn, dt, _p(blk["h"]), _p(dz), 1, _p(blk["S0"]),
                     _p(blk["S1"]), M, H, rpg, kind="colstats", nbytes=2 * M * H * esz)
        fold = blk.get("grn_fold", False)
        afin = bool(blk.get("afin")) and rs_n != "fused" and not gg
        if not fold and not gg and not afin:
            self._op(lst, tag + ":grn.bwd", lib.mpmae_grn_bwd_finalize, _p(blk["S0"]), _p(blk["S1"]), _p(blk["Gx"]),
                     _p(blk["Ainv"]), _p(P[nm["gg"]]), G, H, _p(blk["coef"]), _p(Gd[nm["gg"]]), _p(Gd[nm["gb"]]))
        rsc = rs_n == "fused"
        if afin:
            self._op(lst, tag + ":grn.bwd+bapply", lib.mpmae_grn_bwd_apply_fin, dt, _p(dz), _p(blk["h"]), _p(blk["scale"]), _p(blk["S0"]), _p(blk["S1"]),
                     _p(blk["Gx"]), _p(blk["Ainv"]), _p(P[nm["gg"]]), M, H, _p(blk["coef"]), _p(Gd[nm["gg"]]), _p(Gd[nm["gb"]]),
                     kind="grn_bwd_apply", nbytes=3 * M * H * esz)
        elif not rsc and not gg:
            self._op(lst, tag + ":grn.bapply", lib.mpmae_grn_bwd_apply, dt, _p(dz), _p(blk["h"]), _p(blk["scale"]),
                     _p(blk["coef"]), M, H, rpg, kind="grn_bwd_apply", nbytes=3 * M * H * esz)
        if rsc:  # dh (written over dz) in the operand prologue, pwconv1 data gradient, LayerNorm backward
            dzkw = dict(dz_dout=dout, dz_w2t=w2t["t"], dz_ldw2=w2t["ld"]) if dzr else {}
            if self.lanes and bool(self.opt["ln_fold_defer"]) and blk["sparse"]:
                # its own slab (nothing else may touch it until the stage's fold op has run) + a host-side fold record
                if "ln_slab" not in blk:
                    blk["ln_slab"] = torch.empty(((M + 63) // 64 + 1) * 2 * Cc, dtype=torch.float32, device=self.device)
                if not hasattr(self, "_fold_pending"):
                    self._fold_pending = []
                fd = _lib.FoldDesc()
                self._keepalive.append(fd)
                self._fold_pending.append(fd)
                dzkw = dict(dzkw, ws=blk["ln_slab"], ws_floats=blk["ln_slab"].numel(), defer_fold=C.addressof(fd))
            wgf = bool(blk.get("wgf")) and dzr
            if wgf:      # U = dh^T x-hat, db1 per persistent workgroup into the block's own slab (<= 2 workgroups per CU); dh is not stored
                if "wg_slab" not in blk:
                    cus = torch.cuda.get_device_properties(self.device).multi_processor_count if self.device.type == "cuda" else 256
                    blk["wg_slab"] = torch.empty(2 * cus * (H * Cc + H), dtype=torch.float32, device=self.device)
                    blk["wg_rows"] = C.c_int(0)
                dzkw = dict(dzkw, wg_ws=blk["wg_slab"], wg_ws_floats=blk["wg_slab"].numel(), wg_rows=C.addressof(blk["wg_rows"]))
            self._rs(lst, tag + ":grn.bapply+pw1.dgrad+ln.bwd" + ("+pw1.wgrad" if wgf else ""), 5, blk,
                     ((1 if wgf else 2 if dzr else 3) * M * H + (3 if dzr else 2) * M * Cc) * esz,
                     (4 if dzr else 2) * M * Cc * H,
                     A=dz, A2=blk["h"], W=w1t["t"], ldw=w1t["ld"], v0=blk["scale"], v1=blk["coef"], out=dd,
                     xhat=blk["dhat"], rstd=blk["rstd"], lng=P[nm["ln_w"]], act=act, s0=Gd[nm["ln_w"]],
                     s1=Gd[nm["ln_b"]], rpg=0, **dzkw,
                     **(dict(fin_sum=blk["S1"], fin_sum0=blk["S0"], fin_gamma=P[nm["gg"]], fin_gx=blk["Gx"],
                             fin_ainv=blk["Ainv"], fin_out=blk["coef"], fin_dgamma=Gd[nm["gg"]],
                             fin_dbeta=Gd[nm["gb"]]) if fold else {}))
        elif self._mx_block(blk) or self._mx_sparse(blk):
            qd, qw = self._mx_buf(tag + ".dh", M, H), self._mx_weight(tag + ".W1T")
            self._quant(lst, tag + ":dh.quant", dz, H, qd)
            self._gemm_mx(lst, tag + ":pw1.dgrad", "STORE", qd, qw, C=dxn, M=M, N=Cc, K=H, ldc=Cc)
        else:
            self._gemm(lst, tag + ":pw1.dgrad", "NONE", "STORE", A=dz, B=w1t["t"], C=dxn, M=M, N=Cc, K=H, lda=H,
                       ldb=w1t["ld"], ldc=Cc)
        self._guard(lst, dd)
        if late_w2 and not grouped and not sw:
            self._side_wgrad(lst, tag + ":pw2.wgrad", "NONE", w2_qpro, [dout], **w2_args)
        w1_args = dict(P=dz, Q=blk["xn"], M=M, Nn=H, Kk=Cc, ldp=H, ldq=Cc, dW=Gd[nm["w1"]], sn=Cc, sk=1, db=Gd[nm["b1"]])
        if rsc and bool(blk.get("wgf")) and dzr:
            # second stage of the weight gradient the fused kernel accumulated (LayerNorm affine applied by linearity): nothing on the chain reads it
            def wfold(stream, _b=blk, _c=Cc, _h=H, _g=P[nm["ln_w"]], _bt=P[nm["ln_b"]], _dw=Gd[nm["w1"]], _db=Gd[nm["b1"]]):
                return lib.mpmae_rs_wgrad_fold(_h, _c, _p(_b["wg_slab"]), _b["wg_rows"].value, _p(_g), _p(_bt), _p(_dw), _p(_db), stream)
            if self.lanes and not (self._tail_main() >= 1 and tag == "encoder.stages.0.0"):
                k = self._after(lst)
                self._evseq += 1
                self._op(lst, tag + ":pw1.wgrad.fold", wfold, kind="rs_wgrad_fold", nbytes=blk["wg_slab"].numel() * 4, lane=1, wait=(k,) if k else (),
                         signal=f"s{self._evseq}")
            else:      # (single lane, or the last block of the backward: in order on the main lane like its depthwise weight gradient - tail_main)
                self._op(lst, tag + ":pw1.wgrad.fold", wfold, kind="rs_wgrad_fold", nbytes=blk["wg_slab"].numel() * 4)
        elif grouped or (sw and self._group_ok(blk, "NONE")):
            self._group_add(lst, tag + ":pw1.wgrad", [dz], **w1_args)
        elif not late_all:
            if self.lanes and self._tail_main() >= 2 and tag == "encoder.stages.0.0":
                self._wgrad(lst, tag + ":pw1.wgrad", "NONE", "NONE", **w1_args)      # (tail_main = 2: in order on the main lane)
            else:
                self._side_wgrad(lst, tag + ":pw1.wgrad", "NONE", "NONE", [dz], **w1_args)
        if rs_n is None:
            self._op(lst, tag + ":ln.bwd", self._ln_bwd_callable(Cc), dt, _p(dxn), 1, 1.0, _p(blk["dhat"]), _p(blk["rstd"]),
                     _p(P[nm["ln_w"]]), _p(P[nm["ln_b"]]), 0, _p(dd), 0, _p(Gd[nm["ln_w"]]), _p(Gd[nm["ln_b"]]), M, Cc,
                     _p(act), kind="ln_bwd", nbytes=3 * M * Cc * esz)
            self._guard(lst, dd)
        if late_all and not grouped and not sw:
            self._side_wgrad(lst, tag + ":pw2.wgrad", "NONE", w2_qpro, [dout], **w2_args)
            self._side_wgrad(lst, tag + ":pw1.wgrad", "NONE", "NONE", [dz], **w1_args)
        self._dw_bwd(lst, blk, dd, dout, dx)

    def _dw_bwd(self, lst, blk, dd, dout, dx):
        """depthwise conv backward: weight/bias gradient, then data gradient (+ residual dout)."""
        self._dw_wgrad(lst, blk, dd)
        self._dwconv(lst, blk["prefix"] + ":dw.dgrad", blk, dd, dx, dout, 1, False)
        self._guard(lst, dx)

    def _dw_wgrad(self, lst, blk, dd):
        lib, dt = self.lib, self.dt
        M, Cc = blk["M"], blk["C"]
        act = self.act[blk["stage"]] if blk["sparse"] else None
        tag = blk["prefix"]
        w, gw, b, gb, (skh, skw, sc) = self._dw_weight(blk)
        TP, ts, CC = self._dw_tiling(blk["stage"], Cc)
        a = _lib.DwWgArgs()
        a.x, a.dd, a.dw, a.db = blk["x"].data_ptr(), dd.data_ptr(), gw.data_ptr(), gb.data_ptr()
        a.s_kh, a.s_kw, a.s_c = skh, skw, sc
        a.g = self._geom(blk["stage"])
        a.C, a.CC, a.TP, a.tiles_side = Cc, CC, TP, ts
        a.ntiles_total = self.N * ts * ts
        a.act = act.data_ptr() if act is not None else 0
        a.ws, a.ws_floats = (self.ws3 if self.lanes else self.ws).data_ptr(), self.ws_floats
        self._keepalive.append(a)
        if self.lanes and self._tail_main() >= 1 and tag == "encoder.stages.0.0":
            # in order on the main lane right behind the block's data gradient (its operands are fresh: no event, no scratch-ring guard)
            a.ws = self.ws.data_ptr()              # main-lane scratch
            self._op(lst, tag + ":dw.wgrad", lib.mpmae_dwconv7_wgrad, dt, C.byref(a), 2048, kind="dwconv7_wgrad",
                     nbytes=2 * M * Cc * (4 if dt == F32 else 2), flops=2 * 49 * M * Cc)
            return
        if (self.lanes and dt == BF16 and blk["sparse"] and blk["stage"] >= int(self.opt["dw_group"]) and self.cfg.depths[blk["stage"]] > 1):
            if not hasattr(self, "_dwg_pending"):
                self._dwg_pending = []
            self._dwg_pending.append((tag, a, dd, 2 * M * Cc * 2, 2 * 49 * M * Cc))
            if len(self._dwg_pending) >= min(_lib.DWG_MAX, max(1, len(self.scr_dd2) - 2)):
                self._dwg_flush(lst)
            return
        if self.lanes:
            k = self._after(lst)
            self._evseq += 1
            key = f"s{self._evseq}"
            self._op(lst, tag + ":dw.wgrad", lib.mpmae_dwconv7_wgrad, dt, C.byref(a), 2048, kind="dwconv7_wgrad",
                     nbytes=2 * M * Cc * (4 if dt == F32 else 2), flops=2 * 49 * M * Cc, lane=1, wait=(k,), signal=key)
            self._side_read(key, dd)
        else:
            self._op(lst, tag + ":dw.wgrad", lib.mpmae_dwconv7_wgrad, dt, C.byref(a), 2048, kind="dwconv7_wgrad",
                     nbytes=2 * M * Cc * (4 if dt == F32 else 2), flops=2 * 49 * M * Cc)

    def _dwg_flush(self, lst):
        pend = getattr(self, "_dwg_pending", [])
        if not pend:
            return
        self._dwg_pending = []
        arr = (_lib.DwWgArgs * len(pend))(*[p_[1] for p_ in pend])
        self._keepalive.append(arr)
        stage = pend[0][0].rsplit(".", 1)[0]
        k = self._after(lst)
        self._evseq += 1
        key = f"s{self._evseq}"
        self._op(lst, f"{stage}:dw.wgrad[{len(pend)}]", self.lib.mpmae_dwconv7_wgrad_group, self.dt, arr, len(pend), _p(self.ws3), self.ws_floats,
                 kind="dwconv7_wgrad_group", nbytes=sum(p_[3] for p_ in pend), flops=sum(p_[4] for p_ in pend), lane=1,
                 wait=(k,) if k else (), signal=key)
        self._side_read(key, *[p_[2] for p_ in pend])

    def _block_fwd_fused(self, lst, blk, x):
        P, lib, dt = self.params, self.lib, self.dt
        nm = self._block_names(blk)
        M, Cc, H, G = blk["M"], blk["C"], blk["H"], blk["G"]
        act = self.act[blk["stage"]] if blk["sparse"] else None
        rpg = blk["rpg"]
        eps = 1e-6 if blk["sparse"] else 1e-4
        tag = blk["prefix"]
        blk["x"] = x
        self._dwconv(lst, tag + ":dw", blk, x, blk["d"], None, 0, True)
        self._op(lst, tag + ":ln", lib.mpmae_ln_fwd, dt, _p(blk["d"]), _p(blk["dhat"]), _p(blk["rstd"]), None,
                 None, None, 0, 1e-6, M, Cc, _p(act))
        self._gemm(lst, tag + ":pw1", "LN_AFFINE", "GELU_SUMSQ", A=blk["dhat"], B=self.w[tag + ".W1"]["t"],
                   bias=P[nm["b1"]], C=blk["h"], M=M, N=H, K=Cc, lda=Cc, ldb=self.w[tag + ".W1"]["ld"], ldc=H,
                   p0=P[nm["ln_w"]], p1=P[nm["ln_b"]], rpg=rpg, s0=blk["G2"], act=act)
        self._op(lst, tag + ":grn", lib.mpmae_grn_fwd_finalize, _p(blk["G2"]), _p(P[nm["gg"]]), eps, G, H,
                 _p(blk["Gx"]), _p(blk["Ainv"]), _p(blk["scale"]))
        self._gemm(lst, tag + ":pw2", "GRN", "RESID", A=blk["h"], B=self.w[tag + ".W2"]["t"], bias=P[nm["b2"]],
                   C=blk["out"], R=x, M=M, N=Cc, K=H, lda=H, ldb=self.w[tag + ".W2"]["ld"], ldc=Cc, ldr=Cc,
                   p0=blk["scale"], p1=P[nm["gb"]], rpg=rpg, act=act)
        return blk["out"]

    def _build_forward(self):
        cfg, P, lib, dt, N, L, D = self.cfg, self.params, self.lib, self.dt, self.N, self.L, self.D
        f = self.fwd_ops
        dims = cfg.dims
        C0, p, k = dims[0], self.p, cfg.stem_k
        orig = self.orig_stem = bool(getattr(cfg, "use_orig_stem", False))
        # weight staging only feeds the first GEMM: on the side lane next to mask / activity / im2col (which only read the inputs)
        prep_side = self.lanes and bool(self.opt["prep_side"])
        # (prep_late: issued behind the activity ops instead, see below)
        prep_late = (prep_side and bool(self.opt["prep_late"]) and bool(self.opt["front_side"]) and self.track_activity
                     and bool(self.opt["stem_front"]) and bool(self.opt["stem_fused"]) and bool(self.opt["stem_im2col"]) and dt != F32 and p == 8
                     and k == 1 and C0 % 8 == 0 and C0 <= 48 and cfg.in_chans <= 12 and not orig)      # (= the conditions of the fused stem kernel below)
        if not prep_late:
            self._op(f, "prep", lib.mpmae_prep_weights, dt, _p(self.prep_table), self.prep_n, self.prep_max,
                     **(dict(lane=1, signal="prep_done") if prep_side else {}))
        # the pixel-activity map and its poolings also only read the inputs (and the mask tables): with `front_side` they follow the weight
        # staging on the side lane, so the main lane goes mask -> im2col directly and the stem GEMM waits for ONE side-lane event
        front_side = prep_side and bool(self.opt["front_side"]) and self.track_activity
        if self.dense:
            self._op(f, "mask", lib.mpmae_mask_gen_dense, _p(self.noise), N, L, self.keep_mask, _p(self.mask), _p(self.inv))
        else:
            self._op(f, "mask", lib.mpmae_mask_gen, _p(self.noise), N, L, self.keep, _p(self.mask), _p(self.vis), _p(self.inv),
                     **(dict(signal="mask_done") if front_side else {}))
        img = self.inp["sentinel2"]
        # act_in_stem: the fused stem kernel (below) writes act_full; the poolings are issued behind it
        self._act_in_stem = (front_side and prep_late and bool(self.opt["act_in_stem"]) and self.track_activity and k == 1)
        if self.track_activity and not self._act_in_stem:
            fl = dict(lane=1) if front_side else {}
            self._op(f, "act0", lib.mpmae_activity, _p(img), _p(self.vis), _p(self.act_full), N, cfg.in_chans,
                     cfg.img_size, self.keep, self.grid, p, **(dict(lane=1, wait=("mask_done",)) if front_side else {}))
            if k > 1:
                self._op(f, "actpool_stem", lib.mpmae_activity_pool, _p(self.act_full), _p(self.act[0]), self.M[0], 8, k, **fl)
            for i in range(1, 4):
                self._op(f, f"actpool{i}", lib.mpmae_activity_pool, _p(self.act[i - 1]), _p(self.act[i]), self.M[i], self.S[i], 2, **fl)
            if front_side:
                f[-1][3]["signal"] = "front_done"
        if prep_late:
            self._op(f, "prep", lib.mpmae_prep_weights, dt, _p(self.prep_table), self.prep_n, self.prep_max, lane=1, signal="prep_done")
        wt = self.w["stem.Wt"]
        self.stem_im2col = bool(self.opt["stem_im2col"]) or orig
        self.stem_fused = (k == 1 and C0 % 8 == 0 and bool(self.opt["stem_fused"])) and not orig
        # one launch for the whole stem forward (bf16, patch 8): the convolution output never exists, and the im2col matrix of the weight
        # gradient is written from the kernel's own MFMA operand fragments (no mpmae_im2col3 launch at all)
        self.stem_front = (self.stem_fused and self.stem_im2col and dt != F32 and bool(self.opt["stem_front"]) and p == 8
                           and cfg.in_chans <= 12 and C0 <= 48 and wt["ld"] % 8 == 0)
        if orig:
            # use_orig_stem (convnextv2_sparse.py:99-110,202-203): ONE convolution k = s = patch / 8 + LN. The k x k pixels under every stage-0
            # point are gathered into an operand matrix once per step (mpmae_gather_kxk); the convolution is a plain GEMM with the pooled
            # activity map as row mask (bias only at active outputs), its weight gradient a plain TN product on the same matrix
            self.ldk = _rup(k * k * cfg.in_chans, 8)
            self.col = self._t(self.M[0] * self.ldk)
            self._op(f, "stem:gather", lib.mpmae_gather_kxk, dt, _p(img), None if self.dense else _p(self.vis), _p(self.inv) if self.dense else None, _p(self.col), self.ldk, N, self.keep,
                     self.grid, p, k, cfg.in_chans, cfg.img_size, kind="gather_kxk",
                     nbytes=self.M[0] * self.ldk * (4 if dt == F32 else 2) + N * self.keep * p * p * cfg.in_chans * 4)
            self._gemm(f, "stem:conv", "NONE", "STORE", A=self.col, B=wt["t"], bias=P["encoder.stem_orig.0.bias"], C=self.s0, M=self.M[0], N=C0,
                       K=self.ldk, lda=self.ldk, ldb=wt["ld"], ldc=C0, act=self.act[0])
        elif self.stem_im2col:     # materialise the 3x3 taps once per step: plain (fast) GEMMs forward and for the weight gradient
            self.ldk = _rup(9 * cfg.in_chans, 8)
            self.col = self._t(self.Mfull * self.ldk)
            if not self.stem_front:
                self._op(f, "stem:im2col", lib.mpmae_im2col3, dt, _p(img), _p(self.vis), _p(self.inv), _p(self.col), self.ldk,
                         N, self.keep, self.grid, p, cfg.in_chans, cfg.img_size, kind="im2col3",
                         nbytes=self.Mfull * self.ldk * (4 if dt == F32 else 2) + img.numel() * 4)
            if not self.stem_front:
                self._gemm(f, "stem:conv", "NONE", "STORE", A=self.col, B=wt["t"], bias=P["encoder.initial_conv.0.bias"],
                           C=self.c1, M=self.Mfull, N=C0, K=self.ldk, lda=self.ldk, ldb=wt["ld"], ldc=C0, act=self.act_full)
        else:
            self._gemm(f, "stem:conv", "IM2COL3", "STORE", A=img, B=wt["t"], bias=P["encoder.initial_conv.0.bias"],
                       C=self.c1, M=self.Mfull, N=C0, K=9 * cfg.in_chans, lda=0, ldb=wt["ld"], ldc=C0,
                       vis=self.vis, inv=self.inv, act=self.act_full, keep=self.keep, L=L, S=p, Cseg=cfg.in_chans,
                       grid=self.grid, H=cfg.img_size)
        if self.stem_front:
            a = _lib.StemFrontArgs()
            a.img, a.vis, a.inv = img.data_ptr(), self.vis.data_ptr(), self.inv.data_ptr()
            # the fp32 parameter itself (ME layout [9 Cin][C0], the k order of the im2col matrix): the kernel rounds it to bf16 as the staging
            # does, so the launch waits for nothing on the side lane
            a.W, a.ldw, a.W_master = 0, 0, P["encoder.initial_conv.0.kernel"].data_ptr()
            a.bias = P["encoder.initial_conv.0.bias"].data_ptr()
            a.xhat1, a.rstd1, a.xhat2, a.rstd2 = (t.data_ptr() for t in (self.c1hat, self.rstd1, self.s0hat, self.rstd2))
            a.out = self.x0.data_ptr()
            a.g1, a.b1 = P["encoder.initial_conv.1.ln.weight"].data_ptr(), P["encoder.initial_conv.1.ln.bias"].data_ptr()
            a.w, a.wb = P["encoder.stem.0.kernel"].data_ptr(), P["encoder.stem.0.bias"].data_ptr()
            a.g2, a.b2 = P["encoder.stem.1.ln.weight"].data_ptr(), P["encoder.stem.1.ln.bias"].data_ptr()
            a.N, a.keep, a.grid, a.H, a.Cin, a.C0 = N, self.keep, self.grid, cfg.img_size, cfg.in_chans, C0
            a.track_activity = 1 if self.track_activity else 0
            a.col, a.ldc = self.col.data_ptr(), self.ldk          # the weight gradient's im2col matrix, from the kernel's own A fragments
            a.act_out = self.act_full.data_ptr() if getattr(self, "_act_in_stem", False) else 0
            self._keepalive.append(a)
            self._op(f, "stem:conv+ln+gelu+dw+ln", lib.mpmae_stem_front, C.byref(a), kind="stem_front",
                     nbytes=3 * self.Mfull * C0 * 2 + self.Mfull * self.ldk * 2 + N * self.keep * 100 * cfg.in_chans * 4,
                     flops=2 * self.Mfull * C0 * 9 * cfg.in_chans)
        if getattr(self, "_act_in_stem", False):
            assert self.stem_front
            f[-1][3]["signal"] = "stem_front_done"
            for i in range(1, 4):
                self._op(f, f"actpool{i}", lib.mpmae_activity_pool, _p(self.act[i - 1]), _p(self.act[i]), self.M[i], self.S[i], 2, lane=1,
                         wait=("stem_front_done",) if i == 1 else ())
            f[-1][3]["signal"] = "front_done"
        if prep_side:
            if self.stem_front:       # the fused stem kernel reads the fp32 parameter itself; whatever follows it waits for the side-lane front
                stem_at, rest_key = len(f) - 1 - (3 if getattr(self, "_act_in_stem", False) else 0), ("front_done" if front_side else "prep_done")
            else:
                f[-1][3]["wait"] = tuple(f[-1][3]["wait"]) + (("front_done",) if front_side else ("prep_done",))
        if self.stem_front:
            pass
        elif orig:
            self._op(f, "stem:ln", lib.mpmae_ln_fwd, dt, _p(self.s0), _p(self.s0hat), _p(self.rstd2), _p(self.x0),
                     _p(P["encoder.stem_orig.1.ln.weight"]), _p(P["encoder.stem_orig.1.ln.bias"]), 0, 1e-6, self.M[0], C0, _p(self.act[0]))
        elif self.stem_fused:      # LN + GELU + 1x1 depthwise + LN in one row-wise pass (stemtail.cuh)
            a = _lib.StemTailArgs()
            a.x, a.out = self.c1.data_ptr(), self.x0.data_ptr()
            a.xhat1, a.rstd1, a.xhat2, a.rstd2 = (t.data_ptr() for t in (self.c1hat, self.rstd1, self.s0hat, self.rstd2))
            a.g1, a.b1 = P["encoder.initial_conv.1.ln.weight"].data_ptr(), P["encoder.initial_conv.1.ln.bias"].data_ptr()
            a.w, a.wb = P["encoder.stem.0.kernel"].data_ptr(), P["encoder.stem.0.bias"].data_ptr()
            a.g2, a.b2 = P["encoder.stem.1.ln.weight"].data_ptr(), P["encoder.stem.1.ln.bias"].data_ptr()
            a.act_in = self.act_full.data_ptr() if self.act_full is not None else 0
            a.act_out = self.act[0].data_ptr() if self.act[0] is not None else 0
            a.M, a.C = self.Mfull, C0
            self._keepalive.append(a)
            esz = 4 if dt == F32 else 2
            self._op(f, "stem:ln+gelu+dw+ln", lib.mpmae_stem_tail, dt, 0, C.byref(a), kind="stem_tail_fwd",
                     nbytes=4 * self.Mfull * C0 * esz)
        else:
          self._op(f, "stem:ln1", lib.mpmae_ln_fwd, dt, _p(self.c1), _p(self.c1hat), _p(self.rstd1), _p(self.a1),
                 _p(P["encoder.initial_conv.1.ln.weight"]), _p(P["encoder.initial_conv.1.ln.bias"]), 1, 1e-6,
                 self.Mfull, C0, _p(self.act_full))
          self._op(f, "stem:dw", lib.mpmae_dwstride_fwd, dt, _p(self.a1), _p(self.s0), _p(P["encoder.stem.0.kernel"]),
                 _p(P["encoder.stem.0.bias"]), self.M[0], C0, 8, k, _p(self.act_full), _p(self.act[0]))
          self._op(f, "stem:ln2", lib.mpmae_ln_fwd, dt, _p(self.s0), _p(self.s0hat), _p(self.rstd2), _p(self.x0),
                 _p(P["encoder.stem.1.ln.weight"]), _p(P["encoder.stem.1.ln.bias"]), 0, 1e-6, self.M[0], C0,
                 _p(self.act[0]))
        x = self.x0
        bi = 0
        self._front_rest = (stem_at, rest_key) if (prep_side and self.stem_front) else None
        self._prep_late = prep_late and self._front_rest is not None
        assert not prep_late or self._front_rest is not None
        for i in range(4):
            if i > 0:
                dn = self.down[i - 1]
                pre = f"encoder.downsample_layers.{i - 1}"
                dn["x"] = x
                Ci = dims[i - 1]
                wd = self.w[f"down{i - 1}.Wt"]
                dn["grouped"] = (self.down_grouped and Ci % 8 == 0 and Ci <= 1024 and self.S[i - 1] % 2 == 0)      # (the grouped LayerNorm kernels: C <= 1024; huge has 1408 in front of stage 3)
                # down_fused: the producer of x - the last block's [GRN + pwconv2 + residual] kernel - also does this LayerNorm (its args are patched here)
                last = f[-1]
                fuse = (dn["grouped"] and bool(self.opt["down_fused"]) and dt == BF16 and last[0].endswith(":grn.apply+pw2") and last[3]["kind"] == "rs<4>"
                        and Ci <= 96 and lib.mpmae_get_option(_lib.OPT["RSP_NARROW"]) & (4 if Ci == 80 else 1) == 0)
                dn["fused"] = fuse
                if fuse:
                    if "yg" not in dn:
                        dn["yg"] = self._t(self.M[i] * 4 * Ci)
                    ra = last[2][1]._obj
                    ra.dn_xhat, ra.dn_rstd, ra.dn_y = dn["xhat"].data_ptr(), dn["rstd"].data_ptr(), dn["yg"].data_ptr()
                    ra.dn_gamma, ra.dn_beta, ra.dn_S = P[pre + ".0.ln.weight"].data_ptr(), P[pre + ".0.ln.bias"].data_ptr(), self.S[i - 1]
                    ra.out = 0                      # nothing else reads the stage output
                    f[-1] = (last[0] + "+down.ln", last[1], last[2], dict(last[3], bytes=last[3]["bytes"] + 2 * self.M[i - 1] * Ci * 2))
                    self._gemm(f, pre + ":conv", "NONE", "STORE", A=dn["yg"], B=wd["t"], bias=P[pre + ".1.bias"], C=dn["out"],
                               M=self.M[i], N=dims[i], K=4 * Ci, lda=4 * Ci, ldb=wd["ld"], ldc=dims[i], act=self.act[i])
                elif dn["grouped"]:
                    # LN writes its affine output straight into the [M_i][4*Ci] operand layout of the 2x2/2 convolution,
                    # which then is a plain GEMM (and its weight gradient a plain TN product)
                    if "yg" not in dn:
                        dn["yg"] = self._t(self.M[i] * 4 * Ci)
                    self._op(f, pre + ":ln", lib.mpmae_ln_fwd_down, dt, _p(x), _p(dn["xhat"]), _p(dn["rstd"]), _p(dn["yg"]),
                             _p(P[pre + ".0.ln.weight"]), _p(P[pre + ".0.ln.bias"]), 1e-6, self.M[i - 1], Ci, self.S[i - 1],
                             _p(self.act[i - 1]), kind="ln_fwd_down", nbytes=3 * self.M[i - 1] * Ci * (4 if dt == F32 else 2))
                    self._gemm(f, pre + ":conv", "NONE", "STORE", A=dn["yg"], B=wd["t"], bias=P[pre + ".1.bias"], C=dn["out"],
                               M=self.M[i], N=dims[i], K=4 * Ci, lda=4 * Ci, ldb=wd["ld"], ldc=dims[i], act=self.act[i])
                else:
                    self._op(f, pre + ":ln", lib.mpmae_ln_fwd, dt, _p(x), _p(dn["xhat"]), _p(dn["rstd"]), None, None, None,
                             0, 1e-6, self.M[i - 1], dims[i - 1], _p(self.act[i - 1]))
                    self._gemm(f, pre + ":conv", "DOWN_GATHER", "STORE", A=dn["xhat"], B=wd["t"], bias=P[pre + ".1.bias"],
                               C=dn["out"], M=self.M[i], N=dims[i], K=4 * dims[i - 1], lda=dims[i - 1], ldb=wd["ld"],
                               ldc=dims[i], p0=P[pre + ".0.ln.weight"], p1=P[pre + ".0.ln.bias"], S=self.S[i],
                               Cseg=dims[i - 1], act=self.act[i], act_src=self.act[i - 1])
                x = dn["out"]
            if self._ps_ok(i):
                x = self._stage_fwd_ps(f, i, self.blocks[bi:bi + cfg.depths[i]], x)
                bi += cfg.depths[i]
                continue
            for j in range(cfg.depths[i]):
                x = self._block_fwd(f, self.blocks[bi], x)
                bi += 1
        self.enc_out = x
        wp = self.w["proj.W"]
        self.proj_compact = bool(self.opt["proj_compact"]) and D % 8 == 0
        if self.proj_compact:
            # proj on the COMPACT rows through the plain (fast) NT GEMM; the token kernel then writes the whole decoder input in one pass
            self.proj_rows = self._t(self.M[3], D)
            self._gemm(f, "proj", "NONE", "STORE", A=x, B=wp["t"], bias=P["proj.bias"], C=self.proj_rows, M=self.M[3],
                       N=D, K=dims[3], lda=dims[3], ldb=wp["ld"], ldc=D)
            self._op(f, "mask_token", lib.mpmae_fill_mask_token, dt, _p(self.xdec), _p(P["mask_token"]), _p(self.inv), N * L, D,
                     _p(self.proj_rows), self.keep, L)
        else:
            self._gemm(f, "proj", "NONE", "SCATTER_ROWS", A=x, B=wp["t"], bias=P["proj.bias"], C=self.xdec, M=self.M[3],
                       N=D, K=dims[3], lda=dims[3], ldb=wp["ld"], ldc=D, vis=self.vis, keep=self.keep, L=L)
            self._op(f, "mask_token", lib.mpmae_fill_mask_token, dt, _p(self.xdec), _p(P["mask_token"]), _p(self.inv), N * L, D,
                     None, 0, 0)
        y = self.xdec
        for d_ in self.decs:
            y = self._block_fwd(f, d_, y)
        self.dec_out = y
        # heads
        coff = 0
        self.head_cols = {}
        if self.heads_merged.get("pix"):
            wh = self.w["head.pix.W"]
            self._gemm(f, "head:pix", "NONE", "STORE", A=y, B=wh["t"], bias=P[f"pred_dict.{cfg.pix_mods[0].name}.bias"],
                       C=self.pred_pix, M=N * L, N=self.Wpix, K=D, lda=D, ldb=wh["ld"], ldc=self.pred_pix.shape[1])
        for om in cfg.pix_mods:
            if not self.heads_merged.get("pix"):
                wh = self.w[f"head.{om.name}.W"]
                cview = self.pred_pix.view(-1)[coff:]
                self._gemm(f, f"head:{om.name}", "NONE", "STORE", A=y, B=wh["t"], bias=P[f"pred_dict.{om.name}.bias"],
                           C=cview, M=N * L, N=om.head_out, K=D, lda=D, ldb=wh["ld"], ldc=self.pred_pix.shape[1])
            self.head_cols[om.name] = coff
            coff += om.head_out
        if cfg.img_mods:
            self._op(f, "head:ln", lib.mpmae_ln_fwd, dt, _p(y), _p(self.yhat), _p(self.rstd_y), _p(self.yln),
                     _p(P["layer_norm_tmp.weight"]), _p(P["layer_norm_tmp.bias"]), 0, 1e-6, N * L, D, None)
            self._op(f, "head:pool", lib.mpmae_pool_rows, dt, _p(self.yln), _p(self.pooled), N, L, D)
            coff = 0
            if self.heads_merged.get("img"):
                wh = self.w["head.img.W"]
                self._gemm(f, "head:img", "NONE", "STORE", A=self.pooled, B=wh["t"],
                           bias=P[f"pred_dict.{cfg.img_mods[0].name}.bias"], C=self.pred_img, M=N, N=self.Wimg, K=D, lda=D,
                           ldb=wh["ld"], ldc=self.ldimg)
            for om in cfg.img_mods:
                if not self.heads_merged.get("img"):
                    wh = self.w[f"head.{om.name}.W"]
                    cview = self.pred_img.view(-1)[coff:]
                    self._gemm(f, f"head:{om.name}", "NONE", "STORE", A=self.pooled, B=wh["t"],
                               bias=P[f"pred_dict.{om.name}.bias"], C=cview, M=N, N=om.head_out, K=D, lda=D,
                               ldb=wh["ld"], ldc=self.ldimg)
                self.head_cols[om.name] = coff
                coff += om.head_out
        # losses
        self.loss_args = {}
        ipc = 0
        for t, om in enumerate(cfg.out_mods):
            acc = self.loss_acc[t]
            coef = self.coef[t:t + 1]
            tgt = self.inp[om.name]
            if om.kind == "pix_cont":
                a = _lib.PixContArgs()
                a.pred, a.dpred = self.pred_pix.data_ptr(), self.dpred_pix.data_ptr()
                a.ld, a.coff = self.pred_pix.shape[1], self.head_cols[om.name]
                a.target, a.mask = tgt.data_ptr(), self.mask.data_ptr()
                a.C, a.p, a.grid, a.H, a.L = om.chans, self.p, self.grid, cfg.img_size, L
                a.norm_pix = 1 if (cfg.norm_pix_loss and om.name == "sentinel2") else 0
                a.acc = acc.data_ptr()
                pb = self.patch_buf[ipc]
                ipc += 1
                a.patch_l, a.patch_cnt, a.patch_mean, a.patch_rstd = (pb[i].data_ptr() for i in range(4))
                a.coef = coef.data_ptr()
                self._keepalive.append(a)
                self.loss_args[om.name] = a
                self._op(f, f"loss:{om.name}", lib.mpmae_loss_pix_cont, dt, 0, C.byref(a), N * L)
            elif om.kind == "pix_cat":
                a = _lib.PixCatArgs()
                a.pred, a.dpred = self.pred_pix.data_ptr(), self.dpred_pix.data_ptr()
                a.ld, a.coff = self.pred_pix.shape[1], self.head_cols[om.name]
                a.target, a.mask = tgt.data_ptr(), self.mask.data_ptr()
                a.K, a.p, a.grid, a.H, a.L = om.chans, self.p, self.grid, cfg.img_size, L
                a.acc, a.coef = acc.data_ptr(), coef.data_ptr()
                self._keepalive.append(a)
                self.loss_args[om.name] = a
                self._op(f, f"loss:{om.name}", lib.mpmae_loss_pix_cat, dt, 0, C.byref(a), N * L)
            else:
                a = _lib.ImgArgs()
                a.pred, a.dpred = self.pred_img.data_ptr(), self.dpred_img.data_ptr()
                a.ld, a.coff = self.ldimg, self.head_cols[om.name]
                a.target = tgt.data_ptr()
                a.K, a.N, a.kind = om.chans, N, (0 if om.kind == "img_cat" else 1)
                a.acc, a.coef = acc.data_ptr(), coef.data_ptr()
                self._keepalive.append(a)
                self.loss_args[om.name] = a
                self._op(f, f"loss:{om.name}", lib.mpmae_loss_img, dt, 0, C.byref(a))
        # one launch per loss KIND instead of one per modality (12 small latency-bound kernels -> 3)
        self.loss_multi = bool(self.opt["loss_multi"])
        self.loss_onepass = False
        if self.loss_multi:
            while f and f[-1][0].startswith("loss:"):
                f.pop()
            self._loss_tabs = {}
            # one-pass pixel losses: bf16, both pixel kinds on their row-band / wave kernels (conditions repeated from below), merged heads
            # (one data-gradient GEMM over all pixel heads, one contiguous weight gradient: the scalars become column / row scales)
            cont_m = [om for om in cfg.out_mods if om.kind == "pix_cont"]
            cat_m = [om for om in cfg.out_mods if om.kind == "pix_cat"]
            mc_ = max([om.chans for om in cont_m], default=0)
            mk_ = max([om.chans for om in cat_m], default=0)
            ldp0 = self.pred_pix.shape[1] if cfg.pix_mods else 0
            cont_ok = (not cont_m) or (bool(self.opt["loss_rows"]) and cfg.img_size % 4 == 0 and (self.p * self.p) % 4 == 0
                                       and mc_ * (self.p * cfg.img_size + 4) * 4 <= 150 * 1024
                                       and -(-(mc_ * self.p * (cfg.img_size // 4)) // 512) <= 12 and -(-(mc_ * self.p * self.p // 4) // 64) <= 12)
            cat_ok = (not cat_m) or (bool(self.opt["loss_rows"]) and mk_ <= 16 and ldp0 % 4 == 0
                                     and all(self.head_cols[om.name] % 4 == 0 for om in cat_m) and (self.p * self.p * mk_) % 4 == 0
                                     and 16 * self.p * self.p * mk_ * 4 <= 150 * 1024)
            onepass = self.loss_onepass = (bool(self.opt["loss_onepass"]) and dt == BF16 and bool(cfg.pix_mods) and cont_ok and cat_ok
                                           and bool(self.heads_merged.get("pix")) and self.D % 8 == 0
                                           and bool(self.opt["loss_rows_bwd"]))
            for kind_id, kind, typ in ((0, "pix_cont", _lib.PixContArgs), (1, "pix_cat", _lib.PixCatArgs), (2, "img", None)):
                mods = [om for om in cfg.out_mods if (om.kind == kind if typ else om.kind.startswith("img"))]
                if not mods:
                    continue
                typ = typ or _lib.ImgArgs
                arr = (typ * len(mods))(*[self.loss_args[om.name] for om in mods])
                tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
                self._loss_tabs[kind] = (kind_id, tab, len(mods))
                maxc = max(om.chans for om in mods)
                if (kind == "pix_cont" and bool(self.opt["loss_rows"]) and cfg.img_size % 4 == 0 and (self.p * self.p) % 4 == 0
                        and maxc * (self.p * cfg.img_size + 4) * 4 <= 150 * 1024
                        # the kernel's per-thread vector counts (loss_pix_cont_rows_impl: mv, mp <= 12), else mpmae_loss_multi
                        and -(-(maxc * self.p * (cfg.img_size // 4)) // 512) <= 12 and -(-(maxc * self.p * self.p // 4) // 64) <= 12):
                    # row-band forward: a workgroup per sample walks its patch rows with the target band in LDS (loss.cuh)
                    self._cont_rows = maxc
                    self._op(f, f"loss:{kind}[{len(mods)}]", lib.mpmae_loss_pix_cont_rows_fused if onepass else lib.mpmae_loss_pix_cont_rows,
                             dt, _p(tab), len(mods), N, maxc, self.p, cfg.img_size, kind=f"loss_{kind}_fwd")
                    continue
                ldp_ = self.pred_pix.shape[1] if cfg.pix_mods else 0
                cat_waves = (kind == "pix_cat" and bool(self.opt["loss_rows"]) and maxc <= 16 and ldp_ % 4 == 0
                             and all(self.head_cols[om.name] % 4 == 0 for om in mods) and (self.p * self.p * maxc) % 4 == 0
                             and 16 * self.p * self.p * maxc * 4 <= 150 * 1024)
                if kind == "pix_cat":
                    self._cat_waves = cat_waves
                if cat_waves:      # wave per patch, logits staged through LDS with contiguous vector accesses (loss.cuh)
                    self._op(f, f"loss:{kind}[{len(mods)}]", lib.mpmae_loss_pix_cat_waves, dt, 2 if onepass else 0, _p(tab), len(mods), N,
                             self.p * self.p * maxc, kind=f"loss_{kind}_fwd")
                    continue
                self._op(f, f"loss:{kind}[{len(mods)}]", lib.mpmae_loss_multi, dt, 0, kind_id, _p(tab), len(mods), N,
                         kind=f"loss_{kind}_fwd")
        # image-level head chain (LN, pooling, linear heads, their losses) on the side lane next to the pixel heads and
        # their losses: both only read the decoder output
        self._fwd_join_keys = []
        if self.lanes and self.loss_multi and bool(self.opt["img_side"]):
            names = [op[0] for op in f]
            side_names = {"head:ln", "head:pool", "head:img"} | {n for n in names if n.startswith("loss:img")}
            idx = [i for i, n in enumerate(names) if n in side_names]
            if idx and "head:pix" in names:
                prod = f[names.index("head:pix") - 1][3]          # the op that completes the decoder output
                if prod["signal"] is None:
                    prod["signal"] = "dec_out"
                for j, i in enumerate(idx):
                    m = f[i][3]
                    m["lane"] = 1
                    if j == 0:
                        m["wait"] = tuple(m["wait"]) + (prod["signal"],)
                # (the categorical pixel loss on this lane too, next to the continuous one: two cross-lane events cost more than the overlap returns,
                #  3.645 / 3.641 vs 3.626 / 3.620 ms - profiles/r05/ab_cat_side.txt; removed)
                f[idx[-1]][3]["signal"] = "img_side_done"
                self._fwd_join_keys = ["img_side_done"]
        if self._front_rest is not None:          # the first main-lane op behind the fused stem kernel waits for the rest of the side-lane front
            at, key = self._front_rest
            mains = [op for op in f[at + 1:] if op[3]["lane"] == 0]
            if getattr(self, "_act_in_stem", False):      # act[0] came from the stem kernel (same lane); the poolings are first read at stage 1
                firsts = [op for op in mains if op[0].startswith("encoder.downsample_layers.0")]
                firsts[0][3]["wait"] = tuple(firsts[0][3]["wait"]) + (key,)
            else:
                mains[0][3]["wait"] = tuple(mains[0][3]["wait"]) + (key,)
            if self._prep_late:           # the depthwise kernel reads fp32 taps; the first STAGED weight belongs to the op behind it
                assert mains[0][0].endswith(":dw"), mains[0][0]
                mains[1][3]["wait"] = tuple(mains[1][3]["wait"]) + ("prep_done",)
        self.loss_scale = 1.0
        lv = P.get("loss_fn.log_vars") if cfg.loss_aggr == "uncertainty" else None
        glv = self.grads.get("loss_fn.log_vars") if cfg.loss_aggr == "uncertainty" else None
        self._fin_args = (_p(self.loss_acc), _p(lv), len(cfg.out_mods), _p(self.losses), _p(self.weighted),
                          _p(self.total), _p(self.coef), _p(glv))

    def finalize_loss(self, stream, with_dlogvars: bool, loss_scale: float = 1.0):
        """12 per-modality losses, uncertainty weighting, total, backward coefficients
        (and, with_dlogvars, d total / d log_vars accumulated into the gradient buffer)."""
        a = self._fin_args
        err = self.lib.mpmae_loss_finalize_guarded(a[0], self.loss_slots, a[1], a[2], float(loss_scale), a[3], a[4], a[5], a[6],
                                                   a[7] if with_dlogvars else None, *self._err_words(), stream)
        _lib.check(err, "loss_finalize")

    def _err_words(self):
        """(err_words, n_err, err_stride): grid-barrier error words of the persistent stage kernels (MpmaeMeters) - a timeout poisons this
        rank's loss with +inf in the finalisation, so that the all-reduced guard loss skips the update on every rank."""
        if hasattr(self, "ps_sync"):
            return _p(self.ps_sync), int(self._ps_launches), int(self.ps_sync.shape[1])
        return None, 0, 0

    # ------------------------------------------------------------------ backward program
    def _block_bwd_fused(self, lst, blk, dout, dx):
        """dout: gradient w.r.t. the block output [M,C]; writes the gradient w.r.t. its input into dx."""
        P, Gd, lib, dt = self.params, self.grads, self.lib, self.dt
        nm = self._block_names(blk)
        M, Cc, H, G = blk["M"], blk["C"], blk["H"], blk["G"]
        act = self.act[blk["stage"]] if blk["sparse"] else None
        rpg = blk["rpg"]
        tag = blk["prefix"]
        dz = self.scr_dz[:M * H]
        dxn = self.scr_dxn[:M * Cc]
        dd = self.scr_dd[:M * Cc]
        w2t, w1t = self.w[tag + ".W2T"], self.w[tag + ".W1T"]
        self._gemm(lst, tag + ":pw2.dgrad", "NONE", "DZ_STATS", A=dout, B=w2t["t"], C=dz, R=blk["h"], M=M, N=H, K=Cc,
                   lda=Cc, ldb=w2t["ld"], ldc=H, ldr=H, rpg=rpg, s0=blk["S0"], s1=blk["S1"])
        self._wgrad(lst, tag + ":pw2.wgrad", "NONE", "GRN", P=dout, Q=blk["h"], M=M, Nn=Cc, Kk=H, ldp=Cc, ldq=H,
                    dW=Gd[nm["w2"]], sn=H, sk=1, db=Gd[nm["b2"]], qp0=blk["scale"], qp1=P[nm["gb"]], rpg=rpg)
        self._op(lst, tag + ":grn.bwd", lib.mpmae_grn_bwd_finalize, _p(blk["S0"]), _p(blk["S1"]), _p(blk["Gx"]),
                 _p(blk["Ainv"]), _p(P[nm["gg"]]), G, H, _p(blk["coef"]), _p(Gd[nm["gg"]]), _p(Gd[nm["gb"]]))
        self._gemm(lst, tag + ":pw1.dgrad", "GRN_BWD", "STORE", A=dz, A2=blk["h"], B=w1t["t"], C=dxn, M=M, N=Cc, K=H,
                   lda=H, ldb=w1t["ld"], ldc=Cc, p0=blk["scale"], p1=blk["coef"], rpg=rpg)
        self._wgrad(lst, tag + ":pw1.wgrad", "GRN_BWD", "LN_AFFINE", P=dz, P2=blk["h"], Q=blk["dhat"], M=M, Nn=H,
                    Kk=Cc, ldp=H, ldq=Cc, dW=Gd[nm["w1"]], sn=Cc, sk=1, db=Gd[nm["b1"]], pp0=blk["scale"],
                    pp1=blk["coef"], qp0=P[nm["ln_w"]], qp1=P[nm["ln_b"]], rpg=rpg)
        self._op(lst, tag + ":ln.bwd", self._ln_bwd_fn, dt, _p(dxn), 1, 1.0, _p(blk["dhat"]), _p(blk["rstd"]),
                 _p(P[nm["ln_w"]]), _p(P[nm["ln_b"]]), 0, _p(dd), 0, _p(Gd[nm["ln_w"]]), _p(Gd[nm["ln_b"]]), M, Cc,
                 _p(act))
        self._dw_bwd(lst, blk, dd, dout, dx)

    def _build_backward(self):
        cfg, P, Gd, lib, dt, N, L, D = self.cfg, self.params, self.grads, self.lib, self.dt, self.N, self.L, self.D
        b = self.bwd_ops
        dims = cfg.dims
        y = self.dec_out
        # loss gradients w.r.t. predictions
        if self.loss_onepass:
            # one-pass losses: the pixel losses' gradient already exists WITHOUT its per-modality scalar (written by the forward kernels);
            # the scalars - final after the loss finalisation in front of this program - go into the staged transposed head weights (column
            # segments: the data-gradient GEMM) and into `head_rs`, the row scales of the heads' weight-gradient fold
            cm = torch.zeros(self.Wpix, dtype=torch.uint8)
            for t, om in enumerate(cfg.out_mods):
                if om.kind.startswith("pix"):
                    c0 = self.head_cols[om.name]
                    cm[c0:c0 + om.head_out] = t
            self.head_col_mod = cm.to(self.device)
            self.head_rs = torch.zeros(self.Wpix, dtype=torch.float32, device=self.device)
            # OUT of place: the staged copy is only rewritten by the forward's weight staging, so a second backward behind one forward
            # (retain_graph, the backward-only span replays of tools/) must not compound the scalars into it
            wt_ = self.w["head.pixT"]
            self.head_pixT_scaled = torch.zeros_like(wt_["t"])
            self._op(b, "head:scale", lib.mpmae_head_scale, dt, _p(wt_["t"]), _p(self.head_pixT_scaled), wt_["ld"], D, self.Wpix, _p(self.head_col_mod), _p(self.coef),
                     _p(self.head_rs), kind="head_scale", nbytes=2 * D * self.Wpix * 2)
        if self.loss_multi:
            # (the categorical losses on the side lane next to the continuous ones, forward and gradient: 4.99 vs 4.97 ms, not kept)
            for kind, (kind_id, tab, cnt) in self._loss_tabs.items():
                if self.loss_onepass and kind in ("pix_cont", "pix_cat"):
                    continue
                if kind == "pix_cont" and getattr(self, "_cont_rows", 0) and bool(self.opt["loss_rows_bwd"]):
                    self._op(b, f"dloss:{kind}[{cnt}]", lib.mpmae_loss_pix_cont_rows_bwd, dt, _p(tab), cnt, N, self._cont_rows,
                             self.p, cfg.img_size, kind=f"loss_{kind}_bwd")
                    continue
                if kind == "pix_cat" and getattr(self, "_cat_waves", False):
                    maxc = max(om.chans for om in cfg.out_mods if om.kind == "pix_cat")
                    self._op(b, f"dloss:{kind}[{cnt}]", lib.mpmae_loss_pix_cat_waves, dt, 1, _p(tab), cnt, N,
                             self.p * self.p * maxc, kind=f"loss_{kind}_bwd")
                    continue
                self._op(b, f"dloss:{kind}[{cnt}]", lib.mpmae_loss_multi, dt, 1, kind_id, _p(tab), cnt,
                         N if kind == "img" else N * L, kind=f"loss_{kind}_bwd")
        for om in ([] if self.loss_multi else cfg.out_mods):
            a = self.loss_args[om.name]
            if om.kind == "pix_cont":
                self._op(b, f"dloss:{om.name}", lib.mpmae_loss_pix_cont, dt, 1, C.byref(a), N * L)
            elif om.kind == "pix_cat":
                self._op(b, f"dloss:{om.name}", lib.mpmae_loss_pix_cat, dt, 1, C.byref(a), N * L)
            else:
                self._op(b, f"dloss:{om.name}", lib.mpmae_loss_img, dt, 1, C.byref(a))
        # the last reader of the static input buffers (targets, mask noise -> mask): everything after it may overlap the next input stage
        last_dl = max((i for i, op in enumerate(b) if op[0].startswith(("dloss:", "head:scale"))), default=None)
        self._inputs_free_key = None
        if last_dl is not None:
            if b[last_dl][3]["signal"] is None:
                b[last_dl][3]["signal"] = "inputs_free"
            self._inputs_free_key = b[last_dl][3]["signal"]
        ldp = self.pred_pix.shape[1]

        def contiguous(mods, suffix):
            ts = [Gd[f"pred_dict.{m.name}.{suffix}"] for m in mods]
            return all(a.data_ptr() + a.numel() * 4 == b_.data_ptr() for a, b_ in zip(ts, ts[1:]))

        # The image-level heads' data-gradient GEMM (256 rows: eight workgroups, ~20 us of pure latency) on the weight-gradient lane IN FRONT of the
        # heads' weight gradients: it only needs the image losses' gradient, and its consumer - the LayerNorm backward that accumulates
        # into dy behind the pixel heads' data gradient - waits for its signal
        img_dgrad_key = None
        if cfg.img_mods and self.lanes and bool(self.opt["img_dgrad_side"]):
            wt_i = self.w["head.imgT"]
            k_ = self._after(b)
            self._evseq += 1
            img_dgrad_key = f"s{self._evseq}"
            self._gemm(b, "head:img.dgrad", "NONE", "STORE", A=self.dpred_img, B=wt_i["t"], C=self.dpooled, M=N, N=D,
                       K=self.ldimg, lda=self.ldimg, ldb=wt_i["ld"], ldc=D)
            b[-1][2][-1]._obj.ws = self.ws2.data_ptr()      # (side-lane scratch)
            b[-1][3].update(lane=1, wait=(k_,) if k_ else (), signal=img_dgrad_key)
        if cfg.pix_mods and contiguous(cfg.pix_mods, "weight") and contiguous(cfg.pix_mods, "bias"):
            m0 = cfg.pix_mods[0]        # all pixel heads at once: dW [Wpix, D] and db [Wpix] are contiguous (see _build_params)
            self._side_wgrad(b, "head:pix.wgrad", "NONE", "NONE", [], P=self.dpred_pix, Q=y, M=N * L, Nn=self.Wpix, Kk=D,
                             ldp=ldp, ldq=D, dW=Gd[f"pred_dict.{m0.name}.weight"], sn=D, sk=1,
                             db=Gd[f"pred_dict.{m0.name}.bias"], **(dict(rowscale=self.head_rs) if self.loss_onepass else {}))
        else:
          for om in cfg.pix_mods:
            pv = self.dpred_pix.view(-1)[self.head_cols[om.name]:]
            self._side_wgrad(b, f"head:{om.name}.wgrad", "NONE", "NONE", [], P=pv, Q=y, M=N * L, Nn=om.head_out, Kk=D, ldp=ldp,
                        ldq=D, dW=Gd[f"pred_dict.{om.name}.weight"], sn=D, sk=1, db=Gd[f"pred_dict.{om.name}.bias"])
        have_pix = bool(cfg.pix_mods)
        if have_pix:
            wt = self.w["head.pixT"]
            self._gemm(b, "head:pix.dgrad", "NONE", "STORE", A=self.dpred_pix, B=self.head_pixT_scaled if self.loss_onepass else wt["t"], C=self.dy, M=N * L, N=D,
                       K=self.Wpix, lda=ldp, ldb=wt["ld"], ldc=D)
        if cfg.img_mods:
            if contiguous(cfg.img_mods, "weight") and contiguous(cfg.img_mods, "bias"):
                m0 = cfg.img_mods[0]
                self._side_wgrad(b, "head:img.wgrad", "NONE", "NONE", [], P=self.dpred_img, Q=self.pooled, M=N, Nn=self.Wimg, Kk=D,
                                 ldp=self.ldimg, ldq=D, dW=Gd[f"pred_dict.{m0.name}.weight"], sn=D, sk=1,
                                 db=Gd[f"pred_dict.{m0.name}.bias"])
            else:
              for om in cfg.img_mods:
                pv = self.dpred_img.view(-1)[self.head_cols[om.name]:]
                self._side_wgrad(b, f"head:{om.name}.wgrad", "NONE", "NONE", [], P=pv, Q=self.pooled, M=N, Nn=om.head_out, Kk=D,
                            ldp=self.ldimg, ldq=D, dW=Gd[f"pred_dict.{om.name}.weight"], sn=D, sk=1,
                            db=Gd[f"pred_dict.{om.name}.bias"])
            wt = self.w["head.imgT"]
            # K = the padded width: dpred_img's and the staged weights' padding columns are zero, and a multiple of 8
            # keeps this tiny GEMM on the fast kernel
            if img_dgrad_key is None:
                self._gemm(b, "head:img.dgrad", "NONE", "STORE", A=self.dpred_img, B=wt["t"], C=self.dpooled, M=N, N=D,
                           K=self.ldimg, lda=self.ldimg, ldb=wt["ld"], ldc=D)
            self._op(b, "head:ln.bwd", self._ln_bwd_callable(D), dt, _p(self.dpooled), L, 1.0 / L, _p(self.yhat), _p(self.rstd_y),
                     _p(P["layer_norm_tmp.weight"]), _p(P["layer_norm_tmp.bias"]), 0, _p(self.dy), 1 if have_pix else 0,
                     _p(Gd["layer_norm_tmp.weight"]), _p(Gd["layer_norm_tmp.bias"]), N * L, D, None,
                     wait=(img_dgrad_key,) if img_dgrad_key else ())
        self._fold_flush(b, "head")              # (inside the heads' gradient bucket: the exchange of a bucket must see its folds)
        # decoder block
        dxdec = self.scr_dxA[:N * L * D]
        cur_d = self.dy
        for j in range(len(self.decs) - 1, -1, -1):
            out_d = dxdec if j == 0 else self.dec_dx[j - 1]
            self._block_bwd(b, self.decs[j], cur_d, out_d)
            cur_d = out_d
        wpt = self.w["proj.WT"]
        cur = self.scr_dxB[:self.M[3] * dims[3]]
        if self.proj_compact:
            # the token-gradient pass over dxdec also gathers the visible rows: proj's two gradients are plain GEMMs on [M3, D]
            dyv = self.proj_rows                       # (the forward's compact rows are dead by now)
            if self.dense:      # rows of masked patches exist here and receive no gradient (x * (1 - mask), fcmae.py:255): the gather skips them
                self._op(b, "proj.dy.zero", lib.mpmae_memset_async, _p(dyv), 0, dyv.numel() * dyv.element_size())
            self._op(b, "mask_token.bwd", lib.mpmae_mask_token_bwd, dt, _p(dxdec), _p(self.inv), _p(Gd["mask_token"]), N * L, D,
                     _p(dyv), self.keep, L)
            self._side_wgrad(b, "proj.wgrad", "NONE", "NONE", [dyv], P=dyv, Q=self.enc_out, M=self.M[3], Nn=D, Kk=dims[3], ldp=D,
                             ldq=dims[3], dW=Gd["proj.weight"], sn=dims[3], sk=1, db=Gd["proj.bias"])
            self._gemm(b, "proj.dgrad", "NONE", "STORE", A=dyv, B=wpt["t"], C=cur, M=self.M[3], N=dims[3], K=D, lda=D,
                       ldb=wpt["ld"], ldc=dims[3], act=self.act[3])
        else:
            self._op(b, "mask_token.bwd", lib.mpmae_mask_token_bwd, dt, _p(dxdec), _p(self.inv), _p(Gd["mask_token"]), N * L, D,
                     None, 0, 0)
            self._side_wgrad(b, "proj.wgrad", "ROW_GATHER", "NONE", [dxdec], P=dxdec, Q=self.enc_out, M=self.M[3], Nn=D, Kk=dims[3], ldp=D,
                        ldq=dims[3], dW=Gd["proj.weight"], sn=dims[3], sk=1, db=Gd["proj.bias"], vis=self.vis,
                        keep=self.keep, L=L)
            self._gemm(b, "proj.dgrad", "ROW_GATHER", "STORE", A=dxdec, B=wpt["t"], C=cur, M=self.M[3], N=dims[3], K=D, lda=D,
                       ldb=wpt["ld"], ldc=dims[3], vis=self.vis, keep=self.keep, L=L, act=self.act[3])
        self._guard(b, cur)
        self._fold_flush(b, f"decoder_dict.{cfg.out_mods[0].name}")
        ring, ri = self.scr_dx, 2 % len(self.scr_dx)      # dxdec = ring[0], cur = ring[1]
        other = ring[ri]
        bi = len(self.blocks) - 1
        for i in range(3, -1, -1):
            for j in range(cfg.depths[i] - 1, -1, -1):
                blk = self.blocks[bi]
                nxt = other[:blk["M"] * blk["C"]]
                self._block_bwd(b, blk, cur, nxt)
                ri = (ri + 1) % len(ring)
                other = ring[ri]
                cur = nxt
                bi -= 1
            self._dwg_flush(b)            # the stage's grouped depthwise / pointwise weight gradients: side lane, behind its data-gradient chain
            self._group_flush(b)
            # (tail_main: the last fold group in order on the main lane - the weight-gradient lane is the later one at the end of the step)
            self._fold_flush(b, f"encoder.stages.{i}", lane=0 if (i == 0 and self.lanes and self._tail_main() >= 1) else 1)
            if i > 0:
                dn = self.down[i - 1]
                pre = f"encoder.downsample_layers.{i - 1}"
                Ci, Co = dims[i - 1], dims[i]
                wd = self.w[f"down{i - 1}.W"]
                nxt = other[:self.M[i - 1] * Ci]
                if dn["grouped"]:
                    wx, wy = min(Co, 4 * Ci), max(Co, 4 * Ci)
                    if (self.lanes and bool(self.opt["wgrad_group"]) and dt == BF16
                            and ((wx == 80 and wy % 320 == 0) or (wx % 160 == 0 and wy % 160 == 0)
                                 or (wx == 96 and wy % 384 == 0) or (wx % 192 == 0 and wy % 192 == 0))):
                        # a group of one: the DMA-ring kernel with few row splits instead of the transpose-read kernel's 76 slabs
                        self._group_add(b, pre + ":wgrad", [cur], P=cur, Q=dn["yg"], M=self.M[i], Nn=Co, Kk=4 * Ci,
                                        ldp=Co, ldq=4 * Ci, dW=Gd[pre + ".1.kernel"], sn=1, sk=Co, db=Gd[pre + ".1.bias"])
                        self._group_flush(b, name=pre + ":wgrad")
                    else:
                      self._side_wgrad(b, pre + ":wgrad", "NONE", "NONE", [cur], P=cur, Q=dn["yg"], M=self.M[i], Nn=Co, Kk=4 * Ci,
                                     ldp=Co, ldq=4 * Ci, dW=Gd[pre + ".1.kernel"], sn=1, sk=Co, db=Gd[pre + ".1.bias"])
                    dyg = self.scr_dxn[:self.M[i] * 4 * Ci]
                    self._gemm(b, pre + ":dgrad", "NONE", "STORE", A=cur, B=wd["t"], C=dyg, M=self.M[i], N=4 * Ci, K=Co,
                               lda=Co, ldb=wd["ld"], ldc=4 * Ci)
                    self._op(b, pre + ":ln.bwd", self._ln_bwd_callable(Ci, down=True), dt, _p(dyg), _p(dn["xhat"]), _p(dn["rstd"]),
                             _p(P[pre + ".0.ln.weight"]), _p(nxt), _p(Gd[pre + ".0.ln.weight"]), _p(Gd[pre + ".0.ln.bias"]),
                             self.M[i - 1], Ci, self.S[i - 1], _p(self.act[i - 1]), kind="ln_bwd_down",
                             nbytes=3 * self.M[i - 1] * Ci * (4 if dt == F32 else 2))
                else:
                    self._side_wgrad(b, pre + ":wgrad", "NONE", "DOWN_GATHER", [cur], P=cur, Q=dn["xhat"], M=self.M[i], Nn=Co,
                                     Kk=4 * Ci, ldp=Co, ldq=Ci, dW=Gd[pre + ".1.kernel"], sn=1, sk=Co, db=Gd[pre + ".1.bias"],
                                     qp0=P[pre + ".0.ln.weight"], qp1=P[pre + ".0.ln.bias"], S=self.S[i], Cseg=Ci,
                                     act_src=self.act[i - 1])
                    dxn = self.scr_dxn[:self.M[i - 1] * Ci]
                    self._gemm(b, pre + ":dgrad", "NONE", "DOWN_DGRAD", A=cur, B=wd["t"], C=dxn, M=self.M[i], N=4 * Ci, K=Co,
                               lda=Co, ldb=wd["ld"], ldc=Ci, S=self.S[i], Cseg=Ci, act_src=self.act[i - 1])
                    self._op(b, pre + ":ln.bwd", self._ln_bwd_fn, dt, _p(dxn), 1, 1.0, _p(dn["xhat"]), _p(dn["rstd"]),
                             _p(P[pre + ".0.ln.weight"]), _p(P[pre + ".0.ln.bias"]), 0, _p(nxt), 0,
                             _p(Gd[pre + ".0.ln.weight"]), _p(Gd[pre + ".0.ln.bias"]), self.M[i - 1], Ci, _p(self.act[i - 1]))
                self._guard(b, nxt)
                ri = (ri + 1) % len(ring)
                other = ring[ri]
                cur = nxt
        # stem
        C0, k = dims[0], cfg.stem_k
        dc1 = other[:self.Mfull * C0]
        if self.orig_stem:
            dc1 = other[:self.M[0] * C0]
            self._op(b, "stem:ln.bwd", self._ln_bwd_fn, dt, _p(cur), 1, 1.0, _p(self.s0hat), _p(self.rstd2),
                     _p(P["encoder.stem_orig.1.ln.weight"]), _p(P["encoder.stem_orig.1.ln.bias"]), 0, _p(dc1), 0,
                     _p(Gd["encoder.stem_orig.1.ln.weight"]), _p(Gd["encoder.stem_orig.1.ln.bias"]), self.M[0], C0, _p(self.act[0]))
        elif self.stem_fused:
            a = _lib.StemTailArgs()
            a.x, a.out = cur.data_ptr(), dc1.data_ptr()
            a.xhat1, a.rstd1, a.xhat2, a.rstd2 = (t.data_ptr() for t in (self.c1hat, self.rstd1, self.s0hat, self.rstd2))
            a.g1, a.b1 = P["encoder.initial_conv.1.ln.weight"].data_ptr(), P["encoder.initial_conv.1.ln.bias"].data_ptr()
            a.w, a.wb = P["encoder.stem.0.kernel"].data_ptr(), P["encoder.stem.0.bias"].data_ptr()
            a.g2, a.b2 = P["encoder.stem.1.ln.weight"].data_ptr(), P["encoder.stem.1.ln.bias"].data_ptr()
            a.act_in = self.act_full.data_ptr() if self.act_full is not None else 0
            a.act_out = self.act[0].data_ptr() if self.act[0] is not None else 0
            a.dg1, a.db1 = Gd["encoder.initial_conv.1.ln.weight"].data_ptr(), Gd["encoder.initial_conv.1.ln.bias"].data_ptr()
            a.dw, a.dwb = Gd["encoder.stem.0.kernel"].data_ptr(), Gd["encoder.stem.0.bias"].data_ptr()
            a.dg2, a.db2 = Gd["encoder.stem.1.ln.weight"].data_ptr(), Gd["encoder.stem.1.ln.bias"].data_ptr()
            a.ws, a.ws_floats = self.ws.data_ptr(), self.ws_floats
            a.M, a.C = self.Mfull, C0
            self._keepalive.append(a)
            self._op(b, "stem:ln+gelu+dw+ln.bwd", lib.mpmae_stem_tail, dt, 1, C.byref(a), kind="stem_tail_bwd",
                     nbytes=5 * self.Mfull * C0 * (4 if dt == F32 else 2))
        else:
          ds = self.scr_dd[:self.M[0] * C0]
          self._op(b, "stem:ln2.bwd", self._ln_bwd_fn, dt, _p(cur), 1, 1.0, _p(self.s0hat), _p(self.rstd2),
                   _p(P["encoder.stem.1.ln.weight"]), _p(P["encoder.stem.1.ln.bias"]), 0, _p(ds), 0,
                   _p(Gd["encoder.stem.1.ln.weight"]), _p(Gd["encoder.stem.1.ln.bias"]), self.M[0], C0, _p(self.act[0]))
          self._guard(b, ds)
          da1 = self.scr_dxn[:self.Mfull * C0]
          self._op(b, "stem:dw.bwd", lib.mpmae_dwstride_bwd, dt, _p(ds), _p(self.a1), _p(da1), _p(P["encoder.stem.0.kernel"]),
                   _p(Gd["encoder.stem.0.kernel"]), _p(Gd["encoder.stem.0.bias"]), self.M[0], C0, 8, k, _p(self.act_full),
                   _p(self.ws), self.ws_floats)
          dc1 = other[:self.Mfull * C0]
          self._op(b, "stem:ln1.bwd", self._ln_bwd_fn, dt, _p(da1), 1, 1.0, _p(self.c1hat), _p(self.rstd1),
                   _p(P["encoder.initial_conv.1.ln.weight"]), _p(P["encoder.initial_conv.1.ln.bias"]), 1, _p(dc1), 0,
                   _p(Gd["encoder.initial_conv.1.ln.weight"]), _p(Gd["encoder.initial_conv.1.ln.bias"]), self.Mfull, C0,
                   _p(self.act_full))
        self._guard(b, dc1)
        if self.stem_im2col:
            Kc = (k * k if self.orig_stem else 9) * cfg.in_chans
            Mc = self.M[0] if self.orig_stem else self.Mfull
            kkey, bkey = (("encoder.stem_orig.0.kernel", "encoder.stem_orig.0.bias") if self.orig_stem
                          else ("encoder.initial_conv.0.kernel", "encoder.initial_conv.0.bias"))
            self.dw_stem_pad = torch.zeros(C0 * self.ldk, dtype=torch.float32, device=self.device)
            # zeroed at the START of the backward: in the tail it sat on the critical path between the last data gradient
            # and AdamW (profiles/r01/timeline_final.txt)
            zs = self.lanes and bool(self.opt["zero_side"])      # (side lane: idle at that point, the weight gradient below waits for it)
            self._op(b, "stem:conv.dWpad.zero", lib.mpmae_memset_async, _p(self.dw_stem_pad), 0, C0 * self.ldk * 4,
                     **(dict(lane=1, signal="stem_pad_zero") if zs else {}))
            b.insert(0, b.pop())
            self._wgrad(b, "stem:conv.wgrad", "NONE", "NONE", wait=("stem_pad_zero",) if zs else (), P=dc1, Q=self.col, M=Mc, Nn=C0, Kk=self.ldk, ldp=C0,
                        ldq=self.ldk, dW=self.dw_stem_pad, sn=self.ldk, sk=1, db=Gd[bkey])
            # (C0, 9*Cin) padded row-major -> ME kernel layout (9, Cin, C0)
            self._op(b, "stem:conv.dW.fold", lib.mpmae_strided_add, _p(Gd[kkey]),
                     _p(self.dw_stem_pad), C0, Kc, self.ldk, 1, C0)
        else:
            self._wgrad(b, "stem:conv.wgrad", "NONE", "IM2COL3", P=dc1, Q=self.inp["sentinel2"], M=self.Mfull, Nn=C0,
                        Kk=9 * cfg.in_chans, ldp=C0, ldq=0, dW=Gd["encoder.initial_conv.0.kernel"], sn=1, sk=C0,
                        db=Gd["encoder.initial_conv.0.bias"], vis=self.vis, inv=self.inv, keep=self.keep, L=L, S=self.p,
                        Cseg=cfg.in_chans, grid=self.grid, H=cfg.img_size)
            # this weight gradient gathers its taps from the INPUT IMAGE (and the mask tables): it, not the last loss-gradient op, is the
            # last reader of the static input buffers - the asynchronous input stage of the next batch must wait for it
            if b[-1][3]["signal"] is None:
                b[-1][3]["signal"] = "inputs_free_stem"
            self._inputs_free_key = b[-1][3]["signal"]

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
