"""Engine, part 4 of 6: the forward launch program (fcmae.py:414-456: mask, stem, stages, decoder, heads, losses) and the loss finalisation."""
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


class ForwardMixin:
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
                                     and 16 * self.p * self.p * mk_ * 4 <= 150 * 1024)      # (x 4 although the bf16 slots are 2-byte: with x 2 patch 16 qualifies too -
            # tiny 112/16 then runs the wave kernel + one-pass losses and measured 15.00 / 15.04 vs 14.945 / 14.95 ms, profiles/r06/ab_tiny_onepass_not_kept.txt)
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
