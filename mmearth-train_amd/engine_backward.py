"""Engine, part 5 of 6: the backward launch program (loss gradients, heads, decoder, stages 3..0, stem) with its weight-gradient lane."""
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


class BackwardMixin:
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
