"""Engine, part 3 of 6: the launch programs of ONE ConvNeXtV2 block (forward and backward, 'mat' and 'fused' modes) and of a persistent stage (models/convnextv2_sparse.py:26-60, models/convnextv2.py:18-55 under autograd)."""
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


class BlocksMixin:
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
            # one row per launch: {arrivals, departures, error, -, debug...} + the counters / flags of the XCD-hierarchical grid barrier on their own lines
            self.ps_sync = torch.zeros(8, _lib.PS_SYNC_WORDS, dtype=torch.int32, device=self.device)
            self._ps_launches = 0
        a.sync = self.ps_sync[self._ps_launches].data_ptr()
        a.sync_words = _lib.PS_SYNC_WORDS if int(self.opt["ps_xcd_barrier"]) else 4
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
            self._rs(lst, tag + ":pw2.T+stats", 6, blk, (M * Cc + M * H) * esz + Cc * H * 4, 2 * M * Cc * H, A=dout, R=blk["h"],
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
            self._rs(lst, tag + ":grn.bapply+pw1.dgrad+ln.bwd" + ("+pw1.wg" if wgf else ""), 5, blk,
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
