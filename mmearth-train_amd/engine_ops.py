"""Engine, part 2 of 6: op helpers - one C-ABI launch per op tuple, cross-lane hazard tracking, grouped weight gradients, GEMM / weight-gradient / MX-fp8 / row-streaming / depthwise argument records."""
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


class OpsMixin:
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
