"""Drop-in `fcmae` module: FCMAE with the reference's constructor, forward signature, return
tuple, attribute names, state-dict key layout and size factories
(/root/reference/models/fcmae.py:27-496), running on the MI355X HIP engine.

    model = fcmae.convnextv2_atto(mask_ratio=0.6, decoder_depth=1, decoder_embed_dim=512,
                                  norm_pix_loss=True, patch_size=8, img_size=56, args=args,
                                  loss_fn=UncertaintyWeightingStrategy(12), sparse=True)
    loss, pred, mask, loss_dict, log_vars, normalized = model(samples, mask_ratio=0.6)
    loss.backward(); optimizer.step()

Parameters are nn.Parameters whose storage is one flat fp32 buffer (the engine's), exposed under
the reference's names through a generated module tree, so `state_dict()` / `load_state_dict()` /
`named_parameters()` / timm-style weight-decay grouping behave as with the reference (the shared
decoder block appears under every `decoder_dict.<modality>`). There is no CPU fallback: forward
needs the HIP library and a GPU. `sparse=False` selects the dense ConvNeXtV2 encoder (fcmae.py:103-111; patch_size 16 only, as
in the reference, whose dense stem does not line up with the patch grid at patch 8).
"""
import math
from argparse import Namespace
from collections import OrderedDict
from typing import AnyStr, Dict

import torch
import torch.nn as nn
from torch import Tensor

from .MODALITIES import PIXEL_WISE_MODALITIES
from .config import SIZES, cfg_from_args
from .engine import Engine
from .synth import flat_param_spec, param_view, state_dict_spec


def _trunc_normal_(t, std, gen=None):
    return torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0, generator=gen)


def init_reference_(params: "OrderedDict[str, Tensor]", generator=None):
    """Initial values as left by the reference's FCMAE.apply(_init_weights) (fcmae.py:157-178, which
    overrides SparseConvNeXtV2's own init): ME depthwise kernels, MinkowskiLinear weights and every
    nn.Conv2d weight ~ trunc_normal(std=1, +-2); ME conv kernels and nn.Linear weights std 0.02;
    biases 0; LayerNorm 1/0; GRN gamma/beta 0; mask_token N(0, 0.02); log_vars 0."""
    with torch.no_grad():
        for k, t in params.items():
            if k == "mask_token":
                t.normal_(0.0, 0.02, generator=generator)
            elif k.endswith("bias") or k.endswith(".beta") or k.endswith(".gamma") or k == "loss_fn.log_vars":
                t.zero_()
            elif k.endswith("ln.weight") or k.endswith("norm.weight") or k == "layer_norm_tmp.weight" or (k.endswith(".weight") and t.dim() == 1):
                t.fill_(1.0)                         # (1-d weights are normalisation scales: the dense encoder's LayerNorms are `<layer>.weight`)
            elif k.endswith("dwconv.kernel") or k.endswith("stem.0.kernel") or ".linear.weight" in k:
                _trunc_normal_(t, 1.0, generator)
            elif k.endswith(".kernel"):
                _trunc_normal_(t, 0.02, generator)
            elif t.dim() == 4:                       # nn.Conv2d: proj, decoder dwconv, pixel heads
                _trunc_normal_(t, 1.0, generator)
            else:                                    # nn.Linear: decoder pwconv1/2, image heads
                _trunc_normal_(t, 0.02, generator)


class _Node(nn.Module):
    """Empty container used to reproduce the reference's module / parameter names."""


class _StepFn(torch.autograd.Function):
    """Connects the engine's explicit forward/backward programs to torch autograd so that
    `loss.backward()` fills `.grad` of the module's parameters."""

    @staticmethod
    def forward(ctx, model, loss_scale_token, *params):
        ctx.model = model
        ctx.eng = eng = model._engine
        eng.forward()
        eng.fwd_generation = getattr(eng, "fwd_generation", 0) + 1
        ctx.generation = eng.fwd_generation
        return eng.total.clone().reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        model, eng = ctx.model, ctx.eng
        if getattr(eng, "fwd_generation", 0) != ctx.generation:
            raise RuntimeError("FCMAE.backward: the engine's static activation buffers were overwritten by a later "
                               "forward of the same batch size; call loss.backward() before the next forward")
        eng.backward(zero_grad=True)
        g = eng.gflat * grad_out
        # parameters outside the pretraining graph (the dense encoder's classifier head and final norm, convnextv2.py:151-152: grad is
        # None in the reference, so torch.optim.AdamW skips them - a zero tensor would still apply the decoupled weight decay)
        outs = tuple(None if (eng.dense and k.startswith(("encoder.head.", "encoder.norm."))) else param_view(g[o:o + n], k, p.shape)
                     for k, p, (o, n) in zip(model._pkeys, model._plist, model._poffs))
        return (None, None) + outs


class _PieceOutput(torch.autograd.Function):
    """Marks the outputs of forward_encoder / forward_decoder / forward_loss: they are computed by the engine's forward segments and
    are NOT connected to the fused backward. In the reference (models/fcmae.py:242-412) chaining the three pieces and calling
    loss.backward() trains the model; here that chain must not silently produce no gradients, so its backward raises."""

    @staticmethod
    def forward(ctx, t, anchor, what):
        ctx.what = what
        return t.clone()

    @staticmethod
    def backward(ctx, grad_out):
        raise RuntimeError(f"FCMAE.{ctx.what}: the outputs of forward_encoder / forward_decoder / forward_loss are not connected to "
                           "the fused backward (they would silently yield no gradients); call model(imgs_dict, mask_ratio=...) "
                           "and loss.backward() on ITS loss to train")


class FCMAE(nn.Module):
    """Fully Convolutional Masked Autoencoder with ConvNeXtV2 backbone (MP-MAE), HIP engine."""

    def __init__(self, img_size: int = 112, depths=None, dims=None, decoder_depth: int = 1,
                 decoder_embed_dim: int = 512, patch_size: int = 16, mask_ratio: float = 0.6,
                 norm_pix_loss: bool = False, args: Namespace = None, loss_fn=None, sparse: bool = True,
                 device=None, dtype: str = "bf16"):
        super().__init__()
        depths = depths or [3, 3, 9, 3]
        dims = dims or [96, 192, 384, 768]
        name = next((k for k, (d, c) in SIZES.items() if d == list(depths) and c == list(dims)), None)
        if name is None:
            raise ValueError("depths/dims must be one of the convnextv2_* size presets")
        self.args = args
        self.img_size, self.depths, self.dims = img_size, list(depths), list(dims)
        self.patch_size, self.mask_ratio = patch_size, mask_ratio
        self.num_patches = (img_size // patch_size) ** 2
        self.decoder_embed_dim, self.decoder_depth = decoder_embed_dim, decoder_depth
        self.norm_pix_loss, self.sparse = norm_pix_loss, sparse
        self.cfg = cfg_from_args(name, img_size, patch_size, args, norm_pix_loss, mask_ratio,
                                 decoder_embed_dim, decoder_depth, sparse=sparse)      # (sparse=False: the dense encoder, patch 16 only)
        if (self.cfg.loss_aggr == "uncertainty") != (loss_fn is not None):
            raise ValueError("loss_fn must be given iff args.loss_aggr == 'uncertainty'")
        self.in_chans = self.cfg.in_chans
        self.out_chans = {m: c.chans for m, c in ((om.name, om) for om in self.cfg.out_mods)}
        self.compute_dtype = dtype
        self._device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        self._engine, self._engines = None, {}

        spec = state_dict_spec(self.cfg)
        total = sum(math.prod(s) for _, s, _ in spec)
        self._pflat = torch.zeros(total, dtype=torch.float32, device=self._device)
        self._gflat = torch.zeros(total + 4, dtype=torch.float32, device=self._device)[:total]      # (+ the loss slot of the exchange, engine.gflat_ext)
        self._plist, self._pkeys, self._poffs, views = [], [], [], OrderedDict()
        first = self.cfg.out_mods[0].name
        offs, off = {}, 0
        for key, shape, _ in flat_param_spec(self.cfg):      # the engine's layout of the flat buffers
            offs[key] = off
            off += math.prod(shape)
        for key, shape, _ in spec:                            # registration (state-dict) order = the reference's
            n, off = math.prod(shape), offs[key]
            p = nn.Parameter(param_view(self._pflat[off:off + n], key, shape))
            views[key] = p
            self._plist.append(p)
            self._pkeys.append(key)
            self._poffs.append((off, n))
        init_reference_(OrderedDict((k, p.data) for k, p in views.items()))
        # module tree with the reference's names
        self.loss_fn = loss_fn
        # registration order = the reference's (named_parameters() / optimizer state indices line up with a model
        # built by /root/reference/models/fcmae.py): SparseConvNeXtV2 creates downsample_layers before the stem
        self.add_module("encoder", _Node())
        for child in ("downsample_layers", "initial_conv", "stem", "stages"):
            self.encoder.add_module(child, _Node())
        for key, p in views.items():
            if key == "loss_fn.log_vars":
                loss_fn.log_vars = p
                continue
            parts = key.split(".")
            node = self
            for part in parts[:-1]:
                if not hasattr(node, part) or getattr(node, part) is None:
                    node.add_module(part, _Node())
                node = getattr(node, part)
            node.register_parameter(parts[-1], p)
        shared = getattr(self.decoder_dict, first)          # the same Block objects for every modality (fcmae.py:137,145)
        for om in self.cfg.out_mods[1:]:
            self.decoder_dict.add_module(om.name, shared)

    # ------------------------------------------------------------------ engine plumbing
    def _apply(self, fn, recurse=True):
        raise_if = fn(torch.zeros(1, device=self._pflat.device))
        if raise_if.device != self._pflat.device or raise_if.dtype != torch.float32:
            raise RuntimeError("FCMAE parameters live in one flat fp32 buffer on the construction device; "
                               "pass device=... to the constructor instead of calling .to()/.half()")
        return self

    def _get_engine(self, N: int, mask_ratio=None) -> Engine:
        """One engine (buffer plan + launch program) per (batch size, number of visible patches): the call-time
        mask_ratio decides len_keep, as in the reference (fcmae.py:415,451 - the constructor's value is only stored)."""
        keep = self.cfg.len_keep(mask_ratio)
        eng = self._engines.get((N, keep))
        if eng is None:
            eng = Engine(self.cfg, N, dtype=self.compute_dtype, device=self._device,
                         param_buffers=(self._pflat, self._gflat), mask_ratio=mask_ratio)
            self._engines[(N, keep)] = eng
        return eng

    # ------------------------------------------------------------------ reference helpers
    def patchify(self, imgs: Tensor, modality: str) -> Tensor:
        p = self.patch_size
        assert imgs.shape[2] == imgs.shape[3] and imgs.shape[2] % p == 0
        channels = 1 if modality in ["dynamic_world", "esa_worldcover"] else self.out_chans[modality]
        h = w = imgs.shape[2] // p
        x = imgs.reshape(shape=(imgs.shape[0], channels, h, p, w, p))
        x = torch.einsum("nchpwq->nhwpqc", x)
        return x.reshape(shape=(imgs.shape[0], h * w, p ** 2 * channels))

    def unpatchify(self, x: Tensor) -> Tensor:
        p = self.patch_size
        h = w = self.img_size // p
        x = x.reshape(shape=(x.shape[0], h, w, p, p, self.in_chans))
        x = torch.einsum("nhwpqc->nchpwq", x)
        return x.reshape(shape=(x.shape[0], self.in_chans, h * p, h * p))

    def gen_random_mask(self, x: Tensor, mask_ratio: float) -> Tensor:
        N = x.shape[0]
        L = (x.shape[2] // self.patch_size) ** 2
        len_keep = int(L * (1 - mask_ratio))
        noise = torch.randn(N, L, device=x.device)
        ids_restore = torch.argsort(torch.argsort(noise, dim=1), dim=1)
        mask = torch.ones([N, L], device=x.device)
        mask[:, :len_keep] = 0
        return torch.gather(mask, dim=1, index=ids_restore)

    def upsample_mask(self, mask: Tensor, scale: float):
        assert len(mask.shape) == 2
        p = int(mask.shape[1] ** 0.5)
        return mask.reshape(-1, p, p).repeat_interleave(scale, dim=1).repeat_interleave(scale, dim=2)

    def _crop_windows(self, imgs_dict):
        """Per-sample crop origins (ty, tx) shared by all pixel-wise modalities (kornia RandomCrop, fcmae.py:419-434), or None
        when the tiles already have img_size. Drawn on the device; the crop itself is mpmae_crop (Engine.set_inputs)."""
        S, H = self.img_size, imgs_dict["sentinel2"].shape[-1]
        if H == S:
            return None
        N, dev = imgs_dict["sentinel2"].shape[0], self._device
        return (torch.randint(0, H - S + 1, (N,), device=dev, dtype=torch.int32),
                torch.randint(0, H - S + 1, (N,), device=dev, dtype=torch.int32))

    def _crop(self, imgs_dict, windows=None):
        """Reference-shaped cropped dict (what the reference leaves in the caller's imgs_dict); host-side torch indexing,
        used for the dict hand-back only - the engine crops with the HIP kernel."""
        windows = windows if windows is not None else self._crop_windows(imgs_dict)
        if windows is None:
            return imgs_dict
        S = self.img_size
        ty, tx = windows[0].long(), windows[1].long()
        dev = ty.device
        N = ty.shape[0]
        ar = torch.arange(S, device=dev)
        yy = (ty[:, None] + ar[None, :])[:, None, :, None]
        xx = (tx[:, None] + ar[None, :])[:, None, None, :]
        nn_ = torch.arange(N, device=dev)[:, None, None, None]
        out = {}
        for k, v in imgs_dict.items():
            if k in PIXEL_WISE_MODALITIES:
                cc = torch.arange(v.shape[1], device=dev)[None, :, None, None]
                out[k] = v.to(dev)[nn_, cc, yy, xx]
            else:
                out[k] = v
        return out

    # ------------------------------------------------------------------ stage helpers (fcmae.py:242-412)
    # The three public pieces of the reference's forward. They run the matching segment of the engine's forward
    # launch program and return detached tensors in the reference's shapes; only `forward` is connected to autograd.
    def forward_encoder(self, imgs: Tensor, mask_ratio: float):
        """imgs [N, C, S, S] -> (x [N, dims[-1], S/p, S/p] dense encoder output, mask [N, L])  (fcmae.py:242-247)"""
        N = imgs.shape[0]
        eng = self._engine = self._get_engine(N, mask_ratio)
        eng.inp["sentinel2"].copy_(imgs.reshape(eng.inp["sentinel2"].shape), non_blocking=True)
        eng.noise.copy_(torch.randn(N, self.cfg.num_patches, device=self._device))
        eng.run_segment("encoder")
        x = eng.dense_map(eng.enc_out, self.cfg.dims[-1], 3).to(torch.float32)
        return self._piece(x, "forward_encoder"), eng.mask.clone()

    def forward_decoder(self, x: Tensor, mask: Tensor):
        """x [N, dims[-1], h, w], mask [N, L] (0 keep / 1 remove) -> dict modality -> prediction (fcmae.py:249-265).
        The mask must keep the same number of patches in every sample (what gen_random_mask produces)."""
        N = x.shape[0]
        keep = int((mask[0] == 0).sum().item())
        if not bool(((mask == 0).sum(dim=1) == keep).all()):
            raise ValueError("forward_decoder: every sample must keep the same number of patches")
        eng = self._engine = self._get_engine(N, 1.0 - (keep + 0.5) / self.cfg.num_patches)
        assert eng.keep_mask == keep
        eng.set_mask(mask)
        rows = x.permute(0, 2, 3, 1).reshape(N, self.cfg.num_patches, -1)
        vis = eng.vis.view(N, eng.keep).long()           # (dense encoder: every patch is a row)
        eng.enc_out.copy_(torch.gather(rows, 1, vis[:, :, None].expand(-1, -1, rows.shape[-1]))
                          .reshape(eng.enc_out.shape).to(eng.enc_out.dtype))
        eng.run_segment("decoder")
        return OrderedDict((k, self._piece(v.to(torch.float32), "forward_decoder")) for k, v in eng.preds().items())

    def forward_loss(self, imgs_dict: Dict[AnyStr, Tensor], preds: Dict[AnyStr, Tensor], mask: Tensor):
        """-> (loss, loss_dict, log_vars, normalized_loss_list)  (fcmae.py:267-412)"""
        N = mask.shape[0]
        keep = int((mask[0] == 0).sum().item())
        eng = self._engine = self._get_engine(N, 1.0 - (keep + 0.5) / self.cfg.num_patches)
        eng.set_mask(mask)
        for k, dst in eng.inp.items():
            if k != "sentinel2" or "sentinel2" in preds:
                dst.copy_(imgs_dict[k].reshape(dst.shape), non_blocking=True)
        eng.set_preds(preds)
        eng.run_segment("loss")
        return self._loss_outputs(eng, self._piece(eng.total.clone().reshape(()), "forward_loss"))

    def _piece(self, t, what):
        """Output of a forward piece: a tensor whose backward RAISES (see _PieceOutput) when autograd is recording, else t itself."""
        if not torch.is_grad_enabled():
            return t
        return _PieceOutput.apply(t.detach(), self._plist[0], what)

    def _loss_outputs(self, eng, loss):
        losses = eng.losses.clone()
        loss_dict = OrderedDict((om.name, losses[i]) for i, om in enumerate(self.cfg.out_mods))
        if self.cfg.loss_aggr == "uncertainty":
            return loss, loss_dict, self.loss_fn.log_vars.tolist(), eng.weighted.clone()
        return loss, loss_dict, None, None

    # ------------------------------------------------------------------ forward
    def forward(self, imgs_dict: Dict[AnyStr, Tensor], labels=None, mask_ratio: float = 0.6):
        N = imgs_dict["sentinel2"].shape[0]
        eng = self._engine = self._get_engine(N, mask_ratio)
        windows = self._crop_windows(imgs_dict)
        noise = torch.randn(N, self.cfg.num_patches, device=self._device)
        if windows is not None:
            dev_dict = {k: (v.to(self._device) if k in eng.inp else v) for k, v in imgs_dict.items()}
            eng.set_inputs(dev_dict, noise, crop=windows)
            for k in imgs_dict:              # the reference replaces the caller's dict entries by the cropped tiles
                if k in PIXEL_WISE_MODALITIES and k in eng.inp:
                    imgs_dict[k] = eng.inp[k].clone()
        else:
            eng.set_inputs(imgs_dict, noise)
        loss = _StepFn.apply(self, None, *self._plist)
        pred = eng.preds()
        mask = eng.mask.clone()
        loss, loss_dict, log_vars, normalized = self._loss_outputs(eng, loss)
        return loss, pred, mask, loss_dict, log_vars, normalized


def _factory(name):
    depths, dims = SIZES[name]

    def make(**kwargs):
        return FCMAE(depths=list(depths), dims=list(dims), **kwargs)
    make.__name__ = name
    return make


convnextv2_atto = _factory("convnextv2_atto")
convnextv2_femto = _factory("convnextv2_femto")
convnextv2_pico = _factory("convnextv2_pico")
convnextv2_nano = _factory("convnextv2_nano")
convnextv2_tiny = _factory("convnextv2_tiny")
convnextv2_base = _factory("convnextv2_base")
convnextv2_large = _factory("convnextv2_large")
convnextv2_huge = _factory("convnextv2_huge")
