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
needs the HIP library and a GPU (`sparse=False`, the reference's dense debug path, is not provided).
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
from .synth import flat_param_spec, state_dict_spec


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
            elif k.endswith("ln.weight") or k.endswith("norm.weight") or k == "layer_norm_tmp.weight":
                t.fill_(1.0)
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
        eng = model._engine
        eng.forward()
        return eng.total.clone().reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        model = ctx.model
        eng = model._engine
        eng.backward(zero_grad=True)
        g = eng.gflat * grad_out
        outs = tuple(g[o:o + n].view(p.shape) for p, (o, n) in zip(model._plist, model._poffs))
        return (None, None) + outs


class FCMAE(nn.Module):
    """Fully Convolutional Masked Autoencoder with ConvNeXtV2 backbone (MP-MAE), HIP engine."""

    def __init__(self, img_size: int = 112, depths=None, dims=None, decoder_depth: int = 1,
                 decoder_embed_dim: int = 512, patch_size: int = 16, mask_ratio: float = 0.6,
                 norm_pix_loss: bool = False, args: Namespace = None, loss_fn=None, sparse: bool = True,
                 device=None, dtype: str = "bf16"):
        super().__init__()
        if not sparse:
            raise NotImplementedError("only the sparse encoder (the reference default, main_pretrain.py:157) is provided")
        if getattr(args, "use_orig_stem", False):
            raise NotImplementedError("use_orig_stem=True is not used by any reference recipe (TRAINING.md:39)")
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
                                 decoder_embed_dim, decoder_depth)
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
        self._gflat = torch.zeros(total, dtype=torch.float32, device=self._device)
        self._plist, self._poffs, views = [], [], OrderedDict()
        first = self.cfg.out_mods[0].name
        offs, off = {}, 0
        for key, shape, _ in flat_param_spec(self.cfg):      # the engine's layout of the flat buffers
            offs[key] = off
            off += math.prod(shape)
        for key, shape, _ in spec:                            # registration (state-dict) order = the reference's
            n, off = math.prod(shape), offs[key]
            p = nn.Parameter(self._pflat[off:off + n].view(shape))
            views[key] = p
            self._plist.append(p)
            self._poffs.append((off, n))
        init_reference_(OrderedDict((k, p.data) for k, p in views.items()))
        # module tree with the reference's names
        self.loss_fn = loss_fn
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

    def _get_engine(self, N: int) -> Engine:
        eng = self._engines.get(N)
        if eng is None:
            eng = Engine(self.cfg, N, dtype=self.compute_dtype, device=self._device,
                         param_buffers=(self._pflat, self._gflat))
            self._engines[N] = eng
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

    def _crop(self, imgs_dict):
        """Same random crop window per sample for all pixel-wise modalities (kornia RandomCrop,
        fcmae.py:419-434); identity when the tiles already have img_size."""
        S = self.img_size
        H = imgs_dict["sentinel2"].shape[-1]
        if H == S:
            return imgs_dict
        N = imgs_dict["sentinel2"].shape[0]
        dev = imgs_dict["sentinel2"].device
        ty = torch.randint(0, H - S + 1, (N,), device=dev)
        tx = torch.randint(0, H - S + 1, (N,), device=dev)
        ar = torch.arange(S, device=dev)
        yy = (ty[:, None] + ar[None, :])[:, None, :, None]
        xx = (tx[:, None] + ar[None, :])[:, None, None, :]
        nn_ = torch.arange(N, device=dev)[:, None, None, None]
        out = {}
        for k, v in imgs_dict.items():
            if k in PIXEL_WISE_MODALITIES:
                cc = torch.arange(v.shape[1], device=dev)[None, :, None, None]
                out[k] = v[nn_, cc, yy, xx]
            else:
                out[k] = v
        return out

    # ------------------------------------------------------------------ forward
    def forward(self, imgs_dict: Dict[AnyStr, Tensor], labels=None, mask_ratio: float = 0.6):
        imgs_dict = self._crop(imgs_dict)
        N = imgs_dict["sentinel2"].shape[0]
        if abs(mask_ratio - self.cfg.mask_ratio) > 1e-12:
            raise NotImplementedError("call-time mask_ratio must equal the constructor's (engine buffers are sized by it)")
        eng = self._engine = self._get_engine(N)
        noise = torch.randn(N, self.cfg.num_patches, device=self._device)
        eng.set_inputs(imgs_dict, noise)
        loss = _StepFn.apply(self, None, *self._plist)
        pred = eng.preds()
        mask = eng.mask.clone()
        losses = eng.losses.clone()
        loss_dict = OrderedDict((om.name, losses[i]) for i, om in enumerate(self.cfg.out_mods))
        if self.cfg.loss_aggr == "uncertainty":
            log_vars = self.loss_fn.log_vars.tolist()
            normalized = eng.weighted.clone()
        else:
            log_vars, normalized = None, None
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
