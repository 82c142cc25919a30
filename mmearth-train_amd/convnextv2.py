"""Dense ConvNeXt V2 (the checkpoint CONSUMER of the pretraining path): the network `hubconf.MPMAE`
returns and `helpers.remap_checkpoint_keys` targets (/root/reference/models/convnextv2.py:18-207,
factories :210-246). Inference / fine-tuning of this net is outside the HIP hot path
and runs on stock PyTorch ops; what matters here is the module tree (state-dict keys and shapes) and the
arithmetic the sparse encoder must agree with when nothing is masked:

    initial_conv.0 (3x3, NO padding) -> .1 LN_cf -> GELU -> stem.0 depthwise k=s=patch/8, pad k//2 -> stem.1 LN_cf
    -> stages[0] -> 3 x (downsample_layers[i]: LN_cf + 2x2/2 conv, stages[i+1]) -> mean(H, W) -> norm -> head

Keys: initial_conv.{0,1}.*, stem.{0,1}.*, downsample_layers.i.{0,1}.*, stages.i.j.{dwconv,norm,pwconv1,grn,pwconv2}.*,
norm.*, head.*  (grn.gamma / beta are (1, 1, 1, 4C)).
"""
from argparse import Namespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from .config import SIZES


class LayerNorm(nn.Module):
    """Per-position LayerNorm over channels, biased variance (/root/reference/models/norm_layers.py:7-31)."""

    def __init__(self, dim, eps=1e-6, data_format="channels_last"):
        super().__init__()
        if data_format not in ("channels_last", "channels_first"):
            raise NotImplementedError(data_format)
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))
        self.eps, self.data_format, self.dim = eps, data_format, dim

    def forward(self, x):
        if self.data_format == "channels_last":
            return F.layer_norm(x, (self.dim,), self.weight, self.bias, self.eps)
        y = F.layer_norm(x.permute(0, 2, 3, 1), (self.dim,), self.weight, self.bias, self.eps)
        return y.permute(0, 3, 1, 2)


class GRN(nn.Module):
    """Per-sample global response normalisation over (H, W), eps 1e-4 (norm_layers.py:33-44)."""

    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.zeros(1, 1, 1, dim))
        self.beta = nn.Parameter(torch.zeros(1, 1, 1, dim))

    def forward(self, x):                                    # x: [N, H, W, C]
        g = x.pow(2).sum(dim=(1, 2), keepdim=True).sqrt()
        return x + self.beta + self.gamma * (x * (g / (g.mean(dim=-1, keepdim=True) + 1e-4)))


class Block(nn.Module):
    """dw7x7 -> LN -> Linear C->4C -> GELU -> GRN -> Linear 4C->C, residual (convnextv2.py:18-55)."""

    def __init__(self, dim, drop_path=0.0):
        super().__init__()
        if drop_path:
            raise NotImplementedError("stochastic depth is not used by the pretraining recipes")
        self.dwconv = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, 4 * dim)
        self.act = nn.GELU()
        self.grn = GRN(4 * dim)
        self.pwconv2 = nn.Linear(4 * dim, dim)

    def forward(self, x):
        y = self.norm(self.dwconv(x).permute(0, 2, 3, 1))
        y = self.pwconv2(self.grn(self.act(self.pwconv1(y))))
        return x + y.permute(0, 3, 1, 2)


class ConvNeXtV2(nn.Module):
    def __init__(self, patch_size: int = 8, img_size: int = 56, in_chans: int = 12, num_classes: int = 1000,
                 depths=None, dims=None, drop_path_rate: float = 0.0, head_init_scale: float = 1.0,
                 use_orig_stem: bool = False, args: Namespace = None):
        super().__init__()
        depths = list(depths or [3, 3, 9, 3])
        dims = list(dims or [96, 192, 384, 768])
        if drop_path_rate:
            raise NotImplementedError("drop_path_rate > 0")
        self.depths, self.img_size, self.patch_size, self.use_orig_stem = depths, img_size, patch_size, use_orig_stem
        self.num_stage = len(depths)
        k = patch_size // (2 ** (self.num_stage - 1))
        self.downsample_layers = nn.ModuleList()      # registered first, as in the reference (state-dict key order)
        if use_orig_stem:
            self.stem_orig = nn.Sequential(nn.Conv2d(in_chans, dims[0], kernel_size=k, stride=k),
                                           LayerNorm(dims[0], data_format="channels_first"))
        else:
            self.initial_conv = nn.Sequential(nn.Conv2d(in_chans, dims[0], kernel_size=3, stride=1),
                                              LayerNorm(dims[0], data_format="channels_first"), nn.GELU())
            self.stem = nn.Sequential(nn.Conv2d(dims[0], dims[0], kernel_size=k, stride=k, padding=k // 2, groups=dims[0]),
                                      LayerNorm(dims[0], data_format="channels_first"))
        for i in range(3):
            self.downsample_layers.append(nn.Sequential(LayerNorm(dims[i], data_format="channels_first"),
                                                        nn.Conv2d(dims[i], dims[i + 1], kernel_size=2, stride=2)))
        self.stages = nn.ModuleList(nn.Sequential(*[Block(dims[i]) for _ in range(depths[i])])
                                    for i in range(self.num_stage))
        self.norm = nn.LayerNorm(dims[-1], eps=1e-6)
        self.head = nn.Linear(dims[-1], num_classes)
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.trunc_normal_(m.weight, std=0.02, a=-2.0, b=2.0)
                nn.init.zeros_(m.bias)
        with torch.no_grad():
            self.head.weight.mul_(head_init_scale)
            self.head.bias.mul_(head_init_scale)

    def _trunk(self, x):
        x = self.stem_orig(x) if self.use_orig_stem else self.stem(self.initial_conv(x))
        x = self.stages[0](x)
        for i in range(3):
            x = self.stages[i + 1](self.downsample_layers[i](x))
        return x

    def forward_features(self, x):
        return self.norm(self._trunk(x).mean([-2, -1]))

    def upsample_mask(self, mask, scale):
        assert len(mask.shape) == 2
        p = int(mask.shape[1] ** 0.5)
        return mask.reshape(-1, p, p).repeat_interleave(scale, dim=1).repeat_interleave(scale, dim=2)

    def forward(self, x, mask=None):
        if mask is not None:          # the reference's dense pretraining mode: mask the input once, return the map
            scale = int(self.img_size // (mask.shape[1] ** 0.5))
            x *= 1.0 - self.upsample_mask(mask, scale).unsqueeze(1).type_as(x)
            return self._trunk(x)
        return self.head(self.forward_features(x))


def _factory(name):
    depths, dims = SIZES[name]

    def make(**kwargs):
        return ConvNeXtV2(depths=list(depths), dims=list(dims), **kwargs)
    make.__name__ = name
    return make


convnextv2_atto = _factory("convnextv2_atto")
convnextv2_femto = _factory("convnextv2_femto")
convnext_pico = convnextv2_pico = _factory("convnextv2_pico")      # the reference spells this one `convnext_pico`
convnextv2_nano = _factory("convnextv2_nano")
convnextv2_tiny = _factory("convnextv2_tiny")
convnextv2_base = _factory("convnextv2_base")
convnextv2_large = _factory("convnextv2_large")
convnextv2_huge = _factory("convnextv2_huge")
