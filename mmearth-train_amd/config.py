"""Static description of one FCMAE configuration (sizes, modalities, loss options).

Mirrors the bookkeeping of the reference constructor (/root/reference/models/fcmae.py:30-151)
and its size factories (:459-496) as plain data, so that the oracle, the HIP engine and the
synthetic-input generator agree on shapes and ordering.
"""
from argparse import Namespace
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import List, Tuple

from . import MODALITIES as M

# name -> (depths, dims)   (/root/reference/models/fcmae.py:459-496)
SIZES = {
    "convnextv2_atto": ([2, 2, 6, 2], [40, 80, 160, 320]),
    "convnextv2_femto": ([2, 2, 6, 2], [48, 96, 192, 384]),
    "convnextv2_pico": ([2, 2, 6, 2], [64, 128, 256, 512]),
    "convnextv2_nano": ([2, 2, 8, 2], [80, 160, 320, 640]),
    "convnextv2_tiny": ([3, 3, 9, 3], [96, 192, 384, 768]),
    "convnextv2_base": ([3, 3, 27, 3], [128, 256, 512, 1024]),
    "convnextv2_large": ([3, 3, 27, 3], [192, 384, 768, 1536]),
    "convnextv2_huge": ([3, 3, 27, 3], [352, 704, 1408, 2816]),
}

PIX_CONT = ("sentinel2", "sentinel1", "aster", "canopy_height_eth")
PIX_CAT = ("dynamic_world", "esa_worldcover")
IMG_CAT = ("biome", "eco_region")
IMG_CONT = ("lat", "lon", "month", "era5")


def out_channels(modality: str, modalities: dict, modalities_full: dict) -> int:
    """Channel bookkeeping of /root/reference/models/fcmae.py:65-91."""
    if modality in M.NUM_CLASSES:
        return M.NUM_CLASSES[modality]
    v = modalities[modality]
    return len(modalities_full[modality]) if v == "all" else len(v)


@dataclass
class OutMod:
    name: str
    kind: str          # "pix_cont" | "pix_cat" | "img_cat" | "img_cont"
    chans: int         # out_chans[name]
    head_out: int      # head output width: p*p*chans for pixel heads, chans for image heads
    tgt_chans: int     # channels of the target tensor (1 for class maps)


@dataclass
class ModelCfg:
    name: str = "convnextv2_atto"
    depths: List[int] = field(default_factory=lambda: [2, 2, 6, 2])
    dims: List[int] = field(default_factory=lambda: [40, 80, 160, 320])
    img_size: int = 56
    patch_size: int = 8
    in_chans: int = 12
    decoder_embed_dim: int = 512
    decoder_depth: int = 1
    mask_ratio: float = 0.6
    norm_pix_loss: bool = True
    loss_aggr: str = "uncertainty"
    out_mods: List[OutMod] = field(default_factory=list)
    # False: the dense ConvNeXtV2 encoder (fcmae.py:103-111, models/convnextv2.py:58-199) - every patch is computed, masked pixels are
    # zeroed at the input only; state dict in nn.Conv2d / nn.Linear layouts. Its stem (valid 3x3 convolution, depthwise k = stride =
    # patch / 8 with padding k // 2) only lines up with the decoder's patch grid when patch_size == 16 (convnextv2.py:108-124).
    sparse: bool = True
    # True: the original ConvNeXtV2 patchify stem instead of initial_conv + depthwise stem (convnextv2_sparse.py:99-110,202-203;
    # convnextv2.py:97-106,161-162): ONE convolution k = stride = patch / 8, 12 -> C0, + LayerNorm
    use_orig_stem: bool = False

    # ---- derived sizes ----------------------------------------------------
    @property
    def grid(self) -> int:          # patches per side (7)
        return self.img_size // self.patch_size

    @property
    def num_patches(self) -> int:   # L = 49
        return self.grid * self.grid

    def len_keep(self, mask_ratio=None) -> int:   # fcmae.py:217
        r = self.mask_ratio if mask_ratio is None else mask_ratio
        return int(self.num_patches * (1 - r))

    @property
    def stem_k(self) -> int:        # depthwise stem kernel = stride = patch/8 (convnextv2_sparse.py:123-124)
        return self.patch_size // 8

    def pts_side(self, stage: int) -> int:
        """points per patch side at encoder stage `stage` (0..3): 8,4,2,1."""
        return 8 >> stage

    @property
    def pix_mods(self) -> List[OutMod]:
        return [m for m in self.out_mods if m.kind.startswith("pix")]

    @property
    def img_mods(self) -> List[OutMod]:
        return [m for m in self.out_mods if m.kind.startswith("img")]


def kind_of(name: str) -> str:
    if name in PIX_CONT:
        return "pix_cont"
    if name in PIX_CAT:
        return "pix_cat"
    if name in IMG_CAT:
        return "img_cat"
    if name in IMG_CONT:
        return "img_cont"
    raise KeyError(f"unsupported output modality {name!r}")


def make_cfg(model="convnextv2_atto", img_size=56, patch_size=8, out_modalities=None,
             inp_modalities=None, modalities_full=None, norm_pix_loss=True,
             loss_aggr="uncertainty", mask_ratio=0.6, decoder_embed_dim=512,
             decoder_depth=1, sparse=True, use_orig_stem=False) -> ModelCfg:
    depths, dims = SIZES[model]
    out_modalities = OrderedDict(M.OUT_MODALITIES if out_modalities is None else out_modalities)
    inp_modalities = OrderedDict(M.INP_MODALITIES if inp_modalities is None else inp_modalities)
    modalities_full = M.MODALITIES_FULL if modalities_full is None else modalities_full
    mods = OrderedDict(inp_modalities)
    mods.update(out_modalities)
    assert patch_size % 8 == 0 and img_size % patch_size == 0
    assert decoder_depth >= 1
    if not sparse and patch_size != 16:
        # (110 + 2 (k // 2) - k) // k + 1 == img / 8 only for k = 2: at patch 8 the reference's dense encoder returns a 6 x 6 map for
        # a 7 x 7 mask and fails in forward_decoder (fcmae.py:253); its own test runs the dense mode at 112 / 16 (tests/pretrain_test.py:17)
        raise ValueError("sparse=False needs patch_size == 16 (the dense stem of models/convnextv2.py:108-124 yields img/8 points "
                         "per side only for the 2x2 stride-2 depthwise stem)")
    in_chans = out_channels("sentinel2", mods, modalities_full)
    oms = []
    for name in out_modalities:
        kind = kind_of(name)
        ch = out_channels(name, mods, modalities_full)
        head = patch_size * patch_size * ch if kind.startswith("pix") else ch
        tgt = 1 if kind == "pix_cat" else ch
        oms.append(OutMod(name, kind, ch, head, tgt))
    return ModelCfg(name=model, depths=list(depths), dims=list(dims), img_size=img_size,
                    patch_size=patch_size, in_chans=in_chans, decoder_embed_dim=decoder_embed_dim,
                    decoder_depth=decoder_depth, mask_ratio=mask_ratio, norm_pix_loss=norm_pix_loss,
                    loss_aggr=loss_aggr, out_mods=oms, sparse=bool(sparse), use_orig_stem=bool(use_orig_stem))


def cfg_from_args(model, img_size, patch_size, args: Namespace, norm_pix_loss, mask_ratio,
                  decoder_embed_dim=512, decoder_depth=1, sparse=True) -> ModelCfg:
    """Build a ModelCfg from the reference-style `args` namespace (needs .modalities,
    .out_modalities, .modalities_full, .loss_aggr; /root/reference/models/fcmae.py:65-91,102,126,406)."""
    inp = OrderedDict((k, v) for k, v in args.modalities.items() if k not in args.out_modalities
                      or k == "sentinel2")
    inp = OrderedDict(sentinel2=args.modalities["sentinel2"])
    return make_cfg(model, img_size, patch_size, out_modalities=args.out_modalities,
                    inp_modalities=inp, modalities_full=args.modalities_full,
                    norm_pix_loss=norm_pix_loss, loss_aggr=args.loss_aggr, mask_ratio=mask_ratio,
                    decoder_embed_dim=decoder_embed_dim, decoder_depth=decoder_depth, sparse=sparse,
                    use_orig_stem=bool(getattr(args, "use_orig_stem", False)))


def default_args(out_modalities=None, loss_aggr="uncertainty", use_orig_stem=False) -> Namespace:
    """The `args` namespace main_pretrain.py builds (/root/reference/main_pretrain.py:175-180)."""
    out = OrderedDict(M.OUT_MODALITIES if out_modalities is None else out_modalities)
    mods = OrderedDict(M.INP_MODALITIES)
    mods.update(out)
    return Namespace(inp_modalities=OrderedDict(M.INP_MODALITIES), out_modalities=out,
                     modalities=mods, modalities_full=M.MODALITIES_FULL,
                     use_orig_stem=use_orig_stem, loss_aggr=loss_aggr)
