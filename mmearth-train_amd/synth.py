"""Deterministic synthetic weights and input tiles.

Weights are filled tensor-by-tensor, in state-dict order, from a seeded CPU generator with
scales that exercise every term (non-zero GRN gamma/beta, biases, log_vars) — NOT the
reference's init — so parity tests cannot pass by an identity GRN or zero bias.
Inputs follow the §8d recipe: S2 ~N(0,1); S1 with NaN bands; canopy with 10 % NaN pixels;
class maps with -1 no-data; era5 with 5 % NaN; one-hot biome / eco_region.
"""
from collections import OrderedDict

import torch

from .config import ModelCfg


def state_dict_spec(cfg: ModelCfg):
    """Ordered (key, shape, kind) list in the reference's state-dict layout
    (/root/reference/helpers.py:668-707 for the kernel layouts).
    Decoder-block tensors are listed ONCE under the first output modality; the reference
    state dict holds T aliases of the same tensors (fcmae.py:119-121,137,145)."""
    spec = []
    C = cfg.dims
    k = cfg.stem_k
    add = spec.append
    orig = bool(getattr(cfg, "use_orig_stem", False))
    if getattr(cfg, "sparse", True):
        if orig:      # convnextv2_sparse.py:99-110: MinkowskiConvolution k = s = patch / 8 (kernel (Cin, C0) when k = 1, else (k*k, Cin, C0)) + LN
            add(("encoder.stem_orig.0.kernel", (cfg.in_chans, C[0]) if k == 1 else (k * k, cfg.in_chans, C[0]), "w"))
            add(("encoder.stem_orig.0.bias", (1, C[0]), "b"))
            add(("encoder.stem_orig.1.ln.weight", (C[0],), "g"))
            add(("encoder.stem_orig.1.ln.bias", (C[0],), "b"))
        else:
            add(("encoder.initial_conv.0.kernel", (9, cfg.in_chans, C[0]), "w"))
            add(("encoder.initial_conv.0.bias", (1, C[0]), "b"))
            add(("encoder.initial_conv.1.ln.weight", (C[0],), "g"))
            add(("encoder.initial_conv.1.ln.bias", (C[0],), "b"))
            add(("encoder.stem.0.kernel", (k * k, C[0]), "w"))
            add(("encoder.stem.0.bias", (1, C[0]), "b"))
            add(("encoder.stem.1.ln.weight", (C[0],), "g"))
            add(("encoder.stem.1.ln.bias", (C[0],), "b"))
        for i in range(3):
            p = f"encoder.downsample_layers.{i}"
            add((p + ".0.ln.weight", (C[i],), "g"))
            add((p + ".0.ln.bias", (C[i],), "b"))
            add((p + ".1.kernel", (4, C[i], C[i + 1]), "w"))
            add((p + ".1.bias", (1, C[i + 1]), "b"))
        for i in range(4):
            for j in range(cfg.depths[i]):
                p = f"encoder.stages.{i}.{j}"
                add((p + ".dwconv.kernel", (49, C[i]), "w"))
                add((p + ".dwconv.bias", (1, C[i]), "b"))
                add((p + ".norm.ln.weight", (C[i],), "g"))
                add((p + ".norm.ln.bias", (C[i],), "b"))
                add((p + ".pwconv1.linear.weight", (4 * C[i], C[i]), "w"))
                add((p + ".pwconv1.linear.bias", (4 * C[i],), "b"))
                add((p + ".pwconv2.linear.weight", (C[i], 4 * C[i]), "w"))
                add((p + ".pwconv2.linear.bias", (C[i],), "b"))
                add((p + ".grn.gamma", (1, 4 * C[i]), "gb"))
                add((p + ".grn.beta", (1, 4 * C[i]), "gb"))
    else:
        # dense ConvNeXtV2 (models/convnextv2.py:97-155): nn.Conv2d / nn.Linear / norm_layers.LayerNorm, GRN layouts; `norm` and `head`
        # (:151-152) are part of its state dict but not of the pretraining graph (they never receive a gradient)
        if orig:      # convnextv2.py:97-106: nn.Conv2d k = s = patch / 8 + channels-first LayerNorm
            add(("encoder.stem_orig.0.weight", (C[0], cfg.in_chans, k, k), "w"))
            add(("encoder.stem_orig.0.bias", (C[0],), "b"))
            add(("encoder.stem_orig.1.weight", (C[0],), "g"))
            add(("encoder.stem_orig.1.bias", (C[0],), "b"))
        else:
            add(("encoder.initial_conv.0.weight", (C[0], cfg.in_chans, 3, 3), "w"))
            add(("encoder.initial_conv.0.bias", (C[0],), "b"))
            add(("encoder.initial_conv.1.weight", (C[0],), "g"))
            add(("encoder.initial_conv.1.bias", (C[0],), "b"))
            add(("encoder.stem.0.weight", (C[0], 1, k, k), "w"))
            add(("encoder.stem.0.bias", (C[0],), "b"))
            add(("encoder.stem.1.weight", (C[0],), "g"))
            add(("encoder.stem.1.bias", (C[0],), "b"))
        for i in range(3):
            p = f"encoder.downsample_layers.{i}"
            add((p + ".0.weight", (C[i],), "g"))
            add((p + ".0.bias", (C[i],), "b"))
            add((p + ".1.weight", (C[i + 1], C[i], 2, 2), "w"))
            add((p + ".1.bias", (C[i + 1],), "b"))
        for i in range(4):
            for j in range(cfg.depths[i]):
                p = f"encoder.stages.{i}.{j}"
                add((p + ".dwconv.weight", (C[i], 1, 7, 7), "w"))
                add((p + ".dwconv.bias", (C[i],), "b"))
                add((p + ".norm.weight", (C[i],), "g"))
                add((p + ".norm.bias", (C[i],), "b"))
                add((p + ".pwconv1.weight", (4 * C[i], C[i]), "w"))
                add((p + ".pwconv1.bias", (4 * C[i],), "b"))
                add((p + ".grn.gamma", (1, 1, 1, 4 * C[i]), "gb"))
                add((p + ".grn.beta", (1, 1, 1, 4 * C[i]), "gb"))
                add((p + ".pwconv2.weight", (C[i], 4 * C[i]), "w"))
                add((p + ".pwconv2.bias", (C[i],), "b"))
        add(("encoder.norm.weight", (C[3],), "g"))
        add(("encoder.norm.bias", (C[3],), "b"))
        add(("encoder.head.weight", (1000, C[3]), "w"))
        add(("encoder.head.bias", (1000,), "b"))
    D = cfg.decoder_embed_dim
    add(("proj.weight", (D, C[3], 1, 1), "w"))
    add(("proj.bias", (D,), "b"))
    add(("mask_token", (1, D, 1, 1), "b"))
    first = cfg.out_mods[0].name
    for j in range(cfg.decoder_depth):      # nn.Sequential of decoder_depth Blocks (fcmae.py:119-121)
        p = f"decoder_dict.{first}.{j}"
        add((p + ".dwconv.weight", (D, 1, 7, 7), "w"))
        add((p + ".dwconv.bias", (D,), "b"))
        add((p + ".norm.weight", (D,), "g"))
        add((p + ".norm.bias", (D,), "b"))
        add((p + ".pwconv1.weight", (4 * D, D), "w"))
        add((p + ".pwconv1.bias", (4 * D,), "b"))
        add((p + ".grn.gamma", (1, 1, 1, 4 * D), "gb"))
        add((p + ".grn.beta", (1, 1, 1, 4 * D), "gb"))
        add((p + ".pwconv2.weight", (D, 4 * D), "w"))
        add((p + ".pwconv2.bias", (D,), "b"))
    for m in cfg.out_mods:
        if m.kind.startswith("pix"):
            add((f"pred_dict.{m.name}.weight", (m.head_out, D, 1, 1), "w"))
        else:
            add((f"pred_dict.{m.name}.weight", (m.head_out, D), "w"))
        add((f"pred_dict.{m.name}.bias", (m.head_out,), "b"))
    if cfg.img_mods:
        add(("layer_norm_tmp.weight", (D,), "g"))
        add(("layer_norm_tmp.bias", (D,), "b"))
    if cfg.loss_aggr == "uncertainty":
        add(("loss_fn.log_vars", (len(cfg.out_mods),), "lv"))
    return spec


def _dense_conv_key(key):
    return key in ("encoder.initial_conv.0.weight", "encoder.stem_orig.0.weight") or (key.startswith("encoder.downsample_layers.") and key.endswith(".1.weight"))


def param_view(buf, key, shape):
    """View of a parameter's slice of a flat buffer in the reference's shape. The convolution weights of the DENSE encoder are STORED
    kernel-offset-major, [(kw*k + kh)][Cin][Cout] / [(kw*k + kh)][C] - the MinkowskiEngine layout of the sparse encoder, which is what
    the engine's kernels read (helpers.py:676-688 is the reference's own mapping between the two) - and are exposed here as the
    permuted, non-contiguous view with nn.Conv2d's [Cout, Cin, kh, kw] / [C, 1, kh, kw] indexing. Everything else is stored as shaped."""
    if _dense_conv_key(key):
        co, ci, k, _ = shape
        return buf.view(k, k, ci, co).permute(3, 2, 1, 0)
    if key == "encoder.stem.0.weight":
        c0, _, k, _ = shape
        return buf.view(k, k, c0).permute(2, 1, 0).unsqueeze(1)
    return buf.view(shape)


def dense_aliases(cfg):
    """(engine-internal key, state-dict key, internal shape) for the dense encoder's stem and downsampling layers: the engine's launch
    program addresses them under the sparse encoder's names and layouts (same storage, see param_view)."""
    C, k = cfg.dims, cfg.stem_k
    if getattr(cfg, "use_orig_stem", False):
        out = [("encoder.stem_orig.0.kernel", "encoder.stem_orig.0.weight", (k * k, cfg.in_chans, C[0])),
               ("encoder.stem_orig.1.ln.weight", "encoder.stem_orig.1.weight", (C[0],)),
               ("encoder.stem_orig.1.ln.bias", "encoder.stem_orig.1.bias", (C[0],))]
    else:
        out = [("encoder.initial_conv.0.kernel", "encoder.initial_conv.0.weight", (9, cfg.in_chans, C[0])),
               ("encoder.initial_conv.1.ln.weight", "encoder.initial_conv.1.weight", (C[0],)),
               ("encoder.initial_conv.1.ln.bias", "encoder.initial_conv.1.bias", (C[0],)),
               ("encoder.stem.0.kernel", "encoder.stem.0.weight", (k * k, C[0])),
               ("encoder.stem.1.ln.weight", "encoder.stem.1.weight", (C[0],)),
               ("encoder.stem.1.ln.bias", "encoder.stem.1.bias", (C[0],))]
    for i in range(3):
        p = f"encoder.downsample_layers.{i}"
        out += [(p + ".0.ln.weight", p + ".0.weight", (C[i],)), (p + ".0.ln.bias", p + ".0.bias", (C[i],)),
                (p + ".1.kernel", p + ".1.weight", (4, C[i], C[i + 1]))]
    return out


def _fan_in(key, shape):
    if key.endswith("dwconv.kernel") or key.endswith("stem.0.kernel"):
        return shape[0]
    if key.endswith("dwconv.weight"):
        return 49
    if key == "encoder.stem_orig.0.kernel":
        return shape[0] if len(shape) == 2 else shape[0] * shape[1]
    if key.endswith(".kernel"):
        return shape[0] * shape[1]
    n = 1
    for s in shape[1:]:
        n *= s
    return max(n, 1)


def make_state_dict(cfg: ModelCfg, seed: int = 0, dtype=torch.float32) -> "OrderedDict[str, torch.Tensor]":
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for key, shape, kind in state_dict_spec(cfg):
        if kind == "w":
            t = torch.randn(shape, generator=g) * (1.0 / _fan_in(key, shape)) ** 0.5
        elif kind == "g":      # LN weight around 1
            t = 1.0 + 0.2 * torch.randn(shape, generator=g)
        elif kind == "gb":     # GRN gamma / beta: non-zero so the layer is not the identity
            t = 0.3 * torch.randn(shape, generator=g)
        elif kind == "lv":
            t = 0.5 * torch.randn(shape, generator=g)
        else:                  # biases, mask token
            t = 0.1 * torch.randn(shape, generator=g)
        sd[key] = t.to(dtype)
    return sd


def expand_aliases(cfg: ModelCfg, sd):
    """Reference-layout state dict: replicate the shared decoder block under every output
    modality (aliases of the same tensors, fcmae.py:137,145)."""
    first = cfg.out_mods[0].name
    out = OrderedDict()
    pre = f"decoder_dict.{first}."
    for k, v in sd.items():
        if k.startswith(pre):
            continue
        out[k] = v
    # reference order: encoder..., proj, (mask_token first as Parameter), decoder_dict.*, pred_dict.*
    for m in cfg.out_mods:
        for k, v in sd.items():
            if k.startswith(pre):
                out[f"decoder_dict.{m.name}." + k[len(pre):]] = v
    return out


def make_inputs(cfg: ModelCfg, N: int, seed: int = 1000, clean: bool = False):
    """Synthetic batch (already cropped to img_size) + mask noise. CPU tensors."""
    g = torch.Generator().manual_seed(seed)
    S = cfg.img_size
    names = {m.name for m in cfg.out_mods} | {"sentinel2"}
    d = OrderedDict()
    d["sentinel2"] = torch.randn(N, cfg.in_chans, S, S, generator=g)
    if "sentinel1" in names:
        x = torch.randn(N, 8, S, S, generator=g)
        if not clean:
            x[:, 2:4] = float("nan")
            x[:, 6:8] = float("nan")
            half = torch.rand(N, generator=g) < 0.5
            x[half, 4:8] = float("nan")
        d["sentinel1"] = x
    if "aster" in names:
        d["aster"] = torch.randn(N, 2, S, S, generator=g)
    if "era5" in names:
        x = torch.randn(N, 12, generator=g)
        if not clean:
            x[torch.rand(N, 12, generator=g) < 0.05] = float("nan")
        d["era5"] = x
    if "dynamic_world" in names:
        d["dynamic_world"] = torch.randint(-1, 9, (N, 1, S, S), generator=g)
    if "canopy_height_eth" in names:
        x = torch.randn(N, 2, S, S, generator=g)
        if not clean:
            nanpix = torch.rand(N, 1, S, S, generator=g) < 0.10
            x = torch.where(nanpix.expand_as(x), torch.full_like(x, float("nan")), x)
        d["canopy_height_eth"] = x
    for nm in ("lat", "lon"):
        if nm in names:
            d[nm] = torch.randn(N, 2, generator=g)
    if "biome" in names:
        d["biome"] = torch.nn.functional.one_hot(torch.randint(0, 14, (N,), generator=g), 14)
    if "eco_region" in names:
        d["eco_region"] = torch.nn.functional.one_hot(torch.randint(0, 846, (N,), generator=g), 846)
    if "month" in names:
        d["month"] = torch.randn(N, 2, generator=g)
    if "esa_worldcover" in names:
        d["esa_worldcover"] = torch.randint(-1, 11, (N, 1, S, S), generator=g)
    noise = torch.randn(N, cfg.num_patches, generator=g)
    # dict order = reference OUT_MODALITIES order where possible
    order = ["sentinel2"] + [m.name for m in cfg.out_mods if m.name != "sentinel2"]
    d = OrderedDict((k, d[k]) for k in order)
    return d, noise


def flat_param_spec(cfg):
    """Order of the tensors inside the flat fp32 parameter / gradient buffers: state-dict order, except that
    the prediction heads are regrouped as [pixel weights][pixel biases][image weights][image biases] in
    prediction-column order, so that the weight gradient of ALL pixel heads (and of all image heads) is one
    contiguous [W, D] matrix = one weight-gradient launch instead of one per modality. Everything else
    addresses parameters by key (state dicts keep the reference's order)."""
    spec = state_dict_spec(cfg)
    pred = [e for e in spec if e[0].startswith("pred_dict.")]
    if not pred:
        return spec
    by_key = {e[0]: e for e in pred}
    grouped = []
    for mods in (cfg.pix_mods, cfg.img_mods):
        for suffix in ("weight", "bias"):
            grouped += [by_key.pop(f"pred_dict.{m.name}.{suffix}") for m in mods if f"pred_dict.{m.name}.{suffix}" in by_key]
    grouped += list(by_key.values())
    first = next(i for i, e in enumerate(spec) if e[0].startswith("pred_dict."))
    rest = [e for e in spec if not e[0].startswith("pred_dict.")]
    return rest[:first] + grouped + rest[first:]
