"""CPU ORACLE for the MP-MAE pretraining step — TEST INFRASTRUCTURE, NOT PRODUCT.

A pure-PyTorch fp32 (or fp64) CPU restatement of the reference's algorithm for the hot
path, written in the "dense map + activity mask" formulation so that it shares no code
shape with the HIP path (which works on compacted row lists). Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.

Pinning: checked against golden vectors produced by the reference's OWN modules
(/root/reference/models/fcmae.py, models/convnextv2_sparse.py, models/sparse_norm_layers.py,
models/norm_layers.py, models/convnextv2.py Block and - for FCMAE(sparse=False) - ConvNeXtV2, custom_loss.py) run in the build container
(tests/golden/make_golden.py, tests/test_oracle_golden.py). The arithmetic of the absent
third-party MinkowskiEngine (un-pinned submodule) is restated from its published behaviour and
`helpers.remap_checkpoint_keys` — PARITY AT THE MinkowskiEngine BOUNDARY IS UNPINNED.

Every function cites the reference file:line it follows (paths relative to /root/reference).
All functions are differentiable (torch autograd) so the same module is the gradient oracle.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# masks
# ----------------------------------------------------------------------------
def gen_random_mask(noise: torch.Tensor, len_keep: int) -> torch.Tensor:
    """models/fcmae.py:214-231. noise [N,L] -> mask [N,L] f32, 0 = keep, 1 = remove.
    mask[n,l] = 1 iff rank(noise[n,l]) >= len_keep (two argsorts + gather in the reference)."""
    ids_shuffle = torch.argsort(noise, dim=1, stable=True)
    ids_restore = torch.argsort(ids_shuffle, dim=1, stable=True)
    N, L = noise.shape
    mask = torch.ones(N, L, dtype=torch.float32)
    mask[:, :len_keep] = 0
    return torch.gather(mask, 1, ids_restore)


def upsample_mask(mask: torch.Tensor, scale: int) -> torch.Tensor:
    """models/convnextv2_sparse.py:182-189."""
    p = int(round(mask.shape[1] ** 0.5))
    return mask.reshape(-1, p, p).repeat_interleave(scale, 1).repeat_interleave(scale, 2)


# ----------------------------------------------------------------------------
# MinkowskiEngine semantics on dense maps (see module docstring: unpinned boundary)
# ----------------------------------------------------------------------------
def _me_conv_weight(K: torch.Tensor, ks: int) -> torch.Tensor:
    """ME kernel (ks*ks, Cin, Cout) -> dense W[o,i,kh,kw] = K[kw*ks+kh, i, o]
    (helpers.py:676-683)."""
    kv, cin, cout = K.reshape(ks * ks, K.shape[-2], K.shape[-1]).shape
    return K.reshape(kv, cin, cout).permute(2, 1, 0).reshape(cout, cin, ks, ks).transpose(3, 2)


def _me_dw_weight(K: torch.Tensor, ks: int) -> torch.Tensor:
    """ME depthwise kernel (ks*ks, C) -> W[c,0,kh,kw] = K[kw*ks+kh, c] (helpers.py:684-688)."""
    C = K.shape[1]
    return K.permute(1, 0).reshape(C, 1, ks, ks).transpose(3, 2)


def _ln_map(x, act, w, b, eps=1e-6):
    """MinkowskiLayerNorm (sparse_norm_layers.py:61-77): nn.LayerNorm over C at active sites."""
    y = F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), w, b, eps).permute(0, 3, 1, 2)
    return y * act


def _linear_map(x, act, w, b):
    """MinkowskiLinear on feature rows (convnextv2_sparse.py:41,43)."""
    y = F.linear(x.permute(0, 2, 3, 1), w, b).permute(0, 3, 1, 2)
    return y * act


def _grn_sparse(x, act, gamma, beta):
    """MinkowskiGRN (sparse_norm_layers.py:24-33): L2 norm over ALL active rows of the batch
    (dim 0 of the feature matrix), eps 1e-6."""
    Gx = torch.sqrt((x * x).sum(dim=(0, 2, 3), keepdim=True))          # [1,C,1,1]
    Nx = Gx / (Gx.mean(dim=1, keepdim=True) + 1e-6)
    g = gamma.reshape(1, -1, 1, 1)
    b = beta.reshape(1, -1, 1, 1)
    return (g * (x * Nx) + b + x) * act


def sparse_block(sd, pre, x, act):
    """Sparse ConvNeXtV2 Block (convnextv2_sparse.py:47-56)."""
    C = x.shape[1]
    d = F.conv2d(x, _me_dw_weight(sd[pre + ".dwconv.kernel"], 7), padding=3, groups=C)
    d = (d + sd[pre + ".dwconv.bias"].reshape(1, C, 1, 1)) * act
    xn = _ln_map(d, act, sd[pre + ".norm.ln.weight"], sd[pre + ".norm.ln.bias"])
    h = _linear_map(xn, act, sd[pre + ".pwconv1.linear.weight"], sd[pre + ".pwconv1.linear.bias"])
    g = F.gelu(h) * act
    z = _grn_sparse(g, act, sd[pre + ".grn.gamma"], sd[pre + ".grn.beta"])
    y = _linear_map(z, act, sd[pre + ".pwconv2.linear.weight"], sd[pre + ".pwconv2.linear.bias"])
    return x + y


def sparse_encoder(sd, imgs, mask, cfg, taps=None):
    """SparseConvNeXtV2.forward (convnextv2_sparse.py:191-220), default stem or use_orig_stem.
    imgs [N,Cin,S,S] (NOT modified), mask [N,L] -> dense [N,C3,grid,grid]."""
    L = mask.shape[1]
    scale = int(cfg.img_size // (L ** 0.5))
    up = upsample_mask(mask, scale).unsqueeze(1).type_as(imgs)
    x = imgs * (1.0 - up)                                        # :197 (in place in the reference)
    act = (x.abs().sum(1, keepdim=True) != 0).type_as(x)         # to_sparse (:199)
    C = cfg.dims
    k = cfg.stem_k
    if getattr(cfg, "use_orig_stem", False):
        # stem_orig (:99-110, forward :202-203): MinkowskiConvolution k = s = patch / 8, Cin -> C0, + bias at the active OUTPUT sites
        # (k = 1: same sites; k > 1: a site is active iff any of its k x k children is) -> LN
        w = _me_conv_weight(sd["encoder.stem_orig.0.kernel"], k)
        x = F.conv2d(x, w, stride=k) + sd["encoder.stem_orig.0.bias"].reshape(1, C[0], 1, 1)
        if k > 1:
            act = F.max_pool2d(act, k)
        x = x * act
        x = _ln_map(x, act, sd["encoder.stem_orig.1.ln.weight"], sd["encoder.stem_orig.1.ln.bias"])
    else:
        # initial_conv (:113-119): 3x3 s1 + bias -> LN -> GELU
        w = _me_conv_weight(sd["encoder.initial_conv.0.kernel"], 3)
        x = (F.conv2d(x, w, padding=1) + sd["encoder.initial_conv.0.bias"].reshape(1, C[0], 1, 1)) * act
        x = _ln_map(x, act, sd["encoder.initial_conv.1.ln.weight"], sd["encoder.initial_conv.1.ln.bias"])
        x = F.gelu(x) * act
        # stem (:121-130): depthwise k = s = patch/8 + bias -> LN
        wd = _me_dw_weight(sd["encoder.stem.0.kernel"].reshape(k * k, C[0]), k)
        x = F.conv2d(x, wd, stride=k, groups=C[0]) + sd["encoder.stem.0.bias"].reshape(1, C[0], 1, 1)
        if k > 1:
            act = F.max_pool2d(act, k)
        x = x * act
        x = _ln_map(x, act, sd["encoder.stem.1.ln.weight"], sd["encoder.stem.1.ln.bias"])
    if taps is not None:
        taps["stem_out"] = x
    for i in range(4):
        if i > 0:
            p = f"encoder.downsample_layers.{i - 1}"
            x = _ln_map(x, act, sd[p + ".0.ln.weight"], sd[p + ".0.ln.bias"])
            w = _me_conv_weight(sd[p + ".1.kernel"], 2)
            x = F.conv2d(x, w, stride=2) + sd[p + ".1.bias"].reshape(1, C[i], 1, 1)
            act = F.max_pool2d(act, 2)
            x = x * act
        for j in range(cfg.depths[i]):
            x = sparse_block(sd, f"encoder.stages.{i}.{j}", x, act)
        if taps is not None:
            taps[f"stage{i}_out"] = x
    return x                                                     # .dense()[0] (:218): zeros elsewhere


# ----------------------------------------------------------------------------
# dense encoder (FCMAE(sparse=False))
# ----------------------------------------------------------------------------
def _ln_cf(x, w, b, eps=1e-6):
    """norm_layers.LayerNorm, data_format="channels_first" (norm_layers.py:26-31): biased variance over C."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[:, None, None] * x + b[:, None, None]


def dense_encoder(sd, imgs, mask, cfg, taps=None):
    """ConvNeXtV2.forward with a mask (models/convnextv2.py:183-199): masked pixels are zeroed at the input and nothing else is
    masked - every point of every map is computed. Stem (:108-124): VALID 3x3 convolution (S - 2 points per side) -> LN -> GELU,
    depthwise k = stride = patch / 8 with padding k // 2 -> LN; downsampling (:126-131): LN -> 2x2 stride-2 convolution; the blocks
    are the dense Block (:42-55) with its per-sample GRN. imgs [N,Cin,S,S] (NOT modified) -> [N,C3,grid,grid]."""
    L = mask.shape[1]
    scale = int(cfg.img_size // (L ** 0.5))
    up = upsample_mask(mask, scale).unsqueeze(1).type_as(imgs)
    x = imgs * (1.0 - up)                                        # :190 (in place in the reference)
    C = cfg.dims
    k = cfg.stem_k
    if getattr(cfg, "use_orig_stem", False):      # convnextv2.py:97-106,193-194: Conv2d k = s = patch / 8 (no padding) -> channels-first LN
        x = F.conv2d(x, sd["encoder.stem_orig.0.weight"], sd["encoder.stem_orig.0.bias"], stride=k)
        x = _ln_cf(x, sd["encoder.stem_orig.1.weight"], sd["encoder.stem_orig.1.bias"])
    else:
        x = F.conv2d(x, sd["encoder.initial_conv.0.weight"], sd["encoder.initial_conv.0.bias"])
        x = F.gelu(_ln_cf(x, sd["encoder.initial_conv.1.weight"], sd["encoder.initial_conv.1.bias"]))
        x = F.conv2d(x, sd["encoder.stem.0.weight"], sd["encoder.stem.0.bias"], stride=k, padding=k // 2, groups=C[0])
        x = _ln_cf(x, sd["encoder.stem.1.weight"], sd["encoder.stem.1.bias"])
    if taps is not None:
        taps["stem_out"] = x
    for i in range(4):
        if i > 0:
            p = f"encoder.downsample_layers.{i - 1}"
            x = _ln_cf(x, sd[p + ".0.weight"], sd[p + ".0.bias"])
            x = F.conv2d(x, sd[p + ".1.weight"], sd[p + ".1.bias"], stride=2)
        for j in range(cfg.depths[i]):
            x = dense_block(sd, f"encoder.stages.{i}.{j}", x)
        if taps is not None:
            taps[f"stage{i}_out"] = x
    return x


# ----------------------------------------------------------------------------
# decoder
# ----------------------------------------------------------------------------
def dense_block(sd, pre, x):
    """Dense ConvNeXtV2 Block (models/convnextv2.py:42-55) with LayerNorm channels_last
    (norm_layers.py:23-25) and per-sample GRN eps 1e-4 (norm_layers.py:41-44)."""
    C = x.shape[1]
    d = F.conv2d(x, sd[pre + ".dwconv.weight"], sd[pre + ".dwconv.bias"], padding=3, groups=C)
    t = d.permute(0, 2, 3, 1)
    t = F.layer_norm(t, (C,), sd[pre + ".norm.weight"], sd[pre + ".norm.bias"], 1e-6)
    t = F.linear(t, sd[pre + ".pwconv1.weight"], sd[pre + ".pwconv1.bias"])
    t = F.gelu(t)
    Gx = torch.sqrt((t * t).sum(dim=(1, 2), keepdim=True))
    Nx = Gx / (Gx.mean(dim=-1, keepdim=True) + 1e-4)
    t = sd[pre + ".grn.gamma"] * (t * Nx) + sd[pre + ".grn.beta"] + t
    t = F.linear(t, sd[pre + ".pwconv2.weight"], sd[pre + ".pwconv2.bias"])
    return x + t.permute(0, 3, 1, 2)


def decoder(sd, x, mask, cfg, taps=None):
    """FCMAE.forward_decoder (models/fcmae.py:249-265). The per-modality decoders are the SAME
    Block objects (fcmae.py:119-121,137,145) fed the same input, so the block is evaluated once."""
    x = F.conv2d(x, sd["proj.weight"], sd["proj.bias"])
    n, c, h, w = x.shape
    m = mask.reshape(-1, h, w).unsqueeze(1).type_as(x)
    x = x * (1.0 - m) + sd["mask_token"] * m
    if taps is not None:
        taps["dec_in"] = x
    first = cfg.out_mods[0].name
    y = x
    for j in range(getattr(cfg, "decoder_depth", 1)):      # nn.Sequential of decoder_depth Blocks (fcmae.py:119-121)
        y = dense_block(sd, f"decoder_dict.{first}.{j}", y)
    if taps is not None:
        taps["dec_out"] = y
    pred = OrderedDict()
    pooled = None
    for om in cfg.out_mods:
        W, b = sd[f"pred_dict.{om.name}.weight"], sd[f"pred_dict.{om.name}.bias"]
        if om.kind.startswith("pix"):
            pred[om.name] = F.conv2d(y, W, b)
        else:
            if pooled is None:
                # LayerNorm channels_first (norm_layers.py:26-31), then GAP (fcmae.py:259-262)
                u = y.mean(1, keepdim=True)
                s = (y - u).pow(2).mean(1, keepdim=True)
                t = (y - u) / torch.sqrt(s + 1e-6)
                t = sd["layer_norm_tmp.weight"][:, None, None] * t + sd["layer_norm_tmp.bias"][:, None, None]
                pooled = t.mean(dim=[-2, -1])
            pred[om.name] = F.linear(pooled, W, b)
    return pred


# ----------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------
def patchify(imgs, p, channels):
    """FCMAE.patchify (models/fcmae.py:180-197): [N,C,H,W] -> [N, L, p*p*C], j = (ph*p+pw)*C + c."""
    N = imgs.shape[0]
    h = w = imgs.shape[2] // p
    x = imgs.reshape(N, channels, h, p, w, p)
    x = torch.einsum("nchpwq->nhwpqc", x)
    return x.reshape(N, h * w, p * p * channels)


def _pred_nlc(pred):
    """[N,C,h,w] -> [N,L,C] (fcmae.py:306-309, 369-372)."""
    n, c = pred.shape[:2]
    return pred.reshape(n, c, -1).permute(0, 2, 1)


def loss_pix_cont(pred, target, mask, p, norm_pix):
    """Continuous pixel modalities (fcmae.py:366-403): NaN-aware per-patch MSE, masked patches
    only, denominator = count_nonzero(loss*mask)."""
    pr = _pred_nlc(pred)
    t = patchify(target, p, target.shape[1])
    if norm_pix:
        mean = t.mean(dim=-1, keepdim=True)
        var = t.var(dim=-1, keepdim=True)                 # unbiased (torch default)
        t = (t - mean) / (var + 1.0e-6) ** 0.5
    e = (pr - t) ** 2
    nan = torch.isnan(e)
    cnt = (~nan).sum(-1)
    e = torch.where(nan, torch.zeros_like(e), e)
    l = e.sum(-1) / cnt                                    # 0/0 -> NaN
    q = l * mask
    q = torch.where(torch.isnan(q), torch.zeros_like(q), q)
    return q.sum() / torch.count_nonzero(q)


def loss_pix_cat(pred, target, mask, p, K):
    """Categorical pixel modalities (fcmae.py:302-346): CE over masked patches, target != -1.
    Head channel = pix*K + k with pix = ph*p + pw."""
    pr = _pred_nlc(pred)                                   # [N,L,p*p*K]
    N, L, _ = pr.shape
    pr = pr.reshape(N, L * p * p, K)
    t = patchify(target, p, 1).reshape(N, L * p * p)
    m = mask.unsqueeze(-1).repeat(1, 1, p * p).reshape(N, L * p * p)
    sel = (m == 1) & (t != -1)
    return F.cross_entropy(pr[sel], t[sel].long())


def loss_img_cat(pred, onehot):
    """fcmae.py:282-290."""
    return F.cross_entropy(pred, torch.argmax(onehot, dim=-1))


def loss_img_cont(pred, target):
    """fcmae.py:291-301: MSE over non-NaN target elements."""
    ok = ~torch.isnan(target)
    return F.mse_loss(pred[ok], target[ok])


def uncertainty_weighting(losses, log_vars):
    """UncertaintyWeightingStrategy.forward (custom_loss.py:19-30)."""
    lt = torch.stack(list(losses))
    nz = lt != 0.0
    return (torch.exp(-log_vars) * lt + log_vars) * nz


def forward_loss(sd, imgs_dict, preds, mask, cfg):
    """FCMAE.forward_loss (models/fcmae.py:267-412)."""
    p = cfg.patch_size
    loss_dict = OrderedDict()
    for om in cfg.out_mods:
        pr, tg = preds[om.name], imgs_dict[om.name]
        if om.kind == "img_cat":
            loss_dict[om.name] = loss_img_cat(pr, tg)
        elif om.kind == "img_cont":
            loss_dict[om.name] = loss_img_cont(pr, tg)
        elif om.kind == "pix_cat":
            loss_dict[om.name] = loss_pix_cat(pr, tg, mask, p, om.chans)
        else:
            loss_dict[om.name] = loss_pix_cont(
                pr, tg, mask, p, cfg.norm_pix_loss and om.name == "sentinel2")
    ll = list(loss_dict.values())
    if cfg.loss_aggr == "uncertainty":
        w = uncertainty_weighting(ll, sd["loss_fn.log_vars"])
        return w.sum(), loss_dict, sd["loss_fn.log_vars"].tolist(), w
    return sum(ll), loss_dict, None, None


# ----------------------------------------------------------------------------
# whole step
# ----------------------------------------------------------------------------
def forward(sd, imgs_dict, noise, cfg, mask_ratio=None, taps=None):
    """FCMAE.forward (models/fcmae.py:414-456) on already-cropped inputs with explicit mask
    noise. Returns (loss, pred, mask, loss_dict, log_vars, weighted)."""
    d = OrderedDict((k, v) for k, v in imgs_dict.items())
    imgs = d["sentinel2"]                                   # :439 (before nan_to_num)
    for k in list(d.keys()):
        if k in ("sentinel2", "sentinel1", "aster", "canopy_height_eth"):
            d[k] = torch.nan_to_num(d[k], nan=0.0, posinf=0.0, neginf=0.0)   # :445-449
    mask = gen_random_mask(noise, cfg.len_keep(mask_ratio))
    x = (sparse_encoder if getattr(cfg, "sparse", True) else dense_encoder)(sd, imgs, mask, cfg, taps)      # fcmae.py:94-111
    if taps is not None:
        taps["enc_out"] = x
    pred = decoder(sd, x, mask, cfg, taps)
    loss, loss_dict, log_vars, weighted = forward_loss(sd, d, pred, mask, cfg)
    return loss, pred, mask, loss_dict, log_vars, weighted


def step_grads(sd, imgs_dict, noise, cfg, mask_ratio=None, dtype=torch.float32):
    """Forward + backward; returns (outputs, grads dict keyed like sd)."""
    p = OrderedDict((k, v.detach().to(dtype).clone().requires_grad_(True)) for k, v in sd.items())
    d = OrderedDict((k, (v.to(dtype) if v.is_floating_point() else v)) for k, v in imgs_dict.items())
    out = forward(p, d, noise, cfg, mask_ratio)
    out[0].backward()
    grads = OrderedDict((k, (v.grad if v.grad is not None else torch.zeros_like(v))) for k, v in p.items())
    return out, grads


def adjust_learning_rate(epoch, lr, min_lr, warmup_epochs, epochs):
    """helpers.py:647-665 (warm-up then half-cycle cosine)."""
    if epoch < warmup_epochs:
        return lr * epoch / warmup_epochs
    return min_lr + (lr - min_lr) * 0.5 * (
        1.0 + math.cos(math.pi * (epoch - warmup_epochs) / (epochs - warmup_epochs)))


def adamw_step(param, grad, m, v, step, lr, beta1=0.9, beta2=0.95, eps=1e-8, wd=0.05):
    """torch.optim.AdamW single-tensor update (main_pretrain.py:320: betas (0.9, 0.95))."""
    param = param * (1 - lr * wd)
    m = beta1 * m + (1 - beta1) * grad
    v = beta2 * v + (1 - beta2) * grad * grad
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)) + eps
    return param - (lr / bc1) * m / denom, m, v
