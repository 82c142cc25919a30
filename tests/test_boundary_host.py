"""CPU tests of the drop-in boundary around the hot path: checkpoint contract (save / resume layout of the
fused AdamW, `remap_checkpoint_keys` consumer, `hubconf.MPMAE`), init distributions, the step runner's
gradient-accumulation / bucket ordering logic (world 1 and world 8 over gloo, on a recording fake engine),
and - when /root/reference is present (build container only) - the reference's own dense ConvNeXtV2 as a
second witness for the MinkowskiEngine kernel-offset conventions the oracle's emulator assumes."""
import math
import os
from argparse import Namespace
from collections import OrderedDict

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from mmearth_train_amd import _lib
    return _lib.load()


def _model(device="cpu", subset="all_mod", **kw):
    from mmearth_train_amd import MODALITIES as MM
    from mmearth_train_amd import fcmae
    from mmearth_train_amd.config import default_args
    from mmearth_train_amd.custom_loss import UncertaintyWeightingStrategy
    args = default_args(out_modalities=MM.subset(subset))
    T = len(args.out_modalities)
    return fcmae.convnextv2_atto(mask_ratio=0.6, decoder_depth=1, decoder_embed_dim=512, norm_pix_loss=True,
                                 patch_size=8, img_size=56, args=args, loss_fn=UncertaintyWeightingStrategy(T),
                                 sparse=True, device=device, **kw)


# ----------------------------------------------------------------------------- init
def test_init_reference_distributions(lib):
    """fcmae.py:157-178 as it ends up after FCMAE.apply(_init_weights) overrides the encoder's own init:
    ME depthwise / MinkowskiLinear / nn.Conv2d weights ~ trunc_normal(std 1, +-2) (measured std ~0.88),
    ME conv kernels and nn.Linear std 0.02, biases / GRN 0, LayerNorm 1/0, mask_token N(0, 0.02), log_vars 0."""
    torch.manual_seed(0)
    sd = _model().state_dict()
    tn1 = 0.8796                      # std of N(0,1) truncated at +-2
    def std(k): return sd[k].float().std().item()
    for k in ["encoder.stages.0.0.dwconv.kernel", "encoder.stages.2.3.pwconv1.linear.weight",
              "encoder.stages.3.1.pwconv2.linear.weight", "proj.weight", "decoder_dict.sentinel2.0.dwconv.weight",
              "pred_dict.sentinel2.weight", "pred_dict.esa_worldcover.weight"]:
        assert abs(std(k) - tn1) < 0.05 * tn1 + 2.0 / math.sqrt(sd[k].numel()), (k, std(k))
        assert sd[k].abs().max() <= 2.0
    assert abs(std("encoder.stem.0.kernel") - tn1) < 0.35           # 40 values only
    for k in ["encoder.initial_conv.0.kernel", "encoder.downsample_layers.1.1.kernel",
              "decoder_dict.sentinel2.0.pwconv1.weight", "decoder_dict.sentinel2.0.pwconv2.weight",
              "pred_dict.eco_region.weight", "pred_dict.era5.weight"]:
        assert abs(std(k) - 0.02) < 0.002 + 0.04 / math.sqrt(sd[k].numel()), (k, std(k))
        assert sd[k].abs().max() <= 2.0
    assert abs(std("mask_token") - 0.02) < 0.004
    for k, v in sd.items():
        if k.endswith("bias") or k.endswith(".beta") or k.endswith(".gamma") or k == "loss_fn.log_vars":
            assert (v == 0).all(), k
        if k.endswith("ln.weight") or k.endswith("norm.weight") or k == "layer_norm_tmp.weight":
            assert (v == 1).all(), k


def test_use_orig_stem_state_dict_layout():
    """args.use_orig_stem=True (convnextv2_sparse.py:99-110 / convnextv2.py:97-106; the last constructor option that raised until round 5):
    the encoder's stem tensors are `stem_orig.0.kernel` - (Cin, C0) at patch 8 where k = s = 1, (4, Cin, C0) at patch 16 - `.0.bias (1, C0)`
    and `.1.ln.*`, and there is no initial_conv / stem; the dense encoder carries nn.Conv2d / LayerNorm names and shapes. (That these
    are the reference's keys is pinned by tests/golden/make_golden.py, which loads them into the reference's own model with strict=True.)"""
    from mmearth_train_amd import MODALITIES as MM
    from mmearth_train_amd import fcmae
    from mmearth_train_amd.config import default_args
    from mmearth_train_amd.custom_loss import UncertaintyWeightingStrategy
    args = default_args(out_modalities=MM.subset("all_mod"), use_orig_stem=True)
    mk = lambda **kw: fcmae.convnextv2_atto(mask_ratio=0.6, decoder_depth=1, decoder_embed_dim=512, norm_pix_loss=True, args=args,
                                            loss_fn=UncertaintyWeightingStrategy(12), device="cpu", **kw)
    sd = mk(patch_size=8, img_size=56).state_dict()
    assert tuple(sd["encoder.stem_orig.0.kernel"].shape) == (12, 40) and tuple(sd["encoder.stem_orig.0.bias"].shape) == (1, 40)
    assert "encoder.stem_orig.1.ln.weight" in sd and not any(k.startswith(("encoder.initial_conv", "encoder.stem.")) for k in sd)
    sd = mk(patch_size=16, img_size=112).state_dict()
    assert tuple(sd["encoder.stem_orig.0.kernel"].shape) == (4, 12, 40)
    sd = mk(patch_size=16, img_size=112, sparse=False).state_dict()
    assert tuple(sd["encoder.stem_orig.0.weight"].shape) == (40, 12, 2, 2) and tuple(sd["encoder.stem_orig.1.weight"].shape) == (40,)
    assert not any(k.startswith(("encoder.initial_conv", "encoder.stem.")) for k in sd)


def test_dense_mode_state_dict_is_the_reference_layout_over_kernel_major_storage():
    """FCMAE(sparse=False) (fcmae.py:103-111): parameter names / shapes are those of the dense ConvNeXtV2 (convnextv2.py:97-155,
    incl. its unused `norm` / `head`), initialised as fcmae.py:157-178 leaves them; the convolution weights of the stem and the
    downsampling layers are nn.Conv2d-shaped VIEWS of kernel-offset-major storage (what the engine's kernels read): loading a
    reference-layout state dict and reading the flat buffer back must give helpers.py:676-688's layout. patch 8 is refused, as the
    reference's dense stem cannot serve it."""
    from mmearth_train_amd import MODALITIES as MM
    from mmearth_train_amd import fcmae
    from mmearth_train_amd.config import default_args, make_cfg
    from mmearth_train_amd.custom_loss import UncertaintyWeightingStrategy
    from mmearth_train_amd.synth import dense_aliases, expand_aliases, flat_param_spec, make_state_dict, state_dict_spec
    from oracle.mpmae_ref import _me_conv_weight, _me_dw_weight
    args = default_args(out_modalities=MM.subset("all_mod"))
    mk = lambda **kw: fcmae.convnextv2_atto(mask_ratio=0.6, decoder_depth=1, decoder_embed_dim=512, norm_pix_loss=True, args=args,
                                            loss_fn=UncertaintyWeightingStrategy(12), device="cpu", **kw)
    with pytest.raises(ValueError):
        mk(patch_size=8, img_size=56, sparse=False)
    torch.manual_seed(0)
    m = mk(patch_size=16, img_size=112, sparse=False)
    cfg = make_cfg("convnextv2_atto", 112, 16, out_modalities=MM.subset("all_mod"), sparse=False)
    sd = m.state_dict()
    want = expand_aliases(cfg, OrderedDict((k, s) for k, s, _ in state_dict_spec(cfg)))
    assert set(sd.keys()) == set(want.keys())
    for k, shape in want.items():
        assert tuple(sd[k].shape) == tuple(shape), k
    assert sd["encoder.head.weight"].shape == (1000, 320) and "encoder.norm.weight" in sd
    assert not any(".ln." in k or k.endswith(".kernel") or ".linear." in k for k in sd)
    for k, v in sd.items():      # init: 1-d weights are normalisation scales (1), Conv2d std 1 truncated, Linear std 0.02
        if k.endswith(".weight") and v.dim() == 1:
            assert (v == 1).all(), k
    assert abs(sd["encoder.downsample_layers.1.1.weight"].std().item() - 0.8796) < 0.05
    assert abs(sd["encoder.stages.2.0.pwconv1.weight"].std().item() - 0.02) < 0.002
    # load a seeded reference-layout state dict; the flat storage is kernel-offset-major
    src = make_state_dict(cfg, seed=5)
    m.load_state_dict(expand_aliases(cfg, src), strict=True)
    offs, off = {}, 0
    for key, shape, _ in flat_param_spec(cfg):
        offs[key] = (off, math.prod(shape))
        off += math.prod(shape)
    for akey, key, ashape in dense_aliases(cfg):
        o, n = offs[key]
        stored = m._pflat[o:o + n].view(ashape)
        if key.endswith("conv.0.weight") or key.endswith(".1.weight") and len(ashape) == 3:
            assert torch.equal(_me_conv_weight(stored, int(round(ashape[0] ** 0.5))), src[key]), key
        elif key == "encoder.stem.0.weight":
            assert torch.equal(_me_dw_weight(stored, 2), src[key])
        else:
            assert torch.equal(stored.reshape(-1), src[key].reshape(-1)), key
    back = m.state_dict()
    for k, v in expand_aliases(cfg, src).items():
        assert torch.equal(back[k], v), k


def test_init_matches_reference_statistics():
    from oracle.refharness import load_reference as LR
    if not LR.available():
        pytest.skip("reference tree not present")
    ref = LR.load()
    from mmearth_train_amd import MODALITIES as MM
    from mmearth_train_amd.config import default_args
    args = default_args(out_modalities=MM.subset("all_mod"))
    torch.manual_seed(1)
    rm = ref.fcmae.convnextv2_atto(mask_ratio=0.6, decoder_depth=1, decoder_embed_dim=512, norm_pix_loss=True,
                                   patch_size=8, img_size=56, args=args,
                                   loss_fn=ref.custom_loss.UncertaintyWeightingStrategy(12), sparse=True)
    rsd = rm.state_dict()
    torch.manual_seed(2)
    sd = _model().state_dict()
    assert list(rsd.keys()) == list(sd.keys())                       # same keys in the same order
    assert [n for n, _ in rm.named_parameters()] == [n for n, _ in _model().named_parameters()]   # optimizer-state indices line up
    for k in sd:
        a, b = sd[k].float(), rsd[k].float()
        assert a.shape == b.shape, k
        if a.numel() >= 2000:                                         # same distribution family and scale
            assert abs(a.std().item() - b.std().item()) <= 0.06 * max(b.std().item(), 1e-3), k
            assert abs(a.mean().item() - b.mean().item()) <= 5.0 * b.std().item() / math.sqrt(a.numel()) + 1e-6, k
        elif b.std().item() == 0 or a.numel() == 1:                    # constants: equal
            assert torch.equal(a, b), k


# ----------------------------------------------------------------------------- checkpoint consumer (§8f-2, b6)
def test_remap_and_hub_consume_a_checkpoint_written_here(lib, tmp_path):
    from mmearth_train_amd.helpers import remap_checkpoint_keys
    import hubconf
    torch.manual_seed(3)
    m = _model()
    with torch.no_grad():
        for k, p in m.named_parameters():
            if "grn" in k or k.endswith("bias"):
                p.normal_(0, 0.3)
    sd = OrderedDict((k, v.detach().clone()) for k, v in m.state_dict().items())
    path = tmp_path / "checkpoint-0.pth"
    torch.save({"model": sd, "optimizer": None, "epoch": 0, "scaler": {}, "args": Namespace()}, path)
    dense = hubconf.MPMAE("convnextv2_atto", ckpt_name=str(path), pretrained=True, linear_probe=True, num_classes=7)
    dsd = dense.state_dict()
    # kernel layout rule (helpers.py:676-687): W[o, i, kh, kw] = K[kw*ks + kh, i, o]; depthwise W[c, 0, kh, kw] = K[kw*ks + kh, c]
    K = sd["encoder.initial_conv.0.kernel"]
    W = dsd["initial_conv.0.weight"]
    for kh in range(3):
        for kw in range(3):
            assert torch.equal(W[:, :, kh, kw], K[kw * 3 + kh].t())
    Kd = sd["encoder.stages.1.1.dwconv.kernel"]
    Wd = dsd["stages.1.1.dwconv.weight"]
    assert torch.equal(Wd[:, 0, 2, 5], Kd[5 * 7 + 2])
    Kdn = sd["encoder.downsample_layers.2.1.kernel"]
    assert torch.equal(dsd["downsample_layers.2.1.weight"][:, :, 1, 0], Kdn[0 * 2 + 1].t())
    assert torch.equal(dsd["stages.2.4.pwconv1.weight"], sd["encoder.stages.2.4.pwconv1.linear.weight"])
    assert torch.equal(dsd["stages.2.4.grn.gamma"].reshape(-1), sd["encoder.stages.2.4.grn.gamma"].reshape(-1))
    assert tuple(dsd["stages.2.4.grn.gamma"].shape) == (1, 1, 1, 640)
    assert torch.equal(dsd["downsample_layers.0.0.weight"], sd["encoder.downsample_layers.0.0.ln.weight"])
    assert dsd["stem.0.bias"].dim() == 1 and torch.equal(dsd["stem.0.bias"], sd["encoder.stem.0.bias"].reshape(-1))
    # every encoder tensor found a home; only the final norm / classifier are new
    remapped = remap_checkpoint_keys(OrderedDict((k, v) for k, v in sd.items() if k.startswith("encoder.")))
    assert set(remapped) == set(dsd) - {"norm.weight", "norm.bias", "head.weight", "head.bias"}
    assert tuple(dense(torch.randn(2, 12, 56, 56)).shape) == (2, 7)
    with pytest.raises(ValueError):
        hubconf.MPMAE(ckpt_name="no-such-checkpoint")
    golden = __import__("tests.golden_cases", fromlist=["load_fixture"]).load_fixture("misc")
    assert sorted(remapped.keys()) == sorted(str(k) for k in golden["remap_keys"])   # == the reference's remap output


def test_remap_and_dense_net_equal_the_reference(tmp_path):
    """Same checkpoint through the reference's remap_checkpoint_keys + dense ConvNeXtV2 and through ours."""
    from oracle.refharness import load_reference as LR
    if not LR.available():
        pytest.skip("reference tree not present")
    ref = LR.load()
    import importlib
    rcv = importlib.import_module("reference.models.convnextv2")
    from mmearth_train_amd import convnextv2 as cv
    from mmearth_train_amd.helpers import remap_checkpoint_keys
    from mmearth_train_amd.config import make_cfg
    from mmearth_train_amd.synth import make_state_dict
    cfg = make_cfg("convnextv2_atto", 112, 16)
    sd = make_state_dict(cfg, seed=5)
    enc = OrderedDict((k, v) for k, v in sd.items() if k.startswith("encoder."))
    a, b = remap_checkpoint_keys(enc), ref.helpers.remap_checkpoint_keys(enc)
    assert list(a.keys()) == list(b.keys())
    for k in a:
        assert torch.equal(a[k], b[k]), k
    mine = cv.convnextv2_atto(patch_size=16, img_size=112, num_classes=5)
    theirs = rcv.convnextv2_atto(patch_size=16, img_size=112, num_classes=5)
    assert list(mine.state_dict().keys()) == list(theirs.state_dict().keys())
    mine.load_state_dict(a, strict=False)
    theirs.load_state_dict(b, strict=False)
    theirs.head.load_state_dict(mine.head.state_dict())
    theirs.norm.load_state_dict(mine.norm.state_dict())
    x = torch.randn(2, 12, 112, 112, generator=torch.Generator().manual_seed(6))
    ya, yb = mine(x), theirs(x)
    assert torch.allclose(ya, yb, rtol=1e-4, atol=1e-5)


def test_me_conventions_agree_with_the_reference_dense_encoder():
    """Narrows the unpinned MinkowskiEngine boundary with in-tree evidence: ONE seeded weight
    set through (i) the reference's sparse encoder on the oracle's ME emulator with nothing masked and (ii) the
    reference's dense ConvNeXtV2 fed through the reference's remap_checkpoint_keys must agree in the interior of
    the stage-0 -> downsample-0 output at 112/16. That fixes - from the reference's own code - the offset order
    of the k = s convolutions (2x2/2 stem and downsample), the 3x3 / 7x7 tap order and the (1, C) bias handling.
    GRN is made the identity (gamma = beta = 0) in stage 0: the dense and sparse GRN differ in eps and the dense
    map's border rows differ (no padding in the dense 3x3), and GRN would spread that over the whole map."""
    from oracle.refharness import load_reference as LR
    if not LR.available():
        pytest.skip("reference tree not present")
    ref = LR.load()
    import importlib
    rcv = importlib.import_module("reference.models.convnextv2")
    from mmearth_train_amd import MODALITIES as MM
    from mmearth_train_amd.config import default_args, make_cfg
    from mmearth_train_amd.synth import expand_aliases, make_state_dict
    cfg = make_cfg("convnextv2_atto", 112, 16, out_modalities=MM.subset("S2"))
    sd = make_state_dict(cfg, seed=9)
    for k in sd:
        if k.startswith("encoder.stages.0.") and ".grn." in k:
            sd[k] = torch.zeros_like(sd[k])
    args = default_args(out_modalities=MM.subset("S2"))
    sparse = ref.fcmae.convnextv2_atto(mask_ratio=0.6, decoder_depth=1, decoder_embed_dim=512, norm_pix_loss=True,
                                       patch_size=16, img_size=112, args=args,
                                       loss_fn=ref.custom_loss.UncertaintyWeightingStrategy(1), sparse=True)
    sparse.load_state_dict(expand_aliases(cfg, sd), strict=True)
    dense = rcv.convnextv2_atto(patch_size=16, img_size=112, num_classes=3)
    enc = OrderedDict((k, v) for k, v in sd.items() if k.startswith("encoder."))
    dense.load_state_dict(ref.helpers.remap_checkpoint_keys(enc), strict=False)
    x = torch.randn(1, 12, 112, 112, generator=torch.Generator().manual_seed(10))
    taps = {}
    h1 = sparse.encoder.downsample_layers[0].register_forward_hook(lambda m, i, o: taps.__setitem__("s", o))
    h2 = dense.downsample_layers[0].register_forward_hook(lambda m, i, o: taps.__setitem__("d", o))
    h3 = sparse.encoder.stem.register_forward_hook(lambda m, i, o: taps.__setitem__("s_stem", o))
    h4 = dense.stem.register_forward_hook(lambda m, i, o: taps.__setitem__("d_stem", o))
    with torch.no_grad():
        sparse.encoder(x.clone(), torch.zeros(1, 49))
        dense(x.clone())
    for h in (h1, h2, h3, h4):
        h.remove()
    s_stem, d_stem = taps["s_stem"].dense()[0], taps["d_stem"]
    assert s_stem.shape == d_stem.shape == (1, 40, 56, 56)
    # stem output j covers sparse rows {2j, 2j+1} = dense (unpadded 3x3) rows {2j-1, 2j}: identical except the border ring
    assert torch.allclose(s_stem[..., 1:-1, 1:-1], d_stem[..., 1:-1, 1:-1], rtol=1e-4, atol=1e-5)
    assert not torch.allclose(s_stem[..., 0, :], d_stem[..., 0, :], atol=1e-3)      # the border really differs
    s, d = taps["s"].dense()[0], taps["d"]
    assert s.shape == d.shape == (1, 80, 28, 28)
    # border ring 1 at 56-res grows by 3 per 7x7 block (2 blocks) -> 7 rows -> 4 rows at 28-res
    assert torch.allclose(s[..., 4:-4, 4:-4], d[..., 4:-4, 4:-4], rtol=1e-4, atol=2e-5)
    assert (s[..., 4:-4, 4:-4].abs().mean() > 1e-2)


# ----------------------------------------------------------------------------- fused AdamW state <-> torch layout
class _FakeEngine:
    """Stands in for engine.Engine in the runner logic tests: records the order of launches and fills the flat
    gradient buffer the way the backward segments do (CPU tensors, no HIP)."""

    def __init__(self, rank=0, n=1200):
        self.device = torch.device("cpu")
        self.n_params = n
        self.offsets = OrderedDict([("encoder.initial_conv.0.kernel", (0, 200)), ("encoder.stages.2.0.dwconv.kernel", (200, 300)),
                                    ("encoder.stages.3.0.dwconv.kernel", (500, 100)), ("proj.weight", (600, 300)),
                                    ("pred_dict.sentinel2.weight", (900, 300))])
        self.gflat = torch.zeros(n)
        self.pflat = torch.zeros(n)
        self.mflat, self.vflat = torch.zeros(n), torch.zeros(n)
        self.total = torch.zeros(1)
        self.rank, self.log, self.k = rank, [], 0
        self.bwd_ops = [("dloss", None, (), {}), ("head:pix.wgrad", None, (), {}), ("decoder_dict.sentinel2.0:x", None, (), {}),
                        ("proj.wgrad", None, (), {}), ("encoder.stages.3.0:x", None, (), {}),
                        ("encoder.stages.2.0:x", None, (), {}), ("encoder.downsample_layers.1:x", None, (), {}),
                        ("encoder.stages.0.0:x", None, (), {})]
        self._scale = 1.0

    # launches
    def forward(self):
        self.log.append("fwd"); self.k += 1
        self.total.fill_(float(self.k))

    def finalize_loss(self, st, dlv, scale):
        self._scale = scale

    def _stream(self):
        return None

    def _run(self, ops, st):
        for name, *_ in ops:
            self.log.append(name)
            lo, hi = {"head:pix.wgrad": (900, 1200), "proj.wgrad": (600, 900), "encoder.stages.3.0:x": (500, 600),
                      "encoder.stages.2.0:x": (200, 500), "encoder.stages.0.0:x": (0, 200)}.get(name, (0, 0))
            self.gflat[lo:hi] += self._scale * (self.rank + 1) * self.k

    def set_hyper(self, lr, t, grad_scale=1.0):
        self._hp = (lr, t, grad_scale)

    def launch_adamw(self, wd, note=True, guard_loss=None):
        self.guard_loss = guard_loss                 # data parallel: the all-reduced loss buffer (same skip decision on every rank)
        self.log.append("adamw")
        self.pflat -= self._hp[0] * self._hp[2] * self.gflat

    def note_optimizer_launch(self):
        pass


def test_runner_accumulates_over_update_freq_micro_steps():
    from mmearth_train_amd import dist as mdist
    eng = _FakeEngine()
    run = mdist.StepRunner(eng, world_size=1, lr=0.5, mode="eager", update_freq=3)
    done = [run.step() for _ in range(6)]
    assert done == [False, False, True, False, False, True] and run.t == 2
    # window 2 = micro-steps k = 4, 5, 6, each contributing (1/3) * k; the buffer was zeroed at the window start
    assert torch.allclose(eng.gflat[600:], torch.full((600,), (4 + 5 + 6) / 3.0))
    assert eng.log.count("adamw") == 2 and eng.log.count("fwd") == 6
    assert torch.allclose(eng.pflat[:200], torch.full((200,), -0.5 * ((1 + 2 + 3) / 3.0 + (4 + 5 + 6) / 3.0)))


def _w8(rank, world, port, q, wire=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from mmearth_train_amd import dist as mdist
    mdist.init(backend="gloo")
    eng = _FakeEngine(rank)
    run = mdist.StepRunner(eng, world_size=world, lr=1.0, mode="eager", update_freq=2, allreduce_dtype=wire)
    assert [b for b in run.buckets] == [(900, 1200), (600, 900), (200, 600), (0, 200)]
    assert [[o[0] for o in s] for s in run.segments] == [["dloss", "head:pix.wgrad"], ["decoder_dict.sentinel2.0:x", "proj.wgrad"],
                                                          ["encoder.stages.3.0:x", "encoder.stages.2.0:x"],
                                                          ["encoder.downsample_layers.1:x", "encoder.stages.0.0:x"]]
    run.step(); run.step()
    tot = sum(r + 1 for r in range(world))
    ok = torch.allclose(eng.gflat, torch.full((1200,), tot * (1 + 2) / 2.0))          # SUM over ranks of the window's gradients
    ok = ok and torch.allclose(eng.pflat, -eng.gflat / world)                         # 1/world folded into the optimizer
    ok = ok and abs(run.mean_loss() - 2.0) < 1e-6
    q.put((rank, bool(ok), eng.log))
    mdist.barrier()
    mdist.shutdown()


@pytest.mark.parametrize("wire,port", [(None, 29633), (torch.bfloat16, 29634)])
def test_gloo_world8_runner_ordering_and_exchange(wire, port):
    """8 ranks over gloo: segment / bucket order of the real StepRunner, gradient accumulation, the 1/world fold, the scalar loss
    mean - in fp32 and with the bf16 wire format (the sums below are small integers, exact in bf16)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_w8, args=(r, 8, port, q, wire)) for r in range(8)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    log = res[0][2]
    assert log == ["fwd", "dloss", "head:pix.wgrad", "decoder_dict.sentinel2.0:x", "proj.wgrad", "encoder.stages.3.0:x",
                   "encoder.stages.2.0:x", "encoder.downsample_layers.1:x", "encoder.stages.0.0:x"] * 2 + ["adamw"]


def test_fused_adamw_state_is_a_torch_adamw_state_dict(lib):
    from mmearth_train_amd import dist as mdist
    from mmearth_train_amd.helpers import fused_adamw_state_dict, load_fused_adamw_state_dict, param_groups_weight_decay
    m = _model(subset="S2")
    eng = m._get_engine(2, 0.6)
    run = mdist.StepRunner(eng, world_size=1, lr=3e-4, weight_decay=0.05, mode="eager")
    g = torch.Generator().manual_seed(4)
    eng.mflat.copy_(torch.randn(eng.n_params, generator=g))
    eng.vflat.copy_(torch.rand(eng.n_params, generator=g))
    run.t = 17
    sd = fused_adamw_state_dict(m, run)
    opt = torch.optim.AdamW(param_groups_weight_decay(m, 0.05), lr=1e-3, betas=(0.9, 0.95))
    opt.load_state_dict(sd)                                   # what helpers.auto_load_model does with the entry
    named = dict(m.named_parameters())
    for k in ["encoder.stages.1.0.pwconv1.linear.weight", "mask_token", "loss_fn.log_vars", "encoder.stages.0.0.dwconv.bias"]:
        st = opt.state[named[k]]
        off, n = eng.offsets[k]
        assert torch.equal(st["exp_avg"].reshape(-1), eng.mflat[off:off + n]) and float(st["step"]) == 17
    assert opt.param_groups[0]["weight_decay"] == 0.0 and opt.param_groups[1]["weight_decay"] == 0.05
    assert any(p is named["encoder.stages.0.0.grn.gamma"] for p in opt.param_groups[1]["params"])      # GRN affine IS decayed (timm rule)
    m0, v0 = eng.mflat.clone(), eng.vflat.clone()
    eng.mflat.zero_(); eng.vflat.zero_(); run.t = 0
    load_fused_adamw_state_dict(m, run, opt.state_dict())     # and back, from torch's own serialisation
    assert torch.equal(eng.mflat, m0) and torch.equal(eng.vflat, v0) and run.t == 17
