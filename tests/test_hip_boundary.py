"""GPU tests (-m gpu) of the drop-in boundary: FCMAE.forward_encoder / forward_decoder / forward_loss, call-time
mask_ratio, gradient accumulation (update_freq) on the fused path, the device-side non-finite guard, and
main_pretrain.main end to end (train, checkpoint, auto-resume == uninterrupted run, hub consumer)."""
import math
import os
import re
import subprocess
import sys
from collections import OrderedDict

import pytest
import torch

from tests.golden_cases import CASES, case_cfg, case_data
from tests.test_hip_parity import _engine, _oracle, _rel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _module(cfg, sd, subset="all_mod", dtype="f32"):
    from mmearth_train_amd import MODALITIES as MM
    from mmearth_train_amd import fcmae
    from mmearth_train_amd.config import default_args
    from mmearth_train_amd.custom_loss import UncertaintyWeightingStrategy
    from mmearth_train_amd.synth import expand_aliases
    args = default_args(out_modalities=MM.subset(subset))
    m = fcmae.convnextv2_atto(mask_ratio=0.6, decoder_depth=1, decoder_embed_dim=512, norm_pix_loss=True, patch_size=8,
                              img_size=56, args=args, loss_fn=UncertaintyWeightingStrategy(len(cfg.out_mods)),
                              sparse=True, device="cuda:0", dtype=dtype)
    m.load_state_dict(expand_aliases(cfg, sd), strict=True)
    return m


def test_forward_pieces_equal_forward_and_oracle():
    """forward_encoder -> forward_decoder -> forward_loss (fcmae.py:242-412) chained by the caller reproduce
    forward(): encoder map, every prediction, the 12 losses and the weighted total, all against the oracle."""
    c = CASES["allmod_atto_56"]
    cfg = case_cfg(c)
    sd, inputs, noise = case_data(c, cfg)
    model = _module(cfg, sd)
    dev = {k: v.to("cuda:0") for k, v in inputs.items()}
    torch.manual_seed(c["nseed"])
    x, mask = model.forward_encoder(dev["sentinel2"].clone(), 0.6)
    (oloss, opred, omask, oloss_dict, _, ow), taps, _ = _oracle(cfg, sd, inputs, model._engine.noise.cpu())
    assert torch.equal(mask.cpu(), omask)
    assert tuple(x.shape) == (2, 320, 7, 7)
    assert _rel(x, taps["enc_out"]) < 1e-4
    pred = model.forward_decoder(x, mask)
    for k, v in opred.items():
        assert tuple(pred[k].shape) == tuple(v.shape), k
        assert _rel(pred[k], v) < 2e-4, k
    # forward() hands forward_loss the nan_to_num'ed pixel targets (fcmae.py:445-449); a direct caller does the same
    clean = OrderedDict((k, torch.nan_to_num(v, nan=0.0, posinf=0.0, neginf=0.0) if k in
                         ("sentinel2", "sentinel1", "aster", "canopy_height_eth") else v) for k, v in inputs.items())
    dev = {k: v.to("cuda:0") for k, v in clean.items()}
    loss, loss_dict, log_vars, normalized = model.forward_loss(dev, pred, mask)
    assert abs(loss.item() - oloss.item()) <= 1e-4 * abs(oloss.item())
    for k, v in oloss_dict.items():
        assert abs(loss_dict[k].item() - v.item()) <= 1e-4 * abs(v.item()) + 1e-7, k
    assert len(log_vars) == 12 and torch.allclose(normalized.cpu(), ow.detach(), rtol=1e-4, atol=1e-6)
    # a caller-made prediction (not the decoder's) goes through forward_loss too: zeros -> the plain target energy
    zero = OrderedDict((k, torch.zeros_like(v)) for k, v in pred.items())
    l0, d0, _, _ = model.forward_loss(dev, zero, mask)
    op = OrderedDict((k, torch.zeros_like(v)) for k, v in opred.items())
    from oracle import mpmae_ref as O
    ref0 = O.forward_loss(OrderedDict((k, v.clone()) for k, v in sd.items()), clean, op, omask, cfg)
    assert abs(l0.item() - ref0[0].item()) <= 1e-4 * abs(ref0[0].item())


def test_dense_mode_module_pieces_and_training_step():
    """FCMAE(sparse=False) (fcmae.py:103-111; the mode of the reference's own tests/pretrain_test.py:17) through the module: the three
    forward pieces against the oracle's dense encoder, then model(...) + loss.backward(): every .grad in the reference's nn.Conv2d /
    nn.Linear shapes against the oracle's autograd (fp32 mode bounds), the classifier head untouched (no gradient in the reference)."""
    from mmearth_train_amd import MODALITIES as MM
    from mmearth_train_amd import fcmae
    from mmearth_train_amd.config import default_args
    from mmearth_train_amd.custom_loss import UncertaintyWeightingStrategy
    from mmearth_train_amd.synth import expand_aliases
    c = CASES["allmod_atto_112_dense"]
    cfg = case_cfg(c)
    sd, inputs, noise = case_data(c, cfg)
    args = default_args(out_modalities=MM.subset("all_mod"))
    model = fcmae.convnextv2_atto(mask_ratio=0.6, decoder_depth=1, decoder_embed_dim=512, norm_pix_loss=True, patch_size=16,
                                  img_size=112, args=args, loss_fn=UncertaintyWeightingStrategy(12), sparse=False,
                                  device="cuda:0", dtype="f32")
    model.load_state_dict(expand_aliases(cfg, sd), strict=True)
    dev = {k: v.to("cuda:0") for k, v in inputs.items()}
    x, mask = model.forward_encoder(dev["sentinel2"].clone(), 0.6)
    (oloss, opred, omask, oloss_dict, _, ow), taps, ograds = _oracle(cfg, sd, inputs, model._engine.noise.cpu())
    assert torch.equal(mask.cpu(), omask) and tuple(x.shape) == (2, 320, 7, 7)
    assert _rel(x, taps["enc_out"]) < 1e-4
    assert float(x.detach().abs().reshape(2, 320, 49).amax(1).min()) > 0          # masked patches are computed too (nothing is zero)
    pred = model.forward_decoder(x, mask)
    for k, v in opred.items():
        assert _rel(pred[k], v) < 2e-4, k
    # the training step on the same noise
    torch.manual_seed(123)                                # (forward() draws its own noise: read it back from the engine for the oracle)
    out = model(dev, mask_ratio=0.6)
    out[0].backward()
    nz = model._engine.noise.cpu()
    (oloss, _, omask, _, _, _), _, ograds = _oracle(cfg, sd, inputs, nz)
    assert torch.equal(out[2].cpu(), omask)
    assert abs(out[0].item() - oloss.item()) <= 1e-4 * abs(oloss.item())
    named = dict(model.named_parameters())
    for k, go in ograds.items():
        if k == "loss_fn.log_vars":
            p = model.loss_fn.log_vars
        else:
            p = named[k]
        assert tuple(p.shape) == tuple(go.shape), k
        if k.startswith("encoder.head.") or k.startswith("encoder.norm."):
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert (p.grad.cpu() - go).abs().max().item() <= 2e-4 * go.abs().max().item() + 1e-9, k


@pytest.mark.parametrize("ratio", [0.75, 0.3])
def test_call_time_mask_ratio_decides_len_keep(ratio):
    """fcmae.py:415,451 (quirk q9): the constructor's mask_ratio is only stored; forward(mask_ratio=r) keeps int(L(1-r))."""
    c = CASES["allmod_atto_56"]
    cfg = case_cfg(c)
    sd, inputs, noise = case_data(c, cfg)
    model = _module(cfg, sd)                               # constructed with 0.6
    torch.manual_seed(5)
    loss, pred, mask, loss_dict, _, _ = model({k: v.to("cuda:0") for k, v in inputs.items()}, mask_ratio=ratio)
    keep = int(49 * (1 - ratio))
    assert ((mask == 0).sum(1) == keep).all()
    from mmearth_train_amd.config import make_cfg
    import dataclasses
    ocfg = dataclasses.replace(cfg, mask_ratio=ratio)
    (oloss, opred, omask, _, _, _), _, grads = _oracle(ocfg, sd, inputs, model._engine.noise.cpu())
    assert torch.equal(mask.cpu(), omask)
    assert abs(loss.item() - oloss.item()) <= 1e-4 * abs(oloss.item())
    assert _rel(pred["sentinel2"], opred["sentinel2"]) < 2e-4
    loss.backward()
    g = dict(model.named_parameters())["encoder.stages.1.0.pwconv1.linear.weight"].grad
    assert _rel(g, grads["encoder.stages.1.0.pwconv1.linear.weight"]) < 2e-4
    assert len(model._engines) == 1 and (2, keep) in model._engines


@pytest.mark.parametrize("mode", ["program", "eager"])
def test_update_freq_accumulates_in_the_flat_gradient_buffer(mode):
    """--update_freq 2 on the fused path: two micro-steps on different batches, loss / 2 each, ONE AdamW."""
    from mmearth_train_amd import dist as mdist
    from mmearth_train_amd.synth import make_inputs
    c = CASES["allmod_atto_56"]
    cfg = case_cfg(c)
    sd, inputs, noise = case_data(c, cfg)
    inputs2, noise2 = make_inputs(cfg, c["N"], seed=77)
    ref = _engine(cfg, c["N"], "f32", sd, inputs, noise, lanes=False)
    ref.forward(loss_scale=0.5); ref.backward(zero_grad=True)
    ref.set_inputs(inputs2, noise2)
    ref.forward(loss_scale=0.5); ref.backward(zero_grad=False)
    torch.cuda.synchronize()
    g_ref = ref.gflat.clone()
    ref.optimizer_step(1e-3)
    eng = _engine(cfg, c["N"], "f32", sd, inputs, noise)
    run = mdist.StepRunner(eng, world_size=1, lr=1e-3, mode=mode, update_freq=2)
    p0 = eng.pflat.clone()
    assert run.step() is False
    torch.cuda.synchronize()
    assert torch.equal(eng.pflat, p0)                       # no update on the first micro-step
    eng.set_inputs(inputs2, noise2)
    assert run.step() is True
    torch.cuda.synchronize()
    assert _rel(eng.gflat, g_ref) < 2e-5
    assert _rel(eng.pflat, ref.pflat) < 2e-5 and not torch.equal(eng.pflat, p0)
    # next window starts from a zeroed buffer
    run.step(); torch.cuda.synchronize()
    ref.set_inputs(inputs2, noise2); ref.forward(loss_scale=0.5); ref.backward(zero_grad=True); torch.cuda.synchronize()
    assert _rel(eng.gflat, ref.gflat) < 1e-3               # (parameters differ by one AdamW step's rounding only)


def test_non_finite_loss_skips_the_update_on_the_device():
    """engine_pretrain.py:83-85 stops before the optimizer on a non-finite loss; the fused path has no host sync
    there, so the device guard (mpmae_hp_fetch -> hp[4]) must leave parameters and moments untouched."""
    from mmearth_train_amd import dist as mdist
    c = CASES["allmod_atto_56"]
    cfg = case_cfg(c)
    sd, inputs, noise = case_data(c, cfg)
    eng = _engine(cfg, c["N"], "bf16", sd, inputs, noise)
    run = mdist.StepRunner(eng, world_size=1, lr=1e-3, mode="program")
    run.step(); torch.cuda.synchronize()
    assert run.skipped_steps() == 0
    p1, m1, v1 = eng.pflat.clone(), eng.mflat.clone(), eng.vflat.clone()
    # a diverged run: one uncertainty weight overflowed -> exp(-s) L + s = inf. (A NaN/inf *pixel* does not do it in bf16
    # mode: the packed polynomial GELU clamps through v_med3_f32, which returns a finite value for a NaN input.)
    lv = eng.params["loss_fn.log_vars"]
    keep0 = lv[0].clone()
    lv[0] = float("inf")
    p1, m1, v1 = eng.pflat.clone(), eng.mflat.clone(), eng.vflat.clone()
    run.step(); torch.cuda.synchronize()
    assert not torch.isfinite(eng.total).all()
    assert run.skipped_steps() == 1
    assert torch.equal(eng.pflat, p1) and torch.equal(eng.mflat, m1) and torch.equal(eng.vflat, v1)
    lv[0] = keep0
    eng.set_inputs(inputs, noise)
    run.step(); torch.cuda.synchronize()
    assert torch.isfinite(eng.total).all() and torch.isfinite(eng.pflat).all() and not torch.equal(eng.pflat[:1000], p1[:1000])


def _main(argv):
    import main_pretrain
    return main_pretrain.main(main_pretrain.get_args_parser().parse_args(argv))


def test_main_pretrain_trains_checkpoints_and_auto_resumes(tmp_path):
    """main_pretrain.main end to end on synthetic tiles: 2 epochs in one go == 1 epoch + auto-resumed second epoch
    (model and fused-AdamW state), update_freq 2; the checkpoint loads into a stock torch.optim.AdamW and into hubconf.MPMAE."""
    common = ["--model", "convnextv2_atto", "--input_size", "56", "--patch_size", "8", "--batch_size", "4", "--update_freq", "2",
              "--steps_per_epoch", "4", "--lr", "1e-3", "--min_lr", "1e-3", "--warmup_epochs", "0", "--norm_pix_loss", "True",
              "--compute_dtype", "f32", "--seed", "3"]
    a, b = tmp_path / "a", tmp_path / "b"
    _main(common + ["--epochs", "2", "--output_dir", str(a)])
    _main(common + ["--epochs", "1", "--output_dir", str(b)])
    assert (b / "checkpoint-0.pth").exists()
    _main(common + ["--epochs", "2", "--output_dir", str(b), "--auto_resume", "True"])
    ca = torch.load(a / "checkpoint-1.pth", map_location="cpu", weights_only=False)
    cb = torch.load(b / "checkpoint-1.pth", map_location="cpu", weights_only=False)
    assert ca["epoch"] == cb["epoch"] == 1 and set(ca) == {"model", "optimizer", "epoch", "scaler", "args"}
    assert len(ca["model"]) == 290
    # two separate processes' worth of fp32 atomics (column sums, bias gradients) feed AdamW's g / sqrt(v) at steps 1-4: 1.3e-4 has been
    # observed between two identical runs, so the bound is 1e-3 (a missed resume of the moments or the step count shows up at > 1e-1)
    for k in ca["model"]:
        assert _rel(cb["model"][k], ca["model"][k]) < 1e-3, k
    sa, sb = ca["optimizer"]["state"], cb["optimizer"]["state"]
    assert len(sa) == len(sb) == 180 and float(sa[0]["step"]) == float(sb[0]["step"]) == 4.0      # 2 epochs x 4 micro-steps / 2
    for i in (0, 5, 100, 179):
        assert _rel(sb[i]["exp_avg"], sa[i]["exp_avg"]) < 3e-3
    c0 = torch.load(a / "checkpoint-0.pth", map_location="cpu", weights_only=False)
    assert _rel(c0["model"]["proj.weight"], ca["model"]["proj.weight"]) > 1e-4                    # it did train
    # consumers: torch AdamW built the reference's way (helpers.auto_load_model), and the hub entry point
    from mmearth_train_amd.helpers import param_groups_weight_decay
    from tests.test_boundary_host import _model
    m = _model(device="cuda:0", dtype="f32")
    m.load_state_dict(ca["model"], strict=True)
    opt = torch.optim.AdamW(param_groups_weight_decay(m, 0.05), lr=1e-3, betas=(0.9, 0.95))
    opt.load_state_dict(ca["optimizer"])
    import hubconf
    dense = hubconf.MPMAE("convnextv2_atto", ckpt_name=str(a / "checkpoint-1.pth"), num_classes=4)
    assert torch.equal(dense.state_dict()["stages.3.1.pwconv2.weight"], ca["model"]["encoder.stages.3.1.pwconv2.linear.weight"])


def test_main_pretrain_torch_autograd_path(tmp_path):
    """--fast_path False: FCMAE.forward + loss.backward() + torch.optim.AdamW (update_freq 2), one epoch."""
    stats = _main(["--model", "convnextv2_atto", "--input_size", "56", "--patch_size", "8", "--batch_size", "2", "--update_freq", "2",
                   "--steps_per_epoch", "4", "--epochs", "1", "--lr", "1e-3", "--norm_pix_loss", "True", "--fast_path", "False",
                   "--out_modalities", "pix_mod", "--output_dir", str(tmp_path)])
    assert torch.isfinite(torch.tensor(stats[0]["loss"]))
    ck = torch.load(tmp_path / "checkpoint-0.pth", map_location="cpu", weights_only=False)
    assert len(ck["optimizer"]["state"]) > 0


def test_two_rank_step_driver_with_update_freq_2(tmp_path):
    """world 2 x update_freq 2 through the real driver (two processes on cuda:0 over gloo): the exchange happens once
    per window, on the accumulated buffer; both ranks end equal and match the hand-averaged single-process run."""
    env = dict(os.environ, DDP_PROBE_UPDATE_FREQ="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ddp_probe.py"), str(tmp_path)], capture_output=True,
                       text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    for mode in ("program", "eager"):
        m = re.search(mode + r" step-1 averaged gradient vs reference: max rel ([0-9.e+-]+)", r.stdout)
        assert m and float(m.group(1)) < 1e-5, r.stdout
        assert re.search(mode + r" ranks equal: True", r.stdout), r.stdout


def test_chained_forward_pieces_refuse_backward_instead_of_yielding_no_gradients():
    """/root/reference/models/fcmae.py:242-412: forward_encoder -> forward_decoder -> forward_loss -> loss.backward() trains the
    reference model. The fused path runs those pieces as forward segments only: backward through them must RAISE (they
    used to return detached tensors, i.e. silently no gradients); forward() is the trainable entry point."""
    c = CASES["allmod_atto_56"]
    cfg = case_cfg(c)
    sd, inputs, noise = case_data(c, cfg)
    model = _module(cfg, sd)
    dev = {k: v.cuda() for k, v in inputs.items()}
    x, mask = model.forward_encoder(dev["sentinel2"].clone(), 0.6)
    preds = model.forward_decoder(x, mask)
    loss, loss_dict, log_vars, weighted = model.forward_loss(dev, preds, mask)
    assert loss.requires_grad and torch.isfinite(loss)
    with pytest.raises(RuntimeError, match="not connected to the fused backward"):
        loss.backward()
    with torch.no_grad():                      # inference use of the pieces is unchanged
        x2, _ = model.forward_encoder(dev["sentinel2"].clone(), 0.6)
    assert not x2.requires_grad
    loss2 = model({k: v.clone() for k, v in dev.items()}, mask_ratio=0.6)[0]
    loss2.backward()
    assert model.proj.weight.grad is not None and model.proj.weight.grad.abs().sum() > 0


def test_forward_decoder_sees_weights_changed_after_forward_encoder():
    """forward_decoder / forward_loss re-stage the weights ON THE MAIN LANE before their first GEMM (in the full program
    the staging runs on the side lane and only the stem waits for it). Changing proj.weight between the encoder and the decoder
    segment must change the predictions exactly as a fresh full forward with the new weights does."""
    c = CASES["allmod_atto_56"]
    cfg = case_cfg(c)
    sd, inputs, noise = case_data(c, cfg)
    for dtype in ("bf16", "fp8"):
        eng = _engine(cfg, c["N"], dtype, sd, inputs, noise)
        eng.forward()
        torch.cuda.synchronize()
        before = {k: v.float().clone() for k, v in eng.preds().items()}
        eng.run_segment("encoder")
        torch.cuda.synchronize()
        with torch.no_grad():
            eng.params["proj.weight"].mul_(1.5)
            eng.params["decoder_dict.sentinel2.0.pwconv1.weight"].mul_(0.5)
        eng.run_segment("decoder")
        torch.cuda.synchronize()
        got = {k: v.float().clone() for k, v in eng.preds().items()}
        eng.forward()                                   # full program with the changed weights
        torch.cuda.synchronize()
        # Two forwards of the SAME weights are not bit-identical: the GRN column sums fold through LDS float atomics whose order varies,
        # the fp32 sums differ in their last bits and now and then a bf16 (fp8) rounding downstream flips - measured run-to-run floor on
        # this 2-sample case: <= 1.5 % of max|pred| in bf16, <= 6.2 % with the MX-fp8 decoder. Stale weights are off by far more.
        tol = 4e-2 if dtype == "bf16" else 1.2e-1
        moved = 0.0
        for k, v in eng.preds().items():
            assert _rel(got[k], v.float()) < tol, (dtype, k)
            moved = max(moved, _rel(before[k], v.float()))
        assert moved > 3 * tol, (dtype, moved, "the weight change must move the predictions far beyond the tolerance")


def test_rccl_exchange_runs_at_world_size_one():
    """The "nccl" backend (RCCL) on a one-rank process group drives the bucketed gradient exchange of StepRunner (program events ->
    communication stream -> async all-reduce) in both drivers; gradients, parameters and the logged loss equal the no-exchange step."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_world1_probe.py")], capture_output=True, text=True,
                       timeout=900, cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    for mode in ("program", "eager"):
        m = re.search(mode + r": RCCL world-1 exchange vs no exchange after 2 steps: gradients max rel ([0-9.e+-]+)", r.stdout)
        assert m and float(m.group(1)) < 1e-5, r.stdout
        assert re.search(mode + r": exposed communication tail [0-9.]+ ms per step", r.stdout), r.stdout      # StepRunner.measure_comm_tail


def test_lanes_and_outside_streams_are_probed_for_shared_hardware_queues():
    """ROCm serialises HIP streams that share a hardware queue. The launch program probes its side lanes against the main stream at the
    first replay (and replaces a colliding lane); mpmae_program_stream_overlaps answers the same question for a stream outside the program.
    Whatever the runtime's queue assignment in this process: a stream never 'overlaps' with itself, the stream picked for the gradient
    exchange does overlap with the main stream and the lanes, and the program still steps after the probes."""
    import ctypes as C
    from mmearth_train_amd import dist as mdist
    c = CASES["allmod_atto_56"]
    cfg = case_cfg(c)
    sd, inputs, noise = case_data(c, cfg)
    eng = _engine(cfg, c["N"], "bf16", sd, inputs, noise)
    run = mdist.StepRunner(eng, world_size=1, lr=1e-4, mode="program")
    run.step()
    torch.cuda.synchronize()
    main = torch.cuda.current_stream()
    ms = C.c_void_p(main.cuda_stream)
    assert eng.lib.mpmae_program_stream_overlaps(run.prog, ms, ms) == 0
    s = mdist.pick_concurrent_stream(eng, run.prog)
    assert eng.lib.mpmae_program_stream_overlaps(run.prog, ms, C.c_void_p(s.cuda_stream)) == 1
    assert eng.lib.mpmae_program_stream_overlaps(None, ms, C.c_void_p(s.cuda_stream)) == 1
    run.step()                      # the program still runs after the probes (which synchronise and launch on its lanes)
    torch.cuda.synchronize()
    assert torch.isfinite(eng.total).all()


def test_bench_gpus_flag_starts_ranks_or_fails_loudly():
    """`python bench.py --gpus N` outside a launcher spawns N ranks itself; on a node with fewer GPUs it must refuse (exit code 2 and
    a message), not run one rank and print n_gpus: 1."""
    n = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env={k: v for k, v in os.environ.items() if k != "WORLD_SIZE"})
    assert r.returncode == 2 and f"--gpus {n + 1} requested but this node exposes {n} GPU" in r.stderr, (r.returncode, r.stderr[-500:])
    assert "n_gpus" not in r.stdout


def test_on_device_crop_matches_indexing_and_feeds_forward():
    """Input stage: 64x64 tiles from disk, aligned random 56x56 window per sample shared by all pixel-wise modalities
    (kornia RandomCrop, fcmae.py:419-434) cut by mpmae_crop into the engine's buffers - fp32 bands and int64 class maps - then the
    ordinary forward; checked against torch indexing and against the oracle on the cropped tiles the module hands back."""
    import ctypes as C
    from mmearth_train_amd import _lib
    from mmearth_train_amd.synth import make_inputs
    lib = _lib.load()
    c = CASES["allmod_atto_56"]
    cfg = case_cfg(c)
    sd, _, _ = case_data(c, cfg)
    import dataclasses
    big = dataclasses.replace(cfg, img_size=64)
    inputs64, _ = make_inputs(big, 3, seed=91)
    assert tuple(inputs64["sentinel2"].shape) == (3, 12, 64, 64) and inputs64["esa_worldcover"].dtype == torch.int64
    # kernel vs indexing, both element sizes
    ty = torch.tensor([0, 8, 3], dtype=torch.int32, device="cuda:0")
    tx = torch.tensor([8, 0, 5], dtype=torch.int32, device="cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for k in ("sentinel2", "esa_worldcover"):
        src = inputs64[k].to("cuda:0").contiguous()
        dst = torch.empty(src.shape[0], src.shape[1], 56, 56, dtype=src.dtype, device="cuda:0")
        assert lib.mpmae_crop(src.data_ptr(), dst.data_ptr(), src.element_size(), 3, src.shape[1], 64, 56, ty.data_ptr(), tx.data_ptr(), st) == 0
        ref = torch.stack([src[n, :, int(ty[n]):int(ty[n]) + 56, int(tx[n]):int(tx[n]) + 56] for n in range(3)])
        assert torch.equal(dst, ref), k
    # through the module: the caller's dict ends up holding the cropped tiles, and the step equals the oracle on them
    model = _module(cfg, sd)
    dev = {k: v.to("cuda:0") for k, v in inputs64.items()}
    torch.manual_seed(17)
    loss, pred, mask, loss_dict, _, _ = model(dev, mask_ratio=0.6)
    assert tuple(dev["sentinel2"].shape) == (3, 12, 56, 56) and tuple(dev["dynamic_world"].shape) == (3, 1, 56, 56)
    assert tuple(dev["era5"].shape) == tuple(inputs64["era5"].shape)
    cropped = OrderedDict((k, v.cpu()) for k, v in dev.items())
    # each cropped tile is a window of the original, the SAME window for every pixel-wise modality of a sample
    s2 = inputs64["sentinel2"]
    for n in range(3):
        hits = [(y, x) for y in range(9) for x in range(9) if torch.equal(s2[n, :, y:y + 56, x:x + 56].nan_to_num(), cropped["sentinel2"][n].nan_to_num())]
        assert len(hits) == 1
        y, x = hits[0]
        assert torch.equal(inputs64["esa_worldcover"][n, :, y:y + 56, x:x + 56], cropped["esa_worldcover"][n])
        assert torch.equal(inputs64["sentinel1"][n, :, y:y + 56, x:x + 56].nan_to_num(), cropped["sentinel1"][n].nan_to_num())
    (oloss, _, omask, _, _, _), _, _ = _oracle(cfg, sd, cropped, model._engine.noise.cpu())
    assert torch.equal(mask.cpu(), omask)
    assert abs(loss.item() - oloss.item()) <= 1e-4 * abs(oloss.item())


def test_raw_tile_preparation_fused_into_the_crop_matches_torch():
    """mpmae_crop_norm / mpmae_crop_lut (the per-sample work of MMEarthDataset.__getitem__, /root/reference/mmearth_dataset.py:100-142,
    fused into the aligned crop): no-data -> NaN, per-band z-score, label remap + -1, against torch indexing on the same raw tiles."""
    from mmearth_train_amd.config import make_cfg
    from mmearth_train_amd.engine import Engine
    from mmearth_train_amd.synth import make_inputs, make_state_dict
    cfg = make_cfg()
    N, H, S = 5, 64, 56
    eng = Engine(cfg, N, dtype="bf16", device="cuda:0")
    eng.load_state_dict(make_state_dict(cfg, seed=2))
    g = torch.Generator().manual_seed(9)
    inputs, noise = make_inputs(cfg, N, seed=3)
    s2 = torch.randint(0, 12000, (N, 12, H, H), generator=g).to(torch.uint16)          # digital numbers, 0 = no data
    s2[:, :, :3, :5] = 0
    s1 = torch.randn(N, 8, H, H, generator=g)
    s1[:, 2:4] = float("-inf")                                                          # missing orbit
    ch = torch.randint(0, 256, (N, 2, H, H), generator=g).to(torch.uint8)               # 255 = no data
    esa = torch.tensor([0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 95, 100, 255], dtype=torch.uint8)[torch.randint(0, 13, (N, 1, H, H), generator=g)]
    lut = torch.full((256,), -1, dtype=torch.int32)
    for new, old in enumerate([10, 20, 30, 40, 50, 60, 70, 80, 90, 95, 100]):
        lut[old] = new
    raw_in = dict(inputs)
    raw_in.update(sentinel2=s2, sentinel1=s1, canopy_height_eth=ch, esa_worldcover=esa)
    for k in ("aster", "dynamic_world"):
        big = torch.zeros(N, inputs[k].shape[1], H, H, dtype=inputs[k].dtype)
        big[:, :, 4:4 + S, 4:4 + S] = inputs[k]
        raw_in[k] = big
    dev = {k: v.cuda() for k, v in raw_in.items()}
    stats = {k: (torch.rand(c, generator=g) * 100 + 1, torch.rand(c, generator=g) * 50 + 1) for k, c in (("sentinel2", 12), ("sentinel1", 8), ("canopy_height_eth", 2))}
    raw = {k: dict(mean=m.cuda(), std=sd_.cuda(), nodata=nd) for (k, (m, sd_)), nd in zip(stats.items(), (0.0, float("-inf"), 255.0))}
    raw["esa_worldcover"] = dict(lut=lut.cuda())
    ty = torch.randint(0, H - S + 1, (N,), generator=g, dtype=torch.int32)
    tx = torch.randint(0, H - S + 1, (N,), generator=g, dtype=torch.int32)
    eng.set_inputs_async(dev, noise.cuda(), crop=(ty.cuda(), tx.cuda()), raw=raw)
    eng.wait_inputs()
    torch.cuda.synchronize()
    for k in ("sentinel2", "sentinel1", "canopy_height_eth"):
        m, sd_ = stats[k]
        src = raw_in[k].float()
        want = torch.stack([src[n, :, ty[n]:ty[n] + S, tx[n]:tx[n] + S] for n in range(N)])
        nod = want == {"sentinel2": 0.0, "sentinel1": float("-inf"), "canopy_height_eth": 255.0}[k]
        want = (want - m[None, :, None, None]) / sd_[None, :, None, None]
        want[nod] = float("nan")
        got = eng.inp[k].cpu()
        assert torch.equal(torch.isnan(got), torch.isnan(want)), k
        assert torch.allclose(torch.nan_to_num(got), torch.nan_to_num(want), rtol=1e-6, atol=1e-6), k
    want = torch.stack([lut[raw_in["esa_worldcover"][n, :, ty[n]:ty[n] + S, tx[n]:tx[n] + S].long()] for n in range(N)]).long()
    assert torch.equal(eng.inp["esa_worldcover"].cpu(), want)
    want = torch.stack([raw_in["aster"][n, :, ty[n]:ty[n] + S, tx[n]:tx[n] + S] for n in range(N)])
    assert torch.equal(eng.inp["aster"].cpu(), want)
    eng.forward()
    torch.cuda.synchronize()
    assert torch.isfinite(eng.total).all()


def test_asynchronous_input_stage_equals_the_in_order_one():
    """Engine.set_inputs_async (input stream behind the running step's last reader of the input buffers, next forward waits for its
    event) over several steps with a DIFFERENT batch each: parameters equal the run that stages every batch on the main stream."""
    from mmearth_train_amd import dist as mdist
    from mmearth_train_amd.synth import make_inputs
    c = CASES["allmod_atto_56"]
    cfg = case_cfg(c)
    sd, inputs, noise = case_data(c, cfg)
    batches = [tuple(make_inputs(cfg, c["N"], seed=40 + i)) for i in range(4)]
    dev = [({k: v.cuda() for k, v in b.items()}, nz.cuda()) for b, nz in batches]
    out = []
    for use_async in (False, True):
        eng = _engine(cfg, c["N"], "f32", sd, inputs, noise)
        run = mdist.StepRunner(eng, world_size=1, lr=1e-3, mode="program")
        assert run.inputs_free_signal
        for b, nz in dev:
            if use_async:
                eng.set_inputs_async(b, nz, runner=run)
            else:
                eng.set_inputs(b, nz)
            run.step()
        torch.cuda.synchronize()
        out.append((eng.pflat.clone(), eng.total.item()))
    r_, dl_ = _rel(out[1][0], out[0][0]), abs(out[1][1] - out[0][1]) / abs(out[0][1])
    assert r_ < 2e-5 and dl_ < 2e-5, (r_, dl_)      # (float atomics reorder the fp32 statistics from run to run: 1e-7 per step)


@pytest.mark.parametrize("mode", ["program", "eager"])
def test_device_meters_match_host_recomputation(mode):
    """the MetricLogger / SmoothedValue(window 20) statistics of /root/reference/helpers.py:49-206 kept on the device
    (written by the optimizer launch: mpmae_hp_fetch; gradient norm accumulated inside mpmae_adamw, helpers.get_grad_norm_ :509-526)
    against a host recomputation from per-step read-backs: value / median / window average / global average of the total loss,
    every per-modality loss, its uncertainty-weighted form and the global gradient norm, over more steps than the window holds."""
    from mmearth_train_amd import dist as mdist
    from mmearth_train_amd.synth import make_inputs
    c = CASES["allmod_atto_56"]
    cfg = case_cfg(c)
    sd, inputs, noise = case_data(c, cfg)
    eng = _engine(cfg, c["N"], "f32", sd, inputs, noise)
    run = mdist.StepRunner(eng, world_size=1, lr=2e-4, mode=mode)
    eng.reset_meters()
    hist = []
    for i in range(27):
        b, nz = make_inputs(cfg, c["N"], seed=300 + i)
        eng.set_inputs(b, nz)
        run.step()
        hist.append((eng.total.item(), eng.losses.cpu().clone(), eng.weighted.cpu().clone(), eng.grad_norm().item()))
    m = eng.read_meters()
    assert m["count"] == 27
    tot = torch.tensor([h[0] for h in hist]); gn = torch.tensor([h[3] for h in hist])
    for name, series, n in (("loss", tot, 27), ("grad_norm", gn[:26], 26),
                            ("loss_sentinel1", torch.stack([h[1] for h in hist])[:, 1], 27),
                            ("weighted_esa_worldcover", torch.stack([h[2] for h in hist])[:, 11], 27)):
        win = series[max(0, n - 20):n]
        got = m[name]
        assert abs(got["value"] - series[n - 1].item()) <= 1e-5 * abs(series[n - 1].item()), name
        assert abs(got["median"] - win.median().item()) <= 1e-5 * abs(win.median().item()), name
        assert abs(got["avg"] - win.mean().item()) <= 1e-5 * abs(win.mean().item()), name
        assert abs(got["global_avg"] - series[:n].mean().item()) <= 1e-4 * abs(series[:n].mean().item()), name
    g = eng.meter_global_averages()           # per-epoch form: includes the last update's norm, one (here trivial) cross-rank fold
    assert abs(g["grad_norm"] - gn.mean().item()) <= 1e-4 * gn.mean().item() and abs(g["loss"] - tot.mean().item()) <= 1e-4 * tot.mean().item()


def test_meter_records_the_rank_mean_loss_and_a_barrier_timeout_skips_the_update():
    """(a) with the data-parallel exchange on, mpmae_hp_fetch receives the all-reduced SUM of the rank losses as its guard
    value; the meter's `loss` column must hold the MEAN over ranks (the reference logs all_reduce_mean(loss), engine_pretrain.py:104):
    emulated here at "world 4" by handing the kernel 4 x loss with grad_scale 1/4. (b) A non-zero grid-barrier error word of a
    persistent stage kernel makes the optimizer launch a no-op (parameters unchanged) and is counted in hp[5] / hp[6]."""
    c = CASES["allmod_atto_56"]
    cfg = case_cfg(c)
    sd, inputs, noise = case_data(c, cfg)
    eng = _engine(cfg, c["N"], "bf16", sd, inputs, noise)
    eng.forward(); eng.backward()
    eng.reset_meters()
    L = eng.total.item()
    guard = (eng.total * 4.0).clone()
    eng.step_count += 1
    eng.set_hyper(1e-5, eng.step_count, grad_scale=0.25)
    eng.launch_adamw(guard_loss=guard)
    torch.cuda.synchronize()
    m = eng.read_meters()
    assert abs(m["loss"]["value"] - L) <= 1e-6 * abs(L), (m["loss"]["value"], L)
    assert eng.hp[5].item() == 0 and eng.hp[6].item() == 0
    assert hasattr(eng, "ps_sync"), "the default bf16 engine runs the persistent stage kernels"
    before = eng.pflat.clone()
    eng.ps_sync[0, 2] = 1                       # a timed-out barrier in the first persistent launch
    eng.step_count += 1
    eng.set_hyper(1e-5, eng.step_count)
    eng.launch_adamw()
    torch.cuda.synchronize()
    assert torch.equal(eng.pflat, before), "the update must be skipped"
    assert eng.hp[5].item() == 1 and eng.hp[6].item() == 1
    # one timeout = ONE skipped update (the word is cleared once counted) ...
    assert int(eng.ps_sync[:, 2].sum()) == 0
    # ... a skipped update leaves NO gradient-norm record (`gn = 0` biased the grad-norm statistics) ...
    gsum = float(eng.meter_sums[-2].item())
    eng.step_count += 1
    eng.set_hyper(1e-5, eng.step_count)
    eng.launch_adamw()
    torch.cuda.synchronize()
    assert float(eng.meter_sums[-2].item()) == gsum, "the fetch after a skipped update must not add a grad-norm record"
    assert not torch.equal(eng.pflat, before) and eng.hp[5].item() == 1 and eng.hp[6].item() == 1
    # ... and the skip is COLLECTIVE: the loss finalisation of the timed-out rank writes +inf as its total, which is what the data-parallel
    # exchange all-reduces into every rank's guard loss
    eng.forward()
    eng.ps_sync[0, 2] = 1                       # (a barrier of THIS forward timed out: the word is set by the kernel, i.e. behind the forward's start)
    eng.finalize_loss(eng._stream(), False, 1.0)
    torch.cuda.synchronize()
    assert math.isinf(eng.total.item()) and eng.total.item() > 0
    # the word is only consumed by an optimizer launch; a forward-only / eval caller must not keep reading +inf for forwards that
    # completed - an eager forward starts with clean error words
    eng.forward()
    torch.cuda.synchronize()
    assert int(eng.ps_sync[:, 2].sum()) == 0
    assert math.isfinite(eng.total.item()) and abs(eng.total.item() - L) <= 0.2 * abs(L)      # (two lr = 1e-5 updates lie between the two)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 visible GPUs (one rank per GPU over RCCL)")
def test_rccl_two_gpus_ranks_agree_and_bench_runs(tmp_path):
    """The first N > 1 run over the real backend (skipped on the 1-GPU test box): two ranks, one per GPU, nccl
    (= RCCL): (a) tools/ddp_probe.py - after two optimizer steps both ranks hold identical parameters and the all-reduced gradients
    equal the hand average of the two ranks' gradients, for the program / segments / eager step drivers; (b) `bench.py --gpus 2`
    starts its two ranks and prints one line with n_gpus 2 and a per-rank time for each rank."""
    import json
    env = dict(os.environ, DDP_PROBE_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ddp_probe.py"), str(tmp_path)], capture_output=True, text=True,
                       env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for mode in ("program", "program+segments", "eager"):
        m = re.search(mode.replace("+", r"\+") + r" ranks equal: (\w+)\s+max rel diff vs hand-averaged reference: ([0-9.e+-]+)", r.stdout)
        assert m and m.group(1) == "True" and float(m.group(2)) < 2e-3, r.stdout
        g = re.search(mode.replace("+", r"\+") + r" step-1 averaged gradient vs reference: max rel ([0-9.e+-]+)", r.stdout)
        assert g and float(g.group(1)) < 1e-4, r.stdout
    b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert b.returncode == 0, b.stdout[-2000:] + b.stderr[-2000:]
    line = json.loads(b.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and len(line["per_rank_ms_per_step"]) == 2


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_backward_is_repeatable_behind_one_forward(dtype):
    """with the one-pass pixel losses the per-modality scalars go into the heads' data-gradient weights; scaled IN PLACE a second
    backward behind the same forward compounded them (wrong dy and every encoder / decoder gradient, silently). The scaled copy is a
    buffer of its own now: two backwards give the same gradients, and the staged weights are untouched."""
    c = CASES["allmod_atto_56"]
    cfg = case_cfg(c)
    sd, inputs, noise = case_data(c, cfg)
    eng = _engine(cfg, c["N"], dtype, sd, inputs, noise)
    eng.forward()
    staged = eng.w["head.pixT"]["t"].clone()
    eng.backward()
    torch.cuda.synchronize()
    g1 = eng.gflat.clone()
    eng.backward()
    torch.cuda.synchronize()
    g2 = eng.gflat.clone()
    assert torch.equal(eng.w["head.pixT"]["t"], staged)
    assert g1.abs().max().item() > 0
    # (the default program is not bit-reproducible run to run - float atomics in the statistics folds - so compare at the noise floor)
    tol = 1e-5 if dtype == "f32" else 2e-2
    assert (g1 - g2).abs().max().item() <= tol * g1.abs().max().item(), (g1 - g2).abs().max().item() / g1.abs().max().item()
    k = "encoder.stages.0.0.pwconv1.linear.weight"
    assert _rel(eng.grads[k], g1[eng.offsets[k][0]:eng.offsets[k][0] + eng.offsets[k][1]].view_as(eng.grads[k])) <= tol
