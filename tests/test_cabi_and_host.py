"""CPU tests: the C-ABI library loads and exports every symbol include/mpmae_hip.h declares
(no compute calls without a GPU), ctypes struct layouts match the header, host-side logic."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from mmearth_train_amd import _lib
    return _lib.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "mpmae_hip.h")).read()
    declared = set(re.findall(r"\bint\s+(mpmae_\w+)\s*\(", hdr))
    from mmearth_train_amd import _lib
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    others = set(re.findall(r"\b(?:void|MpmaeProgram\*|long long)\s+(mpmae_\w+)\s*\(", hdr))
    assert others == set(_lib.OTHER_SYMBOLS), others ^ set(_lib.OTHER_SYMBOLS)
    for name in declared | others:
        assert hasattr(lib, name), name
    assert lib.mpmae_arch() == 950


def test_struct_layouts_match_header(tmp_path):
    """sizeof of every args struct as seen by a C compiler == ctypes.sizeof of the binding."""
    from mmearth_train_amd import _lib
    names = dict(MpmaeGeom=_lib.Geom, MpmaeGemmArgs=_lib.GemmArgs, MpmaeWgradArgs=_lib.WgradArgs,
                 MpmaeDwArgs=_lib.DwArgs, MpmaeDwWgArgs=_lib.DwWgArgs, MpmaePrepDesc=_lib.PrepDesc,
                 MpmaePixContArgs=_lib.PixContArgs, MpmaePixCatArgs=_lib.PixCatArgs, MpmaeImgArgs=_lib.ImgArgs,
                 MpmaeRsArgs=_lib.RsArgs, MpmaeStemTailArgs=_lib.StemTailArgs,
                 MpmaePsBlock=_lib.PsBlock, MpmaePsArgs=_lib.PsArgs,
                 MpmaeMeters=_lib.Meters,
                 MpmaeStemFrontArgs=_lib.StemFrontArgs)
    src = tmp_path / "sz.c"
    body = "\n".join(f'  printf("{n} %zu\\n", sizeof({n}));' for n in names)
    src.write_text(f'#include <stdio.h>\n#include "mpmae_hip.h"\nint main(void) {{\n{body}\n  return 0;\n}}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    for line in out.strip().splitlines():
        n, sz = line.split()
        assert ctypes.sizeof(names[n]) == int(sz), (n, sz, ctypes.sizeof(names[n]))


def test_engine_program_builds_on_cpu(lib):
    """The launch program (argument records, buffer plan, prep table) builds without a GPU."""
    from mmearth_train_amd.config import make_cfg
    from mmearth_train_amd.engine import Engine
    from mmearth_train_amd.synth import make_state_dict
    cfg = make_cfg()
    eng = Engine(cfg, 2, dtype="bf16", device="cpu")
    assert eng.n_params == 7580674                    # SURVEY §8a row 13 (all_mod atto)
    assert len(eng.fwd_ops) > 30 and len(eng.bwd_ops) > 50      # (stages 2 / 3 of the forward are ONE persistent launch each)
    sd = make_state_dict(cfg, seed=1)
    eng.load_state_dict(sd)
    full = eng.state_dict()
    # reference state-dict layout: the shared decoder block appears under every output modality
    for om in cfg.out_mods:
        assert f"decoder_dict.{om.name}.0.pwconv1.weight" in full
    assert full["decoder_dict.biome.0.grn.gamma"].data_ptr() == full["decoder_dict.sentinel2.0.grn.gamma"].data_ptr()
    # timm weight-decay grouping: biases / 1-D tensors are not decayed, GRN gamma/beta and mask_token are
    off, n = eng.offsets["encoder.stages.0.0.grn.gamma"]
    assert eng.decay_mask[off:off + n].all()
    off, n = eng.offsets["encoder.stages.0.0.dwconv.bias"]
    assert not eng.decay_mask[off:off + n].any()
    off, n = eng.offsets["mask_token"]
    assert eng.decay_mask[off:off + n].all()
    off, n = eng.offsets["loss_fn.log_vars"]
    assert not eng.decay_mask[off:off + n].any()
    tiny = Engine(make_cfg("convnextv2_tiny", 112, 16), 2, dtype="f32", device="cpu")
    assert tiny.n_params == 36625946


def test_bucket_plan_and_segments(lib):
    from mmearth_train_amd.config import make_cfg
    from mmearth_train_amd.engine import Engine
    from mmearth_train_amd.dist import plan_buckets, split_bwd_segments
    from mmearth_train_amd import MODALITIES as MM
    for subset in ("all_mod", "pix_mod", "S2"):
        eng = Engine(make_cfg(out_modalities=MM.subset(subset)), 2, dtype="bf16", device="cpu")
        b = plan_buckets(eng.offsets, eng.n_params)
        assert len(b) == 4 and b[0][1] == eng.n_params and b[3][0] == 0
        assert all(b[i][0] == b[i + 1][1] for i in range(3))                      # contiguous, in reverse-forward order
        segs = split_bwd_segments(eng.bwd_ops)
        assert sum(len(s) for s in segs) == len(eng.bwd_ops)
        # every gradient written by a segment lies in a bucket that is reduced at or after that segment
        names0 = [o[0] for o in segs[0]]
        assert any(n.startswith("head:") for n in names0) and not any(n.startswith(("decoder_dict.", "encoder.", "proj")) for n in names0)
        names1 = [o[0] for o in segs[1]]
        assert any(n.startswith("proj.wgrad") for n in names1) and not any(n.startswith(("encoder.", "head")) for n in names1)
        names2 = [o[0] for o in segs[2]]
        assert all(n.startswith(("encoder.stages.3.", "encoder.stages.2.", "encoder.downsample_layers.2")) for n in names2)
        # the head bucket holds exactly the heads, their shared LayerNorm and the uncertainty weights
        lo = b[0][0]
        assert all(k.startswith(("pred_dict.", "layer_norm_tmp.", "loss_fn.")) for k, (o, n) in eng.offsets.items() if o >= lo)
    assert b[0][1] - b[0][0] > 0
    # the dense encoder (sparse=False, patch 16): its final norm / classifier head sit behind stages.3 in the state dict and receive no
    # gradient; the bucket plan and the segment cut must take them (plan_buckets asserted on encoder.norm.weight)
    eng = Engine(make_cfg("convnextv2_atto", 112, 16, sparse=False), 2, dtype="bf16", device="cpu")
    b = plan_buckets(eng.offsets, eng.n_params)
    assert len(b) == 4 and b[0][1] == eng.n_params and b[3][0] == 0 and all(b[i][0] == b[i + 1][1] for i in range(3))
    o_norm = eng.offsets["encoder.norm.weight"][0]
    assert b[2][0] <= o_norm < b[2][1]
    assert sum(len(s) for s in split_bwd_segments(eng.bwd_ops)) == len(eng.bwd_ops)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from mmearth_train_amd import dist as mdist
    mdist.init(backend="gloo")
    torch.manual_seed(rank)
    g = torch.randn(1000)
    ref = g.clone()
    mdist.allreduce_buckets_sync(g, [(800, 1000), (600, 800), (200, 600), (0, 200)])
    allg = [torch.zeros(1000) for _ in range(world)]
    dist.all_gather(allg, ref)
    ok = torch.allclose(g, sum(allg), atol=1e-6)
    mx = mdist.max_over_ranks(float(rank + 1))
    mdist.barrier()
    q.put((rank, bool(ok), mx))
    mdist.shutdown()


def test_gloo_world2_bucketed_allreduce():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29611
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res)
    assert all(mx == 2.0 for _, _, mx in res)


def test_fcmae_module_state_dict_layout_cpu(lib):
    """Module tree reproduces the reference's state-dict keys/shapes without a GPU."""
    from mmearth_train_amd import fcmae
    from mmearth_train_amd.config import default_args
    from mmearth_train_amd.custom_loss import UncertaintyWeightingStrategy
    m = fcmae.convnextv2_atto(mask_ratio=0.6, decoder_depth=1, decoder_embed_dim=512, norm_pix_loss=True,
                              patch_size=8, img_size=56, args=default_args(), loss_fn=UncertaintyWeightingStrategy(12),
                              sparse=True, device="cpu")
    sd = m.state_dict()
    assert len(sd) == 290
    assert tuple(sd["encoder.initial_conv.0.kernel"].shape) == (9, 12, 40)
    assert tuple(sd["encoder.stages.2.5.pwconv1.linear.weight"].shape) == (640, 160)
    assert tuple(sd["encoder.downsample_layers.1.1.kernel"].shape) == (4, 80, 160)
    assert tuple(sd["decoder_dict.esa_worldcover.0.grn.gamma"].shape) == (1, 1, 1, 2048)
    assert tuple(sd["pred_dict.sentinel2.weight"].shape) == (768, 512, 1, 1)
    assert tuple(sd["loss_fn.log_vars"].shape) == (12,)
    assert sd["decoder_dict.lat.0.pwconv2.weight"].data_ptr() == sd["decoder_dict.sentinel2.0.pwconv2.weight"].data_ptr()
    assert sum(p.numel() for p in m.parameters()) == 7580674
    assert hasattr(m, "encoder") and hasattr(m, "proj") and hasattr(m, "pred_dict") and hasattr(m, "layer_norm_tmp")
    golden = __import__("tests.golden_cases", fromlist=["load_fixture"]).load_fixture("misc")
    # checkpoint remap contract (helpers.remap_checkpoint_keys): our encoder keys are the ones it expects
    enc_keys = [k for k in sd if k.startswith("encoder.")]
    assert len(enc_keys) == len(golden["remap_keys"])
    import main_pretrain
    flags = {a.dest for a in main_pretrain.get_args_parser()._actions}
    for f in ["model", "input_size", "patch_size", "mask_ratio", "norm_pix_loss", "decoder_depth", "decoder_embed_dim",
              "use_orig_stem", "loss_aggr", "batch_size", "update_freq", "epochs", "warmup_epochs", "blr", "lr", "min_lr",
              "weight_decay", "use_mixed", "sparse", "distributed", "no_ffcv", "output_dir", "auto_resume", "save_ckpt",
              "save_ckpt_freq", "save_ckpt_num", "seed", "device"]:
        assert f in flags, f


@pytest.mark.parametrize("launcher,n", [("self", 2), ("driver", 4)])
def test_bench_multi_rank_dry_run_over_gloo(launcher, n):
    """everything of a first N > 1 run that is not a kernel, rehearsed on the CPU - `bench.py --gpus N --dry-run` spawns
    (or is spawned as, the way the driver does it) N ranks, rendezvous on 127.0.0.1 over gloo, builds the REAL launch program and bucket
    plan, runs the real StepRunner over stand-in launches (every step's all-reduced gradient buffer is checked against its closed form on
    every rank, the loss slot folded into the first bucket) and prints ONE line in the real bench shape."""
    import json
    import socket
    import subprocess
    bench = os.path.join(ROOT, "bench.py")
    args = ["--gpus", str(n), "--dry-run", "--steps", "2", "--warmup", "1"]
    if launcher == "self":
        cmd = [sys.executable, bench] + args
    else:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), bench] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == n and d["config"]["parallelism"] == f"dp{n}" and d["scaling"] == "weak"
    assert len(d["per_rank_ms_per_step"]) == n and "exposed_comm_tail_ms" in d and d["steps"] == 2
    assert sum(d["config"]["buckets"]) == 7580674 and d["config"]["fold_loss"] is True
    for k in ("metric", "value", "unit", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "roofline"):
        assert k in d, k
    assert d["config"]["options"] == {"engine": {}, "library": {}}      # a clean run: every engine / library switch at its default


def test_bench_refuses_timing_experiment_variables_and_stamps_options():
    """the timing hooks are out of Engine (tools/timing_experiment.py patches its own process); bench.py exits non-zero
    when a box still exports their old variables, and prints every non-default switch in config.options."""
    import json
    import subprocess
    bench = os.path.join(ROOT, "bench.py")
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "MPMAE_ENGINE_OPTS")}
    for var in ("MPMAE_SKIP_OPS", "MPMAE_DEFER_EXPERIMENT"):
        r = subprocess.run([sys.executable, bench, "--dry-run", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600,
                           cwd=ROOT, env=dict(base, **{var: ".wgrad"}))
        assert r.returncode == 3 and "refusing" in r.stderr, (r.returncode, r.stderr[-500:])
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=600,
                       cwd=ROOT, env=dict(base, MPMAE_ENGINE_OPTS="tail_main=1,TN3_BLOCKS=64"))
    assert r.returncode == 0, r.stderr[-1500:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["config"]["options"] == {"engine": {"tail_main": 1}, "library": {"TN3_BLOCKS": 64}}
    import glob
    for f in glob.glob(os.path.join(ROOT, "mmearth-train_amd", "engine*.py")):
        src = open(f).read()
        assert "MPMAE_SKIP_OPS" not in src and "MPMAE_DEFER_EXPERIMENT" not in src, f


def test_xcd_tile_order_is_a_permutation_that_keeps_a_row_tile_on_one_xcd():
    """csrc/common.cuh xcd_tile(): workgroup L (x fastest; the hardware deals it to XCD L % 8) -> (row tile, column tile). The mapping must
    visit every tile exactly once for any grid, and the column tiles of a row tile below the last gx % 8 row tiles must share one XCD and
    be consecutive there (they read the same rows of A; round 5: 254 -> 85 MB of fabric fetch per launch at the decoder / head shapes)."""
    def xcd_tile(L, gx, gy):
        gxm = gx & ~7
        if L < gxm * gy:
            c, j = L & 7, L >> 3
            return (j // gy) * 8 + c, j % gy
        t = L - gxm * gy
        return gxm + t // gy, t % gy

    for gx, gy in ((98, 4), (98, 16), (98, 22), (28, 5), (28, 20), (7, 3), (8, 1), (1, 1), (33, 2), (2432, 2)):
        tiles = [xcd_tile(L, gx, gy) for L in range(gx * gy)]
        assert sorted(tiles) == [(m, n) for m in range(gx) for n in range(gy)], (gx, gy)
        by_row = {}
        for L, (m, n) in enumerate(tiles):
            by_row.setdefault(m, []).append((L % 8, L // 8, n))
        for m, lst in by_row.items():
            if m < (gx & ~7):
                assert len({c for c, _, _ in lst}) == 1, (gx, gy, m)                      # one XCD
                js = sorted(j for _, j, _ in lst)
                assert js == list(range(js[0], js[0] + gy)), (gx, gy, m)                  # back to back on it
