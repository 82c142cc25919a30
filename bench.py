#!/usr/bin/env python
"""Benchmark of the MP-MAE pretraining micro-step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = FCMAE forward + 12 losses + backward + gradient all-reduce (N > 1) + AdamW on one
synthetic batch (256 tiles of 12x56x56 per GPU, all_mod atto, bf16 activations / fp32 master
weights), inputs resident in HBM. Rank 0 prints ONE JSON line: whole-job images/sec, plus
  "roofline":     achieved vs peak HBM GB/s of the dominant kernel, timed live with HIP events
                  on the launch stream (algorithmic bytes per launch / average launch duration);
  "cpu_baseline": the CPU oracle (oracle/mpmae_ref.py, the parity checker) timed on the host
                  cores on a bounded sample of the same workload. It is a reported baseline only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mmearth_train_amd import MODALITIES as M  # noqa: E402
from mmearth_train_amd.config import make_cfg  # noqa: E402
from mmearth_train_amd.engine import Engine  # noqa: E402
from mmearth_train_amd.synth import make_inputs, make_state_dict  # noqa: E402
from mmearth_train_amd import dist as mdist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_PEAK_TFS = 2500.0         # dense bf16 MFMA peak (MI355X_MICROARCH.md)
STEP_ROOFLINE_US = {            # sum over layers of max(t_MFMA, t_HBM) at bs256
    ("convnextv2_atto", 56, "all_mod"): 764.0,
    ("convnextv2_atto", 56, "pix_mod"): 759.0,
    ("convnextv2_tiny", 112, "all_mod"): 2699.0,
}


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch size")
    ap.add_argument("--model", default="convnextv2_atto")
    ap.add_argument("--img", type=int, default=56)
    ap.add_argument("--patch", type=int, default=8)
    ap.add_argument("--subset", default="all_mod")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying HIP graphs")
    ap.add_argument("--mode", default="program", choices=["program", "hipgraph", "eager"],
                    help="step driver: native launch program (default), HIP graph replay, or the Python loop")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", type=int, default=0, help=argparse.SUPPRESS)      # child process of cpu_baseline_all: N threads, JSON out
    ap.add_argument("--cpu-batch", type=int, nargs="+", default=[4, 32],
                    help="batch sizes of the CPU-oracle baseline (SURVEY 8d: 4 = BASELINE configs[0], and 32)")
    ap.add_argument("--profile-names", action="store_true", help="per-launch (by op name) time table to stderr")
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--block-mode", default=None, choices=[None, "fused", "mat"], help="override the per-block program policy")
    ap.add_argument("--profile-ops", action="store_true", help="print the per-kernel-kind time table to stderr")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the multi-rank path (no GPU, no kernels): the N ranks are spawned exactly as for a real run, "
                         "rendezvous over gloo, build the REAL launch program and gradient-bucket plan on the host, run the real StepRunner "
                         "(segment order, bucketed all-reduce, loss slot, 1/world fold) over stand-in launches and print the bench line in its "
                         "real shape with \"dry_run\": true - what a first 2/4/8-GPU run can fail on besides the kernels themselves")
    return ap


def parse():
    return build_parser().parse_args()


def per_kernel_times(eng, reps=3, by_name=False, split_lanes=False):
    """Eager pass with HIP events (torch.cuda.Event on the launch stream) around every launch."""
    stream = torch.cuda.current_stream()
    acc = {}
    for _ in range(reps):
        eng.stats.zero_()
        eng.gflat.zero_()
        evs = []
        st = eng._stream()
        for phase, ops in (("fwd", eng.fwd_ops), ("bwd", eng.bwd_ops)):
            if phase == "bwd":   # loss finalisation between the two programs (untimed, 1 tiny block)
                eng.finalize_loss(st, True, 1.0)
            for name, fn, args, meta in ops:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                err = fn(*args, st)
                e1.record(stream)
                assert err == 0, (name, err)
                evs.append((dict(meta, name=name), e0, e1))
        torch.cuda.synchronize()
        for meta, e0, e1 in evs:
            key = meta["name"] if by_name else meta["kind"] + (" [side]" if split_lanes and meta.get("lane") else "")
            d = acc.setdefault(key, dict(ms=0.0, n=0, bytes=0, flops=0))
            d["ms"] += e0.elapsed_time(e1)
            d["n"] += 1
            d["bytes"] += meta["bytes"]
            d["flops"] += meta["flops"]
    return acc


# bench op "kind" -> kernel FAMILY, and the HIP kernel symbols (as rocprofv3 prints them) that move a family's data. The second-stage
# folds an entry point launches are part of its op (HIP events around the C-ABI call) and of its family (tools/families.py).
def family_of(kind):
    if kind.startswith("rs<"):
        return "rs"
    if kind.startswith("wgrad"):
        return "wgrad"
    if kind.startswith("dwconv7_wgrad"):
        return "dwconv7_wgrad"
    if kind.startswith("gemm"):
        return "gemm_nt"
    return kind


FAMILY_SYMBOLS = {
    "rs": ("rsc_wide_kernel", "rsc_wide1_kernel", "rsp_wide_kernel", "rsc_narrow_kernel", "rsp_narrow_kernel", "rsn3_bwd_kernel", "rst_kernel"),
    "wgrad": ("gemm_tn2_kernel", "gemm_tn3_kernel", "gemm_tng_kernel", "gemm_tn_bf16_kernel"),
    "dwconv7": ("dwconv7_mfma_kernel", "dwconv7_v6_kernel", "dwconv7_v6s1_kernel"),
    "dwconv7_wgrad": ("dwconv7_wgrad_mfma_kernel", "dwconv7_wgrad_mfma4_kernel", "dwconv7_wgrad_v5_kernel", "dwconv7_wgrad_v6s1_kernel"),
    "gemm_nt": ("gemm_nt_bf16_kernel", "gemm_nt_ring_kernel", "gemm_nt3_kernel"),
    "ps_fwd": ("ps_fwd_kernel",),
}
FAMILY_TEXT = {
    "rs": "fused pointwise row-streaming kernels (LN + pwconv1 + GELU^2 sums, GRN + pwconv2 + residual, pwconv2 data gradient + GRN statistics, "
          "GRN backward + pwconv1 data gradient + LN backward) with their statistic / LayerNorm-gradient folds",
    "wgrad": "weight-gradient GEMMs dW = P^T Q (transpose-read, DMA-ring and grouped kernels) with their slab folds",
    "dwconv7": "depthwise 7x7 forward / data gradient",
    "dwconv7_wgrad": "depthwise 7x7 weight gradient with its folds",
    "gemm_nt": "dense NT GEMMs (decoder block, heads, downsample, stage-3 pointwise)",
    "ps_fwd": "persistent per-sample stage kernel (all blocks of stage 2 forward; stage 3 with ps = 3)",
}


def _args_match(a, bench_args):
    ref, _ = build_parser().parse_known_args(bench_args.split())
    return all(getattr(ref, k) == getattr(a, k) for k in ("model", "img", "patch", "subset", "batch", "dtype"))


def _latest_profile(a, stem):
    """Newest committed profiles/rNN/<stem>*.json whose meta.bench_args name THIS workload (model / size / subset / batch / dtype):
    counters and traces are per workload and only quoted for a run of the same one. Returns (path, document) or (None, None)."""
    import glob
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    for d in sorted((x for x in (os.listdir(root) if os.path.isdir(root) else [])), reverse=True):
        for path in sorted(glob.glob(os.path.join(root, d, stem + "*.json"))):
            try:
                doc = json.load(open(path))
            except Exception:      # noqa: BLE001
                continue
            if _args_match(a, doc.get("meta", {}).get("bench_args", "")):
                return path, doc
    return None, None


def _latest_families(a):
    """In-step family table (tools/families.py) of this workload, or (None, None)."""
    path, doc = _latest_profile(a, "kernel_families")
    if not path:
        return None, None
    return doc, f"{os.path.relpath(path, os.path.dirname(os.path.abspath(__file__)))} (rocprofv3 --kernel-trace of bench.py, commit {doc['meta'].get('commit', 'n/a')})"


def pmc_traffic(family, path):
    """HBM bytes per launch of a family's main kernels from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE, see
    tools/pmc_traffic.py), averaged over the sampled launches of every symbol of the family; None when it was not sampled."""
    syms = FAMILY_SYMBOLS.get(family)
    if not syms or not path or not os.path.exists(path):
        return None
    ks = json.load(open(path))["kernels"]
    tot = n = 0
    for name, d in ks.items():
        if any(sym in name for sym in syms):
            tot += d["traffic_bytes"] * d["launches_sampled"]
            n += d["launches_sampled"]
    return int(tot / n) if n else None


def step_traffic(fam_doc, pmc_path):
    """HBM bytes of ONE step: the committed per-kernel PMC traffic (bytes per launch) x the committed trace's launches per step, summed over
    every kernel symbol of the step; (bytes, symbols priced, symbols without counters) or None."""
    if not fam_doc or not pmc_path or not os.path.exists(pmc_path):
        return None
    ks = json.load(open(pmc_path))["kernels"]
    tot, hit, miss = 0.0, 0, 0
    for fam in fam_doc["families"].values():
        for name, (_, per_step) in fam["kernels"].items():
            d = next((v for k, v in ks.items() if k.startswith(name)), None)
            if d is None:
                miss += 1
                continue
            hit += 1
            tot += d["traffic_bytes"] * per_step
    return (int(tot), hit, miss) if hit else None


def family_fracs(fams, in_step, min_share=0.03):
    """Roofline fraction of EVERY kernel family with >= min_share of the step's kernel time: max(flops / MFMA peak, algorithmic bytes / HBM
    peak) of the family's ops in one step (live launch program's own counts) / its in-step kernel time (committed trace, folds included)."""
    out = {}
    for k, f in fams.items():
        t = in_step.get(k)
        if not t or t["us_per_step"] <= 0 or t.get("share_of_kernel_time", 0) < min_share or not (f["bytes"] or f["flops"]):
            continue
        b, fl = f["bytes"] / 3, f["flops"] / 3                        # (per_kernel_times runs 3 eager passes)
        roof_us = max(fl / MFMA_PEAK_TFS / 1e12, b / HBM_PEAK_GBS / 1e9) * 1e6
        out[k] = dict(us_per_step=t["us_per_step"], share_of_kernel_time=t["share_of_kernel_time"], algorithmic_bytes_per_step=int(b),
                      algorithmic_flops_per_step=int(fl), roofline_us=round(roof_us, 1), frac=round(roof_us / t["us_per_step"], 4))
    return out


def cpu_baseline(cfg, batches, steps, warm=3, threads=None):
    """Oracle (CPU restatement, kind 'port') forward+backward+AdamW on the host cores: `warm` warm-up + `steps` timed fp32 steps
    per batch size (bs 4 and bs 32, 3 + 5). `value` is the best batch size's images/sec."""
    from oracle import mpmae_ref as O
    ncores = threads or min(os.cpu_count(), 16)      # (every host core is tried too, in a time-boxed child process: cpu_baseline_all)
    torch.set_num_threads(ncores)
    sd = make_state_dict(cfg, seed=0)
    runs = []
    for batch in batches:
        inputs, noise = make_inputs(cfg, batch, seed=1000)
        params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        mom = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in sd.items()}

        def one(t):
            for p in params.values():
                p.grad = None
            out = O.forward(params, inputs, noise, cfg)
            out[0].backward()
            with torch.no_grad():
                for k, p in params.items():
                    if p.grad is None:
                        continue
                    new, m, v = O.adamw_step(p, p.grad, mom[k][0], mom[k][1], t, 1e-4)
                    p.copy_(new)
                    mom[k] = (m, v)

        for w in range(warm):
            one(w + 1)
        t0 = time.perf_counter()
        for i in range(steps):
            one(warm + i + 1)
        dt = (time.perf_counter() - t0) / steps
        runs.append(dict(batch=batch, images_per_sec=round(batch / dt, 2), ms_per_step=round(dt * 1e3, 1)))
    best = max(runs, key=lambda r: r["images_per_sec"])
    return dict(value=best["images_per_sec"], unit="images/sec", cores=ncores, kind="port", runs=runs,
                sample=f"{cfg.name} {cfg.img_size}/{cfg.patch_size} {len(cfg.out_mods)}-modality step, batch "
                       f"{' and '.join(str(b) for b in batches)}, {warm} warm-up + {steps} timed fp32 steps each of "
                       f"oracle/mpmae_ref.py (fwd+bwd+AdamW); value = batch {best['batch']}")


def cpu_baseline_all(a, cfg):
    """SURVEY 8d asks for the CPU port on EVERY host core. On the 256-core GPU host that is slower than 16 threads by orders of magnitude
    (thread oversubscription on small ops: round 4's first attempt ran 40 minutes without finishing), so: the 16-thread leg in-process, the
    all-core leg in a child process with a 60 s wall-clock box; both are reported, `value` is the faster one, `cores` its thread count."""
    import subprocess
    base = cpu_baseline(cfg, a.cpu_batch, a.cpu_steps)
    n_all = os.cpu_count()
    if n_all <= base["cores"]:
        return base
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", str(n_all), "--model", a.model, "--img", str(a.img),
           "--patch", str(a.patch), "--subset", a.subset, "--cpu-steps", str(a.cpu_steps), "--cpu-batch"] + [str(b) for b in a.cpu_batch]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=60)
        allc = json.loads(r.stdout.strip().splitlines()[-1])
        base["all_cores"] = dict(cores=n_all, value=allc["value"], runs=allc["runs"])
        if allc["value"] > base["value"]:
            base.update(value=allc["value"], cores=n_all, runs=allc["runs"], sample=allc["sample"])
    except subprocess.TimeoutExpired:
        base["all_cores"] = dict(cores=n_all, value=None, note="did not finish 3 + 5 steps at batch 4 and 32 within the 60 s box")
    except Exception as e:          # noqa: BLE001 - the baseline must never take the bench line down
        base["all_cores"] = dict(cores=n_all, value=None, note=f"failed: {type(e).__name__}")
    return base


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


class _DryEngine:
    """Stand-in for Engine in `--dry-run`: the real engine's static plan (offsets, flat buffers, backward op names, options) with the launch
    methods replaced by host arithmetic - every backward segment adds (rank + 1) * step into the slice of the flat gradient buffer its
    bucket covers, so the all-reduced buffer is checkable (sum over ranks) after every step."""

    def __init__(self, real, rank, buckets):
        self.real, self.rank, self.k, self.buckets = real, rank, 0, buckets
        for name in ("device", "n_params", "offsets", "gflat", "gflat_ext", "pflat", "mflat", "vflat", "total", "bwd_ops", "opt", "hp"):
            setattr(self, name, getattr(real, name))
        self._scale, self._hp = 1.0, (0.0, 0, 1.0)

    def forward(self):
        self.k += 1
        self.total.fill_(float(self.k))

    def finalize_loss(self, st, dlv, scale):
        self._scale = scale

    def _stream(self):
        return None

    def _run(self, ops, st):
        from mmearth_train_amd.dist import split_bwd_segments
        segs = split_bwd_segments(self.bwd_ops)
        names = [o[0] for o in ops]
        for i, sg in enumerate(segs):
            if sg and sg[0][0] == names[0]:
                lo, hi = self.buckets[i]
                self.gflat[lo:hi] += self._scale * (self.rank + 1) * self.k

    def set_hyper(self, lr, t, grad_scale=1.0):
        self._hp = (lr, t, grad_scale)

    def launch_adamw(self, wd, note=True, guard_loss=None):
        self.pflat -= self._hp[0] * self._hp[2] * self.gflat

    def note_optimizer_launch(self):
        pass


def dry_run(a):
    """`bench.py --gpus N --dry-run`: see the flag's help. Exit code 0 and ONE JSON line on rank 0 when spawn, rendezvous, bucket plan,
    exchange and line assembly all work; the gradients of every step are checked against the closed form on every rank."""
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        _spawn_ranks(a, need_gpus=False)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr, flush=True)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        mdist.init(backend="gloo")
    batch = min(a.batch, 2)                       # host memory: the plan does not depend on the batch size
    cfg = make_cfg(a.model, a.img, a.patch, out_modalities=M.subset(a.subset))
    real = Engine(cfg, batch, dtype=a.dtype, device="cpu")
    buckets = mdist.plan_buckets(real.offsets, real.n_params)
    eng = _DryEngine(real, rank, buckets)
    trainer = mdist.StepRunner(eng, world_size=world, lr=1e-4, mode="eager")
    assert (trainer.buckets == buckets and len(trainer.segments) == 4) if world > 1 else True
    for _ in range(a.warmup):
        trainer.step()
    if world > 1:
        mdist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        real.gflat.zero_()
        trainer.step()
        # every rank's segment added (rank + 1) * k: after the exchange the buffer holds k * sum(rank + 1) wherever a bucket covers it
        want = float(eng.k) * sum(r + 1 for r in range(world))
        assert torch.allclose(real.gflat, torch.full_like(real.gflat, want)), (rank, float(real.gflat[0]), want)
    if world > 1:
        mdist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed]
    if world > 1:
        import torch.distributed as tdist
        t = torch.tensor([elapsed], dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        tdist.all_gather(allt, t)
        per_rank = [float(x.item()) for x in allt]
        elapsed = mdist.max_over_ranks(elapsed)
    if rank == 0:
        out = dict(metric="pretrain images/sec (12x56x56 S2, bs256/GPU)" if a.img == 56 else "pretrain images/sec", dry_run=True,
                   value=round(batch * world * a.steps / elapsed, 1), unit="images/sec", n_gpus=world, steps=a.steps, warmup=a.warmup,
                   ms_per_step=round(elapsed / a.steps * 1e3, 4), higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype=a.dtype, data="synthetic (dry run: no kernels, host stand-ins for the launches)",
                   config=dict(workload=f"{a.subset} {a.model.replace('convnextv2_', '')} {a.img}x{a.img} patch{a.patch} DRY RUN over gloo",
                               per_gpu_batch=batch, global_batch=batch * world, parallelism=f"dp{world}", graph="eager",
                               buckets=[hi - lo for lo, hi in buckets], fold_loss=bool(trainer.fold_loss), options=real.nondefault_options()),
                   roofline=None, vendor_kernels_per_step=0.0, per_rank_ms_per_step=[round(t / a.steps * 1e3, 4) for t in per_rank],
                   exposed_comm_tail_ms=0.0)
        print(json.dumps(out), flush=True)
    if world > 1:
        mdist.barrier()
        mdist.shutdown()


def _spawn_ranks(a, need_gpus=True):
    """`python bench.py --gpus N` outside a launcher: start the N ranks ourselves (one process per GPU under torch.distributed.run,
    rendezvous on 127.0.0.1, like the reference's `torch.distributed.launch --nproc_per_node=N main_pretrain.py`,
    /root/reference/TRAINING.md:18-42) and pass their exit code on. Fails loudly when the node has fewer GPUs."""
    import subprocess
    have = torch.cuda.device_count()
    if need_gpus and have < a.gpus:
        print(f"bench.py: --gpus {a.gpus} requested but this node exposes {have} GPU(s)", file=sys.stderr, flush=True)
        sys.exit(2)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd, env=env).returncode)


def main():
    a = parse()
    # bench integrity: the timing-experiment variables of earlier rounds silenced / re-ordered ops inside the product engine;
    # they are gone from it (tools/timing_experiment.py patches its own process and stamps the line), and a box that still exports them is refused
    stale = [k for k in ("MPMAE_SKIP_OPS", "MPMAE_DEFER_EXPERIMENT") if os.environ.get(k)]
    if stale:
        print(f"bench.py: refusing to run with {stale} set (timing experiments produce INVALID results: use tools/timing_experiment.py)",
              file=sys.stderr, flush=True)
        sys.exit(3)
    if a.dry_run:
        return dry_run(a)
    if a.cpu_baseline_only:
        cfg = make_cfg(a.model, a.img, a.patch, out_modalities=M.subset(a.subset))
        print(json.dumps(cpu_baseline(cfg, a.cpu_batch, a.cpu_steps, threads=a.cpu_baseline_only)), flush=True)
        return
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        _spawn_ranks(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr, flush=True)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        mdist.init(backend="nccl", local_rank=local_rank)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    cfg = make_cfg(a.model, a.img, a.patch, out_modalities=M.subset(a.subset))
    eng = Engine(cfg, a.batch, dtype=a.dtype, device=dev, block_mode=a.block_mode)
    eng.load_state_dict(make_state_dict(cfg, seed=0))
    options = eng.nondefault_options()
    import mmearth_train_amd.engine as _E
    if getattr(_E, "TIMING_EXPERIMENT", None):
        options["TIMING_EXPERIMENT_INVALID"] = _E.TIMING_EXPERIMENT
    if os.environ.get("MPMAE_LIB"):
        options["MPMAE_LIB"] = os.environ["MPMAE_LIB"]
    inputs, noise = make_inputs(cfg, a.batch, seed=1000 + rank)
    eng.set_inputs(inputs, noise)
    torch.cuda.synchronize()
    trainer = mdist.StepRunner(eng, world_size=world, lr=1e-4, mode="eager" if a.no_graph else a.mode)

    for _ in range(a.warmup):
        trainer.step()
    torch.cuda.synchronize()
    if world > 1:
        mdist.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]      # HIP events on the launch stream
    if world > 1:
        trainer.measure_comm_tail = True        # two more events per step on the main stream (exposed_comm_tail_ms in the line)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(a.steps):
        trainer.step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        mdist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    comm_tail_pre = trainer.comm_tail_ms() if world > 1 else None
    trainer.measure_comm_tail = False
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
    median_ms = step_ms[len(step_ms) // 2]
    # the same step with the input stage inside the timed region: a fresh resident batch copied into the engine's
    # static buffers and fresh device mask noise every step, as engine_pretrain.train_one_epoch does
    inputs_dev = {k: v.to(dev) for k, v in inputs.items()}
    torch.cuda.synchronize()
    n_in = max(5, a.steps // 5)
    for _ in range(3):                      # (warm-up: the input stream, its first events)
        eng.set_inputs_async(inputs_dev, None, runner=trainer)
        trainer.step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(n_in):
        eng.set_inputs_async(inputs_dev, None, runner=trainer)
        trainer.step()
    torch.cuda.synchronize()
    ms_with_inputs = (time.perf_counter() - t1) / n_in * 1e3
    comm_tail = comm_tail_pre
    per_rank = [elapsed]
    if world > 1:          # every rank's own wall time of the timed region (the line's value uses the slowest)
        import torch.distributed as tdist
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        allt = [torch.zeros_like(t) for _ in range(world)]
        tdist.all_gather(allt, t)
        per_rank = [float(x.item()) for x in allt]
    elapsed = mdist.max_over_ranks(elapsed) if world > 1 else elapsed
    loss = float(eng.total.item())
    ms_per_step = elapsed / a.steps * 1e3
    value = a.batch * world * a.steps / elapsed

    out = None
    if rank == 0:
        # --- roofline of the dominant kernel FAMILY ---
        # live: HIP events (on the launch stream) around every C-ABI call of an eager pass, an op's second-stage folds included;
        # in-step: the committed rocprofv3 kernel trace of this workload (both lanes live), which also CHOOSES the family when present
        acc = per_kernel_times(eng)
        tot = sum(d["ms"] for d in acc.values())
        fams = {}
        for kind, d in acc.items():
            f = fams.setdefault(family_of(kind), dict(ms=0.0, n=0, bytes=0, flops=0))
            for k in f:
                f[k] += d[k]
        fam_doc, fam_src = _latest_families(a)
        in_step = fam_doc["families"] if fam_doc else {}
        cands = [k for k in fams if k in in_step and in_step[k]["us_per_step"] > 0]
        dom_kind = max(cands, key=lambda k: in_step[k]["us_per_step"]) if cands else max(fams, key=lambda k: fams[k]["ms"])
        dom = fams[dom_kind]
        avg_ms = dom["ms"] / dom["n"]
        avg_bytes = dom["bytes"] / dom["n"]
        avg_flops = dom["flops"] / dom["n"]
        # `algorithmic_bytes` follows SURVEY 8(d): layer-granular - every pointwise layer of a block reads its input and writes its output once,
        # x3 for forward + backward: 60 M C bytes per block in bf16 (20 M C forward, 40 M C backward). The launch program's own count
        # (every tensor an op must read or write, saved x-hat / xn / statistics operands included) stays as `must_move_bytes`.
        must_move = avg_bytes
        if dom_kind == "rs":
            tot_b = 0
            for blk in eng.blocks:
                if blk.get("rs"):
                    fwd_rs = not eng._ps_ok(blk["stage"])            # the persistent stage kernel is its own family
                    tot_b += (20 if fwd_rs else 0) * blk["M"] * blk["C"] + 40 * blk["M"] * blk["C"]
            if tot_b:
                avg_bytes = tot_b / max(dom["n"] // 3, 1)
        mfma_bound = avg_flops / MFMA_PEAK_TFS / 1e12 > avg_bytes / HBM_PEAK_GBS / 1e9
        achieved = (avg_flops / (avg_ms * 1e-3) / 1e12 if mfma_bound else avg_bytes / (avg_ms * 1e-3) / 1e9) if avg_ms > 0 else 0.0
        if a.profile_ops:
            acc2 = per_kernel_times(eng, split_lanes=True)
            tot2 = sum(d["ms"] for d in acc2.values())
            side = sum(d["ms"] for k, d in acc2.items() if k.endswith("[side]"))
            print(f"eager sum {tot2 / 3:.3f} ms/step, side lane {side / 3:.3f} ms/step", file=sys.stderr)
            for k, d in sorted(acc2.items(), key=lambda kv: -kv[1]["ms"]):
                gbs = d["bytes"] / max(d["ms"], 1e-9) / 1e6
                tfs = d["flops"] / max(d["ms"], 1e-9) / 1e9
                print(f"{k:36s} {d['ms'] / 3:9.3f} ms/step {d['n'] // 3:4d} launches "
                      f"{gbs:8.0f} GB/s {tfs:8.1f} TFLOP/s  {100 * d['ms'] / tot:5.1f}%", file=sys.stderr)
        if a.profile_names:
            byn = per_kernel_times(eng, by_name=True)
            for k, d in sorted(byn.items(), key=lambda kv: -kv[1]["ms"])[:400]:
                gbs = d["bytes"] / max(d["ms"], 1e-9) / 1e6
                tfs = d["flops"] / max(d["ms"], 1e-9) / 1e9
                print(f"{k:58s} {d['ms'] / 3 * 1e3:9.1f} us {gbs:8.0f} GB/s {tfs:8.1f} TF/s", file=sys.stderr)
        key = (a.model, a.img, a.subset)
        step_roof = STEP_ROOFLINE_US.get(key)
        pmc_path, pmc_doc = _latest_profile(a, "pmc_traffic")      # counters are per workload
        traffic = pmc_traffic(dom_kind, pmc_path) if pmc_path else None
        peak = MFMA_PEAK_TFS if mfma_bound else HBM_PEAK_GBS
        ops_per_step = dom["n"] // 3
        roof = dict(bound="mfma" if mfma_bound else "hbm", kernel=dom_kind, kernel_family=FAMILY_TEXT.get(dom_kind, dom_kind),
                    kernel_symbols=list(FAMILY_SYMBOLS.get(dom_kind, ())),
                    achieved=round(achieved, 1), peak=peak, unit="TFLOP/s" if mfma_bound else "GB/s",
                    frac=round(achieved / peak, 4), traffic=traffic,
                    algorithmic_bytes=int(avg_bytes), must_move_bytes=int(must_move), algorithmic_flops=int(avg_flops),
                    kernel_avg_us=round(avg_ms * 1e3, 2), kernel_launches_per_step=ops_per_step,
                    kernel_share_of_step=round(dom["ms"] / tot, 3),
                    timing="live: HIP events around every C-ABI call of the family in an eager pass (an op = its kernel + the folds it launches)")
        if dom_kind in in_step:
            # the same family inside the real step (both lanes live): what the step pays; `frac` is the smaller of the two
            us_step = in_step[dom_kind]["us_per_step"]
            per_op = us_step / max(ops_per_step, 1)
            ach_step = (avg_flops / (per_op * 1e-6) / 1e12 if mfma_bound else avg_bytes / (per_op * 1e-6) / 1e9) if per_op > 0 else 0.0
            roof.update(in_step=dict(us_per_step=us_step, avg_us_per_op=round(per_op, 2), achieved=round(ach_step, 1),
                                     frac=round(ach_step / peak, 4), share_of_kernel_time=in_step[dom_kind]["share_of_kernel_time"],
                                     source=fam_src))
            roof["frac"] = round(min(achieved, ach_step) / peak, 4)
            if "mfma_util" in in_step[dom_kind]:      # stand-alone MFMA utilisation of the family (tools/mfma_util.py: a --pmc pass of this workload)
                roof["mfma_util"] = in_step[dom_kind]["mfma_util"]
                roof["mfma_flop_frac"] = in_step[dom_kind]["mfma_flop_frac"]
        if traffic is not None:
            roof["traffic_source"] = (f"{os.path.relpath(pmc_path, os.path.dirname(os.path.abspath(__file__)))} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                      f"passes; commit {pmc_doc.get('meta', {}).get('commit', 'n/a')})")
        if fam_doc:
            # the WORST family beside the largest one (the depthwise weight gradient sat at 0.07 while the line priced `rs`)
            ff = family_fracs(fams, in_step)
            if ff:
                worst = min(ff, key=lambda k: ff[k]["frac"])
                roof["families"] = {k: v["frac"] for k, v in sorted(ff.items(), key=lambda kv: kv[1]["frac"])}
                roof["worst_family"] = dict(kernel=worst, kernel_symbols=list(FAMILY_SYMBOLS.get(worst, ())), **ff[worst])
            roof["launches_per_step"] = fam_doc["meta"].get("launches_per_step")
            st = step_traffic(fam_doc, pmc_path)
            if st:
                roof["step_traffic_bytes"] = st[0]
                roof["step_traffic_note"] = (f"sum over {st[1]} kernel symbols of PMC bytes per launch x launches per step (committed trace + counters)"
                                             + (f"; {st[2]} symbols without counters" if st[2] else ""))
        roof["ops_per_step"] = len(eng.fwd_ops) + len(eng.bwd_ops) + 3      # live: C-ABI ops of the recorded step (+ finalisation, hp_fetch, AdamW)
        if step_roof and a.batch == 256:
            roof["step_roofline_us"] = step_roof
            roof["step_frac"] = round(step_roof / (ms_per_step * 1e3), 4)
        pieces = None
        if world == 1 and getattr(trainer, "prog", None) is not None and hasattr(trainer, "_span"):
            # forward / backward / optimizer split of the recorded step (HIP events around replays of program ranges, unprofiled)
            nseg = len(trainer.segments)

            def timed(span, reps=20):
                for _ in range(3):
                    eng.run_program(trainer.prog, span)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    eng.run_program(trainer.prog, span)
                e1.record()
                torch.cuda.synchronize()
                return round(e0.elapsed_time(e1) / reps * 1e3, 1)
            pieces = dict(forward_us=timed(trainer._span(0, 0)), backward_us=timed(trainer._span(1, 1 + nseg)),
                          forward_backward_us=timed(trainer._span(0, 1 + nseg)))
        out = dict(metric="pretrain images/sec (12x56x56 S2, bs256/GPU)" if a.img == 56 else "pretrain images/sec",
                   value=round(value, 1), unit="images/sec", n_gpus=world, steps=a.steps, warmup=a.warmup,
                   ms_per_step=round(ms_per_step, 4), ms_per_step_median_hip_events=round(median_ms, 4),
                   ms_per_step_with_input_stage=round(ms_with_inputs, 4), higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype={"bf16": "bf16", "fp8": "fp8 (OCP e4m3 + E8M0 MX block scales on the decoder block's pointwise GEMMs and the K = 1280 pointwise GEMMs of encoder stage 3, bf16 elsewhere, fp32 accumulation)"}.get(a.dtype, "f32"), data="synthetic",
                   config=dict(workload=f"{a.subset} {a.model.replace('convnextv2_', '')} {a.img}x{a.img} patch{a.patch} "
                                        f"mask0.6 uncertainty loss, fwd+loss+bwd+allreduce+AdamW",
                               per_gpu_batch=a.batch, global_batch=a.batch * world,
                               parallelism=f"dp{world}", graph=trainer.graph_mode, final_loss=round(loss, 4),
                               # every engine / library switch that is NOT at its default ({} and {} on a clean run), and the stamp of a
                               # timing experiment (tools/timing_experiment.py): a faster, wrong headline cannot appear without a trace
                               options=options,
                               input_stage="outside the timed region of `value` (inputs and mask noise resident in HBM); "
                                           "ms_per_step_with_input_stage includes the D2D batch copy and device randn of every step, issued on an input stream "
                                           "behind the previous step's last reader of the input buffers (Engine.set_inputs_async)"),
                   roofline=roof,
                   # kernels of a vendor library inside the timed step: none can be - since round 6 the library links no vendor BLAS at all
                   # (libmpmae_hip.so: `nm -u` shows the HIP runtime and libstdc++ only; the optional hipBLASLt route of rounds 4-5 is removed)
                   vendor_kernels_per_step=0.0)
        if pieces:
            out["piece_times"] = pieces
        if world > 1:
            out["per_rank_ms_per_step"] = [round(t / a.steps * 1e3, 4) for t in per_rank]
            if comm_tail is not None:      # main stream idle between the last backward kernel and the last bucket's all-reduce (rank 0)
                out["exposed_comm_tail_ms"] = round(comm_tail, 4)
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_all(a, cfg)
        print(json.dumps(out), flush=True)
    if world > 1:
        mdist.barrier()
        mdist.shutdown()


if __name__ == "__main__":
    main()
