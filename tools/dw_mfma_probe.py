"""Developer probe of the matrix-core depthwise kernels (csrc/dwmfma.cuh, MPMAE_OPT_DW = 8) against the VALU kernels (DW = 7) and
torch conv2d: correctness at N = 6 (with activity bits), timing of every stage-0 / stage-1 forward and data-gradient record at
N = 256. CFG=tiny runs the tiny 112/16 geometry (S = 8 at C = 96, S = 4 at C = 192)."""
import ctypes as C, os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from mmearth_train_amd import _lib
from mmearth_train_amd.config import make_cfg
from mmearth_train_amd.engine import Engine
from mmearth_train_amd.synth import make_inputs, make_state_dict
import test_hip_kernels_bf16 as T

lib = _lib.load()
bf, DEV = torch.bfloat16, "cuda:0"
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
tiny = os.environ.get("CFG", "") == "tiny"


def make(N):
    cfg = make_cfg(model="convnextv2_tiny", img_size=112, patch_size=16) if tiny else make_cfg()
    e = Engine(cfg, N, dtype="bf16", device=DEV, options=dict(ps=0, dw_group=9, wgrad_group=0))
    e.load_state_dict(make_state_dict(cfg, seed=21))
    inputs, noise = make_inputs(cfg, N, seed=22)
    if N <= 8:
        z = torch.rand(N, 1, cfg.img_size, cfg.img_size, generator=torch.Generator().manual_seed(23)) < 0.06
        inputs["sentinel2"] = inputs["sentinel2"] * (~z)
    e.set_inputs(inputs, noise)
    names = [o[0] for o in e.fwd_ops]
    stem = next(i for i, n_ in enumerate(names) if n_.startswith("stem:"))
    e._run(e.fwd_ops[:stem], e._stream())
    torch.cuda.synchronize()
    return e


def t_us(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def run(e, check):
    ops = {o[0]: o for o in e.fwd_ops + e.bwd_ops}
    for blk in T._blocks(e):
        if not blk["sparse"] or e.S[blk["stage"]] < 4:
            continue
        tag, M, Cc, S = blk["prefix"], blk["M"], blk["C"], e.S[blk["stage"]]
        act = e.act[blk["stage"]]
        live = act.bool()[:, None]
        Wd, bias = T._dense_dw_weight(e, blk)
        torch.manual_seed(M + Cc)
        for which in (":dw", ":dw.dgrad"):
            name, fn, args, _ = ops[tag + which]
            a = type(args[1]._obj).from_buffer_copy(args[1]._obj)
            x = (torch.randn(M, Cc, device=DEV) * 1.3).to(bf) * live
            add = (torch.randn(M, Cc, device=DEV)).to(bf) * live if a.add else None
            outs = {}
            for dw in (7, 8):
                assert lib.mpmae_set_option(_lib.OPT["DW"], dw) == 0
                out = torch.full((M, Cc), 7.0, device=DEV, dtype=bf)
                a.x, a.out, a.add = x.data_ptr(), out.data_ptr(), (add.data_ptr() if add is not None else 0)
                r = lib.mpmae_dwconv7_fwd(1, C.byref(a), st())
                torch.cuda.synchronize()
                assert r == 0, (name, dw, r)
                outs[dw] = out
                if check:          # bitwise repeatability (no atomics in either kernel)
                    for rep in range(5):
                        out2 = torch.full((M, Cc), 3.0, device=DEV, dtype=bf)
                        a.out = out2.data_ptr()
                        assert lib.mpmae_dwconv7_fwd(1, C.byref(a), st()) == 0
                        torch.cuda.synchronize()
                        assert torch.equal(out2, out), (name, dw, rep, int((out2 != out).sum()))
                    a.out = out.data_ptr()
                us = t_us(lambda: lib.mpmae_dwconv7_fwd(1, C.byref(a), st()))
                print(f"  {name:38s} S={S} C={Cc:4d} M={M:7d} DW={dw}: {us:7.1f} us   ({2 * M * Cc * 2 / us / 1e6:6.2f} TB/s of 1r + 1w)")
            if False:
                assert lib.mpmae_set_option(_lib.OPT["DW"], 9) == 0
                ts0 = a.tiles_side
                for fl, what in ((0, "all"), (1, "no stores"), (2, "no B reads / MFMA"), (4, "no A build"), (8, "no fill loads"), (3, "no stores, MFMA"),
                                 (7, "no stores, MFMA, A"), (15, "nothing but add/act loads + LDS fill"), (9, "no global fill loads, no stores")):
                    a.tiles_side = ts0 | (fl << 16)
                    us = t_us(lambda: lib.mpmae_dwconv7_fwd(1, C.byref(a), st()))
                    print(f"      phases: {what:40s} {us:7.1f} us")
                a.tiles_side = ts0
            if check:
                xm = T._rows_to_map(e, x, S, True)
                for label, W in (("fp32 taps", Wd), ("bf16 taps", Wd.to(bf).float())):
                    if a.flip:
                        ym = F.conv2d(xm, W.flip(2, 3), None, padding=3, groups=Cc)
                    else:
                        ym = F.conv2d(xm, W, bias, padding=3, groups=Cc)
                    ref = T._map_to_rows(e, ym, S, True)
                    if add is not None:
                        ref = ref + add.float()
                    ref = ref * live
                    for dw in (7, 8):
                        err = (outs[dw].float() - ref).abs()
                        tol = (2.0 ** -7) * ref.abs() + (2.0 ** -9) * ref.abs().max()
                        print(f"    {name} DW={dw} vs torch ({label}): max err {err.max().item():.4e} (max|ref| {ref.abs().max().item():.3f}), "
                              f"worst err/tol {(err / tol).max().item():.3f}, rows zero at inactive: {bool((outs[dw][~live[:, 0]] == 0).all())}")


NC = int(os.environ.get("NCHECK", "6"))
print(f"== correctness (N = {NC})")
run(make(NC), True)
print("== timing (N = 256)")
run(make(256), False)


def run_wgrad(e, check):
    """weight gradient: DWW = 5 (VALU kernels) against DWW = 7 (matrix-core kernel at S = 8), against torch at small N"""
    ops = {o[0]: o for o in e.bwd_ops}
    for blk in T._blocks(e):
        if not blk["sparse"] or e.S[blk["stage"]] < 4:
            continue
        tag, M, Cc, S = blk["prefix"], blk["M"], blk["C"], e.S[blk["stage"]]
        live = e.act[blk["stage"]].bool()[:, None]
        name, fn, args, _ = ops[tag + ":dw.wgrad"]
        a = type(args[1]._obj).from_buffer_copy(args[1]._obj)
        torch.manual_seed(3 * M + Cc)
        x = torch.randn(M, Cc, device=DEV).to(bf) * live
        dd = (torch.randn(M, Cc, device=DEV) * 0.2).to(bf) * live
        ws = torch.empty(32 << 20, device=DEV)
        res = {}
        for dww in (5, 7):
            assert lib.mpmae_set_option(_lib.OPT["DWW"], dww) == 0
            dw = torch.zeros(49, Cc, device=DEV); db = torch.zeros(Cc, device=DEV)
            a.x, a.dd, a.dw, a.db, a.ws, a.ws_floats = x.data_ptr(), dd.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel()
            assert lib.mpmae_dwconv7_wgrad(1, C.byref(a), args[2], st()) == 0
            torch.cuda.synchronize()
            res[dww] = (dw.clone(), db.clone())
            us = t_us(lambda: lib.mpmae_dwconv7_wgrad(1, C.byref(a), args[2], st()))
            print(f"  {name:38s} S={S} C={Cc:4d} M={M:7d} DWW={dww}: {us:7.1f} us")
        print(f"    DWW 7 vs 5: dw rel {T._rel(res[7][0], res[5][0]):.2e}  db rel {T._rel(res[7][1], res[5][1]):.2e}")
        if check:
            xm = F.pad(T._rows_to_map(e, x, S, True), (3, 3, 3, 3)); dm = T._rows_to_map(e, dd, S, True); Hh = dm.shape[-1]
            ref = torch.stack([torch.stack([(dm * xm[:, :, kh:kh + Hh, kw:kw + Hh]).sum((0, 2, 3)) for kw in range(7)], 1) for kh in range(7)], 1)
            for dww in (5, 7):
                got = res[dww][0].view(7, 7, Cc).permute(2, 1, 0)
                print(f"    DWW={dww} vs torch: dw rel {T._rel(got, ref):.2e}  db rel {T._rel(res[dww][1], dd.float().sum(0)):.2e}")


print(f"== weight gradient: correctness (N = {NC})")
run_wgrad(make(NC), True)
print("== weight gradient: timing (N = 256)")
run_wgrad(make(256), False)
