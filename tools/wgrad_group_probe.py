"""Developer timing of the GROUPED weight-gradient launch (mpmae_wgrad_group, gemm_tng.cuh) at the encoder stage shapes of the bench
step (bs 256), alone on the GPU, beside the same problems issued one by one through mpmae_wgrad (what round 3 paid)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmearth_train_amd import _lib
lib = _lib.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for kv in os.environ.get("LIBOPTS", "").split(","):
    if kv:
        k, v = kv.split("=")
        assert lib.mpmae_set_option(_lib.OPT[k], int(v)) == 0
ws = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
for M, Cc, blocks, name in [(4864, 320, 2, "stage 3"), (19456, 160, 6, "stage 2"), (77824, 80, 2, "stage 1")]:
    H, count = 4 * Cc, 2 * blocks
    arr = (_lib.WgradArgs * count)()
    keep = []
    torch.manual_seed(1)
    for i in range(count):
        Nn, Kk = (Cc, H) if i % 2 == 0 else (H, Cc)
        Pm = torch.randn(M, Nn, device="cuda").to(torch.bfloat16)
        Qm = torch.randn(M, Kk, device="cuda").to(torch.bfloat16)
        dW = torch.zeros(Nn, Kk, device="cuda"); db = torch.zeros(Nn, device="cuda")
        a = arr[i]
        a.P, a.Q, a.M, a.Nn, a.Kk, a.ldp, a.ldq = Pm.data_ptr(), Qm.data_ptr(), M, Nn, Kk, Nn, Kk
        a.dW, a.sn, a.sk, a.db = dW.data_ptr(), Kk, 1, db.data_ptr()
        a.ws, a.ws_floats = ws.data_ptr(), ws.numel()
        keep.append((Pm, Qm, dW, db))
    assert lib.mpmae_wgrad_group(1, arr, count, C.c_void_p(ws.data_ptr()), ws.numel(), st) == 0
    torch.cuda.synchronize()
    err = max(((dW - Pm.float().t() @ Qm.float()).abs().max() / (Pm.float().t() @ Qm.float()).abs().max()).item() for Pm, Qm, dW, _ in keep)
    us_g = t(lambda: lib.mpmae_wgrad_group(1, arr, count, C.c_void_p(ws.data_ptr()), ws.numel(), st))

    def one_by_one():
        for i in range(count):
            lib.mpmae_wgrad(1, 0, 0, C.byref(arr[i]), 16, st)
    us_1 = t(one_by_one)
    fl = 2.0 * M * Cc * H * count
    by = M * (Cc + H) * 2.0 * count
    print(f"{name}: {count:2d} problems M={M:6d} C={Cc:3d}: grouped {us_g:7.1f} us ({fl/us_g/1e6:5.0f} TF, {by/us_g/1e6:5.2f} TB/s of operands)"
          f" | one by one {us_1:7.1f} us | err {err:.1e}", flush=True)
