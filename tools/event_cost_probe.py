"""What an event record / a cross-stream wait costs the stream it is issued on (GPU time between dependent small kernels)."""
import torch, time
dev = "cuda"
x = torch.zeros(1 << 16, device=dev)
s1 = torch.cuda.Stream(); s2 = torch.cuda.Stream()
N = 400


def run(kind):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    evs = [torch.cuda.Event() for _ in range(N)]
    with torch.cuda.stream(s1):
        e0.record()
        for i in range(N):
            x.add_(1.0)
            if kind in ("record", "record+wait"):
                evs[i].record(s1)
            if kind == "record+wait":
                s2.wait_event(evs[i])
                with torch.cuda.stream(s2):
                    x2.add_(1.0)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N * 1e3


x2 = torch.zeros(1 << 16, device=dev)
for kind in ("plain", "record", "record+wait", "plain", "record"):
    run(kind)
    print(f"{kind:12s} {run(kind):6.2f} us per small kernel on the recording stream")
