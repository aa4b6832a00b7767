"""Host cost of hipGraph replays on this ROCm (GPU box): python tools/graph_probe.py"""
import time
import torch

x = torch.zeros(1 << 16, device="cuda")
def body(n):
    for _ in range(n):
        x.add_(1.0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for n in (8, 32):
    gs = []
    for _ in range(2):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body(n)
        gs.append(g)
    torch.cuda.synchronize()
    for label, streams in (("same stream", [s1, s1]), ("two streams", [s1, s2])):
        for _ in range(5):
            for g, st in zip(gs, streams):
                with torch.cuda.stream(st):
                    g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 100
        for _ in range(reps):
            for g, st in zip(gs, streams):
                with torch.cuda.stream(st):
                    g.replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{n} nodes, {label}: host {1e6*(t1-t0)/(2*reps):.1f} us/replay, total {1e6*(t2-t0)/(2*reps):.1f} us/replay")
    t0 = time.perf_counter()
    for _ in range(100):
        with torch.cuda.stream(s1):
            body(n)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n} kernels, stream launches: host {1e6*(t1-t0)/100:.1f} us, total {1e6*(t2-t0)/100:.1f} us")
