"""Fixed cost per kge_pipeline_run call: ms per step for runs of different lengths (GPU box): python tools/run_len_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

w = bench.workload("DistMult", 0)
model = bench.make_model(w)
pipe, dsamp = bench.make_pipeline(torch, model, w, 0, 20000)
pipe.start()
pipe.run(300, join=False)
torch.cuda.synchronize()
for n in (20, 20, 20, 100, 100, 1000, 20, 1000):
    t0 = time.perf_counter()
    pipe.run(n, join=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"mask {os.environ.get('KGE_PIPE_CUMASK', 'default')} run({n}): {dt * 1e3:.3f} ms total, {dt / n * 1e3:.4f} ms per step, enqueue {(t1 - t0) * 1e3:.3f} ms", flush=True)
pipe.finish()
