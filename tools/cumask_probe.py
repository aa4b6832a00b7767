"""ms per step of the native pipeline driven from a non-default stream (so that CU-masked producer streams, which are blocking
streams, do not synchronise with it): KGE_PIPE_CUMASK=k [KGE_PIPE_CUGROUP=g] python tools/cumask_probe.py  (GPU box)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

w = bench.workload("DistMult", 0)
model = bench.make_model(w)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    pipe, dsamp = bench.make_pipeline(torch, model, w, 0, 4000)
    pipe.start()
    pipe.run(300)
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        pipe.run(1000)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"cumask {os.environ.get('KGE_PIPE_CUMASK', '-')} group {os.environ.get('KGE_PIPE_CUGROUP', '-')} keep {os.environ.get('KGE_PIPE_CUKEEP', '1')}: {dt / 1000 * 1e3:.4f} ms per step", flush=True)
    pipe.finish()
    torch.cuda.synchronize()
    assert int(dsamp.err.item()) == 0
