"""Wall time per step of the native pipeline with stages switched off AFTER a warm-up (the buffers of the skipped stages keep their
last contents, so the remaining stages still work on real batches): python tools/stage_probe.py  (GPU box).
Stage bits: 1 generator rounds, 2 draw, 4 plan, 8 step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

w = bench.workload("DistMult", 0)
model = bench.make_model(w)
pipe, dsamp = bench.make_pipeline(torch, model, w, 0, 6000)
pipe.start()
pipe.run(300, join=False)
torch.cuda.synchronize()
for mask in (15, 13, 12, 14, 15):  # (never a mask with plan but without step: an unconsumed plan leaves its counters set and the next build would rank out of range)
    os.environ["KGE_PIPE_STAGES"] = str(mask)
    pipe.run(50, join=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.run(500, join=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"stages {mask:2d}: {dt / 500 * 1e3:.4f} ms per step", flush=True)
os.environ["KGE_PIPE_STAGES"] = "15"
