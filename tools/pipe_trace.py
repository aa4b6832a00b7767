"""Stage timeline of the native training pipeline on the bench workload (GPU box): python tools/pipe_trace.py [n_launches]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
w = bench.workload("DistMult", 0)
model = bench.make_model(w)
pipe, dsamp = bench.make_pipeline(torch, model, w, 0, 1000)
pipe.start()
pipe.run(300)
torch.cuda.synchronize()
pipe.trace(n)
pipe.run(n)
rows = pipe.trace()
for name, batch, a, b in sorted(rows, key=lambda r: r[2]):
    print(f"{a:9.1f} {b - a:7.1f}  {name:6s} {batch}")
pipe.run(100)
pipe.finish()
# per-kernel durations INSIDE the concurrent pipeline (the library's hipEvent hooks around its dominant kernels)
from kge_keras_amd import _lib  # noqa: E402
w2 = bench.workload("DistMult", 0)
model2 = bench.make_model(w2)
pipe2, _ = bench.make_pipeline(torch, model2, w2, 0, 1000)
pipe2.start()
pipe2.run(200)
torch.cuda.synchronize()
_lib.lib.kge_prof_enable(1)
pipe2.run(200)
torch.cuda.synchronize()
_lib.lib.kge_prof_enable(0)
for lab in ("fwd", "train_grads", "opt", "index", "sampler", "dedup"):
    ms, k = _lib.prof_read(lab)
    print(f"concurrent {lab:12s} avg {ms / max(1, k) * 1e3:8.1f} us over {k} scopes")
pipe2.finish()
