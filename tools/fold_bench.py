"""Per-label kernel times of the default train step on reference-shaped batches (GPU box): python tools/fold_bench.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kge_keras_amd import _lib, models as M  # noqa: E402
from kge_keras_amd.sampler import NumpyStream, PythonStream, build_batch  # noqa: E402

N, R, d, B, ns = 14541, 237, 200, 2048, 100
mdl = M.DistMult(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, seed=1, loss_function="pointwize_hinge")
mdl.compile("adam")
rs = np.random.RandomState(0)
batches = []
for k in range(4):
    if os.environ.get("SKEW"):  # relation and entity frequencies like a real graph: a few dominate (Zipf-like)
        pr = 1.0 / np.arange(1, R + 1) ** 1.1
        pe = 1.0 / np.arange(1, N + 1) ** 0.9
        b = np.stack([rs.choice(N, B, p=pe / pe.sum()), rs.choice(R, B, p=pr / pr.sum()), rs.choice(N, B, p=pe / pe.sum())], 1)
    else:
        b = np.stack([rs.randint(0, N, B), rs.randint(0, R, B), rs.randint(0, N, B)], 1)
    p, n = build_batch(NumpyStream(k), PythonStream(k), b, ns, N)
    batches.append((torch.from_numpy(p.astype(np.int32)).cuda(), torch.from_numpy(n.astype(np.int32)).cuda()))
for i in range(5):
    mdl.train_step(list(batches[i % 4]))
torch.cuda.synchronize()
_lib.lib.kge_prof_enable(1)
reps = 40
t0 = time.perf_counter()
for i in range(reps):
    mdl.train_step(list(batches[i % 4]))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
_lib.lib.kge_prof_enable(0)
out = {k: _lib.prof_read(k) for k in ("dedup", "index", "fwd", "train_grads", "opt")}
print(f"KGE_DEDUP={os.environ.get('KGE_DEDUP', '1')}: {dt*1e3:.3f} ms/step {2*B*ns/dt/1e6:.0f} M/s  " +
      "  ".join(f"{k} {ms/max(1,n)*1e3:.1f}us" for k, (ms, n) in out.items()))
