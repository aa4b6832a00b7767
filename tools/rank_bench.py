"""Times kge_rank_counts alone (GPU box): python tools_rank_bench.py [N d nq model]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from kge_keras_amd import models as M, _lib  # noqa: E402

N, d, nq, model = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]) if len(sys.argv) > 4 else (14541, 200, 20466, "DistMult")
rs = np.random.RandomState(0)
kw = dict(norm=2, gamma=12) if model == "TransE" else {}
mdl = getattr(M, model)(e_dim=d, r_dim=d, num_entities=N, num_relations=11, dp=0, seed=1, **kw)
test = torch.from_numpy(np.stack([rs.randint(0, N, nq), rs.randint(0, 11, nq), rs.randint(0, N, nq)], 1).astype(np.int32)).cuda()
fq, fi = np.zeros(0, np.int32), np.zeros(0, np.int32)
for _ in range(2):
    mdl.rank_counts(test, 0, fq, fi)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    mdl.rank_counts(test, 0, fq, fi)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
K = mdl.e_dim
print(f"{model} N={N} K={K} nq={nq}: {dt*1e3:.3f} ms/sweep  {nq/dt/1e6:.2f} M ranks/s  {2.0*K*N*nq/dt/1e12:.1f} TFLOP/s-equivalent")
