"""Times kge_train_step alone for any model (GPU box): python tools/train_bench.py [model loss d B ns]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kge_keras_amd import models as M  # noqa: E402

model, loss, d, B, ns = (sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else ("HolE", "pointwize_hinge", 200, 2048, 100)
N, R = 14541, 237
kw = dict(norm=2, gamma=12) if model == "TransE" else ({"hidden_dp": float(os.environ.get("HDP", "0"))} if model == "ConvE" else {})
dp = float(os.environ.get("DP", "0"))
mdl = getattr(M, model)(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=dp, seed=1, loss_function=loss, **kw)
mdl.compile("adam")
rs = np.random.RandomState(0)
m = B * ns
pos = torch.from_numpy(np.stack([rs.randint(0, N, m), rs.randint(0, R, m), rs.randint(0, N, m)], 1).astype(np.int32)).cuda()
neg = torch.from_numpy(np.stack([rs.randint(0, N, m), rs.randint(0, R, m), rs.randint(0, N, m)], 1).astype(np.int32)).cuda()
for _ in range(3):
    mdl.train_step([pos, neg])
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    mdl.train_step([pos, neg])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"{model} {loss} d={d} m={m} dp={dp}: {dt*1e3:.3f} ms/step  {2*m/dt/1e6:.1f} M scored triples/s  loss={float(mdl._loss.item()):.4f}")
