"""Times one training step of the reference script's own configuration (tests.py:96-134: FB15k-237 sizes, DistMult d=100, random
5-d literals behind a tanh Dense, pairwize_hinge with the harness's all-pairs broadcast, Adam, B=2048, ns=10) on the GPU box."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kge_keras_amd.harness import build_model  # noqa: E402
from kge_keras_amd.sampler import NumpyStream, PythonStream, build_batch  # noqa: E402

N, R, B, ns = 14541, 237, 2048, 10
rs = np.random.RandomState(0)
for lit in (True, False):
    hp = {'num_entities': N, 'num_relations': R, 'dim': 100, 'embedding_model': 'DistMult', 'loss_function': 'pairwize_hinge', 'margin': 1, 'seed': 1}
    if lit:
        hp.update(literals=rs.rand(N, 5), literal_activation='tanh')
    mdl = build_model(hp)
    b = np.stack([rs.randint(0, N, B), rs.randint(0, R, B), rs.randint(0, N, B)], 1)
    pos, neg = build_batch(NumpyStream(0), PythonStream(0), b, ns, N)
    pos, neg = torch.from_numpy(pos.astype(np.int32)).cuda(), torch.from_numpy(neg.astype(np.int32)).cuda()
    for _ in range(3):
        mdl.train_step([pos, neg])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        mdl.train_step([pos, neg])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"reference script config, literals={'tanh Dense' if lit else 'none'} (dp=0.2 default): {dt*1e3:.3f} ms/step  {2*B*ns/dt/1e6:.1f} M scored triples/s")
