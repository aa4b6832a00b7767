"""Does the 1-vs-all sweep's speed depend on where its operands come from?  Same flops, candidate tables of 14 541 / 4 096 / 2 048 / 1 024 rows
(11.6 MB .. 0.8 MB: past the L2s .. resident in every L2), kernel-only hipEvents of the library: python tools/rank_l2_probe.py  (GPU box)."""
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from kge_keras_amd import models as M, _lib  # noqa: E402

d = 200
pairs = 14541 * 163728
rs = np.random.RandomState(0)
fq, fi = np.zeros(0, np.int32), np.zeros(0, np.int32)
for N in (14541, 4096, 2048, 1024, 14541):
    nq = pairs // N
    mdl = M.DistMult(e_dim=d, r_dim=d, num_entities=N, num_relations=11, dp=0, seed=1)
    test = torch.from_numpy(np.stack([rs.randint(0, N, nq), rs.randint(0, 11, nq), rs.randint(0, N, nq)], 1).astype(np.int32)).cuda()
    for _ in range(2):
        mdl.rank_counts(test, 0, fq, fi)
    torch.cuda.synchronize()
    _lib.lib.kge_prof_enable(1)
    for _ in range(5):
        mdl.rank_counts(test, 0, fq, fi)
    torch.cuda.synchronize()
    _lib.lib.kge_prof_enable(0)
    ms, n = _lib.prof_read("rank_sweep")
    _lib.prof_read("rank")
    t = ms / n
    print(f"N={N:6d} ({N * d * 4 / 1e6:5.1f} MB of candidates) nq={nq:8d}: sweep kernel {t:8.3f} ms  {2.0 * d * N * nq / t / 1e9:7.1f} TFLOP/s", flush=True)
    del mdl, test
    torch.cuda.empty_cache()
