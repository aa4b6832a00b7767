"""Kernel-only time of the 1-vs-all sweep against the number of queries (GPU box): is there a fixed cost per launch?
python tools/rank_scaling.py [N d model] -- hipEvents of the library around the sweep kernel (kge_prof 'rank_sweep')."""
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from kge_keras_amd import models as M, _lib  # noqa: E402

N, d, model = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]) if len(sys.argv) > 3 else (14541, 200, "DistMult")
rs = np.random.RandomState(0)
mdl = getattr(M, model)(e_dim=d, r_dim=d, num_entities=N, num_relations=11, dp=0, seed=1)
fq, fi = np.zeros(0, np.int32), np.zeros(0, np.int32)
K = mdl.e_dim
for nq in (2560, 5120, 10240, 20466, 40932, 81864, 204660):
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
    tot, n2 = _lib.prof_read("rank")
    t = ms / n
    print(f"{model} N={N} K={K} nq={nq:7d}: sweep kernel {t:8.4f} ms  whole call {tot / n2:8.4f} ms  {2.0 * K * N * nq / t / 1e9:7.1f} TFLOP/s  "
          f"units/WG {((nq + 255) // 256) * ((N + 127) // 128) / 256:.2f}")
