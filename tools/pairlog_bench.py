"""pairwize_logistic with the harness's all-pairs broadcast (utils.py:307-312, negative_samples = 1): the loss + both gradients on two
score vectors of m rows each, all-pairs kernels (mode 0: one sweep per side) against the one-pass tiled kernel (mode 1, csrc/pairlog.hip);
library hipEvents around the pairwise kernels + wall time of the whole kge_loss_value call.  python tools/pairlog_bench.py  (GPU box)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kge_keras_amd import models as M, _lib  # noqa: E402

rs = np.random.RandomState(0)
for m, modes in ((20480, (0, 1)), (204800, (1,))):
    t = torch.from_numpy(rs.uniform(-3, 6, m).astype(np.float32)).cuda()
    f = torch.from_numpy(rs.uniform(-6, 3, m).astype(np.float32)).cuda()
    for mode in modes:
        _lib.check(_lib.lib.kge_set_pair_sort_mode(mode))
        for _ in range(2):
            L = M.loss_value("pairwize_logistic", t, f, 1.0, 1, return_grads=True)[0]
        torch.cuda.synchronize()
        reps = 10 if m <= 20480 or mode == 1 else 2
        _lib.lib.kge_prof_enable(1)
        t0 = time.perf_counter()
        for _ in range(reps):
            L = M.loss_value("pairwize_logistic", t, f, 1.0, 1, return_grads=True)[0]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        _lib.lib.kge_prof_enable(0)
        ms, n = _lib.prof_read("pairwise")
        print(f"m = {m:7d} ({m * m:.2e} pairs)  {'one pass' if mode else 'all-pairs kernels, one sweep per side'}: kernels {ms / max(1, n):9.4f} ms  "
              f"whole call {dt * 1e3:9.4f} ms  {m * m / (ms / max(1, n) * 1e-3) / 1e12:6.2f} T pairs/s  loss {float(L.item()):.6f}", flush=True)
_lib.lib.kge_set_pair_sort_mode(-1)
