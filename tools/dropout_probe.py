"""ms per step of the pipeline on the bench workload with the reference's default dropout (dp = 0.2: no duplicate folding, a mask per
oversampled row): python tools/dropout_probe.py  (GPU box)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from kge_keras_amd import models as KM  # noqa: E402

w = bench.workload("DistMult", 0)
C = bench.CFG
model = KM.DistMult(e_dim=C["d"], r_dim=C["d"], num_entities=C["N"], num_relations=C["R"], dp=0.2, init_entities=w["E0"], init_relations=w["R0"],
                    loss_function="pointwize_hinge", margin=1)
model.compile("adam")
pipe, dsamp = bench.make_pipeline(torch, model, w, 0, 3000)
pipe.start()
pipe.run(100)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    pipe.run(500, join=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"dp 0.2 native {pipe._native_active}: {dt / 500 * 1e3:.4f} ms per step = {2 * C['B'] * C['ns'] * 500 / dt / 1e9:.3f} G scored triples/s", flush=True)
pipe.finish()
