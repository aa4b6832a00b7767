"""The benched training path on a table that does NOT fit the 256 MB Infinity Cache (GPU box): the BASELINE configs[1] batch shape
(B = 2048, 100 negatives, DistMult d = 200, pointwize_hinge + Keras Adam) on N entities (default 2 000 000: a 1.6 GB table, 4.8 GB
with the Adam moments, plus its twin), same native pipeline (device sampler, counted plans, steps between twin tables).  Keras' Adam
updates EVERY row every step (dense semantics), so at this size the step is the optimizer's streams: 24 N d bytes per step.
python tools/scale_bench.py [N] [steps]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from kge_keras_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bench.CFG["N"] = N
w = bench.workload("DistMult", 0)
model = bench.make_model(w)
pipe, dsamp = bench.make_pipeline(torch, model, w, 0, 3 * steps + 20)
pipe.start()
pipe.run(10, join=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
pipe.run(steps, join=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
pipe.finish()
# kernel alone: a serial pass of the same object
pipe.use_graphs, pipe.serial = False, True
nb = bench.CFG["n_train"] // bench.CFG["B"]
pipe.load_epoch(pipe.index, [i % nb for i in range(32)])
pipe.start()
torch.cuda.synchronize()
_lib.lib.kge_prof_enable(1)
pipe.run(30)
torch.cuda.synchronize()
_lib.lib.kge_prof_enable(0)
pipe.finish()
k_ms, k_n = _lib.prof_read("train_grads")
L = _lib.CAND_POS_LANES * bench.CFG["B"]
rows = float(sum(int((c[:L].view(_lib.CAND_POS_LANES, bench.CFG["B"]).sum(0) > 0).sum() + (c[L:] > 0).sum()) for _r, c in pipe.cand) / 2)
d = bench.CFG["d"]
algo = rows * 16 * d + 24 * N * d
model._raise_on_flag()
print(json.dumps({"N": N, "table_MB": N * d * 4 / 1e6, "ms_per_step": dt * 1e3, "scored_triples_per_s": 2 * bench.CFG["B"] * bench.CFG["ns"] / dt,
                  "step1_kernel_ms": k_ms / max(1, k_n), "rows_per_launch": rows, "algorithmic_bytes_per_launch": algo,
                  "step1_algorithmic_GBps": algo / (k_ms / max(1, k_n) * 1e-3) / 1e9, "frac_of_8TBps": algo / (k_ms / max(1, k_n) * 1e-3) / 8e12,
                  "whole_step_GBps_on_optimizer_streams_alone": 24 * N * d / dt / 1e9}))
