"""One line per BASELINE.json config on one MI355X (GPU box): python tools/config_bench.py > gpurun_out/configs.json

configs[0] nations DistMult d=32, 10 negatives (plumbing: the reference script's flow on the real nations triples)
configs[1] FB15k-237 DistMult d=200, 100 negatives        -> bench.py is the contract line; here the bare train step for comparison
configs[2] FB15k-237 TransE-L2 d=200, 100 negatives        (train step + distance sweep of the ranking)
configs[3] WN18RR ComplEx d=400, 1-vs-all on the fp32 matrix cores
configs[4] YAGO3-10 DistMult d=200, 1-vs-all over the full table on ONE GPU (the 8-GPU row-sharded form is bench.py --gpus 8)
Train numbers: kge_train_step on pre-built batches of B*ns random rows per array, Adam, no dropout (tools/train_bench.py).
Rank numbers: kge_rank_counts, tail side, no filters (tools/rank_bench.py); TFLOP/s = 2*K*N per rank.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_keras_amd import models as M  # noqa: E402


def train(model, N, R, d, B, ns, reps=10, **kw):
    mdl = getattr(M, model)(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0, seed=1, loss_function="pointwize_hinge", **kw)
    mdl.compile("adam")
    rs = np.random.RandomState(0)
    m = B * ns
    mk = lambda: torch.from_numpy(np.stack([rs.randint(0, N, m), rs.randint(0, R, m), rs.randint(0, N, m)], 1).astype(np.int32)).cuda()  # noqa: E731
    pos, neg = mk(), mk()
    for _ in range(3):
        mdl.train_step([pos, neg])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        mdl.train_step([pos, neg])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return {"ms_per_step": dt * 1e3, "scored_triples_per_s": 2 * m / dt}


def rank(model, N, R, d, nq, reps=5, **kw):
    mdl = getattr(M, model)(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0, seed=1, **kw)
    rs = np.random.RandomState(0)
    test = torch.from_numpy(np.stack([rs.randint(0, N, nq), rs.randint(0, R, nq), rs.randint(0, N, nq)], 1).astype(np.int32)).cuda()
    fq, fi = np.zeros(0, np.int32), np.zeros(0, np.int32)
    for _ in range(2):
        mdl.rank_counts(test, 0, fq, fi)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        mdl.rank_counts(test, 0, fq, fi)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    K = int(mdl.e_dim)
    out = {"ms_per_sweep": dt * 1e3, "ranks_per_s": nq / dt, "queries": nq, "row_floats": K}
    if model != "TransE" or kw.get("norm") == 2:  # TransE-L2 ranks on the matrix cores too (|e|^2 - 2<q,e>)
        out["mfma_tflops_algorithmic"] = 2.0 * K * N * nq / dt / 1e12
        out["mfma_frac_of_157.3"] = out["mfma_tflops_algorithmic"] / 157.3
    return out


def nations():
    """The reference script's flow (tests.py:96-134) on the real nations triples: DataGenerator -> fit -> validate."""
    from kge_keras_amd.harness import DataGenerator, build_model
    from kge_keras_amd.utils import validate
    ds = np.load(os.path.join(ROOT, "tests", "golden", "datasets.npz"))
    tr = ds["nations_train"] if "nations_train" in ds else None
    if tr is None:
        return {"skipped": "no nations split in tests/golden/datasets.npz"}
    N, R = int(tr[:, [0, 2]].max()) + 1, int(tr[:, 1].max()) + 1
    np.random.seed(0)
    import random
    random.seed(0)
    model = build_model(dict(num_entities=N, num_relations=R, dim=32, embedding_model="DistMult", loss_function="pointwize_hinge", margin=1,
                             negative_samples=10, batch_size=32, dp=0.0))
    gen = DataGenerator(tr, ns=10, batch_size=32)
    t0 = time.perf_counter()
    h = model.fit(gen, epochs=5, verbose=0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = 5 * len(gen)
    test = [tuple(int(v) for v in t) for t in tr[:200]]
    res = validate(model, test, N, None, filtering_triples=[])
    return {"steps": steps, "ms_per_step_host_generator_included": dt / steps * 1e3, "scored_triples_per_s": 2 * 320 * steps / dt,
            "final_loss": float(h.history["loss"][-1]) if hasattr(h, "history") else None, "mrr_on_200_train_triples": float(res["mrr"])}


if __name__ == "__main__":
    out = {"device": torch.cuda.get_device_name(0)}
    try:
        out["configs[0] nations DistMult d=32 ns=10"] = nations()
    except Exception as e:  # plumbing case: report, do not hide
        out["configs[0] nations DistMult d=32 ns=10"] = {"error": repr(e)}
    out["configs[1] FB15k-237 DistMult d=200 ns=100"] = {"train_step": train("DistMult", 14541, 237, 200, 2048, 100), "rank": rank("DistMult", 14541, 237, 200, 20466)}
    out["configs[2] FB15k-237 TransE-L2 d=200 ns=100"] = {"train_step": train("TransE", 14541, 237, 200, 2048, 100, norm=2, gamma=12),
                                                          "rank": rank("TransE", 14541, 237, 200, 20466, norm=2, gamma=12)}
    out["configs[3] WN18RR ComplEx d=400 1-vs-all"] = {"rank": rank("ComplEx", 40943, 11, 400, 6268)}
    out["configs[4] YAGO3-10 DistMult d=200 1-vs-all (one GPU, full table)"] = {"rank": rank("DistMult", 123182, 37, 200, 10000)}
    out["other models, FB15k-237 shape"] = {"ComplEx d=200": train("ComplEx", 14541, 237, 200, 2048, 100), "HolE d=200 (frequency domain)": train("HolE", 14541, 237, 200, 2048, 100),
                                            "pRotatE d=100": train("pRotatE", 14541, 237, 100, 2048, 100)}
    print(json.dumps(out, indent=1))
