"""What one rank of an 8-rank sharded evaluation computes, timed on one GPU without any exchange (YAGO3-10 shape): schedule (i) =
1/8 of the queries against the whole table; schedule (ii) = all queries against 1/8 of the rows.  python tools/eval_shard_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from kge_keras_amd import dist as D  # noqa: E402
from kge_keras_amd import models as KM  # noqa: E402

Y = bench.YAGO
N, R, d, nt = Y["N"], Y["R"], Y["d"], Y["n_test"]
rs = np.random.RandomState(4321)
E0 = rs.uniform(-0.1, 0.1, size=(N, d)).astype(np.float32)
R0 = rs.uniform(-0.4, 0.4, size=(R, d)).astype(np.float32)
test = np.stack([rs.randint(0, N, nt), rs.randint(0, R, nt), rs.randint(0, N, nt)], 1).astype(np.int32)
fq_all = np.repeat(np.arange(nt, dtype=np.int32), 3)
fi_all = rs.randint(0, N, 3 * nt).astype(np.int32)
keep = (fi_all != np.repeat(test[:, 2], 3)) & (fi_all != np.repeat(test[:, 0], 3))
fq_all, fi_all = fq_all[keep], fi_all[keep]
model = KM.DistMult(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, init_entities=E0, init_relations=R0)
for world in (1, 2, 4, 8):
    rank = 0
    idx = D.split_queries(nt, rank, world).numpy()
    mine = torch.from_numpy(test[idx]).cuda()
    sel = np.isin(fq_all, idx)
    fq1 = np.searchsorted(idx, fq_all[sel]).astype(np.int32)
    fq1 = torch.from_numpy(np.concatenate([fq1, fq1 + len(idx)])).cuda()  # side 2: head queries numbered behind the tail queries
    fi1 = torch.from_numpy(np.concatenate([fi_all[sel], fi_all[sel]])).cuda()
    lo, hi = D.shard_range(N, rank, world)
    shard = model.E[lo:hi].clone()
    fq2, fi2 = torch.from_numpy(np.concatenate([fq_all, fq_all + nt])).cuda(), torch.from_numpy(np.concatenate([fi_all, fi_all])).cuda()

    def s1():
        return 1 + model.rank_counts(mine, 2, fq1, fi1).long()

    plan2 = D.ResidentShardPlan(test, shard.shape[0], lo, shard.device)

    def s2():
        return plan2.run(lambda ref, local, side: model.rank_counts_shard(ref, local, side, fq2, fi2, shard, lo), shard, both=True)

    for name, fn in (("i ", s1), ("ii", s2)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"world {world} schedule {name}: {dt * 1e3:7.3f} ms per evaluation on one rank (no exchange) -> {2 * nt / dt / 1e6:6.2f} M ranks/s if the exchange were free", flush=True)
