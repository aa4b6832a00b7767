"""Where a rehearsed owner-computes step spends its time (two gloo ranks on ONE GPU; gloo stages through the host, so this says nothing
about xGMI -- it separates the window step from the exchange): python tools/rows_probe.py  (GPU box)."""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(rank, world):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import bench
    from kge_keras_amd import dist as D
    w = bench.workload("DistMult", 0)
    model = bench.make_model(w)
    tr = D.RowShardTrainer(model)
    rs = np.random.RandomState(5)
    N, R, m = bench.CFG["N"], bench.CFG["R"], bench.CFG["B"] * bench.CFG["ns"]
    tri = lambda n: np.stack([rs.randint(0, N, n), rs.randint(0, R, n), rs.randint(0, N, n)], 1).astype(np.int32)  # noqa: E731
    batches = [(tri(m), tri(m)) for _ in range(4)]
    t_plan = t_step = t_ex = 0.0
    K = 12
    for i in range(K + 2):
        pos, neg = batches[i % 4]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        plan = model.plan_batch([pos, neg])
        torch.cuda.synchronize(); t1 = time.perf_counter()
        ex = tr.tables.exchange
        tr.tables.exchange = lambda loss=None: loss  # the window step alone ...
        tr.step(plan)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        tr.tables.exchange = ex
        tr.tables.exchange(tr.loss)                   # ... then the exchange
        tr._install()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        if i >= 2:
            t_plan += t1 - t0; t_step += t2 - t1; t_ex += t3 - t2
    if rank == 0:
        print(f"rank 0 of {world}: plan build {t_plan / K * 1e3:.3f} ms, window step {t_step / K * 1e3:.3f} ms, exchange ({tr.tables.exchanged_bytes()} B received) "
              f"{t_ex / K * 1e3:.3f} ms per step", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(run, args=(2,), nprocs=2, join=True)
