import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from kge_keras_amd import models as M, _lib
N, d, nq = 14541, 200, 20466
rs = np.random.RandomState(0)
for scale in (0.02, 1.0, 30.0):
    E0 = rs.uniform(-scale, scale, (N, d)).astype(np.float32); R0 = rs.uniform(-scale, scale, (11, d)).astype(np.float32)
    mdl = M.DistMult(e_dim=d, r_dim=d, num_entities=N, num_relations=11, dp=0, init_entities=E0, init_relations=R0)
    test = torch.from_numpy(np.stack([rs.randint(0, N, nq), rs.randint(0, 11, nq), rs.randint(0, N, nq)], 1).astype(np.int32)).cuda()
    fq, fi = np.zeros(0, np.int32), np.zeros(0, np.int32)
    for side in (0, 2):
        for _ in range(2): mdl.rank_counts(test, side, fq, fi)
        torch.cuda.synchronize(); _lib.lib.kge_prof_enable(1)
        for _ in range(5): mdl.rank_counts(test, side, fq, fi)
        torch.cuda.synchronize(); _lib.lib.kge_prof_enable(0)
        ms, n = _lib.prof_read("rank_sweep"); _lib.prof_read("rank")
        q = nq * (2 if side == 2 else 1)
        print(f"scale {scale}: side {side}: sweep {ms/n:.4f} ms  {2.0*d*N*q/(ms/n)/1e9:.1f} TFLOP/s")
