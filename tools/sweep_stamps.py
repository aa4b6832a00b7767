"""Diagnostic (GPU box): when does each workgroup of rank_sweep5_kernel finish each of its units?  Needs the stamped build:
tools/build_variant.sh stamp "-DKGE_SWEEP5_STAMP -mllvm -amdgpu-mfma-vgpr-form" rank.hip; KGE_HIP_LIB=kge-keras_amd/csrc/variants/libkge_hip_stamp.so
python tools/sweep_stamps.py [N d nq] -- prints, per decile of a workgroup's units, the mean time per unit (us)."""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from kge_keras_amd import models as M, _lib  # noqa: E402

N, d, nq = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (14541, 200, 20466)
rs = np.random.RandomState(0)
mdl = M.DistMult(e_dim=d, r_dim=d, num_entities=N, num_relations=11, dp=0, seed=1)
test = torch.from_numpy(np.stack([rs.randint(0, N, nq), rs.randint(0, 11, nq), rs.randint(0, N, nq)], 1).astype(np.int32)).cuda()
fq, fi = np.zeros(0, np.int32), np.zeros(0, np.int32)
for _ in range(3):
    mdl.rank_counts(test, 0, fq, fi)
torch.cuda.synchronize()
U = 512
buf = (C.c_ulonglong * (256 * U))()
assert _lib.lib.kge_debug_sweep5_stamps(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(256, U).astype(np.int64)
n = a[:, 0]
t0 = min(a[b, 1] for b in range(256) if n[b] > 0)
ends = np.array([a[b, n[b]] for b in range(256) if 0 < n[b] < U - 1]) - t0
print(f"units per workgroup {n.min()}..{n.max()}; last stamp of a workgroup (us after the first stamp of all): min {ends.min() / 100:.1f} median {np.median(ends) / 100:.1f} max {ends.max() / 100:.1f}")
k = int(n.max())
per = np.full((256, k), np.nan)
for b in range(256):
    if n[b] > 1:
        d_ = np.diff(a[b, 1:1 + n[b]]) / 100.0
        per[b, 1:1 + len(d_)] = d_
print("unit index : mean us per unit over the workgroups (first stamp has no predecessor)")
for lo in range(1, k, max(1, k // 12)):
    hi = min(k, lo + max(1, k // 12))
    print(f"  units {lo:3d}..{hi - 1:3d}: {np.nanmean(per[:, lo:hi]):7.2f} us   (slowest workgroup {np.nanmax(np.nanmean(per[:, lo:hi], axis=1)):7.2f})")
