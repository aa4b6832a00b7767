"""Times kge_build_batch_device alone (GPU box)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kge_keras_amd.sampler import NumpyStream, PythonStream, DeviceSampler  # noqa: E402
from kge_keras_amd import _lib  # noqa: E402

B, ns, N = 2048, 100, 14541
rs = np.random.RandomState(0)
batch = torch.from_numpy(np.stack([rs.randint(0, N, B), rs.randint(0, 237, B), rs.randint(0, N, B)], 1).astype(np.int32)).cuda()
ds = DeviceSampler(NumpyStream(0), PythonStream(0))
for _ in range(3):
    ds.build_batch(batch, ns, N)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 20
for _ in range(reps):
    ds.build_batch(batch, ns, N)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"device sampler alone: {dt*1e3:.3f} ms per batch of {2*B*ns} triples -> {2*B*ns/dt/1e6:.0f} M triples/s")
