"""The three Dense GEMMs of a ConvE call at the bench shape (m = 204 800, F = 2 592, d = 200) through torch.mm (whatever BLAS back end
PyTorch-ROCm picks), for comparison with the rocBLAS sgemm calls of conve.hip: python tools/gemm_probe.py  (GPU box)."""
import os
import time

import torch

m, F, d = 204800, 2592, 200
A = torch.randn(m, F, device="cuda")
W = torch.randn(F, d, device="cuda")
dZ = torch.randn(m, d, device="cuda")
for name, fn in (("Z = A W", lambda: torch.mm(A, W)), ("dA = dZ W^T", lambda: torch.mm(dZ, W.t())), ("dW = A^T dZ", lambda: torch.mm(A.t(), dZ))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{name:14s} {dt * 1e3:.3f} ms  {2.0 * m * F * d / dt / 1e12:.1f} TFLOP/s   (TORCH_BLAS_PREFER_HIPBLASLT={os.environ.get('TORCH_BLAS_PREFER_HIPBLASLT')})", flush=True)
