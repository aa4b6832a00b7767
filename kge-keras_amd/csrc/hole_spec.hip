// HolE without dropout, in the frequency domain.
//
// Reference: KGEkeras/models.py:231-237 computes  score = sum_k p_k * real(ifft(conj(fft(s)) * fft(o)))_k.
// With the DFTs S, P, O of the three (real) rows, Parseval gives
//     score = (1/d) * sum_f Re( conj(P_f) * conj(S_f) * O_f ) = (1/d) * sum_f Re( S_f * P_f * conj(O_f) ),
// which is the ComplEx score (models.py:218-222) of the spectra.  Real rows have Hermitian spectra (X_{d-f} = conj(X_f)), so the
// sum over all d frequencies folds onto the h = floor(d/2)+1 lowest ones with weights c_f = 1 (f = 0, and f = d/2 for even d)
// or 2 (all others).  A training step therefore
//     1. transforms the two tables once:  Ehat = E * WE,  Rhat = Rel * WR           (kge_hole_spec_forward, one sgemm each)
//        WE[i][f] = cos(2 pi f i / d), WE[i][hp+f] = -sin(2 pi f i / d) for f < h, zero columns up to hp (a multiple of 4);
//        WR = WE with column f scaled by c_f / d;
//     2. runs the O(d) ComplEx-shaped kernels of the ordinary training path on (Ehat, Rhat) -- rows of 2*hp floats;
//     3. carries the gradients back through the (linear) transforms:  gE += gEhat * WE^T,  gRel += gRhat * WR^T
//        (kge_hole_spec_adjoint), and re-zeroes the spectral accumulators.
// The O(d^2) correlation per triple of hole.hip disappears; that file still serves dropout steps (every triple masks its
// own rows before the correlation) and the query vectors of the 1-vs-all sweep.
#include <rocblas/rocblas.h>

#include "kge_internal.hpp"

namespace kge {

rocblas_handle blas_handle_shared();  // conve.hip

#define KGE_BLAS_TRY(expr)                                                                  \
  do {                                                                                      \
    rocblas_status s__ = (expr);                                                            \
    if (s__ != rocblas_status_success) return fail(KGE_EHIP, "%s: rocblas status %d", #expr, (int)s__); \
  } while (0)

int hole_spec_half(int d) { return (int)align_up((int64_t)(d / 2 + 1), 4); }

// One thread per (i, f): exact phase by integer reduction of f*i mod d, evaluated in double.
__global__ void hole_basis_kernel(int d, int h, int hp, float* __restrict__ WE, float* __restrict__ WR) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)d * hp) return;
  const int i = (int)(t / hp), f = (int)(t - (int64_t)i * hp);
  float c = 0.f, s = 0.f, w = 0.f;
  if (f < h) {
    const int64_t r = ((int64_t)f * i) % d;
    double sn, cs;
    sincospi(2.0 * (double)r / (double)d, &sn, &cs);
    c = (float)cs;
    s = (float)(-sn);
    const bool self_conj = f == 0 || 2 * f == d;
    w = (self_conj ? 1.f : 2.f) / (float)d;
  }
  const int64_t row = (int64_t)i * 2 * hp;
  WE[row + f] = c;
  WE[row + hp + f] = s;
  WR[row + f] = c * w;
  WR[row + hp + f] = s * w;
}

int hole_spec_basis(int d, float* WE, float* WR, hipStream_t st) {
  const int hp = hole_spec_half(d), h = d / 2 + 1;
  const int64_t n = (int64_t)d * hp;
  hipLaunchKernelGGL(hole_basis_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, d, h, hp, WE, WR);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

// out (rows, 2hp) = X (rows, d) * W (d, 2hp), all row-major.  Column-major view: out^T (2hp x rows) = W^T (2hp x d) * X^T (d x rows).
int hole_spec_forward(const float* X, int64_t rows, int d, const float* W, float* out, hipStream_t st) {
  if (rows == 0) return KGE_OK;
  rocblas_handle h = blas_handle_shared();
  if (!h) return fail(KGE_EHIP, "kge_hole_spec_forward: rocblas handle could not be created");
  const int w2 = 2 * hole_spec_half(d);
  const float one = 1.f, zero = 0.f;
  KGE_BLAS_TRY(rocblas_set_stream(h, st));
  // m = 2hp is a few hundred, n = rows tens of thousands: as ONE GEMM rocBLAS tiles this NN product 256 x 256 and fills ~57 of 256
  // CUs (78 us at FB15k-237, against 25 us for the transposed product of the adjoint).  The rows are independent, so the table is cut
  // into 16 equal row blocks of one strided-batched GEMM (W shared: stride 0) plus a plain GEMM for the < 16 remaining rows.
  const int64_t per = rows / 16;
  const int64_t done = rows >= 4096 ? per * 16 : 0;
  if (done)
    KGE_BLAS_TRY(rocblas_sgemm_strided_batched(h, rocblas_operation_none, rocblas_operation_none, w2, (rocblas_int)per, d, &one, W, w2, 0, X, d,
                                               per * d, &zero, out, w2, per * w2, 16));
  if (done < rows)
    KGE_BLAS_TRY(rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_none, w2, (rocblas_int)(rows - done), d, &one, W, w2, X + done * d, d,
                               &zero, out + done * w2, w2));
  return KGE_OK;
}

// G (rows, d) += Ghat (rows, 2hp) * W^T.  Column-major view: G^T (d x rows) += W (d x 2hp) * Ghat^T (2hp x rows); the row-major
// memory of W is the column-major matrix W^T (2hp x d), hence the transpose flag.
int hole_spec_adjoint(float* Ghat, int64_t rows, int d, const float* W, float* G, int zero_hat, hipStream_t st) {
  if (rows == 0) return KGE_OK;
  rocblas_handle h = blas_handle_shared();
  if (!h) return fail(KGE_EHIP, "kge_hole_spec_adjoint: rocblas handle could not be created");
  const int w2 = 2 * hole_spec_half(d);
  const float one = 1.f;
  KGE_BLAS_TRY(rocblas_set_stream(h, st));
  KGE_BLAS_TRY(rocblas_sgemm(h, rocblas_operation_transpose, rocblas_operation_none, d, (rocblas_int)rows, w2, &one, W, w2, Ghat, w2, &one, G, d));
  if (zero_hat) KGE_HIP_TRY(hipMemsetAsync(Ghat, 0, sizeof(float) * (size_t)rows * w2, st));
  return KGE_OK;
}

}  // namespace kge
