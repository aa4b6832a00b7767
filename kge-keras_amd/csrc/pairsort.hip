// pairwize_hinge with the harness's all-pairs broadcast (KGEkeras/utils.py:300-305 called with negative_samples = 1, tests.py:81:
// `false` reshaped to (m_f, 1) against `true` (m_t,) -> an (m_f, m_t) matrix, mean over all of it) WITHOUT forming the pairs:
//
//     L        = 1/(m_f m_t) sum_i sum_j relu(margin + f_j - t_i)
//     dL/dt_i  = -#{j : f_j > t_i - margin} / (m_f m_t)            ([TF] relu'(0) = 0: strict)
//     dL/df_j  = +#{i : t_i - margin < f_j} / (m_f m_t)
//     L        = 1/(m_f m_t) sum_i ( S_i - c_i a_i ),   a_i = t_i - margin,  c_i = #{j : f_j > a_i},  S_i = sum of those f_j
//
// so with both score vectors sorted (hipCUB radix sort) and a running sum of the sorted `false` scores in double precision, every
// row needs one binary search: O((m_f + m_t) log) instead of O(m_f m_t).  The all-pairs kernels of train.hip take 160 us at the
// reference script's own batch (m = 20 480: 4.2e8 pairs) and 17 ms at the BASELINE batch (m = 204 800: 4.2e10 pairs -- a tensor the
// reference cannot even allocate); this path takes ~0.1 ms at either.  pairwize_logistic / pairwize_square are not sums of a
// function of the order alone and stay on the all-pairs kernels.
// NaN scores (rows with ids outside the tables: the caller's error flag is set) take no part: zero gradient, no loss term.
#include <hipcub/hipcub.hpp>

#include "kge_internal.hpp"

namespace kge {

namespace {

constexpr float kInf = __builtin_inff();
constexpr int kPsMaxBlocks = 4096;  // loss partials a launch may write (train.hip: kMaxBlocks slots)

__global__ __launch_bounds__(256) void ps_prep_kernel(const float* __restrict__ ts, int64_t m_t, const float* __restrict__ fs, int64_t m_f, float margin,
                                                       float* __restrict__ a_in, float* __restrict__ f_in, double* __restrict__ fpre) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i == 0) fpre[0] = 0.0;
  if (i < m_t) {
    const float t = ts[i];
    a_in[i] = t == t ? t - margin : kInf;  // (a NaN row is never below a `false` score)
  } else if (i < m_t + m_f) {
    const float f = fs[i - m_t];
    f_in[i - m_t] = f == f ? f : -kInf;   // (a NaN row is never above a threshold; ps_val drops it from the sums)
  }
}

struct PsVal {  // sorted `false` score -> its term of the running sum
  __host__ __device__ __forceinline__ double operator()(const float& f) const { return f == -kInf ? 0.0 : (double)f; }
};

// first index k with v[k] > x (UPPER) / with v[k] >= x (!UPPER) in the ascending array v[0..n)
template <bool UPPER>
__device__ __forceinline__ int64_t ps_bound(const float* __restrict__ v, int64_t n, float x) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    const bool right = UPPER ? v[mid] <= x : v[mid] < x;
    if (right) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void ps_search_kernel(const float* __restrict__ ts, int64_t m_t, const float* __restrict__ fs, int64_t m_f, float margin,
                                                         const float* __restrict__ a_s, const float* __restrict__ f_s, const double* __restrict__ fpre,
                                                         float* __restrict__ g_t, float* __restrict__ g_f, float* __restrict__ loss_partials) {
  __shared__ float red[16];
  const float inv = 1.f / ((float)m_f * (float)m_t);
  float lsum = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m_t + m_f; i += (int64_t)gridDim.x * 256) {
    if (i < m_t) {
      const float t = ts[i];
      float g = 0.f;
      if (t == t) {
        const float a = t - margin;
        const int64_t k = ps_bound<true>(f_s, m_f, a);  // f_s[k..) > a
        const int64_t c = m_f - k;
        g = -(float)c * inv;
        lsum += (float)((fpre[m_f] - fpre[k] - (double)c * (double)a) * (double)inv);
      }
      g_t[i] = g;
    } else {
      const float f = fs[i - m_t];
      g_f[i - m_t] = f == f ? (float)ps_bound<false>(a_s, m_t, f) * inv : 0.f;  // #{a_i < f}
    }
  }
  const float b = block_sum(lsum, red);
  if (threadIdx.x == 0) loss_partials[blockIdx.x] = b;
}

struct PsLayout { float *a_in, *f_in, *a_s, *f_s; double* fpre; void* temp; size_t temp_bytes; int64_t floats; };

PsLayout ps_layout(float* scratch, int64_t m_t, int64_t m_f, size_t temp_bytes) {
  PsLayout L;
  const int64_t at = align_up(m_t, 64), af = align_up(m_f, 64);
  float* p = scratch;
  L.a_in = p; p += at;
  L.f_in = p; p += af;
  L.a_s = p; p += at;
  L.f_s = p; p += af;
  L.fpre = reinterpret_cast<double*>(p); p += 2 * align_up(m_f + 1, 32);
  L.temp = p;
  L.temp_bytes = temp_bytes;
  L.floats = (p - scratch) + (int64_t)((temp_bytes + 3) / 4) + 64;
  return L;
}

size_t ps_temp_bytes(int64_t m_t, int64_t m_f, hipStream_t st) {
  size_t b1 = 0, b2 = 0, b3 = 0;
  const float* kin = nullptr;
  float* kout = nullptr;
  (void)hipcub::DeviceRadixSort::SortKeys(nullptr, b1, kin, kout, (int)m_t, 0, 32, st);
  (void)hipcub::DeviceRadixSort::SortKeys(nullptr, b2, kin, kout, (int)m_f, 0, 32, st);
  hipcub::TransformInputIterator<double, PsVal, const float*> it(kin, PsVal());
  double* dout = nullptr;
  (void)hipcub::DeviceScan::InclusiveSum(nullptr, b3, it, dout, (int)m_f, st);
  return std::max(b1, std::max(b2, b3)) + 256;
}

// pairwize_square_loss (utils.py:314-319), all pairs: -(f_j - t_i)^2 separates -- four sums decide everything:
//   L = -(m_t sum f^2 - 2 sum f sum t + m_f sum t^2) / (m_f m_t),  dL/dt_i = 2 (sum f - m_f t_i) / (m_f m_t),  dL/df_j = -2 (m_t f_j - sum t) / (m_f m_t)
__global__ __launch_bounds__(256) void ps_square_sums_kernel(const float* __restrict__ ts, int64_t m_t, const float* __restrict__ fs, int64_t m_f,
                                                              double* __restrict__ part /* [grid][4] */) {
  __shared__ double red[4][256];
  double st = 0.0, st2 = 0.0, sf = 0.0, sf2 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m_t + m_f; i += (int64_t)gridDim.x * 256) {
    if (i < m_t) { const double t = ts[i]; st += t; st2 += t * t; }
    else { const double f = fs[i - m_t]; sf += f; sf2 += f * f; }
  }
  red[0][threadIdx.x] = st; red[1][threadIdx.x] = st2; red[2][threadIdx.x] = sf; red[3][threadIdx.x] = sf2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o)
#pragma unroll
      for (int q = 0; q < 4; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 4) part[(int64_t)blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ __launch_bounds__(256) void ps_square_apply_kernel(const float* __restrict__ ts, int64_t m_t, const float* __restrict__ fs, int64_t m_f,
                                                               const double* __restrict__ part, int nparts, float* __restrict__ g_t, float* __restrict__ g_f,
                                                               float* __restrict__ loss_partials) {
  double s[4] = {0.0, 0.0, 0.0, 0.0};  // every thread adds the partials in the same order
  for (int b = 0; b < nparts; ++b)
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] += part[(int64_t)b * 4 + q];
  const double inv = 1.0 / ((double)m_f * (double)m_t);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m_t + m_f; i += (int64_t)gridDim.x * 256) {
    if (i < m_t) g_t[i] = (float)(2.0 * (s[2] - (double)m_f * (double)ts[i]) * inv);
    else g_f[i - m_t] = (float)(-2.0 * ((double)m_t * (double)fs[i - m_t] - s[0]) * inv);
  }
  if (threadIdx.x == 0) loss_partials[blockIdx.x] = blockIdx.x == 0 ? (float)(-((double)m_t * s[3] - 2.0 * s[2] * s[0] + (double)m_f * s[1]) * inv) : 0.f;
}

}  // namespace

// floats of scratch that cover any batch of n = m_t + m_f rows (the temporary storage of the library calls is checked at run time)
// (12 n: the sort path's keys, sorted keys and running sums; 20 n: the 16 + 16 sum planes, padded factors and workgroup sums of the one-pass
// pairwize_logistic, pairlog.hip)
int64_t pair_sort_floats(int64_t n) { return 20 * align_up(n, 64) + 65536; }

// KGE_OK; 1 when the scratch does not cover this call (the caller keeps the all-pairs kernels).  *nb = loss partials written.
int pairwise_hinge_sorted(const float* ts, int64_t m_t, const float* fs, int64_t m_f, float margin, float* g_t, float* g_f, float* loss_partials,
                          int* nb, float* scratch, int64_t scratch_floats, hipStream_t st) {
  if (!scratch || m_t + m_f >= 0x7fffffffll) return 1;
  const size_t tb = ps_temp_bytes(m_t, m_f, st);
  const PsLayout L = ps_layout(reinterpret_cast<float*>(((uintptr_t)scratch + 255) & ~(uintptr_t)255), m_t, m_f, tb);
  if (L.floats + 64 > scratch_floats) return 1;
  const int64_t n = m_t + m_f;
  hipLaunchKernelGGL(ps_prep_kernel, dim3((int)cdiv(n, 256)), dim3(256), 0, st, ts, m_t, fs, m_f, margin, L.a_in, L.f_in, L.fpre);
  size_t bytes = L.temp_bytes;
  KGE_HIP_TRY(hipcub::DeviceRadixSort::SortKeys(L.temp, bytes, (const float*)L.a_in, L.a_s, (int)m_t, 0, 32, st));
  bytes = L.temp_bytes;
  KGE_HIP_TRY(hipcub::DeviceRadixSort::SortKeys(L.temp, bytes, (const float*)L.f_in, L.f_s, (int)m_f, 0, 32, st));
  bytes = L.temp_bytes;
  hipcub::TransformInputIterator<double, PsVal, const float*> it(L.f_s, PsVal());
  KGE_HIP_TRY(hipcub::DeviceScan::InclusiveSum(L.temp, bytes, it, L.fpre + 1, (int)m_f, st));
  const int grid = (int)std::min<int64_t>(cdiv(n, 256), kPsMaxBlocks);
  *nb = grid;
  hipLaunchKernelGGL(ps_search_kernel, dim3(grid), dim3(256), 0, st, ts, m_t, fs, m_f, margin, L.a_s, L.f_s, L.fpre, g_t, g_f, loss_partials);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

// pairwize_square_loss with the all-pairs broadcast from four sums (see ps_square_sums_kernel).  KGE_OK; 1: no scratch.
int pairwise_square_sums(const float* ts, int64_t m_t, const float* fs, int64_t m_f, float* g_t, float* g_f, float* loss_partials, int* nb,
                         float* scratch, int64_t scratch_floats, hipStream_t st) {
  constexpr int kParts = 64;
  if (!scratch || scratch_floats < 2 * 4 * kParts + 64) return 1;
  double* part = reinterpret_cast<double*>(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
  const int64_t n = m_t + m_f;
  const int parts = (int)std::min<int64_t>(cdiv(n, 256), kParts);
  hipLaunchKernelGGL(ps_square_sums_kernel, dim3(parts), dim3(256), 0, st, ts, m_t, fs, m_f, part);
  const int grid = (int)std::min<int64_t>(cdiv(n, 256), kPsMaxBlocks);
  *nb = grid;
  hipLaunchKernelGGL(ps_square_apply_kernel, dim3(grid), dim3(256), 0, st, ts, m_t, fs, m_f, part, parts, g_t, g_f, loss_partials);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

}  // namespace kge
