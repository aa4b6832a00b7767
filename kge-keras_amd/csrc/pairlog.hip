// pairwize_logistic with the harness's all-pairs broadcast (KGEkeras/utils.py:307-312 called with negative_samples = 1, tests.py:81:
// `false` reshaped to (m_f, 1) against `true` (m_t,) -> an (m_f, m_t) matrix, mean over all of it) in ONE tiled pass:
//
//     L        = 1/(m_f m_t) sum_j sum_i log(EPS + 1 + e^(f_j - t_i))
//     dL/df_j  = +1/(m_f m_t) sum_i r_ji,    dL/dt_i = -1/(m_f m_t) sum_j r_ji,    r_ji = e^(f_j - t_i) / (EPS + 1 + e^(f_j - t_i))
//
// Unlike the hinge (a function of the ORDER of the scores: pairsort.hip) or the square (four sums), this is no function of order or
// moments: every pair has to be evaluated.  What it does not have to be is evaluated TWICE with three library transcendentals each --
// round 3's all-pairs kernels ran one sweep per side (expf, logf / a division per pair and side: 0.97 ms at the reference script's
// own batch m = 20 480, ~100 ms extrapolated at the BASELINE batch m = 204 800).  Here
//   * e^(f_j - t_i) = a_j b_i with a_j = e^(f_j), b_i = e^(-t_i) taken ONCE per score (pl_prep_kernel): per pair one multiply, one
//     add, v_rcp_f32, one multiply, v_log_f32 (log2; scaled by ln 2 once at the end) -- two quarter-rate instructions instead of ~40;
//   * every pair is evaluated ONCE: a thread's 8 x 8 register block adds r_ji to its row sums AND its column sums (+ the log term).
// Scores beyond +-40 (e^80 would approach FLT_MAX in a product; no embedding model of the reference gets there) take the exact
// per-pair form e^(f_j - t_i) in the same kernel (a uniform branch on a device flag: no host round trip).
//
// Layout: grid = n_jb x n_ib workgroups of 256 threads (16 x 16); workgroup (jb, ib) owns rows [jb JBW, (jb + 1) JBW) x columns
// [ib IBW, (ib + 1) IBW) and walks them in 128 x 128 steps, rows outer.  Row sums stay in registers over the whole column range and
// leave as one plane entry rowplane[ib][j]; column sums are reduced per step over the 16 row-threads (wave shuffles, then the four
// waves through LDS, summed in wave order) into an LDS accumulator of the workgroup's IBW columns, which leaves as colplane[jb][i].
// A closing kernel adds the planes in plane order.  No atomics anywhere: run-to-run bit-identical.
// Padding: a and b are padded with zeros to the workgroup ranges, so a padded pair has x = 0 exactly: r = 0 (nothing added to any sum)
// and log2(EPS + 1), which the closing kernel takes back pad_pairs times.
#include "kge_internal.hpp"

namespace kge {

namespace {

constexpr int kPlTile = 128;            // pairs tile of a step: 16 x 16 threads x (8 x 8)
constexpr int kPlMaxCols = 14336;       // columns of a workgroup (its LDS accumulator: 56 KB, + 4 KB of step buffers: inside the 64 KB a plain launch may ask for)
constexpr float kPlC1 = 1.f + 1e-6f;    // EPS + 1 (utils.py:12, :309)
constexpr float kPlBig = 40.f;

// a_j = e^(f_j) | f_j (exact form), b_i = e^(-t_i) | t_i; padding 0 | -inf / +inf (x = 0 either way).  flag: a score beyond +-kPlBig.
__global__ __launch_bounds__(256) void pl_flag_kernel(const float* __restrict__ ts, int64_t m_t, const float* __restrict__ fs, int64_t m_f, uint32_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= m_t + m_f) return;
  const float v = i < m_t ? ts[i] : fs[i - m_t];
  if (fabsf(v) > kPlBig) atomicOr(flag, 1u);  // (NaN compares false: it poisons the sums by itself, as in the reference)
}
__global__ __launch_bounds__(256) void pl_prep_kernel(const float* __restrict__ ts, int64_t m_t, int64_t mt_pad, const float* __restrict__ fs, int64_t m_f,
                                                       int64_t mf_pad, const uint32_t* __restrict__ flag, float* __restrict__ a, float* __restrict__ b) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool exact = flag[0] != 0u;
  if (i < mf_pad) a[i] = i < m_f ? (exact ? fs[i] : expf(fs[i])) : (exact ? -__builtin_inff() : 0.f);
  else if (i < mf_pad + mt_pad) {
    const int64_t k = i - mf_pad;
    b[k] = k < m_t ? (exact ? ts[k] : expf(-ts[k])) : (exact ? __builtin_inff() : 0.f);
  }
}

template <bool EXACT>
__device__ __forceinline__ void pl_block(const float (&av)[8], const float (&bv)[8], float (&rs)[8], float (&cs)[8], float& lg) {
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float x = EXACT ? expf(av[r] - bv[c]) : av[r] * bv[c];
      const float den = kPlC1 + x;
      const float q = x * __builtin_amdgcn_rcpf(den);
      rs[r] += q;
      cs[c] += q;
      lg += __builtin_amdgcn_logf(den);  // v_log_f32: log2
    }
}

template <bool EXACT>
__device__ __forceinline__ void pl_body(const float* __restrict__ a, const float* __restrict__ b, int JBW, int IBW, int jb, int ib, int n_jb, int64_t mf_pad,
                                        int64_t mt_pad, float* __restrict__ rowplane, float* __restrict__ colplane, double* __restrict__ lpart,
                                        float* colacc, float (*wsum)[4][kPlTile], double* lred) {
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15, wave = tid >> 6;
  const int64_t j_lo = (int64_t)jb * JBW, i_lo = (int64_t)ib * IBW;
  for (int k = tid; k < IBW; k += 256) colacc[k] = 0.f;
  __syncthreads();
  double lsum = 0.0;
  int step = 0;
  for (int j0 = 0; j0 < JBW; j0 += kPlTile) {
    float av[8], rs[8];
    {
      const float4 a0 = *reinterpret_cast<const float4*>(a + j_lo + j0 + ty * 8), a1 = *reinterpret_cast<const float4*>(a + j_lo + j0 + ty * 8 + 4);
      av[0] = a0.x; av[1] = a0.y; av[2] = a0.z; av[3] = a0.w; av[4] = a1.x; av[5] = a1.y; av[6] = a1.z; av[7] = a1.w;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) rs[r] = 0.f;
    for (int i0 = 0; i0 < IBW; i0 += kPlTile, ++step) {
      float bv[8], cs[8], rstep[8];
      const float4 b0 = *reinterpret_cast<const float4*>(b + i_lo + i0 + tx * 8), b1 = *reinterpret_cast<const float4*>(b + i_lo + i0 + tx * 8 + 4);
      bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
#pragma unroll
      for (int c = 0; c < 8; ++c) { cs[c] = 0.f; rstep[c] = 0.f; }
      float lg = 0.f;
      pl_block<EXACT>(av, bv, rstep, cs, lg);
#pragma unroll
      for (int r = 0; r < 8; ++r) rs[r] += rstep[r];   // (step sums of 8 terms into the long sums: blocked summation)
      lsum += (double)lg;
      // column sums of the step: over the 4 row-threads of this wave (lanes 16, 32 apart), then the 4 waves through LDS in wave order
#pragma unroll
      for (int c = 0; c < 8; ++c) { cs[c] += __shfl_xor(cs[c], 16, 64); cs[c] += __shfl_xor(cs[c], 32, 64); }
      float (*ws)[kPlTile] = wsum[step & 1];
      if ((ty & 3) == 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c) ws[wave][tx * 8 + c] = cs[c];
      }
      __syncthreads();
      if (tid < kPlTile) colacc[i0 + tid] += (ws[0][tid] + ws[1][tid]) + (ws[2][tid] + ws[3][tid]);
    }
    // row sums of this 128-row block over the workgroup's columns: over the 16 column-threads (lanes 1, 2, 4, 8 apart)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      float v = rs[r];
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
      rs[r] = v;
    }
    if (tx == 0) {
      float* dst = rowplane + (int64_t)ib * mf_pad + j_lo + j0 + ty * 8;
      *reinterpret_cast<float4*>(dst) = make_float4(rs[0], rs[1], rs[2], rs[3]);
      *reinterpret_cast<float4*>(dst + 4) = make_float4(rs[4], rs[5], rs[6], rs[7]);
    }
  }
  __syncthreads();
  for (int k = tid; k < IBW; k += 256) colplane[(int64_t)jb * mt_pad + i_lo + k] = colacc[k];
  // the workgroup's log2 sum (fixed order: lanes, then waves)
  for (int o = 32; o; o >>= 1) lsum += __shfl_xor(lsum, o, 64);
  if ((tid & 63) == 0) lred[wave] = lsum;
  __syncthreads();
  if (tid == 0) lpart[(int64_t)ib * n_jb + jb] = (lred[0] + lred[1]) + (lred[2] + lred[3]);
}

__global__ __launch_bounds__(256) void pl_pairs_kernel(const float* __restrict__ a, const float* __restrict__ b, int JBW, int IBW, int n_jb, int64_t mf_pad,
                                                        int64_t mt_pad, const uint32_t* __restrict__ flag, float* __restrict__ rowplane,
                                                        float* __restrict__ colplane, double* __restrict__ lpart) {
  extern __shared__ float colacc[];                    // [IBW]
  __shared__ float wsum[2][4][kPlTile];
  __shared__ double lred[4];
  const int jb = blockIdx.x % n_jb, ib = blockIdx.x / n_jb;
  if (flag[0] != 0u) pl_body<true>(a, b, JBW, IBW, jb, ib, n_jb, mf_pad, mt_pad, rowplane, colplane, lpart, colacc, wsum, lred);
  else pl_body<false>(a, b, JBW, IBW, jb, ib, n_jb, mf_pad, mt_pad, rowplane, colplane, lpart, colacc, wsum, lred);
}

// planes -> gradients (plane order), workgroup log sums -> the loss (one partial)
__global__ __launch_bounds__(256) void pl_finish_kernel(const float* __restrict__ rowplane, int n_ib, int64_t mf_pad, int64_t m_f, const float* __restrict__ colplane,
                                                         int n_jb, int64_t mt_pad, int64_t m_t, const double* __restrict__ lpart, int n_wg, double pad_pairs,
                                                         float* __restrict__ g_t, float* __restrict__ g_f, float* __restrict__ loss_partials) {
  const float inv = 1.f / ((float)m_f * (float)m_t);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m_t + m_f; i += (int64_t)gridDim.x * 256) {
    if (i < m_f) {
      float s = 0.f;
      for (int k = 0; k < n_ib; ++k) s += rowplane[(int64_t)k * mf_pad + i];
      g_f[i] = s * inv;
    } else {
      const int64_t c = i - m_f;
      float s = 0.f;
      for (int k = 0; k < n_jb; ++k) s += colplane[(int64_t)k * mt_pad + c];
      g_t[c] = -s * inv;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double s = 0.0;
    for (int k = 0; k < n_wg; ++k) s += lpart[k];
    s -= pad_pairs * (double)__builtin_amdgcn_logf(kPlC1);   // the padded pairs' log2(EPS + 1), by the instruction that added them
    loss_partials[0] = (float)(s * 0.69314718055994530942 / ((double)m_f * (double)m_t));
  }
}

}  // namespace

// KGE_OK; 1 when this shape is not taken (too few pairs to fill the chip, a column range past the LDS accumulator, scratch too small):
// the caller keeps the all-pairs kernels.  *nb = loss partials written (1).
int pairwise_logistic_onepass(const float* ts, int64_t m_t, const float* fs, int64_t m_f, float* g_t, float* g_f, float* loss_partials, int* nb,
                              float* scratch, int64_t scratch_floats, hipStream_t st) {
  if (!scratch || m_t < 1024 || m_f < 1024) return 1;
  // workgroups: n_jb x n_ib with 128-aligned ranges; ~2 per compute unit where the planes fit the scratch
  auto ranges = [&](int n_jb, int n_ib, int& JBW, int& IBW) {
    JBW = (int)align_up(cdiv(m_f, n_jb), kPlTile);
    IBW = (int)align_up(cdiv(m_t, n_ib), kPlTile);
  };
  int n_jb = 32, n_ib = 16, JBW = 0, IBW = 0;
  auto need = [&](int nj, int ni) {
    int J, I;
    JBW = J = (int)align_up(cdiv(m_f, nj), kPlTile);
    IBW = I = (int)align_up(cdiv(m_t, ni), kPlTile);
    const int64_t mfp = (int64_t)J * nj, mtp = (int64_t)I * ni;
    return (int64_t)ni * mfp + (int64_t)nj * mtp + mfp + mtp + 2 * (int64_t)nj * ni + 256;
  };
  if (need(n_jb, n_ib) > scratch_floats) { n_jb = 16; n_ib = 16; }
  while (IBW > kPlMaxCols && n_ib < 64) { n_ib *= 2; (void)need(n_jb, n_ib); }
  if (need(n_jb, n_ib) > scratch_floats || IBW > kPlMaxCols) return 1;
  ranges(n_jb, n_ib, JBW, IBW);
  const int64_t mf_pad = (int64_t)JBW * n_jb, mt_pad = (int64_t)IBW * n_ib;
  float* p = reinterpret_cast<float*>(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
  uint32_t* flag = reinterpret_cast<uint32_t*>(p); p += 64;
  double* lpart = reinterpret_cast<double*>(p); p += 2 * (int64_t)n_jb * n_ib;
  float* a = p; p += mf_pad;
  float* b = p; p += mt_pad;
  float* rowplane = p; p += (int64_t)n_ib * mf_pad;
  float* colplane = p;
  KGE_HIP_TRY(hipMemsetAsync(flag, 0, 4, st));
  hipLaunchKernelGGL(pl_flag_kernel, dim3((int)cdiv(m_t + m_f, 256)), dim3(256), 0, st, ts, m_t, fs, m_f, flag);
  hipLaunchKernelGGL(pl_prep_kernel, dim3((int)cdiv(mf_pad + mt_pad, 256)), dim3(256), 0, st, ts, m_t, mt_pad, fs, m_f, mf_pad, flag, a, b);
  hipLaunchKernelGGL(pl_pairs_kernel, dim3(n_jb * n_ib), dim3(256), (size_t)IBW * sizeof(float), st, a, b, JBW, IBW, n_jb, mf_pad, mt_pad, flag, rowplane,
                     colplane, lpart);
  const double pad_pairs = (double)mf_pad * (double)mt_pad - (double)m_f * (double)m_t;
  const int grid = (int)std::min<int64_t>(cdiv(m_t + m_f, 256), 2048);
  hipLaunchKernelGGL(pl_finish_kernel, dim3(grid), dim3(256), 0, st, rowplane, n_ib, mf_pad, m_f, colplane, n_jb, mt_pad, m_t, lpart, n_jb * n_ib, pad_pairs, g_t,
                     g_f, loss_partials);
  *nb = 1;
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

}  // namespace kge
