// ConvE (KGEkeras/models.py:239-275):  x = Drop(ReLU(BN2(Dense(Flatten(Drop(ReLU(BN1(Conv3x3x8([s ; p] as a 2w x h image)))))))),
// score = <x, o>.  Unlike the other models this is a small CNN whose layers are shared by all triples: its conv / batch-norm /
// dense parameters live in one flat fp32 vector `theta` (so the optimizer is the same dense sweep as for the tables),
// batch statistics are reductions over the whole call, and the dense layer is a plain GEMM (rocBLAS sgemm, fp32).
// Everything around the GEMMs is hand-written HIP; the per-triple rows dS, dP, dO that come out of the backward pass are
// summed into the embedding tables' gradients by the same per-destination lists as the other models (pull.hip).
//
// theta = [K(KH,KW,C) | bc(C) | g1(C) | b1(C) | Wd(F,d) | bd(d) | g2(d) | b2(d)],   stats = [mm1(C) | mv1(C) | mm2(d) | mv2(d)]
// [TF semantics restated in oracle/kge_oracle.py conve_*]: BN eps 1e-3, momentum 0.99, the 4-D BN's moving variance takes
// Bessel's correction (fused kernel), Flatten order (H, W, C); the two calls of the wrapper (pos, neg) are separate BN batches.
// The biases bc and bd feed straight into a batch norm, so their gradients are identically zero; they are left untouched.
#include <rocblas/rocblas.h>

#include "dropout.hpp"
#include "kge_internal.hpp"
#include "pull.hpp"

namespace kge {

constexpr float kBnEps = 1e-3f, kBnMom = 0.99f;

struct CvShape { int d, w, h, H, W, Ho, Wo, C, F, KH, KW; uint32_t mWo, mW; };  // mWo, mW = ceil(2^32 / Wo), ceil(2^32 / W) (cv_div)
// Conv2D(conv_filters, (conv_size_w, conv_size_h)) of models.py:254: a (KH, KW) kernel over the (2w rows, h columns) image, 'valid',
// stride 1.  kge_model_desc.conv_kh / conv_kw == 0 stand for the reference default 3.
static CvShape cv_shape(const kge_model_desc* md) {
  const int d = md->d_e;
  int w = 1;
  for (int i = 1; i * i <= d; ++i) if (d % i == 0) w = i;  // largest i <= sqrt(d) dividing d (models.py:250-251)
  CvShape s;
  s.KH = md->conv_kh > 0 ? md->conv_kh : 3;
  s.KW = md->conv_kw > 0 ? md->conv_kw : 3;
  s.C = md->conv_filters;
  s.d = d; s.w = w; s.h = d / w; s.H = 2 * w; s.W = s.h; s.Ho = s.H - s.KH + 1; s.Wo = s.W - s.KW + 1; s.F = s.Ho * s.Wo * s.C;
  s.mWo = s.Wo > 1 ? (uint32_t)((0x100000000ull + s.Wo - 1) / s.Wo) : 0u;
  s.mW = s.W > 1 ? (uint32_t)((0x100000000ull + s.W - 1) / s.W) : 0u;
  return s;
}
// the register-tiled kernels below are written for the reference defaults (8 filters of 3x3); every other shape takes the cvg_* kernels
static bool cv_fast(const CvShape& s) {
  static const bool force_generic = [] { const char* v = getenv("KGE_CONVE_GENERIC"); return v && atoi(v) != 0; }();
  return s.C == 8 && s.KH == 3 && s.KW == 3 && !force_generic;
}
struct CvOff { int64_t K, bc, g1, b1, Wd, bd, g2, b2, total; };
static CvOff cv_off(const CvShape& s) {
  CvOff o;
  o.K = 0; o.bc = (int64_t)s.KH * s.KW * s.C; o.g1 = o.bc + s.C; o.b1 = o.g1 + s.C; o.Wd = o.b1 + s.C; o.bd = o.Wd + (int64_t)s.F * s.d;
  o.g2 = o.bd + s.d; o.b2 = o.g2 + s.d; o.total = o.b2 + s.d;
  return o;
}

// x / dv for 0 <= x < 2^16, 1 <= dv < 2^16 with magic = ceil(2^32 / dv) (0 for dv == 1): the positions of an image, per lane and triple
__device__ __forceinline__ int cv_div(int x, uint32_t magic) { return magic ? (int)__umulhi((uint32_t)x, magic) : x; }

// ------------------------------------------------------------------------------------------------ forward kernels
// One wave per triple: the 2w x h image of (s ; p) is staged in LDS, every lane computes the C channels of positions lane, lane+64, ...
// The default shape (8 filters of 3x3) never stores the convolution's output: 2 592 floats per triple would be written once and read three
// times (batch-norm apply, its backward sums, the convolution's backward) -- 2.1 GB per array of the bench shape each time -- while
// recomputing them from the 400-float image in LDS costs 72 FMAs per position (cv_conv_at: one FMA order everywhere, so the recomputed
// values are bit-identical wherever they are formed).  Other shapes: the cvg_* kernels further down, which store it.
template <int C> struct CvKern { float k[9][C]; float b[C]; };
template <int C>
__device__ __forceinline__ void cv_kern_load(const float* __restrict__ theta, CvKern<C>& K) {
#pragma unroll
  for (int a = 0; a < 9; ++a)
#pragma unroll
    for (int c = 0; c < C; ++c) K.k[a][c] = theta[a * C + c];
#pragma unroll
  for (int c = 0; c < C; ++c) K.b[c] = theta[9 * C + c];
}
// the 3x3 window at (r, q) and the C channel values of that position
template <int C>
__device__ __forceinline__ void cv_conv_at(const float* img, int W, int r, int q, const CvKern<C>& K, float (&xv)[9], float (&acc)[C]) {
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) xv[a * 3 + b] = img[(r + a) * W + q + b];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = K.b[c];
#pragma unroll
  for (int a = 0; a < 9; ++a)
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = fmaf(xv[a], K.k[a][c], acc[c]);
}
// image of (s ; p) of triple i into this wave's LDS slice; false (and the error flag, if given) for ids outside the tables
template <bool DROP>
__device__ __forceinline__ bool cv_stage(float* img, const float* __restrict__ E, const float* __restrict__ Rel, const int32_t* __restrict__ tri,
                                         int64_t i, int N, int R, int d, uint32_t* err, uint64_t seed, int64_t row_offset, uint32_t thr, float dscale,
                                         int lane) {
  const int s = tri[3 * i], p = tri[3 * i + 1], o = tri[3 * i + 2];
  const bool ok = (unsigned)s < (unsigned)N && (unsigned)p < (unsigned)R && (unsigned)o < (unsigned)N;
  if (err && !ok && lane == 0) atomicOr(err, 1u);
  const float* es = E + (int64_t)(ok ? s : 0) * d;
  const float* rp = Rel + (int64_t)(ok ? p : 0) * d;
  DropKey ks{0, 0}, kp{0, 0};
  if (DROP) { ks = drop_key(seed, row_offset + i, 0); kp = drop_key(seed, row_offset + i, 1); }
  for (int k = lane; k < d; k += 64) {
    float a = es[k], b = rp[k];
    if (DROP) { a *= drop_mul(ks, k, thr, dscale); b *= drop_mul(kp, k, thr, dscale); }
    img[k] = ok ? a : 0.f;       // rows 0..w-1   = s reshaped (w, h)
    img[d + k] = ok ? b : 0.f;   // rows w..2w-1  = p reshaped (w, h)
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  return ok;
}

// pass 1 of a training call: per-block partial sums of y and y^2 per channel, nothing else written
template <bool DROP, int C>
__global__ __launch_bounds__(256) void cv_conv_stats(const float* __restrict__ E, const float* __restrict__ Rel, const int32_t* __restrict__ tri,
                                                      int64_t m, int N, int R, CvShape sh, const float* __restrict__ theta,
                                                      float* __restrict__ psum, uint32_t* __restrict__ err, uint64_t seed, int64_t row_offset,
                                                      uint32_t thr, float dscale) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float red[4][2 * C];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* img = lds + wave * (sh.H * sh.W);
  CvKern<C> K;
  cv_kern_load<C>(theta, K);
  float s1[C], s2[C];
#pragma unroll
  for (int c = 0; c < C; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
  for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < m; i += (int64_t)gridDim.x * 4) {
    cv_stage<DROP>(img, E, Rel, tri, i, N, R, sh.d, err, seed, row_offset, thr, dscale, lane);
    for (int pos = lane; pos < sh.Ho * sh.Wo; pos += 64) {
      const int r = cv_div(pos, sh.mWo), q = pos - r * sh.Wo;
      float xv[9], acc[C];
      cv_conv_at<C>(img, sh.W, r, q, K, xv, acc);
#pragma unroll
      for (int c = 0; c < C; ++c) { s1[c] += acc[c]; s2[c] = fmaf(acc[c], acc[c], s2[c]); }
    }
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int c = 0; c < C; ++c) { s1[c] = group_sum<64>(s1[c]); s2[c] = group_sum<64>(s2[c]); }
  if (lane == 0)
#pragma unroll
    for (int c = 0; c < C; ++c) { red[wave][c] = s1[c]; red[wave][C + c] = s2[c]; }
  __syncthreads();
  if (threadIdx.x < 2 * C) psum[(int64_t)blockIdx.x * 2 * C + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// pass 2 (the only pass in inference mode): A = Drop(ReLU(g1 * (y - mu) * inv + b1)) with y recomputed.  A wave's 64 positions x 8
// channels are turned in LDS so that every store instruction covers 1 KB in a row (lane-per-position stores leave half-filled lines
// behind each instruction: 1.08 ms per 204 800-row call against 0.46 ms).
template <bool DROPE, bool DROPH, int C>
__global__ __launch_bounds__(256) void cv_conv_act(const float* __restrict__ E, const float* __restrict__ Rel, const int32_t* __restrict__ tri,
                                                    int64_t m, int N, int R, CvShape sh, const float* __restrict__ theta,
                                                    const float* __restrict__ theta_g1, const float* __restrict__ theta_b1,
                                                    const float* __restrict__ stat, float* __restrict__ A, uint32_t* __restrict__ err,
                                                    uint64_t seed, int64_t row_offset, uint32_t thr_e, float ds_e, uint32_t thr_h, float ds_h) {
  static_assert(C == 8, "two float4 per position");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ __attribute__((aligned(16))) float4 turn[4][128];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* img = lds + wave * (sh.H * sh.W);
  CvKern<C> K;
  cv_kern_load<C>(theta, K);
  float sc[C], mu[C], be[C];
#pragma unroll
  for (int c = 0; c < C; ++c) { sc[c] = theta_g1[c] * stat[C + c]; mu[c] = stat[c]; be[c] = theta_b1[c]; }
  const int npos = sh.Ho * sh.Wo;
  for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < m; i += (int64_t)gridDim.x * 4) {
    cv_stage<DROPE>(img, E, Rel, tri, i, N, R, sh.d, err, seed, row_offset, thr_e, ds_e, lane);
    DropKey kh{0, 0};
    if (DROPH) kh = drop_key(seed, row_offset + i, 3);
    float4* arow4 = reinterpret_cast<float4*>(A + i * (int64_t)sh.F);
    for (int p0 = 0; p0 < npos; p0 += 64) {
      const int pos = p0 + lane;
      if (pos < npos) {
        const int r = cv_div(pos, sh.mWo), q = pos - r * sh.Wo;
        float xv[9], acc[C];
        cv_conv_at<C>(img, sh.W, r, q, K, xv, acc);
#pragma unroll
        for (int c = 0; c < C; ++c) {
          float a = fmaf(sc[c], acc[c] - mu[c], be[c]);
          a = a > 0.f ? a : 0.f;
          if (DROPH) a *= drop_mul(kh, (uint32_t)(pos * C + c), thr_h, ds_h);
          acc[c] = a;
        }
        turn[wave][2 * lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        turn[wave][2 * lane + 1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int n4 = 2 * min(64, npos - p0);  // float4 of this round
      if (lane < n4) arow4[2 * p0 + lane] = turn[wave][lane];
      if (lane + 64 < n4) arow4[2 * p0 + 64 + lane] = turn[wave][64 + lane];
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// Block-wide double sums of two strided columns of the partials matrix: one workgroup per channel, fixed tree order.
__device__ __forceinline__ void cv_col_sums(const float* __restrict__ partials, int nb, int stride, int c1, int c2, double& a, double& b) {
  __shared__ double sa[256], sb[256];
  double x = 0.0, y = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) { x += partials[(int64_t)i * stride + c1]; y += partials[(int64_t)i * stride + c2]; }
  sa[threadIdx.x] = x; sb[threadIdx.x] = y;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { sa[threadIdx.x] += sa[threadIdx.x + o]; sb[threadIdx.x] += sb[threadIdx.x + o]; }
    __syncthreads();
  }
  a = sa[0]; b = sb[0];
}

// partials[nb][2*nch] (sum, sum of squares) over `count` values per channel -> callstat[0..nch) = mean, [nch..2nch) = 1/sqrt(var+eps);
// training: moving statistics updated (bessel: the fused 4-D batch norm).  grid = nch workgroups.
__global__ __launch_bounds__(256) void cv_bn_finalize(const float* __restrict__ partials, int nb, int nch, double count, int bessel,
                                                       float* __restrict__ callstat, float* __restrict__ mov_mean,
                                                       float* __restrict__ mov_var, int use_moving) {
  const int c = blockIdx.x;
  float mean, var;
  if (use_moving) { mean = mov_mean[c]; var = mov_var[c]; }
  else {
    double a, b;
    cv_col_sums(partials, nb, 2 * nch, c, nch + c, a, b);
    const double mu = a / count;
    double v = b / count - mu * mu;
    if (v < 0.0) v = 0.0;
    mean = (float)mu; var = (float)v;
    if (threadIdx.x == 0) {
      mov_mean[c] = mov_mean[c] * kBnMom + mean * (1.f - kBnMom);
      const float vu = bessel ? (float)(v * (count / (count > 1.0 ? count - 1.0 : 1.0))) : var;
      mov_var[c] = mov_var[c] * kBnMom + vu * (1.f - kBnMom);
    }
  }
  if (threadIdx.x == 0) {
    callstat[c] = mean;
    callstat[nch + c] = 1.f / sqrtf(var + kBnEps);
  }
}

// Z += bd (per column) and, TRAIN, per-block partial column sums of z and z^2.  Block = 256 threads over columns, 64 rows.
template <bool TRAIN>
__global__ __launch_bounds__(256) void cv_z_stats(float* __restrict__ Z, int64_t m, int d, const float* __restrict__ bd,
                                                   float* __restrict__ psum) {
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  for (int k = threadIdx.x; k < d; k += 256) {
    const float b = bd[k];
    float s1 = 0.f, s2 = 0.f;
    for (int64_t r = r0; r < min(m, r0 + 64); ++r) {
      const float z = Z[r * d + k] + b;
      Z[r * d + k] = z;
      s1 += z; s2 = fmaf(z, z, s2);
    }
    if (TRAIN) { psum[(int64_t)blockIdx.x * 2 * d + k] = s1; psum[(int64_t)blockIdx.x * 2 * d + d + k] = s2; }
  }
}

// x = Drop(ReLU(g2 * (Z - mu2) * inv2 + b2));  X[i] = x;  score_i = <x, o'>.  One wave per triple.
template <bool DROP>
__global__ __launch_bounds__(256) void cv_head_out(const float* __restrict__ E, const int32_t* __restrict__ tri, int64_t m, int N,
                                                    int d, const float* __restrict__ Z, const float* __restrict__ g2,
                                                    const float* __restrict__ b2, const float* __restrict__ stat,
                                                    float* __restrict__ X, float* __restrict__ scores, uint64_t seed,
                                                    int64_t row_offset, uint32_t thr_e, float dscale_e, uint32_t thr_h, float dscale_h,
                                                    int want_score) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < m; i += (int64_t)gridDim.x * 4) {
    const int o = tri[3 * i + 2];
    const bool ok = (unsigned)tri[3 * i] < (unsigned)N && (unsigned)o < (unsigned)N;
    const float* eo = E + (int64_t)(ok ? o : 0) * d;
    DropKey ko{0, 0}, kh{0, 0};
    if (DROP) { ko = drop_key(seed, row_offset + i, 2); kh = drop_key(seed, row_offset + i, 4); }
    float part = 0.f;
    for (int k = lane; k < d; k += 64) {
      float x = fmaf(g2[k] * stat[d + k], Z[i * d + k] - stat[k], b2[k]);
      x = x > 0.f ? x : 0.f;
      if (DROP) x *= drop_mul(kh, k, thr_h, dscale_h);
      X[i * d + k] = x;
      if (want_score) {
        float ov = eo[k];
        if (DROP) ov *= drop_mul(ko, k, thr_e, dscale_e);
        part = fmaf(x, ov, part);
      }
    }
    if (want_score) {
      part = group_sum<64>(part);
      if (lane == 0) scores[i] = ok ? part : __builtin_nanf("");
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward kernels
// dO[t] = mask_o * g * x ; dy2 = g * o' * (x > 0 ? hscale : 0) written over X ; per-block partial column sums of dy2 and dy2*xhat2.
template <bool DROP>
__global__ __launch_bounds__(256) void cv_head_bwd(const float* __restrict__ E, const int32_t* __restrict__ tri, int64_t m, int N,
                                                    int d, const float* __restrict__ Z, const float* __restrict__ stat,
                                                    const float* __restrict__ g, float* __restrict__ X, float* __restrict__ dO,
                                                    float* __restrict__ psum, uint64_t seed, int64_t row_offset, uint32_t thr_e,
                                                    float dscale_e, float dscale_h) {
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  for (int k = threadIdx.x; k < d; k += 256) {
    const float mu = stat[k], inv = stat[d + k];
    float s1 = 0.f, s2 = 0.f;
    for (int64_t i = r0; i < min(m, r0 + 64); ++i) {
      const int o = tri[3 * i + 2];
      const bool ok = (unsigned)tri[3 * i] < (unsigned)N && (unsigned)o < (unsigned)N;
      const float gi = ok ? g[i] : 0.f;
      float ov = E[(int64_t)(ok ? o : 0) * d + k], mo = 1.f;
      if (DROP) { mo = drop_mul(drop_key(seed, row_offset + i, 2), k, thr_e, dscale_e); ov *= mo; }
      const float x = X[i * d + k];
      dO[i * d + k] = gi * x * mo;
      const float dy = x > 0.f ? gi * ov * (DROP ? dscale_h : 1.f) : 0.f;
      X[i * d + k] = dy;
      const float xh = (Z[i * d + k] - mu) * inv;
      s1 += dy; s2 = fmaf(dy, xh, s2);
    }
    psum[(int64_t)blockIdx.x * 2 * d + k] = s1;
    psum[(int64_t)blockIdx.x * 2 * d + d + k] = s2;
  }
}

// partials[nb][2*nch] -> bst[0..nch) = mean(gamma*dy), bst[nch..2nch) = mean(gamma*dy*xhat); dgamma += sum(dy*xhat), dbeta += sum(dy).
// grid = nch workgroups.
__global__ __launch_bounds__(256) void cv_bn_bwd_finalize(const float* __restrict__ partials, int nb, int nch, double count,
                                                           const float* __restrict__ gamma, float* __restrict__ bst,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x;
  double a, b;
  cv_col_sums(partials, nb, 2 * nch, c, nch + c, a, b);
  if (threadIdx.x == 0) {
    dbeta[c] += (float)a;
    dgamma[c] += (float)b;
    bst[c] = (float)(a / count) * gamma[c];
    bst[nch + c] = (float)(b / count) * gamma[c];
  }
}

// dZ = inv2 * (g2*dy2 - mean(g2 dy2) - xhat2 * mean(g2 dy2 xhat2)) written over Z (dy2 is in X)
__global__ __launch_bounds__(256) void cv_bn2_bwd_apply(float* __restrict__ Z, const float* __restrict__ X, int64_t m, int d,
                                                         const float* __restrict__ g2, const float* __restrict__ stat,
                                                         const float* __restrict__ bst) {
  const int64_t total = m * (int64_t)d;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i % d);
    const float inv = stat[d + k];
    const float xh = (Z[i] - stat[k]) * inv;
    Z[i] = inv * (g2[k] * X[i] - bst[k] - xh * bst[d + k]);
  }
}

// ---- backward of the default shape with y recomputed (see cv_conv_stats): dy1 = dA * (A > 0 ? hscale : 0) is formed on the fly in both
// kernels -- A > 0 <=> the pre-activation is positive and the element was kept by the hidden dropout -- and never written.
// per-block partial channel sums of dy1 and dy1 * xhat1, reading dA alone
template <bool DROPE, bool DROPH, int C>
__global__ __launch_bounds__(256) void cv_bn1_bwd_rc(const float* __restrict__ E, const float* __restrict__ Rel, const int32_t* __restrict__ tri,
                                                      int64_t m, int N, int R, CvShape sh, const float* __restrict__ theta,
                                                      const float* __restrict__ theta_g1, const float* __restrict__ theta_b1,
                                                      const float* __restrict__ stat, const float* __restrict__ dA, float* __restrict__ psum,
                                                      uint64_t seed, int64_t row_offset, uint32_t thr_e, float ds_e, uint32_t thr_h, float ds_h) {
  static_assert(C == 8, "two float4 per position");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float red[4][2 * C];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* img = lds + wave * (sh.H * sh.W);
  CvKern<C> K;
  cv_kern_load<C>(theta, K);
  float sc[C], mu[C], be[C], inv[C];
#pragma unroll
  for (int c = 0; c < C; ++c) { inv[c] = stat[C + c]; sc[c] = theta_g1[c] * inv[c]; mu[c] = stat[c]; be[c] = theta_b1[c]; }
  float s1[C], s2[C];
#pragma unroll
  for (int c = 0; c < C; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
  const float hscale = DROPH ? ds_h : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < m; i += (int64_t)gridDim.x * 4) {
    cv_stage<DROPE>(img, E, Rel, tri, i, N, R, sh.d, nullptr, seed, row_offset, thr_e, ds_e, lane);
    DropKey kh{0, 0};
    if (DROPH) kh = drop_key(seed, row_offset + i, 3);
    const float* da = dA + i * (int64_t)sh.F;
    for (int pos = lane; pos < sh.Ho * sh.Wo; pos += 64) {
      // (lane-per-position loads: turning the row in LDS as cv_conv_act does for its stores measured slower here, 0.80 against 0.67 ms)
      const float4 d0 = reinterpret_cast<const float4*>(da + pos * C)[0], d1 = reinterpret_cast<const float4*>(da + pos * C)[1];
      const float dv[C] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
      const int r = cv_div(pos, sh.mWo), q = pos - r * sh.Wo;
      float xv[9], acc[C];
      cv_conv_at<C>(img, sh.W, r, q, K, xv, acc);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        bool keep = fmaf(sc[c], acc[c] - mu[c], be[c]) > 0.f;
        if (DROPH) keep = keep && drop_mul(kh, (uint32_t)(pos * C + c), thr_h, ds_h) != 0.f;
        const float dy = keep ? dv[c] * hscale : 0.f;
        s1[c] += dy;
        s2[c] = fmaf(dy, (acc[c] - mu[c]) * inv[c], s2[c]);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int c = 0; c < C; ++c) { s1[c] = group_sum<64>(s1[c]); s2[c] = group_sum<64>(s2[c]); }
  if (lane == 0)
#pragma unroll
    for (int c = 0; c < C; ++c) { red[wave][c] = s1[c]; red[wave][C + c] = s2[c]; }
  __syncthreads();
  if (threadIdx.x < 2 * C) psum[(int64_t)blockIdx.x * 2 * C + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// One wave per triple: dY = inv1 * (g1*dy1 - mean - xhat1*mean2) from dA and the recomputed y, kept in LDS; dK partial sums in the same
// pass over the positions (the window is in registers already); dX gathered per pixel from up to nine positions -> dS, dP rows.
// (Tried and dropped: the position's nine contributions added into a 2w x h image with ds_add_f32 instead of the dY slice -- LDS float
// atomics serialise their lanes: 3.54 ms per 204 800-row call against 2.46 ms; the dA row staged through LDS in whole-line loads: 2.64 ms.)
template <bool DROPE, bool DROPH, int C>
__global__ __launch_bounds__(256) void cv_conv_bwd_rc(const float* __restrict__ E, const float* __restrict__ Rel, const int32_t* __restrict__ tri,
                                                       int64_t m, int N, int R, CvShape sh, const float* __restrict__ theta,
                                                       const float* __restrict__ theta_g1, const float* __restrict__ theta_b1,
                                                       const float* __restrict__ stat, const float* __restrict__ bst, const float* __restrict__ dA,
                                                       float* __restrict__ dS, float* __restrict__ dP, float* __restrict__ pk, uint64_t seed,
                                                       int64_t row_offset, uint32_t thr_e, float ds_e, uint32_t thr_h, float ds_h) {
  static_assert(C == 8, "two float4 per position");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float red[4][9 * C];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int npos = sh.Ho * sh.Wo;
  float* img = lds + wave * (sh.H * sh.W + npos * C);
  float* dy = img + sh.H * sh.W;  // [npos][C]
  CvKern<C> K;
  cv_kern_load<C>(theta, K);
  float sc[C], mu[C], be[C], inv[C], g1[C], m1[C], m2[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    inv[c] = stat[C + c]; g1[c] = theta_g1[c]; sc[c] = g1[c] * inv[c]; mu[c] = stat[c]; be[c] = theta_b1[c]; m1[c] = bst[c]; m2[c] = bst[C + c];
  }
  float dk[9][C];
#pragma unroll
  for (int a = 0; a < 9; ++a)
#pragma unroll
    for (int c = 0; c < C; ++c) dk[a][c] = 0.f;
  const float hscale = DROPH ? ds_h : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < m; i += (int64_t)gridDim.x * 4) {
    const bool ok = cv_stage<DROPE>(img, E, Rel, tri, i, N, R, sh.d, nullptr, seed, row_offset, thr_e, ds_e, lane);
    DropKey ks{0, 0}, kp{0, 0}, kh{0, 0};
    if (DROPE) { ks = drop_key(seed, row_offset + i, 0); kp = drop_key(seed, row_offset + i, 1); }
    if (DROPH) kh = drop_key(seed, row_offset + i, 3);
    const float* da = dA + i * (int64_t)sh.F;
    for (int pos = lane; pos < npos; pos += 64) {
      const float4 d0 = reinterpret_cast<const float4*>(da + pos * C)[0], d1 = reinterpret_cast<const float4*>(da + pos * C)[1];
      const float dv0[C] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
      const int r = cv_div(pos, sh.mWo), q = pos - r * sh.Wo;
      float xv[9], acc[C], dv[C];
      cv_conv_at<C>(img, sh.W, r, q, K, xv, acc);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        bool keep = fmaf(sc[c], acc[c] - mu[c], be[c]) > 0.f;
        if (DROPH) keep = keep && drop_mul(kh, (uint32_t)(pos * C + c), thr_h, ds_h) != 0.f;
        const float dy1 = keep ? dv0[c] * hscale : 0.f;
        const float xh = (acc[c] - mu[c]) * inv[c];
        dv[c] = ok ? inv[c] * (g1[c] * dy1 - m1[c] - xh * m2[c]) : 0.f;
      }
      reinterpret_cast<float4*>(dy + pos * C)[0] = make_float4(dv[0], dv[1], dv[2], dv[3]);
      reinterpret_cast<float4*>(dy + pos * C)[1] = make_float4(dv[4], dv[5], dv[6], dv[7]);
      // dK[a][b][c] += X[r+a][q+b] * dY[pos][c]
#pragma unroll
      for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int c = 0; c < C; ++c) dk[a][c] = fmaf(xv[a], dv[c], dk[a][c]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // dX[u][v] = sum_{a,b,c} dY[u-a][v-b][c] * K[a][b][c]  (positions inside the valid output only)
    for (int k = lane; k < 2 * sh.d; k += 64) {
      const int u = cv_div(k, sh.mW), v = k - u * sh.W;
      float acc = 0.f;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int r = u - a;
        if (r < 0 || r >= sh.Ho) continue;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const int q = v - b;
          if (q < 0 || q >= sh.Wo) continue;
          const float4 y0 = reinterpret_cast<const float4*>(dy + (r * sh.Wo + q) * C)[0], y1 = reinterpret_cast<const float4*>(dy + (r * sh.Wo + q) * C)[1];
          const float yv[C] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
#pragma unroll
          for (int c = 0; c < C; ++c) acc = fmaf(yv[c], K.k[a * 3 + b][c], acc);
        }
      }
      if (k < sh.d) {
        if (DROPE) acc *= drop_mul(ks, k, thr_e, ds_e);
        dS[i * sh.d + k] = acc;
      } else {
        const int kk2 = k - sh.d;
        if (DROPE) acc *= drop_mul(kp, kk2, thr_e, ds_e);
        dP[i * sh.d + kk2] = acc;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int a = 0; a < 9; ++a)
#pragma unroll
    for (int c = 0; c < C; ++c) dk[a][c] = group_sum<64>(dk[a][c]);
  if (lane == 0)
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
      for (int c = 0; c < C; ++c) red[wave][a * C + c] = dk[a][c];
  __syncthreads();
  if (threadIdx.x < 9 * C) pk[(int64_t)blockIdx.x * 9 * C + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void cv_sum_partials(const float* __restrict__ partials, int nb, int n, float* __restrict__ out) {
  const int c = blockIdx.x;  // grid = n workgroups
  double a, b;
  cv_col_sums(partials, nb, n, c, c, a, b);
  if (threadIdx.x == 0) out[c] += (float)a;
}

// gradient of the embedding tables: sum of the per-triple rows over the destination lists (entries of pull.hip's fill_kernel:
// .w = t, entity keys carry the role in bit 31 of .x).  One wave per (destination row, chunk).
__global__ __launch_bounds__(256) void cv_rowsum_kernel(float* __restrict__ gE, float* __restrict__ gR, int N, int NK, int d,
                                                         const int32_t* __restrict__ cnt, const int32_t* __restrict__ off,
                                                         const int32_t* __restrict__ eoff, const int32_t* __restrict__ poff,
                                                         const int32_t* __restrict__ totals, const int2* __restrict__ emap,
                                                         const int4* __restrict__ entries, float* __restrict__ slab, int Cc,
                                                         const float* __restrict__ dS, const float* __restrict__ dP,
                                                         const float* __restrict__ dO) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int unit = blockIdx.x * 4 + wave;
  int key, chunk;
  if (unit < NK) { key = unit; chunk = 0; }
  else {
    const int x = unit - NK;
    if (x >= totals[0]) return;
    const int2 kc = emap[x];
    key = kc.x; chunk = kc.y;
  }
  const int c = cnt[key];
  if (c == 0) return;
  const int nch = (c + Cc - 1) / Cc;
  const int start = off[key] + chunk * Cc, len = min(Cc, c - chunk * Cc);
  const bool is_rel = key >= N;
  float* dst = nch == 1 ? (is_rel ? gR + (int64_t)(key - N) * d : gE + (int64_t)key * d) : slab + (int64_t)(poff[key] + chunk) * d;
  for (int k = lane; k < d; k += 64) {
    float acc = 0.f;
    for (int i = 0; i < len; ++i) {
      const int4 e = entries[start + i];
      const float* src = is_rel ? dP : (((unsigned)e.x >> 31) ? dO : dS);
      acc += src[(int64_t)e.w * d + k];
    }
    dst[k] = nch == 1 ? dst[k] + acc : acc;
  }
}

// Same sums for rows of d % 4 == 0, d <= 64 * NV floats: four 16-lane sub-groups of the wave take different list entries,
// a lane reads its NV float4 chunks of an entry's row, the sub-groups meet in a shuffle reduction (one pass over the list
// with four entries in flight instead of ceil(d / 64) passes with one).
template <int NV>
__global__ __launch_bounds__(256) void cv_rowsum4_kernel(float* __restrict__ gE, float* __restrict__ gR, int N, int NK, int d,
                                                          const int32_t* __restrict__ cnt, const int32_t* __restrict__ off,
                                                          const int32_t* __restrict__ eoff, const int32_t* __restrict__ poff,
                                                          const int32_t* __restrict__ totals, const int2* __restrict__ emap,
                                                          const int4* __restrict__ entries, float* __restrict__ slab, int Cc,
                                                          const float* __restrict__ dS, const float* __restrict__ dP,
                                                          const float* __restrict__ dO) {
  const int wave = threadIdx.x >> 6, lane64 = threadIdx.x & 63, lane = lane64 & 15, sub = lane64 >> 4;
  const int unit = blockIdx.x * 4 + wave;
  int key, chunk;
  if (unit < NK) { key = unit; chunk = 0; }
  else {
    const int x = unit - NK;
    if (x >= totals[0]) return;
    const int2 kc = emap[x];
    key = kc.x; chunk = kc.y;
  }
  const int c = cnt[key];
  if (c == 0) return;
  const int nch = (c + Cc - 1) / Cc, nq = d >> 2;
  const int start = off[key] + chunk * Cc, len = min(Cc, c - chunk * Cc);
  const bool is_rel = key >= N;
  float4 acc[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 2
  for (int i = sub; i < len; i += 4) {
    const int4 e = entries[start + i];
    const float* src = (is_rel ? dP : (((unsigned)e.x >> 31) ? dO : dS)) + (int64_t)e.w * d;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int q = lane + 16 * j;
      if (q < nq) {
        const float4 v = *reinterpret_cast<const float4*>(src + 4 * q);
        acc[j].x += v.x; acc[j].y += v.y; acc[j].z += v.z; acc[j].w += v.w;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
      acc[j].x += __shfl_xor(acc[j].x, o, 64); acc[j].y += __shfl_xor(acc[j].y, o, 64);
      acc[j].z += __shfl_xor(acc[j].z, o, 64); acc[j].w += __shfl_xor(acc[j].w, o, 64);
    }
  }
  if (sub != 0) return;
  float* dst = nch == 1 ? (is_rel ? gR + (int64_t)(key - N) * d : gE + (int64_t)key * d) : slab + (int64_t)(poff[key] + chunk) * d;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int q = lane + 16 * j;
    if (q < nq) {
      float4* p4 = reinterpret_cast<float4*>(dst + 4 * q);
      if (nch == 1) { const float4 o4 = *p4; *p4 = make_float4(o4.x + acc[j].x, o4.y + acc[j].y, o4.z + acc[j].z, o4.w + acc[j].w); }
      else *p4 = acc[j];
    }
  }
}

// ------------------------------------------------------------------------------------------------ any conv shape
// The same layers for any conv_filters / conv_size_* (models.py:241-245): one thread per output element, weights read through the
// caches, runtime loop bounds; the accumulation order of the convolution is that of the register-tiled kernels (bias, then a, b
// ascending), so the default shape gives the same Y either way.  No speed claims: the hot configurations take the kernels above.
template <bool DROP>
__global__ __launch_bounds__(256) void cvg_conv_fwd(const float* __restrict__ E, const float* __restrict__ Rel, const int32_t* __restrict__ tri,
                                                     int64_t m, int N, int R, CvShape sh, const float* __restrict__ theta, float* __restrict__ Y,
                                                     uint32_t* __restrict__ err, uint64_t seed, int64_t row_offset, uint32_t thr, float dscale) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* img = lds + wave * (sh.H * sh.W);
  const int C = sh.C, KK = sh.KH * sh.KW;
  for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < m; i += (int64_t)gridDim.x * 4) {
    const int s = tri[3 * i], p = tri[3 * i + 1], o = tri[3 * i + 2];
    const bool ok = (unsigned)s < (unsigned)N && (unsigned)p < (unsigned)R && (unsigned)o < (unsigned)N;
    if (!ok && lane == 0) atomicOr(err, 1u);
    const float* es = E + (int64_t)(ok ? s : 0) * sh.d;
    const float* rp = Rel + (int64_t)(ok ? p : 0) * sh.d;
    DropKey ks{0, 0}, kp{0, 0};
    if (DROP) { ks = drop_key(seed, row_offset + i, 0); kp = drop_key(seed, row_offset + i, 1); }
    for (int k = lane; k < sh.d; k += 64) {
      float a = es[k], b = rp[k];
      if (DROP) { a *= drop_mul(ks, k, thr, dscale); b *= drop_mul(kp, k, thr, dscale); }
      img[k] = ok ? a : 0.f;
      img[sh.d + k] = ok ? b : 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float* yrow = Y + i * (int64_t)sh.F;
    for (int e = lane; e < sh.F; e += 64) {
      const int pos = e / C, c = e - pos * C;
      const int r = pos / sh.Wo, q = pos - r * sh.Wo;
      float acc = theta[KK * C + c];
      for (int a = 0; a < sh.KH; ++a)
        for (int b = 0; b < sh.KW; ++b) acc = fmaf(img[(r + a) * sh.W + q + b], theta[(a * sh.KW + b) * C + c], acc);
      yrow[e] = acc;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// Per-block partial channel sums over a slab of 16 rows of an (m, F) activation whose channel is index % C, laid out like the
// partial sums of the kernels above (psum[block][2C]).  MODE 0: sums of y and y^2 (batch-norm statistics).  MODE 1 (backward):
// dy1 = A > 0 ? dA * hscale : 0 written over V (= dA), sums of dy1 and dy1 * xhat1.  Thread t owns channel t % C and every
// (256 / C)-th position; the phases meet in LDS in fixed order.
template <int MODE>
__global__ __launch_bounds__(256) void cvg_chan_sums(float* __restrict__ V, const float* __restrict__ A, const float* __restrict__ Y, int64_t m,
                                                      int F, int C, const float* __restrict__ stat, float hscale, float* __restrict__ psum) {
  __shared__ float red[2][256];
  const int64_t r0 = (int64_t)blockIdx.x * 16, r1 = min(m, r0 + 16);
  const int npos = F / C;
  for (int c0 = 0; c0 < C; c0 += 256) {  // (C > 256: channel groups of 256, one phase)
    const int Cg = min(C - c0, 256);
    const int phases = 256 / Cg;
    const int c = c0 + (int)threadIdx.x % Cg, ph = (int)threadIdx.x / Cg;
    float s1 = 0.f, s2 = 0.f;
    if (ph < phases) {
      for (int64_t r = r0; r < r1; ++r)
        for (int pos = ph; pos < npos; pos += phases) {
          const int64_t idx = r * F + (int64_t)pos * C + c;
          if (MODE == 0) {
            const float y = V[idx];
            s1 += y;
            s2 = fmaf(y, y, s2);
          } else {
            const float dy = A[idx] > 0.f ? V[idx] * hscale : 0.f;
            V[idx] = dy;
            s1 += dy;
            s2 = fmaf(dy, (Y[idx] - stat[c]) * stat[C + c], s2);
          }
        }
    }
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    if ((int)threadIdx.x < Cg) {
      float a = 0.f, b = 0.f;
      for (int q = 0; q < phases; ++q) { a += red[0][q * Cg + threadIdx.x]; b += red[1][q * Cg + threadIdx.x]; }
      psum[(int64_t)blockIdx.x * 2 * C + c0 + threadIdx.x] = a;
      psum[(int64_t)blockIdx.x * 2 * C + C + c0 + threadIdx.x] = b;
    }
    __syncthreads();
  }
}

// A = Drop(ReLU(g1 * (Y - mu) * inv + b1)), one element per thread (channel = index % C).
template <bool DROP>
__global__ __launch_bounds__(256) void cvg_bn1_apply(const float* __restrict__ Y, float* __restrict__ A, int64_t m, int F, int C,
                                                      const float* __restrict__ theta_g1, const float* __restrict__ theta_b1,
                                                      const float* __restrict__ stat, uint64_t seed, int64_t row_offset, uint32_t thr, float dscale) {
  const int64_t total = m * (int64_t)F;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c = (int)(idx % C);
    const int64_t row = idx / F;
    const int e = (int)(idx - row * F);
    float a = fmaf(theta_g1[c] * stat[C + c], Y[idx] - stat[c], theta_b1[c]);
    a = a > 0.f ? a : 0.f;
    if (DROP) a *= drop_mul(drop_key(seed, row_offset + row, 3), (uint32_t)e, thr, dscale);
    A[idx] = a;
  }
}

// One wave per triple: dY on the fly (dy1 in dA), dX -> dS, dP rows, and the kernel gradient: weight j = (a*KW + b)*C + c is owned by
// lane j % 64 of the wave, summed over the positions in ascending order into the wave's LDS slice, the four slices meet at the end.
template <bool DROP>
__global__ __launch_bounds__(256) void cvg_conv_bwd(const float* __restrict__ E, const float* __restrict__ Rel, const int32_t* __restrict__ tri,
                                                     int64_t m, int N, int R, CvShape sh, const float* __restrict__ theta,
                                                     const float* __restrict__ stat, const float* __restrict__ bst, const float* __restrict__ dA,
                                                     const float* __restrict__ Y, float* __restrict__ dS, float* __restrict__ dP,
                                                     float* __restrict__ pk, uint64_t seed, int64_t row_offset, uint32_t thr, float dscale) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int C = sh.C, KK = sh.KH * sh.KW, npos = sh.Ho * sh.Wo, nk = KK * C;
  const int per = sh.H * sh.W + sh.F + nk;
  float* img = lds + wave * per;
  float* dy = img + sh.H * sh.W;  // [npos][C]
  float* dk = dy + sh.F;          // [KK][C]
  const float* g1 = theta + (int64_t)nk + C;
  for (int j = lane; j < nk; j += 64) dk[j] = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < m; i += (int64_t)gridDim.x * 4) {
    const int s = tri[3 * i], p = tri[3 * i + 1], o = tri[3 * i + 2];
    const bool ok = (unsigned)s < (unsigned)N && (unsigned)p < (unsigned)R && (unsigned)o < (unsigned)N;
    const float* es = E + (int64_t)(ok ? s : 0) * sh.d;
    const float* rp = Rel + (int64_t)(ok ? p : 0) * sh.d;
    DropKey ks{0, 0}, kp{0, 0};
    if (DROP) { ks = drop_key(seed, row_offset + i, 0); kp = drop_key(seed, row_offset + i, 1); }
    for (int k = lane; k < sh.d; k += 64) {
      float a = es[k], b = rp[k];
      if (DROP) { a *= drop_mul(ks, k, thr, dscale); b *= drop_mul(kp, k, thr, dscale); }
      img[k] = ok ? a : 0.f;
      img[sh.d + k] = ok ? b : 0.f;
    }
    const float* da = dA + i * (int64_t)sh.F;
    const float* yr = Y + i * (int64_t)sh.F;
    for (int e = lane; e < sh.F; e += 64) {
      const int c = e % C;
      const float inv = stat[C + c];
      const float xh = (yr[e] - stat[c]) * inv;
      dy[e] = ok ? inv * (g1[c] * da[e] - bst[c] - xh * bst[C + c]) : 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < nk; j += 64) {  // dK[a][b][c] += sum_pos X[r+a][q+b] * dY[pos][c]
      const int ab = j / C, c = j - ab * C, a = ab / sh.KW, b = ab - a * sh.KW;
      float acc = 0.f;
      for (int pos = 0; pos < npos; ++pos) {
        const int r = pos / sh.Wo, q = pos - r * sh.Wo;
        acc = fmaf(img[(r + a) * sh.W + q + b], dy[pos * C + c], acc);
      }
      dk[j] += acc;
    }
    for (int k = lane; k < 2 * sh.d; k += 64) {  // dX[u][v] = sum_{a,b,c} dY[u-a][v-b][c] * K[a][b][c]
      const int u = k / sh.W, v = k - u * sh.W;
      float acc = 0.f;
      for (int a = 0; a < sh.KH; ++a) {
        const int r = u - a;
        if (r < 0 || r >= sh.Ho) continue;
        for (int b = 0; b < sh.KW; ++b) {
          const int q = v - b;
          if (q < 0 || q >= sh.Wo) continue;
          for (int c = 0; c < C; ++c) acc = fmaf(dy[(r * sh.Wo + q) * C + c], theta[(a * sh.KW + b) * C + c], acc);
        }
      }
      if (k < sh.d) {
        if (DROP) acc *= drop_mul(ks, k, thr, dscale);
        dS[i * sh.d + k] = acc;
      } else {
        const int kk2 = k - sh.d;
        if (DROP) acc *= drop_mul(kp, kk2, thr, dscale);
        dP[i * sh.d + kk2] = acc;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  for (int j = threadIdx.x; j < nk; j += 256)
    pk[(int64_t)blockIdx.x * nk + j] = ((lds[0 * per + sh.H * sh.W + sh.F + j] + lds[1 * per + sh.H * sh.W + sh.F + j]) + lds[2 * per + sh.H * sh.W + sh.F + j]) +
                                       lds[3 * per + sh.H * sh.W + sh.F + j];
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
static rocblas_handle blas_handle() {
  static rocblas_handle h = nullptr;
  static int state = 0;
  if (state == 0) state = rocblas_create_handle(&h) == rocblas_status_success ? 1 : -1;
  return state == 1 ? h : nullptr;
}

rocblas_handle blas_handle_shared() { return blas_handle(); }

#define KGE_BLAS_TRY(expr)                                                                  \
  do {                                                                                      \
    rocblas_status s__ = (expr);                                                            \
    if (s__ != rocblas_status_success) return fail(KGE_EHIP, "%s: rocblas status %d", #expr, (int)s__); \
  } while (0)

struct CvBuf {  // per call
  float *Y, *A, *Z, *X, *stat1, *stat2;
};
struct CvWs {
  CvBuf call[2];
  float *dA, *dS, *dP, *dO, *psum, *bst, *g, *sc;
  void* pull_hdr;
  void* pull_body;
  int64_t psum_floats;
};

static int64_t c256(int64_t x) { return align_up(x, 256); }
int pull_chunk();
int64_t train_workspace_bytes(const kge_model_desc*, int64_t, int64_t);
int loss_value(const kge_loss_desc*, const float*, int64_t, const float*, int64_t, float*, float*, float*, void*, int64_t, hipStream_t);
int pull2_fill_entries(const kge_model_desc* md, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, const PullPlan& p,
                       hipStream_t st);
int pull_finish_launch(const kge_model_desc* md, float* gE, float* gR, const PullPlan& p, hipStream_t st);

static int64_t cv_psum_floats(const CvShape& s, int64_t m) {
  const int64_t nb_conv = std::min<int64_t>(cdiv(m, 4), 2048);
  const int64_t nb_rows = cdiv(m, 64);
  const int64_t nb_bn1 = std::max<int64_t>(cdiv(m * s.F / 4, 4096), cdiv(m, 16));  // (register-tiled / cvg_chan_sums partials)
  return std::max<int64_t>({nb_conv * (int64_t)s.KH * s.KW * s.C, nb_rows * 2 * s.d, nb_bn1 * 2 * s.C, nb_conv * 2 * s.C}) + 64;
}

int64_t conve_workspace_bytes(const kge_model_desc* md, int64_t m_pos, int64_t m_neg) {
  const CvShape s = cv_shape(md);
  const int64_t n = m_pos + m_neg, mm = std::max(m_pos, m_neg);
  int64_t b = 0;
  const int nf = cv_fast(s) ? 1 : 2;  // A, and Y where the convolution's output is stored (the general kernels)
  for (int64_t m : {m_pos, m_neg}) b += nf * c256(m * s.F * 4) + 2 * c256(m * s.d * 4) + c256(2 * s.C * 4) + c256(2 * s.d * 4);
  b += c256(mm * s.F * 4) + 3 * c256(n * s.d * 4) + c256(cv_psum_floats(s, mm) * 4) + c256(2 * (s.d + s.C) * 4) + 2 * c256(n * 4);
  b += pull_header_bytes(md) + pull_workspace_bytes(md, n) + c256(train_workspace_bytes(nullptr, m_pos, m_neg)) + 4096;
  return b;
}

static void cv_carve(void* ws, const kge_model_desc* md, int64_t m_pos, int64_t m_neg, CvWs* w, void** loss_ws, int64_t* loss_ws_bytes) {
  const CvShape s = cv_shape(md);
  const int64_t n = m_pos + m_neg, mm = std::max(m_pos, m_neg);
  char* q = reinterpret_cast<char*>(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  auto take = [&](int64_t bytes) { char* r = q; q += c256(bytes); return r; };
  w->pull_hdr = take(pull_header_bytes(md));  // fixed place: zero on entry, left zero
  const int64_t ms[2] = {m_pos, m_neg};
  for (int c = 0; c < 2; ++c) {
    w->call[c].Y = cv_fast(s) ? nullptr : (float*)take(ms[c] * s.F * 4);
    w->call[c].A = (float*)take(ms[c] * s.F * 4);
    w->call[c].Z = (float*)take(ms[c] * s.d * 4);
    w->call[c].X = (float*)take(ms[c] * s.d * 4);
    w->call[c].stat1 = (float*)take(2 * s.C * 4);
    w->call[c].stat2 = (float*)take(2 * s.d * 4);
  }
  w->dA = (float*)take(mm * s.F * 4);
  w->dS = (float*)take(n * s.d * 4);
  w->dP = (float*)take(n * s.d * 4);
  w->dO = (float*)take(n * s.d * 4);
  w->psum_floats = cv_psum_floats(s, mm);
  w->psum = (float*)take(w->psum_floats * 4);
  w->bst = (float*)take(2 * (s.d + s.C) * 4);
  w->g = (float*)take(n * 4);
  w->sc = (float*)take(n * 4);
  w->pull_body = take(pull_workspace_bytes(md, n));
  *loss_ws_bytes = train_workspace_bytes(nullptr, m_pos, m_neg);
  *loss_ws = take(*loss_ws_bytes);
}

int conve_sizes(const kge_model_desc* md, int64_t* n_theta, int64_t* n_stats) {
  if (md->conv_filters < 1 || md->conv_kh < 0 || md->conv_kw < 0)
    return fail(KGE_EINVAL, "ConvE: conv_filters = %d, kernel %d x %d", md->conv_filters, md->conv_kh, md->conv_kw);
  const CvShape s = cv_shape(md);
  if (s.w < 2 && s.h < 2) return fail(KGE_ESHAPE, "ConvE: e_dim %d has no factor pair with w > 1 or h > 1 (models.py:252)", md->d_e);
  if (s.Ho < 1 || s.Wo < 1) return fail(KGE_ESHAPE, "ConvE: image %dx%d is smaller than the %dx%d kernel", s.H, s.W, s.KH, s.KW);
  if ((size_t)4 * (s.H * s.W + s.F + (cv_fast(s) ? 0 : s.KH * s.KW * s.C)) * sizeof(float) > 64 * 1024)
    return fail(KGE_EUNSUPPORTED, "ConvE: %d filters of %dx%d on a %dx%d image need more than 64 KB of LDS per workgroup in the backward pass", s.C,
                s.KH, s.KW, s.H, s.W);
  if (s.d > 1024) return fail(KGE_EUNSUPPORTED, "ConvE: e_dim above 1024 is not built");
  *n_theta = cv_off(s).total;
  *n_stats = 2 * s.C + 2 * s.d;
  return KGE_OK;
}

// forward of one call (rows of `tri`).  training: batch statistics + dropout; else moving statistics.
static int cv_forward(const kge_model_desc* md, const float* E, const float* Rel, const float* theta, float* stats, const int32_t* tri,
                      int64_t m, int64_t row_offset, bool training, const CvBuf& b, float* psum, float* scores, int want_score,
                      uint32_t* err, uint64_t seed, hipStream_t st) {
  const CvShape s = cv_shape(md);
  const int C = s.C;
  const bool fast = cv_fast(s);
  const CvOff o = cv_off(s);
  const bool drop_e = training && md->dropout > 0.f, drop_h = training && md->hidden_dropout > 0.f;
  const uint32_t thr_e = drop_thr(md->dropout), thr_h = drop_thr(md->hidden_dropout);
  const float ds_e = 1.f / (1.f - md->dropout), ds_h = 1.f / (1.f - md->hidden_dropout);
  const int nb_conv = (int)std::min<int64_t>(cdiv(m, 4), 2048);
  const size_t smem = (size_t)4 * s.H * s.W * sizeof(float);
  int nb_stat1 = nb_conv;
  if (fast) {  // (the convolution's output is not stored: pass 1 files its channel sums, pass 2 -- below -- recomputes it)
    if (training) {
      if (drop_e) hipLaunchKernelGGL((cv_conv_stats<true, 8>), dim3(nb_conv), dim3(256), smem, st, E, Rel, tri, m, md->num_entities, md->num_relations, s, theta, psum, err, seed, row_offset, thr_e, ds_e);
      else hipLaunchKernelGGL((cv_conv_stats<false, 8>), dim3(nb_conv), dim3(256), smem, st, E, Rel, tri, m, md->num_entities, md->num_relations, s, theta, psum, err, seed, row_offset, thr_e, ds_e);
    }
  } else {
    if (drop_e) hipLaunchKernelGGL((cvg_conv_fwd<true>), dim3(nb_conv), dim3(256), smem, st, E, Rel, tri, m, md->num_entities, md->num_relations, s, theta, b.Y, err, seed, row_offset, thr_e, ds_e);
    else hipLaunchKernelGGL((cvg_conv_fwd<false>), dim3(nb_conv), dim3(256), smem, st, E, Rel, tri, m, md->num_entities, md->num_relations, s, theta, b.Y, err, seed, row_offset, thr_e, ds_e);
    nb_stat1 = (int)cdiv(m, 16);
    if (training) hipLaunchKernelGGL((cvg_chan_sums<0>), dim3(nb_stat1), dim3(256), 0, st, b.Y, nullptr, nullptr, m, s.F, C, nullptr, 1.f, psum);
  }
  hipLaunchKernelGGL(cv_bn_finalize, dim3(C), dim3(256), 0, st, psum, nb_stat1, C, (double)m * s.Ho * s.Wo, 1, b.stat1, stats, stats + C, training ? 0 : 1);
  if (fast) {
#define KGE_ACT(DE, DH) hipLaunchKernelGGL((cv_conv_act<DE, DH, 8>), dim3(nb_conv), dim3(256), smem, st, E, Rel, tri, m, md->num_entities, md->num_relations, \
                                           s, theta, theta + o.g1, theta + o.b1, b.stat1, b.A, err, seed, row_offset, thr_e, ds_e, thr_h, ds_h)
    if (drop_e) { if (drop_h) KGE_ACT(true, true); else KGE_ACT(true, false); }
    else { if (drop_h) KGE_ACT(false, true); else KGE_ACT(false, false); }
#undef KGE_ACT
  } else {
    const int nb_el = (int)std::min<int64_t>(cdiv(m * s.F, 256), 8192);
    if (drop_h) hipLaunchKernelGGL((cvg_bn1_apply<true>), dim3(nb_el), dim3(256), 0, st, b.Y, b.A, m, s.F, C, theta + o.g1, theta + o.b1, b.stat1, seed, row_offset, thr_h, ds_h);
    else hipLaunchKernelGGL((cvg_bn1_apply<false>), dim3(nb_el), dim3(256), 0, st, b.Y, b.A, m, s.F, C, theta + o.g1, theta + o.b1, b.stat1, seed, row_offset, thr_h, ds_h);
  }
  rocblas_handle h = blas_handle();
  if (!h) return fail(KGE_EHIP, "ConvE: rocblas handle could not be created");
  KGE_BLAS_TRY(rocblas_set_stream(h, st));
  const float one = 1.f, zero = 0.f;
  // row-major Z(m,d) = A(m,F) * Wd(F,d)  ==  column-major Z'(d,m) = Wd'(d,F) * A'(F,m)
  KGE_BLAS_TRY(rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_none, s.d, (rocblas_int)m, s.F, &one, theta + o.Wd, s.d, b.A, s.F, &zero, b.Z, s.d));
  const int nb_rows = (int)cdiv(m, 64);
  if (training) hipLaunchKernelGGL((cv_z_stats<true>), dim3(nb_rows), dim3(256), 0, st, b.Z, m, s.d, theta + o.bd, psum);
  else hipLaunchKernelGGL((cv_z_stats<false>), dim3(nb_rows), dim3(256), 0, st, b.Z, m, s.d, theta + o.bd, psum);
  hipLaunchKernelGGL(cv_bn_finalize, dim3(s.d), dim3(256), 0, st, psum, nb_rows, s.d, (double)m, 0, b.stat2, stats + 2 * C, stats + 2 * C + s.d, training ? 0 : 1);
  const int nb_w = (int)std::min<int64_t>(cdiv(m, 4), 4096);
  if (drop_e || drop_h)
    hipLaunchKernelGGL((cv_head_out<true>), dim3(nb_w), dim3(256), 0, st, E, tri, m, md->num_entities, s.d, b.Z, theta + o.g2, theta + o.b2, b.stat2, b.X,
                       scores, seed, row_offset, drop_e ? thr_e : 0u, drop_e ? ds_e : 1.f, drop_h ? thr_h : 0u, drop_h ? ds_h : 1.f, want_score);
  else
    hipLaunchKernelGGL((cv_head_out<false>), dim3(nb_w), dim3(256), 0, st, E, tri, m, md->num_entities, s.d, b.Z, theta + o.g2, theta + o.b2, b.stat2, b.X,
                       scores, seed, row_offset, 0u, 1.f, 0u, 1.f, want_score);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

static int cv_backward(const kge_model_desc* md, const float* E, const float* Rel, const float* theta, float* gtheta, const int32_t* tri,
                       int64_t m, int64_t row_offset, const CvBuf& b, const CvWs& w, const float* g, uint64_t seed, hipStream_t st) {
  const CvShape s = cv_shape(md);
  const int C = s.C, nk = s.KH * s.KW * s.C;
  const bool fast = cv_fast(s);
  const CvOff o = cv_off(s);
  const bool drop_e = md->dropout > 0.f, drop_h = md->hidden_dropout > 0.f;
  const uint32_t thr_e = drop_thr(md->dropout);
  const float ds_e = 1.f / (1.f - md->dropout), ds_h = drop_h ? 1.f / (1.f - md->hidden_dropout) : 1.f;
  float* dS = w.dS + row_offset * s.d;
  float* dP = w.dP + row_offset * s.d;
  float* dO = w.dO + row_offset * s.d;
  float* bst2 = w.bst;
  float* bst1 = w.bst + 2 * s.d;
  const int nb_rows = (int)cdiv(m, 64);
  if (drop_e || drop_h)
    hipLaunchKernelGGL((cv_head_bwd<true>), dim3(nb_rows), dim3(256), 0, st, E, tri, m, md->num_entities, s.d, b.Z, b.stat2, g, b.X, dO, w.psum, seed,
                       row_offset, drop_e ? thr_e : 0u, drop_e ? ds_e : 1.f, ds_h);
  else
    hipLaunchKernelGGL((cv_head_bwd<false>), dim3(nb_rows), dim3(256), 0, st, E, tri, m, md->num_entities, s.d, b.Z, b.stat2, g, b.X, dO, w.psum, seed,
                       row_offset, 0u, 1.f, 1.f);
  hipLaunchKernelGGL(cv_bn_bwd_finalize, dim3(s.d), dim3(256), 0, st, w.psum, nb_rows, s.d, (double)m, theta + o.g2, bst2, gtheta + o.g2, gtheta + o.b2);
  const int nb_el2 = (int)std::min<int64_t>(cdiv(m * s.d, 256), 8192);
  hipLaunchKernelGGL(cv_bn2_bwd_apply, dim3(nb_el2), dim3(256), 0, st, b.Z, b.X, m, s.d, theta + o.g2, b.stat2, bst2);
  rocblas_handle h = blas_handle();
  if (!h) return fail(KGE_EHIP, "ConvE: rocblas handle could not be created");
  KGE_BLAS_TRY(rocblas_set_stream(h, st));
  const float one = 1.f, zero = 0.f;
  // dA(m,F) = dZ(m,d) * Wd^T     ==  dA'(F,m) = Wd'(d,F)^T * dZ'(d,m)
  KGE_BLAS_TRY(rocblas_sgemm(h, rocblas_operation_transpose, rocblas_operation_none, s.F, (rocblas_int)m, s.d, &one, theta + o.Wd, s.d, b.Z, s.d, &zero, w.dA, s.F));
  // dWd(F,d) += A^T(F,m) * dZ(m,d)  ==  dWd'(d,F) += dZ'(d,m) * A'(F,m)^T
  KGE_BLAS_TRY(rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_transpose, s.d, s.F, (rocblas_int)m, &one, b.Z, s.d, b.A, s.F, &one, gtheta + o.Wd, s.d));
  const int64_t total4 = m * (int64_t)s.F / 4;
  int nb_bn1 = (int)cdiv(total4, 4096);
  const int nb_conv = (int)std::min<int64_t>(cdiv(m, 4), 2048);
  const uint32_t thr_h = drop_thr(md->hidden_dropout);
  if (fast) {
    const size_t smem1 = (size_t)4 * s.H * s.W * sizeof(float);
    nb_bn1 = nb_conv;
#define KGE_BN1(DE, DH) hipLaunchKernelGGL((cv_bn1_bwd_rc<DE, DH, 8>), dim3(nb_conv), dim3(256), smem1, st, E, Rel, tri, m, md->num_entities, md->num_relations, \
                                           s, theta, theta + o.g1, theta + o.b1, b.stat1, w.dA, w.psum, seed, row_offset, thr_e, ds_e, thr_h, ds_h)
    if (drop_e) { if (drop_h) KGE_BN1(true, true); else KGE_BN1(true, false); }
    else { if (drop_h) KGE_BN1(false, true); else KGE_BN1(false, false); }
#undef KGE_BN1
  } else {
    nb_bn1 = (int)cdiv(m, 16);
    hipLaunchKernelGGL((cvg_chan_sums<1>), dim3(nb_bn1), dim3(256), 0, st, w.dA, b.A, b.Y, m, s.F, C, b.stat1, ds_h, w.psum);
  }
  hipLaunchKernelGGL(cv_bn_bwd_finalize, dim3(C), dim3(256), 0, st, w.psum, nb_bn1, C, (double)m * s.Ho * s.Wo, theta + o.g1, bst1, gtheta + o.g1, gtheta + o.b1);
  if (fast) {
    const size_t smem = (size_t)4 * (s.H * s.W + s.Ho * s.Wo * C) * sizeof(float);
#define KGE_CBW(DE, DH) hipLaunchKernelGGL((cv_conv_bwd_rc<DE, DH, 8>), dim3(nb_conv), dim3(256), smem, st, E, Rel, tri, m, md->num_entities, md->num_relations, \
                                           s, theta, theta + o.g1, theta + o.b1, b.stat1, bst1, w.dA, dS, dP, w.psum, seed, row_offset, thr_e, ds_e, thr_h, ds_h)
    if (drop_e) { if (drop_h) KGE_CBW(true, true); else KGE_CBW(true, false); }
    else { if (drop_h) KGE_CBW(false, true); else KGE_CBW(false, false); }
#undef KGE_CBW
  } else {
    const size_t smem = (size_t)4 * (s.H * s.W + s.F + nk) * sizeof(float);
    if (drop_e)
      hipLaunchKernelGGL((cvg_conv_bwd<true>), dim3(nb_conv), dim3(256), smem, st, E, Rel, tri, m, md->num_entities, md->num_relations, s, theta, b.stat1,
                         bst1, w.dA, b.Y, dS, dP, w.psum, seed, row_offset, thr_e, ds_e);
    else
      hipLaunchKernelGGL((cvg_conv_bwd<false>), dim3(nb_conv), dim3(256), smem, st, E, Rel, tri, m, md->num_entities, md->num_relations, s, theta, b.stat1,
                         bst1, w.dA, b.Y, dS, dP, w.psum, seed, row_offset, thr_e, ds_e);
  }
  hipLaunchKernelGGL(cv_sum_partials, dim3(nk), dim3(256), 0, st, w.psum, nb_conv, nk, gtheta + o.K);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

// scores (want_hidden == 0) or the hidden vectors x(s,p) (want_hidden != 0; out is (n, d)) in inference mode
int conve_infer(const kge_model_desc* md, const float* E, const float* Rel, const float* theta, float* stats, const int32_t* tri, int64_t n,
                float* out, int want_hidden, void* ws, int64_t wbytes, uint32_t* err, hipStream_t st) {
  int64_t nt, ns;
  if (int rc = conve_sizes(md, &nt, &ns)) return rc;
  const CvShape s = cv_shape(md);
  const int64_t CH = 32768;  // rows per pass: bounds the activation buffers
  const int64_t m = std::min(n, CH);
  const int64_t need = (cv_fast(s) ? 1 : 2) * c256(m * s.F * 4) + 2 * c256(m * s.d * 4) + c256(2 * s.C * 4) + c256(2 * s.d * 4) + c256(cv_psum_floats(s, m) * 4) + 1024;
  if (wbytes < need) return fail(KGE_EWORKSPACE, "ConvE inference: workspace has %lld bytes, needs %lld", (long long)wbytes, (long long)need);
  char* q = reinterpret_cast<char*>(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  auto take = [&](int64_t bytes) { char* r = q; q += c256(bytes); return r; };
  CvBuf b;
  b.Y = cv_fast(s) ? nullptr : (float*)take(m * s.F * 4); b.A = (float*)take(m * s.F * 4); b.Z = (float*)take(m * s.d * 4);
  float* Xtmp = (float*)take(m * s.d * 4);
  b.stat1 = (float*)take(2 * s.C * 4); b.stat2 = (float*)take(2 * s.d * 4);
  float* psum = (float*)take(cv_psum_floats(s, m) * 4);
  ProfScope prof(KGE_PROF_SCORE, st);
  for (int64_t a = 0; a < n; a += CH) {
    const int64_t mm = std::min(CH, n - a);
    b.X = want_hidden ? out + a * s.d : Xtmp;
    if (int rc = cv_forward(md, E, Rel, theta, stats, tri + 3 * a, mm, 0, false, b, psum, want_hidden ? nullptr : out + a, want_hidden ? 0 : 1, err, 0, st)) return rc;
  }
  return KGE_OK;
}

// list-sum of per-triple rows into the table gradients (also used by literal.hip)
static void rowsum_dispatch(int grid, hipStream_t st, float* gE, float* gR, int N, int NK, int d, const PullPlan& plan, const float* dS, const float* dP,
                            const float* dO) {
#define KGE_RS_ARGS gE, gR, N, NK, d, plan.cnt, plan.off, plan.eoff, plan.poff, plan.totals, plan.emap, plan.entries, plan.slab, plan.chunk, dS, dP, dO
  const bool al = (d & 3) == 0 && (((uintptr_t)gE | (uintptr_t)gR | (uintptr_t)plan.slab | (uintptr_t)dS | (uintptr_t)dP | (uintptr_t)dO) & 15) == 0;
  if (al && d <= 64) hipLaunchKernelGGL((cv_rowsum4_kernel<1>), dim3(grid), dim3(256), 0, st, KGE_RS_ARGS);
  else if (al && d <= 128) hipLaunchKernelGGL((cv_rowsum4_kernel<2>), dim3(grid), dim3(256), 0, st, KGE_RS_ARGS);
  else if (al && d <= 256) hipLaunchKernelGGL((cv_rowsum4_kernel<4>), dim3(grid), dim3(256), 0, st, KGE_RS_ARGS);
  else hipLaunchKernelGGL(cv_rowsum_kernel, dim3(grid), dim3(256), 0, st, KGE_RS_ARGS);
#undef KGE_RS_ARGS
}

int cv_rowsum_launch(const kge_model_desc* md, float* gE, float* gR, const PullPlan& plan, const float* dS, const float* dP, const float* dO,
                     hipStream_t st) {
  const int N = md->num_entities, NK = N + md->num_relations;
  rowsum_dispatch((int)cdiv(NK + plan.max_extra, 4), st, gE, gR, N, NK, md->d_e, plan, dS, dP, dO);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

int64_t conve_infer_workspace_bytes(const kge_model_desc* md, int64_t n) {
  const CvShape s = cv_shape(md);
  const int64_t m = std::min<int64_t>(n, 32768);
  return (cv_fast(s) ? 1 : 2) * c256(m * s.F * 4) + 2 * c256(m * s.d * 4) + c256(2 * s.C * 4) + c256(2 * s.d * 4) + c256(cv_psum_floats(s, m) * 4) + 2048;
}

// rows != null: the per-triple gradient rows are handed out ({dS, dP, dO}, (m_pos + m_neg, d) each, valid until the workspace is used
// again) instead of being summed into gE / gR -- what the literal path needs: there the "tables" are staged rows of the batch itself.
int conve_train_grads_rows(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, const float* theta, float* stats,
                           float* gE, float* gR, float* gtheta, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg,
                           float* loss_out, float* scores_out, uint64_t seed, void* ws, int64_t wbytes, uint32_t* err, hipStream_t st,
                           const float** rows);
int conve_train_grads(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, const float* theta, float* stats,
                      float* gE, float* gR, float* gtheta, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg,
                      float* loss_out, float* scores_out, uint64_t seed, void* ws, int64_t wbytes, uint32_t* err, hipStream_t st) {
  return conve_train_grads_rows(md, ld, E, Rel, theta, stats, gE, gR, gtheta, pos, m_pos, neg, m_neg, loss_out, scores_out, seed, ws, wbytes, err, st, nullptr);
}
int conve_train_grads_rows(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, const float* theta, float* stats,
                           float* gE, float* gR, float* gtheta, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg,
                           float* loss_out, float* scores_out, uint64_t seed, void* ws, int64_t wbytes, uint32_t* err, hipStream_t st,
                           const float** rows) {
  int64_t nt, ns;
  if (int rc = conve_sizes(md, &nt, &ns)) return rc;
  if (wbytes < conve_workspace_bytes(md, m_pos, m_neg))
    return fail(KGE_EWORKSPACE, "ConvE: workspace has %lld bytes, needs %lld", (long long)wbytes, (long long)conve_workspace_bytes(md, m_pos, m_neg));
  CvWs w;
  void* loss_ws;
  int64_t loss_ws_bytes;
  cv_carve(ws, md, m_pos, m_neg, &w, &loss_ws, &loss_ws_bytes);
  float* sc = scores_out ? scores_out : w.sc;
  // destination lists (ids only)
  PullPlan plan;
  if (!rows)
    if (int rc = pull_prepare(md, pos, m_pos, neg, m_neg, w.pull_hdr, w.pull_body, st, &plan)) return rc;
  {
    ProfScope prof(KGE_PROF_FWD, st);
    if (int rc = cv_forward(md, E, Rel, theta, stats, pos, m_pos, 0, true, w.call[0], w.psum, sc, 1, err, seed, st)) return rc;
    if (int rc = cv_forward(md, E, Rel, theta, stats, neg, m_neg, m_pos, true, w.call[1], w.psum, sc + m_pos, 1, err, seed, st)) return rc;
  }
  if (int rc = loss_value(ld, sc, m_pos, sc + m_pos, m_neg, loss_out, w.g, w.g + m_pos, loss_ws, loss_ws_bytes, st)) return rc;
  {
    ProfScope prof(KGE_PROF_TRAIN_GRADS, st);
    if (int rc = cv_backward(md, E, Rel, theta, gtheta, pos, m_pos, 0, w.call[0], w, w.g, seed, st)) return rc;
    if (int rc = cv_backward(md, E, Rel, theta, gtheta, neg, m_neg, m_pos, w.call[1], w, w.g + m_pos, seed, st)) return rc;
    if (rows) {
      rows[0] = w.dS; rows[1] = w.dP; rows[2] = w.dO;
      KGE_HIP_TRY(hipGetLastError());
      return KGE_OK;
    }
    if (int rc = pull2_fill_entries(md, pos, m_pos, neg, m_neg, plan, st)) return rc;
    const int N = md->num_entities, NK = N + md->num_relations;
    rowsum_dispatch((int)cdiv(NK + plan.max_extra, 4), st, gE, gR, N, NK, md->d_e, plan, w.dS, w.dP, w.dO);
    if (int rc = pull_finish_launch(md, gE, gR, plan, st)) return rc;
  }
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

}  // namespace kge
