// LiteralE-style input path (KGEkeras/models.py:114-120,152-158; used by the reference's own script, tests.py:113,125-126):
//     s_hat = act(Dense([dp(E[s]) ; dp(Lit[s])])),   o_hat likewise with the SAME Dense,   score = func(s_hat, dp(Rel[p]), o_hat).
// The Dense layer is shared by all rows, so it is one GEMM over the stacked (2n, d+L) input (rocBLAS sgemm, fp32); the gather,
// activation, per-triple score / row gradients and the compaction around it are HIP kernels; the per-triple entity rows
// that come out of the backward GEMM are summed into the table gradient by the destination lists of pull.hip.
// theta_l = [W (d+L, d) row-major | b (d)].  Dropout multipliers: roles 0/1/2 = s/p/o rows, 5/6 = literal rows of s/o.
#include <rocblas/rocblas.h>

#include "dropout.hpp"
#include "kge_internal.hpp"
#include "pull.hpp"

namespace kge {

rocblas_handle blas_handle_shared();  // conve.hip
// conve.hip: ConvE's gradient half with the per-triple rows handed out instead of summed (rows = {dS, dP, dO})
int conve_train_grads_rows(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, const float* theta, float* stats,
                           float* gE, float* gR, float* gtheta, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg,
                           float* loss_out, float* scores_out, uint64_t seed, void* ws, int64_t wbytes, uint32_t* err, hipStream_t st,
                           const float** rows);
int64_t conve_workspace_bytes(const kge_model_desc* md, int64_t m_pos, int64_t m_neg);
// ConvE behind the literal transform (lit_train_grads): the head's parameters, moving statistics and parameter gradients
struct LitConve { const float* theta; float* stats; float* gtheta; };
// hake.hip: HAKE's gradient half (dense accumulators gE / gR += per-row gradients by float atomics) and its scratch
int hake_train_grads(const kge_model_desc* md, const kge_loss_desc* ld, const kge_hake_params* hp, const float* E, const float* Rel, float* gE, float* gR,
                     const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, float* loss_out, float* scores_out, uint64_t seed,
                     void* workspace, int64_t wbytes, uint32_t* err, hipStream_t st);
int64_t hake_workspace_bytes(int64_t n, int64_t chunk);
int cv_rowsum_launch(const kge_model_desc* md, float* gE, float* gR, const PullPlan& plan, const float* dS, const float* dP, const float* dO,
                     hipStream_t st);  // conve.hip
int pull2_fill_entries(const kge_model_desc* md, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, const PullPlan& p,
                       hipStream_t st);
int pull_finish_launch(const kge_model_desc* md, float* gE, float* gR, const PullPlan& p, hipStream_t st);
int64_t train_workspace_bytes(const kge_model_desc*, int64_t, int64_t);
int loss_value(const kge_loss_desc*, const float*, int64_t, const float*, int64_t, float*, float*, float*, void*, int64_t, hipStream_t);

__device__ __forceinline__ void act_eval(int act, float z, float& h, float& dh) {
  switch (act) {
    case 1: h = z > 0.f ? z : 0.f; dh = z > 0.f ? 1.f : 0.f; break;                                     // relu
    case 2: h = 1.f / (1.f + expf(-z)); dh = h * (1.f - h); break;                                      // sigmoid
    case 3: h = tanhf(z); dh = 1.f - h * h; break;                                                      // tanh
    case 4: h = z > 20.f ? z : log1pf(expf(z)); dh = 1.f / (1.f + expf(-z)); break;                     // softplus
    case 5: { const float a = 1.f + fabsf(z); h = z / a; dh = 1.f / (a * a); break; }                   // softsign
    case 6: { const float e = expf(fminf(z, 0.f)); h = z > 0.f ? z : e - 1.f; dh = z > 0.f ? 1.f : e; break; }  // elu
    case 7: { const float e = expf(fminf(z, 0.f)); const float A = 1.6732632423543772f, S = 1.0507009873554805f;
              h = S * (z > 0.f ? z : A * (e - 1.f)); dh = S * (z > 0.f ? 1.f : A * e); break; }          // selu
    case 8: h = expf(z); dh = h; break;                                                                 // exponential
    default: h = z; dh = 1.f; break;                                                                    // linear
  }
}

// A[t] = [E[s]*m | Lit[s]*m], A[n+t] = [E[o]*m | Lit[o]*m]; ids == nullptr: identity rows (table transform), single role.
template <bool DROP>
__global__ __launch_bounds__(256) void lit_gather(const float* __restrict__ E, const float* __restrict__ Lit, const int32_t* __restrict__ pos,
                                                   int64_t m_pos, const int32_t* __restrict__ neg, int64_t m_neg, int N, int NL, int d, int L,
                                                   float* __restrict__ A, uint32_t* __restrict__ err, uint64_t seed, uint32_t thr, float dscale) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t n = m_pos + m_neg;
  const int K1 = d + L;
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < n; t += (int64_t)gridDim.x * 4) {
    int s, o;
    if (pos) {
      const int32_t* tr = t < m_pos ? pos + 3 * t : neg + 3 * (t - m_pos);
      s = tr[0]; o = tr[2];
    } else { s = (int)t; o = (int)t; }
    const bool ok = (unsigned)s < (unsigned)N && (unsigned)o < (unsigned)N && (unsigned)s < (unsigned)NL && (unsigned)o < (unsigned)NL;
    if (!ok && lane == 0) atomicOr(err, 1u);
    const int roles = pos ? 2 : 1;
    for (int r = 0; r < roles; ++r) {
      const int id = ok ? (r == 0 ? s : o) : 0;
      float* dst = A + (r == 0 ? t : n + t) * K1;
      DropKey ke{0, 0}, kl{0, 0};
      if (DROP) { ke = drop_key(seed, t, r == 0 ? 0 : 2); kl = drop_key(seed, t, r == 0 ? 5 : 6); }
      for (int k = lane; k < K1; k += 64) {
        float v = k < d ? E[(int64_t)id * d + k] : Lit[(int64_t)id * L + (k - d)];
        if (DROP) v *= k < d ? drop_mul(ke, k, thr, dscale) : drop_mul(kl, k - d, thr, dscale);
        dst[k] = ok ? v : 0.f;
      }
    }
  }
}

// Zb = Z + b ; H = act(Zb)
__global__ __launch_bounds__(256) void lit_act_fwd(float* __restrict__ Z, float* __restrict__ H, int64_t rows, int d, const float* __restrict__ b,
                                                    int act) {
  const int64_t total = rows * d;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const float z = Z[i] + b[i % d];
    float h, dh;
    act_eval(act, z, h, dh);
    Z[i] = z;
    H[i] = h;
  }
}

// act 9 = softmax over the Dense layer's e_dim outputs (keras.activations.softmax, axis -1): the one activation of the reference's
// `literal_activation` choices that is not element-wise.  One wave per row: z = Z + b, h = exp(z - max z) / sum; Zb keeps h (the backward
// needs h, not z).  Backward: dz_i = h_i (dh_i - sum_j dh_j h_j), in place on dH, then the same per-block column sums as lit_act_bwd.
constexpr int kActSoftmax = 9;
__global__ __launch_bounds__(256) void lit_softmax_fwd(float* __restrict__ Z, float* __restrict__ H, int64_t rows, int d, const float* __restrict__ b) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
    float mx = -__builtin_inff();
    for (int k = lane; k < d; k += 64) mx = fmaxf(mx, Z[r * d + k] + b[k]);
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int k = lane; k < d; k += 64) sum += expf(Z[r * d + k] + b[k] - mx);
    sum = group_sum<64>(sum);
    const float inv = 1.f / sum;
    for (int k = lane; k < d; k += 64) {
      const float h = expf(Z[r * d + k] + b[k] - mx) * inv;
      H[r * d + k] = h;
      Z[r * d + k] = h;
    }
  }
}
__global__ __launch_bounds__(256) void lit_softmax_bwd_rows(float* __restrict__ dH, const float* __restrict__ Hs, int64_t rows, int d) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
    float dot = 0.f;
    for (int k = lane; k < d; k += 64) dot += dH[r * d + k] * Hs[r * d + k];
    dot = group_sum<64>(dot);
    for (int k = lane; k < d; k += 64) dH[r * d + k] = Hs[r * d + k] * (dH[r * d + k] - dot);
  }
}

constexpr int kLitHoleMax = 512;  // HolE rows in LDS: 4 waves x 3 rows x 512 floats = 24 KB

// per triple: s_hat = H[t], o_hat = H[n+t], p' = Rel[p]*mask.  MODE 0: score.  MODE 1: row gradients dH[t], dH[n+t], dP[t] (masked).
template <int MODEL, int NORM, int MODE, bool DROP>
__global__ __launch_bounds__(256) void lit_rows(const float* __restrict__ Rel, const int32_t* __restrict__ pos, int64_t m_pos,
                                                 const int32_t* __restrict__ neg, int64_t m_neg, int R, int d, float gamma, float modulus,
                                                 const float* __restrict__ H, const float* __restrict__ g, float* __restrict__ scores,
                                                 float* __restrict__ dH, float* __restrict__ dP, uint64_t seed, uint32_t thr, float dscale) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t n = m_pos + m_neg;
  const int h = d >> 1;
  const int width = MODEL == KGE_COMPLEX || MODEL == KGE_ROTATE ? h : d;
  const int dr = MODEL == KGE_ROTATE ? h : d;  // width of a relation row (RotatE: h phases, models.py:428-429)
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < n; t += (int64_t)gridDim.x * 4) {
    const int32_t* tr = t < m_pos ? pos + 3 * t : neg + 3 * (t - m_pos);
    const int p = tr[1];
    const bool ok = (unsigned)p < (unsigned)R;
    const float* hs = H + t * d;
    const float* ho = H + (n + t) * d;
    const float* rp = Rel + (int64_t)(ok ? p : 0) * dr;
    DropKey kp{0, 0};
    if (DROP) kp = drop_key(seed, t, 1);
    auto mp = [&](int e) { return DROP ? drop_mul(kp, e, thr, dscale) : 1.f; };
    if (MODEL == KGE_HOLE) {
      // models.py:231-237 on the transformed rows: score = sum_k p'_k c_k,  c_k = sum_i s_i o_{(i+k) mod d} (circular cross-correlation);
      // d/ds_i = g sum_k p'_k o_{(i+k)},  d/do_j = g sum_k p'_k s_{(j-k)},  d/dp_k = g c_k (times p's dropout factor).
      // The three rows of the triple sit in LDS (this wave's slice); direct O(d^2) sums, ascending index order.
      __shared__ float hb[4][3][kLitHoleMax];
      float* ls = hb[wave][0];
      float* lo = hb[wave][1];
      float* lp = hb[wave][2];
      for (int k = lane; k < d; k += 64) { ls[k] = hs[k]; lo[k] = ho[k]; lp[k] = rp[k] * mp(k); }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float gi = 0.f;
      if (MODE == 1) { gi = ok ? g[t] : 0.f; if (!(gi == gi)) gi = 0.f; }
      float accp = 0.f;
      for (int k = lane; k < d; k += 64) {
        float c = 0.f;
        int j = k;
        for (int i = 0; i < d; ++i) { c += ls[i] * lo[j]; j = j + 1 == d ? 0 : j + 1; }
        accp += lp[k] * c;
        if (MODE == 1) dP[t * d + k] = gi * c * mp(k);
      }
      if (MODE == 0) {
        accp = group_sum<64>(accp);
        if (lane == 0) scores[t] = ok ? accp : __builtin_nanf("");
      } else {
        for (int i = lane; i < d; i += 64) {
          float a = 0.f, b = 0.f;
          int jo = i, js = i;                    // o index (i + k) mod d, s index (i - k) mod d
          for (int k = 0; k < d; ++k) {
            a += lp[k] * lo[jo];
            b += lp[k] * ls[js];
            jo = jo + 1 == d ? 0 : jo + 1;
            js = js == 0 ? d - 1 : js - 1;
          }
          dH[t * d + i] = gi * a;
          dH[(n + t) * d + i] = gi * b;
        }
      }
      __builtin_amdgcn_wave_barrier();  // (the next triple of this wave overwrites the slice)
      continue;
    }
    if (MODEL == KGE_ROTATE) {
      // models.py:437-459 on the transformed rows: the rotated subject minus the object, summed with its signs --
      // S = sum_k (sr c - si s - or) + (sr s + si c - oi), c / s = cos / sin of the (masked) phase; score = gamma - S (gamma > 0) or S.
      // Linear in s and o: dS/dsr = c + s, dS/dsi = c - s, dS/do = -1; dS/dphase = sr (c - s) - si (c + s).
      // dP rows are d wide like every model's (the destination sums run on rows of d): the upper half is written as zeros.
      float acc = 0.f;
      for (int k = lane; k < h; k += 64) {
        float sn, cs;
        sincosf(rp[k] * mp(k), &sn, &cs);
        acc += (hs[k] * cs - hs[h + k] * sn - ho[k]) + (hs[k] * sn + hs[h + k] * cs - ho[h + k]);
      }
      acc = group_sum<64>(acc);
      if (MODE == 0) {
        if (lane == 0) scores[t] = ok ? (gamma > 0.f ? gamma - acc : acc) : __builtin_nanf("");
        continue;
      }
      float gi = ok ? g[t] : 0.f;
      if (!(gi == gi)) gi = 0.f;
      const float sg = gamma > 0.f ? -gi : gi;
      for (int k = lane; k < h; k += 64) {
        const float b = mp(k);
        float sn, cs;
        sincosf(rp[k] * b, &sn, &cs);
        dH[t * d + k] = sg * (cs + sn); dH[t * d + h + k] = sg * (cs - sn);
        dH[(n + t) * d + k] = -sg; dH[(n + t) * d + h + k] = -sg;
        dP[t * d + k] = sg * (hs[k] * (cs - sn) - hs[h + k] * (cs + sn)) * b;
        dP[t * d + h + k] = 0.f;
      }
      continue;
    }
    float acc = 0.f;
    for (int k = lane; k < width; k += 64) {
      if (MODEL == KGE_COMPLEX) {
        const float sr = hs[k], si = hs[h + k], pr = rp[k] * mp(k), pi = rp[h + k] * mp(h + k), orr = ho[k], oi = ho[h + k];
        acc += sr * pr * orr + pr * si * oi + pi * sr * oi - pi * si * orr;
      } else {
        const float sv = hs[k], pv = rp[k] * mp(k), ov = ho[k];
        if (MODEL == KGE_DISTMULT) acc += sv * pv * ov;
        else {  // NORM 1 / 2: |x|, x^2;  3: the pRotatE phase distance |sin(x/2)| (models.py:479-489);  5: |x|^p, p in `modulus`
          const float x = sv + pv - ov;
          acc += NORM == 1 ? fabsf(x) : (NORM == 3 ? phase_abs(x) : (NORM == 5 ? powf(fabsf(x), modulus) : x * x));
        }
      }
    }
    acc = group_sum<64>(acc);
    float dist = acc, score = acc;
    if (MODEL == KGE_TRANSE) {
      if (NORM == 2) dist = sqrtf(acc);
      if (NORM == 3) dist = modulus * acc;
      if (NORM == 5) dist = powf(acc, 1.f / modulus);
      score = gamma > 0.f ? gamma - dist : dist;
    }
    if (MODE == 0) {
      if (lane == 0) scores[t] = ok ? score : __builtin_nanf("");
      continue;
    }
    float gi = ok ? g[t] : 0.f;
    if (!(gi == gi)) gi = 0.f;
    float* ds = dH + t * d;
    float* dob = dH + (n + t) * d;
    float* dp = dP + t * d;
    if (MODEL == KGE_TRANSE) {
      const float sgn = gamma > 0.f ? -gi : gi;
      // d dist / d x_k:  1: sign(x);  2: x / dist;  3: modulus/2 * sign(sin(x/2)) cos(x/2);  5: sign(x) |x|^(p-1) dist^(1-p)
      float rn = NORM == 2 ? (dist > 0.f ? sgn / dist : 0.f) : sgn;
      if (NORM == 3) rn = 0.5f * modulus * sgn;
      if (NORM == 5) rn = dist > 0.f ? sgn * powf(dist, 1.f - modulus) : 0.f;
      for (int k = lane; k < width; k += 64) {
        const float b = mp(k);
        const float x = hs[k] + rp[k] * b - ho[k];
        float dx;
        if (NORM == 1) dx = x > 0.f ? rn : (x < 0.f ? -rn : 0.f);
        else if (NORM == 3) dx = rn * phase_dterm(x);
        else if (NORM == 5) dx = x == 0.f ? 0.f : rn * copysignf(powf(fabsf(x), modulus - 1.f), x);
        else dx = rn * x;
        ds[k] = dx; dob[k] = -dx; dp[k] = dx * b;
      }
    } else if (MODEL == KGE_DISTMULT) {
      for (int k = lane; k < width; k += 64) {
        const float b = mp(k);
        const float sv = hs[k], pv = rp[k] * b, ov = ho[k];
        ds[k] = gi * pv * ov; dob[k] = gi * sv * pv; dp[k] = gi * sv * ov * b;
      }
    } else {
      for (int k = lane; k < width; k += 64) {
        const float br = mp(k), bi = mp(h + k);
        const float sr = hs[k], si = hs[h + k], pr = rp[k] * br, pi = rp[h + k] * bi, orr = ho[k], oi = ho[h + k];
        ds[k] = gi * (pr * orr + pi * oi);      ds[h + k] = gi * (pr * oi - pi * orr);
        dp[k] = gi * (sr * orr + si * oi) * br; dp[h + k] = gi * (sr * oi - si * orr) * bi;
        dob[k] = gi * (sr * pr - pi * si);      dob[h + k] = gi * (pr * si + pi * sr);
      }
    }
  }
}

// ConvE behind the literal transform: the head runs on STAGED tables -- entity rows = the 2n transformed rows of the batch (H), relation
// rows = the batch's n relation rows with their dropout masks applied (Pm), triples (t, t, n + t) -- so its gradient rows with respect
// to those tables are dL/d(transformed rows) and dL/d(masked relation rows).  One wave per batch row.
template <bool DROP>
__global__ __launch_bounds__(256) void lit_stage_rel(const float* __restrict__ Rel, const int32_t* __restrict__ pos, int64_t m_pos,
                                                      const int32_t* __restrict__ neg, int64_t m_neg, int R, int d, float* __restrict__ Pm,
                                                      int32_t* __restrict__ tri, uint64_t seed, uint32_t thr, float dscale, uint32_t* __restrict__ err) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t n = m_pos + m_neg;
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < n; t += (int64_t)gridDim.x * 4) {
    const int32_t* tr = t < m_pos ? pos + 3 * t : neg + 3 * (t - m_pos);
    const int p = tr[1];
    const bool ok = (unsigned)p < (unsigned)R;
    if (!ok && lane == 0) atomicOr(err, 1u);
    const float* rp = Rel + (int64_t)(ok ? p : 0) * d;
    DropKey kp{0, 0};
    if (DROP) kp = drop_key(seed, t, 1);
    for (int k = lane; k < d; k += 64) Pm[t * d + k] = ok ? rp[k] * (DROP ? drop_mul(kp, k, thr, dscale) : 1.f) : 0.f;
    if (lane == 0) { tri[3 * t] = (int32_t)t; tri[3 * t + 1] = (int32_t)t; tri[3 * t + 2] = (int32_t)(n + t); }
  }
}
// dP[t] = rows[t] * mask(t, role 1): the gradient with respect to the UNMASKED relation row of batch row t
template <bool DROP>
__global__ __launch_bounds__(256) void lit_mask_rel(const float* __restrict__ rows, int64_t n, int d, float* __restrict__ dP, uint64_t seed, uint32_t thr,
                                                     float dscale) {
  const int64_t total = n * d;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / d;
    const int k = (int)(i - t * d);
    dP[i] = rows[i] * (DROP ? drop_mul(drop_key(seed, t, 1), k, thr, dscale) : 1.f);
  }
}

// HAKE behind the literal transform (relation rows 3h wide, entity rows 2h: the destination sums of this path run on rows of ONE width):
// gR[p_t] += rows[t] * mask(t, role 1) by float atomics, like HAKE's own gradient path
template <bool DROP>
__global__ __launch_bounds__(256) void lit_scatter_rel(const float* __restrict__ rows, const int32_t* __restrict__ pos, int64_t m_pos,
                                                        const int32_t* __restrict__ neg, int64_t m_neg, int R, int dr, float* __restrict__ gR, uint64_t seed,
                                                        uint32_t thr, float dscale) {
  const int64_t n = m_pos + m_neg, total = n * dr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / dr;
    const int k = (int)(i - t * dr);
    const int p = (t < m_pos ? pos + 3 * t : neg + 3 * (t - m_pos))[1];
    if ((unsigned)p >= (unsigned)R) continue;
    const float v = rows[i] * (DROP ? drop_mul(drop_key(seed, t, 1), k, thr, dscale) : 1.f);
    if (v != 0.f) unsafeAtomicAdd(gR + (int64_t)p * dr + k, v);
  }
}

// dZ = dH * act'(Zb) in place; per-block partial column sums (64 rows per block) for the bias gradient.
__global__ __launch_bounds__(256) void lit_act_bwd(float* __restrict__ dH, const float* __restrict__ Zb, int64_t rows, int d, int act,
                                                    float* __restrict__ psum) {
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  for (int k = threadIdx.x; k < d; k += 256) {
    float s = 0.f;
    for (int64_t r = r0; r < min(rows, r0 + 64); ++r) {
      float h, dh;
      act_eval(act, Zb[r * d + k], h, dh);
      const float v = dH[r * d + k] * dh;
      dH[r * d + k] = v;
      s += v;
    }
    psum[(int64_t)blockIdx.x * d + k] = s;
  }
}

__global__ __launch_bounds__(256) void lit_colsum(const float* __restrict__ partials, int nb, int n, float* __restrict__ out) {
  __shared__ double sa[256];
  const int c = blockIdx.x;
  double x = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) x += partials[(int64_t)i * n + c];
  sa[threadIdx.x] = x;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sa[threadIdx.x] += sa[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) out[c] += (float)sa[0];
}

// dS[t] = dA[t][:d] * mask(t, role 0), dO[t] = dA[n+t][:d] * mask(t, role 2)
template <bool DROP>
__global__ __launch_bounds__(256) void lit_compact(const float* __restrict__ dA, int64_t n, int d, int K1, float* __restrict__ dS,
                                                    float* __restrict__ dO, uint64_t seed, uint32_t thr, float dscale) {
  const int64_t total = n * d;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / d;
    const int k = (int)(i - t * d);
    float a = dA[t * K1 + k], b = dA[(n + t) * K1 + k];
    if (DROP) { a *= drop_mul(drop_key(seed, t, 0), k, thr, dscale); b *= drop_mul(drop_key(seed, t, 2), k, thr, dscale); }
    dS[i] = a;
    dO[i] = b;
  }
}

// ---------------------------------------------------------------------------------------------------------------
#define KGE_BLAS_TRY(expr)                                                                  \
  do {                                                                                      \
    rocblas_status s__ = (expr);                                                            \
    if (s__ != rocblas_status_success) return fail(KGE_EHIP, "%s: rocblas status %d", #expr, (int)s__); \
  } while (0)

static int64_t l256(int64_t x) { return align_up(x, 256); }

static int lit_check(const kge_model_desc* md, int L, int act) {
  if (md->model != KGE_DISTMULT && md->model != KGE_TRANSE && md->model != KGE_COMPLEX && md->model != KGE_HOLE && md->model != KGE_ROTATE &&
      md->model != KGE_CONVE && md->model != KGE_HAKE)
    return fail(KGE_EUNSUPPORTED, "literal path: model %d", md->model);
  if (md->model == KGE_ROTATE && md->d_e != 2 * md->d_r) return fail(KGE_EINVAL, "literal path under RotatE: d_e must be 2 * d_r");
  if (md->model == KGE_HOLE && md->d_e > kLitHoleMax)
    return fail(KGE_EUNSUPPORTED, "literal path under HolE: rows of at most %d floats (e_dim = %d)", kLitHoleMax, md->d_e);
  if (L <= 0) return fail(KGE_EINVAL, "literal path: literal width %d", L);
  if (act < 0 || act > kActSoftmax) return fail(KGE_EUNSUPPORTED, "literal path: activation id %d (custom callables are not built)", act);
  return KGE_OK;
}

// Ehat (rows, d) = act([E | Lit] W + b) for table rows 0..rows-1, inference mode (no dropout)
int lit_transform(const kge_model_desc* md, int act, const float* E, const float* Lit, int L, int lit_rows_n, const float* theta, int rows,
                  float* out, void* ws, int64_t wbytes, uint32_t* err, hipStream_t st) {
  if (int rc = lit_check(md, L, act)) return rc;
  const int d = md->d_e, K1 = d + L;
  const int64_t need = l256((int64_t)rows * K1 * 4) + 512;
  if (wbytes < need) return fail(KGE_EWORKSPACE, "literal transform: workspace has %lld bytes, needs %lld", (long long)wbytes, (long long)need);
  float* A = reinterpret_cast<float*>(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  hipLaunchKernelGGL((lit_gather<false>), dim3((int)std::min<int64_t>(cdiv(rows, 4), 4096)), dim3(256), 0, st, E, Lit, (const int32_t*)nullptr, (int64_t)rows,
                     (const int32_t*)nullptr, (int64_t)0, md->num_entities, lit_rows_n, d, L, A, err, 0, 0u, 1.f);
  rocblas_handle h = blas_handle_shared();
  if (!h) return fail(KGE_EHIP, "literal path: rocblas handle could not be created");
  KGE_BLAS_TRY(rocblas_set_stream(h, st));
  const float one = 1.f, zero = 0.f;
  KGE_BLAS_TRY(rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_none, d, rows, K1, &one, theta, d, A, K1, &zero, out, d));
  if (act == kActSoftmax)
    hipLaunchKernelGGL(lit_softmax_fwd, dim3((int)std::min<int64_t>(cdiv(rows, 4), 8192)), dim3(256), 0, st, out, out, (int64_t)rows, d, theta + (int64_t)K1 * d);
  else
  hipLaunchKernelGGL(lit_act_fwd, dim3((int)std::min<int64_t>(cdiv((int64_t)rows * d, 256), 8192)), dim3(256), 0, st, out, out, (int64_t)rows, d,
                     theta + (int64_t)K1 * d, act);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

int64_t lit_transform_workspace_bytes(const kge_model_desc* md, int L, int rows) { return l256((int64_t)rows * (md->d_e + L) * 4) + 1024; }

// The weight gradient dW (d x K1) = dZ^T A contracts over all 2n stacked rows: one output tile, i.e. ONE workgroup in a plain
// GEMM (2.2 ms at 2n = 81 920).  It is computed as `splits` slabs of the contraction (strided-batched GEMM, one tile each)
// whose partial matrices a small kernel adds up in slab order.
static int lit_splits(int64_t rows2n, int64_t d, int64_t K1) {
  const int64_t by_rows = rows2n / 2048, by_bytes = (32ll << 20) / std::max<int64_t>(1, d * K1 * 4);
  return (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(by_rows, by_bytes), 256));
}

__global__ __launch_bounds__(256) void lit_splitk_sum(const float* __restrict__ part, int splits, int64_t elems, float* __restrict__ acc) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < elems; i += (int64_t)gridDim.x * 256) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += part[(int64_t)k * elems + i];
    acc[i] += s;
  }
}

// ConvE behind the transform: its staged descriptor (2n entity rows, n relation rows, input dropout already applied) and what it adds to the workspace
static kge_model_desc lit_conve_desc(const kge_model_desc* md, int64_t n) {
  kge_model_desc m2 = *md;
  m2.num_entities = (int32_t)(2 * n);
  m2.num_relations = (int32_t)n;
  m2.dropout = 0.f;
  return m2;
}
static int64_t lit_conve_bytes(const kge_model_desc* md, int64_t m_pos, int64_t m_neg) {
  const int64_t n = m_pos + m_neg;
  const kge_model_desc m2 = lit_conve_desc(md, n);
  return l256(n * md->d_e * 4) + l256(n * 12) + l256(conve_workspace_bytes(&m2, m_pos, m_neg)) + 1024;
}
static int64_t lit_hake_bytes(const kge_model_desc* md, int64_t m_pos, int64_t m_neg) {
  const int64_t n = m_pos + m_neg;
  return 2 * l256(n * md->d_r * 4) + l256(n * 12) + l256(hake_workspace_bytes(n, 0)) + 1024;
}
int64_t lit_workspace_bytes(const kge_model_desc* md, int L, int64_t m_pos, int64_t m_neg) {
  const int64_t n = m_pos + m_neg, d = md->d_e, K1 = d + L;
  return l256((int64_t)lit_splits(2 * n, d, K1) * d * K1 * 4) + pull_header_bytes(md) + 2 * l256(2 * n * K1 * 4) + 3 * l256(2 * n * d * 4) + 3 * l256(n * d * 4) + 2 * l256(n * 4) +
         l256(cdiv(2 * n, 64) * d * 4) + pull_workspace_bytes(md, n) + l256(train_workspace_bytes(nullptr, m_pos, m_neg)) + 4096 +
         (md->model == KGE_ROTATE || md->model == KGE_HAKE ? l256((int64_t)md->num_relations * d * 4) : 0) +
         (md->model == KGE_CONVE ? lit_conve_bytes(md, m_pos, m_neg) : 0) + (md->model == KGE_HAKE ? lit_hake_bytes(md, m_pos, m_neg) : 0);
}

// RotatE: the relation rows are h = d / 2 wide, the destination sums run on rows of d.  The per-triple rows dP carry zeros in their
// upper half and are summed into a padded table gRp (R x d, all-zero between steps); this folds its lower half into gR and clears it.
__global__ __launch_bounds__(256) void lit_fold_rel(float* __restrict__ gRp, float* __restrict__ gR, int R, int d) {
  const int h = d >> 1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)R * d; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / d), k = (int)(i % d);
    if (gR && k < h) gR[(int64_t)r * h + k] += gRp[i];
    gRp[i] = 0.f;
  }
}

template <int MODEL, int NORM, int MODE>
static void launch_rows(bool drop, int grid, hipStream_t st, const float* Rel, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg,
                        const kge_model_desc* md, const float* H, const float* g, float* scores, float* dH, float* dP, uint64_t seed) {
  const uint32_t thr = drop_thr(md->dropout);
  const float dscale = 1.f / (1.f - md->dropout);
  if (drop) hipLaunchKernelGGL((lit_rows<MODEL, NORM, MODE, true>), dim3(grid), dim3(256), 0, st, Rel, pos, m_pos, neg, m_neg, md->num_relations, md->d_e, md->gamma, md->modulus, H, g, scores, dH, dP, seed, thr, dscale);
  else hipLaunchKernelGGL((lit_rows<MODEL, NORM, MODE, false>), dim3(grid), dim3(256), 0, st, Rel, pos, m_pos, neg, m_neg, md->num_relations, md->d_e, md->gamma, md->modulus, H, g, scores, dH, dP, seed, thr, dscale);
}

template <int MODE>
static void launch_rows_model(bool drop, int grid, hipStream_t st, const float* Rel, const int32_t* pos, int64_t m_pos, const int32_t* neg,
                              int64_t m_neg, const kge_model_desc* md, const float* H, const float* g, float* scores, float* dH, float* dP,
                              uint64_t seed) {
  if (md->model == KGE_DISTMULT) launch_rows<KGE_DISTMULT, 1, MODE>(drop, grid, st, Rel, pos, m_pos, neg, m_neg, md, H, g, scores, dH, dP, seed);
  else if (md->model == KGE_COMPLEX) launch_rows<KGE_COMPLEX, 1, MODE>(drop, grid, st, Rel, pos, m_pos, neg, m_neg, md, H, g, scores, dH, dP, seed);
  else if (md->model == KGE_HOLE) launch_rows<KGE_HOLE, 1, MODE>(drop, grid, st, Rel, pos, m_pos, neg, m_neg, md, H, g, scores, dH, dP, seed);
  else if (md->model == KGE_ROTATE) launch_rows<KGE_ROTATE, 1, MODE>(drop, grid, st, Rel, pos, m_pos, neg, m_neg, md, H, g, scores, dH, dP, seed);
  else if (md->norm == 1) launch_rows<KGE_TRANSE, 1, MODE>(drop, grid, st, Rel, pos, m_pos, neg, m_neg, md, H, g, scores, dH, dP, seed);
  else if (md->norm == 3) launch_rows<KGE_TRANSE, 3, MODE>(drop, grid, st, Rel, pos, m_pos, neg, m_neg, md, H, g, scores, dH, dP, seed);
  else if (md->norm == 5) launch_rows<KGE_TRANSE, 5, MODE>(drop, grid, st, Rel, pos, m_pos, neg, m_neg, md, H, g, scores, dH, dP, seed);
  else launch_rows<KGE_TRANSE, 2, MODE>(drop, grid, st, Rel, pos, m_pos, neg, m_neg, md, H, g, scores, dH, dP, seed);
}

int lit_train_grads(const kge_model_desc* md, const kge_loss_desc* ld, int act, const float* E, const float* Rel, const float* Lit, int L,
                    int lit_rows_n, const float* theta, float* gE, float* gR, float* gtheta, const int32_t* pos, int64_t m_pos,
                    const int32_t* neg, int64_t m_neg, float* loss_out, float* scores_out, uint64_t seed, void* ws, int64_t wbytes, uint32_t* err,
                    hipStream_t st, const LitConve* conve, const kge_hake_params* hake) {
  if (int rc = lit_check(md, L, act)) return rc;
  if ((md->model == KGE_CONVE) != (conve != nullptr)) return fail(KGE_EINVAL, "literal path: ConvE goes through kge_lit_conve_train_grads (and only ConvE)");
  if ((md->model == KGE_HAKE) != (hake != nullptr)) return fail(KGE_EINVAL, "literal path: HAKE goes through kge_lit_hake_train_grads (and only HAKE)");
  if (2 * (m_pos + m_neg) > 0x7fffffff) return fail(KGE_EINVAL, "literal path: batch too large");
  if (wbytes < lit_workspace_bytes(md, L, m_pos, m_neg))
    return fail(KGE_EWORKSPACE, "literal path: workspace has %lld bytes, needs %lld", (long long)wbytes, (long long)lit_workspace_bytes(md, L, m_pos, m_neg));
  const int64_t n = m_pos + m_neg;
  const int d = md->d_e, K1 = d + L;
  char* q = reinterpret_cast<char*>(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  auto take = [&](int64_t bytes) { char* r = q; q += l256(bytes); return r; };
  void* pull_hdr = take(pull_header_bytes(md));
  // (RotatE: right behind the header, in the part of the workspace that is zero on entry and left zero -- see lit_fold_rel)
  float* gRp = md->model == KGE_ROTATE || md->model == KGE_HAKE ? (float*)take((int64_t)md->num_relations * md->d_e * 4) : nullptr;
  float* A = (float*)take(2 * n * K1 * 4);
  float* dA = (float*)take(2 * n * K1 * 4);
  float* Zb = (float*)take(2 * n * d * 4);
  float* H = (float*)take(2 * n * d * 4);
  float* dH = (float*)take(2 * n * d * 4);
  float* dPb = (float*)take(n * d * 4);
  float* dS = (float*)take(n * d * 4);
  float* dO = (float*)take(n * d * 4);
  float* g = (float*)take(n * 4);
  float* scb = (float*)take(n * 4);
  const int nb_rows = (int)cdiv(2 * n, 64);
  float* psum = (float*)take((int64_t)nb_rows * d * 4);
  void* pull_body = take(pull_workspace_bytes(md, n));
  const int64_t loss_ws_bytes = train_workspace_bytes(nullptr, m_pos, m_neg);
  void* loss_ws = take(loss_ws_bytes);
  const int splits = lit_splits(2 * n, d, K1);
  float* wpart = (float*)take((int64_t)splits * d * K1 * 4);
  float* Pm = nullptr;
  int32_t* tri2 = nullptr;
  void* cws = nullptr;
  kge_model_desc md2 = *md;
  int64_t cws_bytes = 0;
  float* dPm = nullptr;
  const int dr = md->d_r;
  if (conve) {
    md2 = lit_conve_desc(md, n);
    Pm = (float*)take(n * d * 4);
    tri2 = (int32_t*)take(n * 12);
    cws_bytes = conve_workspace_bytes(&md2, m_pos, m_neg);
    cws = take(cws_bytes);
  } else if (hake) {
    md2 = lit_conve_desc(md, n);  // (the same staging: 2n entity rows, n relation rows, input dropout already applied)
    Pm = (float*)take(n * dr * 4);
    dPm = (float*)take(n * dr * 4);
    tri2 = (int32_t*)take(n * 12);
    cws_bytes = hake_workspace_bytes(n, 0);
    cws = take(cws_bytes);
  }
  float* sc = scores_out ? scores_out : scb;
  const bool drop = md->dropout > 0.f;
  const uint32_t thr = drop_thr(md->dropout);
  const float dscale = 1.f / (1.f - md->dropout);
  PullPlan plan;
  if (int rc = pull_prepare(md, pos, m_pos, neg, m_neg, pull_hdr, pull_body, st, &plan)) return rc;
  rocblas_handle hb = blas_handle_shared();
  if (!hb) return fail(KGE_EHIP, "literal path: rocblas handle could not be created");
  KGE_BLAS_TRY(rocblas_set_stream(hb, st));
  const float one = 1.f, zero = 0.f;
  const int gw = (int)std::min<int64_t>(cdiv(n, 4), 4096);
  const int gel = (int)std::min<int64_t>(cdiv(2 * n * d, 256), 8192);
  {
    ProfScope prof(KGE_PROF_FWD, st);
    if (drop) hipLaunchKernelGGL((lit_gather<true>), dim3(gw), dim3(256), 0, st, E, Lit, pos, m_pos, neg, m_neg, md->num_entities, lit_rows_n, d, L, A, err, seed, thr, dscale);
    else hipLaunchKernelGGL((lit_gather<false>), dim3(gw), dim3(256), 0, st, E, Lit, pos, m_pos, neg, m_neg, md->num_entities, lit_rows_n, d, L, A, err, seed, thr, dscale);
    KGE_BLAS_TRY(rocblas_sgemm(hb, rocblas_operation_none, rocblas_operation_none, d, (rocblas_int)(2 * n), K1, &one, theta, d, A, K1, &zero, Zb, d));
    if (act == kActSoftmax) hipLaunchKernelGGL(lit_softmax_fwd, dim3((int)std::min<int64_t>(cdiv(2 * n, 4), 8192)), dim3(256), 0, st, Zb, H, 2 * n, d, theta + (int64_t)K1 * d);
    else hipLaunchKernelGGL(lit_act_fwd, dim3(gel), dim3(256), 0, st, Zb, H, 2 * n, d, theta + (int64_t)K1 * d, act);
    if (!conve && !hake) launch_rows_model<0>(drop, gw, st, Rel, pos, m_pos, neg, m_neg, md, H, nullptr, sc, nullptr, nullptr, seed);
  }
  if (hake) {
    // models.py:152-158 in front of HAKE.func (:383-400): one call of the model per side on the staged rows (the call-wide norm is the
    // side's, as in HAKE's own train step); its dense accumulators over the staged tables ARE the per-row gradients
    if (drop) hipLaunchKernelGGL((lit_stage_rel<true>), dim3(gw), dim3(256), 0, st, Rel, pos, m_pos, neg, m_neg, md->num_relations, dr, Pm, tri2, seed, thr, dscale, err);
    else hipLaunchKernelGGL((lit_stage_rel<false>), dim3(gw), dim3(256), 0, st, Rel, pos, m_pos, neg, m_neg, md->num_relations, dr, Pm, tri2, seed, thr, dscale, err);
    KGE_HIP_TRY(hipMemsetAsync(dH, 0, (size_t)2 * n * d * 4, st));
    KGE_HIP_TRY(hipMemsetAsync(dPm, 0, (size_t)n * dr * 4, st));
    KGE_HIP_TRY(hipMemsetAsync(dPb, 0, (size_t)n * d * 4, st));  // (the relation keys of the destination sums add zeros into the padded table)
    if (int rc = hake_train_grads(&md2, ld, hake, H, Pm, dH, dPm, tri2, m_pos, tri2 + 3 * m_pos, m_neg, loss_out, sc, seed, cws, cws_bytes, err, st)) return rc;
    const int gm = (int)std::min<int64_t>(cdiv(n * dr, 256), 8192);
    if (drop) hipLaunchKernelGGL((lit_scatter_rel<true>), dim3(gm), dim3(256), 0, st, dPm, pos, m_pos, neg, m_neg, md->num_relations, dr, gR, seed, thr, dscale);
    else hipLaunchKernelGGL((lit_scatter_rel<false>), dim3(gm), dim3(256), 0, st, dPm, pos, m_pos, neg, m_neg, md->num_relations, dr, gR, seed, thr, dscale);
  } else if (conve) {
    // models.py:152-158 in front of ConvE.func (:264-275): the head (batch statistics, hidden dropout, its own loss call) on the staged rows
    if (drop) hipLaunchKernelGGL((lit_stage_rel<true>), dim3(gw), dim3(256), 0, st, Rel, pos, m_pos, neg, m_neg, md->num_relations, d, Pm, tri2, seed, thr, dscale, err);
    else hipLaunchKernelGGL((lit_stage_rel<false>), dim3(gw), dim3(256), 0, st, Rel, pos, m_pos, neg, m_neg, md->num_relations, d, Pm, tri2, seed, thr, dscale, err);
    const float* rows[3] = {nullptr, nullptr, nullptr};
    if (int rc = conve_train_grads_rows(&md2, ld, H, Pm, conve->theta, conve->stats, nullptr, nullptr, conve->gtheta, tri2, m_pos, tri2 + 3 * m_pos, m_neg, loss_out,
                                        sc, seed, cws, cws_bytes, err, st, rows))
      return rc;
    KGE_HIP_TRY(hipMemcpyAsync(dH, rows[0], (size_t)n * d * 4, hipMemcpyDeviceToDevice, st));
    KGE_HIP_TRY(hipMemcpyAsync(dH + n * d, rows[2], (size_t)n * d * 4, hipMemcpyDeviceToDevice, st));
    const int gm = (int)std::min<int64_t>(cdiv(n * d, 256), 8192);
    if (drop) hipLaunchKernelGGL((lit_mask_rel<true>), dim3(gm), dim3(256), 0, st, rows[1], n, d, dPb, seed, thr, dscale);
    else hipLaunchKernelGGL((lit_mask_rel<false>), dim3(gm), dim3(256), 0, st, rows[1], n, d, dPb, seed, thr, dscale);
  } else if (int rc = loss_value(ld, sc, m_pos, sc + m_pos, m_neg, loss_out, g, g + m_pos, loss_ws, loss_ws_bytes, st)) {
    return rc;
  }
  {
    ProfScope prof(KGE_PROF_TRAIN_GRADS, st);
    if (!conve && !hake) launch_rows_model<1>(drop, gw, st, Rel, pos, m_pos, neg, m_neg, md, H, g, nullptr, dH, dPb, seed);
    // (softmax: rows first -- dz = h (dh - <dh, h>) -- then the column sums through the linear branch of lit_act_bwd)
    if (act == kActSoftmax) hipLaunchKernelGGL(lit_softmax_bwd_rows, dim3((int)std::min<int64_t>(cdiv(2 * n, 4), 8192)), dim3(256), 0, st, dH, Zb, 2 * n, d);
    hipLaunchKernelGGL(lit_act_bwd, dim3(nb_rows), dim3(256), 0, st, dH, Zb, 2 * n, d, act == kActSoftmax ? 0 : act, psum);
    hipLaunchKernelGGL(lit_colsum, dim3(d), dim3(256), 0, st, psum, nb_rows, d, gtheta + (int64_t)K1 * d);
    // dA(2n,K1) = dZ(2n,d) W^T ; dW(K1,d) += A^T dZ   (column-major views as in conve.hip)
    KGE_BLAS_TRY(rocblas_sgemm(hb, rocblas_operation_transpose, rocblas_operation_none, K1, (rocblas_int)(2 * n), d, &one, theta, d, dH, d, &zero, dA, K1));
    if (splits > 1) {
      const int64_t kc = (2 * n) / splits, tail = 2 * n - kc * splits;
      KGE_BLAS_TRY(rocblas_sgemm_strided_batched(hb, rocblas_operation_none, rocblas_operation_transpose, d, K1, (rocblas_int)kc, &one, dH, d, kc * d, A, K1,
                                                 kc * K1, &zero, wpart, d, (rocblas_stride)d * K1, splits));
      hipLaunchKernelGGL(lit_splitk_sum, dim3((int)cdiv((int64_t)d * K1, 256)), dim3(256), 0, st, wpart, splits, (int64_t)d * K1, gtheta);
      if (tail) KGE_BLAS_TRY(rocblas_sgemm(hb, rocblas_operation_none, rocblas_operation_transpose, d, K1, (rocblas_int)tail, &one, dH + kc * splits * d, d,
                                           A + kc * splits * K1, K1, &one, gtheta, d));
    } else {
      KGE_BLAS_TRY(rocblas_sgemm(hb, rocblas_operation_none, rocblas_operation_transpose, d, K1, (rocblas_int)(2 * n), &one, dH, d, A, K1, &one, gtheta, d));
    }
    const int gc = (int)std::min<int64_t>(cdiv(n * d, 256), 8192);
    if (drop) hipLaunchKernelGGL((lit_compact<true>), dim3(gc), dim3(256), 0, st, dA, n, d, K1, dS, dO, seed, thr, dscale);
    else hipLaunchKernelGGL((lit_compact<false>), dim3(gc), dim3(256), 0, st, dA, n, d, K1, dS, dO, seed, thr, dscale);
    if (int rc = pull2_fill_entries(md, pos, m_pos, neg, m_neg, plan, st)) return rc;
    if (int rc = cv_rowsum_launch(md, gE, gRp ? gRp : gR, plan, dS, dPb, dO, st)) return rc;
    if (int rc = pull_finish_launch(md, gE, gRp ? gRp : gR, plan, st)) return rc;
    if (gRp) hipLaunchKernelGGL(lit_fold_rel, dim3((int)std::min<int64_t>(cdiv((int64_t)md->num_relations * d, 256), 1024)), dim3(256), 0, st, gRp, hake ? nullptr : gR, md->num_relations, d);
  }
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

}  // namespace kge
