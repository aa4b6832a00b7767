// HolE (KGEkeras/models.py:224-237): score = sum_k p_k * x_k,  x = circular cross-correlation of s and o,
//     x_k = sum_i s_i * o_{(i+k) mod d}        (the reference computes it as real(ifft(conj(fft(s)) * fft(o)))).
//
// There is no operand shared between triples, so this is VALU work, O(d^2) per triple, done directly (d is a few
// hundred; no FFT, no power-of-two constraint): one wave per triple, lane l owns outputs 4l..4l+3, both operands live
// in LDS (u = the row that is broadcast, w2 = the row that slides, stored twice back to back so the wrap-around needs no
// modulo).  Per 4 values of i a lane issues one broadcast ds_read_b128 of u, one aligned ds_read_b128 of the window and
// 16 FMAs.
//
// Every HolE quantity on the path is the same primitive ccorr(u, w)_k = sum_i u_i w_{(i+k)}:
//     forward     x        = ccorr(s, o)
//     d/dp        dp       = ccorr(s, o)
//     d/ds        ds       = ccorr(p, o)
//     d/do        do       = ccorr(rev(s), p),  rev(s)_i = s_{(-i) mod d}     (circular convolution of p and s)
//     1-vs-all    tail q   = ccorr(rev(s), p)   (score(s,p,o') = <o', q>)     head q = ccorr(p, o)  (score = <s', q>)
// so the ranking sweep is the same fp32-MFMA  Q . E^T  as DistMult's once the query vectors are built.
#include "dropout.hpp"
#include "kge_internal.hpp"
#include "pull.hpp"

namespace kge {

constexpr int kHoleMaxD = 1024;

__device__ __forceinline__ int hole_dpad(int d) { return (d + 3) & ~3; }
__host__ __device__ __forceinline__ int hole_lds_floats(int d) { return ((d + 3) & ~3) + ((2 * d + 16 + 3) & ~3); }

// Stage one row into the broadcast slot (optionally reversed: u_i = row[(d - i) mod d]) with an optional dropout mask.
template <bool DROP, int NL = 64>
__device__ __forceinline__ void stage_u(float* u, const float* __restrict__ row, int d, bool reverse, int lane, DropKey key,
                                        uint32_t thr, float dscale) {
  const int dp = hole_dpad(d);
  for (int i = lane; i < dp; i += NL) {
    float v = 0.f;
    if (i < d) {
      const int src = reverse ? (i == 0 ? 0 : d - i) : i;
      v = row[src];
      if (DROP) v *= drop_mul(key, (uint32_t)src, thr, dscale);
    }
    u[i] = v;
  }
}

// Stage one row twice back to back (w2[j] = row[j mod d]) plus zero padding.
template <bool DROP, int NL = 64>
__device__ __forceinline__ void stage_w2(float* w2, const float* __restrict__ row, int d, int lane, DropKey key, uint32_t thr,
                                         float dscale) {
  for (int i = lane; i < d; i += NL) {
    float v = row[i];
    if (DROP) v *= drop_mul(key, (uint32_t)i, thr, dscale);
    w2[i] = v;
    w2[d + i] = v;
  }
  if (lane < 16) w2[2 * d + lane] = 0.f;
}

// out[j] = ccorr(u, w)_{k0 + j}, j = 0..3, k0 = 4 * lane + pass * 256 (k0 may run past d: those outputs are garbage-free
// reads of the padded buffer and are discarded by the caller).
__device__ __forceinline__ void ccorr4(const float* __restrict__ u, const float* __restrict__ w2, int d, int k0, float out[4]) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (k0 < d) {
    float4 w = *reinterpret_cast<const float4*>(w2 + k0);
    const int d4 = d & ~3;
    for (int i = 0; i < d4; i += 4) {
      const float4 uu = *reinterpret_cast<const float4*>(u + i);
      const float4 nx = *reinterpret_cast<const float4*>(w2 + k0 + i + 4);
      a0 = fmaf(uu.x, w.x, a0); a1 = fmaf(uu.x, w.y, a1); a2 = fmaf(uu.x, w.z, a2); a3 = fmaf(uu.x, w.w, a3);
      a0 = fmaf(uu.y, w.y, a0); a1 = fmaf(uu.y, w.z, a1); a2 = fmaf(uu.y, w.w, a2); a3 = fmaf(uu.y, nx.x, a3);
      a0 = fmaf(uu.z, w.z, a0); a1 = fmaf(uu.z, w.w, a1); a2 = fmaf(uu.z, nx.x, a2); a3 = fmaf(uu.z, nx.y, a3);
      a0 = fmaf(uu.w, w.w, a0); a1 = fmaf(uu.w, nx.x, a1); a2 = fmaf(uu.w, nx.y, a2); a3 = fmaf(uu.w, nx.z, a3);
      w = nx;
    }
    for (int i = d4; i < d; ++i) {  // d % 4 tail
      const float ui = u[i];
      a0 = fmaf(ui, w2[k0 + i], a0); a1 = fmaf(ui, w2[k0 + i + 1], a1);
      a2 = fmaf(ui, w2[k0 + i + 2], a2); a3 = fmaf(ui, w2[k0 + i + 3], a3);
    }
  }
  out[0] = a0; out[1] = a1; out[2] = a2; out[3] = a3;
}

// The same primitive with EIGHT outputs per lane (k0 = 8 * lane of a 32-lane half-wave, d <= 256): per 4 values of i one broadcast
// ds_read_b128 of u and one ds_read_b128 of the window feed 32 FMAs instead of 16 -- ccorr4 keeps the LDS pipe of a CU ~75 % busy (four
// SIMDs x two b128 reads per 16 FMAs), so the two halves of a wave each take a problem of their own (two triples, two list entries) and
// the FMAs, not the reads, set the pace.  Every output's sum runs over i in ascending order, as in ccorr4: the values are bit-identical.
constexpr int kHolePairD = 256;
__device__ __forceinline__ void ccorr8(const float* __restrict__ u, const float* __restrict__ w2, int d, int k0, float out[8]) {
  float a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = 0.f;
  if (k0 < d) {
    float4 w0 = *reinterpret_cast<const float4*>(w2 + k0), w1 = *reinterpret_cast<const float4*>(w2 + k0 + 4);
    const int d4 = d & ~3;
    for (int i = 0; i < d4; i += 4) {
      const float4 uu = *reinterpret_cast<const float4*>(u + i);
      const float4 nx = *reinterpret_cast<const float4*>(w2 + k0 + i + 8);
      const float e[11] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, nx.x, nx.y, nx.z};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a[j] = fmaf(uu.x, e[j], a[j]);
        a[j] = fmaf(uu.y, e[j + 1], a[j]);
        a[j] = fmaf(uu.z, e[j + 2], a[j]);
        a[j] = fmaf(uu.w, e[j + 3], a[j]);
      }
      w0 = w1; w1 = nx;
    }
    for (int i = d4; i < d; ++i) {  // d % 4 tail
      const float ui = u[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = fmaf(ui, w2[k0 + i + j], a[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) out[j] = a[j];
}


// Forward.  LOSS >= 0: pointwise loss fused; LOSS < 0: scores only (predict, pairwise two-phase path).
// Dynamic LDS: 4 waves x hole_lds_floats(d) floats.
template <int LOSS, bool DROP>
__global__ __launch_bounds__(256) void hole_fwd_kernel(const float* __restrict__ E, const float* __restrict__ Rel,
                                                        const int32_t* __restrict__ pos, int64_t m_pos,
                                                        const int32_t* __restrict__ neg, int64_t m_neg, int N, int R, int d,
                                                        float margin, float* __restrict__ coef, float* __restrict__ scores,
                                                        float* __restrict__ loss_partials, uint32_t* __restrict__ err,
                                                        uint64_t seed, uint32_t thr, float dscale) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float red[16];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* u = lds + wave * hole_lds_floats(d);
  float* w2 = u + hole_dpad(d);
  const int64_t n = m_pos + m_neg;
  const float inv_pos = 1.f / (float)(m_pos > 0 ? m_pos : 1), inv_neg = 1.f / (float)(m_neg > 0 ? m_neg : 1);
  float lsum = 0.f;
  if (d <= kHolePairD) {  // two triples per wave, one per half (ccorr8); LDS: two slices per wave
    const int hw = lane >> 5, hl = lane & 31;
    float* u2 = lds + (wave * 2 + hw) * hole_lds_floats(d);
    float* w22 = u2 + hole_dpad(d);
    for (int64_t t0 = ((int64_t)blockIdx.x * 4 + wave) * 2; t0 < n; t0 += (int64_t)gridDim.x * 8) {
      const int64_t t = t0 + hw;
      const bool live = t < n;
      const bool is_pos = t < m_pos;
      const int32_t* tr = live ? (is_pos ? pos + 3 * t : neg + 3 * (t - m_pos)) : pos;
      const int s = live ? tr[0] : 0, p = live ? tr[1] : 0, o = live ? tr[2] : 0;
      const bool ok = live && (unsigned)s < (unsigned)N && (unsigned)p < (unsigned)R && (unsigned)o < (unsigned)N;
      float part = 0.f;
      DropKey ks{0, 0}, kp{0, 0}, ko{0, 0};
      if (DROP) { ks = drop_key(seed, t, 0); kp = drop_key(seed, t, 1); ko = drop_key(seed, t, 2); }
      if (ok) {
        stage_u<DROP, 32>(u2, E + (int64_t)s * d, d, false, hl, ks, thr, dscale);
        stage_w2<DROP, 32>(w22, E + (int64_t)o * d, d, hl, ko, thr, dscale);
      }
      __builtin_amdgcn_wave_barrier();
      if (ok) {
        const float* rp = Rel + (int64_t)p * d;
        const int k0 = 8 * hl;
        float x[8];
        ccorr8(u2, w22, d, k0, x);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (k0 + j < d) {
            float pv = rp[k0 + j];
            if (DROP) pv *= drop_mul(kp, (uint32_t)(k0 + j), thr, dscale);
            part = fmaf(pv, x[j], part);
          }
      }
      __builtin_amdgcn_wave_barrier();
      const float score0 = group_sum<32>(part);
      if (hl == 0 && live) {
        float score = score0;
        if (!ok) { score = __builtin_nanf(""); atomicOr(err, 1u); }
        if (scores) scores[t] = score;
        if (LOSS >= 0) {
          float g = 0.f, l = 0.f;
          if (ok) pointwise_term<LOSS>(score, is_pos, margin, is_pos ? inv_pos : inv_neg, g, l);
          lsum += l;
          coef[t] = g;
        }
      }
    }
  } else
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < n; t += (int64_t)gridDim.x * 4) {
    const bool is_pos = t < m_pos;
    const int32_t* tr = is_pos ? pos + 3 * t : neg + 3 * (t - m_pos);
    const int s = tr[0], p = tr[1], o = tr[2];
    const bool ok = (unsigned)s < (unsigned)N && (unsigned)p < (unsigned)R && (unsigned)o < (unsigned)N;
    float part = 0.f;
    if (ok) {
      DropKey ks{0, 0}, kp{0, 0}, ko{0, 0};
      if (DROP) { ks = drop_key(seed, t, 0); kp = drop_key(seed, t, 1); ko = drop_key(seed, t, 2); }
      stage_u<DROP>(u, E + (int64_t)s * d, d, false, lane, ks, thr, dscale);
      stage_w2<DROP>(w2, E + (int64_t)o * d, d, lane, ko, thr, dscale);
      __builtin_amdgcn_wave_barrier();  // LDS is in order within a wave; keep the compiler from reordering across
      const float* rp = Rel + (int64_t)p * d;
      for (int k0 = 4 * lane; k0 < d; k0 += 256) {
        float x[4];
        ccorr4(u, w2, d, k0, x);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (k0 + j < d) {
            float pv = rp[k0 + j];
            if (DROP) pv *= drop_mul(kp, (uint32_t)(k0 + j), thr, dscale);
            part = fmaf(pv, x[j], part);
          }
      }
      __builtin_amdgcn_wave_barrier();
    }
    const float score0 = group_sum<64>(part);
    if (lane == 0) {
      float score = score0;
      if (!ok) { score = __builtin_nanf(""); atomicOr(err, 1u); }
      if (scores) scores[t] = score;
      if (LOSS >= 0) {
        float g = 0.f, l = 0.f;
        if (ok) pointwise_term<LOSS>(score, is_pos, margin, is_pos ? inv_pos : inv_neg, g, l);
        lsum += l;
        coef[t] = g;
      }
    }
  }
  if (LOSS >= 0) {
    const float b = block_sum(lsum, red);
    if (threadIdx.x == 0) loss_partials[blockIdx.x] = b;
  }
}

// Gradient gather-reduce for HolE over the same lists as pull_kernel (pull.hip): one wave per (destination row, chunk).
template <bool DROP>
__global__ __launch_bounds__(256) void hole_pull_kernel(const float* __restrict__ E, const float* __restrict__ Rel,
                                                         float* __restrict__ gE, float* __restrict__ gR, int N, int NK, int d,
                                                         const int32_t* __restrict__ cnt, const int32_t* __restrict__ off,
                                                         const int32_t* __restrict__ eoff, const int32_t* __restrict__ poff,
                                                         const int32_t* __restrict__ totals, const int2* __restrict__ emap,
                                                         const int4* __restrict__ entries, float* __restrict__ slab, int C,
                                                         uint64_t seed, uint32_t thr, float dscale) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* u = lds + wave * hole_lds_floats(d);
  float* w2 = u + hole_dpad(d);
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
  const int nch = (c + C - 1) / C;
  const int start = off[key] + chunk * C;
  const int len = min(C, c - chunk * C);
  const bool is_rel = key >= N;
  if (d <= kHolePairD) {  // two list entries per round, one per half-wave (ccorr8); the halves' sums meet at the end
    const int hw = lane >> 5, hl = lane & 31, k0 = 8 * hl;
    float* u2 = lds + (wave * 2 + hw) * hole_lds_floats(d);
    float* w22 = u2 + hole_dpad(d);
    float a8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a8[j] = 0.f;
    for (int i = 0; i < len; i += 2) {
      const bool live = i + hw < len;
      const int4 e = entries[start + (live ? i + hw : i)];
      const float cf = __int_as_float(e.z);
      const bool act = live && cf != 0.f;
      const int kind1 = is_rel ? 0 : ((unsigned)e.x >> 31);
      const float *ur, *wr;
      int ru, rw, rd;
      bool rev = false;
      if (is_rel) { ur = E + (int64_t)e.x * d; wr = E + (int64_t)e.y * d; ru = 0; rw = 2; rd = 1; }
      else if (!kind1) { ur = Rel + (int64_t)(e.x & 0x7fffffff) * d; wr = E + (int64_t)e.y * d; ru = 1; rw = 2; rd = 0; }
      else { ur = E + (int64_t)e.y * d; wr = Rel + (int64_t)(e.x & 0x7fffffff) * d; ru = 0; rw = 1; rd = 2; rev = true; }
      DropKey ku{0, 0}, kw{0, 0}, kd{0, 0};
      if (DROP) { ku = drop_key(seed, e.w, ru); kw = drop_key(seed, e.w, rw); kd = drop_key(seed, e.w, rd); }
      if (act) {
        stage_u<DROP, 32>(u2, ur, d, rev, hl, ku, thr, dscale);
        stage_w2<DROP, 32>(w22, wr, d, hl, kw, thr, dscale);
      }
      __builtin_amdgcn_wave_barrier();
      if (act) {
        float x[8];
        ccorr8(u2, w22, d, k0, x);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float v = cf * x[j];
          if (DROP && k0 + j < d) v *= drop_mul(kd, (uint32_t)(k0 + j), thr, dscale);
          a8[j] += v;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) a8[j] += __shfl_xor(a8[j], 32, 64);
    float* dst2;
    bool add2;
    if (nch == 1) { dst2 = (is_rel ? gR + (int64_t)(key - N) * d : gE + (int64_t)key * d); add2 = true; }
    else { dst2 = slab + (int64_t)(poff[key] + chunk) * d; add2 = false; }
    if (hw == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k0 + j < d) dst2[k0 + j] = add2 ? dst2[k0 + j] + a8[j] : a8[j];
    }
    return;
  }
  constexpr int NP = kHoleMaxD / 256;  // passes of 256 outputs
  float acc[NP][4];
#pragma unroll
  for (int q = 0; q < NP; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[q][j] = 0.f;
  for (int i = 0; i < len; ++i) {
    const int4 e = entries[start + i];
    const float cf = __int_as_float(e.z);
    if (cf == 0.f) continue;
    const int kind1 = is_rel ? 0 : ((unsigned)e.x >> 31);
    // operands of ccorr(u, w) and their roles in triple e.w
    const float *ur, *wr;
    int ru, rw, rd;
    bool rev = false;
    if (is_rel) { ur = E + (int64_t)e.x * d; wr = E + (int64_t)e.y * d; ru = 0; rw = 2; rd = 1; }                                  // dp = ccorr(s, o)
    else if (!kind1) { ur = Rel + (int64_t)(e.x & 0x7fffffff) * d; wr = E + (int64_t)e.y * d; ru = 1; rw = 2; rd = 0; }           // ds = ccorr(p, o)
    else { ur = E + (int64_t)e.y * d; wr = Rel + (int64_t)(e.x & 0x7fffffff) * d; ru = 0; rw = 1; rd = 2; rev = true; }           // do = ccorr(rev(s), p)
    DropKey ku{0, 0}, kw{0, 0}, kd{0, 0};
    if (DROP) { ku = drop_key(seed, e.w, ru); kw = drop_key(seed, e.w, rw); kd = drop_key(seed, e.w, rd); }
    stage_u<DROP>(u, ur, d, rev, lane, ku, thr, dscale);
    stage_w2<DROP>(w2, wr, d, lane, kw, thr, dscale);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int k0 = 4 * lane + q * 256;
      if (k0 < d) {
        float x[4];
        ccorr4(u, w2, d, k0, x);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = cf * x[j];
          if (DROP && k0 + j < d) v *= drop_mul(kd, (uint32_t)(k0 + j), thr, dscale);
          acc[q][j] += v;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  float* dst;
  bool add;
  if (nch == 1) { dst = (is_rel ? gR + (int64_t)(key - N) * d : gE + (int64_t)key * d); add = true; }
  else { dst = slab + (int64_t)(poff[key] + chunk) * d; add = false; }
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int k0 = 4 * lane + q * 256;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (k0 + j < d) dst[k0 + j] = add ? dst[k0 + j] + acc[q][j] : acc[q][j];
  }
}

// Query vectors of the 1-vs-all sweep (zero padded to Kpad): tail q = ccorr(rev(s), p), head q = ccorr(p, o).
__global__ __launch_bounds__(256) void hole_build_q_kernel(const float* __restrict__ E, const float* __restrict__ Rel,
                                                            const int32_t* __restrict__ triples, int64_t nq, int side, int N,
                                                            int R, int d, int Kpad, float* __restrict__ Qv,
                                                            uint32_t* __restrict__ err) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* u = lds + wave * hole_lds_floats(d);
  float* w2 = u + hole_dpad(d);
  const int64_t q = (int64_t)blockIdx.x * 4 + wave;
  if (q >= nq) return;
  const int s = triples[3 * q + 0], p = triples[3 * q + 1], o = triples[3 * q + 2];
  const bool ok = (unsigned)s < (unsigned)N && (unsigned)p < (unsigned)R && (unsigned)o < (unsigned)N;
  float* out = Qv + q * Kpad;
  if (!ok) {
    if (lane == 0) atomicOr(err, 1u);
    for (int k = lane; k < Kpad; k += 64) out[k] = 0.f;
    return;
  }
  const DropKey nokey{0, 0};
  if (side == 0) {
    stage_u<false>(u, E + (int64_t)s * d, d, true, lane, nokey, 0u, 1.f);
    stage_w2<false>(w2, Rel + (int64_t)p * d, d, lane, nokey, 0u, 1.f);
  } else {
    stage_u<false>(u, Rel + (int64_t)p * d, d, false, lane, nokey, 0u, 1.f);
    stage_w2<false>(w2, E + (int64_t)o * d, d, lane, nokey, 0u, 1.f);
  }
  __builtin_amdgcn_wave_barrier();
  for (int k0 = 4 * lane; k0 < Kpad; k0 += 256) {
    float x[4];
    ccorr4(u, w2, d, k0, x);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (k0 + j < Kpad) out[k0 + j] = k0 + j < d ? x[j] : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------------
static int hole_check(const kge_model_desc* md, const char* who) {
  if (md->d_e > kHoleMaxD) return fail(KGE_EUNSUPPORTED, "%s: HolE rows wider than %d floats are not built", who, kHoleMaxD);
  return KGE_OK;
}

static size_t hole_smem(int d) { return (size_t)(d <= kHolePairD ? 8 : 4) * hole_lds_floats(d) * sizeof(float); }  // (two slices per wave for the paired path)

int hole_forward(const kge_model_desc* md, int loss, float margin, const float* E, const float* Rel, const int32_t* pos,
                 int64_t m_pos, const int32_t* neg, int64_t m_neg, float* coef, float* scores, float* loss_partials,
                 int* n_partials, uint32_t* err, uint64_t seed, bool training, hipStream_t st) {
  if (int rc = hole_check(md, "HolE forward")) return rc;
  const int64_t n = m_pos + m_neg;
  const int grid = (int)std::min<int64_t>(cdiv(n, 4), 4096);
  if (n_partials) *n_partials = grid;
  const bool drop = training && md->dropout > 0.f;
  const uint32_t thr = drop_thr(md->dropout);
  const float dscale = 1.f / (1.f - md->dropout);
  const int d = md->d_e;
  ProfScope prof(training ? KGE_PROF_FWD : KGE_PROF_SCORE, st);
#define KGE_HF(LOSSV, DR) hipLaunchKernelGGL((hole_fwd_kernel<LOSSV, DR>), dim3(grid), dim3(256), hole_smem(d), st, E, Rel, pos, m_pos, neg, m_neg, \
                                             md->num_entities, md->num_relations, d, margin, coef, scores, loss_partials, err, seed, thr, dscale)
#define KGE_HF2(LOSSV) do { if (drop) KGE_HF(LOSSV, true); else KGE_HF(LOSSV, false); } while (0)
  switch (loss) {
    case KGE_POINTWIZE_HINGE: KGE_HF2(KGE_POINTWIZE_HINGE); break;
    case KGE_POINTWIZE_LOGISTIC: KGE_HF2(KGE_POINTWIZE_LOGISTIC); break;
    case KGE_POINTWIZE_SQUARE: KGE_HF2(KGE_POINTWIZE_SQUARE); break;
    case KGE_POINTWIZE_CROSS_ENTROPY: KGE_HF2(KGE_POINTWIZE_CROSS_ENTROPY); break;
    default: KGE_HF2(-1); break;
  }
#undef KGE_HF2
#undef KGE_HF
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

int hole_backward(const kge_model_desc* md, const float* E, const float* Rel, float* gE, float* gR, const PullPlan& p,
                  uint64_t seed, hipStream_t st) {
  if (int rc = hole_check(md, "HolE backward")) return rc;
  const int N = md->num_entities, NK = N + md->num_relations, d = md->d_e;
  const int units = NK + p.max_extra;
  const int grid = (int)cdiv(units, 4);
  const bool drop = md->dropout > 0.f;
  const uint32_t thr = drop_thr(md->dropout);
  const float dscale = 1.f / (1.f - md->dropout);
  ProfScope prof(KGE_PROF_TRAIN_GRADS, st);
  if (drop)
    hipLaunchKernelGGL((hole_pull_kernel<true>), dim3(grid), dim3(256), hole_smem(d), st, E, Rel, gE, gR, N, NK, d, p.cnt, p.off, p.eoff,
                       p.poff, p.totals, p.emap, p.entries, p.slab, p.chunk, seed, thr, dscale);
  else
    hipLaunchKernelGGL((hole_pull_kernel<false>), dim3(grid), dim3(256), hole_smem(d), st, E, Rel, gE, gR, N, NK, d, p.cnt, p.off, p.eoff,
                       p.poff, p.totals, p.emap, p.entries, p.slab, p.chunk, seed, thr, dscale);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

int hole_build_q(const kge_model_desc* md, const float* E, const float* Rel, const int32_t* triples, int64_t nq, int side, int Kpad,
                 float* Qv, uint32_t* err, hipStream_t st) {
  if (int rc = hole_check(md, "HolE ranking")) return rc;
  hipLaunchKernelGGL(hole_build_q_kernel, dim3((int)cdiv(nq, 4)), dim3(256), hole_smem(md->d_e), st, E, Rel, triples, nq, side,
                     md->num_entities, md->num_relations, md->d_e, Kpad, Qv, err);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

}  // namespace kge
