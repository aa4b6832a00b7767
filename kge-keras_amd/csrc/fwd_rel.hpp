// The relation-grouped forward of the training step (fwd_rel_kernel), shared by pull2.hip (its own launch) and step1.hip (the
// relation role of the one-launch step).  Reference: EmbeddingModel.call x2 + loss (KGEkeras/models.py:134-222; utils.py:288-298).
#pragma once
#include "dropout.hpp"
#include "kge_internal.hpp"
#include "pull.hpp"

namespace kge {

__device__ __forceinline__ float4 f4(float x) { return make_float4(x, x, x, x); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) { return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w)); }
__device__ __forceinline__ float sum4(float4 a) { return (a.x + a.y) + (a.z + a.w); }
__device__ __forceinline__ float4 abs4(float4 a) { return make_float4(fabsf(a.x), fabsf(a.y), fabsf(a.z), fabsf(a.w)); }
__device__ __forceinline__ float sg(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
__device__ __forceinline__ float4 sgn4(float4 a) { return make_float4(sg(a.x), sg(a.y), sg(a.z), sg(a.w)); }
__device__ __forceinline__ float4 phase_abs4(float4 a) { return make_float4(phase_abs(a.x), phase_abs(a.y), phase_abs(a.z), phase_abs(a.w)); }
__device__ __forceinline__ float4 phase_dterm4(float4 a) { return make_float4(phase_dterm(a.x), phase_dterm(a.y), phase_dterm(a.z), phase_dterm(a.w)); }
__device__ __forceinline__ float4 shx4(float4 v, int off) {
  return make_float4(__shfl_xor(v.x, off, 64), __shfl_xor(v.y, off, 64), __shfl_xor(v.z, off, 64), __shfl_xor(v.w, off, 64));
}

struct FwdRelArgs {
  const float* E; const float* Rel;
  int64_t m_pos, m_neg;
  int d_e, nchunk;
  float gamma, modulus, margin;
  const int32_t* rcnt; const int32_t* roff; const int32_t* rtotal; const int2* rumap; const int4* rentries;
  float* coef; float* scores; float* rslab;
  uint64_t seed; uint32_t thr; float dscale;
  int rel_lo = 0, rel_hi = 0x7fffffff;  // relation window of an owner-computes step (kge_train_step_swap_window): units of other relations are skipped
  // chained steps (StepChain, pull.hpp): non-zero = the unit's partial row is stored write-through and the wave waits for the
  // acknowledgement -- a closer workgroup of the same launch sums the rows
  uint32_t chain_tok = 0;
};

__device__ __forceinline__ bool rel_in_window(const FwdRelArgs& a, int unit) {
  if (a.rel_lo == 0 && a.rel_hi == 0x7fffffff) return true;
  const int p = a.rumap[unit].x;
  return p >= a.rel_lo && p < a.rel_hi;
}

// One wave per (relation, chunk).  64/G sub-groups of G lanes take different triples; lane l of a sub-group owns float4
// chunks l, l+G, ... (NV of them) of the row (of each half row for ComplEx).  Adds the unit's loss terms to lsum (lane 0 of every
// sub-group), writes the unit's partial relation-gradient row to rslab[unit].  Returns the relation; *len_out = rows of the unit, *c_out = rows of the relation.
template <int MODEL, int NORM, int LOSS, int NV, bool DROP, int G = 16>
__device__ __forceinline__ int fwd_rel_unit(const FwdRelArgs& a, int unit, int R, float& lsum, int* len_out, int* c_out) {
  constexpr int NA = MODEL == KGE_COMPLEX ? 2 * NV : NV;
  const float* __restrict__ E = a.E;
  const float* __restrict__ Rel = a.Rel;
  const int d_e = a.d_e, nchunk = a.nchunk;
  const float gamma = a.gamma, modulus = a.modulus, margin = a.margin, dscale = a.dscale;
  const uint64_t seed = a.seed;
  const uint32_t thr = a.thr;
  const int2* __restrict__ rumap = a.rumap;
  const int32_t* __restrict__ roff = a.roff;
  const int4* __restrict__ rentries = a.rentries;
  float* __restrict__ coef = a.coef;
  float* __restrict__ scores = a.scores;
  float* __restrict__ rslab = a.rslab;
  const int lane64 = threadIdx.x & 63, lane = lane64 & (G - 1), sub = lane64 / G;
  const int h = d_e >> 1, hc = h / 4;
  const float inv_pos = 1.f / (float)a.m_pos, inv_neg = 1.f / (float)a.m_neg;
  (void)hc; (void)seed; (void)thr; (void)dscale;
  const int2 pc = rumap[unit];
  const int p = pc.x;
  const int c = a.rcnt[p];
  const int chunk = rel_chunk_rows(c);
  const int start = roff[p] + pc.y * chunk;
  const int len = min(chunk, c - pc.y * chunk);
  const float* rp = Rel + (int64_t)p * d_e;
  float4 pv[NA], acc[NA];
  int4 e_nx = make_int4(0, 0, 0, 0);
  if (sub < len) e_nx = rentries[start + sub];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int ch = lane + j * G;
    pv[j] = ch < nchunk ? *reinterpret_cast<const float4*>(rp + ch * 4) : f4(0.f);
    acc[j] = f4(0.f);
    if (MODEL == KGE_COMPLEX) { pv[NV + j] = ch < nchunk ? *reinterpret_cast<const float4*>(rp + h + ch * 4) : f4(0.f); acc[NV + j] = f4(0.f); }
  }
  // the next entry is fetched while this one's rows are gathered and multiplied (entry -> rows is a dependent load per round otherwise)
  for (int i = sub; i < len; i += 64 / G) {
    const int4 e = e_nx;
    if (i + 64 / G < len) e_nx = rentries[start + i + 64 / G];
    const int64_t t = e.z;
    const bool is_pos = e.w > 0;
    const float mult = fabsf((float)e.w);  // how many input rows this one stands for
    const float* es = E + (int64_t)e.x * d_e;
    const float* eo = E + (int64_t)e.y * d_e;
    DropKey ks{0, 0}, kp{0, 0}, ko{0, 0};
    if (DROP) { ks = drop_key(seed, t, 0); kp = drop_key(seed, t, 1); ko = drop_key(seed, t, 2); }
    // tv: d score / d (relation row), element-wise (TransE: its element factor) -- for the trilinear models the score is
    // <relation row, tv>, so the gathered rows die as soon as tv is formed (two row-sized register arrays less than keeping them)
    float4 tv[NA];
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int ch = lane + j * G;
      const bool in = ch < nchunk;
      float4 s0 = in ? *reinterpret_cast<const float4*>(es + ch * 4) : f4(0.f);
      float4 o0 = in ? *reinterpret_cast<const float4*>(eo + ch * 4) : f4(0.f);
      float4 p0 = pv[j];
      float4 s1 = f4(0.f), o1 = f4(0.f), p1 = f4(0.f);
      if (MODEL == KGE_COMPLEX) {
        s1 = in ? *reinterpret_cast<const float4*>(es + h + ch * 4) : f4(0.f);
        o1 = in ? *reinterpret_cast<const float4*>(eo + h + ch * 4) : f4(0.f);
        p1 = pv[NV + j];
      }
      if (DROP) {
        s0 = mul4(s0, drop_mul4(ks, ch, thr, dscale)); o0 = mul4(o0, drop_mul4(ko, ch, thr, dscale));
        p0 = mul4(p0, drop_mul4(kp, ch, thr, dscale));
        if (MODEL == KGE_COMPLEX) {
          s1 = mul4(s1, drop_mul4(ks, hc + ch, thr, dscale)); o1 = mul4(o1, drop_mul4(ko, hc + ch, thr, dscale));
          p1 = mul4(p1, drop_mul4(kp, hc + ch, thr, dscale));
        }
      }
      if (MODEL == KGE_DISTMULT) {
        tv[j] = mul4(s0, o0);
        part += sum4(mul4(p0, tv[j]));
      } else if (MODEL == KGE_TRANSE) {
        const float4 x = sub4(add4(s0, p0), o0);
        part += NORM == 1 ? sum4(abs4(x)) : (NORM == 3 ? sum4(phase_abs4(x)) : sum4(mul4(x, x)));
        tv[j] = NORM == 1 ? sgn4(x) : (NORM == 3 ? phase_dterm4(x) : x);
      } else {  // Re<s, p, conj o> = pr (sr or + si oi) + pi (sr oi - si or)
        tv[j] = add4(mul4(s0, o0), mul4(s1, o1));
        tv[NV + j] = sub4(mul4(s0, o1), mul4(s1, o0));
        part += sum4(add4(mul4(p0, tv[j]), mul4(p1, tv[NV + j])));
      }
    }
    part = group_sum<G>(part);
    float score = part, dist = part;
    if (MODEL == KGE_TRANSE) {
      if (NORM == 2) dist = sqrtf(part);
      if (NORM == 3) dist = modulus * part;
      score = gamma > 0.f ? gamma - dist : dist;
    }
    float g, l;
    pointwise_term<LOSS>(score, is_pos, margin, (is_pos ? inv_pos : inv_neg) * mult, g, l);
    if (MODEL == KGE_TRANSE) {
      g = gamma > 0.f ? -g : g;
      if (NORM == 2) g = dist > 0.f ? g / dist : 0.f;
      if (NORM == 3) g *= 0.5f * modulus;
    }
    if (lane == 0) {
      lsum += l;
      if (coef) coef[t] = g;
      if (scores) scores[t] = score;
    }
    if (g != 0.f) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        float4 term = tv[j];
        if (DROP) term = mul4(term, drop_mul4(kp, lane + j * G, thr, dscale));
        acc[j] = fma4(f4(g), term, acc[j]);
        if (MODEL == KGE_COMPLEX) {
          float4 ti = tv[NV + j];
          if (DROP) ti = mul4(ti, drop_mul4(kp, hc + lane + j * G, thr, dscale));
          acc[NV + j] = fma4(f4(g), ti, acc[NV + j]);
        }
      }
    }
  }
  // meet the 4 sub-groups, one slab row per unit
#pragma unroll
  for (int o2 = G; o2 < 64; o2 <<= 1)
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = add4(acc[i], shx4(acc[i], o2));
  if (sub == 0) {
    float* dst = rslab + (int64_t)unit * d_e;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int ch = lane + j * G;
      if (ch < nchunk) {
        if (a.chain_tok) {
          store4_sc1(dst + ch * 4, acc[j]);
          if (MODEL == KGE_COMPLEX) store4_sc1(dst + h + ch * 4, acc[NV + j]);
        } else {
          *reinterpret_cast<float4*>(dst + ch * 4) = acc[j];
          if (MODEL == KGE_COMPLEX) *reinterpret_cast<float4*>(dst + h + ch * 4) = acc[NV + j];
        }
      }
    }
  }
  if (a.chain_tok) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the row is at the memory side before this workgroup's flag says so
  *len_out = len;
  *c_out = c;
  return p;
}

template <int MODEL, int NORM, int LOSS, int NV, bool DROP>
__global__ __launch_bounds__(256) void fwd_rel_kernel(FwdRelArgs a, float* __restrict__ loss_partials) {
  __shared__ float red[16];
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
  float lsum = 0.f;
  int len, c;
  if (unit < a.rtotal[0] && rel_in_window(a, unit)) fwd_rel_unit<MODEL, NORM, LOSS, NV, DROP>(a, unit, 0, lsum, &len, &c);
  const float b = block_sum(lsum, red);
  if (threadIdx.x == 0) loss_partials[blockIdx.x] = b;
}

}  // namespace kge
