// The self-contained entity half of a step between twin tables (see pullself.hip), as a per-wave device function shared by
// pull_self_kernel (pullself.hip) and the one-launch step (step1.hip).
#pragma once
#include "pull_kernel.hpp"

namespace kge {

struct SelfArgs {
  const float* E; const float* Rel;
  int N, d_e, nchunk;
  int row_lo, row_n;  // destination rows [row_lo, row_lo + row_n) are this launch's (an owner-computes window; the whole table: 0, N)
  float inv_pos, inv_neg, gamma, modulus, margin;
  const int32_t* off; const int32_t* poff; const int32_t* totals; const int2* emap; const int4* entries;
  float* slab; int32_t* cnt; int C;
  uint64_t seed; uint32_t thr; float dscale;  // dropout (DROP): step seed, keep threshold, 1 / (1 - rate)
  int kind;  // KGE_OPT_SGD | KGE_OPT_ADAM_TF: apply the optimizer, E_in row -> E_out row; < 0: no optimizer, the gradient row -> E_out (= gradE)
  float lr, b1, b2, eps;
  const float* lr_state;  // non-null: lr_t of this update is read from the device-resident step state (a whole step in one call) ...
  float* lr_keep;         // ... and copied here by the launch's first thread, for the closing launch (rel_tail_kernel)
  float *E_out, *mE, *vE;
  // 1: Adam's moment rows (read once, written once per step) go non-temporal; 2: also the parameter rows of entities no triple touches.
  // For tables past the Infinity Cache: there the streams only push each other out of the caches (10 M entities: 9.14 -> 8.74 ms with 1);
  // on a table that fits (the FB15k-237 bench: 47 MB with moments) they cost 5 % instead, so the host sets this by size.
  int stream_nt = 0;
};

// One wave: entity unit `unit` (a destination row, or an extra chunk of a hub row); `mine`: NACC * 16 float4 of LDS of this wave.
template <int MODEL, int NORM, int LOSS, int NV, int G = 16, bool DROP = false>
__device__ __forceinline__ void pull_self_unit(const SelfArgs& a, int unit, float4* mine) {
  constexpr int NSUB = 64 / G;
  constexpr int NACC = MODEL == KGE_COMPLEX ? 2 * NV : NV;
  constexpr int NPER = (NACC + NSUB - 1) / NSUB;
  using L = V<4>;
  const int lane64 = threadIdx.x & 63, lane = lane64 & (G - 1), sub = lane64 / G;
  const int N = a.N, d_e = a.d_e, nchunk = a.nchunk, h = d_e >> 1;
  int key, chunk;
  if (unit < a.row_n) { key = a.row_lo + unit; chunk = 0; }
  else {
    const int x = unit - a.row_n;
    if (x >= a.totals[0]) return;
    const int2 kc = a.emap[x];
    key = kc.x; chunk = kc.y;
    if ((unsigned)(key - a.row_lo) >= (unsigned)a.row_n) return;  // a hub row of another window
  }
  // (the counter of a hub row is the ticket its chunk waves count down, so the length comes from the list starts)
  const int c = (key + 1 < N ? a.off[key + 1] : a.totals[2]) - a.off[key];
  const int nch = (c + a.C - 1) / a.C;
  const int start = a.off[key] + chunk * a.C;
  const int len = min(a.C, c - chunk * a.C);
  const float* self = a.E + (int64_t)key * d_e;

  // A row no triple of the batch touches: only the optimizer's streams (Keras updates every row of the table, models.py:87-112 with the
  // default 'adam' of tests.py:86), all of the row's loads issued at once -- on a table far larger than the batch (most rows untouched)
  // this is what the launch spends its time on, and the general path below would chain three memory round trips per wave (list bounds
  // -> own row -> moments).  Same arithmetic as the general path with a zero gradient row.
  if (c == 0) {
    if (a.kind < 0) return;  // (gradient tables are all-zero between steps: a row nobody touched stays so)
    float* wout0 = a.E_out + (int64_t)key * d_e;
    float* mrow0 = a.mE + (int64_t)key * d_e;
    float* vrow0 = a.vE + (int64_t)key * d_e;
    const float lr0 = a.lr_state ? a.lr_state[0] : a.lr;
#pragma unroll
    for (int k = 0; k < NPER; ++k) {
      const int i = sub + k * NSUB;
      const int ch = lane + (i % NV) * G;
      if (i < NACC && ch < nchunk) {
        const int at = (i / NV) * h + ch * 4;
        float4 w = a.stream_nt > 1 ? L::ld_nt(self + at) : L::ld(self + at), m = L::zero(), v = L::zero();
        if (a.kind == KGE_OPT_ADAM_TF) {
          if (a.stream_nt) { m = L::ld_nt(mrow0 + at); v = L::ld_nt(vrow0 + at); }
          else { m = L::ld(mrow0 + at); v = L::ld(vrow0 + at); }
        }
        tail_upd(a.kind, lr0, a.b1, a.b2, a.eps, w, L::zero(), m, v);
        if (a.stream_nt > 1) L::st_nt(wout0 + at, w); else L::st(wout0 + at, w);
        if (a.kind == KGE_OPT_ADAM_TF) {
          if (a.stream_nt) { L::st_nt(mrow0 + at, m); L::st_nt(vrow0 + at, v); }
          else { L::st(mrow0 + at, m); L::st(vrow0 + at, v); }
        }
      }
    }
    return;
  }

  // own row -> LDS, accumulator layout: slot i*G + lane holds float4 chunk lane + (i % NV)*G of half i / NV
#pragma unroll
  for (int k = 0; k < NPER; ++k) {
    const int i = sub + k * NSUB;
    const int ch = lane + (i % NV) * G;
    if (i < NACC) mine[i * G + lane] = ch < nchunk ? L::ld(self + (i / NV) * h + ch * 4) : L::zero();
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  float4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = L::zero();

  int4 e_nx = make_int4(0, 0, 0, 0);
  if (sub < len) e_nx = a.entries[start + sub];
  for (int i = sub; i < len; i += NSUB) {
    asm volatile("" ::: "memory");  // (the own row stays in LDS: without this the compiler hoists its reads into 16 more registers)
    const int4 e = e_nx;
    if (i + NSUB < len) e_nx = a.entries[start + i + NSUB];
    const bool dest_o = (unsigned)e.x >> 31;  // 0: this row is the subject of the triple, 1: its object
    const float* ra = a.Rel + (int64_t)(e.x & 0x7fffffff) * d_e;
    const float* rb = a.E + (int64_t)e.y * d_e;
    const bool is_pos = e.z > 0;
    const float mult = fabsf((float)e.z);  // how many input rows this one stands for
    // dropout (models.py:122,144-148): the masks of the triple's three gathered rows, regenerated from (seed, batch row, role)
    // exactly as the forward does; this row's own mask multiplies both its value in the score and its gradient
    DropKey kd{0, 0}, ka{0, 0}, kb{0, 0};
    const int hc = h >> 2;
    if (DROP) { kd = drop_key(a.seed, e.w, dest_o ? 2 : 0); ka = drop_key(a.seed, e.w, 1); kb = drop_key(a.seed, e.w, dest_o ? 0 : 2); }
    float4 t[NACC];                         // d score / d (own row), element-wise
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int ch = lane + j * G;
      const bool in = ch < nchunk;
      if (MODEL == KGE_DISTMULT) {
        float4 av = in ? L::ld(ra + ch * 4) : L::zero(), bv = in ? L::ld(rb + ch * 4) : L::zero();
        if (DROP) av = av * drop_mul4(ka, ch, a.thr, a.dscale) * drop_mul4(kb, ch, a.thr, a.dscale) * drop_mul4(kd, ch, a.thr, a.dscale);
        t[j] = av * bv;
        part += hsum(mine[j * G + lane] * t[j]);
      } else if (MODEL == KGE_TRANSE) {
        float4 av = in ? L::ld(ra + ch * 4) : L::zero(), bv = in ? L::ld(rb + ch * 4) : L::zero();
        float4 m = mine[j * G + lane], md1 = L::zero();
        if (DROP) { md1 = drop_mul4(kd, ch, a.thr, a.dscale); av = av * drop_mul4(ka, ch, a.thr, a.dscale); bv = bv * drop_mul4(kb, ch, a.thr, a.dscale); m = m * md1; }
        const float4 x = dest_o ? (bv + av) - m : (m + av) - bv;  // s + p - o
        part += NORM == 1 ? hsum(vabs(x)) : (NORM == 3 ? hsum(vphase_abs(x)) : hsum(x * x));
        t[j] = NORM == 1 ? vsgn(x) : (NORM == 3 ? vphase_dterm(x) : x);
        if (DROP) t[j] = t[j] * md1;
      } else {
        float4 ar = in ? L::ld(ra + ch * 4) : L::zero(), ai = in ? L::ld(ra + h + ch * 4) : L::zero();
        float4 br = in ? L::ld(rb + ch * 4) : L::zero(), bi = in ? L::ld(rb + h + ch * 4) : L::zero();
        if (DROP) {
          ar = ar * drop_mul4(ka, ch, a.thr, a.dscale); ai = ai * drop_mul4(ka, hc + ch, a.thr, a.dscale);
          br = br * drop_mul4(kb, ch, a.thr, a.dscale); bi = bi * drop_mul4(kb, hc + ch, a.thr, a.dscale);
        }
        // subject: d/dsr = pr*or + pi*oi, d/dsi = pr*oi - pi*or;  object: d/dor = sr*pr - pi*si, d/doi = pr*si + pi*sr   (a = p, b = other)
        t[j] = dest_o ? br * ar - ai * bi : ar * br + ai * bi;
        t[NV + j] = dest_o ? ar * bi + ai * br : ar * bi - ai * br;
        if (DROP) { t[j] = t[j] * drop_mul4(kd, ch, a.thr, a.dscale); t[NV + j] = t[NV + j] * drop_mul4(kd, hc + ch, a.thr, a.dscale); }
        part += hsum(mine[j * G + lane] * t[j] + mine[(NV + j) * G + lane] * t[NV + j]);  // trilinear: score = <own row, d score / d own row>
      }
    }
    part = group_sum<G>(part);
    float score = part, dist = part;
    if (MODEL == KGE_TRANSE) {
      if (NORM == 2) dist = sqrtf(part);
      if (NORM == 3) dist = a.modulus * part;
      score = a.gamma > 0.f ? a.gamma - dist : dist;
    }
    float g, l;
    pointwise_term<LOSS>(score, is_pos, a.margin, (is_pos ? a.inv_pos : a.inv_neg) * mult, g, l);
    if (MODEL == KGE_TRANSE) {
      g = a.gamma > 0.f ? -g : g;
      if (NORM == 2) g = dist > 0.f ? g / dist : 0.f;
      if (NORM == 3) g *= 0.5f * a.modulus;
      if (dest_o) g = -g;  // d x / d o = -1
    }
    if (g != 0.f) {
#pragma unroll
      for (int i2 = 0; i2 < NACC; ++i2) acc[i2] = acc[i2] + g * t[i2];
    }
  }
  // meet the sub-groups
#pragma unroll
  for (int o2 = G; o2 < 64; o2 <<= 1)
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = acc[i] + shx(acc[i], o2);

  if (nch > 1) {
    // hub row: this chunk's partial row goes to the slab; the chunk wave that brings the row's counter to zero sums the slab rows in
    // chunk order (fixed order) and carries on (release / acquire by the fences around the ticket)
    float* part = a.slab + (int64_t)(a.poff[key] + chunk) * d_e;
    if (sub == 0) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int ch = lane + j * G;
        if (ch < nchunk) {
          L::st(part + ch * 4, acc[j]);
          if (MODEL == KGE_COMPLEX) L::st(part + h + ch * 4, acc[NV + j]);
        }
      }
    }
    __threadfence();
    int old = 0;
    if (lane64 == 0) old = atomicSub(a.cnt + key, len);
    old = __shfl(old, 0, 64);
    if (old != len) return;
    __threadfence();
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = L::zero();
    const float* src = a.slab + (int64_t)a.poff[key] * d_e;
    for (int jc = 0; jc < nch; ++jc) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int ch = lane + j * G;
        if (ch < nchunk) {
          acc[j] = acc[j] + L::ld(src + (int64_t)jc * d_e + ch * 4);
          if (MODEL == KGE_COMPLEX) acc[NV + j] = acc[NV + j] + L::ld(src + (int64_t)jc * d_e + h + ch * 4);
        }
      }
    }
  } else if (lane64 == 0 && c) {
    a.cnt[key] = 0;
  }
  // optimizer (or, kind < 0: the gradient row itself into the dense accumulator E_out): sub-group s takes the accumulators s, s + NSUB, ...
  float* wout = a.E_out + (int64_t)key * d_e;
  float* mrow = a.mE + (int64_t)key * d_e;
  float* vrow = a.vE + (int64_t)key * d_e;
  if (a.kind < 0 && c == 0) return;  // (gradient tables are all-zero between steps: a row nobody touched stays so)
#pragma unroll
  for (int k = 0; k < NPER; ++k) {
    float4 g = L::zero();
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      if (i % NSUB == sub && i / NSUB == k) g = acc[i];
    const int i = sub + k * NSUB;
    const int ch = lane + (i % NV) * G;
    if (i < NACC && ch < nchunk) {
      const int at = (i / NV) * h + ch * 4;
      if (a.kind < 0) { L::st(wout + at, L::ld(wout + at) + g); continue; }  // (accumulates, like every gradient path of the library)
      float4 w = mine[i * G + lane], m = L::zero(), v = L::zero();
      if (a.kind == KGE_OPT_ADAM_TF) {
        if (a.stream_nt) { m = L::ld_nt(mrow + at); v = L::ld_nt(vrow + at); }
        else { m = L::ld(mrow + at); v = L::ld(vrow + at); }
      }
      tail_upd(a.kind, a.lr_state ? a.lr_state[0] : a.lr, a.b1, a.b2, a.eps, w, g, m, v);
      L::st(wout + at, w);
      if (a.kind == KGE_OPT_ADAM_TF) {
        if (a.stream_nt) { L::st_nt(mrow + at, m); L::st_nt(vrow + at, v); }
        else { L::st(mrow + at, m); L::st(vrow + at, v); }
      }
    }
  }
}

template <int MODEL, int NORM, int LOSS, int NV, int G = 16, bool DROP = false>
__global__ __launch_bounds__(256) void pull_self_kernel(SelfArgs a) {
  constexpr int NACC = MODEL == KGE_COMPLEX ? 2 * NV : NV;
  __shared__ float4 selfrow[4][NACC * G];
  const int wave = threadIdx.x >> 6;
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.lr_keep) a.lr_keep[0] = a.lr_state[0];
  pull_self_unit<MODEL, NORM, LOSS, NV, G, DROP>(a, blockIdx.x * 4 + wave, selfrow[wave]);
}

}  // namespace kge
