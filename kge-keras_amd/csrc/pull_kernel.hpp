// pull_kernel, its launcher and the element helpers of the gather-reduce kernels; shared by pull.hip and the kernels of the step
// between twin tables (pull_self.hpp, step1.hip).
#pragma once
#include "kge_internal.hpp"
#include "pull.hpp"
#include "dropout.hpp"

namespace kge {

template <int VEC> struct V;
template <> struct V<4> {
  using T = float4;
  static __device__ __forceinline__ T ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ void st(float* p, T v) { *reinterpret_cast<float4*>(p) = v; }
  static __device__ __forceinline__ T zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
  // Non-temporal forms for streams with no reuse inside a step, on tables past the caches (pull_self.hpp: SelfArgs.stream_nt)
  typedef float v4f_ __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ T ld_nt(const float* p) {
    const v4f_ x = __builtin_nontemporal_load(reinterpret_cast<const v4f_*>(p));
    return make_float4(x.x, x.y, x.z, x.w);
  }
  static __device__ __forceinline__ void st_nt(float* p, T v) { __builtin_nontemporal_store(v4f_{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f_*>(p)); }
};
template <> struct V<1> {
  using T = float;
  static __device__ __forceinline__ T ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, T v) { *p = v; }
  static __device__ __forceinline__ T zero() { return 0.f; }
};

template <int VEC> struct DM;
template <> struct DM<4> { static __device__ __forceinline__ float4 mul(DropKey k, int c, uint32_t thr, float sc) { return drop_mul4(k, (uint32_t)c, thr, sc); } };
template <> struct DM<1> { static __device__ __forceinline__ float mul(DropKey k, int c, uint32_t thr, float sc) { return drop_mul(k, (uint32_t)c, thr, sc); } };

#define KGE_F4(expr_x, expr_y, expr_z, expr_w) make_float4(expr_x, expr_y, expr_z, expr_w)
__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return KGE_F4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return KGE_F4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator*(float4 a, float4 b) { return KGE_F4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 operator*(float a, float4 b) { return KGE_F4(a * b.x, a * b.y, a * b.z, a * b.w); }
__device__ __forceinline__ float hsum(float a) { return a; }
__device__ __forceinline__ float hsum(float4 a) { return (a.x + a.y) + (a.z + a.w); }
__device__ __forceinline__ float vabs(float a) { return fabsf(a); }
__device__ __forceinline__ float4 vabs(float4 a) { return KGE_F4(fabsf(a.x), fabsf(a.y), fabsf(a.z), fabsf(a.w)); }
__device__ __forceinline__ float sgn1(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
__device__ __forceinline__ float vsgn(float a) { return sgn1(a); }
__device__ __forceinline__ float4 vsgn(float4 a) { return KGE_F4(sgn1(a.x), sgn1(a.y), sgn1(a.z), sgn1(a.w)); }
__device__ __forceinline__ float vphase_abs(float a) { return phase_abs(a); }
__device__ __forceinline__ float4 vphase_abs(float4 a) { return KGE_F4(phase_abs(a.x), phase_abs(a.y), phase_abs(a.z), phase_abs(a.w)); }
__device__ __forceinline__ float vphase_dterm(float a) { return phase_dterm(a); }
__device__ __forceinline__ float4 vphase_dterm(float4 a) { return KGE_F4(phase_dterm(a.x), phase_dterm(a.y), phase_dterm(a.z), phase_dterm(a.w)); }
__device__ __forceinline__ float shx(float v, int off) { return __shfl_xor(v, off, 64); }
__device__ __forceinline__ float4 shx(float4 v, int off) {
  return KGE_F4(__shfl_xor(v.x, off, 64), __shfl_xor(v.y, off, 64), __shfl_xor(v.z, off, 64), __shfl_xor(v.w, off, 64));
}

// ------------------------------------------------------------------------------------------------ pull
// One wave per unit.  Lane l of a G-lane sub-group owns chunks l, l+G, ... (NV of them) of the row (of each HALF row for
// ComplEx); the 64/G sub-groups of the wave take different list entries and meet in a shuffle reduction at the end.
// Optimizer update of one float4 / float of a row (the expressions of train.hip's opt_sweep, gscale = 1, no L3 term).
// Every fused multiply-add is spelled out and the compiler's own contraction is off: left to itself it fuses different products in the
// copies of this function inlined into different kernels (rel_tail_kernel, the closers of a chained step, the flush: one ulp apart
// on a few elements), and the rows of a step must not depend on which launch happens to close them.
__device__ __forceinline__ void tail_upd(int kind, float lr, float b1, float b2, float eps, float& w, float g, float& m, float& v) {
#pragma clang fp contract(off)
  if (kind == KGE_OPT_ADAM_TF) {
    m = __fmaf_rn(b1, m, (1.f - b1) * g);
    v = __fmaf_rn(b2, v, ((1.f - b2) * g) * g);
    w = w - (lr * m) / (sqrtf(v) + eps);
  } else {
    w = __fmaf_rn(-lr, g, w);
  }
}
__device__ __forceinline__ void tail_upd(int kind, float lr, float b1, float b2, float eps, float4& w, float4 g, float4& m, float4& v) {
  tail_upd(kind, lr, b1, b2, eps, w.x, g.x, m.x, v.x);
  tail_upd(kind, lr, b1, b2, eps, w.y, g.y, m.y, v.y);
  tail_upd(kind, lr, b1, b2, eps, w.z, g.z, m.z, v.z);
  tail_upd(kind, lr, b1, b2, eps, w.w, g.w, m.w, v.w);
}

// StepTail (the step between twin tables): one workgroup per relation row.  Wave w sums the row's partial rows of units [w*per, (w+1)*per) with four loads in
// flight, the four sums meet in LDS in wave order (fixed order: run-to-run deterministic), wave 0 applies the optimizer.
__device__ __forceinline__ void tail_relation_row(const StepTail& t, const float* __restrict__ Rel, int row) {
  __shared__ float4 part[4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int d4 = t.d_r >> 2;
  const int c = (row + 1 < t.R ? t.roff[row + 1] : t.rtotal[1]) - t.roff[row];
  const int nch = rel_units(c);
  const int per = (nch + 3) >> 2;
  const int j0 = min(wave * per, nch), j1 = min(j0 + per, nch);
  const float lr = t.lr_slot ? t.lr_slot[0] : t.lr_t;
  if (threadIdx.x == 0) t.rcnt[row] = 0;
  for (int q0 = 0; q0 < d4; q0 += 64) {
    const int q = q0 + lane;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < d4) {
      const float4* src = reinterpret_cast<const float4*>(t.rslab) + (int64_t)t.ruoff[row] * d4 + q;
      int j = j0;
      for (; j + 4 <= j1; j += 4) {
        const float4 a = src[(int64_t)j * d4], b = src[(int64_t)(j + 1) * d4], c2 = src[(int64_t)(j + 2) * d4], d = src[(int64_t)(j + 3) * d4];
        s = (((s + a) + b) + c2) + d;
      }
      for (; j < j1; ++j) s = s + src[(int64_t)j * d4];
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && q < d4) {
      const float4 g = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
      const int64_t at = (int64_t)row * t.d_r + q * 4;
      float4 w = *reinterpret_cast<const float4*>(Rel + at), m = make_float4(0.f, 0.f, 0.f, 0.f), v = m;
      if (t.kind == KGE_OPT_ADAM_TF) { m = *reinterpret_cast<const float4*>(t.mR + at); v = *reinterpret_cast<const float4*>(t.vR + at); }
      tail_upd(t.kind, lr, t.b1, t.b2, t.eps, w, g, m, v);
      *reinterpret_cast<float4*>(t.Rel_out + at) = w;
      if (t.kind == KGE_OPT_ADAM_TF) { *reinterpret_cast<float4*>(t.mR + at) = m; *reinterpret_cast<float4*>(t.vR + at) = v; }
    }
    __syncthreads();
  }
}

template <int MODEL, int NORM, int G, int NV, int VEC, bool DROP>
__global__ __launch_bounds__(256) void pull_kernel(const float* __restrict__ E, const float* __restrict__ Rel,
                                                    float* __restrict__ gE, float* __restrict__ gR, int N, int NK, int d_e,
                                                    int nchunk, const int32_t* __restrict__ cnt, const int32_t* __restrict__ off,
                                                    const int32_t* __restrict__ eoff, const int32_t* __restrict__ poff,
                                                    const int32_t* __restrict__ totals, const int2* __restrict__ emap,
                                                    const int4* __restrict__ entries, float* __restrict__ slab, int C,
                                                    uint64_t seed, uint32_t thr, float dscale, const float* __restrict__ coef_of_row, LossFinish lf) {
  if (lf.partials && blockIdx.x == gridDim.x - 1) {  // (launched with one workgroup more than the units need)
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < lf.n; i += 256) s += lf.partials[i];
    const float b = block_sum(s, red);
    if (threadIdx.x == 0) {
      lf.loss_out[0] = b;
      step_advance(lf.adv, b);
    }
    return;
  }
  using T = typename V<VEC>::T;
  using L = V<VEC>;
  using D = DM<VEC>;
  const int hc = (d_e >> 1) / VEC;
  constexpr int NSUB = 64 / G;
  constexpr int NACC = MODEL == KGE_COMPLEX ? 2 * NV : NV;
  const int wave = threadIdx.x >> 6, lane64 = threadIdx.x & 63;
  const int lane = lane64 & (G - 1), sub = lane64 / G;
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
  const int h = d_e >> 1;
  const float* self = is_rel ? Rel + (int64_t)(key - N) * d_e : E + (int64_t)key * d_e;

  T acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = L::zero();
  T me[MODEL == KGE_TRANSE ? NV : 1];
  if (MODEL == KGE_TRANSE) {
#pragma unroll
    for (int i = 0; i < NV; ++i) { const int ch = lane + i * G; me[i] = ch < nchunk ? L::ld(self + ch * VEC) : L::zero(); }
  }

  // The coefficient rides in the entry (lists filled after the forward), or is looked up by row number (lists built ahead, ids
  // only).  Entry and coefficient of the NEXT round are fetched while this round's rows are gathered and multiplied: the lookup is
  // a dependent load (entry -> coefficient -> rows) that would otherwise sit in front of every round.
  int4 e_nx = make_int4(0, 0, 0, 0);
  float cf_nx = 0.f;
  if (sub < len) { e_nx = entries[start + sub]; cf_nx = coef_of_row ? coef_of_row[e_nx.w] : __int_as_float(e_nx.z); }
#pragma unroll 2
  for (int i = sub; i < len; i += NSUB) {
    const int4 e = e_nx;
    const float cf = cf_nx;
    if (i + NSUB < len) { e_nx = entries[start + i + NSUB]; cf_nx = coef_of_row ? coef_of_row[e_nx.w] : __int_as_float(e_nx.z); }
    if (cf == 0.f) continue;  // inactive hinge rows carry no gradient
    const int kind1 = is_rel ? 0 : ((unsigned)e.x >> 31);
    const float* ra = is_rel ? E + (int64_t)e.x * d_e : Rel + (int64_t)(e.x & 0x7fffffff) * d_e;
    const float* rb = E + (int64_t)e.y * d_e;
    // dropout: roles of (destination, a, b) in the triple e.w
    DropKey kd{0, 0}, ka{0, 0}, kb{0, 0};
    if (DROP) {
      const int rd = is_rel ? 1 : (kind1 ? 2 : 0), rra = is_rel ? 0 : 1, rrb = is_rel ? 2 : (kind1 ? 0 : 2);
      kd = drop_key(seed, e.w, rd); ka = drop_key(seed, e.w, rra); kb = drop_key(seed, e.w, rrb);
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int ch = lane + j * G;
      if (ch < nchunk) {
        if (MODEL == KGE_DISTMULT) {
          T a = L::ld(ra + ch * VEC), b = L::ld(rb + ch * VEC);
          if (DROP) a = a * D::mul(ka, ch, thr, dscale) * D::mul(kb, ch, thr, dscale) * D::mul(kd, ch, thr, dscale);
          acc[j] = acc[j] + cf * (a * b);
        } else if (MODEL == KGE_TRANSE) {
          // x = s + p - o with the destination row in its own role
          T a = L::ld(ra + ch * VEC), b = L::ld(rb + ch * VEC), m = me[j];
          T md1;
          if (DROP) { md1 = D::mul(kd, ch, thr, dscale); a = a * D::mul(ka, ch, thr, dscale); b = b * D::mul(kb, ch, thr, dscale); m = m * md1; }
          T x;
          float sg = cf;
          if (is_rel) x = a + m - b;              // dest p: a = s, b = o
          else if (!kind1) x = m + a - b;         // dest s: a = p, b = o
          else { x = b + a - m; sg = -cf; }        // dest o: a = p, b = s ; d/do = -d/dx
          T term = sg * (NORM == 1 ? vsgn(x) : (NORM == 3 ? vphase_dterm(x) : x));
          if (DROP) term = term * md1;
          acc[j] = acc[j] + term;
        } else {  // ComplEx: halves [real | imag]
          T ar = L::ld(ra + ch * VEC), ai = L::ld(ra + h + ch * VEC);
          T br = L::ld(rb + ch * VEC), bi = L::ld(rb + h + ch * VEC);
          if (DROP) {
            ar = ar * D::mul(ka, ch, thr, dscale); ai = ai * D::mul(ka, hc + ch, thr, dscale);
            br = br * D::mul(kb, ch, thr, dscale); bi = bi * D::mul(kb, hc + ch, thr, dscale);
          }
          T gr, gi;
          if (is_rel) { gr = ar * br + ai * bi; gi = ar * bi - ai * br; }            // a = s, b = o
          else if (!kind1) { gr = ar * br + ai * bi; gi = ar * bi - ai * br; }       // a = p, b = o : pr*or+pi*oi | pr*oi-pi*or
          else { gr = br * ar - ai * bi; gi = ar * bi + ai * br; }                   // a = p, b = s : sr*pr-pi*si | pr*si+pi*sr
          if (DROP) { gr = gr * D::mul(kd, ch, thr, dscale); gi = gi * D::mul(kd, hc + ch, thr, dscale); }
          acc[j] = acc[j] + cf * gr;
          acc[NV + j] = acc[NV + j] + cf * gi;
        }
      }
    }
  }
  // meet the sub-groups
#pragma unroll
  for (int o2 = G; o2 < 64; o2 <<= 1)
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = acc[i] + shx(acc[i], o2);
  if (sub != 0) return;
  float* dst;
  bool add;
  if (nch == 1) { dst = (is_rel ? gR + (int64_t)(key - N) * d_e : gE + (int64_t)key * d_e); add = true; }
  else { dst = slab + (int64_t)(poff[key] + chunk) * d_e; add = false; }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int ch = lane + j * G;
    if (ch < nchunk) {
      if (add) {
        L::st(dst + ch * VEC, L::ld(dst + ch * VEC) + acc[j]);
        if (MODEL == KGE_COMPLEX) L::st(dst + h + ch * VEC, L::ld(dst + h + ch * VEC) + acc[NV + j]);
      } else {
        L::st(dst + ch * VEC, acc[j]);
        if (MODEL == KGE_COMPLEX) L::st(dst + h + ch * VEC, acc[NV + j]);
      }
    }
  }
}

template <int MODEL, int NORM, int VEC>
static void launch_pull(int nchunk, int grid, hipStream_t st, const float* E, const float* Rel, float* gE, float* gR,
                        const kge_model_desc* md, const PullPlan& p, uint64_t seed, int nk_override = -1, const float* coef_of_row = nullptr,
                        const LossFinish* lfp = nullptr) {
  const LossFinish lf = lfp ? *lfp : LossFinish{nullptr, 0, nullptr, StepAdvance{nullptr, nullptr, 0, 0.f, 0.f, 0.f}};
  if (lfp) grid += 1;
  const int N = md->num_entities, NK = nk_override >= 0 ? nk_override : N + md->num_relations;
  const bool drop = md->dropout > 0.f;
  const uint32_t thr = drop_thr(md->dropout);
  const float dscale = 1.f / (1.f - md->dropout);
#define KGE_PULL1(GW, NVV, DR) hipLaunchKernelGGL((pull_kernel<MODEL, NORM, GW, NVV, VEC, DR>), dim3(grid), dim3(256), 0, st, E, Rel, gE, gR, N, NK, md->d_e, \
                                                  nchunk, p.cnt, p.off, p.eoff, p.poff, p.totals, p.emap, p.entries, p.slab, p.chunk, seed, thr, dscale, coef_of_row, lf)
#define KGE_PULL(GW, NVV) do { if (drop) KGE_PULL1(GW, NVV, true); else KGE_PULL1(GW, NVV, false); } while (0)
  // 16-lane sub-groups: four list entries of a row in flight per wave (measured at d=200: 114.6 us vs 149.7 us with one
  // 64-lane group on lists of ~56 entries, 49.0 vs 53.7 us on lists of ~18)
  if (nchunk <= 8) KGE_PULL(8, 1);
  else if (nchunk <= 16) KGE_PULL(16, 1);
  else if (nchunk <= 32) KGE_PULL(16, 2);
  else if (nchunk <= 64) KGE_PULL(16, 4);
  else if (nchunk <= 128) KGE_PULL(64, 2);
  else KGE_PULL(64, 4);
#undef KGE_PULL
#undef KGE_PULL1
}

struct Shape { bool v4; int nchunk; };
static Shape shape_of(const kge_model_desc* md, const void* a, const void* b, const void* c, const void* d) {
  const int width = md->model == KGE_COMPLEX ? md->d_e / 2 : md->d_e;
  const bool v4 = width % 4 == 0 && md->d_e % 4 == 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) % 16 == 0);
  return Shape{v4, v4 ? width / 4 : width};
}

}  // namespace kge
