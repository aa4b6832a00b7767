// Pull gradient path, variant 2 (pointwise losses; DistMult / TransE / ComplEx with rows of <= 256 floats per half):
// the forward gather walks the triples GROUPED BY RELATION, so the relation row is read once per work unit and the
// relation gradient  d/dp = g * f(s, o)  is accumulated in registers from the very rows the score needed -- the
// backward pull then only has the entity rows left (2 list entries and 2 entity-row reads per triple instead of 3 and 4).
//
//   main stream                                              side stream (joins before ent_fill)
//   rel_hist_kernel   LDS histogram of p, rank per triple     ent_hist_kernel  returning int atomics on s and o
//   rel_plan_kernel   offsets, unit map (relation, chunk)     scan_kernel / emap_kernel over the N entity keys
//   rel_fill_kernel   relation lists {s, o, t}
//   fwd_rel_kernel    one wave per (relation, chunk of 64): score, loss, coef[t], partial d/dp row -> slab
//   ent_fill_kernel   entity lists {p | role, other entity, coef, t}
//   pull_kernel       (pull.hip) over entity keys only
//   rel_finish_kernel d/dp rows = sum of the relation's slab rows (chunk order); pull_finish for hub entities
#include "dropout.hpp"
#include "kge_internal.hpp"
#include "pull.hpp"

namespace kge {

constexpr int kRelMax = 4096;   // relations handled by the LDS histogram / single-workgroup plan
constexpr int kRelChunk = 64;   // triples per forward work unit
constexpr int kRelBlockTriples = 1024;

struct Plan2 {
  // relation side
  int32_t* rcnt;    // [R]
  int32_t* roff;    // [R]  list start
  int32_t* ruoff;   // [R]  first unit of the relation
  int32_t* rtotal;  // [4]  {units}
  int2* rumap;      // [max_units] (relation, chunk)
  int32_t* rrank;   // [n]  rank of triple t in its relation list (-1: invalid ids)
  int4* rentries;   // [n]  {s, o, t, 0}
  float* rslab;     // [max_units][d]
  int max_units;
  // entity side (same meaning as PullPlan, keys = entities only)
  PullPlan ent;
  int32_t* erank;   // [2][n]
};

__global__ __launch_bounds__(256) void rel_hist_kernel(const int32_t* __restrict__ pos, int64_t m_pos,
                                                        const int32_t* __restrict__ neg, int64_t m_neg, int N, int R,
                                                        int32_t* __restrict__ rcnt, int32_t* __restrict__ rrank) {
  __shared__ int lrel[kRelMax];
  for (int i = threadIdx.x; i < R; i += 256) lrel[i] = 0;
  __syncthreads();
  const int64_t n = m_pos + m_neg;
  const int64_t t0 = (int64_t)blockIdx.x * kRelBlockTriples;
  constexpr int PER = kRelBlockTriples / 256;
  int rp[PER], pp[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int64_t t = t0 + threadIdx.x + i * 256;
    rp[i] = pp[i] = -1;
    if (t < n) {
      const int32_t* tr = t < m_pos ? pos + 3 * t : neg + 3 * (t - m_pos);
      const int s = tr[0], p = tr[1], o = tr[2];
      if ((unsigned)s < (unsigned)N && (unsigned)p < (unsigned)R && (unsigned)o < (unsigned)N) { pp[i] = p; rp[i] = atomicAdd(&lrel[p], 1); }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < R; i += 256) { const int v = lrel[i]; lrel[i] = v ? atomicAdd(&rcnt[i], v) : 0; }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int64_t t = t0 + threadIdx.x + i * 256;
    if (t < n) rrank[t] = pp[i] < 0 ? -1 : lrel[pp[i]] + rp[i];
  }
}

// One workgroup of 256 threads: exclusive scans over R <= 4096 relations (list offsets, unit offsets) and the unit map.
__global__ __launch_bounds__(256) void rel_plan_kernel(const int32_t* __restrict__ rcnt, int R, int32_t* __restrict__ roff,
                                                        int32_t* __restrict__ ruoff, int32_t* __restrict__ rtotal,
                                                        int2* __restrict__ rumap, int max_units) {
  __shared__ int wa[4], wb[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (R + 255) / 256;  // <= 16
  const int k0 = tid * per, k1 = min(R, k0 + per);
  int a = 0, b = 0;
  for (int k = k0; k < k1; ++k) { const int c = rcnt[k]; a += c; b += (c + kRelChunk - 1) / kRelChunk; }
  int ia = a, ib = b;
  for (int d = 1; d < 64; d <<= 1) {
    const int ua = __shfl_up(ia, d, 64), ub = __shfl_up(ib, d, 64);
    if (lane >= d) { ia += ua; ib += ub; }
  }
  if (lane == 63) { wa[wave] = ia; wb[wave] = ib; }
  __syncthreads();
  int ra = ia - a, rb = ib - b;
  for (int w = 0; w < wave; ++w) { ra += wa[w]; rb += wb[w]; }
  for (int k = k0; k < k1; ++k) {
    const int c = rcnt[k], nch = (c + kRelChunk - 1) / kRelChunk;
    roff[k] = ra; ruoff[k] = rb;
    for (int j = 0; j < nch; ++j) if (rb + j < max_units) rumap[rb + j] = make_int2(k, j);
    ra += c; rb += nch;
  }
  if (tid == 255) rtotal[0] = rb;
}

__global__ __launch_bounds__(256) void rel_fill_kernel(const int32_t* __restrict__ pos, int64_t m_pos,
                                                        const int32_t* __restrict__ neg, int64_t m_neg,
                                                        const int32_t* __restrict__ roff, const int32_t* __restrict__ rrank,
                                                        int4* __restrict__ rentries, float* __restrict__ coef,
                                                        float* __restrict__ scores, uint32_t* __restrict__ err) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m_pos + m_neg) return;
  const int r = rrank[t];
  if (r < 0) {  // ids outside the tables: the row takes no part in the step, scores NaN, error flag raised
    coef[t] = 0.f;
    if (scores) scores[t] = __builtin_nanf("");
    atomicOr(err, 1u);
    return;
  }
  const int32_t* tr = t < m_pos ? pos + 3 * t : neg + 3 * (t - m_pos);
  rentries[roff[tr[1]] + r] = make_int4(tr[0], tr[2], (int)t, 0);
}

__global__ __launch_bounds__(256) void ent_hist_kernel(const int32_t* __restrict__ pos, int64_t m_pos,
                                                        const int32_t* __restrict__ neg, int64_t m_neg, int N, int R,
                                                        int32_t* __restrict__ cnt, int32_t* __restrict__ erank) {
  const int64_t n = m_pos + m_neg;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int32_t* tr = t < m_pos ? pos + 3 * t : neg + 3 * (t - m_pos);
  const int s = tr[0], p = tr[1], o = tr[2];
  int rs = -1, ro = -1;
  if ((unsigned)s < (unsigned)N && (unsigned)p < (unsigned)R && (unsigned)o < (unsigned)N) {
    rs = atomicAdd(&cnt[s], 1);
    ro = atomicAdd(&cnt[o], 1);
  }
  erank[t] = rs;
  erank[n + t] = ro;
}

__global__ __launch_bounds__(256) void ent_fill_kernel(const int32_t* __restrict__ pos, int64_t m_pos,
                                                        const int32_t* __restrict__ neg, int64_t m_neg,
                                                        const float* __restrict__ coef, const int32_t* __restrict__ off,
                                                        const int32_t* __restrict__ erank, int4* __restrict__ entries) {
  const int64_t n = m_pos + m_neg;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int r0 = erank[t];
  if (r0 < 0) return;
  const int32_t* tr = t < m_pos ? pos + 3 * t : neg + 3 * (t - m_pos);
  const int s = tr[0], p = tr[1], o = tr[2];
  const int cb = __float_as_int(coef[t]);
  entries[off[s] + r0] = make_int4(p, o, cb, (int)t);
  entries[off[o] + erank[n + t]] = make_int4(p | (int)0x80000000, s, cb, (int)t);
}


__device__ __forceinline__ float4 f4(float x) { return make_float4(x, x, x, x); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) { return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w)); }
__device__ __forceinline__ float sum4(float4 a) { return (a.x + a.y) + (a.z + a.w); }
__device__ __forceinline__ float4 abs4(float4 a) { return make_float4(fabsf(a.x), fabsf(a.y), fabsf(a.z), fabsf(a.w)); }
__device__ __forceinline__ float sg(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
__device__ __forceinline__ float4 sgn4(float4 a) { return make_float4(sg(a.x), sg(a.y), sg(a.z), sg(a.w)); }
__device__ __forceinline__ float4 shx4(float4 v, int off) {
  return make_float4(__shfl_xor(v.x, off, 64), __shfl_xor(v.y, off, 64), __shfl_xor(v.z, off, 64), __shfl_xor(v.w, off, 64));
}

// One wave per (relation, chunk).  4 sub-groups of 16 lanes take different triples; lane l of a sub-group owns float4
// chunks l, l+16, ... (NV of them) of the row (of each half row for ComplEx).
template <int MODEL, int NORM, int LOSS, int NV, bool DROP>
__global__ __launch_bounds__(256) void fwd_rel_kernel(const float* __restrict__ E, const float* __restrict__ Rel, int64_t m_pos,
                                                       int64_t m_neg, int d_e, int nchunk, float gamma, float margin,
                                                       const int32_t* __restrict__ rcnt, const int32_t* __restrict__ roff,
                                                       const int32_t* __restrict__ rtotal, const int2* __restrict__ rumap,
                                                       const int4* __restrict__ rentries, float* __restrict__ coef,
                                                       float* __restrict__ scores, float* __restrict__ rslab,
                                                       float* __restrict__ loss_partials, uint64_t seed, uint32_t thr, float dscale) {
  __shared__ float red[16];
  constexpr int G = 16;
  constexpr int NA = MODEL == KGE_COMPLEX ? 2 * NV : NV;
  const int wave = threadIdx.x >> 6, lane64 = threadIdx.x & 63, lane = lane64 & (G - 1), sub = lane64 / G;
  const int unit = blockIdx.x * 4 + wave;
  const int h = d_e >> 1, hc = h / 4;
  const float inv_pos = 1.f / (float)m_pos, inv_neg = 1.f / (float)m_neg;
  float lsum = 0.f;
  if (unit < rtotal[0]) {
    const int2 pc = rumap[unit];
    const int p = pc.x;
    const int c = rcnt[p];
    const int start = roff[p] + pc.y * kRelChunk;
    const int len = min(kRelChunk, c - pc.y * kRelChunk);
    const float* rp = Rel + (int64_t)p * d_e;
    float4 pv[NA], acc[NA];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int ch = lane + j * G;
      pv[j] = ch < nchunk ? *reinterpret_cast<const float4*>(rp + ch * 4) : f4(0.f);
      acc[j] = f4(0.f);
      if (MODEL == KGE_COMPLEX) { pv[NV + j] = ch < nchunk ? *reinterpret_cast<const float4*>(rp + h + ch * 4) : f4(0.f); acc[NV + j] = f4(0.f); }
    }
    for (int i = sub; i < len; i += 4) {
      const int4 e = rentries[start + i];
      const int64_t t = e.z;
      const bool is_pos = t < m_pos;
      const float* es = E + (int64_t)e.x * d_e;
      const float* eo = E + (int64_t)e.y * d_e;
      DropKey ks{0, 0}, kp{0, 0}, ko{0, 0};
      if (DROP) { ks = drop_key(seed, t, 0); kp = drop_key(seed, t, 1); ko = drop_key(seed, t, 2); }
      float4 sv[NA], ov[NA], pm[NA];
      float part = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int ch = lane + j * G;
        const bool in = ch < nchunk;
        sv[j] = in ? *reinterpret_cast<const float4*>(es + ch * 4) : f4(0.f);
        ov[j] = in ? *reinterpret_cast<const float4*>(eo + ch * 4) : f4(0.f);
        pm[j] = pv[j];
        if (MODEL == KGE_COMPLEX) {
          sv[NV + j] = in ? *reinterpret_cast<const float4*>(es + h + ch * 4) : f4(0.f);
          ov[NV + j] = in ? *reinterpret_cast<const float4*>(eo + h + ch * 4) : f4(0.f);
          pm[NV + j] = pv[NV + j];
        }
        if (DROP) {
          sv[j] = mul4(sv[j], drop_mul4(ks, ch, thr, dscale)); ov[j] = mul4(ov[j], drop_mul4(ko, ch, thr, dscale));
          pm[j] = mul4(pm[j], drop_mul4(kp, ch, thr, dscale));
          if (MODEL == KGE_COMPLEX) {
            sv[NV + j] = mul4(sv[NV + j], drop_mul4(ks, hc + ch, thr, dscale)); ov[NV + j] = mul4(ov[NV + j], drop_mul4(ko, hc + ch, thr, dscale));
            pm[NV + j] = mul4(pm[NV + j], drop_mul4(kp, hc + ch, thr, dscale));
          }
        }
        if (MODEL == KGE_DISTMULT) part += sum4(mul4(mul4(sv[j], pm[j]), ov[j]));
        else if (MODEL == KGE_TRANSE) { const float4 x = sub4(add4(sv[j], pm[j]), ov[j]); part += NORM == 1 ? sum4(abs4(x)) : sum4(mul4(x, x)); }
        else {
          const float4 sr = sv[j], si = sv[NV + j], pr = pm[j], pi = pm[NV + j], orr = ov[j], oi = ov[NV + j];
          part += sum4(add4(add4(mul4(mul4(sr, pr), orr), mul4(mul4(pr, si), oi)), sub4(mul4(mul4(pi, sr), oi), mul4(mul4(pi, si), orr))));
        }
      }
      part = group_sum<G>(part);
      float score = part, dist = part;
      if (MODEL == KGE_TRANSE) {
        if (NORM == 2) dist = sqrtf(part);
        score = gamma > 0.f ? gamma - dist : dist;
      }
      float g, l;
      pointwise_term<LOSS>(score, is_pos, margin, is_pos ? inv_pos : inv_neg, g, l);
      if (MODEL == KGE_TRANSE) {
        g = gamma > 0.f ? -g : g;
        if (NORM == 2) g = dist > 0.f ? g / dist : 0.f;
      }
      if (lane == 0) {
        lsum += l;
        coef[t] = g;
        if (scores) scores[t] = score;
      }
      if (g != 0.f) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          if (MODEL == KGE_DISTMULT) {
            float4 term = mul4(sv[j], ov[j]);
            if (DROP) term = mul4(term, drop_mul4(kp, lane + j * G, thr, dscale));
            acc[j] = fma4(f4(g), term, acc[j]);
          } else if (MODEL == KGE_TRANSE) {
            const float4 x = sub4(add4(sv[j], pm[j]), ov[j]);
            float4 term = NORM == 1 ? sgn4(x) : x;
            if (DROP) term = mul4(term, drop_mul4(kp, lane + j * G, thr, dscale));
            acc[j] = fma4(f4(g), term, acc[j]);
          } else {
            const float4 sr = sv[j], si = sv[NV + j], orr = ov[j], oi = ov[NV + j];
            float4 gr = add4(mul4(sr, orr), mul4(si, oi)), gi = sub4(mul4(sr, oi), mul4(si, orr));
            if (DROP) { gr = mul4(gr, drop_mul4(kp, lane + j * G, thr, dscale)); gi = mul4(gi, drop_mul4(kp, hc + lane + j * G, thr, dscale)); }
            acc[j] = fma4(f4(g), gr, acc[j]);
            acc[NV + j] = fma4(f4(g), gi, acc[NV + j]);
          }
        }
      }
    }
    // meet the 4 sub-groups, one slab row per unit
#pragma unroll
    for (int i = 0; i < NA; ++i) { acc[i] = add4(acc[i], shx4(acc[i], 16)); acc[i] = add4(acc[i], shx4(acc[i], 32)); }
    if (sub == 0) {
      float* dst = rslab + (int64_t)unit * d_e;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int ch = lane + j * G;
        if (ch < nchunk) {
          *reinterpret_cast<float4*>(dst + ch * 4) = acc[j];
          if (MODEL == KGE_COMPLEX) *reinterpret_cast<float4*>(dst + h + ch * 4) = acc[NV + j];
        }
      }
    }
  }
  const float b = block_sum(lsum, red);
  if (threadIdx.x == 0) loss_partials[blockIdx.x] = b;
}

// gR[p] += sum of the relation's slab rows, in chunk order (one wave per relation, 8 independent loads in flight)
// Also: leaves rcnt all-zero for the next step (the workspace is self-cleaning; it is zeroed once at allocation), and the
// extra last block reduces the forward's loss partials (fixed order) into *loss_out.
__global__ __launch_bounds__(256) void rel_finish_kernel(float* __restrict__ gR, int R, int d_e, int32_t* __restrict__ rcnt,
                                                          const int32_t* __restrict__ ruoff, const float* __restrict__ rslab,
                                                          const float* __restrict__ loss_partials, int n_partials,
                                                          float* __restrict__ loss_out) {
  __shared__ float red[16];
  if (blockIdx.x == gridDim.x - 1) {  // loss = sum of block partials
    float s = 0.f;
    for (int i = threadIdx.x; i < n_partials; i += 256) s += loss_partials[i];
    const float b = block_sum(s, red);
    if (threadIdx.x == 0) loss_out[0] = b;
    return;
  }
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= R) return;
  const int lane = threadIdx.x & 63;
  const int c = rcnt[p];
  if (lane == 0) rcnt[p] = 0;
  const int nch = (c + kRelChunk - 1) / kRelChunk;
  if (nch == 0) return;
  const float* src = rslab + (int64_t)ruoff[p] * d_e;
  float* dst = gR + (int64_t)p * d_e;
  for (int k = lane; k < d_e; k += 64) {
    float s = 0.f;
    int j = 0;
    for (; j + 8 <= nch; j += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(j + u) * d_e + k];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; j < nch; ++j) s += src[(int64_t)j * d_e + k];
    dst[k] += s;
  }
}

// ---------------------------------------------------------------------------------------------------------------
static int64_t b256(int64_t x) { return align_up(x, 256); }
int pull_chunk();

bool pull2_supported(const kge_model_desc* md, const kge_loss_desc* ld) {
  if (md->model != KGE_DISTMULT && md->model != KGE_TRANSE && md->model != KGE_COMPLEX) return false;
  if (ld->loss != KGE_POINTWIZE_HINGE && ld->loss != KGE_POINTWIZE_LOGISTIC && ld->loss != KGE_POINTWIZE_SQUARE &&
      ld->loss != KGE_POINTWIZE_CROSS_ENTROPY)
    return false;
  if (md->num_relations > kRelMax) return false;
  const int width = md->model == KGE_COMPLEX ? md->d_e / 2 : md->d_e;
  return width % 4 == 0 && md->d_e % 4 == 0 && width / 4 <= 64;
}

int64_t pull2_workspace_bytes(const kge_model_desc* md, int64_t n) {
  const int64_t R = md->num_relations, N = md->num_entities;
  const int64_t max_units = n / kRelChunk + R + 1;
  const int C = pull_chunk();
  const int64_t max_extra = 2 * n / C + 1;
  return b256(R * 4) * 3 + 256 + b256(max_units * 8) + b256(n * 4) + b256(n * 16) + b256(max_units * md->d_e * 4) +
         b256(N * 4) * 4 + 256 + b256(max_extra * 8) + b256(2 * n * 16) + b256(2 * n * 4) + b256(n * 4) + b256(2 * max_extra * md->d_e * 4) + 2048;
}

static void carve2(void* hdr, void* ws, const kge_model_desc* md, int64_t n, Plan2* p) {
  const int64_t R = md->num_relations, N = md->num_entities;
  const int64_t max_units = n / kRelChunk + R + 1;
  const int C = pull_chunk();
  const int64_t max_extra = 2 * n / C + 1;
  char* q = reinterpret_cast<char*>(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  auto take = [&](int64_t bytes) { char* r = q; q += b256(bytes); return r; };
  char* hq = reinterpret_cast<char*>(((uintptr_t)hdr + 255) & ~(uintptr_t)255);
  p->ent.cnt = (int32_t*)hq;                 // header (fixed place, zero on entry, left zero): entity counters ...
  p->rcnt = (int32_t*)(hq + b256(N * 4));    // ... then relation counters
  p->roff = (int32_t*)take(R * 4);
  p->ruoff = (int32_t*)take(R * 4);
  p->rtotal = (int32_t*)take(16);
  p->rumap = (int2*)take(max_units * 8);
  p->rrank = (int32_t*)take(n * 4);
  p->rentries = (int4*)take(n * 16);
  p->rslab = (float*)take(max_units * md->d_e * 4);
  p->max_units = (int)max_units;
  p->ent.off = (int32_t*)take(N * 4);
  p->ent.eoff = (int32_t*)take(N * 4);
  p->ent.poff = (int32_t*)take(N * 4);
  p->ent.totals = (int32_t*)take(16);
  p->ent.emap = (int2*)take(max_extra * 8);
  p->ent.entries = (int4*)take(2 * n * 16);
  p->erank = (int32_t*)take(2 * n * 4);
  p->ent.ranks = p->erank;
  p->ent.coef = (float*)take(n * 4);
  p->ent.slab = (float*)take(2 * max_extra * md->d_e * 4);
  p->ent.max_extra = (int)max_extra;
  p->ent.chunk = C;
}

// kernels of pull.hip reused on the entity keys
void pull2_entity_plan(const PullPlan& e, int N, hipStream_t st);
int pull2_entity_pull(const kge_model_desc* md, const float* E, const float* Rel, float* gE, float* gR, const PullPlan& e, uint64_t seed,
                      hipStream_t st);

template <int MODEL, int NORM, int LOSS>
static void launch_fwd_rel(int nv, bool drop, int grid, hipStream_t st, const float* E, const float* Rel, int64_t m_pos, int64_t m_neg,
                           const kge_model_desc* md, int nchunk, float margin, const Plan2& p, float* scores, float* partials,
                           uint64_t seed) {
  const uint32_t thr = drop_thr(md->dropout);
  const float dscale = 1.f / (1.f - md->dropout);
#define KGE_FR(NVV, DR) hipLaunchKernelGGL((fwd_rel_kernel<MODEL, NORM, LOSS, NVV, DR>), dim3(grid), dim3(256), 0, st, E, Rel, m_pos, m_neg, md->d_e, \
                                           nchunk, md->gamma, margin, p.rcnt, p.roff, p.rtotal, p.rumap, p.rentries, p.ent.coef, scores, p.rslab,  \
                                           partials, seed, thr, dscale)
#define KGE_FR2(NVV) do { if (drop) KGE_FR(NVV, true); else KGE_FR(NVV, false); } while (0)
  if (nv <= 1) KGE_FR2(1);
  else if (nv <= 2) KGE_FR2(2);
  else KGE_FR2(4);
#undef KGE_FR2
#undef KGE_FR
}

template <int MODEL, int NORM>
static void launch_fwd_rel_loss(int loss, int nv, bool drop, int grid, hipStream_t st, const float* E, const float* Rel, int64_t m_pos,
                                int64_t m_neg, const kge_model_desc* md, int nchunk, float margin, const Plan2& p, float* scores,
                                float* partials, uint64_t seed) {
  if (loss == KGE_POINTWIZE_HINGE) launch_fwd_rel<MODEL, NORM, KGE_POINTWIZE_HINGE>(nv, drop, grid, st, E, Rel, m_pos, m_neg, md, nchunk, margin, p, scores, partials, seed);
  else if (loss == KGE_POINTWIZE_LOGISTIC) launch_fwd_rel<MODEL, NORM, KGE_POINTWIZE_LOGISTIC>(nv, drop, grid, st, E, Rel, m_pos, m_neg, md, nchunk, margin, p, scores, partials, seed);
  else if (loss == KGE_POINTWIZE_SQUARE) launch_fwd_rel<MODEL, NORM, KGE_POINTWIZE_SQUARE>(nv, drop, grid, st, E, Rel, m_pos, m_neg, md, nchunk, margin, p, scores, partials, seed);
  else launch_fwd_rel<MODEL, NORM, KGE_POINTWIZE_CROSS_ENTROPY>(nv, drop, grid, st, E, Rel, m_pos, m_neg, md, nchunk, margin, p, scores, partials, seed);
}

// `side` may be null (then everything runs on `st`).
int train_grads_pull2(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, float* gE, float* gR,
                      const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, float* scores, float* loss_partials,
                      int* n_partials, float* loss_out, void* hdr, void* ws, uint32_t* err, uint64_t seed, hipStream_t st, hipStream_t side,
                      hipEvent_t fork, hipEvent_t join) {
  const int64_t n = m_pos + m_neg;
  Plan2 p;
  carve2(hdr, ws, md, n, &p);
  const int N = md->num_entities, R = md->num_relations;
  const int tgrid = (int)cdiv(n, 256);
  const int width = md->model == KGE_COMPLEX ? md->d_e / 2 : md->d_e;
  const int nchunk = width / 4;
  const bool drop = md->dropout > 0.f;
  // ---- entity index on the side stream
  hipStream_t es = side ? side : st;
  if (side) { KGE_HIP_TRY(hipEventRecord(fork, st)); KGE_HIP_TRY(hipStreamWaitEvent(side, fork, 0)); }
  {
    ProfScope prof(KGE_PROF_INDEX, es);
    hipLaunchKernelGGL(ent_hist_kernel, dim3(tgrid), dim3(256), 0, es, pos, m_pos, neg, m_neg, N, R, p.ent.cnt, p.erank);
    pull2_entity_plan(p.ent, N, es);
  }
  if (side) KGE_HIP_TRY(hipEventRecord(join, side));
  // ---- relation index + forward on the main stream
  hipLaunchKernelGGL(rel_hist_kernel, dim3((int)cdiv(n, kRelBlockTriples)), dim3(256), 0, st, pos, m_pos, neg, m_neg, N, R, p.rcnt, p.rrank);
  hipLaunchKernelGGL(rel_plan_kernel, dim3(1), dim3(256), 0, st, p.rcnt, R, p.roff, p.ruoff, p.rtotal, p.rumap, p.max_units);
  hipLaunchKernelGGL(rel_fill_kernel, dim3(tgrid), dim3(256), 0, st, pos, m_pos, neg, m_neg, p.roff, p.rrank, p.rentries, p.ent.coef, scores, err);
  const int fgrid = (int)cdiv(p.max_units, 4);
  *n_partials = fgrid;
  {
    ProfScope prof(KGE_PROF_FWD, st);
    const int nv = (int)cdiv(nchunk, 16);
    if (md->model == KGE_DISTMULT) launch_fwd_rel_loss<KGE_DISTMULT, 1>(ld->loss, nv, drop, fgrid, st, E, Rel, m_pos, m_neg, md, nchunk, ld->margin, p, scores, loss_partials, seed);
    else if (md->model == KGE_COMPLEX) launch_fwd_rel_loss<KGE_COMPLEX, 1>(ld->loss, nv, drop, fgrid, st, E, Rel, m_pos, m_neg, md, nchunk, ld->margin, p, scores, loss_partials, seed);
    else if (md->norm == 1) launch_fwd_rel_loss<KGE_TRANSE, 1>(ld->loss, nv, drop, fgrid, st, E, Rel, m_pos, m_neg, md, nchunk, ld->margin, p, scores, loss_partials, seed);
    else launch_fwd_rel_loss<KGE_TRANSE, 2>(ld->loss, nv, drop, fgrid, st, E, Rel, m_pos, m_neg, md, nchunk, ld->margin, p, scores, loss_partials, seed);
  }
  if (side) KGE_HIP_TRY(hipStreamWaitEvent(st, join, 0));
  hipLaunchKernelGGL(ent_fill_kernel, dim3(tgrid), dim3(256), 0, st, pos, m_pos, neg, m_neg, p.ent.coef, p.ent.off, p.erank, p.ent.entries);
  if (int rc = pull2_entity_pull(md, E, Rel, gE, gR, p.ent, seed, st)) return rc;
  hipLaunchKernelGGL(rel_finish_kernel, dim3((int)cdiv(R, 4) + 1), dim3(256), 0, st, gR, R, md->d_e, p.rcnt, p.ruoff, p.rslab, loss_partials,
                     *n_partials, loss_out);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

}  // namespace kge
