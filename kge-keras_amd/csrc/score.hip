// Forward scoring: the fused "3 row gathers + func" of EmbeddingModel.call in predict mode.
//
// Replaces (reference repo): tf.unstack + 3x Embedding gather (KGEkeras/models.py:140-150) and
// DistMult.func (:171-172), TransE.func (:186-190), ComplEx.func (:212-222), called through
// model.predict at KGEkeras/utils.py:211,231.
//
// Kernel shape (gfx950): a sub-group of G lanes owns one triple; lane j reads the 16-byte chunks
// j, j+G, ... of the three rows (global_load_dwordx4, 16*G contiguous bytes per row per instruction),
// multiplies/accumulates in registers and the G partial sums meet in a log2(G)-step cross-lane
// reduction.  No LDS, no re-read: algorithmic bytes = 4*(2*d_e + d_r) + 12 + 4 per triple, and the
// kernel is bound by how many row bytes the chip keeps in flight (HBM / Infinity Cache / L2).
#include "dropout.hpp"
#include "kge_internal.hpp"
#include "pull.hpp"

namespace kge {

template <int VEC>
struct Chunk;
template <>
struct Chunk<4> {
  using T = float4;
  static __device__ __forceinline__ T ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
};
template <>
struct Chunk<1> {
  using T = float;
  static __device__ __forceinline__ T ld(const float* p) { return *p; }
};

__device__ __forceinline__ float dm3(float s, float p, float o) { return s * p * o; }
__device__ __forceinline__ float dm3(float4 s, float4 p, float4 o) {
  return s.x * p.x * o.x + s.y * p.y * o.y + s.z * p.z * o.z + s.w * p.w * o.w;
}
template <int NORM>
__device__ __forceinline__ float te1(float s, float p, float o, float pw) {
  float x = s + p - o;
  return NORM == 1 ? fabsf(x) : (NORM == 3 ? phase_abs(x) : (NORM == 5 ? powf(fabsf(x), pw) : x * x));  // 5: general p-norm, exponent pw
}
template <int NORM>
__device__ __forceinline__ float te1(float4 s, float4 p, float4 o, float pw) {
  return te1<NORM>(s.x, p.x, o.x, pw) + te1<NORM>(s.y, p.y, o.y, pw) + te1<NORM>(s.z, p.z, o.z, pw) + te1<NORM>(s.w, p.w, o.w, pw);
}
__device__ __forceinline__ float cx1(float sr, float si, float pr, float pi, float orr, float oi) {
  // models.py:218-222  s1+s2+s3-s4
  return sr * pr * orr + pr * si * oi + pi * sr * oi - pi * si * orr;
}
__device__ __forceinline__ float cx1(float4 sr, float4 si, float4 pr, float4 pi, float4 orr, float4 oi) {
  return cx1(sr.x, si.x, pr.x, pi.x, orr.x, oi.x) + cx1(sr.y, si.y, pr.y, pi.y, orr.y, oi.y) +
         cx1(sr.z, si.z, pr.z, pi.z, orr.z, oi.z) + cx1(sr.w, si.w, pr.w, pi.w, orr.w, oi.w);
}

// RotatE as written (models.py:439-459): (re_s cos p - im_s sin p - re_o) + (re_s sin p + im_s cos p - im_o), summed with its signs.
// The reference wraps the phase first (atan2(sin p, cos p)); cos / sin of the wrapped phase are cos p / sin p.
__device__ __forceinline__ float rt1(float sr, float si, float ph, float orr, float oi) {
  float sn, c;
  sincosf(ph, &sn, &c);
  return (sr * c - si * sn - orr) + (sr * sn + si * c - oi);
}
__device__ __forceinline__ float rt1(float4 sr, float4 si, float4 ph, float4 orr, float4 oi) {
  return rt1(sr.x, si.x, ph.x, orr.x, oi.x) + rt1(sr.y, si.y, ph.y, orr.y, oi.y) + rt1(sr.z, si.z, ph.z, orr.z, oi.z) + rt1(sr.w, si.w, ph.w, orr.w, oi.w);
}

// nchunk: chunks (of VEC floats) per row for DistMult/TransE, per HALF row for ComplEx and RotatE.
// DROP: EmbeddingModel.call(x, training=True) -- the Dropout layer on the three gathered rows is active (models.py:122,144-148): row i of
// the call takes the masks of batch row row0 + i of step `seed`, the same counter-based masks the training kernels regenerate
// (dropout.hpp), kept entries scaled by 1 / (1 - rate).
template <int VEC> struct DMul;
template <> struct DMul<4> {
  static __device__ __forceinline__ float4 m(DropKey k, int c, uint32_t thr, float sc) { return drop_mul4(k, (uint32_t)c, thr, sc); }
  static __device__ __forceinline__ float4 mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
};
template <> struct DMul<1> {
  static __device__ __forceinline__ float m(DropKey k, int c, uint32_t thr, float sc) { return drop_mul(k, (uint32_t)c, thr, sc); }
  static __device__ __forceinline__ float mul(float a, float b) { return a * b; }
};

template <int MODEL, int NORM, int G, int VEC, bool DROP = false>
__global__ __launch_bounds__(256) void score_kernel(const float* __restrict__ E, const float* __restrict__ Rel,
                                                     const int32_t* __restrict__ triples, int64_t n, int N, int R,
                                                     int d_e, int d_r, int nchunk, float gamma, float modulus,
                                                     float* __restrict__ out, uint32_t* __restrict__ err, uint64_t seed = 0,
                                                     int64_t row0 = 0, uint32_t thr = 0, float dscale = 1.f) {
  using C = Chunk<VEC>;
  using D = DMul<VEC>;
  const int lane = threadIdx.x & (G - 1);
  const int64_t group = (int64_t)(blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / G;
  const int64_t ngroups = (int64_t)gridDim.x * blockDim.x / G;
  for (int64_t t = group; t < n; t += ngroups) {
    const int s = triples[3 * t + 0], p = triples[3 * t + 1], o = triples[3 * t + 2];
    const bool ok = (unsigned)s < (unsigned)N && (unsigned)p < (unsigned)R && (unsigned)o < (unsigned)N;
    float acc = 0.f;
    if (ok) {
      const float* es = E + (int64_t)s * d_e;
      const float* rp = Rel + (int64_t)p * d_r;
      const float* eo = E + (int64_t)o * d_e;
      DropKey ks{0, 0}, kp{0, 0}, ko{0, 0};
      if (DROP) { ks = drop_key(seed, row0 + t, 0); kp = drop_key(seed, row0 + t, 1); ko = drop_key(seed, row0 + t, 2); }
      if (MODEL == KGE_COMPLEX) {
        const int h = d_e >> 1, hc = h / VEC;
#pragma unroll 2
        for (int c = lane; c < nchunk; c += G) {
          auto sr = C::ld(es + c * VEC), si = C::ld(es + h + c * VEC);
          auto pr = C::ld(rp + c * VEC), pi = C::ld(rp + h + c * VEC);
          auto orr = C::ld(eo + c * VEC), oi = C::ld(eo + h + c * VEC);
          if (DROP) {
            sr = D::mul(sr, D::m(ks, c, thr, dscale)); si = D::mul(si, D::m(ks, hc + c, thr, dscale));
            pr = D::mul(pr, D::m(kp, c, thr, dscale)); pi = D::mul(pi, D::m(kp, hc + c, thr, dscale));
            orr = D::mul(orr, D::m(ko, c, thr, dscale)); oi = D::mul(oi, D::m(ko, hc + c, thr, dscale));
          }
          acc += cx1(sr, si, pr, pi, orr, oi);
        }
      } else if (MODEL == KGE_ROTATE) {
        const int h = d_e >> 1, hc = h / VEC;
#pragma unroll 2
        for (int c = lane; c < nchunk; c += G) {
          auto sr = C::ld(es + c * VEC), si = C::ld(es + h + c * VEC), ph = C::ld(rp + c * VEC), orr = C::ld(eo + c * VEC), oi = C::ld(eo + h + c * VEC);
          if (DROP) {
            sr = D::mul(sr, D::m(ks, c, thr, dscale)); si = D::mul(si, D::m(ks, hc + c, thr, dscale)); ph = D::mul(ph, D::m(kp, c, thr, dscale));
            orr = D::mul(orr, D::m(ko, c, thr, dscale)); oi = D::mul(oi, D::m(ko, hc + c, thr, dscale));
          }
          acc += rt1(sr, si, ph, orr, oi);
        }
      } else {
#pragma unroll 4
        for (int c = lane; c < nchunk; c += G) {
          auto sv = C::ld(es + c * VEC), pv = C::ld(rp + c * VEC), ov = C::ld(eo + c * VEC);
          if (DROP) { sv = D::mul(sv, D::m(ks, c, thr, dscale)); pv = D::mul(pv, D::m(kp, c, thr, dscale)); ov = D::mul(ov, D::m(ko, c, thr, dscale)); }
          acc += (MODEL == KGE_DISTMULT) ? dm3(sv, pv, ov) : te1<NORM>(sv, pv, ov, modulus);
        }
      }
    }
    acc = group_sum<G>(acc);
    if (lane == 0) {
      float sc = acc;
      if (MODEL == KGE_TRANSE) {
        if (NORM == 2) sc = sqrtf(sc);
        if (NORM == 3) sc *= modulus;
        if (NORM == 5) sc = powf(sc, 1.f / modulus);
        sc = gamma > 0.f ? gamma - sc : sc;  // models.py:187-190, 486-489
      }
      if (MODEL == KGE_ROTATE) sc = gamma > 0.f ? gamma - sc : sc;  // models.py:455-458
      if (!ok) {
        sc = __builtin_nanf("");
        atomicOr(err, 1u);
      }
      out[t] = sc;
    }
  }
}

struct ScoreDrop { bool on; uint64_t seed; int64_t row0; };  // training-mode scores (dropout active): masks of batch rows row0.. of step `seed`

template <int MODEL, int NORM, int G, int VEC>
static int launch_score(const kge_model_desc* md, const float* E, const float* Rel, const int32_t* triples, int64_t n,
                        int nchunk, float* out, uint32_t* err, hipStream_t st, const ScoreDrop& dr) {
  const int64_t want = cdiv(n * G, 256);
  const int grid = (int)(want < (int64_t)kCUs * 16 ? want : (int64_t)kCUs * 16);
  ProfScope prof(KGE_PROF_SCORE, st);
  if (dr.on)
    hipLaunchKernelGGL((score_kernel<MODEL, NORM, G, VEC, true>), dim3(grid), dim3(256), 0, st, E, Rel, triples, n, md->num_entities, md->num_relations,
                       md->d_e, md->d_r, nchunk, md->gamma, md->modulus, out, err, dr.seed, dr.row0, drop_thr(md->dropout), 1.f / (1.f - md->dropout));
  else
  hipLaunchKernelGGL((score_kernel<MODEL, NORM, G, VEC>), dim3(grid), dim3(256), 0, st, E, Rel, triples, n,
                     md->num_entities, md->num_relations, md->d_e, md->d_r, nchunk, md->gamma, md->modulus, out, err);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

template <int MODEL, int NORM, int VEC>
static int pick_g(int G, const kge_model_desc* md, const float* E, const float* Rel, const int32_t* tr, int64_t n,
                  int nchunk, float* out, uint32_t* err, hipStream_t st, const ScoreDrop& dr) {
  switch (G) {
    case 4: return launch_score<MODEL, NORM, 4, VEC>(md, E, Rel, tr, n, nchunk, out, err, st, dr);
    case 8: return launch_score<MODEL, NORM, 8, VEC>(md, E, Rel, tr, n, nchunk, out, err, st, dr);
    case 16: return launch_score<MODEL, NORM, 16, VEC>(md, E, Rel, tr, n, nchunk, out, err, st, dr);
    case 32: return launch_score<MODEL, NORM, 32, VEC>(md, E, Rel, tr, n, nchunk, out, err, st, dr);
    default: return launch_score<MODEL, NORM, 64, VEC>(md, E, Rel, tr, n, nchunk, out, err, st, dr);
  }
}

int score_group_width(int nchunk);  // tuning.cpp

static int score_dispatch_impl(const kge_model_desc* md, const float* E, const float* Rel, const int32_t* triples, int64_t n,
                               float* out, uint32_t* err, hipStream_t st, const ScoreDrop& dr) {
  const int width = (md->model == KGE_COMPLEX || md->model == KGE_ROTATE) ? md->d_e / 2 : md->d_e;
  const bool v4 = (width % 4 == 0) && (md->d_e % 4 == 0) && (((uintptr_t)E | (uintptr_t)Rel) % 16 == 0);
  const int nchunk = v4 ? width / 4 : width;
  const int G = score_group_width(nchunk);
#define KGE_GO(MODEL, NORM)                                                                          \
  return v4 ? pick_g<MODEL, NORM, 4>(G, md, E, Rel, triples, n, nchunk, out, err, st, dr)            \
            : pick_g<MODEL, NORM, 1>(G, md, E, Rel, triples, n, nchunk, out, err, st, dr)
  switch (md->model) {
    case KGE_DISTMULT: KGE_GO(KGE_DISTMULT, 1);
    case KGE_TRANSE:
      if (md->norm == 1) { KGE_GO(KGE_TRANSE, 1); }
      if (md->norm == 3) { KGE_GO(KGE_TRANSE, 3); }
      if (md->norm == 5) { KGE_GO(KGE_TRANSE, 5); }
      KGE_GO(KGE_TRANSE, 2);
    case KGE_COMPLEX: KGE_GO(KGE_COMPLEX, 1);
    case KGE_ROTATE: KGE_GO(KGE_ROTATE, 1);
    case KGE_HOLE:
      if (dr.on && dr.row0 != 0) return fail(KGE_EUNSUPPORTED, "kge_score_training: HolE takes row_offset 0 only");
      return hole_forward(md, -1, 0.f, E, Rel, triples, n, nullptr, 0, nullptr, out, nullptr, nullptr, err, dr.on ? dr.seed : 0, dr.on, st);
    default: return fail(KGE_EUNSUPPORTED, "kge_score: model %d has no HIP scoring kernel yet", md->model);
  }
#undef KGE_GO
}

int score_dispatch(const kge_model_desc* md, const float* E, const float* Rel, const int32_t* triples, int64_t n,
                   float* out, uint32_t* err, hipStream_t st) {
  return score_dispatch_impl(md, E, Rel, triples, n, out, err, st, ScoreDrop{false, 0, 0});
}
// scores with the Dropout layer active (md->dropout > 0; rate 0: plain scores)
int score_dispatch_training(const kge_model_desc* md, const float* E, const float* Rel, const int32_t* triples, int64_t n, uint64_t seed,
                            int64_t row0, float* out, uint32_t* err, hipStream_t st) {
  return score_dispatch_impl(md, E, Rel, triples, n, out, err, st, ScoreDrop{md->dropout > 0.f, seed, row0});
}

}  // namespace kge
