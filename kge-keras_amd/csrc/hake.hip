// HAKE (KGEkeras/models.py:362-400), exactly as written.  Entity rows are [phase | modulus] (2h floats), relation rows
// [phase | modulus | bias] (3h floats; models.py:371-372).  For a row t of the tensor handed to ONE call of the model:
//
//   phase_x   = x_phase / (embedding_range / pi)                                                     models.py:391-393
//   bias      = clip(bias_p, -inf, 1);  bias = where(bias < -|mod_p|, -|mod_p|, bias)                models.py:395-396
//   M[t][k]   = mod_s (mod_p + bias) - |mod_o| (1 - bias)
//   p_score_t = phase_weight * sum_k |sin((phase_s + phase_p - phase_o) / 2)|                        models.py:399  (ord=1, axis=-1)
//   r_score   = mod_weight * sqrt(sum over ALL t, k of M^2)                                          models.py:398  (tf.norm ord=2, NO axis)
//   score_t   = gamma - (p_score_t + r_score)                                                        models.py:400
//
// tf.norm without an axis is one vector norm over the whole tensor [TF], so the modulus part is ONE scalar per call: over a side of
// the training batch (the two-input wrapper calls the model once on the positives and once on the negatives, tests.py:77-78), or over
// a chunk of 32 rows in model.predict (Keras' default predict batch size [TF]) -- the score of a triple depends on the rows it is
// scored with, and this file reproduces exactly that: every entry point takes the rows of a call as a SEGMENT.
//
//   hake_rows_kernel   one wave per row: P_t = sum_k |sin(y)|, Q_t = sum_k M^2 (dropout masks regenerated as everywhere, dropout.hpp)
//   hake_seg_kernel    one workgroup per segment: r = sqrt(sum_t Q_t) (fixed order), score_t = gamma - (pw P_t + mw r)
//   (loss and dL/dscore: the library's loss kernels on the score vectors, utils.py:288-319)
//   hake_gsum_kernel   G_side = sum_t dL/dscore_t (every row's modulus gradient carries the sum over its segment)
//   hake_grad_kernel   one wave per row: the three gradient rows, added to the dense gradient tables with float atomics
//
// A composite of launches like the ConvE / literal steps: HAKE is API surface, not the metric's hot path.
#include "dropout.hpp"
#include "kge_internal.hpp"

namespace kge {

int loss_value(const kge_loss_desc*, const float*, int64_t, const float*, int64_t, float*, float*, float*, void*, int64_t, hipStream_t);  // train.hip
int64_t train_workspace_bytes(const kge_model_desc* md, int64_t m_pos, int64_t m_neg);                                                    // train.hip

struct HakeArgs {
  const float* E; const float* Rel;
  const int32_t* pos; int64_t m_pos; const int32_t* neg; int64_t m_neg;
  int N, R, h;
  float gamma, pw, mw, inv_scale;          // inv_scale = pi / embedding_range
  uint64_t seed; uint32_t thr; float dscale;
};

// bias as the score uses it, and which branch it came from: 0 = bias_p itself (gradient passes), 1 = clipped to 1, 2 = -|mod_p|
__device__ __forceinline__ float hake_bias(float bp, float mp, int* branch) {
  float b = fminf(bp, 1.f);
  int br = bp > 1.f ? 1 : 0;
  if (b < -fabsf(mp)) { b = -fabsf(mp); br = 2; }
  *branch = br;
  return b;
}

// the row's ids; false (and the error flag) when one of them is outside the tables
__device__ __forceinline__ bool hake_ids(const HakeArgs& a, int64_t t, int* s, int* p, int* o, uint32_t* err) {
  const int32_t* tr = t < a.m_pos ? a.pos + 3 * t : a.neg + 3 * (t - a.m_pos);
  *s = tr[0]; *p = tr[1]; *o = tr[2];
  const bool ok = (unsigned)*s < (unsigned)a.N && (unsigned)*p < (unsigned)a.R && (unsigned)*o < (unsigned)a.N;
  if (!ok && (threadIdx.x & 63) == 0) atomicOr(err, 1u);
  return ok;
}

template <bool DROP>
__global__ __launch_bounds__(256) void hake_rows_kernel(HakeArgs a, float* __restrict__ P, float* __restrict__ Q, uint32_t* __restrict__ err) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = a.h;
  const int64_t n = a.m_pos + a.m_neg;
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < n; t += (int64_t)gridDim.x * 4) {
    int s, p, o;
    const bool ok = hake_ids(a, t, &s, &p, &o, err);
    float accP = 0.f, accQ = 0.f;
    if (ok) {
      const float* rs = a.E + (int64_t)s * 2 * h;
      const float* ro = a.E + (int64_t)o * 2 * h;
      const float* rp = a.Rel + (int64_t)p * 3 * h;
      DropKey ks{0, 0}, kp{0, 0}, ko{0, 0};
      if (DROP) { ks = drop_key(a.seed, t, 0); kp = drop_key(a.seed, t, 1); ko = drop_key(a.seed, t, 2); }
      auto ms_ = [&](int e) { return DROP ? drop_mul(ks, e, a.thr, a.dscale) : 1.f; };
      auto mp_ = [&](int e) { return DROP ? drop_mul(kp, e, a.thr, a.dscale) : 1.f; };
      auto mo_ = [&](int e) { return DROP ? drop_mul(ko, e, a.thr, a.dscale) : 1.f; };
      for (int k = lane; k < h; k += 64) {
        const float y = 0.5f * a.inv_scale * (rs[k] * ms_(k) + rp[k] * mp_(k) - ro[k] * mo_(k));
        accP += fabsf(sinf(y));
        const float mods = rs[h + k] * ms_(h + k), modo = ro[h + k] * mo_(h + k), modp = rp[h + k] * mp_(h + k);
        int br;
        const float b = hake_bias(rp[2 * h + k] * mp_(2 * h + k), modp, &br);
        const float M = mods * (modp + b) - fabsf(modo) * (1.f - b);
        accQ += M * M;
      }
    }
    accP = group_sum<64>(accP);
    accQ = group_sum<64>(accQ);
    if (lane == 0) { P[t] = ok ? accP : __builtin_nanf(""); Q[t] = ok ? accQ : 0.f; }
  }
}

// Segment sg covers rows [off[sg], off[sg + 1]) when a list of offsets is given (the calls of many model.predict runs in one launch), else
// [b[sg], b[sg + 1]) with b = {0, m0, n} (two segments: the sides of a training batch) or, chunk > 0, rows
// [sg * chunk, min(n, (sg + 1) * chunk)) (one model.predict).  r[sg] = sqrt(sum of Q over the segment), scores filled.
__global__ __launch_bounds__(256) void hake_seg_kernel(const float* __restrict__ P, const float* __restrict__ Q, int64_t n, int64_t m0, int64_t chunk,
                                                        const int64_t* __restrict__ off, float gamma, float pw, float mw, float* __restrict__ scores,
                                                        float* __restrict__ rnorm) {
  __shared__ float red[16];
  const int64_t lo = off ? max((int64_t)0, off[blockIdx.x]) : chunk > 0 ? (int64_t)blockIdx.x * chunk : (blockIdx.x == 0 ? 0 : m0);
  const int64_t hi = off ? min(n, off[blockIdx.x + 1]) : chunk > 0 ? min(n, lo + chunk) : (blockIdx.x == 0 ? m0 : n);
  float s = 0.f;
  for (int64_t t = lo + threadIdx.x; t < hi; t += 256) s += Q[t];
  const float tot = block_sum(s, red);
  __shared__ float r_s;
  if (threadIdx.x == 0) { r_s = sqrtf(tot); rnorm[blockIdx.x] = r_s; }
  __syncthreads();
  const float r = r_s;
  for (int64_t t = lo + threadIdx.x; t < hi; t += 256) scores[t] = gamma - (pw * P[t] + mw * r);
}

// counts[q] = #{rows t of query q, [qoff[q], qoff[q + 1]), with score_t > score of the query's first row}: utils.py:211-215 (the target is row 0 of
// the candidate list; real_rank = n - rankdata(.,'max')[0] + 1 = 1 + that count)
__global__ __launch_bounds__(256) void hake_rank_kernel(const float* __restrict__ scores, const int64_t* __restrict__ qoff, int64_t n, int32_t* __restrict__ counts) {
  __shared__ float red[16];
  const int64_t lo = max((int64_t)0, qoff[blockIdx.x]), hi = min(n, qoff[blockIdx.x + 1]);
  if (lo >= hi) { if (threadIdx.x == 0) counts[blockIdx.x] = 0; return; }
  const float tg = scores[lo];
  int c = 0;
  for (int64_t t = lo + 1 + threadIdx.x; t < hi; t += 256) c += scores[t] > tg ? 1 : 0;
  const float tot = block_sum((float)c, red);  // (exact: at most 2^24 rows per query)
  if (threadIdx.x == 0) counts[blockIdx.x] = (int32_t)tot;
}

// G[0] = sum of g over the positives, G[1] over the negatives (fixed order)
__global__ __launch_bounds__(256) void hake_gsum_kernel(const float* __restrict__ g, int64_t m0, int64_t n, float* __restrict__ G) {
  __shared__ float red[16];
  const int64_t lo = blockIdx.x == 0 ? 0 : m0, hi = blockIdx.x == 0 ? m0 : n;
  float s = 0.f;
  for (int64_t t = lo + threadIdx.x; t < hi; t += 256) { const float v = g[t]; s += v == v ? v : 0.f; }
  const float tot = block_sum(s, red);
  if (threadIdx.x == 0) G[blockIdx.x] = tot;
}

template <bool DROP>
__global__ __launch_bounds__(256) void hake_grad_kernel(HakeArgs a, const float* __restrict__ g, const float* __restrict__ G, const float* __restrict__ rnorm,
                                                         float* __restrict__ gE, float* __restrict__ gR) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = a.h;
  const int64_t n = a.m_pos + a.m_neg;
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < n; t += (int64_t)gridDim.x * 4) {
    const int32_t* tr = t < a.m_pos ? a.pos + 3 * t : a.neg + 3 * (t - a.m_pos);
    const int s = tr[0], p = tr[1], o = tr[2];
    if (!((unsigned)s < (unsigned)a.N && (unsigned)p < (unsigned)a.R && (unsigned)o < (unsigned)a.N)) continue;  // (flagged by the forward)
    const int side = t < a.m_pos ? 0 : 1;
    float gi = g[t];
    if (!(gi == gi)) gi = 0.f;
    const float r = rnorm[side];
    const float cM = r > 0.f ? -a.mw * G[side] / r : 0.f;  // dL/dM = cM * M  ([TF]: 0/0 = NaN at r == 0; zero here, like TransE-L2 at x = 0)
    const float cP = -a.pw * gi * 0.5f * a.inv_scale;      // dL/d(raw phase) = cP * sign(sin y) cos y  (+ for s and p, - for o)
    const float* rs = a.E + (int64_t)s * 2 * h;
    const float* ro = a.E + (int64_t)o * 2 * h;
    const float* rp = a.Rel + (int64_t)p * 3 * h;
    float* ds = gE + (int64_t)s * 2 * h;
    float* dob = gE + (int64_t)o * 2 * h;
    float* dp = gR + (int64_t)p * 3 * h;
    DropKey ks{0, 0}, kp{0, 0}, ko{0, 0};
    if (DROP) { ks = drop_key(a.seed, t, 0); kp = drop_key(a.seed, t, 1); ko = drop_key(a.seed, t, 2); }
    auto ms_ = [&](int e) { return DROP ? drop_mul(ks, e, a.thr, a.dscale) : 1.f; };
    auto mp_ = [&](int e) { return DROP ? drop_mul(kp, e, a.thr, a.dscale) : 1.f; };
    auto mo_ = [&](int e) { return DROP ? drop_mul(ko, e, a.thr, a.dscale) : 1.f; };
    for (int k = lane; k < h; k += 64) {
      const float fs = ms_(k), fp = mp_(k), fo = mo_(k);
      const float y = 0.5f * a.inv_scale * (rs[k] * fs + rp[k] * fp - ro[k] * fo);
      float sn, cs;
      sincosf(y, &sn, &cs);
      const float dph = cP * (sn > 0.f ? cs : (sn < 0.f ? -cs : 0.f));
      const float fsm = ms_(h + k), fpm = mp_(h + k), fom = mo_(h + k), fpb = mp_(2 * h + k);
      const float mods = rs[h + k] * fsm, modo = ro[h + k] * fom, modp = rp[h + k] * fpm;
      int br;
      const float b = hake_bias(rp[2 * h + k] * fpb, modp, &br);
      const float dM = cM * (mods * (modp + b) - fabsf(modo) * (1.f - b));
      const float db = dM * (mods + fabsf(modo));
      const float sgo = modo > 0.f ? 1.f : (modo < 0.f ? -1.f : 0.f), sgp = modp > 0.f ? 1.f : (modp < 0.f ? -1.f : 0.f);
      const float dms = dM * (modp + b);
      const float dmo = -dM * sgo * (1.f - b);
      const float dmp = dM * mods + (br == 2 ? -sgp * db : 0.f);
      const float dbp = br == 0 ? db : 0.f;
      if (dph != 0.f) {
        if (fs != 0.f) atomicAdd(ds + k, dph * fs);
        if (fp != 0.f) atomicAdd(dp + k, dph * fp);
        if (fo != 0.f) atomicAdd(dob + k, -dph * fo);
      }
      if (dms * fsm != 0.f) atomicAdd(ds + h + k, dms * fsm);
      if (dmo * fom != 0.f) atomicAdd(dob + h + k, dmo * fom);
      if (dmp * fpm != 0.f) atomicAdd(dp + h + k, dmp * fpm);
      if (dbp * fpb != 0.f) atomicAdd(dp + 2 * h + k, dbp * fpb);
    }
  }
}

static int64_t h256(int64_t x) { return (x + 255) & ~(int64_t)255; }

// P | Q | scores | g (n floats each) | r norms (one per segment) | G (2) | chunk == 0 (a training batch of n = m_pos + m_neg rows) only:
// the loss kernels' workspace for those n rows.  The scoring entry point runs no loss: kge_hake_score is called with up to 64 M
// candidate rows per chunk by HAKE.rank / rank_triples, where the loss scratch (pairwise sort planes: ~56 B per row) would be 7 GB.
int64_t hake_workspace_bytes(int64_t n, int64_t chunk) {
  const int64_t nseg = chunk > 0 ? cdiv(n, chunk) : std::max<int64_t>(2, -chunk);  // (chunk < 0: -chunk segments given by offsets)
  const int64_t loss_ws = chunk == 0 ? h256(train_workspace_bytes(nullptr, n, 0)) : 0;  // (the loss workspace depends on m_pos + m_neg only)
  return 4 * h256(n * 4) + h256(nseg * 4) + 256 + loss_ws + 512;
}

static HakeArgs hake_args(const kge_model_desc* md, const kge_hake_params* hp, const float* E, const float* Rel, const int32_t* pos, int64_t m_pos,
                          const int32_t* neg, int64_t m_neg, uint64_t seed, bool training) {
  HakeArgs a;
  a.E = E; a.Rel = Rel; a.pos = pos; a.m_pos = m_pos; a.neg = neg; a.m_neg = m_neg;
  a.N = md->num_entities; a.R = md->num_relations; a.h = md->d_e / 2;
  a.gamma = md->gamma; a.pw = hp->phase_weight; a.mw = hp->mod_weight;
  a.inv_scale = 3.14159265358979323846f / hp->embedding_range;
  const float rate = training ? md->dropout : 0.f;
  a.seed = seed; a.thr = drop_thr(rate); a.dscale = 1.f / (1.f - rate);
  return a;
}

int hake_score(const kge_model_desc* md, const kge_hake_params* hp, const float* E, const float* Rel, const int32_t* triples, int64_t n, int64_t chunk,
               const int64_t* seg_off, int64_t n_seg, float* out, void* workspace, int64_t wbytes, uint32_t* err, hipStream_t st) {
  if (seg_off) chunk = -n_seg;
  else if (chunk <= 0 || chunk > n) chunk = n;  // one call over all rows
  if (wbytes < hake_workspace_bytes(n, chunk)) return fail(KGE_EWORKSPACE, "kge_hake_score: workspace has %lld bytes, needs %lld", (long long)wbytes, (long long)hake_workspace_bytes(n, chunk));
  char* q = reinterpret_cast<char*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  float* P = (float*)q; q += h256(n * 4);
  float* Q = (float*)q; q += 3 * h256(n * 4);
  float* rn = (float*)q;
  const HakeArgs a = hake_args(md, hp, E, Rel, triples, n, triples, 0, 0, false);
  ProfScope prof(KGE_PROF_SCORE, st);
  hipLaunchKernelGGL((hake_rows_kernel<false>), dim3((int)std::min<int64_t>(cdiv(n, 4), kCUs * 16)), dim3(256), 0, st, a, P, Q, err);
  hipLaunchKernelGGL(hake_seg_kernel, dim3((int)(seg_off ? n_seg : cdiv(n, chunk))), dim3(256), 0, st, P, Q, n, 0, chunk, seg_off, a.gamma, a.pw, a.mw, out, rn);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

int hake_rank_counts(const float* scores, int64_t n, const int64_t* q_off, int64_t n_q, int32_t* counts, hipStream_t st) {
  hipLaunchKernelGGL(hake_rank_kernel, dim3((int)n_q), dim3(256), 0, st, scores, q_off, n, counts);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

int hake_train_grads(const kge_model_desc* md, const kge_loss_desc* ld, const kge_hake_params* hp, const float* E, const float* Rel, float* gE, float* gR,
                     const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, float* loss_out, float* scores_out, uint64_t seed,
                     void* workspace, int64_t wbytes, uint32_t* err, hipStream_t st) {
  const int64_t n = m_pos + m_neg;
  if (wbytes < hake_workspace_bytes(n, 0)) return fail(KGE_EWORKSPACE, "kge_hake_train_grads: workspace has %lld bytes, needs %lld", (long long)wbytes, (long long)hake_workspace_bytes(n, 0));
  char* q = reinterpret_cast<char*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  float* P = (float*)q; q += h256(n * 4);
  float* Q = (float*)q; q += h256(n * 4);
  float* sc = (float*)q; q += h256(n * 4);
  float* g = (float*)q; q += h256(n * 4);
  float* rn = (float*)q; q += h256(2 * 4);
  float* G = (float*)q; q += 256;
  void* lws = q;
  const int64_t lbytes = wbytes - (q - reinterpret_cast<char*>(workspace));
  const HakeArgs a = hake_args(md, hp, E, Rel, pos, m_pos, neg, m_neg, seed, true);
  const bool drop = md->dropout > 0.f;
  const int grid = (int)std::min<int64_t>(cdiv(n, 4), kCUs * 16);
  ProfScope prof(KGE_PROF_TRAIN_GRADS, st);
  if (drop) hipLaunchKernelGGL((hake_rows_kernel<true>), dim3(grid), dim3(256), 0, st, a, P, Q, err);
  else hipLaunchKernelGGL((hake_rows_kernel<false>), dim3(grid), dim3(256), 0, st, a, P, Q, err);
  hipLaunchKernelGGL(hake_seg_kernel, dim3(2), dim3(256), 0, st, P, Q, n, m_pos, 0, (const int64_t*)nullptr, a.gamma, a.pw, a.mw, sc, rn);
  KGE_HIP_TRY(hipGetLastError());
  if (int rc = loss_value(ld, sc, m_pos, sc + m_pos, m_neg, loss_out, g, g + m_pos, lws, lbytes, st)) return rc;
  hipLaunchKernelGGL(hake_gsum_kernel, dim3(2), dim3(256), 0, st, g, m_pos, n, G);
  if (drop) hipLaunchKernelGGL((hake_grad_kernel<true>), dim3(grid), dim3(256), 0, st, a, g, G, rn, gE, gR);
  else hipLaunchKernelGGL((hake_grad_kernel<false>), dim3(grid), dim3(256), 0, st, a, g, G, rn, gE, gR);
  if (scores_out) KGE_HIP_TRY(hipMemcpyAsync(scores_out, sc, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

}  // namespace kge
