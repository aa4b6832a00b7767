// One training batch on the GPU: forward on pos and neg, loss, sparse gradients (duplicates summed),
// optimizer update.  Replaces what the default Keras train_step does for the reference's two-input
// functional wrapper (tests.py:63-89,130-134): EmbeddingModel.call x2 (KGEkeras/models.py:134-162),
// loss_function (KGEkeras/utils.py:288-319), autodiff of the three Embedding gathers into IndexedSlices
// and the Keras 'adam' / 'sgd' apply (TensorFlow semantics: SURVEY.md 8(a) rows O1/O2).
//
// "push" gradient path: one kernel gathers the three rows of a triple, reduces the score across the
// wave, turns it into dL/dscore in registers (pointwise losses) and immediately adds the three
// gradient rows into the dense fp32 accumulators with no-return global_atomic_add_f32, shaped as
// 256 contiguous bytes per wave-instruction (lane l owns elements l, l+G, ...).  Hinge rows with a
// zero subgradient issue no atomics at all.
#include "kge_internal.hpp"
#include "pull.hpp"
#include "dropout.hpp"

namespace kge {


// dL/dscore and the loss term of one row for the pointwise losses (utils.py:288-295).
// `inv` = 1/len(true) or 1/len(false) (reduce_mean).  [TF] relu'(0) = 0.

// LOSS < 0: dL/dscore comes from gbuf (two-phase pairwise path) and no loss term is produced here.
template <int MODEL, int NORM, int LOSS, int G, bool DROP>
__global__ __launch_bounds__(256) void push_kernel(const float* __restrict__ E, const float* __restrict__ Rel,
                                                    float* __restrict__ gE, float* __restrict__ gR,
                                                    const int32_t* __restrict__ pos, int64_t m_pos,
                                                    const int32_t* __restrict__ neg, int64_t m_neg, int N, int R, int d_e,
                                                    int d_r, float gamma, float modulus, float margin, const float* __restrict__ gbuf,
                                                    float* __restrict__ scores, float* __restrict__ loss_partials,
                                                    uint32_t* __restrict__ err, uint64_t seed, uint32_t thr, float dscale) {
  __shared__ float red[16];
  const int lane = threadIdx.x & (G - 1);
  const int64_t group = (int64_t)(blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / G;
  const int64_t ngroups = (int64_t)gridDim.x * blockDim.x / G;
  const int64_t n = m_pos + m_neg;
  const int h = d_e >> 1;                               // ComplEx / RotatE half width
  const int width = (MODEL == KGE_COMPLEX || MODEL == KGE_ROTATE) ? h : d_e;  // elements each lane strides over
  const float inv_pos = 1.f / (float)m_pos, inv_neg = 1.f / (float)m_neg;
  float lsum = 0.f;
  for (int64_t t = group; t < n; t += ngroups) {
    const bool is_pos = t < m_pos;
    const int32_t* tr = is_pos ? pos + 3 * t : neg + 3 * (t - m_pos);
    const int s = tr[0], p = tr[1], o = tr[2];
    const bool ok = (unsigned)s < (unsigned)N && (unsigned)p < (unsigned)R && (unsigned)o < (unsigned)N;
    const float* es = E + (int64_t)(ok ? s : 0) * d_e;
    const float* rp = Rel + (int64_t)(ok ? p : 0) * d_r;
    const float* eo = E + (int64_t)(ok ? o : 0) * d_e;
    // inverted-dropout multipliers of element e of the s / p / o row of this batch row (1 when DROP is off)
    DropKey ks{0, 0}, kp{0, 0}, ko{0, 0};
    if (DROP) { ks = drop_key(seed, t, 0); kp = drop_key(seed, t, 1); ko = drop_key(seed, t, 2); }
    auto ms = [&](int e) { return DROP ? drop_mul(ks, e, thr, dscale) : 1.f; };
    auto mp = [&](int e) { return DROP ? drop_mul(kp, e, thr, dscale) : 1.f; };
    auto mo = [&](int e) { return DROP ? drop_mul(ko, e, thr, dscale) : 1.f; };
    // ---- pass 1: score
    float acc = 0.f;
    for (int k = lane; k < width; k += G) {
      if (MODEL == KGE_COMPLEX) {
        const float sr = es[k] * ms(k), si = es[h + k] * ms(h + k), pr = rp[k] * mp(k), pi = rp[h + k] * mp(h + k),
                    orr = eo[k] * mo(k), oi = eo[h + k] * mo(h + k);
        acc += sr * pr * orr + pr * si * oi + pi * sr * oi - pi * si * orr;
      } else if (MODEL == KGE_ROTATE) {  // models.py:439-453
        const float sr = es[k] * ms(k), si = es[h + k] * ms(h + k), orr = eo[k] * mo(k), oi = eo[h + k] * mo(h + k);
        float sn, c;
        sincosf(rp[k] * mp(k), &sn, &c);
        acc += (sr * c - si * sn - orr) + (sr * sn + si * c - oi);
      } else {
        const float sv = es[k] * ms(k), pv = rp[k] * mp(k), ov = eo[k] * mo(k);
        if (MODEL == KGE_DISTMULT) acc += sv * pv * ov;
        else { const float x = sv + pv - ov; acc += NORM == 1 ? fabsf(x) : (NORM == 3 ? phase_abs(x) : (NORM == 5 ? powf(fabsf(x), modulus) : x * x)); }
      }
    }
    acc = group_sum<G>(acc);
    float score = acc, dist = acc;
    if (MODEL == KGE_TRANSE) {
      if (NORM == 2) dist = sqrtf(acc);
      if (NORM == 3) dist = modulus * acc;
      if (NORM == 5) dist = powf(acc, 1.f / modulus);
      score = gamma > 0.f ? gamma - dist : dist;
    }
    if (MODEL == KGE_ROTATE) score = gamma > 0.f ? gamma - acc : acc;
    float g, l = 0.f;
    if (LOSS >= 0) pointwise_term<LOSS>(score, is_pos, margin, is_pos ? inv_pos : inv_neg, g, l);
    else g = gbuf[t];
    if (!ok) { g = 0.f; l = 0.f; score = __builtin_nanf(""); }
    if (lane == 0) {
      lsum += l;
      if (scores) scores[t] = score;
      if (!ok) atomicOr(err, 1u);
    }
    if (g == 0.f) continue;  // wave-uniform per sub-group: no gradient, no atomics
    // ---- pass 2: gradient rows (rows are L1/L2-hot from pass 1)
    float* gs = gE + (int64_t)s * d_e;
    float* gp = gR + (int64_t)p * d_r;
    float* go = gE + (int64_t)o * d_e;
    if (MODEL == KGE_TRANSE) {
      // d score / d x = -d dist/dx (gamma > 0) or +d dist/dx; L2: x/dist, 0 at dist == 0 (TF yields NaN there)
      const float sgn = gamma > 0.f ? -g : g;
      // general p: d dist / d x_k = |x_k|^(p-1) sign(x_k) dist^(1-p)
      const float rn = (NORM == 2) ? (dist > 0.f ? sgn / dist : 0.f) : (NORM == 3 ? 0.5f * modulus * sgn : (NORM == 5 ? (dist > 0.f ? sgn * powf(dist, 1.f - modulus) : 0.f) : sgn));
      for (int k = lane; k < width; k += G) {
        const float a = ms(k), b = mp(k), c = mo(k);
        const float x = es[k] * a + rp[k] * b - eo[k] * c;
        const float dx = NORM == 1 ? (x > 0.f ? rn : (x < 0.f ? -rn : 0.f))
                                   : (NORM == 3 ? rn * phase_dterm(x) : (NORM == 5 ? (x == 0.f ? 0.f : rn * copysignf(powf(fabsf(x), modulus - 1.f), x)) : rn * x));
        atomicAdd(gs + k, dx * a);
        atomicAdd(gp + k, dx * b);
        atomicAdd(go + k, -dx * c);
      }
    } else if (MODEL == KGE_ROTATE) {
      // linear in s and o: d/ds = [cos + sin | cos - sin], d/do = -1, d/dp = re_s (cos - sin) - im_s (cos + sin)  (d atan2(sin p, cos p)/dp = 1)
      const float gg = gamma > 0.f ? -g : g;
      for (int k = lane; k < width; k += G) {
        const float asr = ms(k), asi = ms(h + k), ap = mp(k), aor = mo(k), aoi = mo(h + k);
        const float sr = es[k] * asr, si = es[h + k] * asi;
        float sn, c;
        sincosf(rp[k] * ap, &sn, &c);
        atomicAdd(gs + k, gg * (c + sn) * asr);
        atomicAdd(gs + h + k, gg * (c - sn) * asi);
        atomicAdd(gp + k, gg * (sr * (c - sn) - si * (c + sn)) * ap);
        atomicAdd(go + k, -gg * aor);
        atomicAdd(go + h + k, -gg * aoi);
      }
    } else if (MODEL == KGE_DISTMULT) {
      for (int k = lane; k < width; k += G) {
        const float a = ms(k), b = mp(k), c = mo(k);
        const float sv = es[k] * a, pv = rp[k] * b, ov = eo[k] * c;
        atomicAdd(gs + k, g * pv * ov * a);
        atomicAdd(gp + k, g * sv * ov * b);
        atomicAdd(go + k, g * sv * pv * c);
      }
    } else {  // ComplEx
      for (int k = lane; k < width; k += G) {
        const float asr = ms(k), asi = ms(h + k), apr = mp(k), api = mp(h + k), aor = mo(k), aoi = mo(h + k);
        const float sr = es[k] * asr, si = es[h + k] * asi, pr = rp[k] * apr, pi = rp[h + k] * api, orr = eo[k] * aor, oi = eo[h + k] * aoi;
        atomicAdd(gs + k, g * (pr * orr + pi * oi) * asr);
        atomicAdd(gs + h + k, g * (pr * oi - pi * orr) * asi);
        atomicAdd(gp + k, g * (sr * orr + si * oi) * apr);
        atomicAdd(gp + h + k, g * (sr * oi - si * orr) * api);
        atomicAdd(go + k, g * (sr * pr - pi * si) * aor);
        atomicAdd(go + h + k, g * (pr * si + pi * sr) * aoi);
      }
    }
  }
  if (LOSS >= 0) {
    const float b = block_sum(lsum, red);
    if (threadIdx.x == 0) loss_partials[blockIdx.x] = b;
  }
}

// ---- pairwise losses, all-pairs broadcast (utils.py:300-319 with negative_samples == 1 as called at
// tests.py:81, or len(true) == 1): tmp[j][i] = phi(false_j - true_i); mean over m_f*m_t entries.
// One thread per output element sweeps the other vector through LDS tiles.  O(m_f*m_t) by definition.
template <int LOSS>
__device__ __forceinline__ void pair_term(float z, float margin, float& val, float& dz) {
  if (LOSS == KGE_PAIRWIZE_HINGE) {
    const float y = margin + z;
    val = y > 0.f ? y : 0.f;
    dz = y > 0.f ? 1.f : 0.f;
  } else if (LOSS == KGE_PAIRWIZE_LOGISTIC) {
    const float e = expf(z), den = kLossEps + 1.f + e;
    val = logf(den);
    dz = e / den;
  } else {
    val = -z * z;
    dz = -2.f * z;
  }
}

// side 0: one thread per false j, sweeps true  -> g_f[j] = +sum_i dz / (m_f m_t), loss partials
// side 1: one thread per true  i, sweeps false -> g_t[i] = -sum_j dz / (m_f m_t)
// The sweep over the other vector is cut into gridDim.y column chunks (a row per thread alone leaves the chip idle: m = 20 480
// rows are 80 workgroups): chunk c of row idx leaves its partial sums in part[c][idx] (and lpart[c][idx] for the loss on
// side 0); pairwise_finish_kernel adds the chunks in order.  part == nullptr: one chunk, results written directly.
template <int LOSS, int SIDE>
__global__ __launch_bounds__(256) void pairwise_allpairs_kernel(const float* __restrict__ tscore, int64_t m_t,
                                                                 const float* __restrict__ fscore, int64_t m_f,
                                                                 float margin, float* __restrict__ g_t,
                                                                 float* __restrict__ g_f, float* __restrict__ loss_partials,
                                                                 float* __restrict__ part, float* __restrict__ lpart, int64_t chunk_len) {
  __shared__ float tile[256];
  __shared__ float red[16];
  const float* mine = SIDE == 0 ? fscore : tscore;
  const float* other = SIDE == 0 ? tscore : fscore;
  const int64_t n_mine = SIDE == 0 ? m_f : m_t, n_other = SIDE == 0 ? m_t : m_f;
  const float inv = 1.f / ((float)m_f * (float)m_t);
  const int64_t o0 = part ? (int64_t)blockIdx.y * chunk_len : 0;
  const int64_t o1 = part ? (o0 + chunk_len < n_other ? o0 + chunk_len : n_other) : n_other;
  float lsum = 0.f;
  for (int64_t base = (int64_t)blockIdx.x * 256; base < n_mine; base += (int64_t)gridDim.x * 256) {
    const int64_t idx = base + threadIdx.x;
    const float v = idx < n_mine ? mine[idx] : 0.f;
    float gs = 0.f, ls = 0.f;
    for (int64_t ob = o0; ob < o1; ob += 256) {
      __syncthreads();
      tile[threadIdx.x] = ob + threadIdx.x < o1 ? other[ob + threadIdx.x] : 0.f;
      __syncthreads();
      const int cnt = (int)(o1 - ob < 256 ? o1 - ob : 256);
      for (int q = 0; q < cnt; ++q) {
        const float z = SIDE == 0 ? v - tile[q] : tile[q] - v;
        float val, dz;
        pair_term<LOSS>(z, margin, val, dz);
        gs += dz;
        ls += val;
      }
    }
    if (idx < n_mine) {
      if (part) {
        part[(int64_t)blockIdx.y * n_mine + idx] = gs;
        if (SIDE == 0) lpart[(int64_t)blockIdx.y * n_mine + idx] = ls;
      } else if (SIDE == 0) { g_f[idx] = gs * inv; lsum += ls * inv; }
      else g_t[idx] = -gs * inv;
    }
  }
  if (SIDE == 0 && !part) {
    const float b = block_sum(lsum, red);
    if (threadIdx.x == 0) loss_partials[blockIdx.x] = b;
  }
}

template <int SIDE>
__global__ __launch_bounds__(256) void pairwise_finish_kernel(const float* __restrict__ part, const float* __restrict__ lpart, int nchunk,
                                                               int64_t n_mine, float inv, float* __restrict__ g, float* __restrict__ loss_partials) {
  __shared__ float red[16];
  float lsum = 0.f;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n_mine; idx += (int64_t)gridDim.x * 256) {
    float gs = 0.f, ls = 0.f;
    for (int c = 0; c < nchunk; ++c) {
      gs += part[(int64_t)c * n_mine + idx];
      if (SIDE == 0) ls += lpart[(int64_t)c * n_mine + idx];
    }
    g[idx] = (SIDE == 0 ? gs : -gs) * inv;
    lsum += ls * inv;
  }
  if (SIDE == 0) {
    const float b = block_sum(lsum, red);
    if (threadIdx.x == 0) loss_partials[blockIdx.x] = b;
  }
}

// ---- optimizer sweeps (dense over the whole table, as Keras does for IndexedSlices + 'adam'; for 'sgd' the dense
// sweep equals the sparse scatter because untouched rows have g == 0 exactly).  Also re-zeroes the accumulator and,
// when the L3 regulariser is on (models.py:10-11), adds 3w*x*|x| to the gradient and reduces sum|x|^3 for the loss.
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
// The rest of the gradient of float4 element i of a table whose rows are d4 float4s wide, for a step whose finish kernel was left
// out (OptFinish, pull.hpp): SEG 0 = entity table (hub rows: sum of the slab rows), SEG 1 = relation table (sum of the units' partial
// rows); the thread that owns the first element of a row leaves that row's counter zero.
template <int SEG>
__device__ __forceinline__ float4 finish_grad(const OptFinish& f, int64_t i, int d4, float4 gv) {
  const int row = (int)(i / d4), k4 = (int)(i - (int64_t)row * d4);
  if (SEG == 0) {
    if (k4 == 0) f.ecnt[row] = 0;
    const int c = (row + 1 < f.N ? f.eoff[row + 1] : f.etotals[2]) - f.eoff[row];
    if (c <= f.C) return gv;
    const int nch = (c + f.C - 1) / f.C;
    const float4* src = reinterpret_cast<const float4*>(f.eslab) + (int64_t)f.epoff[row] * d4 + k4;
    for (int j = 0; j < nch; ++j) gv = f4add(gv, src[(int64_t)j * d4]);
    return gv;
  }
  if (k4 == 0) f.rcnt[row] = 0;
  const int c = (row + 1 < f.R ? f.roff[row + 1] : f.rtotal[1]) - f.roff[row];
  const int nch = rel_units(c);
  const float4* src = reinterpret_cast<const float4*>(f.rslab) + (int64_t)f.ruoff[row] * d4 + k4;
  int j = 0;
  for (; j + 4 <= nch; j += 4) {  // four loads in flight, summed in unit order
    const float4 a = src[(int64_t)j * d4], b = src[(int64_t)(j + 1) * d4], c2 = src[(int64_t)(j + 2) * d4], d = src[(int64_t)(j + 3) * d4];
    gv = f4add(f4add(f4add(f4add(gv, a), b), c2), d);
  }
  for (; j < nch; ++j) gv = f4add(gv, src[(int64_t)j * d4]);
  return gv;
}

template <int KIND, bool REG, int SEG = -1>
__device__ __forceinline__ void opt_sweep(float* __restrict__ w, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                          int64_t n, int block, int nblocks, float lr_t, float b1, float b2, float eps, float gscale,
                                          float regw, float* __restrict__ reg_partials, float* red, const OptFinish* fin = nullptr, int d4 = 1) {
  float rsum = 0.f;
  const int64_t stride = (int64_t)nblocks * blockDim.x;
  const int64_t n4 = n >> 2;
  float4* w4 = reinterpret_cast<float4*>(w);
  float4* g4 = reinterpret_cast<float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  auto upd = [&](float& wx, float gx, float& mx, float& vx) {
    gx *= gscale;
    if (REG) {
      const float a = fabsf(wx);
      rsum += a * a * a;
      gx += 3.f * regw * wx * a;
    }
    if (KIND == KGE_OPT_ADAM_TF) {
      mx = b1 * mx + (1.f - b1) * gx;
      vx = b2 * vx + (1.f - b2) * gx * gx;
      wx -= lr_t * mx / (sqrtf(vx) + eps);
    } else {
      wx -= lr_t * gx;
    }
  };
  for (int64_t i = (int64_t)block * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 wv = w4[i], gv = g4[i], mv, vv;
    if (KIND == KGE_OPT_ADAM_TF) { mv = m4[i]; vv = v4[i]; }
    if (SEG >= 0) gv = finish_grad<SEG>(*fin, i, d4, gv);
    upd(wv.x, gv.x, mv.x, vv.x);
    upd(wv.y, gv.y, mv.y, vv.y);
    upd(wv.z, gv.z, mv.z, vv.z);
    upd(wv.w, gv.w, mv.w, vv.w);
    w4[i] = wv;
    g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KIND == KGE_OPT_ADAM_TF) { m4[i] = mv; v4[i] = vv; }
  }
  // scalar tail (n % 4)
  for (int64_t i = (n4 << 2) + (int64_t)block * blockDim.x + threadIdx.x; i < n; i += stride) {
    float wx = w[i], gx = g[i], mx = 0.f, vx = 0.f;
    if (KIND == KGE_OPT_ADAM_TF) { mx = m[i]; vx = v[i]; }
    upd(wx, gx, mx, vx);
    w[i] = wx;
    g[i] = 0.f;
    if (KIND == KGE_OPT_ADAM_TF) { m[i] = mx; v[i] = vx; }
  }
  if (REG) {
    const float b = block_sum(rsum, red);
    if (threadIdx.x == 0) reg_partials[block] = b;
  }
}

// One launch for both tables: blocks [0, grid_a) sweep table a (the entity table; REG applies to it), the rest table b.
struct OptSeg { float *w, *g, *m, *v; int64_t n; };
template <int KIND, bool REG>
__global__ __launch_bounds__(256) void opt_kernel(OptSeg a, OptSeg b, int grid_a, float lr_t, float b1, float b2, float eps, float gscale,
                                                   float regw, float* __restrict__ reg_partials, const float* __restrict__ lr_dev) {
  __shared__ float red[16];
  if (lr_dev) lr_t = lr_dev[0];  // device-resident step counter (graph replay): lr_t of this step
  if ((int)blockIdx.x < grid_a) opt_sweep<KIND, REG>(a.w, a.g, a.m, a.v, a.n, blockIdx.x, grid_a, lr_t, b1, b2, eps, gscale, regw, reg_partials, red);
  else opt_sweep<KIND, false>(b.w, b.g, b.m, b.v, b.n, blockIdx.x - grid_a, gridDim.x - grid_a, lr_t, b1, b2, eps, gscale, 0.f, nullptr, red);
}
// The same sweep closing the gradient half of a relation-grouped step itself (OptFinish; rows of d4 float4s in both tables).
template <int KIND, bool REG>
__global__ __launch_bounds__(256) void opt_finish_kernel(OptSeg a, OptSeg b, int grid_a, float lr_t, float b1, float b2, float eps, float gscale,
                                                          float regw, float* __restrict__ reg_partials, const float* __restrict__ lr_dev,
                                                          OptFinish fin, int d4) {
  __shared__ float red[16];
  if (lr_dev) lr_t = lr_dev[0];
  if ((int)blockIdx.x < grid_a) opt_sweep<KIND, REG, 0>(a.w, a.g, a.m, a.v, a.n, blockIdx.x, grid_a, lr_t, b1, b2, eps, gscale, regw, reg_partials, red, &fin, d4);
  else opt_sweep<KIND, false, 1>(b.w, b.g, b.m, b.v, b.n, blockIdx.x - grid_a, gridDim.x - grid_a, lr_t, b1, b2, eps, gscale, 0.f, nullptr, red, &fin, d4);
}

// loss_out = (accumulate ? loss_out : 0) + scale * sum(partials[0..n)), fixed order -> run-to-run deterministic
__global__ __launch_bounds__(256) void finalize_kernel(const float* __restrict__ partials, int n, float scale,
                                                        int accumulate, float* __restrict__ loss_out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
  const float b = block_sum(s, red);
  if (threadIdx.x == 0) loss_out[0] = (accumulate ? loss_out[0] : 0.f) + scale * b;
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
constexpr int kMaxBlocks = 4096;  // partial-sum slots per kernel

struct Workspace {
  float* loss_partials;  // kMaxBlocks
  float* reg_partials;   // kMaxBlocks
  float* gbuf;           // m_pos + m_neg (pairwise, push path)
  float* sbuf;           // m_pos + m_neg (pairwise, when scores_out is null)
  float* pair;           // all-pairs partial sums (null when the batch is too large for it)
  float* lr_slot;        // learning rate of the running update when the gradient half advances the step state (StepAdvance)
  void* pull_hdr;        // pull-path counters: fixed place, zero on entry, left zero by every step
  void* pull;            // pull-path plan and lists
};

int grad_mode();  // api.hip: 0 = pull (default), 1 = push (float atomics)
struct SideStream { hipStream_t stream; hipEvent_t fork, join; };
int pull_variant();          // api.hip: 2 (default) = relation-grouped forward (pull2.hip), 1 = generic lists (pull.hip)
SideStream* side_stream();  // api.hip: lazily created helper stream (nullptr when disabled by KGE_NO_SIDE_STREAM=1)

// all-pairs pairwise losses: partial sums of up to kPairChunks column chunks per row (two planes for the `false` side: gradient
// and loss), reserved for batches small enough for an O(m_f * m_t) loss to be usable at all
constexpr int kPairChunks = 16;
constexpr int64_t kPairMaxRows = 131072;
// pairsort.hip: pairwize_hinge's all-pairs mean from the two sorted score vectors (no pairs formed); its scratch is carved from the same
// place, so every batch size has what that path needs and batches up to kPairMaxRows also the chunk planes of the all-pairs kernels
int64_t pair_sort_floats(int64_t n);
int pairwise_hinge_sorted(const float* ts, int64_t m_t, const float* fs, int64_t m_f, float margin, float* g_t, float* g_f, float* loss_partials,
                          int* nb, float* scratch, int64_t scratch_floats, hipStream_t st);
int pairwise_square_sums(const float* ts, int64_t m_t, const float* fs, int64_t m_f, float* g_t, float* g_f, float* loss_partials, int* nb,
                         float* scratch, int64_t scratch_floats, hipStream_t st);
// pairlog.hip: pairwize_logistic's all-pairs mean in one tiled pass (every pair once, e^f and e^-t once per score)
int pairwise_logistic_onepass(const float* ts, int64_t m_t, const float* fs, int64_t m_f, float* g_t, float* g_f, float* loss_partials, int* nb,
                              float* scratch, int64_t scratch_floats, hipStream_t st);
static int64_t pair_floats(int64_t n) { return std::max<int64_t>(n <= kPairMaxRows ? 3 * kPairChunks * align_up(n, 64) : 0, pair_sort_floats(n)); }
static int64_t base_floats(int64_t n) { return 2 * kMaxBlocks + 64 + 2 * align_up(n, 64) + pair_floats(n); }

int64_t train_workspace_bytes(const kge_model_desc* md, int64_t m_pos, int64_t m_neg) {
  return (int64_t)sizeof(float) * base_floats(m_pos + m_neg) + 512 +
         (md ? pull_header_bytes(md) + std::max(pull_workspace_bytes(md, m_pos + m_neg), pull2_workspace_bytes(md, m_pos + m_neg)) : 0);
}

static int carve(void* ws, int64_t bytes, const kge_model_desc* md, int64_t m_pos, int64_t m_neg, Workspace* out) {
  if (!ws) return fail(KGE_EINVAL, "workspace is null");
  if (bytes < train_workspace_bytes(md, m_pos, m_neg))
    return fail(KGE_EWORKSPACE, "workspace has %lld bytes, needs %lld", (long long)bytes,
                (long long)train_workspace_bytes(md, m_pos, m_neg));
  uintptr_t p = ((uintptr_t)ws + 255) & ~(uintptr_t)255;
  out->pull_hdr = reinterpret_cast<void*>(p);
  if (md) p += (uintptr_t)align_up(pull_header_bytes(md), 256);
  float* f = reinterpret_cast<float*>(p);
  out->loss_partials = f;
  out->reg_partials = f + kMaxBlocks;
  out->lr_slot = f + 2 * kMaxBlocks;
  out->gbuf = f + 2 * kMaxBlocks + 64;
  out->sbuf = out->gbuf + align_up(m_pos + m_neg, 64);
  out->pair = pair_floats(m_pos + m_neg) ? out->sbuf + align_up(m_pos + m_neg, 64) : nullptr;
  out->pull = f + base_floats(m_pos + m_neg);
  return KGE_OK;
}

// Chained steps (StepChain, pull.hpp): the words between lr_slot and gbuf hold the two-slot lr ring [8..10) and the give-up count [20]; the
// forward workgroups' flags live in the scratch of pull2.hip (timeouts non-null marks the step as chained until they are attached).
static void chain_of(const Workspace& ws, int64_t seq, uint32_t* err, StepChain* ch) {
  ch->lr_ring = ws.lr_slot + 8;
  ch->timeouts = reinterpret_cast<uint32_t*>(ws.lr_slot + 20);
  ch->err = err;
  ch->seq = seq;
}

int score_dispatch(const kge_model_desc* md, const float* E, const float* Rel, const int32_t* triples, int64_t n,
                   float* out, uint32_t* err, hipStream_t st);
int push_group_width(int width);  // tuning.cpp

template <int MODEL, int NORM, int LOSS>
static int launch_push(int G, const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, float* gE,
                       float* gR, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, const float* gbuf,
                       float* scores, float* partials, uint32_t* err, hipStream_t st, int* nblocks, uint64_t seed = 0) {
  const int64_t n = m_pos + m_neg;
  const int64_t want = cdiv(n * G, 256);
  const int grid = (int)(want < kMaxBlocks ? want : kMaxBlocks);
  *nblocks = grid;
  ProfScope prof(KGE_PROF_TRAIN_GRADS, st);
  const bool drop = md->dropout > 0.f;
  const uint32_t thr = drop_thr(md->dropout);
  const float dscale = 1.f / (1.f - md->dropout);
#define KGE_PUSH(GW, DR)                                                                                                \
  hipLaunchKernelGGL((push_kernel<MODEL, NORM, LOSS, GW, DR>), dim3(grid), dim3(256), 0, st, E, Rel, gE, gR, pos, m_pos, \
                     neg, m_neg, md->num_entities, md->num_relations, md->d_e, md->d_r, md->gamma, md->modulus, ld->margin, gbuf, \
                     scores, partials, err, seed, thr, dscale)
  if (drop) { if (G == 16) KGE_PUSH(16, true); else if (G == 32) KGE_PUSH(32, true); else KGE_PUSH(64, true); }
  else { if (G == 16) KGE_PUSH(16, false); else if (G == 32) KGE_PUSH(32, false); else KGE_PUSH(64, false); }
#undef KGE_PUSH
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

template <int LOSS>
static int push_by_model(int G, const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, float* gE,
                         float* gR, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, const float* gbuf,
                         float* scores, float* partials, uint32_t* err, hipStream_t st, int* nb, uint64_t seed = 0) {
  switch (md->model) {
    case KGE_DISTMULT:
      return launch_push<KGE_DISTMULT, 1, LOSS>(G, md, ld, E, Rel, gE, gR, pos, m_pos, neg, m_neg, gbuf, scores, partials, err, st, nb, seed);
    case KGE_TRANSE:
      if (md->norm == 1)
        return launch_push<KGE_TRANSE, 1, LOSS>(G, md, ld, E, Rel, gE, gR, pos, m_pos, neg, m_neg, gbuf, scores, partials, err, st, nb, seed);
      if (md->norm == 3)
        return launch_push<KGE_TRANSE, 3, LOSS>(G, md, ld, E, Rel, gE, gR, pos, m_pos, neg, m_neg, gbuf, scores, partials, err, st, nb, seed);
      if (md->norm == 5)
        return launch_push<KGE_TRANSE, 5, LOSS>(G, md, ld, E, Rel, gE, gR, pos, m_pos, neg, m_neg, gbuf, scores, partials, err, st, nb, seed);
      return launch_push<KGE_TRANSE, 2, LOSS>(G, md, ld, E, Rel, gE, gR, pos, m_pos, neg, m_neg, gbuf, scores, partials, err, st, nb, seed);
    case KGE_COMPLEX:
      return launch_push<KGE_COMPLEX, 1, LOSS>(G, md, ld, E, Rel, gE, gR, pos, m_pos, neg, m_neg, gbuf, scores, partials, err, st, nb, seed);
    case KGE_ROTATE:
      return launch_push<KGE_ROTATE, 1, LOSS>(G, md, ld, E, Rel, gE, gR, pos, m_pos, neg, m_neg, gbuf, scores, partials, err, st, nb, seed);
    default:
      return fail(KGE_EUNSUPPORTED, "kge_train_step: model %d has no HIP training kernel yet", md->model);
  }
}

// Column-aligned broadcast: false.reshape(-1, ns) against true of length ns > 1 (utils.py:301,308,315):
// tmp[i][j] = phi(F[i][j] - true_j), mean over the m_f entries.  SIDE 0: one thread per false element (g_f, loss partials);
// SIDE 1: one thread per column j walks the rows in order (g_t[j] = -sum_i dz / m_f; fixed summation order).
template <int LOSS, int SIDE>
__global__ __launch_bounds__(256) void pairwise_columns_kernel(const float* __restrict__ tscore, int64_t ns, const float* __restrict__ fscore,
                                                                int64_t m_f, float margin, float* __restrict__ g_t,
                                                                float* __restrict__ g_f, float* __restrict__ loss_partials) {
  __shared__ float red[16];
  const float inv = 1.f / (float)m_f;
  if (SIDE == 0) {
    float lsum = 0.f;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < m_f; k += (int64_t)gridDim.x * 256) {
      float val, dz;
      pair_term<LOSS>(fscore[k] - tscore[k % ns], margin, val, dz);
      g_f[k] = dz * inv;
      lsum += val * inv;
    }
    const float b = block_sum(lsum, red);
    if (threadIdx.x == 0) loss_partials[blockIdx.x] = b;
  } else {
    const int64_t rows = m_f / ns;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < ns; j += (int64_t)gridDim.x * 256) {
      const float tj = tscore[j];
      float gs = 0.f;
      for (int64_t i = 0; i < rows; ++i) {
        float val, dz;
        pair_term<LOSS>(fscore[i * ns + j] - tj, margin, val, dz);
        gs += dz;
      }
      g_t[j] = -gs * inv;
    }
  }
}

// NumPy/TF broadcast of false.reshape(-1, ns) (rows, ns) against true (m_t,):
//   ns == 1 or m_t == 1 -> all pairs;   ns == m_t -> column aligned;   anything else is a broadcast error in the reference.
static int pairwise_shape(const kge_loss_desc* ld, int64_t m_t, int64_t m_f) {
  if (ld->negative_samples < 1 || m_f % ld->negative_samples)
    return fail(KGE_ESHAPE, "pairwize loss: len(false)=%lld is not divisible by negative_samples=%d", (long long)m_f, ld->negative_samples);
  if (!(ld->negative_samples == 1 || m_t == 1 || ld->negative_samples == m_t))
    return fail(KGE_ESHAPE, "pairwize loss: (%lld,%d) does not broadcast against (%lld,)", (long long)(m_f / ld->negative_samples),
                ld->negative_samples, (long long)m_t);
  return KGE_OK;
}

// KGE_PAIR_SORT (0 / 1: never / always take the sorted / four-sums forms; unset: by pair count) -- read once; kge_set_pair_sort_mode
// (tests, A/B timing) overrides it.  Note: past the thresholds below the pairwise losses change ALGORITHM, hence summation order, with
// the batch size: all-pairs kernels below 2^27 pairs (hinge) / 2^22 pairs (square), sorted scores + running double sum / four sums
// above -- results agree to fp32 rounding across the switch, not bit for bit.
static int g_pair_sort_mode = -2;
static int pair_sort_mode() {
  if (g_pair_sort_mode == -2) { const char* v = getenv("KGE_PAIR_SORT"); g_pair_sort_mode = v ? atoi(v) : -1; }
  return g_pair_sort_mode;
}
void set_pair_sort_mode(int mode) { g_pair_sort_mode = mode < -1 ? -1 : mode; }

template <int LOSS>
static int launch_pairwise_t(const float* ts, int64_t m_t, const float* fs, int64_t m_f, int64_t ns, float margin, float* g_t, float* g_f,
                             float* partials, float* pair, hipStream_t st, int* nb) {
  const int gf = (int)std::min<int64_t>(cdiv(m_f, 256), kMaxBlocks);
  const int gt = (int)std::min<int64_t>(cdiv(m_t, 256), kMaxBlocks);
  *nb = gf;
  ProfScope prof(KGE_PROF_PAIRWISE, st);
  if (ns == 1 || m_t == 1) {
    // the hinge's all-pairs mean is a function of the order of the scores: past ~1e8 pairs two sorts and a binary search per row are
    // cheaper than the pairs themselves (KGE_PAIR_SORT=0 / 1: never / always)
    const int sort_mode = pair_sort_mode();
    if (LOSS == KGE_PAIRWIZE_HINGE && pair && sort_mode != 0 && (sort_mode == 1 || (double)m_t * (double)m_f >= 134217728.0)) {
      const int rc = pairwise_hinge_sorted(ts, m_t, fs, m_f, margin, g_t, g_f, partials, nb, pair, pair_floats(m_t + m_f), st);
      if (rc != 1) return rc;  // (1: the library calls want more temporary storage than the scratch holds -- the all-pairs kernels run)
    }
    // ... and the square loss separates into four sums (same switch)
    if (LOSS == KGE_PAIRWIZE_SQUARE && pair && sort_mode != 0 && (sort_mode == 1 || (double)m_t * (double)m_f >= 4194304.0)) {
      const int rc = pairwise_square_sums(ts, m_t, fs, m_f, g_t, g_f, partials, nb, pair, pair_floats(m_t + m_f), st);
      if (rc != 1) return rc;
    }
    // ... and the logistic loss, which no order or moment decides, takes every pair ONCE in a tiled pass (pairlog.hip) from 2^22 pairs on
    if (LOSS == KGE_PAIRWIZE_LOGISTIC && pair && sort_mode != 0 && (sort_mode == 1 || (double)m_t * (double)m_f >= 4194304.0)) {
      const int rc = pairwise_logistic_onepass(ts, m_t, fs, m_f, g_t, g_f, partials, nb, pair, pair_floats(m_t + m_f), st);
      if (rc != 1) return rc;
    }
    const bool planes = pair && m_t + m_f <= kPairMaxRows;  // (larger batches: the scratch is the sort path's, one chunk per row here)
    // column chunks of at least 1024 scores, at most kPairChunks of them (scratch: pair = [chunks][m_f] g | [chunks][m_f] loss | [chunks][m_t] g)
    auto chunks = [](int64_t n_other) { return (int)std::max<int64_t>(1, std::min<int64_t>(kPairChunks, n_other / 1024)); };
    const int cf = planes ? chunks(m_t) : 1, ct = planes ? chunks(m_f) : 1;
    const int64_t stride = align_up(m_t + m_f, 64);
    float* pf = planes && cf > 1 ? pair : nullptr;
    float* lf = pf ? pair + kPairChunks * stride : nullptr;
    float* pt = planes && ct > 1 ? pair + 2 * kPairChunks * stride : nullptr;
    const int64_t lenf = align_up(cdiv(m_t, cf), 256), lent = align_up(cdiv(m_f, ct), 256);
    const float inv = 1.f / ((float)m_f * (float)m_t);
    hipLaunchKernelGGL((pairwise_allpairs_kernel<LOSS, 0>), dim3(gf, pf ? (int)cdiv(m_t, lenf) : 1), dim3(256), 0, st, ts, m_t, fs, m_f, margin, g_t, g_f, partials, pf, lf, lenf);
    if (pf) hipLaunchKernelGGL((pairwise_finish_kernel<0>), dim3(gf), dim3(256), 0, st, pf, lf, (int)cdiv(m_t, lenf), m_f, inv, g_f, partials);
    hipLaunchKernelGGL((pairwise_allpairs_kernel<LOSS, 1>), dim3(gt, pt ? (int)cdiv(m_f, lent) : 1), dim3(256), 0, st, ts, m_t, fs, m_f, margin, g_t, g_f, partials, pt, (float*)nullptr, lent);
    if (pt) hipLaunchKernelGGL((pairwise_finish_kernel<1>), dim3(gt), dim3(256), 0, st, pt, (const float*)nullptr, (int)cdiv(m_f, lent), m_t, inv, g_t, partials);
  } else {
    hipLaunchKernelGGL((pairwise_columns_kernel<LOSS, 0>), dim3(gf), dim3(256), 0, st, ts, ns, fs, m_f, margin, g_t, g_f, partials);
    hipLaunchKernelGGL((pairwise_columns_kernel<LOSS, 1>), dim3(gt), dim3(256), 0, st, ts, ns, fs, m_f, margin, g_t, g_f, partials);
  }
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

static int launch_pairwise(const kge_loss_desc* ld, const float* ts, int64_t m_t, const float* fs, int64_t m_f, float* g_t, float* g_f,
                           float* partials, float* pair, hipStream_t st, int* nb) {
  const int64_t ns = ld->negative_samples;
  if (ld->loss == KGE_PAIRWIZE_HINGE) return launch_pairwise_t<KGE_PAIRWIZE_HINGE>(ts, m_t, fs, m_f, ns, ld->margin, g_t, g_f, partials, pair, st, nb);
  if (ld->loss == KGE_PAIRWIZE_LOGISTIC) return launch_pairwise_t<KGE_PAIRWIZE_LOGISTIC>(ts, m_t, fs, m_f, ns, ld->margin, g_t, g_f, partials, pair, st, nb);
  return launch_pairwise_t<KGE_PAIRWIZE_SQUARE>(ts, m_t, fs, m_f, ns, ld->margin, g_t, g_f, partials, pair, st, nb);
}

// reduce_mean=False of the pairwize losses (utils.py:303-305,310-312,317-319): the raw (rows, cols) pair tensor of the
// broadcast, rows = len(false)/ns, cols = max(ns, len(true)).
template <int LOSS>
__global__ __launch_bounds__(256) void pair_tensor_kernel(const float* __restrict__ ts, int64_t m_t, const float* __restrict__ fs, int64_t ns,
                                                           int64_t rows, int64_t cols, float margin, float* __restrict__ out) {
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < rows * cols; k += (int64_t)gridDim.x * 256) {
    const int64_t i = k / cols, j = k - i * cols;
    float val, dz;
    pair_term<LOSS>(fs[i * ns + (ns == 1 ? 0 : j)] - ts[m_t == 1 ? 0 : j], margin, val, dz);
    out[k] = val;
  }
}

int loss_pairs(const kge_loss_desc* ld, const float* ts, int64_t m_t, const float* fs, int64_t m_f, float* out, int64_t out_elems, hipStream_t st) {
  if (ld->loss != KGE_PAIRWIZE_HINGE && ld->loss != KGE_PAIRWIZE_LOGISTIC && ld->loss != KGE_PAIRWIZE_SQUARE)
    return fail(KGE_EINVAL, "kge_loss_pairs: loss %d is not a pairwize loss (the pointwize ones always reduce)", ld->loss);
  if (int rc = pairwise_shape(ld, m_t, m_f)) return rc;
  const int64_t ns = ld->negative_samples, rows = m_f / ns, cols = std::max<int64_t>(ns, m_t);
  if (out_elems != rows * cols) return fail(KGE_ESHAPE, "kge_loss_pairs: output holds %lld floats, the pair tensor is (%lld, %lld)", (long long)out_elems, (long long)rows, (long long)cols);
  const int nb = (int)std::min<int64_t>(cdiv(rows * cols, 256), kMaxBlocks);
  ProfScope prof(KGE_PROF_PAIRWISE, st);
  if (ld->loss == KGE_PAIRWIZE_HINGE) hipLaunchKernelGGL((pair_tensor_kernel<KGE_PAIRWIZE_HINGE>), dim3(nb), dim3(256), 0, st, ts, m_t, fs, ns, rows, cols, ld->margin, out);
  else if (ld->loss == KGE_PAIRWIZE_LOGISTIC) hipLaunchKernelGGL((pair_tensor_kernel<KGE_PAIRWIZE_LOGISTIC>), dim3(nb), dim3(256), 0, st, ts, m_t, fs, ns, rows, cols, ld->margin, out);
  else hipLaunchKernelGGL((pair_tensor_kernel<KGE_PAIRWIZE_SQUARE>), dim3(nb), dim3(256), 0, st, ts, m_t, fs, ns, rows, cols, ld->margin, out);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

// ---- the reference's loss functions as callables on two score vectors (utils.py:288-319): value and, optionally,
// d loss / d true and d loss / d false
template <int LOSS>
__global__ __launch_bounds__(256) void pointwise_loss_kernel(const float* __restrict__ ts, int64_t m_t, const float* __restrict__ fs,
                                                              int64_t m_f, float margin, float* __restrict__ g_t,
                                                              float* __restrict__ g_f, float* __restrict__ partials) {
  __shared__ float red[16];
  const float inv_t = 1.f / (float)m_t, inv_f = 1.f / (float)m_f;
  float lsum = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m_t + m_f; i += (int64_t)gridDim.x * 256) {
    const bool is_pos = i < m_t;
    float g, l;
    pointwise_term<LOSS>(is_pos ? ts[i] : fs[i - m_t], is_pos, margin, is_pos ? inv_t : inv_f, g, l);
    lsum += l;
    if (is_pos) { if (g_t) g_t[i] = g; } else if (g_f) g_f[i - m_t] = g;
  }
  const float b = block_sum(lsum, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = b;
}

int loss_value(const kge_loss_desc* ld, const float* ts, int64_t m_t, const float* fs, int64_t m_f, float* loss_out, float* g_t,
               float* g_f, void* workspace, int64_t wbytes, hipStream_t st) {
  Workspace ws;
  if (int rc = carve(workspace, wbytes, nullptr, m_t, m_f, &ws)) return rc;
  int nb = 0;
  const bool pairwise = ld->loss == KGE_PAIRWIZE_HINGE || ld->loss == KGE_PAIRWIZE_LOGISTIC || ld->loss == KGE_PAIRWIZE_SQUARE;
  if (pairwise) {
    if (int rc = pairwise_shape(ld, m_t, m_f)) return rc;
    float* gt = g_t ? g_t : ws.gbuf;
    float* gf = g_f ? g_f : ws.gbuf + m_t;
    if (int rc = launch_pairwise(ld, ts, m_t, fs, m_f, gt, gf, ws.loss_partials, ws.pair, st, &nb)) return rc;
  } else {
    nb = (int)std::min<int64_t>(cdiv(m_t + m_f, 256), kMaxBlocks);
    if (ld->loss == KGE_POINTWIZE_HINGE) hipLaunchKernelGGL((pointwise_loss_kernel<KGE_POINTWIZE_HINGE>), dim3(nb), dim3(256), 0, st, ts, m_t, fs, m_f, ld->margin, g_t, g_f, ws.loss_partials);
    else if (ld->loss == KGE_POINTWIZE_LOGISTIC) hipLaunchKernelGGL((pointwise_loss_kernel<KGE_POINTWIZE_LOGISTIC>), dim3(nb), dim3(256), 0, st, ts, m_t, fs, m_f, ld->margin, g_t, g_f, ws.loss_partials);
    else if (ld->loss == KGE_POINTWIZE_SQUARE) hipLaunchKernelGGL((pointwise_loss_kernel<KGE_POINTWIZE_SQUARE>), dim3(nb), dim3(256), 0, st, ts, m_t, fs, m_f, ld->margin, g_t, g_f, ws.loss_partials);
    else if (ld->loss == KGE_POINTWIZE_CROSS_ENTROPY) hipLaunchKernelGGL((pointwise_loss_kernel<KGE_POINTWIZE_CROSS_ENTROPY>), dim3(nb), dim3(256), 0, st, ts, m_t, fs, m_f, ld->margin, g_t, g_f, ws.loss_partials);
    else return fail(KGE_EINVAL, "unknown loss %d", ld->loss);
  }
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, st, ws.loss_partials, nb, 1.f, 0, loss_out);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

// ---- forward-only loss of one [pos, neg] batch in inference mode (what Keras reports as val_loss: tests.py:130-134 passes
// validation_data; Model.evaluate runs the wrapper's add_loss with the layers in inference mode): scores of both arrays (score.hip),
// the loss on them, nothing else -- no gradient lists, no gradient rows, no optimizer.  running_sum (optional, device double):
// += the loss, so that a caller sums an epoch of validation batches without reading any of them back.
__global__ void loss_running_sum_kernel(const float* __restrict__ loss, double* __restrict__ sum) { sum[0] += (double)loss[0]; }

int loss_only(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, const int32_t* pos, int64_t m_pos,
              const int32_t* neg, int64_t m_neg, float* loss_out, double* running_sum, void* workspace, int64_t wbytes, uint32_t* err, hipStream_t st) {
  Workspace ws;
  if (int rc = carve(workspace, wbytes, nullptr, m_pos, m_neg, &ws)) return rc;
  if (int rc = score_dispatch(md, E, Rel, pos, m_pos, ws.sbuf, err, st)) return rc;
  if (int rc = score_dispatch(md, E, Rel, neg, m_neg, ws.sbuf + m_pos, err, st)) return rc;
  if (int rc = loss_value(ld, ws.sbuf, m_pos, ws.sbuf + m_pos, m_neg, loss_out, nullptr, nullptr, workspace, wbytes, st)) return rc;
  if (running_sum) hipLaunchKernelGGL(loss_running_sum_kernel, dim3(1), dim3(1), 0, st, loss_out, running_sum);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

// w * sum |E|^3 over n table elements (l3_reg, models.py:10-11,78-92: part of every Keras loss value when regularization != 0), added
// to *loss_inout; block partials summed in a fixed order.
__global__ __launch_bounds__(256) void l3_partials_kernel(const float* __restrict__ E, int64_t n, float* __restrict__ partials) {
  __shared__ float red[16];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const float a = fabsf(E[i]); s += a * a * a; }
  const float b = block_sum(s, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = b;
}
int l3_value(const float* E, int64_t n, float weight, float* loss_inout, void* workspace, int64_t wbytes, hipStream_t st) {
  Workspace ws;
  if (int rc = carve(workspace, wbytes, nullptr, 0, 0, &ws)) return rc;
  const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(n, 1024), kMaxBlocks));
  hipLaunchKernelGGL(l3_partials_kernel, dim3(nb), dim3(256), 0, st, E, n, ws.reg_partials);
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, st, ws.reg_partials, nb, weight, 1, loss_inout);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

int train_grads(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, float* gE, float* gR,
                const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, float* loss_out, float* scores_out,
                uint64_t seed, void* workspace, int64_t wbytes, uint32_t* err, hipStream_t st) {
  Workspace ws;
  if (int rc = carve(workspace, wbytes, md, m_pos, m_neg, &ws)) return rc;
  const bool pairwise = ld->loss == KGE_PAIRWIZE_HINGE || ld->loss == KGE_PAIRWIZE_LOGISTIC || ld->loss == KGE_PAIRWIZE_SQUARE;
  if (pairwise) {
    if (int rc = pairwise_shape(ld, m_pos, m_neg)) return rc;
  } else if (ld->loss != KGE_POINTWIZE_HINGE && ld->loss != KGE_POINTWIZE_LOGISTIC && ld->loss != KGE_POINTWIZE_SQUARE &&
             ld->loss != KGE_POINTWIZE_CROSS_ENTROPY) {
    return fail(KGE_EINVAL, "unknown loss %d", ld->loss);
  }
  int nb = 0;
  if (grad_mode() == 0 && pull_variant() == 2 && pull2_supported(md, ld)) {
    return train_grads_pull2(md, ld, E, Rel, gE, gR, pos, m_pos, neg, m_neg, scores_out, ws.loss_partials, &nb, loss_out, ws.pull_hdr, ws.pull, err, seed, st);
  }
  if ((grad_mode() == 0 || md->model == KGE_HOLE) && pull_supported(md)) {
    // The index build depends on the ids only and the forward gather does not need it: fork it onto the library's side
    // stream and join before the list fill (event fork/join, capturable into a hipGraph).
    PullPlan plan;
    SideStream* ss = side_stream();
    if (ss) {
      KGE_HIP_TRY(hipEventRecord(ss->fork, st));
      KGE_HIP_TRY(hipStreamWaitEvent(ss->stream, ss->fork, 0));
      if (int rc = pull_prepare(md, pos, m_pos, neg, m_neg, ws.pull_hdr, ws.pull, ss->stream, &plan)) return rc;
      KGE_HIP_TRY(hipEventRecord(ss->join, ss->stream));
    } else {
      if (int rc = pull_prepare(md, pos, m_pos, neg, m_neg, ws.pull_hdr, ws.pull, st, &plan)) return rc;
    }
    float* sc = scores_out ? scores_out : (pairwise ? ws.sbuf : nullptr);
    if (int rc = pull_forward(md, pairwise ? -1 : ld->loss, ld->margin, E, Rel, pos, m_pos, neg, m_neg, plan, sc, ws.loss_partials, &nb, err, seed, st)) return rc;
    if (ss) KGE_HIP_TRY(hipStreamWaitEvent(st, ss->join, 0));
    if (pairwise) {
      if (int rc = launch_pairwise(ld, sc, m_pos, sc + m_pos, m_neg, plan.coef, plan.coef + m_pos, ws.loss_partials, ws.pair, st, &nb)) return rc;
    }
    if (int rc = pull_backward(md, E, Rel, gE, gR, pos, m_pos, neg, m_neg, plan, pairwise, sc, seed, st)) return rc;
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, st, ws.loss_partials, nb, 1.f, 0, loss_out);
    KGE_HIP_TRY(hipGetLastError());
    return KGE_OK;
  }
  if (pairwise && md->dropout > 0.f)
    return fail(KGE_EUNSUPPORTED, "pairwize loss with dropout needs the pull gradient path (kge_set_grad_mode(0))");
  const int width = (md->model == KGE_COMPLEX || md->model == KGE_ROTATE) ? md->d_e / 2 : md->d_e;
  const int G = push_group_width(width);
  switch (ld->loss) {
    case KGE_POINTWIZE_HINGE:
      if (int rc = push_by_model<KGE_POINTWIZE_HINGE>(G, md, ld, E, Rel, gE, gR, pos, m_pos, neg, m_neg, nullptr, scores_out, ws.loss_partials, err, st, &nb, seed)) return rc;
      break;
    case KGE_POINTWIZE_LOGISTIC:
      if (int rc = push_by_model<KGE_POINTWIZE_LOGISTIC>(G, md, ld, E, Rel, gE, gR, pos, m_pos, neg, m_neg, nullptr, scores_out, ws.loss_partials, err, st, &nb, seed)) return rc;
      break;
    case KGE_POINTWIZE_SQUARE:
      if (int rc = push_by_model<KGE_POINTWIZE_SQUARE>(G, md, ld, E, Rel, gE, gR, pos, m_pos, neg, m_neg, nullptr, scores_out, ws.loss_partials, err, st, &nb, seed)) return rc;
      break;
    case KGE_POINTWIZE_CROSS_ENTROPY:
      if (int rc = push_by_model<KGE_POINTWIZE_CROSS_ENTROPY>(G, md, ld, E, Rel, gE, gR, pos, m_pos, neg, m_neg, nullptr, scores_out, ws.loss_partials, err, st, &nb, seed)) return rc;
      break;
    case KGE_PAIRWIZE_HINGE:
    case KGE_PAIRWIZE_LOGISTIC:
    case KGE_PAIRWIZE_SQUARE: {
      float* sc = scores_out ? scores_out : ws.sbuf;
      // phase 1: scores of pos then neg
      if (int rc = score_dispatch(md, E, Rel, pos, m_pos, sc, err, st)) return rc;
      if (int rc = score_dispatch(md, E, Rel, neg, m_neg, sc + m_pos, err, st)) return rc;
      // phase 2: dL/dscore for every row + loss
      int rc;
      if ((rc = launch_pairwise(ld, sc, m_pos, sc + m_pos, m_neg, ws.gbuf, ws.gbuf + m_pos, ws.loss_partials, ws.pair, st, &nb))) return rc;
      // phase 3: gradient rows
      int nb2 = 0;
      if ((rc = push_by_model<-1>(G, md, ld, E, Rel, gE, gR, pos, m_pos, neg, m_neg, ws.gbuf, nullptr, nullptr, err, st, &nb2, seed))) return rc;
      break;
    }
    default:
      return fail(KGE_EINVAL, "unknown loss %d", ld->loss);
  }
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, st, ws.loss_partials, nb, 1.f, 0, loss_out);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

// Device-resident optimizer step state (kge_opt_desc.step_dev; layout: kge_internal.hpp).  Advanced by one tiny kernel at the end of
// every kge_opt_apply.
__global__ void opt_state_init_kernel(int64_t* step, int64_t done, int kind, float lr, float b1, float b2) {
  step[0] = done;
  step_state_lr(step)[0] = lr_of_step(kind, lr, b1, b2, done + 1);
  step_state_lr(step)[1] = 0.f;
  step_state_loss(step)[0] = 0.0;
}
__global__ void opt_state_post_kernel(int64_t* step, int kind, float lr, float b1, float b2, const float* loss) {
  const int64_t t = step[0] + 1;
  step[0] = t;
  step_state_lr(step)[0] = lr_of_step(kind, lr, b1, b2, t + 1);
  if (loss) step_state_loss(step)[0] += (double)loss[0];
}

int opt_state_init(const kge_opt_desc* od, hipStream_t st) {
  hipLaunchKernelGGL(opt_state_init_kernel, dim3(1), dim3(1), 0, st, od->step_dev, od->step - 1, od->kind, od->lr, od->beta1, od->beta2);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

template <int KIND>
static int launch_opt(bool reg, OptSeg a, OptSeg b, float lr_t, const kge_opt_desc* od, float gscale, float regw, float* reg_partials,
                      hipStream_t st, int* nb, const float* lr_slot = nullptr, const OptFinish* fin = nullptr, int d4 = 1) {
  // device-resident step state: lr_t of this update comes from the state (prepared by the previous update's post kernel), or from
  // the slot the gradient half just wrote (StepAdvance)
  const float* lr_dev = lr_slot ? lr_slot : (od->step_dev ? reinterpret_cast<const float*>(od->step_dev + 1) : nullptr);
  auto blocks = [](int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(cdiv(n, 4), 256), kCUs * 8)); };
  const int ga = blocks(a.n), gb = b.n > 0 ? blocks(b.n) : 0;
  *nb = ga;
  if (fin) {
    if (reg) hipLaunchKernelGGL((opt_finish_kernel<KIND, true>), dim3(ga + gb), dim3(256), 0, st, a, b, ga, lr_t, od->beta1, od->beta2, od->eps, gscale, regw, reg_partials, lr_dev, *fin, d4);
    else hipLaunchKernelGGL((opt_finish_kernel<KIND, false>), dim3(ga + gb), dim3(256), 0, st, a, b, ga, lr_t, od->beta1, od->beta2, od->eps, gscale, regw, reg_partials, lr_dev, *fin, d4);
    KGE_HIP_TRY(hipGetLastError());
    return KGE_OK;
  }
  if (reg) hipLaunchKernelGGL((opt_kernel<KIND, true>), dim3(ga + gb), dim3(256), 0, st, a, b, ga, lr_t, od->beta1, od->beta2, od->eps, gscale, regw, reg_partials, lr_dev);
  else hipLaunchKernelGGL((opt_kernel<KIND, false>), dim3(ga + gb), dim3(256), 0, st, a, b, ga, lr_t, od->beta1, od->beta2, od->eps, gscale, regw, reg_partials, lr_dev);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

static float host_lr_t(const kge_opt_desc* od) {
  if (od->kind != KGE_OPT_ADAM_TF) return od->lr;
  const double t = (double)od->step;
  return (float)((double)od->lr * std::sqrt(1.0 - std::pow((double)od->beta2, t)) / (1.0 - std::pow((double)od->beta1, t)));
}

bool plan_supported(const kge_model_desc* md, const kge_loss_desc* ld) {
  return grad_mode() == 0 && pull_variant() == 2 && pull2_supported(md, ld);
}

// advance != null (a whole step in one call, no L3 term): the closing kernel of the gradient half advances the step state of
// that optimizer descriptor (StepAdvance); the optimizer half is then opt_apply(..., advanced = true).
// finish != null (a whole step in one call): the finish kernel is left out and *finish says what the optimizer sweep has to do in
// its place (OptFinish); the optimizer half is then opt_apply(..., finish).
int train_grads_with_plan(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, float* gE, float* gR, void* plan,
                          int64_t m_pos, int64_t m_neg, float* loss_out, uint64_t seed, void* workspace, int64_t wbytes, hipStream_t st,
                          const kge_opt_desc* advance, OptFinish* finish, int grad_parts) {
  Workspace ws;
  if (int rc = carve(workspace, wbytes, md, m_pos, m_neg, &ws)) return rc;
  int nb = 0;
  StepAdvance adv{nullptr, nullptr, 0, 0.f, 0.f, 0.f};
  if (advance) adv = StepAdvance{advance->step_dev, ws.lr_slot, advance->kind, advance->lr, advance->beta1, advance->beta2};
  return train_grads_planned(md, ld, E, Rel, gE, gR, plan, m_pos, m_neg, ws.loss_partials, &nb, loss_out, ws.pull, seed, st, adv, finish, nullptr, grad_parts);
}

// The step between twin tables (step1.hip, pullself.hip): parameters read from (E_in, Rel_in), written to (E_out, Rel_out); moments in
// place; no gradient tables.  Needs the device-resident step state and no L3 term.
bool step_swap_supported(const kge_model_desc* md, const kge_loss_desc* ld, const kge_opt_desc* od) {
  return plan_supported(md, ld) && pull_step_supported(md) && od->step_dev && ld->regularization == 0.f &&
         (od->kind == KGE_OPT_SGD || (od->kind == KGE_OPT_ADAM_TF && od->mE && od->vE && od->mRel && od->vRel));
}
int train_step_swap(const kge_model_desc* md, const kge_loss_desc* ld, const kge_opt_desc* od, const float* E_in, const float* Rel_in, float* E_out,
                    float* Rel_out, void* plan, int64_t m_pos, int64_t m_neg, float* loss_out, uint64_t seed, void* workspace, int64_t wbytes,
                    int parts, hipStream_t st, const kge_row_window* win, int64_t chain_seq, uint32_t* chain_err) {
  Workspace ws;
  if (int rc = carve(workspace, wbytes, md, m_pos, m_neg, &ws)) return rc;
  int nb = 0;
  StepAdvance adv{od->step_dev, ws.lr_slot, od->kind, od->lr, od->beta1, od->beta2};
  StepTail t{};
  step_tail_whole(t, md->num_entities, md->num_relations);
  if (win) { t.ent_lo = win->ent_lo; t.ent_hi = win->ent_hi; t.rel_lo = win->rel_lo; t.rel_hi = win->rel_hi; }
  t.kind = od->kind;
  t.E_out = E_out; t.mE = od->mE; t.vE = od->vE;
  t.Rel_out = Rel_out; t.mR = od->mRel; t.vR = od->vRel;
  t.b1 = od->beta1; t.b2 = od->beta2; t.eps = od->eps;
  t.parts = parts;
  if (chain_seq >= 0) chain_of(ws, chain_seq, chain_err, &t.ch);
  if (parts == 3) {
    // the whole step on one stream: lr_t of this update is read from the step state (prepared by the previous update, or by
    // kge_opt_state_init) -- nothing of the step is a host value, so a captured step replays right.  The first launch copies it into
    // the workspace slot the closing launch reads, whose last workgroup meanwhile moves the state on.
    t.lr_state = reinterpret_cast<const float*>(od->step_dev + 1);
    t.lr_slot = ws.lr_slot;
    adv.lr_slot = nullptr;
  } else {
    // the halves may be issued apart, even side by side on two streams, so lr_t of this update travels by value: od->step is its number
    if (od->step < 1) return fail(KGE_EINVAL, "kge_train_step_swap: od->step (the number of this update, 1-based) must be >= 1 when the halves are issued apart");
    t.lr_t = host_lr_t(od);
  }
  return train_grads_planned(md, ld, E_in, Rel_in, nullptr, nullptr, plan, m_pos, m_neg, ws.loss_partials, &nb, loss_out, ws.pull, seed, st, adv, nullptr, &t);
}

int opt_apply_vector(const kge_opt_desc* od, float* w, float* g, int64_t n, float gscale, hipStream_t st) {
  int nb = 0;
  if (od->step_dev) return fail(KGE_EUNSUPPORTED, "kge_opt_apply_vector: device-resident step counters cover kge_opt_apply only");
  const OptSeg none{nullptr, nullptr, nullptr, nullptr, 0};
  if (od->kind == KGE_OPT_ADAM_TF) {
    if (!od->mE || !od->vE) return fail(KGE_EINVAL, "adam: moment buffers (mE, vE) are null");
    if (od->step < 1) return fail(KGE_EINVAL, "adam: step must be >= 1");
    return launch_opt<KGE_OPT_ADAM_TF>(false, OptSeg{w, g, od->mE, od->vE, n}, none, host_lr_t(od), od, gscale, 0.f, nullptr, st, &nb);
  }
  if (od->kind == KGE_OPT_SGD) return launch_opt<KGE_OPT_SGD>(false, OptSeg{w, g, nullptr, nullptr, n}, none, od->lr, od, gscale, 0.f, nullptr, st, &nb);
  return fail(KGE_EINVAL, "unknown optimizer %d", od->kind);
}

int opt_apply(const kge_model_desc* md, const kge_loss_desc* ld, const kge_opt_desc* od, float* E, float* Rel, float* gE,
              float* gR, float gscale, float* loss_out, void* workspace, int64_t wbytes, hipStream_t st, bool advanced, const OptFinish* finish) {
  Workspace ws;
  if (int rc = carve(workspace, wbytes, md, 0, 0, &ws)) return rc;
  const float* lr_slot = advanced ? ws.lr_slot : nullptr;
  const int d4 = md->d_e / 4;
  const int64_t nE = (int64_t)md->num_entities * md->d_e, nR = (int64_t)md->num_relations * md->d_r;
  const bool reg = ld->regularization != 0.f;
  int nb = 0;
  {
    ProfScope prof(KGE_PROF_OPT, st);
    if (od->kind == KGE_OPT_ADAM_TF) {
      if (!od->mE || !od->vE || !od->mRel || !od->vRel) return fail(KGE_EINVAL, "adam: moment buffers are null");
      if (!od->step_dev && od->step < 1) return fail(KGE_EINVAL, "adam: step must be >= 1");
      if (int rc = launch_opt<KGE_OPT_ADAM_TF>(reg, OptSeg{E, gE, od->mE, od->vE, nE}, OptSeg{Rel, gR, od->mRel, od->vRel, nR}, host_lr_t(od), od, gscale,
                                               ld->regularization, ws.reg_partials, st, &nb, lr_slot, finish, d4)) return rc;
    } else if (od->kind == KGE_OPT_SGD) {
      if (int rc = launch_opt<KGE_OPT_SGD>(reg, OptSeg{E, gE, nullptr, nullptr, nE}, OptSeg{Rel, gR, nullptr, nullptr, nR}, od->lr, od, gscale,
                                           ld->regularization, ws.reg_partials, st, &nb, lr_slot, finish, d4)) return rc;
    } else {
      return fail(KGE_EINVAL, "unknown optimizer %d", od->kind);
    }
  }
  if (reg && loss_out) hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, st, ws.reg_partials, nb, ld->regularization, 1, loss_out);
  if (od->step_dev && !advanced) hipLaunchKernelGGL(opt_state_post_kernel, dim3(1), dim3(1), 0, st, od->step_dev, od->kind, od->lr, od->beta1, od->beta2, loss_out);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

}  // namespace kge
