/* oracle/kge_oracle_c.c -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Plain-C restatement of one reference training batch, used as (a) a fast checker at full BASELINE sizes and (b) the
 * `cpu_baseline` leg of bench.py (OpenMP over the host cores).  Same algorithm as oracle/kge_oracle.py (which cites
 * the reference lines): EmbeddingModel.call x2 (KGEkeras/models.py:134-162) with DistMult (:171-172), TransE
 * (:186-190) or ComplEx (:212-222); pointwize_hinge / pointwize_logistic (KGEkeras/utils.py:288-292); the dense
 * equivalents of the IndexedSlices gradients (duplicates summed) and Keras 'adam' / 'sgd' [TF semantics].
 * Arithmetic is fp32 like the reference's graph; gradient accumulation is per-thread private then reduced in a fixed
 * order, so results do not depend on the thread count beyond fp32 summation order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EPSILON 1e-6f

static float score_row(int model, int norm, float gamma, const float* s, const float* p, const float* o, int d) {
  float acc = 0.f;
  if (model == 0) {
    for (int k = 0; k < d; ++k) acc += s[k] * p[k] * o[k];
    return acc;
  }
  if (model == 1) {
    for (int k = 0; k < d; ++k) { float x = s[k] + p[k] - o[k]; acc += norm == 1 ? fabsf(x) : x * x; }
    if (norm == 2) acc = sqrtf(acc);
    return gamma > 0.f ? gamma - acc : acc;
  }
  const int h = d / 2;
  for (int k = 0; k < h; ++k) {
    const float sr = s[k], si = s[h + k], pr = p[k], pi = p[h + k], orr = o[k], oi = o[h + k];
    acc += sr * pr * orr + pr * si * oi + pi * sr * oi - pi * si * orr;
  }
  return acc;
}

/* loss: 0 = pointwize_hinge, 1 = pointwize_logistic.  Returns the batch loss; gE/gR must be zero on entry. */
float kge_oracle_grads(int model, int norm, float gamma, int loss, float margin, const float* E, const float* Rel, int N, int R,
                       int d, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, float* gE, float* gR,
                       float* scores) {
  const int64_t n = m_pos + m_neg;
  double total = 0.0;
  int nt = 1;
#ifdef _OPENMP
  nt = omp_get_max_threads();
#endif
  float** pe = (float**)calloc(nt, sizeof(float*));
  float** pr = (float**)calloc(nt, sizeof(float*));
#pragma omp parallel reduction(+ : total)
  {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    float* le = (float*)calloc((size_t)N * d, sizeof(float));
    float* lr = (float*)calloc((size_t)R * d, sizeof(float));
    pe[tid] = le;
    pr[tid] = lr;
#pragma omp for schedule(static)
    for (int64_t t = 0; t < n; ++t) {
      const int is_pos = t < m_pos;
      const int32_t* tr = is_pos ? pos + 3 * t : neg + 3 * (t - m_pos);
      const float *s = E + (size_t)tr[0] * d, *p = Rel + (size_t)tr[1] * d, *o = E + (size_t)tr[2] * d;
      const float sc = score_row(model, norm, gamma, s, p, o, d);
      if (scores) scores[t] = sc;
      const float inv = 1.f / (float)(is_pos ? m_pos : m_neg);
      float g, l;
      if (loss == 0) {
        const float a = is_pos ? margin - sc : margin + sc;
        l = a > 0.f ? a * inv : 0.f;
        g = a > 0.f ? (is_pos ? -inv : inv) : 0.f;
      } else {
        const float e = expf(is_pos ? -sc : sc), den = EPSILON + 1.f + e;
        l = logf(den) * inv;
        g = (is_pos ? -e : e) / den * inv;
      }
      total += l;
      if (g == 0.f) continue;
      float *ds = le + (size_t)tr[0] * d, *dp = lr + (size_t)tr[1] * d, *dobj = le + (size_t)tr[2] * d;
      if (model == 0) {
        for (int k = 0; k < d; ++k) { ds[k] += g * p[k] * o[k]; dp[k] += g * s[k] * o[k]; dobj[k] += g * s[k] * p[k]; }
      } else if (model == 1) {
        float dist = 0.f;
        if (norm == 2) { for (int k = 0; k < d; ++k) { float x = s[k] + p[k] - o[k]; dist += x * x; } dist = sqrtf(dist); }
        const float c = (gamma > 0.f ? -g : g) / (norm == 2 ? (dist > 0.f ? dist : INFINITY) : 1.f);
        for (int k = 0; k < d; ++k) {
          const float x = s[k] + p[k] - o[k];
          const float dx = norm == 1 ? c * (x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f)) : c * x;
          ds[k] += dx; dp[k] += dx; dobj[k] -= dx;
        }
      } else {
        const int h = d / 2;
        for (int k = 0; k < h; ++k) {
          const float sr = s[k], si = s[h + k], pr_ = p[k], pi = p[h + k], orr = o[k], oi = o[h + k];
          ds[k] += g * (pr_ * orr + pi * oi);     ds[h + k] += g * (pr_ * oi - pi * orr);
          dp[k] += g * (sr * orr + si * oi);      dp[h + k] += g * (sr * oi - si * orr);
          dobj[k] += g * (sr * pr_ - pi * si);    dobj[h + k] += g * (pr_ * si + pi * sr);
        }
      }
    }
  }
  /* fixed-order reduction of the per-thread tables */
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)N * d; ++i) { float a = 0.f; for (int t = 0; t < nt; ++t) if (pe[t]) a += pe[t][i]; gE[i] += a; }
  for (int64_t i = 0; i < (int64_t)R * d; ++i) { float a = 0.f; for (int t = 0; t < nt; ++t) if (pr[t]) a += pr[t][i]; gR[i] += a; }
  for (int t = 0; t < nt; ++t) { free(pe[t]); free(pr[t]); }
  free(pe); free(pr);
  return (float)total;
}

/* Keras 'adam' (kind 1; dense, eps outside the bias correction) or 'sgd' (kind 0); zeroes g. */
void kge_oracle_apply(int kind, float* w, float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, int64_t step) {
  float lr_t = lr;
  if (kind == 1) lr_t = (float)((double)lr * sqrt(1.0 - pow((double)b2, (double)step)) / (1.0 - pow((double)b1, (double)step)));
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const float gx = g[i];
    if (kind == 1) {
      m[i] = b1 * m[i] + (1.f - b1) * gx;
      v[i] = b2 * v[i] + (1.f - b2) * gx * gx;
      w[i] -= lr_t * m[i] / (sqrtf(v[i]) + eps);
    } else {
      w[i] -= lr_t * gx;
    }
    g[i] = 0.f;
  }
}

/* cap the OpenMP team (per-thread private gradient tables make very wide teams counter-productive) */
void kge_oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int kge_oracle_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
