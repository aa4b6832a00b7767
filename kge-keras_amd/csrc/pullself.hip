// The entity half of a step between twin tables (kge_train_step_swap), self-contained: one wave per destination entity row walks the
// row's list of (relation, other entity, signed multiplicity) entries, gathers the two other rows, forms the triple's score against
// its own row (kept in LDS), takes dL/dscore from it, accumulates the gradient row and applies the optimizer to the row on the spot
// (parameters from the input table, written to the twin table; moments in place).  It needs nothing from the relation-grouped
// forward (no coefficient lookup: the dependent load entry -> coefficient is gone), so the two run side by side; the relation half
// (fwd_rel_kernel + rel_tail_kernel, pull2.hip) forms the same scores again in its own summation order -- for the hinge loss a row
// whose margin - score vanishes to fp32 rounding could be switched on in one half and off in the other; its gradient is bounded by
// 1/len(batch side) times a row product, three orders below the 1e-4 parity tolerance.
// Reference: EmbeddingModel.call x2 + loss + the default Keras train step (KGEkeras/models.py:134-222; utils.py:288-298;
// tests.py:63-89), optimizer 'adam' / 'sgd' with Keras semantics (every row of the table is updated).
#include "pull_self.hpp"

namespace kge {

// The relation rows of the step: workgroup r < R sums the forward's partial rows of relation r and applies the optimizer
// (tail_relation_row); the last workgroup reduces the forward's loss partials (fixed order) and moves the step state on.
__global__ __launch_bounds__(256) void rel_tail_kernel(StepTail t, const float* __restrict__ Rel, LossFinish lf) {
  if ((int)blockIdx.x > t.R) {  // (window steps only) the plan's entity counters outside the window: no wave of the entity half met them
    const int i = ((int)blockIdx.x - t.R - 1) * 1024 + threadIdx.x * 4;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u < t.N && (i + u < t.ent_lo || i + u >= t.ent_hi)) t.ecnt[i + u] = 0;
    return;
  }
  if ((int)blockIdx.x == t.R) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < lf.n; i += 256) s += lf.partials[i];
    const float b = block_sum(s, red);
    if (threadIdx.x == 0) { lf.loss_out[0] = b; step_advance(lf.adv, b); }
    return;
  }
  if ((int)blockIdx.x < t.rel_lo || (int)blockIdx.x >= t.rel_hi) {  // another rank's relation row: only the plan's counter is left zero
    if (threadIdx.x == 0) t.rcnt[blockIdx.x] = 0;
    return;
  }
  tail_relation_row(t, Rel, blockIdx.x);
}

bool pull_step_supported(const kge_model_desc* md) {
  const int width = md->model == KGE_COMPLEX ? md->d_e / 2 : md->d_e;
  return width % 4 == 0 && md->d_r % 4 == 0 && width / 4 <= 64;
}

template <int MODEL, int NORM, int LOSS>
static void launch_self_nv(int nv, bool drop, int grid, hipStream_t st, const SelfArgs& a) {
#define KGE_PS(NVV) do { if (drop) hipLaunchKernelGGL((pull_self_kernel<MODEL, NORM, LOSS, NVV, 16, true>), dim3(grid), dim3(256), 0, st, a); \
                         else hipLaunchKernelGGL((pull_self_kernel<MODEL, NORM, LOSS, NVV, 16, false>), dim3(grid), dim3(256), 0, st, a); } while (0)
  if (nv <= 1) KGE_PS(1);
  else if (nv <= 2) KGE_PS(2);
  else KGE_PS(4);
#undef KGE_PS
}
template <int MODEL, int NORM>
static void launch_self(int loss, int nv, bool drop, int grid, hipStream_t st, const SelfArgs& a) {
  if (loss == KGE_POINTWIZE_HINGE) launch_self_nv<MODEL, NORM, KGE_POINTWIZE_HINGE>(nv, drop, grid, st, a);
  else if (loss == KGE_POINTWIZE_LOGISTIC) launch_self_nv<MODEL, NORM, KGE_POINTWIZE_LOGISTIC>(nv, drop, grid, st, a);
  else if (loss == KGE_POINTWIZE_SQUARE) launch_self_nv<MODEL, NORM, KGE_POINTWIZE_SQUARE>(nv, drop, grid, st, a);
  else launch_self_nv<MODEL, NORM, KGE_POINTWIZE_CROSS_ENTROPY>(nv, drop, grid, st, a);
}

void self_args_of(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, int64_t m_pos, int64_t m_neg, const PullPlan& e,
                  const StepTail& tail, float lr_t, int nchunk, SelfArgs* out) {
  SelfArgs& a = *out;
  a.E = E; a.Rel = Rel;
  a.N = md->num_entities; a.d_e = md->d_e; a.nchunk = nchunk;
  a.inv_pos = 1.f / (float)m_pos; a.inv_neg = 1.f / (float)m_neg;
  a.gamma = md->gamma; a.modulus = md->modulus; a.margin = ld->margin;
  a.off = e.off; a.poff = e.poff; a.totals = e.totals; a.emap = e.emap; a.entries = e.entries; a.slab = e.slab; a.cnt = e.cnt; a.C = e.chunk;
  a.seed = 0; a.thr = drop_thr(md->dropout); a.dscale = 1.f / (1.f - md->dropout);  // (the caller sets the step's seed)
  a.kind = tail.kind; a.lr = lr_t; a.lr_state = nullptr; a.lr_keep = nullptr; a.b1 = tail.b1; a.b2 = tail.b2; a.eps = tail.eps;
  a.E_out = tail.E_out; a.mE = tail.mE; a.vE = tail.vE;
  a.row_lo = tail.ent_lo; a.row_n = tail.ent_hi - tail.ent_lo;
  // tables + moments past the 256 MB Infinity Cache: their streams go non-temporal (SelfArgs.stream_nt)
  a.stream_nt = (int64_t)md->num_entities * md->d_e * 4 * 3 > (256ll << 20) ? 1 : 0;  // (2 -- the parameter rows too -- measured no different from 1)
}

// Entity half: every entity row of E_out, moments in place, entity counters of the plan left zero.  lr_t by value.
int pull2_entity_self_step(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, int64_t m_pos, int64_t m_neg,
                           const PullPlan& e, const StepTail& tail, float lr_t, uint64_t seed, hipStream_t st) {
  if (!pull_step_supported(md)) return fail(KGE_EUNSUPPORTED, "step between twin tables: row width not covered");
  const Shape sh = shape_of(md, E, Rel, tail.E_out, tail.mE);
  if (!sh.v4) return fail(KGE_EINVAL, "tables must be 16-byte aligned");
  SelfArgs a;
  self_args_of(md, ld, E, Rel, m_pos, m_neg, e, tail, lr_t, sh.nchunk, &a);
  a.seed = seed;
  a.lr_state = tail.lr_state;  // (a whole step launch by launch: this launch comes first and keeps lr_t for the closing one)
  a.lr_keep = tail.lr_state ? tail.lr_slot : nullptr;
  const int grid = (int)cdiv(a.row_n + e.max_extra, 4);
  const int nv = (int)cdiv(sh.nchunk, 16);
  {
    ProfScope prof(KGE_PROF_TRAIN_GRADS, st);
    if (md->model == KGE_DISTMULT) launch_self<KGE_DISTMULT, 1>(ld->loss, nv, md->dropout > 0.f, grid, st, a);
    else if (md->model == KGE_COMPLEX) launch_self<KGE_COMPLEX, 1>(ld->loss, nv, md->dropout > 0.f, grid, st, a);
    else if (md->norm == 1) launch_self<KGE_TRANSE, 1>(ld->loss, nv, md->dropout > 0.f, grid, st, a);
    else if (md->norm == 3) launch_self<KGE_TRANSE, 3>(ld->loss, nv, md->dropout > 0.f, grid, st, a);
    else launch_self<KGE_TRANSE, 2>(ld->loss, nv, md->dropout > 0.f, grid, st, a);
  }
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

// The entity gradient rows alone (data-parallel training: the tables are exchanged before the optimizer runs): the same launch with
// the gradient row written to gE instead of the optimizer applied.  Needs nothing from the forward, so the caller may start the
// exchange of gE while the relation half still runs.
int pull2_entity_self_grads(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, int64_t m_pos, int64_t m_neg,
                            const PullPlan& e, float* gE, uint64_t seed, hipStream_t st) {
  StepTail t{};
  step_tail_whole(t, md->num_entities, md->num_relations);
  t.kind = -1;
  t.E_out = gE; t.mE = gE; t.vE = gE;  // (moments unused)
  return pull2_entity_self_step(md, ld, E, Rel, m_pos, m_neg, e, t, 0.f, seed, st);
}

// Relation half, second launch (the first is the relation-grouped forward): relation rows of Rel_out, relation counters left zero,
// loss reduced, step state moved on.
int rel_tail_launch(const StepTail& t, const float* Rel, const LossFinish& lf, hipStream_t st) {
  ProfScope prof(KGE_PROF_OPT, st);
  const bool windowed = t.ecnt && (t.ent_lo > 0 || t.ent_hi < t.N);
  hipLaunchKernelGGL(rel_tail_kernel, dim3(t.R + 1 + (windowed ? (int)cdiv(t.N, 1024) : 0)), dim3(256), 0, st, t, Rel, lf);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

}  // namespace kge
