// The relation-grouped step closed by its pull launch (kge_train_step_swap): pull_kernel<..., TAIL = true> applies the optimizer to
// every entity row in the wave that has just summed that row's gradient, and to the relation rows in the first R workgroups of the
// same launch.  The step is then two launches (fwd_rel_kernel, pull_kernel) and writes no gradient table: against the three-launch
// form it saves the 11.6 MB gradient table written, read and re-zeroed per step at the FB15k-237 shape, and one stream-order gap.
// Reference: the default Keras train step over build_model's wrapper (tests.py:63-89,130-134), optimizer 'adam' / 'sgd' (tests.py:83).
#include "pull_kernel.hpp"

namespace kge {

int pull2_entity_step(const kge_model_desc* md, const float* E, const float* Rel, const PullPlan& e, uint64_t seed, hipStream_t st,
                      const LossFinish* lf, const StepTail* tail) {
  const int N = md->num_entities;
  if (!pull_step_supported(md)) return fail(KGE_EUNSUPPORTED, "step closed by the pull launch: dropout or row width not covered");
  const Shape sh = shape_of(md, E, Rel, tail->E_out, tail->Rel_out);
  if (!sh.v4) return fail(KGE_EINVAL, "tables must be 16-byte aligned");
  {
    ProfScope prof(KGE_PROF_TRAIN_GRADS, st);
    const int pgrid = (int)cdiv(N + e.max_extra, 4);
    if (md->model == KGE_DISTMULT) launch_pull<KGE_DISTMULT, 1, 4, true>(sh.nchunk, pgrid, st, E, Rel, nullptr, nullptr, md, e, seed, N, e.coef, lf, tail);
    else if (md->model == KGE_COMPLEX) launch_pull<KGE_COMPLEX, 1, 4, true>(sh.nchunk, pgrid, st, E, Rel, nullptr, nullptr, md, e, seed, N, e.coef, lf, tail);
    else if (md->norm == 1) launch_pull<KGE_TRANSE, 1, 4, true>(sh.nchunk, pgrid, st, E, Rel, nullptr, nullptr, md, e, seed, N, e.coef, lf, tail);
    else if (md->norm == 3) launch_pull<KGE_TRANSE, 3, 4, true>(sh.nchunk, pgrid, st, E, Rel, nullptr, nullptr, md, e, seed, N, e.coef, lf, tail);
    else launch_pull<KGE_TRANSE, 2, 4, true>(sh.nchunk, pgrid, st, E, Rel, nullptr, nullptr, md, e, seed, N, e.coef, lf, tail);
  }
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

}  // namespace kge
