// extern "C" surface of libkge_hip.so: argument validation, error text, dispatch.  See include/kge_hip.h.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "kge_internal.hpp"
#include "pull.hpp"

namespace kge {

// (err_buf / fail / kge_last_error: errors.cpp -- host-only, shared with the sanitizer build of the host objects)

// hipGetLastError() is per thread and sticky across callers: an error some other library left behind (PyTorch probing the
// platform at start-up has been seen to leave "no ROCm-capable device") must not be reported as ours after the next launch.
static inline void clear_stale_error() { (void)hipGetLastError(); }

int check_model(const kge_model_desc* md, const char* who) {
  clear_stale_error();
  if (!md) return fail(KGE_EINVAL, "%s: model descriptor is null", who);
  if (md->model < KGE_DISTMULT || md->model > KGE_HAKE) return fail(KGE_EINVAL, "%s: unknown model %d", who, md->model);
  if (md->num_entities <= 0 || md->num_relations <= 0 || md->d_e <= 0 || md->d_r <= 0)
    return fail(KGE_EINVAL, "%s: non-positive table shape (%d,%d) (%d,%d)", who, md->num_entities, md->d_e, md->num_relations, md->d_r);
  // s*p*o, s+p-o and the ComplEx products broadcast row-against-row: TensorFlow rejects unequal widths
  if (md->model == KGE_ROTATE) {  // rotation phases: one per complex component (models.py:428-429: e_dim doubled, r_dim not)
    if (md->d_e != 2 * md->d_r) return fail(KGE_ESHAPE, "%s: RotatE needs e_dim %d == 2 * r_dim %d (tf.split halves against the phase row)", who, md->d_e, md->d_r);
  } else if (md->model == KGE_HAKE) {  // [phase | modulus] against [phase | modulus | bias] (models.py:371-372,387-389)
    if ((md->d_e & 1) || 3 * (md->d_e / 2) != md->d_r) return fail(KGE_ESHAPE, "%s: HAKE needs e_dim %d = 2h and r_dim %d = 3h", who, md->d_e, md->d_r);
  } else if (md->d_e != md->d_r) return fail(KGE_ESHAPE, "%s: e_dim %d and r_dim %d do not broadcast", who, md->d_e, md->d_r);
  if (md->model == KGE_COMPLEX && (md->d_e & 1)) return fail(KGE_ESHAPE, "%s: ComplEx rows must split in two halves (tf.split), width %d", who, md->d_e);
  if (md->model == KGE_TRANSE && md->norm != 1 && md->norm != 2 && md->norm != 3 && md->norm != 5)
    return fail(KGE_EUNSUPPORTED, "%s: TransE norm id %d (1, 2, the pRotatE phase distance 3 and the general p-norm 5 are built)", who, md->norm);
  if (md->model == KGE_TRANSE && md->norm == 5 && !(md->modulus > 0.f))
    return fail(KGE_EINVAL, "%s: the general p-norm (norm id 5) takes its exponent p > 0 in kge_model_desc.modulus (got %f)", who, md->modulus);
  if (!(md->dropout >= 0.f && md->dropout < 1.f)) return fail(KGE_EINVAL, "%s: dropout rate %f outside [0,1)", who, md->dropout);
  return KGE_OK;
}

// ---- kernel timing hooks
namespace {
struct ProfState {
  bool on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[KGE_PROF_NLABELS];
  hipEvent_t pending[KGE_PROF_NLABELS] = {};
};
ProfState g_prof;
}  // namespace

bool prof_is_on() { return g_prof.on; }

void prof_begin(int label, hipStream_t st) {
  if (!g_prof.on) return;
  hipEvent_t a;
  if (hipEventCreate(&a) != hipSuccess) return;
  (void)hipEventRecord(a, st);
  g_prof.pending[label] = a;
}

void prof_end(int label, hipStream_t st) {
  if (!g_prof.on || !g_prof.pending[label]) return;
  hipEvent_t b;
  if (hipEventCreate(&b) != hipSuccess) return;
  (void)hipEventRecord(b, st);
  g_prof.ev[label].emplace_back(g_prof.pending[label], b);
  g_prof.pending[label] = nullptr;
}

// ---- tuning constants (what measured fastest; the environment switches of rounds 1-3 that selected the slower alternatives are gone:
// profiles/r0*_experiments.txt keep their numbers) and the few switches that select a code PATH (tests and A/B timing use them)
static int env_int(const char* name, int dflt) {
  const char* v = std::getenv(name);
  return v && *v ? std::atoi(v) : dflt;
}

int score_group_width(int nchunk) {
  int p2 = 1;
  while (p2 < nchunk) p2 <<= 1;
  int g = p2 / 4;
  return g < 4 ? 4 : (g > 64 ? 64 : g);
}

int fwd_group_width(int nchunk) {
  return score_group_width(nchunk);
}

int rank_blocks_target() {
  return 16 * kCUs;
}

int pull_chunk() {
  return 128;  // list entries per chunk wave of a hub row
}

int pull_variant() {
  static const int v = env_int("KGE_PULL_VARIANT", 2);
  return v == 1 ? 1 : 2;
}

static int g_dedup_mode = -1;
int dedup_mode() {  // 1 (default): fold duplicate rows of batches >= 4096 rows; 0: never; 2: always.  KGE_DEDUP overrides the default.
  if (g_dedup_mode < 0) { const int v = env_int("KGE_DEDUP", 1); g_dedup_mode = v < 0 || v > 2 ? 1 : v; }
  return g_dedup_mode;
}

static int g_grad_mode = -1;
int grad_mode() {
  if (g_grad_mode < 0) g_grad_mode = env_int("KGE_GRAD_MODE", 0) ? 1 : 0;
  return g_grad_mode;
}

struct SideStream { hipStream_t stream; hipEvent_t fork, join; };
SideStream* side_stream() {
  // created once, by whichever thread comes first (a function-local static is initialised exactly once); nullptr when disabled or
  // when creation failed.  The stream and its fork / join events are shared by every caller: the library is thread-compatible,
  // not thread-safe (kge_hip.h) -- two host threads must not run training steps through it at the same time.
  static SideStream ss;
  static const bool ok = [] {
    return !env_int("KGE_NO_SIDE_STREAM", 0) && hipStreamCreateWithFlags(&ss.stream, hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&ss.join, hipEventDisableTiming) == hipSuccess;
  }();
  return ok ? &ss : nullptr;
}

int push_group_width(int width) {
  return width <= 16 ? 16 : (width <= 32 ? 32 : 64);
}

int score_dispatch(const kge_model_desc*, const float*, const float*, const int32_t*, int64_t, float*, uint32_t*, hipStream_t);
int64_t train_workspace_bytes(const kge_model_desc*, int64_t, int64_t);
int train_grads(const kge_model_desc*, const kge_loss_desc*, const float*, const float*, float*, float*, const int32_t*, int64_t,
                const int32_t*, int64_t, float*, float*, uint64_t, void*, int64_t, uint32_t*, hipStream_t);
bool plan_supported(const kge_model_desc*, const kge_loss_desc*);
int opt_state_init(const kge_opt_desc*, hipStream_t);
int64_t batch_plan_bytes(const kge_model_desc*, int64_t);
int batch_plan_build(const kge_model_desc*, const int32_t*, int64_t, const int32_t*, int64_t, const int32_t*, const int32_t*, int, int64_t, int64_t, void*,
                     int, uint32_t*, hipStream_t);
int train_grads_with_plan(const kge_model_desc*, const kge_loss_desc*, const float*, const float*, float*, float*, void*, int64_t, int64_t, float*,
                          uint64_t, void*, int64_t, hipStream_t, const kge_opt_desc* advance = nullptr, OptFinish* finish = nullptr, int grad_parts = 3);
bool grad_halves_supported(const kge_model_desc*);
int opt_apply(const kge_model_desc*, const kge_loss_desc*, const kge_opt_desc*, float*, float*, float*, float*, float, float*, void*,
              int64_t, hipStream_t, bool advanced = false, const OptFinish* finish = nullptr);

void set_pair_sort_mode(int mode);  // train.hip
void set_deterministic_mode(int on);  // pull2.hip
bool deterministic_mode();
int loss_only(const kge_model_desc*, const kge_loss_desc*, const float*, const float*, const int32_t*, int64_t, const int32_t*, int64_t, float*, double*,
              void*, int64_t, uint32_t*, hipStream_t);
int l3_value(const float*, int64_t, float, float*, void*, int64_t, hipStream_t);
int score_dispatch_training(const kge_model_desc*, const float*, const float*, const int32_t*, int64_t, uint64_t, int64_t, float*, uint32_t*, hipStream_t);
bool step_swap_supported(const kge_model_desc*, const kge_loss_desc*, const kge_opt_desc*);
int train_step_swap(const kge_model_desc*, const kge_loss_desc*, const kge_opt_desc*, const float*, const float*, float*, float*, void*, int64_t,
                    int64_t, float*, uint64_t, void*, int64_t, int, hipStream_t, const kge_row_window* win = nullptr, int64_t chain_seq = -1,
                    uint32_t* chain_err = nullptr);

int loss_value(const kge_loss_desc*, const float*, int64_t, const float*, int64_t, float*, float*, float*, void*, int64_t, hipStream_t);
int loss_pairs(const kge_loss_desc*, const float*, int64_t, const float*, int64_t, float*, int64_t, hipStream_t);
int64_t rank_workspace_bytes(const kge_model_desc*, int64_t);
int conve_sizes(const kge_model_desc*, int64_t*, int64_t*);
int64_t conve_workspace_bytes(const kge_model_desc*, int64_t, int64_t);
int64_t conve_infer_workspace_bytes(const kge_model_desc*, int64_t);
int conve_infer(const kge_model_desc*, const float*, const float*, const float*, float*, const int32_t*, int64_t, float*, int, void*, int64_t,
                uint32_t*, hipStream_t);
int conve_train_grads(const kge_model_desc*, const kge_loss_desc*, const float*, const float*, const float*, float*, float*, float*, float*,
                      const int32_t*, int64_t, const int32_t*, int64_t, float*, float*, uint64_t, void*, int64_t, uint32_t*, hipStream_t);
int opt_apply_vector(const kge_opt_desc*, float*, float*, int64_t, float, hipStream_t);
int64_t lit_workspace_bytes(const kge_model_desc*, int, int64_t, int64_t);
int hole_spec_half(int);
int hole_spec_basis(int, float*, float*, hipStream_t);
int hole_spec_forward(const float*, int64_t, int, const float*, float*, hipStream_t);
int hole_spec_adjoint(float*, int64_t, int, const float*, float*, int, hipStream_t);
int64_t lit_transform_workspace_bytes(const kge_model_desc*, int, int);
int lit_transform(const kge_model_desc*, int, const float*, const float*, int, int, const float*, int, float*, void*, int64_t, uint32_t*, hipStream_t);
struct LitConve { const float* theta; float* stats; float* gtheta; };  // literal.hip
int lit_train_grads(const kge_model_desc*, const kge_loss_desc*, int, const float*, const float*, const float*, int, int, const float*, float*, float*,
                    float*, const int32_t*, int64_t, const int32_t*, int64_t, float*, float*, uint64_t, void*, int64_t, uint32_t*, hipStream_t,
                    const LitConve* conve = nullptr, const kge_hake_params* hake = nullptr);
int64_t rank_sweep_q_workspace_bytes(int64_t, int);
int rank_sweep_q(const float*, int64_t, int, const float*, int, int, const int32_t*, const int32_t*, const int32_t*, int64_t, int32_t*, int,
                 void*, int64_t, hipStream_t);
int rank_counts(const kge_model_desc*, const float*, const float*, const float*, int, int, const int32_t*, int64_t, int,
                const int32_t*, const int32_t*, int64_t, int32_t*, int, float*, void*, int64_t, uint32_t*, hipStream_t);
int rank_chain_scores(const kge_model_desc*, const float*, const float*, const int32_t*, int64_t, int, float*, void*, int64_t,
                      uint32_t*, hipStream_t);

int64_t sampler_workspace_bytes(int64_t, int64_t, int64_t);
int build_batch_device(void*, void*, const int32_t*, const int32_t*, const int64_t*, const int64_t*, int64_t*, int64_t, int64_t, int64_t, int32_t*, int32_t*, int32_t*, int32_t*, const void*, void*, int64_t, uint32_t*,
                       hipStream_t);
int sampler_plan_host(int64_t, int64_t, int64_t, uint64_t*);
int64_t sampler_stream_bytes(int64_t, int64_t, int64_t);
int sampler_stream_round();
int sampler_stream_plan_host(int64_t, int64_t, int64_t, uint64_t*);
int sampler_stream_init(void*, int64_t, const void*, const void*, int64_t, int64_t, int64_t, hipStream_t);
int sampler_stream_produce(void*, int64_t, int64_t, int64_t, int64_t, const void*, hipStream_t);
int sampler_stream_draw(void*, int64_t, void*, void*, const int32_t*, const int32_t*, const int64_t*, const int64_t*, int64_t*, int64_t, int64_t, int64_t,
                        int32_t*, int32_t*, int32_t*, int32_t*, uint32_t*, hipStream_t);
int64_t sampler_plan_bytes();

// hake.hip
int64_t hake_workspace_bytes(int64_t, int64_t);
int hake_score(const kge_model_desc*, const kge_hake_params*, const float*, const float*, const int32_t*, int64_t, int64_t, const int64_t*, int64_t, float*, void*,
               int64_t, uint32_t*, hipStream_t);
int hake_rank_counts(const float*, int64_t, const int64_t*, int64_t, int32_t*, hipStream_t);
int hake_train_grads(const kge_model_desc*, const kge_loss_desc*, const kge_hake_params*, const float*, const float*, float*, float*, const int32_t*, int64_t,
                     const int32_t*, int64_t, float*, float*, uint64_t, void*, int64_t, uint32_t*, hipStream_t);
}  // namespace kge

using namespace kge;

extern "C" {

int kge_abi_version(void) { return KGE_ABI_VERSION; }

int kge_set_grad_mode(int mode) {
  if (mode != 0 && mode != 1) return fail(KGE_EINVAL, "kge_set_grad_mode: 0 = pull (gather-reduce), 1 = push (float atomics)");
  g_grad_mode = mode;
  return KGE_OK;
}

int kge_prof_enable(int on) {
  g_prof.on = on != 0;
  return KGE_OK;
}

int kge_prof_read(int label, double* total_ms, int64_t* launches) {
  if (label < 0 || label >= KGE_PROF_NLABELS || !total_ms || !launches) return fail(KGE_EINVAL, "kge_prof_read: bad arguments");
  double tot = 0.0;
  for (auto& pr : g_prof.ev[label]) {
    KGE_HIP_TRY(hipEventSynchronize(pr.second));
    float ms = 0.f;
    KGE_HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
    tot += ms;
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  *total_ms = tot;
  *launches = (int64_t)g_prof.ev[label].size();
  g_prof.ev[label].clear();
  return KGE_OK;
}

int kge_device_props(int64_t props[4]) {
  if (!props) return fail(KGE_EINVAL, "kge_device_props: null output");
  int dev = 0;
  KGE_HIP_TRY(hipGetDevice(&dev));
  hipDeviceProp_t p;
  KGE_HIP_TRY(hipGetDeviceProperties(&p, dev));
  props[0] = p.multiProcessorCount;
  props[1] = p.warpSize;
  props[2] = (int64_t)p.maxSharedMemoryPerMultiProcessor;
  props[3] = (int64_t)p.totalGlobalMem;
  return KGE_OK;
}

int kge_score(const kge_model_desc* md, const float* E, const float* Rel, const int32_t* triples, int64_t n, float* out,
              uint32_t* err_flag, kge_stream_t stream) {
  if (int rc = check_model(md, "kge_score")) return rc;
  if (n < 0) return fail(KGE_EINVAL, "kge_score: n = %lld", (long long)n);
  if (n == 0) return KGE_OK;
  if (!E || !Rel || !triples || !out || !err_flag) return fail(KGE_EINVAL, "kge_score: null pointer");
  return score_dispatch(md, E, Rel, triples, n, out, err_flag, static_cast<hipStream_t>(stream));
}

int kge_score_training(const kge_model_desc* md, const float* E, const float* Rel, const int32_t* triples, int64_t n, uint64_t seed,
                       int64_t row_offset, float* out, uint32_t* err_flag, kge_stream_t stream) {
  if (int rc = check_model(md, "kge_score_training")) return rc;
  if (n < 0 || row_offset < 0) return fail(KGE_EINVAL, "kge_score_training: n = %lld, row_offset = %lld", (long long)n, (long long)row_offset);
  if (n == 0) return KGE_OK;
  if (!E || !Rel || !triples || !out || !err_flag) return fail(KGE_EINVAL, "kge_score_training: null pointer");
  return score_dispatch_training(md, E, Rel, triples, n, seed, row_offset, out, err_flag, static_cast<hipStream_t>(stream));
}

int kge_loss_only(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, const int32_t* pos, int64_t m_pos,
                  const int32_t* neg, int64_t m_neg, float* loss_out, double* running_sum, void* workspace, int64_t workspace_bytes,
                  uint32_t* err_flag, kge_stream_t stream) {
  clear_stale_error();
  if (int rc = check_model(md, "kge_loss_only")) return rc;
  if (!ld || !E || !Rel || !pos || !neg || !loss_out || !workspace || !err_flag) return fail(KGE_EINVAL, "kge_loss_only: null pointer");
  if (m_pos <= 0 || m_neg <= 0) return fail(KGE_EINVAL, "kge_loss_only: empty batch side (reduce_mean of nothing)");
  return loss_only(md, ld, E, Rel, pos, m_pos, neg, m_neg, loss_out, running_sum, workspace, workspace_bytes, err_flag, static_cast<hipStream_t>(stream));
}

int kge_l3_value(const float* E, int64_t n, float weight, float* loss_inout, void* workspace, int64_t workspace_bytes, kge_stream_t stream) {
  if (!E || n < 0 || !loss_inout || !workspace) return fail(KGE_EINVAL, "kge_l3_value: bad arguments");
  return l3_value(E, n, weight, loss_inout, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int64_t kge_train_step_workspace_bytes(const kge_model_desc* md, const kge_loss_desc* ld, int64_t m_pos, int64_t m_neg) {
  (void)ld;
  if (check_model(md, "kge_train_step_workspace_bytes")) return KGE_EINVAL;
  if (m_pos < 0 || m_neg < 0) return fail(KGE_EINVAL, "negative batch size");
  return train_workspace_bytes(md, m_pos, m_neg);
}

static int check_train(const char* who, const kge_model_desc* md, const kge_loss_desc* ld, const void* E, const void* Rel,
                       const void* gE, const void* gR, const void* pos, int64_t m_pos, const void* neg, int64_t m_neg,
                       const void* loss_out, const void* err) {
  if (int rc = check_model(md, who)) return rc;
  if (!ld) return fail(KGE_EINVAL, "%s: loss descriptor is null", who);
  if (!E || !Rel || !gE || !gR || !pos || !neg || !loss_out || !err) return fail(KGE_EINVAL, "%s: null pointer", who);
  // reduce_mean over an empty tensor is NaN in TF; a Keras batch is never empty
  if (m_pos <= 0 || m_neg <= 0) return fail(KGE_EINVAL, "%s: empty batch (m_pos=%lld, m_neg=%lld)", who, (long long)m_pos, (long long)m_neg);
  if (m_pos + m_neg > 0x7FFFFFFFll) return fail(KGE_EUNSUPPORTED, "%s: batch above 2^31 rows", who);
  return KGE_OK;
}

int kge_train_grads(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, float* gradE,
                    float* gradRel, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, float* loss_out,
                    float* scores_out, uint64_t seed, void* workspace, int64_t workspace_bytes, uint32_t* err_flag,
                    kge_stream_t stream) {
  if (int rc = check_train("kge_train_grads", md, ld, E, Rel, gradE, gradRel, pos, m_pos, neg, m_neg, loss_out, err_flag)) return rc;
  return train_grads(md, ld, E, Rel, gradE, gradRel, pos, m_pos, neg, m_neg, loss_out, scores_out, seed, workspace,
                     workspace_bytes, err_flag, static_cast<hipStream_t>(stream));
}

int kge_opt_apply(const kge_model_desc* md, const kge_loss_desc* ld, const kge_opt_desc* od, float* E, float* Rel, float* gradE,
                  float* gradRel, float grad_scale, float* loss_out, void* workspace, int64_t workspace_bytes,
                  kge_stream_t stream) {
  if (int rc = check_model(md, "kge_opt_apply")) return rc;
  if (!ld || !od || !E || !Rel || !gradE || !gradRel) return fail(KGE_EINVAL, "kge_opt_apply: null pointer");
  return opt_apply(md, ld, od, E, Rel, gradE, gradRel, grad_scale, loss_out, workspace, workspace_bytes,
                   static_cast<hipStream_t>(stream));
}

int kge_opt_state_init(const kge_opt_desc* od, kge_stream_t stream) {
  clear_stale_error();
  if (!od || !od->step_dev) return fail(KGE_EINVAL, "kge_opt_state_init: optimizer descriptor without step_dev");
  if (od->kind != KGE_OPT_ADAM_TF && od->kind != KGE_OPT_SGD) return fail(KGE_EINVAL, "unknown optimizer %d", od->kind);
  if (od->step < 1) return fail(KGE_EINVAL, "kge_opt_state_init: step must be >= 1 (the number of the next update)");
  return opt_state_init(od, static_cast<hipStream_t>(stream));
}

int kge_train_step(const kge_model_desc* md, const kge_loss_desc* ld, const kge_opt_desc* od, float* E, float* Rel, float* gradE,
                   float* gradRel, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, float* loss_out,
                   float* scores_out, uint64_t seed, void* workspace, int64_t workspace_bytes, uint32_t* err_flag,
                   kge_stream_t stream) {
  if (int rc = check_train("kge_train_step", md, ld, E, Rel, gradE, gradRel, pos, m_pos, neg, m_neg, loss_out, err_flag)) return rc;
  if (!od) return fail(KGE_EINVAL, "kge_train_step: optimizer descriptor is null");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (int rc = train_grads(md, ld, E, Rel, gradE, gradRel, pos, m_pos, neg, m_neg, loss_out, scores_out, seed, workspace,
                           workspace_bytes, err_flag, st)) return rc;
  return opt_apply(md, ld, od, E, Rel, gradE, gradRel, 1.f, loss_out, workspace, workspace_bytes, st);
}

static int check_plan(const char* who, const kge_model_desc* md, const kge_loss_desc* ld, int64_t m_pos, int64_t m_neg) {
  if (int rc = check_model(md, who)) return rc;
  if (!ld) return fail(KGE_EINVAL, "%s: loss descriptor is null", who);
  if (m_pos <= 0 || m_neg <= 0) return fail(KGE_EINVAL, "%s: empty batch (m_pos=%lld, m_neg=%lld)", who, (long long)m_pos, (long long)m_neg);
  if (m_pos + m_neg > 0x7FFFFFFFll) return fail(KGE_EUNSUPPORTED, "%s: batch above 2^31 rows", who);
  if (!plan_supported(md, ld))
    return fail(KGE_EUNSUPPORTED, "%s: batch plans exist for the relation-grouped pull path (DistMult / TransE / ComplEx, pointwize losses, "
                "gradient mode 0); use kge_train_step", who);
  return KGE_OK;
}

int64_t kge_batch_plan_bytes(const kge_model_desc* md, const kge_loss_desc* ld, int64_t m_pos, int64_t m_neg) {
  if (int rc = check_plan("kge_batch_plan_bytes", md, ld, m_pos, m_neg)) return rc;
  return batch_plan_bytes(md, m_pos + m_neg);
}

int kge_batch_plan_build(const kge_model_desc* md, const kge_loss_desc* ld, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg,
                         void* plan, int64_t plan_bytes, int32_t parts, uint32_t* err_flag, kge_stream_t stream) {
  if (int rc = check_plan("kge_batch_plan_build", md, ld, m_pos, m_neg)) return rc;
  if (!pos || !neg || !plan || !err_flag) return fail(KGE_EINVAL, "kge_batch_plan_build: null pointer");
  if (plan_bytes < batch_plan_bytes(md, m_pos + m_neg))
    return fail(KGE_EWORKSPACE, "kge_batch_plan_build: plan buffer has %lld bytes, needs %lld", (long long)plan_bytes, (long long)batch_plan_bytes(md, m_pos + m_neg));
  if (parts < 1 || parts > 15 || !(parts & 7)) return fail(KGE_EINVAL, "kge_batch_plan_build: parts mask %d outside 1..15", parts);
  return batch_plan_build(md, pos, m_pos, neg, m_neg, nullptr, nullptr, 1, 0, 0, plan, parts, err_flag, static_cast<hipStream_t>(stream));
}

int kge_batch_plan_build_counted(const kge_model_desc* md, const kge_loss_desc* ld, const int32_t* cand_rows, const int32_t* cand_counts,
                                 int32_t pos_lanes, int64_t n_pos_cand, int64_t n_neg_cand, int64_t m_pos, int64_t m_neg, void* plan,
                                 int64_t plan_bytes, int32_t parts, uint32_t* err_flag, kge_stream_t stream) {
  if (pos_lanes < 1 || pos_lanes > 64) return fail(KGE_EINVAL, "kge_batch_plan_build_counted: pos_lanes %d outside 1..64", pos_lanes);
  if (int rc = check_plan("kge_batch_plan_build_counted", md, ld, m_pos, m_neg)) return rc;
  if (!cand_rows || !cand_counts || !plan || !err_flag) return fail(KGE_EINVAL, "kge_batch_plan_build_counted: null pointer");
  if (n_pos_cand <= 0 || n_neg_cand <= 0 || n_pos_cand + n_neg_cand > m_pos + m_neg)
    return fail(KGE_EINVAL, "kge_batch_plan_build_counted: %lld + %lld candidate rows for %lld + %lld batch rows", (long long)n_pos_cand,
                (long long)n_neg_cand, (long long)m_pos, (long long)m_neg);
  if (md->dropout > 0.f) return fail(KGE_EUNSUPPORTED, "kge_batch_plan_build_counted: folded batches and dropout exclude each other (every row has its own mask)");
  if (plan_bytes < batch_plan_bytes(md, m_pos + m_neg))
    return fail(KGE_EWORKSPACE, "kge_batch_plan_build_counted: plan buffer has %lld bytes, needs %lld", (long long)plan_bytes,
                (long long)batch_plan_bytes(md, m_pos + m_neg));
  if (parts < 1 || parts > 15 || !(parts & 7)) return fail(KGE_EINVAL, "kge_batch_plan_build_counted: parts mask %d outside 1..15", parts);
  return batch_plan_build(md, nullptr, m_pos, nullptr, m_neg, cand_rows, cand_counts, pos_lanes, n_pos_cand, n_neg_cand, plan, parts, err_flag,
                          static_cast<hipStream_t>(stream));
}

int kge_train_grads_planned(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, float* gradE, float* gradRel,
                            void* plan, int64_t plan_bytes, int64_t m_pos, int64_t m_neg, float* loss_out, uint64_t seed, void* workspace,
                            int64_t workspace_bytes, kge_stream_t stream) {
  if (int rc = check_plan("kge_train_grads_planned", md, ld, m_pos, m_neg)) return rc;
  if (!E || !Rel || !gradE || !gradRel || !plan || !loss_out) return fail(KGE_EINVAL, "kge_train_grads_planned: null pointer");
  if (plan_bytes < batch_plan_bytes(md, m_pos + m_neg))
    return fail(KGE_EWORKSPACE, "kge_train_grads_planned: plan buffer has %lld bytes, needs %lld", (long long)plan_bytes, (long long)batch_plan_bytes(md, m_pos + m_neg));
  return train_grads_with_plan(md, ld, E, Rel, gradE, gradRel, plan, m_pos, m_neg, loss_out, seed, workspace, workspace_bytes,
                               static_cast<hipStream_t>(stream));
}

int kge_train_grads_halves_supported(const kge_model_desc* md, const kge_loss_desc* ld) {
  if (int rc = check_model(md, "kge_train_grads_halves_supported")) return rc;
  if (!ld) return fail(KGE_EINVAL, "kge_train_grads_halves_supported: loss descriptor is null");
  return plan_supported(md, ld) && grad_halves_supported(md) ? 1 : 0;
}

int kge_train_grads_planned_parts(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, float* gradE, float* gradRel,
                                  void* plan, int64_t plan_bytes, int64_t m_pos, int64_t m_neg, float* loss_out, uint64_t seed, void* workspace,
                                  int64_t workspace_bytes, int32_t parts, kge_stream_t stream) {
  if (parts < 1 || parts > 3) return fail(KGE_EINVAL, "kge_train_grads_planned_parts: parts mask %d outside 1..3", parts);
  if (int rc = check_plan("kge_train_grads_planned_parts", md, ld, m_pos, m_neg)) return rc;
  if (!E || !Rel || !gradE || !gradRel || !plan || !loss_out) return fail(KGE_EINVAL, "kge_train_grads_planned_parts: null pointer");
  if (plan_bytes < batch_plan_bytes(md, m_pos + m_neg))
    return fail(KGE_EWORKSPACE, "kge_train_grads_planned_parts: plan buffer has %lld bytes, needs %lld", (long long)plan_bytes, (long long)batch_plan_bytes(md, m_pos + m_neg));
  if (parts != 3 && !grad_halves_supported(md))
    return fail(KGE_EUNSUPPORTED, "kge_train_grads_planned_parts: the halves come apart only for rows of at most 256 floats (per half row for ComplEx)");
  return train_grads_with_plan(md, ld, E, Rel, gradE, gradRel, plan, m_pos, m_neg, loss_out, seed, workspace, workspace_bytes,
                               static_cast<hipStream_t>(stream), nullptr, nullptr, parts);
}

int kge_train_step_planned(const kge_model_desc* md, const kge_loss_desc* ld, const kge_opt_desc* od, float* E, float* Rel, float* gradE,
                           float* gradRel, void* plan, int64_t plan_bytes, int64_t m_pos, int64_t m_neg, float* loss_out, uint64_t seed,
                           void* workspace, int64_t workspace_bytes, kge_stream_t stream) {
  if (!od) return fail(KGE_EINVAL, "kge_train_step_planned: optimizer descriptor is null");
  if (int rc = check_plan("kge_train_step_planned", md, ld, m_pos, m_neg)) return rc;
  if (!E || !Rel || !gradE || !gradRel || !plan || !loss_out) return fail(KGE_EINVAL, "kge_train_step_planned: null pointer");
  if (plan_bytes < batch_plan_bytes(md, m_pos + m_neg))
    return fail(KGE_EWORKSPACE, "kge_train_step_planned: plan buffer has %lld bytes, needs %lld", (long long)plan_bytes, (long long)batch_plan_bytes(md, m_pos + m_neg));
  // device-resident step state and no L3 term: the gradient half's closing kernel advances the state, the step ends with the sweep
  const bool fused = od->step_dev && ld->regularization == 0.f;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // and the optimizer sweep closes the gradient half itself (OptFinish): no finish kernel between the pull and the sweep
  OptFinish fin;
  if (int rc = train_grads_with_plan(md, ld, E, Rel, gradE, gradRel, plan, m_pos, m_neg, loss_out, seed, workspace, workspace_bytes, st, fused ? od : nullptr, &fin))
    return rc;
  return opt_apply(md, ld, od, E, Rel, gradE, gradRel, 1.f, loss_out, workspace, workspace_bytes, st, fused, &fin);
}

int kge_train_step_swap_supported(const kge_model_desc* md, const kge_loss_desc* ld, const kge_opt_desc* od) {
  if (!od) return fail(KGE_EINVAL, "kge_train_step_swap_supported: optimizer descriptor is null");
  if (int rc = check_model(md, "kge_train_step_swap_supported")) return rc;
  if (!ld) return fail(KGE_EINVAL, "kge_train_step_swap_supported: loss descriptor is null");
  return step_swap_supported(md, ld, od) ? 1 : 0;
}

int kge_train_step_swap(const kge_model_desc* md, const kge_loss_desc* ld, const kge_opt_desc* od, const float* E_in, const float* Rel_in,
                        float* E_out, float* Rel_out, void* plan, int64_t plan_bytes, int64_t m_pos, int64_t m_neg, float* loss_out,
                        uint64_t seed, void* workspace, int64_t workspace_bytes, int32_t parts, kge_stream_t stream) {
  if (!od) return fail(KGE_EINVAL, "kge_train_step_swap: optimizer descriptor is null");
  if (parts < 1 || parts > 3) return fail(KGE_EINVAL, "kge_train_step_swap: parts mask %d outside 1..3", parts);
  if (int rc = check_plan("kge_train_step_swap", md, ld, m_pos, m_neg)) return rc;
  if (!E_in || !Rel_in || !E_out || !Rel_out || !plan || !loss_out) return fail(KGE_EINVAL, "kge_train_step_swap: null pointer");
  if (E_in == E_out || Rel_in == Rel_out) return fail(KGE_EINVAL, "kge_train_step_swap: input and output tables must be distinct buffers");
  if (!step_swap_supported(md, ld, od))
    return fail(KGE_EUNSUPPORTED, "kge_train_step_swap: needs the device-resident step state, no L3 term, rows of at most 256 floats (per half row for ComplEx)");
  if (plan_bytes < batch_plan_bytes(md, m_pos + m_neg))
    return fail(KGE_EWORKSPACE, "kge_train_step_swap: plan buffer has %lld bytes, needs %lld", (long long)plan_bytes, (long long)batch_plan_bytes(md, m_pos + m_neg));
  return train_step_swap(md, ld, od, E_in, Rel_in, E_out, Rel_out, plan, m_pos, m_neg, loss_out, seed, workspace, workspace_bytes, parts,
                         static_cast<hipStream_t>(stream));
}

int kge_train_step_swap_window(const kge_model_desc* md, const kge_loss_desc* ld, const kge_opt_desc* od, const float* E_in, const float* Rel_in,
                               float* E_out, float* Rel_out, void* plan, int64_t plan_bytes, int64_t m_pos, int64_t m_neg, float* loss_out,
                               uint64_t seed, void* workspace, int64_t workspace_bytes, const kge_row_window* win, kge_stream_t stream) {
  if (!od || !win) return fail(KGE_EINVAL, "kge_train_step_swap_window: null descriptor");
  if (int rc = check_plan("kge_train_step_swap_window", md, ld, m_pos, m_neg)) return rc;
  if (!E_in || !Rel_in || !E_out || !Rel_out || !plan || !loss_out) return fail(KGE_EINVAL, "kge_train_step_swap_window: null pointer");
  if (E_in == E_out || Rel_in == Rel_out) return fail(KGE_EINVAL, "kge_train_step_swap_window: input and output tables must be distinct buffers");
  if (win->ent_lo < 0 || win->ent_hi < win->ent_lo || win->ent_hi > md->num_entities || win->rel_lo < 0 || win->rel_hi < win->rel_lo ||
      win->rel_hi > md->num_relations)
    return fail(KGE_EINVAL, "kge_train_step_swap_window: window [%d, %d) x [%d, %d) outside the tables (%d, %d)", win->ent_lo, win->ent_hi, win->rel_lo,
                win->rel_hi, md->num_entities, md->num_relations);
  if (md->dropout > 0.f || !step_swap_supported(md, ld, od))
    return fail(KGE_EUNSUPPORTED, "kge_train_step_swap_window: needs what kge_train_step_swap needs, and no dropout");
  if (plan_bytes < batch_plan_bytes(md, m_pos + m_neg))
    return fail(KGE_EWORKSPACE, "kge_train_step_swap_window: plan buffer has %lld bytes, needs %lld", (long long)plan_bytes, (long long)batch_plan_bytes(md, m_pos + m_neg));
  return train_step_swap(md, ld, od, E_in, Rel_in, E_out, Rel_out, plan, m_pos, m_neg, loss_out, seed, workspace, workspace_bytes, 3,
                         static_cast<hipStream_t>(stream), win);
}

int kge_train_step_swap_chain(const kge_model_desc* md, const kge_loss_desc* ld, const kge_opt_desc* od, const float* E_in, const float* Rel_in,
                              float* E_out, float* Rel_out, void* plan, int64_t plan_bytes, int64_t m_pos, int64_t m_neg, float* loss_out,
                              uint64_t seed, void* workspace, int64_t workspace_bytes, int64_t seq, uint32_t* err_flag, kge_stream_t stream) {
  if (!od) return fail(KGE_EINVAL, "kge_train_step_swap_chain: optimizer descriptor is null");
  if (seq < 0) return fail(KGE_EINVAL, "kge_train_step_swap_chain: seq must be >= 0");
  if (int rc = check_plan("kge_train_step_swap_chain", md, ld, m_pos, m_neg)) return rc;
  if (!E_in || !Rel_in || !E_out || !Rel_out || !plan || !loss_out) return fail(KGE_EINVAL, "kge_train_step_swap_chain: null pointer");
  if (E_in == E_out || Rel_in == Rel_out) return fail(KGE_EINVAL, "kge_train_step_swap_chain: input and output tables must be distinct buffers");
  if (!step_swap_supported(md, ld, od))
    return fail(KGE_EUNSUPPORTED, "kge_train_step_swap_chain: needs what kge_train_step_swap needs");
  if (plan_bytes < batch_plan_bytes(md, m_pos + m_neg))
    return fail(KGE_EWORKSPACE, "kge_train_step_swap_chain: plan buffer has %lld bytes, needs %lld", (long long)plan_bytes, (long long)batch_plan_bytes(md, m_pos + m_neg));
  return train_step_swap(md, ld, od, E_in, Rel_in, E_out, Rel_out, plan, m_pos, m_neg, loss_out, seed, workspace, workspace_bytes, 3,
                         static_cast<hipStream_t>(stream), nullptr, seq, err_flag);
}

int64_t kge_rank_workspace_bytes(const kge_model_desc* md, int64_t n_queries) {
  if (check_model(md, "kge_rank_workspace_bytes")) return KGE_EINVAL;
  if (n_queries < 0) return fail(KGE_EINVAL, "negative query count");
  return rank_workspace_bytes(md, n_queries);
}

int kge_rank_counts(const kge_model_desc* md, const float* E, const float* Rel, const float* E_cand, int32_t cand_offset,
                    int32_t cand_rows, const int32_t* triples, int64_t n_queries, int32_t side, const int32_t* filt_query,
                    const int32_t* filt_ids, int64_t n_filt, int32_t* counts, int32_t accumulate, float* target_scores_out,
                    void* workspace, int64_t workspace_bytes, uint32_t* err_flag, kge_stream_t stream) {
  if (int rc = check_model(md, "kge_rank_counts")) return rc;
  if (n_queries < 0 || n_filt < 0) return fail(KGE_EINVAL, "kge_rank_counts: negative size");
  if (n_queries == 0) return KGE_OK;
  if (side < 0 || side > 2) return fail(KGE_EINVAL, "kge_rank_counts: side must be 0 (tail), 1 (head) or 2 (both: tail queries, then head queries)");
  if (md->model == KGE_HAKE) return fail(KGE_EUNSUPPORTED, "kge_rank_counts: a HAKE score depends on the rows it is predicted with (chunks of 32 in the reference's validate): rank through kge_hake_score");
  if (!E || !Rel || (!E_cand && cand_rows > 0) || !triples || !counts || !workspace || !err_flag || (n_filt && (!filt_query || !filt_ids)))
    return fail(KGE_EINVAL, "kge_rank_counts: null pointer");
  if (cand_offset < 0 || cand_rows < 0 || (int64_t)cand_offset + cand_rows > 0x7FFFFFFFll)
    return fail(KGE_EINVAL, "kge_rank_counts: bad candidate range [%d,%d+%d)", cand_offset, cand_offset, cand_rows);
  if (cand_rows == 0) {  // an empty row shard counts nothing
    if (!accumulate) KGE_HIP_TRY(hipMemsetAsync(counts, 0, sizeof(int32_t) * n_queries * (side == 2 ? 2 : 1), static_cast<hipStream_t>(stream)));
    return KGE_OK;
  }
  return rank_counts(md, E, Rel, E_cand, cand_offset, cand_rows, triples, n_queries, side, filt_query, filt_ids, n_filt, counts,
                     accumulate, target_scores_out, workspace, workspace_bytes, err_flag, static_cast<hipStream_t>(stream));
}

int kge_rank_chain_scores(const kge_model_desc* md, const float* E, const float* Rel, const int32_t* triples, int64_t n_queries,
                          int32_t side, float* out, void* workspace, int64_t workspace_bytes, uint32_t* err_flag,
                          kge_stream_t stream) {
  if (int rc = check_model(md, "kge_rank_chain_scores")) return rc;
  if (n_queries <= 0 || !E || !Rel || !triples || !out || !workspace || !err_flag) return fail(KGE_EINVAL, "kge_rank_chain_scores: bad arguments");
  return rank_chain_scores(md, E, Rel, triples, n_queries, side, out, workspace, workspace_bytes, err_flag,
                           static_cast<hipStream_t>(stream));
}

int64_t kge_sampler_workspace_bytes(int64_t B, int64_t ns, int64_t num_e) {
  if (B <= 0 || ns <= 0 || num_e <= 0 || num_e > 0x7FFFFFFFll || B * ns > 0x3FFFFFFFll) return fail(KGE_EINVAL, "kge_sampler_workspace_bytes: bad sizes");
  return sampler_workspace_bytes(B, ns, num_e);
}

int64_t kge_sampler_plan_bytes(void) { return sampler_plan_bytes(); }

int kge_sampler_plan_host(int64_t B, int64_t ns, int64_t num_e, void* plan_host) {
  if (!plan_host || B <= 0 || ns <= 0 || num_e <= 0 || num_e > 0x7FFFFFFFll || B * ns > 0x3FFFFFFFll) return fail(KGE_EINVAL, "kge_sampler_plan_host: bad arguments");
  return sampler_plan_host(B, ns, num_e, static_cast<uint64_t*>(plan_host));
}

int kge_build_batch_device(void* np_state_dev, void* py_state_dev, const int32_t* batch_dev, int64_t B, int64_t ns, int64_t num_e,
                           int32_t* pos_out_dev, int32_t* neg_out_dev, const void* plan_dev, void* workspace, int64_t workspace_bytes,
                           uint32_t* err_flag, kge_stream_t stream) {
  clear_stale_error();
  if (!np_state_dev || !py_state_dev || !batch_dev || !pos_out_dev || !neg_out_dev || !workspace || !err_flag)
    return fail(KGE_EINVAL, "kge_build_batch_device: null pointer");
  if (B <= 0 || ns <= 0 || num_e <= 0 || num_e > 0x7FFFFFFFll || B * ns > 0x3FFFFFFFll) return fail(KGE_EINVAL, "kge_build_batch_device: bad sizes");
  return build_batch_device(np_state_dev, py_state_dev, batch_dev, nullptr, nullptr, nullptr, nullptr, B, ns, num_e, pos_out_dev, neg_out_dev,
                            nullptr, nullptr, plan_dev, workspace, workspace_bytes, err_flag, static_cast<hipStream_t>(stream));
}

int kge_build_batch_device_indexed(void* np_state_dev, void* py_state_dev, const int32_t* kg_dev, const int64_t* index_dev,
                                   const int64_t* starts_dev, int64_t* cursor_dev, int64_t B, int64_t ns, int64_t num_e, int32_t* pos_out_dev,
                                   int32_t* neg_out_dev, int32_t* cand_rows_dev, int32_t* cand_counts_dev, const void* plan_dev, void* workspace,
                                   int64_t workspace_bytes, uint32_t* err_flag, kge_stream_t stream) {
  clear_stale_error();
  if ((cand_rows_dev == nullptr) != (cand_counts_dev == nullptr))
    return fail(KGE_EINVAL, "kge_build_batch_device_indexed: cand_rows and cand_counts go together");
  if (!np_state_dev || !py_state_dev || !kg_dev || !index_dev || !starts_dev || !cursor_dev || !workspace || !err_flag)
    return fail(KGE_EINVAL, "kge_build_batch_device_indexed: null pointer");
  if ((pos_out_dev == nullptr) != (neg_out_dev == nullptr) || (!pos_out_dev && !cand_rows_dev))
    return fail(KGE_EINVAL, "kge_build_batch_device_indexed: give pos_out and neg_out, or cand_rows and cand_counts, or both");
  if (B <= 0 || ns <= 0 || num_e <= 0 || num_e > 0x7FFFFFFFll || B * ns > 0x3FFFFFFFll)
    return fail(KGE_EINVAL, "kge_build_batch_device_indexed: bad sizes");
  return build_batch_device(np_state_dev, py_state_dev, nullptr, kg_dev, index_dev, starts_dev, cursor_dev, B, ns, num_e, pos_out_dev, neg_out_dev,
                            cand_rows_dev, cand_counts_dev, plan_dev, workspace, workspace_bytes, err_flag, static_cast<hipStream_t>(stream));
}

static int check_sampler_shape(const char* who, int64_t B, int64_t ns, int64_t num_e) {
  if (B <= 0 || ns <= 0 || num_e <= 0 || num_e > 0x7FFFFFFFll || B * ns > 0x3FFFFFFFll) return fail(KGE_EINVAL, "%s: bad sizes", who);
  return KGE_OK;
}

int64_t kge_sampler_stream_bytes(int64_t B, int64_t ns, int64_t num_e) {
  if (int rc = check_sampler_shape("kge_sampler_stream_bytes", B, ns, num_e)) return rc;
  return sampler_stream_bytes(B, ns, num_e);
}

int32_t kge_sampler_stream_round(void) { return sampler_stream_round(); }

int kge_sampler_stream_plan_host(int64_t B, int64_t ns, int64_t num_e, void* plan_host) {
  if (int rc = check_sampler_shape("kge_sampler_stream_plan_host", B, ns, num_e)) return rc;
  if (!plan_host) return fail(KGE_EINVAL, "kge_sampler_stream_plan_host: null output");
  return sampler_stream_plan_host(B, ns, num_e, static_cast<uint64_t*>(plan_host));
}

int kge_sampler_stream_init(void* stream_buf, int64_t stream_bytes, const void* np_state_dev, const void* py_state_dev, int64_t B, int64_t ns,
                            int64_t num_e, kge_stream_t stream) {
  clear_stale_error();
  if (int rc = check_sampler_shape("kge_sampler_stream_init", B, ns, num_e)) return rc;
  if (!stream_buf || !np_state_dev || !py_state_dev) return fail(KGE_EINVAL, "kge_sampler_stream_init: null pointer");
  return sampler_stream_init(stream_buf, stream_bytes, np_state_dev, py_state_dev, B, ns, num_e, static_cast<hipStream_t>(stream));
}

int kge_sampler_stream_produce(void* stream_buf, int64_t stream_bytes, int64_t B, int64_t ns, int64_t num_e, const void* plan_dev,
                               kge_stream_t stream) {
  clear_stale_error();
  if (int rc = check_sampler_shape("kge_sampler_stream_produce", B, ns, num_e)) return rc;
  if (!stream_buf) return fail(KGE_EINVAL, "kge_sampler_stream_produce: null pointer");
  return sampler_stream_produce(stream_buf, stream_bytes, B, ns, num_e, plan_dev, static_cast<hipStream_t>(stream));
}

int kge_sampler_stream_draw(void* stream_buf, int64_t stream_bytes, void* np_state_dev, void* py_state_dev, const int32_t* batch_dev,
                            const int32_t* kg_dev, const int64_t* index_dev, const int64_t* starts_dev, int64_t* cursor_dev, int64_t B,
                            int64_t ns, int64_t num_e, int32_t* pos_out_dev, int32_t* neg_out_dev, int32_t* cand_rows_dev,
                            int32_t* cand_counts_dev, uint32_t* err_flag, kge_stream_t stream) {
  clear_stale_error();
  if (int rc = check_sampler_shape("kge_sampler_stream_draw", B, ns, num_e)) return rc;
  if (!stream_buf || !np_state_dev || !py_state_dev || !err_flag) return fail(KGE_EINVAL, "kge_sampler_stream_draw: null pointer");
  if (!batch_dev && (!kg_dev || !index_dev || !starts_dev || !cursor_dev))
    return fail(KGE_EINVAL, "kge_sampler_stream_draw: give the batch rows, or kg / index / starts / cursor");
  if ((cand_rows_dev == nullptr) != (cand_counts_dev == nullptr)) return fail(KGE_EINVAL, "kge_sampler_stream_draw: cand_rows and cand_counts go together");
  if ((pos_out_dev == nullptr) != (neg_out_dev == nullptr) || (!pos_out_dev && !cand_rows_dev))
    return fail(KGE_EINVAL, "kge_sampler_stream_draw: give pos_out and neg_out, or cand_rows and cand_counts, or both");
  return sampler_stream_draw(stream_buf, stream_bytes, np_state_dev, py_state_dev, batch_dev, kg_dev, index_dev, starts_dev, cursor_dev, B, ns, num_e,
                             pos_out_dev, neg_out_dev, cand_rows_dev, cand_counts_dev, err_flag, static_cast<hipStream_t>(stream));
}

int64_t kge_loss_workspace_bytes(int64_t m_true, int64_t m_false) {
  if (m_true < 0 || m_false < 0) return fail(KGE_EINVAL, "negative size");
  return train_workspace_bytes(nullptr, m_true, m_false);
}

int kge_loss_value(const kge_loss_desc* ld, const float* true_scores, int64_t m_true, const float* false_scores, int64_t m_false,
                   float* loss_out, float* d_true, float* d_false, void* workspace, int64_t workspace_bytes, kge_stream_t stream) {
  clear_stale_error();
  if (!ld || !true_scores || !false_scores || !loss_out || !workspace) return fail(KGE_EINVAL, "kge_loss_value: null pointer");
  if (m_true <= 0 || m_false <= 0) return fail(KGE_EINVAL, "kge_loss_value: empty score vector (reduce_mean of nothing)");
  return loss_value(ld, true_scores, m_true, false_scores, m_false, loss_out, d_true, d_false, workspace, workspace_bytes,
                    static_cast<hipStream_t>(stream));
}

int kge_set_pair_sort_mode(int mode) {
  if (mode < -1 || mode > 1) return fail(KGE_EINVAL, "kge_set_pair_sort_mode: mode %d (-1 = by pair count, 0 = all-pairs kernels, 1 = sorted scores / four sums)", mode);
  set_pair_sort_mode(mode);
  return KGE_OK;
}

int kge_set_deterministic(int on) {
  set_deterministic_mode(on);
  return KGE_OK;
}
int kge_get_deterministic(void) { return deterministic_mode() ? 1 : 0; }

int kge_set_dedup_mode(int mode) {
  if (mode < 0 || mode > 2) return fail(KGE_EINVAL, "kge_set_dedup_mode: mode %d (0 = off, 1 = batches of 4096+ rows, 2 = always)", mode);
  g_dedup_mode = mode;
  return KGE_OK;
}

int kge_loss_pairs(const kge_loss_desc* ld, const float* true_scores, int64_t m_true, const float* false_scores, int64_t m_false,
                   float* out, int64_t out_elems, kge_stream_t stream) {
  clear_stale_error();
  if (!ld || !true_scores || !false_scores || !out) return fail(KGE_EINVAL, "kge_loss_pairs: null pointer");
  if (m_true <= 0 || m_false <= 0) return fail(KGE_EINVAL, "kge_loss_pairs: empty score vector");
  return loss_pairs(ld, true_scores, m_true, false_scores, m_false, out, out_elems, static_cast<hipStream_t>(stream));
}

static int check_conve(const kge_model_desc* md, const char* who) {
  if (int rc = check_model(md, who)) return rc;
  if (md->model != KGE_CONVE) return fail(KGE_EINVAL, "%s: model %d is not ConvE", who, md->model);
  if (!(md->hidden_dropout >= 0.f && md->hidden_dropout < 1.f)) return fail(KGE_EINVAL, "%s: hidden dropout rate %f outside [0,1)", who, md->hidden_dropout);
  return KGE_OK;
}

// ---- HAKE (hake.hip)
static int check_hake(const kge_model_desc* md, const kge_hake_params* hp, const char* who) {
  if (int rc = check_model(md, who)) return rc;
  if (md->model != KGE_HAKE) return fail(KGE_EINVAL, "%s: model %d is not HAKE", who, md->model);
  if (!hp || !(hp->embedding_range > 0.f)) return fail(KGE_EINVAL, "%s: kge_hake_params with embedding_range > 0 needed", who);
  return KGE_OK;
}

int64_t kge_hake_workspace_bytes(int64_t n, int64_t chunk) {
  if (n < 0) return fail(KGE_EINVAL, "kge_hake_workspace_bytes: negative size");
  return hake_workspace_bytes(n, chunk > n ? n : chunk);  // (0: the two sides of a training batch; < 0: -chunk segments by offsets)
}

int kge_hake_score(const kge_model_desc* md, const kge_hake_params* hp, const float* E, const float* Rel, const int32_t* triples, int64_t n,
                   int64_t chunk, const int64_t* seg_off, int64_t n_seg, float* out, void* workspace, int64_t workspace_bytes, uint32_t* err_flag,
                   kge_stream_t stream) {
  if (int rc = check_hake(md, hp, "kge_hake_score")) return rc;
  if (n == 0) return KGE_OK;
  if (n < 0 || n > 0x7FFFFFFFll || !E || !Rel || !triples || !out || !workspace || !err_flag) return fail(KGE_EINVAL, "kge_hake_score: bad arguments");
  if (seg_off && (n_seg <= 0 || n_seg > 0x7FFFFFFFll)) return fail(KGE_EINVAL, "kge_hake_score: %lld segments", (long long)n_seg);
  return hake_score(md, hp, E, Rel, triples, n, chunk, seg_off, n_seg, out, workspace, workspace_bytes, err_flag, static_cast<hipStream_t>(stream));
}

int kge_hake_rank_counts(const float* scores, int64_t n, const int64_t* q_off, int64_t n_q, int32_t* counts, kge_stream_t stream) {
  if (n_q == 0) return KGE_OK;
  if (n < 0 || n_q < 0 || n_q > 0x7FFFFFFFll || !scores || !q_off || !counts) return fail(KGE_EINVAL, "kge_hake_rank_counts: bad arguments");
  return hake_rank_counts(scores, n, q_off, n_q, counts, static_cast<hipStream_t>(stream));
}

int kge_hake_train_grads(const kge_model_desc* md, const kge_loss_desc* ld, const kge_hake_params* hp, const float* E, const float* Rel,
                         float* gradE, float* gradRel, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, float* loss_out,
                         float* scores_out, uint64_t seed, void* workspace, int64_t workspace_bytes, uint32_t* err_flag, kge_stream_t stream) {
  if (int rc = check_hake(md, hp, "kge_hake_train_grads")) return rc;
  if (int rc = check_train("kge_hake_train_grads", md, ld, E, Rel, gradE, gradRel, pos, m_pos, neg, m_neg, loss_out, err_flag)) return rc;
  if (!workspace) return fail(KGE_EINVAL, "kge_hake_train_grads: null workspace");
  return hake_train_grads(md, ld, hp, E, Rel, gradE, gradRel, pos, m_pos, neg, m_neg, loss_out, scores_out, seed, workspace, workspace_bytes, err_flag,
                          static_cast<hipStream_t>(stream));
}

int kge_conve_sizes(const kge_model_desc* md, int64_t* n_theta, int64_t* n_stats) {
  if (int rc = check_conve(md, "kge_conve_sizes")) return rc;
  if (!n_theta || !n_stats) return fail(KGE_EINVAL, "kge_conve_sizes: null output");
  return conve_sizes(md, n_theta, n_stats);
}

int64_t kge_conve_workspace_bytes(const kge_model_desc* md, int64_t m_pos, int64_t m_neg) {
  if (check_conve(md, "kge_conve_workspace_bytes")) return KGE_EINVAL;
  if (m_pos < 0 || m_neg < 0) return fail(KGE_EINVAL, "negative batch size");
  return conve_workspace_bytes(md, m_pos, m_neg);
}

int64_t kge_conve_infer_workspace_bytes(const kge_model_desc* md, int64_t n) {
  if (check_conve(md, "kge_conve_infer_workspace_bytes")) return KGE_EINVAL;
  if (n < 0) return fail(KGE_EINVAL, "negative size");
  return conve_infer_workspace_bytes(md, n);
}

int kge_conve_score(const kge_model_desc* md, const float* E, const float* Rel, const float* theta, float* stats, const int32_t* triples,
                    int64_t n, float* out, void* workspace, int64_t workspace_bytes, uint32_t* err_flag, kge_stream_t stream) {
  if (int rc = check_conve(md, "kge_conve_score")) return rc;
  if (n == 0) return KGE_OK;
  if (n < 0 || !E || !Rel || !theta || !stats || !triples || !out || !workspace || !err_flag) return fail(KGE_EINVAL, "kge_conve_score: bad arguments");
  return conve_infer(md, E, Rel, theta, stats, triples, n, out, 0, workspace, workspace_bytes, err_flag, static_cast<hipStream_t>(stream));
}

int kge_conve_hidden(const kge_model_desc* md, const float* E, const float* Rel, const float* theta, float* stats, const int32_t* triples,
                     int64_t n, float* hidden_out, void* workspace, int64_t workspace_bytes, uint32_t* err_flag, kge_stream_t stream) {
  if (int rc = check_conve(md, "kge_conve_hidden")) return rc;
  if (n == 0) return KGE_OK;
  if (n < 0 || !E || !Rel || !theta || !stats || !triples || !hidden_out || !workspace || !err_flag) return fail(KGE_EINVAL, "kge_conve_hidden: bad arguments");
  return conve_infer(md, E, Rel, theta, stats, triples, n, hidden_out, 1, workspace, workspace_bytes, err_flag, static_cast<hipStream_t>(stream));
}

int kge_conve_train_grads(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, const float* theta,
                          float* stats, float* gradE, float* gradRel, float* gtheta, const int32_t* pos, int64_t m_pos,
                          const int32_t* neg, int64_t m_neg, float* loss_out, float* scores_out, uint64_t seed, void* workspace,
                          int64_t workspace_bytes, uint32_t* err_flag, kge_stream_t stream) {
  if (int rc = check_conve(md, "kge_conve_train_grads")) return rc;
  if (int rc = check_train("kge_conve_train_grads", md, ld, E, Rel, gradE, gradRel, pos, m_pos, neg, m_neg, loss_out, err_flag)) return rc;
  if (!theta || !stats || !gtheta || !workspace) return fail(KGE_EINVAL, "kge_conve_train_grads: null pointer");
  return conve_train_grads(md, ld, E, Rel, theta, stats, gradE, gradRel, gtheta, pos, m_pos, neg, m_neg, loss_out, scores_out, seed, workspace,
                           workspace_bytes, err_flag, static_cast<hipStream_t>(stream));
}

int kge_opt_apply_vector(const kge_opt_desc* od, float* w, float* g, int64_t n, float grad_scale, kge_stream_t stream) {
  clear_stale_error();
  if (!od || !w || !g || n < 0) return fail(KGE_EINVAL, "kge_opt_apply_vector: bad arguments");
  return opt_apply_vector(od, w, g, n, grad_scale, static_cast<hipStream_t>(stream));
}

int64_t kge_rank_sweep_q_workspace_bytes(int64_t n_queries, int32_t K) {
  if (n_queries < 0 || K <= 0) return fail(KGE_EINVAL, "kge_rank_sweep_q_workspace_bytes: bad sizes");
  return rank_sweep_q_workspace_bytes(n_queries, K);
}

int kge_rank_sweep_q(const float* Q, int64_t n_queries, int32_t K, const float* Cand, int32_t cand_offset, int32_t cand_rows,
                     const int32_t* target_ids, const int32_t* filt_query, const int32_t* filt_ids, int64_t n_filt, int32_t* counts,
                     int32_t accumulate, void* workspace, int64_t workspace_bytes, kge_stream_t stream) {
  clear_stale_error();
  if (n_queries == 0) return KGE_OK;
  if (n_queries < 0 || K <= 0 || !Q || !Cand || !target_ids || !counts || !workspace || cand_rows <= 0 || cand_offset < 0 || n_filt < 0 ||
      (n_filt && (!filt_query || !filt_ids)))
    return fail(KGE_EINVAL, "kge_rank_sweep_q: bad arguments");
  return rank_sweep_q(Q, n_queries, K, Cand, cand_offset, cand_rows, target_ids, filt_query, filt_ids, n_filt, counts, accumulate, workspace,
                      workspace_bytes, static_cast<hipStream_t>(stream));
}

int32_t kge_hole_spec_width(int32_t d) {
  if (d <= 0) return fail(KGE_EINVAL, "kge_hole_spec_width: d must be positive");
  return 2 * hole_spec_half(d);
}

int kge_hole_spec_basis(int32_t d, float* WE, float* WR, kge_stream_t stream) {
  clear_stale_error();
  if (d <= 0 || !WE || !WR) return fail(KGE_EINVAL, "kge_hole_spec_basis: bad arguments");
  return hole_spec_basis(d, WE, WR, static_cast<hipStream_t>(stream));
}

int kge_hole_spec_forward(const float* X, int64_t rows, int32_t d, const float* W, float* out, kge_stream_t stream) {
  clear_stale_error();
  if (rows < 0 || rows > 0x7fffffffLL || d <= 0 || (rows && (!X || !W || !out))) return fail(KGE_EINVAL, "kge_hole_spec_forward: bad arguments");
  return hole_spec_forward(X, rows, d, W, out, static_cast<hipStream_t>(stream));
}

int kge_hole_spec_adjoint(float* Ghat, int64_t rows, int32_t d, const float* W, float* G, int32_t zero_hat, kge_stream_t stream) {
  clear_stale_error();
  if (rows < 0 || rows > 0x7fffffffLL || d <= 0 || (rows && (!Ghat || !W || !G))) return fail(KGE_EINVAL, "kge_hole_spec_adjoint: bad arguments");
  return hole_spec_adjoint(Ghat, rows, d, W, G, zero_hat, static_cast<hipStream_t>(stream));
}

int64_t kge_lit_workspace_bytes(const kge_model_desc* md, int32_t L, int64_t m_pos, int64_t m_neg) {
  if (check_model(md, "kge_lit_workspace_bytes")) return KGE_EINVAL;
  if (L <= 0 || m_pos < 0 || m_neg < 0) return fail(KGE_EINVAL, "kge_lit_workspace_bytes: bad sizes");
  return lit_workspace_bytes(md, L, m_pos, m_neg);
}

int64_t kge_lit_transform_workspace_bytes(const kge_model_desc* md, int32_t L, int32_t rows) {
  if (check_model(md, "kge_lit_transform_workspace_bytes")) return KGE_EINVAL;
  if (L <= 0 || rows < 0) return fail(KGE_EINVAL, "kge_lit_transform_workspace_bytes: bad sizes");
  return lit_transform_workspace_bytes(md, L, rows);
}

int kge_lit_transform(const kge_model_desc* md, int32_t act, const float* E, const float* Lit, int32_t L, int32_t lit_rows, const float* theta_l,
                      int32_t rows, float* out, void* workspace, int64_t workspace_bytes, uint32_t* err_flag, kge_stream_t stream) {
  if (int rc = check_model(md, "kge_lit_transform")) return rc;
  if (rows == 0) return KGE_OK;
  if (rows < 0 || rows > md->num_entities || !E || !Lit || !theta_l || !out || !workspace || !err_flag) return fail(KGE_EINVAL, "kge_lit_transform: bad arguments");
  return lit_transform(md, act, E, Lit, L, lit_rows, theta_l, rows, out, workspace, workspace_bytes, err_flag, static_cast<hipStream_t>(stream));
}

int kge_lit_train_grads(const kge_model_desc* md, const kge_loss_desc* ld, int32_t act, const float* E, const float* Rel, const float* Lit,
                        int32_t L, int32_t lit_rows, const float* theta_l, float* gradE, float* gradRel, float* gtheta_l, const int32_t* pos,
                        int64_t m_pos, const int32_t* neg, int64_t m_neg, float* loss_out, float* scores_out, uint64_t seed, void* workspace,
                        int64_t workspace_bytes, uint32_t* err_flag, kge_stream_t stream) {
  if (int rc = check_train("kge_lit_train_grads", md, ld, E, Rel, gradE, gradRel, pos, m_pos, neg, m_neg, loss_out, err_flag)) return rc;
  if (!Lit || !theta_l || !gtheta_l || !workspace) return fail(KGE_EINVAL, "kge_lit_train_grads: null pointer");
  return lit_train_grads(md, ld, act, E, Rel, Lit, L, lit_rows, theta_l, gradE, gradRel, gtheta_l, pos, m_pos, neg, m_neg, loss_out, scores_out, seed,
                         workspace, workspace_bytes, err_flag, static_cast<hipStream_t>(stream));
}

int kge_lit_conve_train_grads(const kge_model_desc* md, const kge_loss_desc* ld, int32_t act, const float* E, const float* Rel, const float* Lit,
                              int32_t L, int32_t lit_rows, const float* theta_l, const float* theta_c, float* bn_stats, float* gradE, float* gradRel,
                              float* gtheta_l, float* gtheta_c, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, float* loss_out,
                              float* scores_out, uint64_t seed, void* workspace, int64_t workspace_bytes, uint32_t* err_flag, kge_stream_t stream) {
  if (int rc = check_train("kge_lit_conve_train_grads", md, ld, E, Rel, gradE, gradRel, pos, m_pos, neg, m_neg, loss_out, err_flag)) return rc;
  if (md->model != KGE_CONVE) return fail(KGE_EINVAL, "kge_lit_conve_train_grads: the descriptor is not a ConvE one");
  if (!Lit || !theta_l || !gtheta_l || !theta_c || !bn_stats || !gtheta_c || !workspace) return fail(KGE_EINVAL, "kge_lit_conve_train_grads: null pointer");
  const LitConve cv{theta_c, bn_stats, gtheta_c};
  return lit_train_grads(md, ld, act, E, Rel, Lit, L, lit_rows, theta_l, gradE, gradRel, gtheta_l, pos, m_pos, neg, m_neg, loss_out, scores_out, seed,
                         workspace, workspace_bytes, err_flag, static_cast<hipStream_t>(stream), &cv);
}

int kge_lit_hake_train_grads(const kge_model_desc* md, const kge_loss_desc* ld, const kge_hake_params* hp, int32_t act, const float* E, const float* Rel,
                             const float* Lit, int32_t L, int32_t lit_rows, const float* theta_l, float* gradE, float* gradRel, float* gtheta_l,
                             const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, float* loss_out, float* scores_out, uint64_t seed,
                             void* workspace, int64_t workspace_bytes, uint32_t* err_flag, kge_stream_t stream) {
  if (int rc = check_train("kge_lit_hake_train_grads", md, ld, E, Rel, gradE, gradRel, pos, m_pos, neg, m_neg, loss_out, err_flag)) return rc;
  if (md->model != KGE_HAKE || !hp) return fail(KGE_EINVAL, "kge_lit_hake_train_grads: needs a HAKE descriptor and its parameters");
  if (!Lit || !theta_l || !gtheta_l || !workspace) return fail(KGE_EINVAL, "kge_lit_hake_train_grads: null pointer");
  return lit_train_grads(md, ld, act, E, Rel, Lit, L, lit_rows, theta_l, gradE, gradRel, gtheta_l, pos, m_pos, neg, m_neg, loss_out, scores_out, seed,
                         workspace, workspace_bytes, err_flag, static_cast<hipStream_t>(stream), nullptr, hp);
}

// ---- RCCL collectives of the sharded evaluation and of data-parallel training, on a communicator the caller made (ncclComm_t passed
// as void*).  RCCL is resolved at run time from the process (PyTorch-ROCm brings its own librccl) or from librccl.so.1: the library
// has no link-time dependency on it and single-GPU users never load it.
namespace {
typedef int (*nccl_allgather_t)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*nccl_allreduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*nccl_errstr_t)(int);
struct Rccl { nccl_allgather_t allgather = nullptr; nccl_allreduce_t allreduce = nullptr; nccl_errstr_t errstr = nullptr; };
const Rccl* rccl() {
  static const Rccl r = [] {
    Rccl x;
    void* h = RTLD_DEFAULT;
    if (!dlsym(h, "ncclAllReduce")) {
      h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (!h) return x;
    }
    x.allgather = reinterpret_cast<nccl_allgather_t>(dlsym(h, "ncclAllGather"));
    x.allreduce = reinterpret_cast<nccl_allreduce_t>(dlsym(h, "ncclAllReduce"));
    x.errstr = reinterpret_cast<nccl_errstr_t>(dlsym(h, "ncclGetErrorString"));
    return x;
  }();
  return (r.allgather && r.allreduce) ? &r : nullptr;
}
constexpr int kNcclInt32 = 2, kNcclFloat32 = 7, kNcclSum = 0;  // ncclDataType_t / ncclRedOp_t values (nccl.h)
int rccl_fail(const Rccl* r, const char* who, int rc) {
  return fail(KGE_EHIP, "%s: RCCL error %d (%s)", who, rc, r->errstr ? r->errstr(rc) : "?");
}
}  // namespace

int kge_allgather_rows(void* comm, const float* shard, int64_t rows_per_rank, int32_t d, float* full, kge_stream_t stream) {
  clear_stale_error();
  if (!comm || !shard || !full || rows_per_rank < 0 || d <= 0) return fail(KGE_EINVAL, "kge_allgather_rows: bad arguments");
  const Rccl* r = rccl();
  if (!r) return fail(KGE_EUNSUPPORTED, "kge_allgather_rows: RCCL (librccl.so.1) is not available in this process");
  const int rc = r->allgather(shard, full, (size_t)rows_per_rank * (size_t)d, kNcclFloat32, comm, static_cast<hipStream_t>(stream));
  return rc == 0 ? KGE_OK : rccl_fail(r, "kge_allgather_rows", rc);
}

int kge_allreduce_counts(void* comm, int32_t* counts, int64_t n, kge_stream_t stream) {
  clear_stale_error();
  if (!comm || !counts || n < 0) return fail(KGE_EINVAL, "kge_allreduce_counts: bad arguments");
  const Rccl* r = rccl();
  if (!r) return fail(KGE_EUNSUPPORTED, "kge_allreduce_counts: RCCL (librccl.so.1) is not available in this process");
  const int rc = r->allreduce(counts, counts, (size_t)n, kNcclInt32, kNcclSum, comm, static_cast<hipStream_t>(stream));
  return rc == 0 ? KGE_OK : rccl_fail(r, "kge_allreduce_counts", rc);
}

int kge_allreduce_f32(void* comm, float* buf, int64_t n, kge_stream_t stream) {
  clear_stale_error();
  if (!comm || !buf || n < 0) return fail(KGE_EINVAL, "kge_allreduce_f32: bad arguments");
  const Rccl* r = rccl();
  if (!r) return fail(KGE_EUNSUPPORTED, "kge_allreduce_f32: RCCL (librccl.so.1) is not available in this process");
  const int rc = r->allreduce(buf, buf, (size_t)n, kNcclFloat32, kNcclSum, comm, static_cast<hipStream_t>(stream));
  return rc == 0 ? KGE_OK : rccl_fail(r, "kge_allreduce_f32", rc);
}

}  // extern "C"
