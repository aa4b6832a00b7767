"""ctypes binding of libkge_hip.so (include/kge_hip.h).  No fallback: a missing library is an ImportError."""
import ctypes as C
import os

# PyTorch-ROCm owns the device memory and the streams, so libkge_hip.so must talk to the SAME HIP runtime instance PyTorch
# loaded.  PyTorch wheels bundle their own libamdhip64 / librocblas; loading ours first would pull /opt/rocm's copies into the
# process beside them, and every launch on a torch pointer would then fail with "no ROCm-capable device is detected".  With
# torch imported first the loader resolves our DT_NEEDED entries (same SONAMEs) to the copies already in the process.
import torch  # noqa: F401  (must precede the CDLL below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KGE_HIP_LIB") or os.path.join(_HERE, "csrc", "libkge_hip.so")  # (KGE_HIP_LIB: another build of the same library, for A/B timing)

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
        "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
lib = C.CDLL(LIB_PATH)
ABI_VERSION = 10  # KGE_ABI_VERSION of include/kge_hip.h this binding was written against (struct layouts, argument lists, the 24-byte step state)
lib.kge_abi_version.restype = C.c_int
if lib.kge_abi_version() != ABI_VERSION:
    raise ImportError(f"{LIB_PATH} reports ABI version {lib.kge_abi_version()}, this binding needs {ABI_VERSION}: rebuild the library "
                      f"(`make -C {os.path.join(_HERE, 'csrc')}`) -- a stale build would mis-read descriptors silently.")

KGE_OK, KGE_EINVAL, KGE_ESHAPE, KGE_EHIP, KGE_EWORKSPACE, KGE_EUNSUPPORTED = 0, -1, -2, -3, -4, -5
MODEL_IDS = {"DistMult": 0, "TransE": 1, "ComplEx": 2, "HolE": 3, "ConvE": 4, "RotatE": 5, "HAKE": 6}
LOSS_IDS = {"pointwize_hinge": 0, "pointwize_logistic": 1, "pairwize_hinge": 2, "pairwize_logistic": 3,
            "pointwize_square_loss": 4, "pairwize_square_loss": 5, "pointwize_cross_entropy": 6}
OPT_IDS = {"sgd": 0, "adam": 1}
ACT_IDS = {None: 0, "linear": 0, "relu": 1, "sigmoid": 2, "tanh": 3, "softplus": 4, "softsign": 5, "elu": 6, "selu": 7, "exponential": 8, "softmax": 9}
MT_STATE_BYTES = 2504
CAND_POS_LANES = 16  # KGE_CAND_POS_LANES
STEP_STATE_WORDS = 3  # KGE_STEP_STATE_BYTES / 8: {int64 updates done, float lr_t of the next update + pad, double loss sum}


class ModelDesc(C.Structure):
    _fields_ = [("model", C.c_int32), ("num_entities", C.c_int32), ("num_relations", C.c_int32), ("d_e", C.c_int32),
                ("d_r", C.c_int32), ("norm", C.c_int32), ("gamma", C.c_float), ("dropout", C.c_float),
                ("conv_filters", C.c_int32), ("hidden_dropout", C.c_float), ("modulus", C.c_float),
                ("conv_kh", C.c_int32), ("conv_kw", C.c_int32)]


class LossDesc(C.Structure):
    _fields_ = [("loss", C.c_int32), ("negative_samples", C.c_int32), ("margin", C.c_float), ("regularization", C.c_float)]


class OptDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("step", C.c_int64), ("mE", C.c_void_p), ("vE", C.c_void_p), ("mRel", C.c_void_p), ("vRel", C.c_void_p),
                ("step_dev", C.c_void_p)]


class RowWindow(C.Structure):
    """kge_row_window: the rows of both tables one rank of an owner-computes step updates."""
    _fields_ = [("ent_lo", C.c_int32), ("ent_hi", C.c_int32), ("rel_lo", C.c_int32), ("rel_hi", C.c_int32)]


class HakeParams(C.Structure):
    _fields_ = [("phase_weight", C.c_float), ("mod_weight", C.c_float), ("embedding_range", C.c_float)]


class PipelineDesc(C.Structure):
    """kge_pipeline_desc (include/kge_hip.h): everything the native pipeline driver borrows."""
    _fields_ = [("md", ModelDesc), ("ld", LossDesc), ("od", OptDesc),
                ("E", C.c_void_p), ("Rel", C.c_void_p), ("gradE", C.c_void_p), ("gradRel", C.c_void_p), ("loss_out", C.c_void_p),
                ("train_ws", C.c_void_p), ("train_ws_bytes", C.c_int64), ("err_flag", C.c_void_p),
                ("np_state", C.c_void_p), ("py_state", C.c_void_p), ("kg", C.c_void_p), ("index", C.c_void_p), ("starts", C.c_void_p),
                ("cursor", C.c_void_p), ("B", C.c_int64), ("ns", C.c_int64), ("num_e", C.c_int64),
                ("sampler_plan", C.c_void_p), ("sampler_ws", C.c_void_p), ("sampler_ws_bytes", C.c_int64), ("sampler_err", C.c_void_p),
                ("sampler_stream", C.c_void_p), ("sampler_stream_bytes", C.c_int64), ("sampler_stream_plan", C.c_void_p),
                ("cand_rows", C.c_void_p * 2), ("cand_counts", C.c_void_p * 2), ("plan", C.c_void_p * 2), ("plan_bytes", C.c_int64),
                ("pos", C.c_void_p * 2), ("neg", C.c_void_p * 2), ("E_twin", C.c_void_p), ("Rel_twin", C.c_void_p), ("plan3", C.c_void_p), ("plan_more", C.c_void_p * 3),
                ("comm", C.c_void_p), ("world", C.c_int32), ("grad_flat", C.c_void_p), ("grad_flat_n", C.c_int64)]


_P, _I64, _I32, _U64, _F = C.c_void_p, C.c_int64, C.c_int32, C.c_uint64, C.c_float
_MD, _LD, _OD = C.POINTER(ModelDesc), C.POINTER(LossDesc), C.POINTER(OptDesc)

# every symbol include/kge_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "kge_last_error": (C.c_char_p, []),
    "kge_abi_version": (C.c_int, []),
    "kge_device_props": (C.c_int, [_P]),
    "kge_rank_workspace_bytes": (_I64, [_MD, _I64]),
    "kge_rank_counts": (C.c_int, [_MD, _P, _P, _P, _I32, _I32, _P, _I64, _I32, _P, _P, _I64, _P, _I32, _P, _P, _I64, _P, _P]),
    "kge_rank_chain_scores": (C.c_int, [_MD, _P, _P, _P, _I64, _I32, _P, _P, _I64, _P, _P]),
    "kge_conve_sizes": (C.c_int, [_MD, _P, _P]),
    "kge_conve_workspace_bytes": (_I64, [_MD, _I64, _I64]),
    "kge_conve_infer_workspace_bytes": (_I64, [_MD, _I64]),
    "kge_conve_score": (C.c_int, [_MD, _P, _P, _P, _P, _P, _I64, _P, _P, _I64, _P, _P]),
    "kge_conve_hidden": (C.c_int, [_MD, _P, _P, _P, _P, _P, _I64, _P, _P, _I64, _P, _P]),
    "kge_conve_train_grads": (C.c_int, [_MD, _LD, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _I64, _P, _P, _U64, _P, _I64, _P, _P]),
    "kge_lit_workspace_bytes": (_I64, [_MD, _I32, _I64, _I64]),
    "kge_lit_transform_workspace_bytes": (_I64, [_MD, _I32, _I32]),
    "kge_lit_transform": (C.c_int, [_MD, _I32, _P, _P, _I32, _I32, _P, _I32, _P, _P, _I64, _P, _P]),
    "kge_lit_train_grads": (C.c_int, [_MD, _LD, _I32, _P, _P, _P, _I32, _I32, _P, _P, _P, _P, _P, _I64, _P, _I64, _P, _P, _U64, _P, _I64, _P, _P]),
    "kge_lit_conve_train_grads": (C.c_int, [_MD, _LD, _I32, _P, _P, _P, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _I64, _P, _P, _U64, _P, _I64, _P, _P]),
    "kge_lit_hake_train_grads": (C.c_int, [_MD, _LD, _P, _I32, _P, _P, _P, _I32, _I32, _P, _P, _P, _P, _P, _I64, _P, _I64, _P, _P, _U64, _P, _I64, _P, _P]),
    "kge_hole_spec_width": (_I32, [_I32]),
    "kge_hole_spec_basis": (C.c_int, [_I32, _P, _P, _P]),
    "kge_hole_spec_forward": (C.c_int, [_P, _I64, _I32, _P, _P, _P]),
    "kge_hole_spec_adjoint": (C.c_int, [_P, _I64, _I32, _P, _P, _I32, _P]),
    "kge_hake_workspace_bytes": (_I64, [_I64, _I64]),
    "kge_hake_score": (C.c_int, [_MD, C.POINTER(HakeParams), _P, _P, _P, _I64, _I64, _P, _I64, _P, _P, _I64, _P, _P]),
    "kge_hake_rank_counts": (C.c_int, [_P, _I64, _P, _I64, _P, _P]),
    "kge_hake_train_grads": (C.c_int, [_MD, _LD, C.POINTER(HakeParams), _P, _P, _P, _P, _P, _I64, _P, _I64, _P, _P, _U64, _P, _I64, _P, _P]),
    "kge_opt_state_init": (C.c_int, [_OD, _P]),
    "kge_opt_apply_vector": (C.c_int, [_OD, _P, _P, _I64, _F, _P]),
    "kge_rank_sweep_q_workspace_bytes": (_I64, [_I64, _I32]),
    "kge_rank_sweep_q": (C.c_int, [_P, _I64, _I32, _P, _I32, _I32, _P, _P, _P, _I64, _P, _I32, _P, _I64, _P]),
    "kge_loss_workspace_bytes": (_I64, [_I64, _I64]),
    "kge_loss_value": (C.c_int, [_LD, _P, _I64, _P, _I64, _P, _P, _P, _P, _I64, _P]),
    "kge_batch_plan_bytes": (_I64, [_MD, _LD, _I64, _I64]),
    "kge_batch_plan_build": (C.c_int, [_MD, _LD, _P, _I64, _P, _I64, _P, _I64, _I32, _P, _P]),
    "kge_batch_plan_build_counted": (C.c_int, [_MD, _LD, _P, _P, _I32, _I64, _I64, _I64, _I64, _P, _I64, _I32, _P, _P]),
    "kge_train_grads_planned": (C.c_int, [_MD, _LD, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P, C.c_uint64, _P, _I64, _P]),
    "kge_train_step_planned": (C.c_int, [_MD, _LD, _OD, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P, C.c_uint64, _P, _I64, _P]),
    "kge_pipeline_set_seed": (C.c_int, [_P, C.c_uint64]),
    "kge_train_grads_halves_supported": (C.c_int, [_MD, _LD]),
    "kge_train_grads_planned_parts": (C.c_int, [_MD, _LD, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P, C.c_uint64, _P, _I64, _I32, _P]),
    "kge_train_step_swap_supported": (C.c_int, [_MD, _LD, _OD]),
    "kge_train_step_swap": (C.c_int, [_MD, _LD, _OD, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P, C.c_uint64, _P, _I64, _I32, _P]),
    "kge_train_step_swap_window": (C.c_int, [_MD, _LD, _OD, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P, C.c_uint64, _P, _I64, _P, _P]),
    "kge_train_step_swap_chain": (C.c_int, [_MD, _LD, _OD, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P, C.c_uint64, _P, _I64, _I64, _P, _P]),
    "kge_loss_pairs": (C.c_int, [_LD, _P, _I64, _P, _I64, _P, _I64, _P]),
    "kge_set_grad_mode": (C.c_int, [C.c_int]),
    "kge_set_dedup_mode": (C.c_int, [C.c_int]),
    "kge_set_pair_sort_mode": (C.c_int, [C.c_int]),
    "kge_set_deterministic": (C.c_int, [C.c_int]),
    "kge_get_deterministic": (C.c_int, []),
    "kge_prof_enable": (C.c_int, [C.c_int]),
    "kge_prof_read": (C.c_int, [C.c_int, _P, _P]),
    "kge_score": (C.c_int, [_MD, _P, _P, _P, _I64, _P, _P, _P]),
    "kge_score_training": (C.c_int, [_MD, _P, _P, _P, _I64, _U64, _I64, _P, _P, _P]),
    "kge_loss_only": (C.c_int, [_MD, _LD, _P, _P, _P, _I64, _P, _I64, _P, _P, _P, _I64, _P, _P]),
    "kge_l3_value": (C.c_int, [_P, _I64, _F, _P, _P, _I64, _P]),
    "kge_train_step_workspace_bytes": (_I64, [_MD, _LD, _I64, _I64]),
    "kge_train_step": (C.c_int, [_MD, _LD, _OD, _P, _P, _P, _P, _P, _I64, _P, _I64, _P, _P, _U64, _P, _I64, _P, _P]),
    "kge_train_grads": (C.c_int, [_MD, _LD, _P, _P, _P, _P, _P, _I64, _P, _I64, _P, _P, _U64, _P, _I64, _P, _P]),
    "kge_opt_apply": (C.c_int, [_MD, _LD, _OD, _P, _P, _P, _P, _F, _P, _P, _I64, _P]),
    "kge_mt_seed_numpy": (C.c_int, [_P, C.c_uint32]),
    "kge_mt_seed_python": (C.c_int, [_P, _P, C.c_int]),
    "kge_mt_randint": (C.c_int, [_P, _I64, _I64, _P, _I64]),
    "kge_mt_shuffle": (C.c_int, [_P, _P, _I64]),
    "kge_mt_choices": (C.c_int, [_P, _I64, _P, _I64]),
    "kge_mt_raw": (C.c_int, [_P, _P, _I64]),
    "kge_filter_coo_host": (C.c_int, [_P, _I64, _I32, _P, _I64, _I64, _P, _P, _P]),
    "kge_sampler_workspace_bytes": (_I64, [_I64, _I64, _I64]),
    "kge_sampler_plan_bytes": (_I64, []),
    "kge_sampler_plan_host": (C.c_int, [_I64, _I64, _I64, _P]),
    "kge_build_batch_device": (C.c_int, [_P, _P, _P, _I64, _I64, _I64, _P, _P, _P, _P, _I64, _P, _P]),
    "kge_build_batch_device_indexed": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _I64, _P, _P]),
    "kge_mt_jump_poly": (C.c_int, [_I64, _P]),
    "kge_mt_jump_host": (C.c_int, [_P, _I64]),
    "kge_build_batch_host": (C.c_int, [_P, _P, _P, _I64, _I64, _I64, _P, _P]),
    "kge_allgather_rows": (C.c_int, [_P, _P, _I64, _I32, _P, _P]),
    "kge_allreduce_counts": (C.c_int, [_P, _P, _I64, _P]),
    "kge_allreduce_f32": (C.c_int, [_P, _P, _I64, _P]),
    "kge_kg_open": (C.c_int, [C.POINTER(C.c_char_p), _I32, C.POINTER(C.c_void_p)]),
    "kge_kg_sizes": (C.c_int, [_P, C.POINTER(_I64), C.POINTER(_I64), _P]),
    "kge_kg_triples": (C.c_int, [_P, _I32, _P]),
    "kge_kg_names": (C.c_int, [_P, _I32, _P, _I64, _P]),
    "kge_kg_close": (C.c_int, [_P]),
    "kge_sampler_stream_bytes": (_I64, [_I64, _I64, _I64]),
    "kge_sampler_stream_round": (_I32, []),
    "kge_sampler_stream_plan_host": (C.c_int, [_I64, _I64, _I64, _P]),
    "kge_sampler_stream_init": (C.c_int, [_P, _I64, _P, _P, _I64, _I64, _I64, _P]),
    "kge_sampler_stream_produce": (C.c_int, [_P, _I64, _I64, _I64, _I64, _P, _P]),
    "kge_sampler_stream_draw": (C.c_int, [_P, _I64, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P]),
    "kge_pipeline_create": (C.c_int, [C.POINTER(PipelineDesc), _P, C.POINTER(C.c_void_p)]),
    "kge_pipeline_start": (C.c_int, [_P, _I64, _P]),
    "kge_pipeline_run": (C.c_int, [_P, _I64, _P]),
    "kge_pipeline_finish": (C.c_int, [_P, _P]),
    "kge_pipeline_state": (C.c_int, [_P, _P]),
    "kge_pipeline_stats": (C.c_int, [_P, _P, _I32]),
    "kge_pipeline_trace": (_I64, [_P, _I64, _P, _I64]),
    "kge_pipeline_destroy": (C.c_int, [_P]),
}
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = header and library out of sync
    _fn.restype = _res
    _fn.argtypes = _args


PROF_LABELS = {"train_grads": 0, "opt": 1, "score": 2, "pairwise": 3, "rank": 4, "sampler": 5, "fwd": 6, "index": 7, "dedup": 8, "rank_sweep": 9}


def prof_read(label):
    """(total_ms, launches) recorded for one kernel label since the last read (kge_prof_read)."""
    ms, n = C.c_double(0.0), C.c_int64(0)
    check(lib.kge_prof_read(PROF_LABELS[label], C.byref(ms), C.byref(n)))
    return ms.value, n.value


class KgeError(RuntimeError):
    pass


_EXC = {KGE_EINVAL: ValueError, KGE_ESHAPE: ValueError, KGE_EHIP: KgeError, KGE_EWORKSPACE: KgeError,
        KGE_EUNSUPPORTED: NotImplementedError}


def check(rc):
    """Raise the Python exception matching a negative kge_status."""
    if rc < 0:
        msg = lib.kge_last_error().decode("utf-8", "replace")
        raise _EXC.get(rc, KgeError)(f"[libkge_hip {rc}] {msg}")
    return rc


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("kge_keras_amd needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
