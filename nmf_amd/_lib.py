"""ctypes binding of libnmfx.so (the C ABI declared in include/nmfx.h).

The library is the product's only compute path: if it is missing or no MI355X
is visible, loading fails loudly -- there is no CPU fallback in this package.
"""
import ctypes as C
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NMFX_LIB") or os.path.join(_HERE, "lib", "libnmfx.so")     # NMFX_LIB: an experiment build (A/B runs on one box)

NMFX_OK, NMFX_E_ARG, NMFX_E_HIP, NMFX_E_NOTPD, NMFX_E_STATE, NMFX_E_NOMEM, NMFX_E_RCCL = 0, -1, -2, -3, -4, -5, -6
F32, F64 = 0, 1
EU, KL = 0, 1
PROX = {"nn": 0, "l1n": 1, "l2n": 2, "l1inf": 3, "l1inf_transpose": 4}

_i64, _i32, _dbl, _vp = C.c_int64, C.c_int, C.c_double, C.c_void_p
_pd = C.POINTER(C.c_double)

# name -> (restype, argtypes); mirrors include/nmfx.h one to one
SIGNATURES = {
    "nmfx_create": (_i32, [C.POINTER(_vp), _i32, _i64, _i64, _i32]),
    "nmfx_destroy": (_i32, [_vp]),
    "nmfx_last_error": (C.c_char_p, [_vp]),
    "nmfx_version": (_i32, []),
    "nmfx_device_count": (_i32, []),
    "nmfx_set_stream": (_i32, [_vp, _vp]),
    "nmfx_reset_stream": (_i32, [_vp]),
    "nmfx_synchronize": (_i32, [_vp]),
    "nmfx_set_precision": (_i32, [_vp, _i32]),
    "nmfx_get_precision": (_i32, [_vp]),
    "nmfx_get_note": (C.c_char_p, [_vp]),
    "nmfx_upload_v": (_i32, [_vp, _vp, _i32, _i64, _i64, _i64]),
    "nmfx_upload_v_device": (_i32, [_vp, _vp, _i32, _i64, _i64, _i64]),
    "nmfx_set_factors": (_i32, [_vp, _vp, _vp]),
    "nmfx_get_factors": (_i32, [_vp, _vp, _vp]),
    "nmfx_get_matrix": (_i32, [_vp, C.c_char_p, _vp]),
    "nmfx_set_matrix": (_i32, [_vp, C.c_char_p, _vp]),
    "nmfx_prox_apply": (_i32, [_vp, _i32, _i32, _dbl, _dbl, _i32]),
    "nmfx_get_state": (_i32, [_vp, C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i64)]),
    "nmfx_get_objectives": (_i32, [_vp, _i64, _i64, _vp]),
    "nmfx_mur_run": (_i32, [_vp, _i32, _dbl, _dbl, _i64, _dbl, _dbl, _i64, _i64]),
    "nmfx_mur_finish": (_i32, [_vp, _i32, _i64, _dbl, _dbl, _i64]),
    "nmfx_mur_phase_a": (_i32, [_vp, _i32, _dbl, _i64]),
    "nmfx_mur_phase_b": (_i32, [_vp, _i32, _dbl, _i64, _dbl, _dbl, _i64]),
    "nmfx_mur_finish_a": (_i32, [_vp, _i32, _i64]),
    "nmfx_mur_finish_b": (_i32, [_vp, _i64, _dbl, _dbl, _i64]),
    "nmfx_exchange_sizes": (_i32, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "nmfx_aoadmm_phase_h_products": (_i32, [_vp, _i64]),
    "nmfx_aoadmm_phase_h_solve": (_i32, [_vp, _i32, _dbl, _i32, _i64, _dbl, _dbl, _i64]),
    "nmfx_aoadmm_phase_w_products": (_i32, [_vp, _i64, _dbl, _dbl, _i64]),
    "nmfx_aoadmm_phase_w_round": (_i32, [_vp, _i32, _dbl, _i32]),
    "nmfx_aoadmm_phase_w_close": (_i32, [_vp, _i32, _i64]),
    "nmfx_aoadmm_phase_w_fused": (_i32, [_vp, _i32, _dbl, _i32]),
    "nmfx_aoadmm_phase_w_repair": (_i32, [_vp, _i32, _dbl, _i32, _i64]),
    "nmfx_aoadmm_kl_phase_h_products": (_i32, [_vp, _i64, _i32]),
    "nmfx_aoadmm_kl_phase_h_round": (_i32, [_vp, _i32, _dbl, _i32, _i64, _dbl, _dbl, _i64]),
    "nmfx_aoadmm_kl_phase_h_close": (_i32, [_vp, _i32, _i64, _dbl, _dbl, _i64]),
    "nmfx_aoadmm_kl_phase_w_round": (_i32, [_vp, _i32, _dbl, _i32]),
    "nmfx_aoadmm_kl_phase_w_close": (_i32, [_vp, _i32, _i64]),
    "nmfx_objective_partial": (_i32, [_vp]),
    "nmfx_anls_phase_objective": (_i32, [_vp, _i64]),
    "nmfx_anls_phase_w": (_i32, [_vp, _dbl, _i64, _dbl, _dbl, _i64]),
    "nmfx_anls_phase_h": (_i32, [_vp, _dbl, _i64]),
    "nmfx_topk_svd": (_i32, [_vp, _i32, _i32, _dbl, _i32, C.c_uint64, _vp, _vp, _vp, C.POINTER(_i32), C.POINTER(_dbl)]),
    "nmfx_reserve_objectives": (_i32, [_vp, _i64]),
    "nmfx_shift_iteration_base": (_i32, [_vp, _i64]),
    "nmfx_set_exchange_buffers": (_i32, [_vp, _vp, _i64, _vp, _i64]),
    "nmfx_comm_unique_id": (_i32, [_vp]),
    "nmfx_comm_init_rank": (_i32, [_vp, _vp, _i32, _i32]),
    "nmfx_comm_destroy": (_i32, [_vp]),
    "nmfx_comm_info": (_i32, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "nmfx_comm_negotiate": (_i32, [_vp]),
    "nmfx_comm_all_reduce": (_i32, [_vp, _i32, _i64, _i64]),
    "nmfx_comm_all_min": (_i32, [_vp, C.POINTER(_i64), _i32]),
    "nmfx_comm_barrier": (_i32, [_vp]),
    "nmfx_comm_set_graph": (_i32, [_vp, _i32]),
    "nmfx_comm_graph_replays": (_i32, [_vp, C.POINTER(_i64)]),
    "nmfx_comm_set_exchange": (_i32, [_vp, _i32]),
    "nmfx_comm_get_exchange": (_i32, [_vp, C.POINTER(_i32)]),
    "nmfx_mur_slice_info": (_i32, [_vp, _i32, _i32, C.POINTER(_i64), C.POINTER(_i64)]),
    "nmfx_mur_phase_b_slice": (_i32, [_vp, _i32, _dbl, _i64, _dbl, _dbl, _i64, _i64, _i64]),
    "nmfx_mur_phase_b_rest": (_i32, [_vp, _i32, _i64, _i64]),
    "nmfx_mur_run_sharded": (_i32, [_vp, _i32, _dbl, _dbl, _i64, _dbl, _dbl, _i64, _i64]),
    "nmfx_mur_finish_sharded": (_i32, [_vp, _i32, _i64, _dbl, _dbl, _i64]),
    "nmfx_objective_f64": (_i32, [_vp, _pd]),
    "nmfx_set_stop_guard": (_i32, [_vp, _dbl]),
    "nmfx_resume": (_i32, [_vp]),
    "nmfx_mur_pair_run": (_i32, [_vp, _pd, _pd, _i64, _dbl, _dbl, _i64, _i64]),
    "nmfx_mur_pair_finish": (_i32, [_vp, _i64, _dbl, _dbl, _i64]),
    "nmfx_pair_get_state": (_i32, [_vp, _i32, C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i64)]),
    "nmfx_pair_get_objectives": (_i32, [_vp, _i32, _i64, _i64, _vp]),
    "nmfx_pair_get_factors": (_i32, [_vp, _i32, _i32, _vp, _vp]),
    "nmfx_set_exchange_rank": (_i32, [_vp, _i32, _i32]),
    "nmfx_get_exchange_buffers": (_i32, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "nmfx_aoadmm_run": (_i32, [_vp, _i32, _i32, _dbl, _i32, _dbl, _i32, _i64, _dbl, _dbl, _i64, _i64]),
    "nmfx_aoadmm_finish": (_i32, [_vp, _i64, _dbl, _dbl, _i64]),
    "nmfx_get_inner_counts": (_i32, [_vp, _i64, _i64, _vp]),
    "nmfx_set_l2n_operator": (_i32, [_vp, _i32, _vp]),
    "nmfx_admm_run": (_i32, [_vp, _i32, _dbl, _i32, _dbl, _i32, _dbl, _i64, _dbl, _dbl, _i64, _i64]),
    "nmfx_admm_phase_products": (_i32, [_vp, _i32, _dbl, _i32, _i32, _i64]),
    "nmfx_admm_phase_update": (_i32, [_vp, _i32, _dbl, _i32, _dbl, _i32, _dbl, _i64, _dbl, _dbl, _i64]),
    "nmfx_anls_set_distance": (_i32, [_vp, _i32]),
    "nmfx_anls_run": (_i32, [_vp, _dbl, _dbl, _i64, _dbl, _dbl, _i64, _i64]),
    "nmfx_get_diagnostics": (_i32, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "nmfx_get_nnls_fallbacks": (_i32, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "nmfx_get_inner_paths": (_i32, [_vp, C.POINTER(_i64)]),
    "nmfx_mur_chunk_info": (_i32, [_vp, _i32, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "nmfx_mur_phase_a_head": (_i32, [_vp, _i32, _dbl, _i64]),
    "nmfx_mur_phase_a_cols": (_i32, [_vp, _i32, _i64, _i64]),
    "nmfx_profile_enable": (_i32, [_vp, _i32]),
    "nmfx_profile_get": (_i32, [_vp, C.c_char_p, _pd, C.POINTER(_i64)]),
    "nmfx_profile_reset": (_i32, [_vp]),
    "nmfx_profile_repeat": (_i32, [_vp, C.c_char_p, _i32, _i32, _pd]),
}

_lib = None


class NmfxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libnmfx error {code}: {msg}")
        self.code = code


def load():
    """dlopen libnmfx.so once.  When torch is installed it is imported first so
    that both share ONE HIP runtime (torch bundles its own libamdhip64.so.7;
    loading ours first would map a second copy and break pointer sharing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m nmf_amd.build` "
            "(hipcc, gfx950).  nmf_amd has no CPU fallback.")
    if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None \
            and os.environ.get("NMF_AMD_NO_TORCH") != "1":
        import torch  # noqa: F401  (ordering only)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    lax = os.environ.get("NMFX_LIB_LAX") == "1"      # (A/B runs against an OLDER build through NMFX_LIB: entry points it lacks are skipped)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None) if lax else getattr(lib, name)      # AttributeError = header/library mismatch
        if fn is None:
            continue
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc, handle=None):
    if rc == NMFX_OK:
        return
    msg = load().nmfx_last_error(handle)
    raise NmfxError(rc, msg.decode() if msg else "")


def require_gpu():
    lib = load()
    if lib.nmfx_device_count() <= 0:
        raise RuntimeError("nmf_amd: no HIP device visible; the solvers run only on an "
                           "MI355X (gfx950) and there is no CPU fallback")
    return lib
