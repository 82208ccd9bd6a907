"""ctypes binding of libseld_hip.so (include/seld_hip.h).

The HIP library IS the product: there is no CPU or PyTorch fallback.  `load()` raises
`SeldLibraryError` when the shared object is missing or does not export a declared symbol.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# SELD_HIP_LIB: another build of the same library (A/B timing of two builds on one box, tools/); still a HIP build of csrc/
LIB_PATH = os.environ.get("SELD_HIP_LIB") or os.path.join(_HERE, "libseld_hip.so")
CSRC = os.path.join(_HERE, "csrc")

SELD_OK = 0
SELD_DOA_MSE, SELD_DOA_MMSE, SELD_DOA_MAE, SELD_DOA_MSLE = 0, 1, 2, 3
SELD_DTYPE_F32 = 0
SELD_DTYPE_F64 = 1
SELD_DTYPE_BF16 = 2
SELD_ACT = {None: 0, "linear": 0, "sigmoid": 1, "tanh": 2, "relu": 3}      # simple_dense_block's dense_activation -> seld_arch.*_dense_act
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)   # seld_allreduce_fn
MAX_LAYERS = 4
ERR_NAMES = {-1: "SELD_ERR_INVALID", -2: "SELD_ERR_UNSUPPORTED", -3: "SELD_ERR_HIP", -4: "SELD_ERR_NOMEM"}


class SeldLibraryError(RuntimeError):
    pass


class SeldError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class Arch(C.Structure):
    _fields_ = [("in_ch", C.c_int32), ("n_freq", C.c_int32), ("n_conv", C.c_int32),
                ("filters", C.c_int32 * MAX_LAYERS), ("pool_t", C.c_int32 * MAX_LAYERS),
                ("pool_f", C.c_int32 * MAX_LAYERS), ("n_gru", C.c_int32), ("gru_units", C.c_int32 * MAX_LAYERS),
                ("n_sed_dense", C.c_int32), ("sed_units", C.c_int32 * MAX_LAYERS),
                ("n_doa_dense", C.c_int32), ("doa_units", C.c_int32 * MAX_LAYERS), ("n_classes", C.c_int32),
                ("first_kind", C.c_int32), ("xc_blocks", C.c_int32), ("rn_filters", C.c_int32), ("rn_blocks", C.c_int32 * 4),
                ("sed_dense_act", C.c_int32), ("doa_dense_act", C.c_int32),
                ("sed_kernel_size", C.c_int32), ("doa_kernel_size", C.c_int32), ("sed_dropout", C.c_float), ("doa_dropout", C.c_float),
                ("output_coupling", C.c_int32),
                ("conv_dropout", C.c_float), ("gru_dropout", C.c_float)]


class LossCfg(C.Structure):
    _fields_ = [("doa_loss", C.c_int32), ("w_sed", C.c_float), ("w_doa", C.c_float),
                ("sed_grad_scale", C.c_float), ("mmse_den", C.c_float)]


_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_F = C.c_float
_FP = C.POINTER(C.c_float)

# name -> (restype, argtypes); must list every symbol include/seld_hip.h declares
SIGNATURES = {
    "seld_abi_sizes": (_I, [C.POINTER(C.c_int32), _I]),
    "seld_create": (_I, [C.POINTER(Arch), _I, _I, _I, _I, C.POINTER(_P)]),
    "seld_destroy": (None, [_P]),
    "seld_last_error": (C.c_char_p, [_P]),
    "seld_set_stream": (_I, [_P, _P]),
    "seld_set_batch": (_I, [_P, _I]),
    "seld_sync": (_I, [_P]),
    "seld_set_option": (_I, [_P, C.c_char_p, _I]),
    "seld_k_set_option": (_I, [C.c_char_p, _I]),
    "seld_param_count": (_L, [_P]),
    "seld_state_count": (_L, [_P]),
    "seld_variable_count": (_I, [_P, _I]),
    "seld_variable_info": (_I, [_P, _I, _I, C.c_char_p, _I, C.POINTER(_L), C.POINTER(C.c_int32), C.POINTER(_L)]),
    "seld_set_weights_host": (_I, [_P, _P, _L]),
    "seld_get_weights_host": (_I, [_P, _P, _L]),
    "seld_set_state_host": (_I, [_P, _P, _L]),
    "seld_get_state_host": (_I, [_P, _P, _L]),
    "seld_get_grads_host": (_I, [_P, _P, _L]),
    "seld_get_adam_host": (_I, [_P, _P, _P, _L]),
    "seld_set_adam_host": (_I, [_P, _P, _P, _L, _L]),
    "seld_param_ptr": (_P, [_P]),
    "seld_grad_ptr": (_P, [_P]),
    "seld_forward": (_I, [_P, _P, _P, _P, _I]),
    "seld_train_fwd_bwd": (_I, [_P, _P, _P, _P, C.POINTER(LossCfg), _P, _P, _P, _P]),
    "seld_grads_tail_ready": (_I, [_P, _P, C.POINTER(_L)]),
    "seld_grads_bucket_count": (_I, [_P]),
    "seld_grads_bucket_ready": (_I, [_P, _I, _P, C.POINTER(_L), C.POINTER(_L)]),
    "seld_set_sync_bn": (_I, [_P, _P, _P, _I]),
    "seld_dp_available": (_I, []),
    "seld_dp_unique_id": (_I, [_P]),
    "seld_dp_init": (_I, [_P, _I, _I, _P]),
    "seld_dp_world": (_I, [_P]),
    "seld_dp_allreduce_grads": (_I, [_P]),
    "seld_dp_set_sync_bn": (_I, [_P, _I]),
    "seld_dp_destroy": (_I, [_P]),
    "seld_adam_step": (_I, [_P, _F, _F, _F, _F, _I]),
    "seld_train_step": (_I, [_P, _P, _P, _P, C.POINTER(LossCfg), _F, _I, _P, _P, _P, _P]),
    "seld_test_step": (_I, [_P, _P, _P, _P, C.POINTER(LossCfg), _P, _P, _P, _P]),
    "seld_mmse_den": (_I, [_P, _P, _P]),
    "seld_feat_create": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, C.POINTER(_P)]),
    "seld_feat_destroy": (None, [_P]),
    "seld_feat_last_error": (C.c_char_p, [_P]),
    "seld_feat_set_option": (_I, [_P, C.c_char_p, _I]),
    "seld_feat_frames": (_L, [_P, _L]),
    "seld_feat_channels": (_I, [_P]),
    "seld_feat_extract": (_I, [_P, _P, _I, _L, _P, _P]),
    "seld_feat_extract_batch": (_I, [_P, _P, _I, _I, _L, _P, _P]),
    "seld_feat_normalize": (_I, [_P, _P, _P, _P, _L, _L, _I, _F, _P]),
    "seld_feat_stats_scratch_doubles": (_L, [_I]),
    "seld_feat_stats_accumulate": (_I, [_P, _L, _I, _P, _P, _P]),
    "seld_feat_stats_finalize": (_I, [_P, _I, _P, _P, _P]),
    "seld_frame_windows": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "seld_overlap_average": (_I, [_P, _P, _I, _I, _I, _P]),
    "seld_aug_mask": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "seld_aug_gather_sign": (_I, [_P, _I, _L, _I, _L, _P, _P, _P]),
    "seld_metrics_state_size": (_I, [_I]),
    "seld_metrics_scratch_floats": (_L, [_I, _I, _I, _I]),
    "seld_metrics_update": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P]),
    "seld_profile_enable": (_I, [_P, _I]),
    "seld_profile_count": (_I, [_P]),
    "seld_profile_get": (_I, [_P, _I, C.c_char_p, _I, C.POINTER(_L), C.POINTER(C.c_double)]),
    "seld_profile_reset": (_I, [_P]),
    "seld_k_conv3x3_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I]),
    "seld_k_conv_first_fwd_pool": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I]),
    "seld_k_conv1_gram": (_I, [_P, _P, _I, _I, _I]),
    "seld_k_conv1_train_gram": (_I, [_P] * 11 + [_I] * 3),
    "seld_k_bn_relu_ext": (_I, [_P, _P, _P, _P, _L]),
    "seld_k_conv3x3_dgrad": (_I, [_P, _P, _P, _I, _I, _I, _I, _I]),
    "seld_k_conv3x3_wgrad": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I]),
    "seld_k_conv1_bwd_fused": (_I, [_P] * 11 + [_I] * 5),
    "seld_k_bn_relu_pool_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I]),
    "seld_k_bn_relu_pool_bwd": (_I, [_P] * 9 + [_I] * 6),
    "seld_k_gemm": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I]),
    "seld_k_gemm_pair_n": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I]),
    "seld_k_gemm_pair_k": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I]),
    "seld_k_gemm_sb": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I]),
    "seld_k_xc_dw_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I]),
    "seld_k_gemm_tn": (_I, [_P, _P, _P, _P, _I, _I, _I]),
    "seld_k_gru_fwd": (_I, [_P] * 11 + [_I] * 3),
    "seld_k_gru_bwd": (_I, [_P] * 11 + [_I] * 3),
    "seld_k_losses": (_I, [_P, _P, _P, _P, C.POINTER(LossCfg), _P, _P, _P, _P, _I, _I, _I]),
    "seld_k_adam": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _L]),
    "seld_debug_pool_routing": (_I, [_P, _I, _P, _P]),
    "seld_debug_relu_output": (_I, [_P, _I, _I, _P, _L, C.POINTER(C.c_int64)]),
    "seld_debug_set_routing": (_I, [_P, _I, _L, _P, _P]),
    "seld_debug_set_relu_gates": (_I, [_P, _I, _I, _L, _P, _P]),
    "seld_k_gru_timing": (_I, [_I, _P, _I]),
    "seld_m_conv_out": (_I, [_I, _I]),
    "seld_m_im2col": (_I, [_P, _P] + [_I] * 8 + [_P]),
    "seld_m_col2im": (_I, [_P, _P] + [_I] * 9 + [_P]),
    "seld_m_gemm": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "seld_m_gemm_tn_scratch": (_L, [_I, _I]),
    "seld_m_gemm_tn": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "seld_m_bn_scratch": (_L, [_I]),
    "seld_m_bn_stats": (_I, [_P, _L, _I, _P, _P, _P, _P]),
    "seld_m_bn_apply": (_I, [_P, _P, _P, _P, _P, _F, _P, _L, _I, _I, _P]),
    "seld_m_bn_moving": (_I, [_P, _P, _P, _P, _I, _F, _L, _P]),
    "seld_m_bn_bwd": (_I, [_P, _P, _P, _P, _P, _F, _P, _P, _P, _L, _I, _P, _P]),
    "seld_m_act": (_I, [_P, _P, _L, _I, _P]),
    "seld_m_act_bwd": (_I, [_P, _P, _P, _L, _I, _I, _P]),
    "seld_m_axpy": (_I, [_P, _P, _L, _F, _P]),
    "seld_m_copy_channels": (_I, [_P, _P, _L, _I, _I, _I, _I, _P]),
    "seld_m_mean_hw": (_I, [_P, _P, _I, _I, _I, _P]),
    "seld_m_scale_hw": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "seld_m_scale_hw_bwd_ds": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "seld_m_scale_hw_bwd_dx": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "seld_m_gru_fwd": (_I, [_P] * 11 + [_I] * 3 + [_P]),
    "seld_m_gru_bwd": (_I, [_P] * 11 + [_I] * 3 + [_P]),
    "seld_m_losses_scratch": (_L, [_I]),
    "seld_m_losses": (_I, [_P, _P, _P, _P, C.POINTER(LossCfg), _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "seld_m_adam": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _L, _P]),
    "seld_k_rn_conv": (_I, [_P, _P, _P] + [_I] * 7),
    "seld_k_rn_conv_bwd": (_I, [_P, _P, _P, _P, _P] + [_I] * 7),
    "seld_k_rn_bn": (_I, [_P] * 7 + [_L, _I, _I]),
    "seld_k_rn_bn_bwd": (_I, [_P] * 7 + [_L, _I]),
    "seld_device_clocks": (_I, [_I] + [C.POINTER(C.c_int)] * 4),
    "seld_k_valu_clock_mhz": (_I, [_I, C.POINTER(C.c_double)]),
}

_lib = None


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 into seld_amd/libseld_hip.so (hipcc cross-compiles
    without a GPU).  Returns the library path."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=True)
    r = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if r.returncode != 0:
        raise SeldLibraryError("building libseld_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return LIB_PATH


def load():
    """Load libseld_hip.so and bind every declared symbol.  Fails loudly; never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SeldLibraryError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C {CSRC}`.  seld_amd has no CPU/PyTorch fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise SeldLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise SeldLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    sizes = (C.c_int32 * 2)()
    if lib.seld_abi_sizes(sizes, 2) != 2 or sizes[0] != C.sizeof(Arch) or sizes[1] != C.sizeof(LossCfg):
        raise SeldLibraryError(f"{LIB_PATH}: struct layout mismatch (library seld_arch {sizes[0]} B, seld_loss_cfg {sizes[1]} B; "
                               f"binding {C.sizeof(Arch)} / {C.sizeof(LossCfg)} B): rebuild the library or update seld_amd/_lib.py")
    _lib = lib
    return lib


def check(rc: int, ctx=None):
    if rc != SELD_OK:
        msg = load().seld_last_error(ctx)
        raise SeldError(rc, msg.decode() if msg else "")
