"""ctypes binding of libdua_hip.so (include/dua_hip.h).

There is no CPU fallback: if the library is missing or a GPU entry point is
called without a device, this module raises.  torch is imported first so that
the library binds to the libamdhip64 already mapped by PyTorch-ROCm (same
SONAME) and both share one HIP runtime, device context and stream objects.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL: maps libamdhip64.so.7)

_HERE = os.path.dirname(os.path.abspath(__file__))
# The shipped library; with DUA_DEBUG=1 in the environment, DUA_HIP_LIB may name a diagnostic build of the same library
# (tools/build_diag.sh: stamps, ablations) instead.  Without the debug flag the environment cannot redirect the package.
LIB_PATH = os.path.join(_HERE, "libdua_hip.so")
if os.environ.get("DUA_DEBUG") == "1" and os.environ.get("DUA_HIP_LIB"):
    LIB_PATH = os.environ["DUA_HIP_LIB"]

F32, F16 = 0, 1
ERR_ARG = -22


class NativeLibraryMissing(RuntimeError):
    pass


class Conv3Desc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("N", C.c_int), ("D", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("Cin", C.c_int), ("Cin_stride", C.c_int), ("Cin_off", C.c_int),
                ("Cout", C.c_int), ("Cout_stride", C.c_int), ("Cout_off", C.c_int), ("tap_channel_plus1", C.c_int),
                ("background", C.c_int), ("layout", C.c_int), ("policy", C.c_int)]


class InNorm(C.Structure):
    _fields_ = [("stats", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("add", C.c_void_p),
                ("add_stride", C.c_int), ("c_pad", C.c_int), ("count", C.c_longlong), ("eps", C.c_float),
                ("slope", C.c_float)]


class UpConvDesc(C.Structure):
    """dua_upconv_desc."""
    _fields_ = [(n, C.c_int) for n in ("dtype", "N", "D", "H", "W", "Cskip", "Cskip_stride", "Cskip_off", "Cu", "Cu_stride", "Cu_off",
                                       "Cout", "Cout_stride", "Cout_off", "layout")]


class MaterializeDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("N", C.c_int), ("D", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int),
                ("raw_stride", C.c_int), ("emb_stride", C.c_int), ("out_stride", C.c_int), ("out_off", C.c_int),
                ("pool_stride", C.c_int), ("out_blocked", C.c_int)]


class NormBwdDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("N", C.c_int), ("voxels", C.c_long), ("C", C.c_int), ("da_stride", C.c_int),
                ("da_off", C.c_int), ("raw_stride", C.c_int), ("raw_off", C.c_int), ("out_stride", C.c_int),
                ("out_off", C.c_int)]


class TailDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("N", C.c_int), ("voxels", C.c_long), ("K", C.c_int), ("raw_stride", C.c_int),
                ("C", C.c_int), ("CX", C.c_int), ("mode", C.c_int), ("xin_stride", C.c_int),
                ("seed", C.c_ulonglong), ("seed_dev", C.c_void_p)]


class TailResidual(C.Structure):
    _fields_ = [("res", C.c_void_p), ("res_stride", C.c_int), ("res_norm", InNorm), ("ra_src", C.c_void_p),
                ("ra_stride", C.c_int), ("ra_off", C.c_int), ("channels", C.c_int)]


IN_BLOCKED, OUT_BLOCKED = 1, 2            # dua_conv3_desc.layout bits
POLICY_NO_FINISH = 256                    # dua_conv3_desc.policy bit: skip the split-K finish kernel (timing only)
MODE_LOGITS, MODE_DDPM, MODE_DDIM = 0, 1, 2
OP_CONV3, OP_MATERIALIZE, OP_DECONV, OP_UPCONV = 1, 2, 3, 4


class StepOp(C.Structure):
    _fields_ = [("kind", C.c_int), ("has_norm", C.c_int), ("conv", Conv3Desc), ("mat", MaterializeDesc), ("norm", InNorm),
                ("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p), ("stats", C.c_void_p),
                ("emb", C.c_void_p), ("pooled", C.c_void_p), ("up", UpConvDesc), ("u", C.c_void_p), ("wu", C.c_void_p)]


class DenoiserPlan(C.Structure):
    _fields_ = [("N", C.c_int), ("P", C.c_int),
                ("temb_table", C.c_void_p), ("table_rows", C.c_int),
                ("rows_per_sample", C.c_void_p),
                ("row_of_step", C.c_void_p), ("nsteps", C.c_int), ("coef_table", C.c_void_p), ("counter", C.c_void_p),
                ("cur_add", C.c_void_p), ("cur_coef", C.c_void_p), ("step_word", C.c_void_p), ("err_word", C.c_void_p),
                ("stat_arena", C.c_void_p), ("stat_bytes", C.c_long),
                ("ops", C.POINTER(StepOp)), ("n_ops", C.c_int),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_long),
                ("tail", TailDesc), ("tail_raw", C.c_void_p), ("tail_norm", InNorm), ("wf", C.c_void_p), ("bf", C.c_void_p),
                ("x_state", C.c_void_p), ("noise", C.c_void_p), ("xin", C.c_void_p), ("xstart_sum", C.c_void_p),
                ("logits", C.c_void_p), ("xstart", C.c_void_p)]


class WindowGeom(C.Structure):
    """dua_window_geom."""
    _fields_ = [(n, C.c_int) for n in ("B", "D", "H", "W", "C", "wd", "wh", "ww", "sd", "sh", "sw")]


class TokenLinearDesc(C.Structure):
    """dua_token_linear_desc."""
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int), ("M", C.c_long), ("K", C.c_int), ("N", C.c_int),
                ("W", C.c_void_p), ("bias", C.c_void_p), ("mode", C.c_int), ("samples", C.c_int),
                ("out", C.c_void_p), ("ldc", C.c_int), ("out_off", C.c_int), ("x", C.c_void_p),
                ("stats", C.c_void_p), ("c_pad", C.c_int), ("geom", WindowGeom), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("eps", C.c_float), ("ln_out", C.c_void_p), ("background", C.c_int)]


TOKLIN_PLAIN, TOKLIN_GELU, TOKLIN_STATS, TOKLIN_RESIDUAL, TOKLIN_SCATTER = range(5)

TEMB_MAX_BLOCKS, ADAMW_MAX_TENSORS = 16, 64


class TembBlocks(C.Structure):
    """dua_temb_blocks."""
    _fields_ = [("nblocks", C.c_int), ("cout", C.c_int * TEMB_MAX_BLOCKS), ("w", C.c_void_p * TEMB_MAX_BLOCKS),
                ("b", C.c_void_p * TEMB_MAX_BLOCKS), ("dw", C.c_void_p * TEMB_MAX_BLOCKS), ("db", C.c_void_p * TEMB_MAX_BLOCKS)]


class PackItem(C.Structure):
    """dua_pack_item."""
    _fields_ = [("kind", C.c_int), ("Cout", C.c_int), ("Cin", C.c_int), ("packed", C.c_int), ("w", C.c_void_p), ("out", C.c_void_p)]


class AdamWList(C.Structure):
    """dua_adamw_list."""
    _fields_ = [("count", C.c_int), ("numel", C.c_long * ADAMW_MAX_TENSORS), ("p", C.c_void_p * ADAMW_MAX_TENSORS),
                ("g", C.c_void_p * ADAMW_MAX_TENSORS), ("m", C.c_void_p * ADAMW_MAX_TENSORS), ("v", C.c_void_p * ADAMW_MAX_TENSORS)]

_P = C.c_void_p
ABI_VERSION = 8          # DUA_ABI_VERSION of include/dua_hip.h this binding was written against

_SIGS = {
    "dua_abi_version": (C.c_int, []),
    "dua_prepare": (C.c_int, []),
    "dua_prepared_kernels": (C.c_int, []),
    "dua_mfma_probe": (C.c_int, [C.c_int, C.c_int, _P, _P, _P]),
    "dua_chain_probe": (C.c_int, [C.c_int, C.c_int, _P, _P, _P, _P]),
    "dua_deconv_k2s2_fwd": (C.c_int, [C.POINTER(Conv3Desc), _P, _P, _P, C.POINTER(InNorm), _P, _P]),
    "dua_conv3d_k3_dgrad_reduce_supported": (C.c_int, [C.POINTER(Conv3Desc)]),
    "dua_conv3d_k3_dgrad_reduce": (C.c_int, [C.POINTER(Conv3Desc), _P, _P, _P, _P, _P, C.c_int, C.c_int, C.POINTER(InNorm), _P, _P]),
    "dua_deconv_k2s2_res_supported": (C.c_int, [C.POINTER(Conv3Desc), C.c_int]),
    "dua_deconv_k2s2_res_fwd": (C.c_int, [C.POINTER(Conv3Desc), _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "dua_q_sample": (C.c_int, [C.c_int, C.c_long, _P, _P, _P, _P, _P]),
    "dua_sampler_step": (C.c_int, [C.c_int, C.c_int, C.c_long, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dua_final_conv_sampler": (C.c_int, [C.POINTER(TailDesc), _P, C.POINTER(InNorm)] + [_P] * 11),
    "dua_final_conv_sampler_res": (C.c_int, [C.POINTER(TailDesc), _P, C.POINTER(InNorm), C.POINTER(TailResidual)] + [_P] * 11),
    "dua_window_attention_fwd": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int,
                                           C.c_float, _P, _P]),
    "dua_patch_merge_norm": (C.c_int, [C.c_int] * 7 + [_P, _P, _P, _P, C.c_float, _P, _P]),
    "dua_residual_norm_act": (C.c_int, [C.c_int, C.c_int, C.c_long, C.c_int, _P, C.c_int, C.POINTER(InNorm), _P, C.c_int,
                                        C.POINTER(InNorm), _P, C.c_int, C.c_int, C.c_float, _P, C.c_int, C.c_int, _P, C.c_int,
                                        C.c_int, C.c_int, _P]),
    "dua_window_gather_norm": (C.c_int, [C.c_int, C.POINTER(WindowGeom), _P, _P, _P, _P, C.c_float, _P, _P]),
    "dua_window_scatter_add_norm": (C.c_int, [C.c_int, C.POINTER(WindowGeom), _P, _P, _P, _P, C.c_float, _P, _P]),
    "dua_stage_out": (C.c_int, [C.c_int, C.c_int, C.c_long, C.c_int, _P, _P, C.c_int, C.c_float, _P, _P, _P, C.c_int, C.c_int, _P]),
    "dua_patch_embed": (C.c_int, [C.c_int] * 8 + [_P, _P, _P, _P, C.c_int, C.c_float, _P, _P, _P, C.c_int, C.c_int, _P]),
    "dua_instnorm_stats": (C.c_int, [C.c_int, C.c_int, C.c_long, C.c_int, _P, C.c_int, C.c_int, _P, C.c_int, _P]),
    "dua_gelu": (C.c_int, [C.c_int, C.c_long, _P, _P]),
    "dua_token_linear": (C.c_int, [C.POINTER(TokenLinearDesc), _P]),
    "dua_token_gemm": (C.c_int, [C.POINTER(TokenLinearDesc), _P, C.c_long, _P]),
    "dua_token_gemm_workspace": (C.c_long, [C.c_long, C.c_int, C.c_int]),
    "dua_swin_mlp": (C.c_int, [C.c_long, C.c_int, _P, _P, _P, _P, _P, _P, _P]),
    "dua_denoiser_step": (C.c_int, [C.POINTER(DenoiserPlan), _P]),
    "dua_temb_table": (C.c_int, [C.c_int, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, _P, _P]),
    "dua_step_begin": (C.c_int, [C.c_int, C.c_int, _P, C.c_int, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P, _P]),
    "dua_step_begin_clear": (C.c_int, [C.c_int, C.c_int, _P, C.c_int, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P, _P, C.c_long, _P]),
    "dua_conv3d_k3_workspace": (C.c_long, [C.POINTER(Conv3Desc)]),
    "dua_conv3d_k3_kernel_kind": (C.c_int, [C.POINTER(Conv3Desc), C.c_int, C.c_int]),
    "dua_deconv_k2s2_kernel_kind": (C.c_int, [C.POINTER(Conv3Desc)]),
    "dua_conv3d_k3_fwd": (C.c_int, [C.POINTER(Conv3Desc), _P, _P, _P, C.POINTER(InNorm), _P, _P, _P, C.c_long, _P]),
    "dua_upconv_k3_supported": (C.c_int, [C.POINTER(UpConvDesc)]),
    "dua_pack_upconv_weights": (C.c_long, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P]),
    "dua_upconv_k3_fwd": (C.c_int, [C.POINTER(UpConvDesc), _P, _P, C.POINTER(InNorm), _P, _P, _P, _P, _P, _P]),
    "dua_conv3d_k3_wgrad_workspace": (C.c_long, [C.POINTER(Conv3Desc)]),
    "dua_conv3d_k3_wgrad": (C.c_int, [C.POINTER(Conv3Desc), _P, _P, _P, C.c_int, _P, _P, C.c_long, _P]),
    "dua_pack_conv3_weights": (C.c_long, [C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "dua_pack_conv3_weights_tap": (C.c_long, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "dua_pack_conv3_weights_dgrad": (C.c_long, [C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "dua_instnorm_finalize": (C.c_int, [C.c_int, C.c_int, C.POINTER(InNorm), _P, _P, _P]),
    "dua_instnorm_bwd_reduce": (C.c_int, [C.POINTER(NormBwdDesc), _P, _P, C.POINTER(InNorm), _P, _P]),
    "dua_instnorm_bwd_apply": (C.c_int, [C.POINTER(NormBwdDesc), _P, _P, C.POINTER(InNorm), _P, _P, _P, _P, _P, _P]),
    "dua_maxpool2_bwd_add": (C.c_int, [C.c_int] * 6 + [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, C.c_int, _P, C.c_int, _P]),
    "dua_pack_deconv_weights_dgrad": (C.c_long, [C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "dua_deconv_k2s2_bwd_workspace": (C.c_long, [C.POINTER(Conv3Desc)]),
    "dua_deconv_k2s2_bwd": (C.c_int, [C.POINTER(Conv3Desc), _P, _P, _P, _P, _P, _P, C.c_long, _P]),
    "dua_head_fwd": (C.c_int, [C.c_int, C.c_long, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P, C.c_int, _P]),
    "dua_head_bwd_workspace": (C.c_long, [C.c_long]),
    "dua_head_bwd": (C.c_int, [C.c_int, C.c_long, C.c_int, C.c_int, _P, C.c_int, _P, C.c_int, _P, _P, C.c_int, _P, _P, _P, C.c_long, _P]),
    "dua_seg_loss_reduce": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_long, _P, C.c_int, _P, _P, _P]),
    "dua_seg_loss_grad": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_long, _P, C.c_int, _P, _P, _P, C.c_float, C.c_float,
                                    C.c_float, _P, C.c_int, _P]),
    "dua_materialize": (C.c_int, [C.POINTER(MaterializeDesc), _P, C.POINTER(InNorm), _P, _P, _P, _P]),
    "dua_pack_deconv_weights": (C.c_long, [C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "dua_to_channels_last": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_long, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "dua_from_channels_last": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_long, _P, C.c_int, C.c_int, _P, _P]),
    "dua_pack_conv3_weights_batch": (C.c_int, [C.c_int, C.c_int, C.POINTER(PackItem), _P]),
    "dua_to_channels_last_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, C.c_long, _P, C.c_int, _P]),
    "dua_linear_f32": (C.c_int, [C.c_long, C.c_int, C.c_int, _P, C.c_long, _P, _P, _P, C.c_long, C.c_int, _P]),
    "dua_stats_channel_sums": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "dua_seg_loss_finish": (C.c_int, [C.c_int, C.c_int, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "dua_q_sample_affine": (C.c_int, [C.c_int, C.c_long, _P, C.c_float, C.c_float, _P, _P, C.c_int, _P, _P, _P]),
    "dua_temb_train_fwd": (C.c_int, [C.c_int, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P, C.POINTER(TembBlocks), _P, _P, _P]),
    "dua_temb_train_bwd": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, C.POINTER(TembBlocks), _P, _P, _P, _P, _P, _P, _P, _P]),
    "dua_grads_nonfinite": (C.c_int, [C.POINTER(AdamWList), _P, _P]),
    "dua_adamw_step": (C.c_int, [C.POINTER(AdamWList), C.c_float, _P, C.c_float, C.c_float, C.c_float, C.c_float, _P, _P, _P,
                                 C.c_int, _P]),
    "dua_adamw_advance": (C.c_int, [_P, _P, _P, _P, C.c_float, C.c_float, C.c_int, _P, _P]),
}

_lib = None


def exported_symbols():
    """Every entry point include/dua_hip.h declares (kept in step by tests/test_abi.py)."""
    return sorted(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        got = L.dua_abi_version()
        if got != ABI_VERSION:
            raise NativeLibraryMissing(f"{LIB_PATH} has ABI version {got}, this binding needs {ABI_VERSION}: rebuild the library")
        _lib = L
    return _lib


def prepare(device=None):
    """dua_prepare() on ``device`` (default: the current one): every function attribute the library's launchers need is
    set before the first launch -- plans and trainers call this at construction, i.e. before any stream capture and
    before autograd's worker thread issues a launch."""
    L = lib()
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        check(L.dua_prepare(), "dua_prepare")


def ptr(t):
    """Device pointer of a tensor (or None)."""
    if t is None:
        return None
    assert t.is_cuda, "libdua_hip.so entry points take device pointers"
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {'invalid argument' if rc == ERR_ARG else f'hipError {rc}'}")


def dt_code(dtype: torch.dtype) -> int:
    if dtype == torch.float16:
        return F16
    if dtype == torch.float32:
        return F32
    raise ValueError(f"unsupported activation dtype {dtype}")
