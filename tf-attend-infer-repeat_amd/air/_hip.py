"""ctypes binding of libair_hip.so (C ABI declared in include/air_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol
cannot be resolved the import fails loudly.  Build it with
``python tf-attend-infer-repeat_amd/build.py`` (hipcc --offload-arch=gfx950).
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- FIRST: torch brings its own libamdhip64; loading libair_hip.so before it binds the library to a
#                              second HIP runtime that sees no device ("no ROCm-capable device is detected" at the first launch)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AIR_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "libair_hip.so")   # override: A/B builds in tools/

ABI_VERSION = 5

# enums (keep in sync with include/air_hip.h)
DYN_PRIOR_LOG_ODDS, DYN_TEMPERATURE, DYN_STOP_THRESHOLD, DYN_LEARNING_RATE, DYN_CLIP_NORM = 0, 1, 2, 3, 4
DYN_SCALE_PM, DYN_SCALE_PV, DYN_SHIFT_PM, DYN_SHIFT_PV, DYN_VAE_PM, DYN_VAE_PV = 5, 6, 7, 8, 9, 10
DYN_LIK_STD, DYN_GRAD_SCALE, DYN_COUNT = 11, 12, 16
DYN_SCALE_PLV, DYN_SHIFT_PLV, DYN_VAE_PLV = 13, 14, 15
IST_GLOBAL_STEP, IST_COUNT = 0, 4
ATT_S, ATT_X, ATT_Y, ATT_ZPRE, ATT_Z, ATT_ZPROB = 0, 1, 2, 3, 4, 5
ATT_KL_Z, ATT_KL_SCALE, ATT_KL_SHIFT, ATT_KL_VAE, ATT_MASK_PREV, ATT_MASK, ATT_ST_BACK = 6, 7, 8, 9, 10, 11, 12
ATT_STRIDE, OUT_STRIDE = 16, 8
ACT_NONE, ACT_RELU, ACT_SOFTPLUS, ACT_SIGMOID_NOISE = 0, 1, 2, 3
GRAD_NONE, GRAD_RELU, GRAD_SOFTPLUS = 0, 1, 2
EPI_GENERIC, EPI_LSTM_FWD, EPI_REPARAM_FWD, EPI_LSTM_BWD, EPI_REPARAM_BWD, EPI_LSTM_BWD_TAIL, EPI_LSTM_FWD0 = 0, 1, 2, 3, 4, 5, 6
SCHED_STAIRCASE, SCHED_HAS_MIN, SCHED_HAS_MAX, SCHED_LOG = 1, 2, 4, 8

_p = C.c_void_p
_i = C.c_int32
_f = C.c_float


class Schedule(C.Structure):
    _fields_ = [("slot", _i), ("flags", _i), ("init", _f), ("iters", _f), ("factor", _f),
                ("vmin", _f), ("vmax", _f)]


class StepJob(C.Structure):
    _fields_ = [("sched", _p), ("nsched", _i), ("dyn", _p), ("istate", _p),
                ("normals", _p), ("n_normal", C.c_int64), ("uniforms", _p), ("n_uniform", C.c_int64),
                ("seed", C.c_uint64), ("twin_src", _p), ("twin_dst", _p), ("twin_n", C.c_int64)]


class Gemm(C.Structure):
    _fields_ = [("A", _p), ("B", _p), ("C", _p),
                ("M", _i), ("N", _i), ("K", _i), ("lda", _i), ("ldb", _i), ("ldc", _i),
                ("transA", _i), ("transB", _i),
                ("bias", _p), ("addend", _p), ("ldadd", _i), ("aux", _p), ("ldaux", _i),
                ("aux_scale", _f), ("act", _i), ("actgrad", _i), ("accumulate", _i), ("precision", _i),
                ("epi", _i), ("tile_m", _i), ("tile_n", _i), ("ksplit", _i), ("addend_slabs", _i), ("i0", _i),
                ("p0", _p), ("p1", _p), ("p2", _p), ("p3", _p), ("q0", _p), ("q1", _p), ("q2", _p),
                ("step_job", C.POINTER(StepJob)),
                ("A16", _p), ("B16", _p), ("C16", _p), ("q0_16", _p), ("q2_16", _p), ("B16p", _p)]


class Panel(C.Structure):
    """air_panel_t: one row-major [K, N] matrix of the flat variable buffer and where its panel-blocked bf16 twin lives"""
    _fields_ = [("src_off", C.c_int64), ("dst_off", C.c_int64), ("K", _i), ("N", _i), ("gates", _i), ("exclusive", _i)]


MAX_PANELS = 16


class Colsum(C.Structure):
    _fields_ = [("src", _p), ("dst", _p), ("rows", _i), ("cols", _i), ("ld", _i), ("accumulate", _i)]


class Wgrad(C.Structure):
    _fields_ = [("A", _p), ("dY", _p), ("dW", _p), ("db", _p),
                ("M", _i), ("N", _i), ("K", _i), ("lda", _i), ("ldb", _i), ("ldc", _i),
                ("head_pack", _i), ("Hs", _i), ("Hh", _i), ("Hz", _i), ("A16", _p), ("dY16", _p)]


class AttendFwd(C.Structure):
    _fields_ = [("hid", _p), ("wout", _p), ("bout", _p), ("canvas", _p),
                ("eps_scale", _p), ("eps_shift", _p), ("u", _p), ("dyn", _p),
                ("out7", _p), ("att", _p), ("window", _p),
                ("B", _i), ("N", _i), ("C", _i), ("w", _i), ("Hs", _i), ("Hh", _i), ("Hz", _i),
                ("wout_ld", _i), ("train", _i), ("window16", _p)]


class AttendBwd(C.Structure):
    _fields_ = [("hid", _p), ("wout", _p), ("canvas", _p), ("eps_scale", _p), ("eps_shift", _p),
                ("dyn", _p), ("out7", _p), ("att", _p), ("d_window", _p), ("d_sxy_write", _p),
                ("d_hid", _p), ("d_out7", _p),
                ("B", _i), ("N", _i), ("C", _i), ("w", _i), ("Hs", _i), ("Hh", _i), ("Hz", _i), ("wout_ld", _i),
                ("literal", _i), ("d_hid16", _p)]


class WriteFwd(C.Structure):
    _fields_ = [("vrec", _p), ("ml", _p), ("images", _p), ("dyn", _p), ("att", _p), ("recon", _p),
                ("rec_loss", _p), ("d_recon", _p), ("run_loss", _p), ("run_digits", _p), ("loss_item", _p),
                ("B", _i), ("N", _i), ("C", _i), ("w", _i), ("Z", _i), ("wb_order", _p)]


class WriteBwd(C.Structure):
    _fields_ = [("d_recon", _p), ("vrec", _p), ("att", _p), ("d_gen_pre", _p), ("d_sxy_write", _p),
                ("B", _i), ("N", _i), ("C", _i), ("w", _i), ("literal", _i),
                ("fin_loss_item", _p), ("fin_targets", _p), ("fin_digits", _p), ("fin_scalars", _p), ("d_gen_pre16", _p),
                ("order", _p)]


class BottleneckFwd(C.Structure):
    _fields_ = [("X", _p), ("Wml", _p), ("bml", _p), ("eps", _p), ("Wg", _p), ("bg", _p), ("ml", _p), ("z", _p), ("g", _p),
                ("M", _i), ("K1", _i), ("Z", _i), ("H", _i), ("ldx", _i), ("z16", _p), ("g16", _p),
                ("X16", _p), ("Wml16", _p), ("Wg16", _p), ("exact_fp32", _i), ("ldz", _i)]


class BottleneckBwd(C.Structure):
    _fields_ = [("dG", _p), ("Wg", _p), ("ml", _p), ("eps", _p), ("att", _p), ("dyn", _p), ("Wml", _p), ("x", _p),
                ("d_ml", _p), ("d_x", _p), ("M", _i), ("K1", _i), ("Z", _i), ("H", _i), ("d_ml16", _p), ("d_x16", _p),
                ("dG16", _p), ("Wg16", _p), ("Wml16", _p), ("exact_fp32", _i)]


class ShuffleBatch(C.Structure):
    _fields_ = [("queue", _p), ("state", _p), ("picks", _p), ("capacity", _i), ("batch", _i), ("min_after_dequeue", _i),
                ("n_records", _i), ("seed", C.c_uint64)]


class Summaries(C.Structure):
    _fields_ = [("att", _p), ("targets", _p), ("digits", _p), ("rec_loss", _p), ("loss_item", _p), ("scalars", _p),
                ("out", _p), ("B", _i), ("N", _i), ("max_digits", _i)]


_SIGNATURES = {
    "air_abi_version": (C.c_int, []),
    "air_strerror": (C.c_char_p, [C.c_int]),
    "air_gemm": (C.c_int, [C.POINTER(Gemm), _p]),
    "air_gemm_kernel_name": (C.c_int, [C.POINTER(Gemm), C.c_char_p, C.c_int]),
    "air_gemm_slabs": (C.c_int, [C.c_int, C.c_int]),
    "air_bf16_twin": (C.c_int, [_p, _p, C.c_int64, _p]),
    "air_panel_shadow": (C.c_int, [_p, _p, C.POINTER(Panel), C.c_int, _p]),
    "air_adam_clip_step_panels": (C.c_int, [_p, _p, _p, _p, C.c_int64, _p, C.c_int, _p, _p, _f, _f, _f, _f, _p,
                                            C.POINTER(Panel), C.c_int, _p, _p, _p]),
    "air_colsum": (C.c_int, [C.POINTER(Colsum), C.c_int, _p]),
    "air_wgrad_num_blocks": (C.c_int, [C.POINTER(Wgrad), C.c_int]),
    "air_wgrad_num_workgroups": (C.c_int, [C.POINTER(Wgrad), C.c_int, C.c_int]),
    "air_wgrad_grouped": (C.c_int, [C.POINTER(Wgrad), C.c_int, C.c_int, _p, _p, _p]),
    "air_lstm_gates_fwd": (C.c_int, [_p, _p, _p, _p, _p, C.c_int, C.c_int, _p]),
    "air_lstm_first_step": (C.c_int, [_p, C.c_int, _p, _p, _p, _p, _p, C.c_int, C.c_int, _p]),
    "air_lstm_gates_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, C.c_int, C.c_int, C.c_int, _p]),
    "air_transformer_fwd": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "air_transformer_bwd": (C.c_int, [_p, _p, _p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "air_attend_fwd": (C.c_int, [C.POINTER(AttendFwd), _p]),
    "air_attend_bwd": (C.c_int, [C.POINTER(AttendBwd), _p]),
    "air_heads_out_wgrad": (C.c_int, [_p, _p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "air_reparam_fwd": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, _p]),
    "air_reparam_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, C.c_int, C.c_int, _p]),
    "air_write_fwd": (C.c_int, [C.POINTER(WriteFwd), _p]),
    "air_write_bwd": (C.c_int, [C.POINTER(WriteBwd), _p]),
    "air_bce_fwd_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, C.c_int, C.c_int, _p]),
    "air_finalize": (C.c_int, [_p, _p, _p, _p, _p, _p, C.c_int, _p]),
    "air_step_begin": (C.c_int, [_p, C.c_int, _p, _p, _p, C.c_int64, _p, C.c_int64, C.c_uint64, _p, _p, C.c_int64, _p]),
    "air_optim_num_partials": (C.c_int, [C.c_int64]),
    "air_grad_sqnorm": (C.c_int, [_p, C.c_int64, _p, _p, _p]),
    "air_write_bwd_kernel_name": (C.c_int, [C.POINTER(WriteBwd), C.c_char_p, C.c_int]),
    "air_vae_bottleneck_fwd": (C.c_int, [C.POINTER(BottleneckFwd), _p]),
    "air_vae_bottleneck_bwd": (C.c_int, [C.POINTER(BottleneckBwd), _p]),
    "air_adam_clip_step": (C.c_int, [_p, _p, _p, _p, C.c_int64, _p, C.c_int, _p, _p, _f, _f, _f, _f, _p, _p, _p]),
    "air_shuffle_batch_init": (C.c_int, [C.POINTER(ShuffleBatch), _p]),
    "air_shuffle_batch_dequeue": (C.c_int, [C.POINTER(ShuffleBatch), _p]),
    "air_shuffle_batch_dequeue_many": (C.c_int, [C.POINTER(ShuffleBatch), C.c_int, _p, _p]),
    "air_batch_gather": (C.c_int, [_p, _p, _p, _p, _p, C.c_int, C.c_int, _p]),
    "air_summaries_count": (C.c_int, [C.c_int, C.c_int]),
    "air_summaries": (C.c_int, [C.POINTER(Summaries), _p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class AirHipError(RuntimeError):
    pass


def load(path: str = LIB_PATH):
    if not os.path.exists(path):
        raise AirHipError(
            "libair_hip.so not found at %s -- the AIR hot path has no CPU fallback. "
            "Build it: python tf-attend-infer-repeat_amd/build.py" % path)
    lib = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise AirHipError("libair_hip.so lacks symbol %s (stale build?)" % name) from e
        fn.restype = res
        fn.argtypes = args
    if lib.air_abi_version() != ABI_VERSION:
        raise AirHipError("libair_hip.so ABI %d != binding ABI %d" % (lib.air_abi_version(), ABI_VERSION))
    return lib


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = load()
    return _LIB


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().air_strerror(rc)
        raise AirHipError("%s failed with code %d: %s" % (what or "libair_hip call", rc,
                                                          msg.decode() if msg else "?"))
