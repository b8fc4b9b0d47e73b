"""ctypes binding of libtsff.so (the C ABI of include/tsff.h).

There is no CPU fallback: if the shared library is missing, does not load, or the process has no
HIP device, the product path raises.  The oracle under ``oracle/`` is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtsff.so")
LIB_PATH = os.environ.get("TSFF_LIBRARY", LIB_PATH)  # A/B experiments: another in-tree build of the same ABI

ABI_VERSION = 8
MAX_ION = 4
NBINS = 1024
NXI1 = 1024
NXI2 = 1640
DLM_NM = 31

# parameter slots (include/tsff.h)
P_TE, P_NE, P_M, P_LAM, P_AMP1, P_AMP2, P_AMP3, P_NE_GRADIENT, P_TE_GRADIENT, P_UD, P_VA, P_ION0 = range(12)
ION_TI, ION_Z, ION_A, ION_FRACT = range(4)
FE_SHARED, FE_PER_LINEOUT, FE_DLM = range(3)
LOSS_METHODS = {"l2": 0, "l1": 1, "log-cosh": 2, "poisson": 3}
FEATURE_ELE, FEATURE_ION = 0, 1
OPT_DENOM_MODE = 1
OPT_LAUNCH_PLAN = 2
OPT_DLM_BLOCKS = 3
ERR_LDS = -7  # TSFF_ERR_LDS: more LDS needed than a CU has


def n_params(n_ion: int) -> int:
    return P_ION0 + 4 * n_ion


c_double_p = C.POINTER(C.c_double)
c_uint8_p = C.POINTER(C.c_uint8)


class TsffConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("lamrangE", C.c_double * 2),
        ("lamrangI", C.c_double * 2),
        ("npts", C.c_int32),
        ("load_ele", C.c_int32),
        ("load_ion", C.c_int32),
        ("ele_lam_shift", C.c_double),
        ("n_angles", C.c_int32),
        ("sa_deg", c_double_p),
        ("sa_weights", c_double_p),
        ("num_grad_points", C.c_int32),
        ("n_ion", C.c_int32),
        ("nvx", C.c_int32),
        ("fe_mode", C.c_int32),
        ("fe_shared", c_double_p),
        ("dlm_table", c_double_p),
        ("xi1", c_double_p),
        ("xi2", c_double_p),
        ("zprime_re", c_double_p),
        ("zprime_im", c_double_p),
        ("lg_table", c_double_p),
        ("n_taps_ele", C.c_int32),
        ("tap_off_ele", C.c_int32),
        ("taps_ele", c_double_p),
        ("n_taps_ion", C.c_int32),
        ("tap_off_ion", C.c_int32),
        ("taps_ion", c_double_p),
        ("norm", C.c_int32),
        ("ele_filter", c_double_p),
        ("p_scale", c_double_p),
        ("p_shift", c_double_p),
        ("p_sigmoid", c_uint8_p),
        ("ti_same", C.c_uint8 * MAX_ION),
        ("loss_method", C.c_int32),
        ("mask_ele", c_uint8_p),
        ("mask_ion", c_uint8_p),
    ]


class TsffAtsConfig(C.Structure):
    _fields_ = [
        ("n_px", C.c_int32),
        ("weights", c_double_p),
        ("n_taps_ang", C.c_int32),
        ("tap_off_ang", C.c_int32),
        ("taps_ang", c_double_p),
        ("n_taps_lam", C.c_int32),
        ("tap_off_lam", C.c_int32),
        ("taps_lam", c_double_p),
        ("lam_step", C.c_int32),
        ("ang_step", C.c_int32),
        ("row_start", C.c_int32),
        ("row_end", C.c_int32),
        ("lam_axis", c_double_p),
    ]


_vp = C.c_void_p
_SIGNATURES = {
    "tsff_abi_version": (C.c_int, []),
    "tsff_create": (C.c_int, [C.POINTER(TsffConfig), C.POINTER(_vp)]),
    "tsff_destroy": (None, [_vp]),
    "tsff_last_error": (C.c_char_p, [_vp]),
    "tsff_set_stream": (C.c_int, [_vp, _vp]),
    "tsff_set_option": (C.c_int, [_vp, C.c_int32, C.c_int32]),
    "tsff_reserve": (C.c_int, [_vp, C.c_int32]),
    "tsff_get_axes": (C.c_int, [_vp, c_double_p, c_double_p]),
    "tsff_chi_table": (C.c_int, [_vp, _vp, C.c_int32, _vp]),
    "tsff_form_factor": (C.c_int, [_vp, C.c_int32, _vp, _vp, C.c_int32, _vp]),
    "tsff_form_factor_grad": (C.c_int, [_vp, C.c_int32, _vp, _vp, C.c_int32, _vp, _vp, _vp]),
    "tsff_form_factor_2d": (C.c_int, [_vp, C.c_int32, _vp, _vp, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_int32, _vp]),
    "tsff_form_factor_2d_range": (C.c_int, [_vp, C.c_int32, _vp, _vp, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_int32,
                                            C.c_int64, C.c_int64, _vp]),
    "tsff_form_factor_2d_grad": (C.c_int, [_vp, C.c_int32, _vp, _vp, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_int64, C.c_int64, C.c_uint64, _vp, _vp, _vp]),
    "tsff_form_factor_2d_save": (C.c_int, [_vp, C.c_int32, _vp, _vp, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_int64, C.c_int64, _vp,
                                           C.POINTER(C.c_uint64)]),
    "tsff_ats_setup": (C.c_int, [_vp, C.POINTER(TsffAtsConfig)]),
    "tsff_ats_spectrum": (C.c_int, [_vp, _vp, _vp, C.c_double, C.c_double, C.c_double, _vp]),
    "tsff_ats_adjoint": (C.c_int, [_vp, _vp, _vp, C.c_double, C.c_double, C.c_double, _vp, _vp, c_double_p]),
    "tsff_forward": (C.c_int, [_vp] + [_vp] * 6 + [C.c_int32, _vp, _vp]),
    "tsff_loss_grad": (C.c_int, [_vp] + [_vp] * 8 + [C.c_int32, c_double_p, c_uint8_p, _vp, _vp, _vp, _vp]),
    "tsff_loss_grad_packed": (C.c_int, [_vp] + [_vp] * 8 + [C.c_int32, c_double_p, c_uint8_p, C.POINTER(C.c_int32), C.c_int32, C.c_int64, C.c_int64,
                                        _vp, _vp, _vp]),
    "tsff_loss_grad_fe": (C.c_int, [_vp] + [_vp] * 8 + [C.c_int32, c_double_p, c_uint8_p, _vp, _vp, _vp, _vp, _vp]),
    "tsff_pack_fe_rows": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int64, C.c_int64, _vp]),
    "tsff_array_loss": (C.c_int, [_vp] + [_vp] * 8 + [C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "tsff_enable_timing": (C.c_int, [_vp, C.c_int32]),
    "tsff_kernel_times": (C.c_int, [_vp, C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_int32)]),
    "tsff_fp64_fma_peak": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "tsff_fp64_mfma_peak": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "tsff_l1_read_peak": (C.c_int, [_vp, C.POINTER(C.c_double)]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


class TsffError(RuntimeError):
    pass


def load():
    """Load libtsff.so (built in-tree by ``__graft_entry__.build()`` / ``tsadar_amd.build``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TsffError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  tsadar_amd has no CPU fallback."
        )
    # torch first: libtsff.so needs libamdhip64, and the HIP runtime of the process must be the one torch brings (device memory, streams
    # and events are shared with it).  Loaded before torch, the library binds /opt/rocm's copy, torch then loads its own, and
    # tsff_create finds "no HIP device" (seen with build() and smoke() in one process).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export the symbol
        fn.restype = res
        fn.argtypes = args
    if lib.tsff_abi_version() != ABI_VERSION:
        raise TsffError(f"libtsff ABI {lib.tsff_abi_version()} != binding ABI {ABI_VERSION}")
    _lib = lib
    return lib


def check(lib, handle, rc):
    if rc != 0:
        msg = lib.tsff_last_error(handle)
        raise TsffError(f"libtsff error {rc}: {msg.decode() if msg else '?'}")
