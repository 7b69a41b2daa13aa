"""ctypes binding of libresnmtf_hip.so (the C-ABI declared in include/resnmtf_hip.h).

The shared library is built in-tree by ``__graft_entry__.build()`` /
``python -m resnmtf_amd.build``.  There is deliberately no fallback: if the library is
missing, or no gfx950 device is usable, every compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libresnmtf_hip.so")

OK = 0
ERR_NAMES = {1: "INVALID", 2: "NO_DEVICE", 3: "HIP", 4: "ALLOC", 5: "STATE"}
FACTOR_F, FACTOR_G, FACTOR_S, FACTOR_FBLOCK, FACTOR_FBLOCK_ALL = 0, 1, 2, 3, 4
FACTOR_GBLOCK, FACTOR_GBLOCK_ALL, FACTOR_SBLOCK, FACTOR_SBLOCK_ALL = 5, 6, 7, 8
PHASE_F, PHASE_G, PHASE_S, PHASE_F_ALL, PHASE_LOCAL_SWEEP = 0, 1, 2, 3, 4
PHASE_XTF, PHASE_G_ALL, PHASE_XG, PHASE_S_ALL = 5, 6, 7, 8
PHASE_SLICE_F, PHASE_SLICE_XTF, PHASE_SLICE_G, PHASE_SLICE_XG = 9, 10, 11, 12
FACTOR_U_SEND, FACTOR_U_RECV, FACTOR_FNEW_SEND, FACTOR_FNEW_RECV = 9, 10, 11, 12
FACTOR_T_SEND, FACTOR_T_RECV, FACTOR_GNEW_SEND, FACTOR_GNEW_RECV, FACTOR_F_SLICE, FACTOR_G_SLICE = 13, 14, 15, 16, 17, 18
TIMED_KINDS = ("xg", "xtf", "f_chain", "g_chain", "s_chain", "pack")
ABI_VERSION = 2
MAX_K = 64


class ResnmtfError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__(f"resnmtf_hip error {code} ({ERR_NAMES.get(code, '?')}): {text}")
        self.code = code


class Options(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int), ("device_id", C.c_int), ("stream", C.c_void_p),
        ("use_graph", C.c_int), ("check_every", C.c_int), ("target_workgroups", C.c_int),
        ("time_kernels", C.c_int), ("pass_waves", C.c_int), ("pass_splits_xg", C.c_int),
        ("pass_splits_xtf", C.c_int), ("pass_lds_pad_kb", C.c_int), ("update_blocks", C.c_int),
        ("no_pitch_pad", C.c_int), ("kk_mode", C.c_int), ("bf16_split", C.c_int), ("replicate_f", C.c_int),
        ("no_f_chain", C.c_int), ("x_half", C.c_int), ("half_unroll", C.c_int), ("replicate_gs", C.c_int),
        ("wait_mode", C.c_int), ("slice_chains", C.c_int), ("slice_index", C.c_int), ("slice_count", C.c_int),
        ("slice_p2p", C.c_int), ("xcd_order", C.c_int), ("fuse_updates", C.c_int),
    ]


class PassTiming(C.Structure):
    _fields_ = [
        ("xg_ms_total", C.c_double), ("xtf_ms_total", C.c_double),
        ("xg_launches", C.c_longlong), ("xtf_launches", C.c_longlong),
        ("xg_bytes", C.c_double), ("xtf_bytes", C.c_double),
        ("xg_flops", C.c_double), ("xtf_flops", C.c_double),
    ]


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_h = C.c_void_p

# name -> (restype, argtypes); must list every symbol include/resnmtf_hip.h declares
SIGNATURES = {
    "resnmtf_abi_version": (C.c_int, []),
    "resnmtf_device_count": (C.c_int, []),
    "resnmtf_default_options": (None, [C.POINTER(Options)]),
    "resnmtf_last_error": (C.c_char_p, [_h]),
    "resnmtf_create": (C.c_int, [C.c_int, _ip, _ip, _ip, _ip, C.POINTER(Options), C.POINTER(_h)]),
    "resnmtf_destroy": (C.c_int, [_h]),
    "resnmtf_set_view": (C.c_int, [_h, C.c_int, _dp]),
    "resnmtf_set_view_raw": (C.c_int, [_h, C.c_int, _dp, _ip]),
    "resnmtf_copy_view": (C.c_int, [_h, C.c_int, _h, C.c_int]),
    "resnmtf_shuffle_view": (C.c_int, [_h, C.c_int, _h, C.c_int, C.c_ulonglong, C.c_int]),
    "resnmtf_subsample_view": (C.c_int, [_h, C.c_int, _h, C.c_int, _ip, _ip]),
    "resnmtf_view_empty_lines": (C.c_int, [_h, C.c_int, _ip, _ip, C.POINTER(C.c_ubyte), C.POINTER(C.c_ubyte)]),
    "resnmtf_get_view": (C.c_int, [_h, C.c_int, _dp]),
    "resnmtf_set_factors": (C.c_int, [_h, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "resnmtf_init_svd": (C.c_int, [_h, C.c_int, C.c_ulonglong, C.c_double, C.c_int, _dp]),
    "resnmtf_set_restrictions": (C.c_int, [_h, _dp, _dp, _dp]),
    "resnmtf_set_shared_rows": (C.c_int, [_h, C.c_int, C.c_int, C.c_int, _ip, _ip]),
    "resnmtf_set_shared_cols": (C.c_int, [_h, C.c_int, C.c_int, C.c_int, _ip, _ip]),
    "resnmtf_run": (C.c_int, [_h, C.c_int, C.c_double, C.c_int, _dp, C.c_int, _ip]),
    "resnmtf_get_factors": (C.c_int, [_h, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "resnmtf_finalise": (C.c_int, [_h, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "resnmtf_reserve_sweeps": (C.c_int, [_h, C.c_int]),
    "resnmtf_prepare": (C.c_int, [_h]),
    "resnmtf_phase": (C.c_int, [_h, C.c_int, C.c_int, C.c_int]),
    "resnmtf_factor_device_ptr": (C.c_int, [_h, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "resnmtf_view_errors": (C.c_int, [_h, C.c_int, C.c_int, C.c_int, _dp]),
    "resnmtf_synchronize": (C.c_int, [_h]),
    "resnmtf_pass_timings": (C.c_int, [_h, C.POINTER(PassTiming), C.c_int]),
    "resnmtf_view_image_info": (C.c_int, [_h, C.c_int, _ip, _dp]),
    "resnmtf_kernel_timings": (C.c_int, [_h, _dp, C.POINTER(C.c_longlong), C.c_int]),
    "resnmtf_set_stop_tolerance": (C.c_int, [_h, C.c_double]),
    "resnmtf_loop_state": (C.c_int, [_h, _ip, _ip, _ip]),
    "resnmtf_slice_info": (C.c_int, [_h, _ip, _ip]),
    "resnmtf_p2p_export": (C.c_int, [_h, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "resnmtf_p2p_import": (C.c_int, [_h, C.c_int, C.c_void_p, C.c_size_t]),
    "resnmtf_p2p_selftest": (C.c_int, [_h, C.c_int]),
}

_lib = None


def _pin_hip_runtime():
    """One HIP runtime per process.  PyTorch wheels bundle their own ``libamdhip64.so`` (+ HSA runtime); this library is
    linked against the system's.  Whichever is loaded first serves both (same SONAME) -- and if it is the system's,
    a later ``import torch`` finds no device ("No HIP GPUs are available": its own HSA runtime comes up beside a foreign
    HIP).  So when torch is installed but not yet imported, its copy is loaded here first; without torch (an R process
    calling the C-ABI) the system's runtime is used as linked."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load() -> C.CDLL:
    """Load the in-tree HIP library; raises ImportError loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    _pin_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -m resnmtf_amd.build` "
            "(hipcc --offload-arch=gfx950).  resnmtf_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.resnmtf_abi_version() != ABI_VERSION:
        raise ImportError("libresnmtf_hip.so ABI version mismatch")
    _lib = lib
    return lib
