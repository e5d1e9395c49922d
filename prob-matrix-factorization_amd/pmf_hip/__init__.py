"""ctypes binding of libpmf_hip.so (C-ABI declared in include/pmf_hip.h).

The shared library is the product: there is no CPU fallback.  Importing this
module never loads it; the first use does, and raises `PmfLibraryError` if the
library has not been built (`python -c "import __graft_entry__ as g; g.build()"`).
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpmf_hip.so")
# The TEST build of the same sources (-DPMF_TEST_TRANSPORT): adds the `hostshm` rehearsal transport (ranks sharing
# one GPU exchange through POSIX shared memory) and its entry point pmf_comm_init_hostshm.  Never loaded unless the
# process asks for that transport (PMF_COMM_TRANSPORT=hostshm) or sets PMF_HIP_TEST_LIBRARY=1.
TEST_LIB_PATH = os.path.join(_HERE, "libpmf_hip_test.so")
TEST_ONLY_SIGNATURES = ("pmf_comm_init_hostshm",)

F32, F64 = 0, 1
USER, ITEM = 0, 1
(ARR_FACTOR, ARR_SHAPE, ARR_RATE, ARR_PRIOR_RATE, ARR_HYPER_RATE, ARR_COV, ARR_BIAS, ARR_SCALE, ARR_SCALE_SHAPE,
 ARR_SCALE_RATE) = range(10)
PREDICT_BIAS, PREDICT_SCALE = 1, 2
KERNEL_NAMES = ("gamma_sweep", "gamma_final", "gauss_accum", "gauss_solve", "gauss_bias",
                "eval", "predict", "topk", "gauss_combine", "gauss_sgd", "comm_allreduce", "comm_wait")
UNIQUE_ID_BYTES = 128
TRANSPORT_RCCL, TRANSPORT_HOSTSHM = 0, 1
OP_SUM, OP_MAX = 0, 1
EXCHANGE = {"auto": 0, "allreduce": 1, "scatter_gather": 2}
MAX_LABELS = 32


class PmfLibraryError(RuntimeError):
    """libpmf_hip.so is missing or does not export the declared ABI."""


class PmfError(RuntimeError):
    """A C-ABI call returned a negative status."""


_p = C.c_void_p
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_f64p = C.POINTER(C.c_double)

# name -> (restype, argtypes); must list every function include/pmf_hip.h declares
SIGNATURES = {
    "pmf_abi_version": (C.c_int, []),
    "pmf_last_error": (C.c_char_p, []),
    "pmf_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "pmf_ctx_create": (C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(_p)]),
    "pmf_ctx_destroy": (C.c_int, [_p]),
    "pmf_ctx_set_stream": (C.c_int, [_p, _p]),
    "pmf_ctx_sync": (C.c_int, [_p]),
    "pmf_gauss_sgd_sweep": (C.c_int, [_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]),
    "pmf_ctx_sgd_stats_width": (C.c_int, [_p, C.POINTER(C.c_int)]),
    "pmf_gauss_sgd_accumulate": (C.c_int, [_p, C.c_int, _p, C.c_double, C.c_double, C.c_double, C.c_double]),
    "pmf_gauss_sgd_finalize": (C.c_int, [_p, C.c_int, _p]),
    "pmf_ctx_set_row_chunks": (C.c_int, [_p, C.c_int, C.c_int]),
    "pmf_ctx_chunk_rows": (C.c_int, [_p, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pmf_ctx_select_chunk": (C.c_int, [_p, C.c_int, C.c_int]),
    "pmf_ctx_device_bytes": (C.c_int, [_p, _i64p]),
    "pmf_ctx_set_ratings": (C.c_int, [_p, C.c_int64, _i32p, _i32p, _f64p]),
    "pmf_set_array": (C.c_int, [_p, C.c_int, C.c_int, _f64p]),
    "pmf_get_array": (C.c_int, [_p, C.c_int, C.c_int, _f64p]),
    "pmf_get_array_rows": (C.c_int, [_p, C.c_int, C.c_int, C.c_int64, _i64p, _f64p]),
    "pmf_set_array_rows": (C.c_int, [_p, C.c_int, C.c_int, C.c_int64, _i64p, _f64p]),
    "pmf_set_cov_identity": (C.c_int, [_p, C.c_int, C.c_double]),
    "pmf_gamma_sweep": (C.c_int, [_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double]),
    "pmf_gamma_ext_sweep": (C.c_int, [_p, C.c_int, C.c_double, C.c_double]),
    "pmf_ctx_kpad": (C.c_int, [_p, C.POINTER(C.c_int)]),
    "pmf_gamma_accumulate": (C.c_int, [_p, C.c_int, _p]),
    "pmf_gamma_finalize": (C.c_int, [_p, C.c_int, _p, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double]),
    "pmf_gauss_factor_sweep": (C.c_int, [_p, C.c_int, C.c_double, C.c_double]),
    "pmf_gauss_bias_sweep": (C.c_int, [_p, C.c_int, C.c_double, C.c_double]),
    "pmf_ctx_cov_stride": (C.c_int, [_p, C.POINTER(C.c_int)]),
    "pmf_gauss_factor_accumulate": (C.c_int, [_p, C.c_int, _p]),
    "pmf_gauss_factor_finalize": (C.c_int, [_p, C.c_int, _p, C.c_double, C.c_double]),
    "pmf_gauss_bias_accumulate": (C.c_int, [_p, C.c_int, _p]),
    "pmf_gauss_bias_finalize": (C.c_int, [_p, C.c_int, _p, C.c_double, C.c_double]),
    "pmf_predict": (C.c_int, [_p, C.c_int64, _i32p, _i32p, C.c_int, C.c_double, _f64p]),
    "pmf_eval_set": (C.c_int, [_p, C.c_int64, _i32p, _i32p, _f64p, _i32p, C.c_int]),
    "pmf_eval_run": (C.c_int, [_p, C.c_int, C.c_double, _f64p, _f64p, _i64p]),
    "pmf_topk_items": (C.c_int, [_p, C.c_int64, _i32p, C.c_int, C.c_int, _i32p, _f64p]),
    "pmf_prof_enable": (C.c_int, [_p, C.c_int]),
    "pmf_prof_reset": (C.c_int, [_p]),
    "pmf_prof_get": (C.c_int, [_p, C.c_int, _f64p, _i64p]),
    "pmf_prof_gather_ceiling": (C.c_int, [_p, C.c_int, C.c_int, _f64p]),
    "pmf_comm_unique_id": (C.c_int, [_p]),
    "pmf_comm_init": (C.c_int, [_p, C.c_int, C.c_int, _p]),
    "pmf_comm_set_exchange": (C.c_int, [_p, C.c_int]),
    "pmf_comm_attach": (C.c_int, [_p, _p]),
    "pmf_comm_destroy": (C.c_int, [_p]),
    "pmf_comm_info": (C.c_int, [_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pmf_comm_barrier": (C.c_int, [_p]),
    "pmf_comm_allreduce_host": (C.c_int, [_p, _f64p, C.c_int64, C.c_int]),
    "pmf_comm_gather_user_rows": (C.c_int, [_p, C.c_int, _i64p, _f64p]),
}


def wants_test_library():
    return os.environ.get("PMF_HIP_TEST_LIBRARY") == "1" or os.environ.get("PMF_COMM_TRANSPORT") == "hostshm"

_lib = None
_loaded_before_torch = False


def loaded_before_torch():
    """True when libpmf_hip.so (and with it /opt/rocm's HIP runtime) was mapped into a process that
    had not imported torch yet: a later `import torch` would then find no GPU."""
    return _lib is not None and _loaded_before_torch


def load():
    """Load libpmf_hip.so and bind every declared entry point (once)."""
    global _lib
    if _lib is not None:
        return _lib
    path = TEST_LIB_PATH if wants_test_library() else LIB_PATH
    if os.environ.get("PMF_HIP_LIBRARY"):      # an explicit build of the same ABI (tools/: diagnostic builds with in-kernel stamps)
        path = os.environ["PMF_HIP_LIBRARY"]
    if not os.path.exists(path):
        raise PmfLibraryError(
            f"{path} not found: the HIP engine has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` from the repository root. "
            "There is no CPU fallback.")
    # The CAVI engine does not use PyTorch.  One interaction has to be known, though: PyTorch-ROCm
    # wheels bundle their own HIP runtime, and if /opt/rocm's libamdhip64 (this library's dependency)
    # is mapped first, a later `import torch` ends up with two runtimes and silently reports no GPU.
    # A process that ALSO wants torch on the GPU (the hpf_pytorch model, the comparison drivers)
    # must therefore import torch before the first engine call -- src/models/hpf_pytorch.py and the
    # drivers do so at import time; PMF_HIP_TORCH_PRELOAD=1 forces it here.
    global _loaded_before_torch
    _loaded_before_torch = "torch" not in sys.modules
    # RCCL between processes needs dmabuf IPC on hosts whose driver has no legacy IPC (hipIpcGetMemHandle:
    # invalid argument otherwise); the runtime reads this when it initialises, i.e. after this point.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if _loaded_before_torch and os.environ.get("PMF_HIP_TORCH_PRELOAD") == "1":
        try:
            import torch  # noqa: F401
            _loaded_before_torch = False
        except Exception:
            pass
    try:
        lib = C.CDLL(path)
    except OSError as exc:  # missing libamdhip64 etc.
        raise PmfLibraryError(f"cannot load {path}: {exc}") from exc
    sigs = dict(SIGNATURES)
    if path == TEST_LIB_PATH:
        sigs["pmf_comm_init_hostshm"] = SIGNATURES["pmf_comm_init"]
    for name, (res, args) in sigs.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise PmfLibraryError(f"{path} does not export {name}") from exc
        fn.restype = res
        fn.argtypes = args
    lib.pmf_path = path
    _lib = lib
    return lib


def check(status, what=""):
    if status != 0:
        msg = load().pmf_last_error()
        raise PmfError(f"{what} failed ({status}): {msg.decode() if msg else ''}")


def device_count():
    n = C.c_int(0)
    rc = load().pmf_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def as_i32(ids, what="ids"):
    """int ids -> contiguous int32 (the ABI's id type); refuses values that do not fit."""
    a = np.asarray(ids)
    if a.dtype != np.int32:
        a64 = a.astype(np.int64, copy=False)
        if a64.size and (a64.max() > np.iinfo(np.int32).max or a64.min() < np.iinfo(np.int32).min):
            raise ValueError(f"{what}: values do not fit int32")
        a = a64.astype(np.int32)
    return np.ascontiguousarray(a)


def as_f64(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))


def ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


from .engine import Context  # noqa: E402,F401
