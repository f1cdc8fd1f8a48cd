"""ctypes binding of libposeidon_mi355x.so (the C ABI declared in include/poseidon_mi355x.h).

There is deliberately no fallback: if the shared library is missing, import fails; if no HIP device is
usable, context creation raises PmxError (PMX_ERR_HIP).
"""
from __future__ import annotations

import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libposeidon_mi355x.so")
# the same library with the device-group test hooks compiled in (csrc/Makefile: -DPMX_TEST_HOOKS; include/poseidon_mi355x_testing.h).
# Only tests and rehearsals load it, and only by asking for it in code (use_test_library) - never by an environment variable.
TEST_LIB_PATH = os.path.join(HERE, "libposeidon_mi355x_test.so")

PMX_OK = 0
PMX_ERR_CONFIG = -1
PMX_ERR_ARG = -2
PMX_ERR_HIP = -3
PMX_ERR_UNSUPPORTED = -4
PMX_ERR_RCCL = -5
PMX_ERR_HOST = -6

ABI_VERSION = 5
UNIQUE_ID_BYTES = 128
MAX_LOCAL_DEVICES = 16

MODE_ABSORBING = 0
MODE_SQUEEZING = 1


class PmxError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"pmx error {code}: {message}")
        self.code = code


class PmxConfig(ctypes.Structure):
    _fields_ = [
        ("full_rounds", ctypes.c_uint32), ("partial_rounds", ctypes.c_uint32),
        ("alpha", ctypes.c_uint64),
        ("rate", ctypes.c_uint32), ("capacity", ctypes.c_uint32),
        ("modulus", ctypes.c_uint64 * 4),
        ("ark", ctypes.c_void_p), ("mds", ctypes.c_void_p),
    ]


class PmxMgpuInfo(ctypes.Structure):
    _fields_ = [
        ("world", ctypes.c_int), ("n_local", ctypes.c_int), ("first_rank", ctypes.c_int), ("width", ctypes.c_int),
        ("rccl_version", ctypes.c_int), ("comm_ranks", ctypes.c_int), ("comm_first_rank", ctypes.c_int),
        ("devices", ctypes.c_int * 16),
    ]


OP_PERMUTE, OP_HASH, OP_COMPRESS, OP_ABSORB, OP_SQUEEZE = range(5)


class PmxEngineInfo(ctypes.Structure):
    _fields_ = [
        ("engine", ctypes.c_char * 64), ("width", ctypes.c_int), ("threads", ctypes.c_int), ("waves_per_simd", ctypes.c_int),
        ("lds_bytes", ctypes.c_int), ("optimised", ctypes.c_int), ("row_tables", ctypes.c_int), ("lane_tables", ctypes.c_int),
        ("mfma_dense", ctypes.c_int), ("launches", ctypes.c_int), ("partial_window", ctypes.c_int),
    ]


class PmxValuPeak(ctypes.Structure):
    _fields_ = [
        ("lane_mads_per_s", ctypes.c_double), ("lane_mads_per_s_vcc", ctypes.c_double),
        ("lane_mads_per_s_sgpr", ctypes.c_double), ("best_lane_mads_per_s", ctypes.c_double),
        ("shader_clock_hz", ctypes.c_double), ("theoretical_lane_mads_per_s", ctypes.c_double),
        ("compute_units", ctypes.c_int), ("launches", ctypes.c_int),
    ]


class PmxIssueSlot(ctypes.Structure):
    _fields_ = [
        ("ns_12mad_4simple", ctypes.c_double), ("ns_4mad_12simple", ctypes.c_double), ("ns_16mad", ctypes.c_double),
        ("ns_floor", ctypes.c_double), ("waves_per_simd", ctypes.c_int), ("compute_units", ctypes.c_int), ("launches", ctypes.c_int),
    ]


_u64p = ctypes.c_void_p
_u32p = ctypes.c_void_p
_sz = ctypes.c_size_t

# name -> (restype, argtypes); one entry per function declared in include/poseidon_mi355x.h
SIGNATURES = {
    "pmx_abi_version": (ctypes.c_int, []),
    "pmx_last_error": (ctypes.c_char_p, []),
    "pmx_device_count": (ctypes.c_int, []),
    "pmx_host_alloc": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), _sz]),
    "pmx_host_free": (ctypes.c_int, [ctypes.c_void_p]),
    "pmx_device_alloc": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), _sz]),
    "pmx_device_free": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p]),
    "pmx_device_upload": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, _sz, ctypes.c_void_p]),
    "pmx_device_download": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, _sz, ctypes.c_void_p]),
    "pmx_stream_synchronize": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p]),
    "pmx_find_poseidon_ark_and_mds": (ctypes.c_int, [_u64p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32,
                                                     ctypes.c_uint32, ctypes.c_uint32, _u64p, _u64p]),
    "pmx_mont_constants": (ctypes.c_int, [_u64p, _u64p, _u64p, _u64p]),
    "pmx_to_mont": (ctypes.c_int, [_u64p, _u64p, _sz]),
    "pmx_from_mont": (ctypes.c_int, [_u64p, _u64p, _sz]),
    "pmx_ctx_create": (ctypes.c_int, [ctypes.POINTER(PmxConfig), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "pmx_ctx_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "pmx_ctx_acquire": (ctypes.c_int, [ctypes.POINTER(PmxConfig), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "pmx_ctx_release": (ctypes.c_int, [ctypes.c_void_p]),
    "pmx_ctx_cache_clear": (ctypes.c_int, []),
    "pmx_ctx_width": (ctypes.c_int, [ctypes.c_void_p]),
    "pmx_ctx_engine_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _sz, _sz, ctypes.POINTER(PmxEngineInfo)]),
    "pmx_permute_batch": (ctypes.c_int, [ctypes.c_void_p, _u64p, _sz]),
    "pmx_permute_batch_dev": (ctypes.c_int, [ctypes.c_void_p, _u64p, _sz, ctypes.c_void_p]),
    "pmx_hash_batch": (ctypes.c_int, [ctypes.c_void_p, _u64p, _sz, _u64p, _sz, _sz]),
    "pmx_hash_batch_dev": (ctypes.c_int, [ctypes.c_void_p, _u64p, _sz, _u64p, _sz, _sz, ctypes.c_void_p]),
    "pmx_sponge_absorb_batch": (ctypes.c_int, [ctypes.c_void_p, _u64p, _u32p, _u32p, _u64p, _sz, _sz]),
    "pmx_sponge_squeeze_batch": (ctypes.c_int, [ctypes.c_void_p, _u64p, _u32p, _u32p, _u64p, _sz, _sz]),
    "pmx_sponge_absorb_batch_dev": (ctypes.c_int, [ctypes.c_void_p, _u64p, _u32p, _u32p, _u64p, _sz, _sz,
                                                   ctypes.c_void_p]),
    "pmx_sponge_squeeze_batch_dev": (ctypes.c_int, [ctypes.c_void_p, _u64p, _u32p, _u32p, _u64p, _sz, _sz,
                                                    ctypes.c_void_p]),
    "pmx_merkle_2to1": (ctypes.c_int, [ctypes.c_void_p, _u64p, _sz, _u64p, _u64p]),
    "pmx_merkle_2to1_dev": (ctypes.c_int, [ctypes.c_void_p, _u64p, _sz, ctypes.c_void_p]),
    "pmx_merkle_2to1_forest": (ctypes.c_int, [ctypes.c_void_p, _u64p, _sz, _sz, _u64p, _u64p]),
    "pmx_merkle_2to1_forest_dev": (ctypes.c_int, [ctypes.c_void_p, _u64p, _sz, _sz, ctypes.c_void_p]),
    "pmx_merkle_paths": (ctypes.c_int, [_u64p, _sz, _u64p, _sz, _u64p]),
    "pmx_merkle_verify_paths": (ctypes.c_int, [ctypes.c_void_p, _u64p, _u64p, _u64p, _sz, _sz, _u64p, ctypes.c_void_p]),
    "pmx_merkle_verify_paths_dev": (ctypes.c_int, [ctypes.c_void_p, _u64p, _u64p, _u64p, _sz, _sz, _u64p, ctypes.c_void_p, _u64p,
                                                   ctypes.c_void_p]),
    # device groups
    "pmx_shard_bounds": (ctypes.c_int, [_sz, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_sz), ctypes.POINTER(_sz)]),
    "pmx_mgpu_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "pmx_mgpu_create": (ctypes.c_int, [ctypes.POINTER(PmxConfig), ctypes.c_int, ctypes.c_void_p,
                                       ctypes.POINTER(ctypes.c_void_p)]),
    "pmx_mgpu_create_rank": (ctypes.c_int, [ctypes.POINTER(PmxConfig), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]),
    "pmx_mgpu_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "pmx_mgpu_get_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(PmxMgpuInfo)]),
    "pmx_mgpu_stream": (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int]),
    "pmx_mgpu_ctx": (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int]),
    "pmx_mgpu_synchronize": (ctypes.c_int, [ctypes.c_void_p]),
    "pmx_mgpu_permute_batch": (ctypes.c_int, [ctypes.c_void_p, _u64p, _sz]),
    "pmx_mgpu_hash_batch": (ctypes.c_int, [ctypes.c_void_p, _u64p, _sz, _u64p, _sz, _sz]),
    "pmx_mgpu_permute_shards_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, _sz]),
    "pmx_mgpu_all_gather_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _sz, _sz]),
    "pmx_mgpu_gather_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _sz, _sz, ctypes.c_int]),
    "pmx_mgpu_permute_gather_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _sz, ctypes.c_int, ctypes.c_int]),
    "pmx_mgpu_merkle_2to1_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _sz]),
    "pmx_mgpu_merkle_2to1": (ctypes.c_int, [ctypes.c_void_p, _u64p, _sz, _u64p]),
}

# include/poseidon_mi355x_diag.h: libposeidon_mi355x_diag.so, the benchmark diagnostics (bench.py) - not part of the shipped library
DIAG_LIB_PATH = os.path.join(HERE, "libposeidon_mi355x_diag.so")
DIAG_SIGNATURES = {
    "pmx_diag_last_error": (ctypes.c_char_p, []),
    "pmx_diag_int_valu_peak": (ctypes.c_int, [ctypes.c_int, ctypes.c_double, ctypes.POINTER(PmxValuPeak)]),
    "pmx_diag_issue_slot": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.POINTER(PmxIssueSlot)]),
}

# include/poseidon_mi355x_testing.h: exported by libposeidon_mi355x_test.so only
TEST_HOOK_SIGNATURES = {
    "pmx_test_hooks_enabled": (ctypes.c_int, []),
    "pmx_mgpu_test_fault": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "pmx_mgpu_test_shared_device": (ctypes.c_int, [ctypes.c_int]),
}

_lib = None
_path = LIB_PATH
_hooks = False


def use_library(path: str, test_hooks: bool = False) -> None:
    """Bind another build of the library (tests only: the test-hook build, a coverage build).  Before the first lib()."""
    global _path, _hooks
    if _lib is not None and os.path.abspath(path) != os.path.abspath(_path):
        raise RuntimeError("use_library() after another build of the library was loaded")
    _path, _hooks = path, test_hooks


def use_test_library() -> None:
    """Bind libposeidon_mi355x_test.so (the shipped objects + the device-group test hooks) instead of the shipped library.
    Must be called before the first lib() of the process; tests and rehearsals only."""
    use_library(TEST_LIB_PATH, test_hooks=True)


def is_test_library() -> bool:
    return _hooks


def library_path() -> str:
    return _path


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_path):
            raise ImportError(
                f"{_path} is missing: build it with `make -C sponge_amd/csrc` "
                "(or __graft_entry__.build()). There is no CPU fallback.")
        handle = ctypes.CDLL(_path)
        sigs = list(SIGNATURES.items()) + (list(TEST_HOOK_SIGNATURES.items()) if _hooks else [])
        for name, (restype, argtypes) in sigs:
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


_diag = None


def diag_lib() -> ctypes.CDLL:
    """libposeidon_mi355x_diag.so (benchmark diagnostics; a library of its own)."""
    global _diag
    if _diag is None:
        if not os.path.exists(DIAG_LIB_PATH):
            raise ImportError(f"{DIAG_LIB_PATH} is missing: build it with `make -C sponge_amd/csrc`")
        handle = ctypes.CDLL(DIAG_LIB_PATH)
        for name, (restype, argtypes) in DIAG_SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _diag = handle
    return _diag


def check_diag(rc: int) -> None:
    if rc != PMX_OK:
        raise PmxError(rc, diag_lib().pmx_diag_last_error().decode("utf-8", "replace"))


def check(rc: int) -> None:
    if rc != PMX_OK:
        raise PmxError(rc, lib().pmx_last_error().decode("utf-8", "replace"))
