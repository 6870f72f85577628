"""ctypes binding of ``include/dcs_beamformer.h`` (libdcs_beamformer.so).

Fails loudly when the library has not been built: there is no CPU fallback.
"""
from __future__ import annotations

import ctypes
from ctypes import POINTER, c_char_p, c_float, c_int, c_size_t, c_uint32, c_uint64, c_void_p
from pathlib import Path

from .parameters import CParams

import os

# DCS_LIB_PATH: load another build of the SAME library (A/B of compiler flags in tools/); never a fallback.
LIB_PATH = Path(os.environ.get("DCS_LIB_PATH") or (Path(__file__).resolve().parent / "csrc" / "libdcs_beamformer.so"))

# status codes (include/dcs_beamformer.h)
DCS_OK = 0
DCS_ERR_INVALID_ARGUMENT = -1
DCS_ERR_UNSUPPORTED = -2
DCS_ERR_NOT_READY = -3
DCS_ERR_OUT_OF_RANGE = -4
DCS_ERR_NO_DEVICE = -5
DCS_ERR_WRONG_DEVICE = -6

# enum dcs_bf_kernel / dcs_bf_bitwidth
NAIVE = 0
MULTIPLE_CHANNELS = 1
MULTIPLE_CHANNELS_AND_TIMESTAMPS = 2
COMBINED_COEFF_GEN_AND_BEAMFORMER_SINGLE_CHANNEL = 3
B16 = 0
B32 = 1


class Timespec(ctypes.Structure):
    """``struct timespec`` (x86-64 Linux: two longs) -- the reference kernels' time arguments."""

    _fields_ = [("tv_sec", ctypes.c_long), ("tv_nsec", ctypes.c_long)]


class DcsError(RuntimeError):
    """A non-zero status from the C-ABI (the reference's GPU_ERRCHK prints and
    exits, ``common/Utils.cpp:8-16``; here it raises)."""

    def __init__(self, status: int, where: str):
        self.status = status
        msg = _lib().dcs_error_string(status).decode() if _LIB is not None else str(status)
        super().__init__(f"{where}: {msg} (status {status})")


# (name, restype, argtypes) -- every symbol include/dcs_beamformer.h declares.
_VP = c_void_p
SIGNATURES = [
    ("dcs_error_string", c_char_p, [c_int]),
    ("dcs_abi_version", c_int, []),
    ("dcs_bf_default_params", c_int, [POINTER(CParams)]),
    ("dcs_bf_output_bytes", c_int, [POINTER(CParams), c_int, c_uint32, POINTER(c_size_t)]),
    ("dcs_bf_delta_times", c_int, [POINTER(CParams), c_uint64, c_uint32, POINTER(c_float)]),
    ("dcs_bf_ts_diff", c_int, [POINTER(Timespec), POINTER(Timespec), POINTER(c_float)]),
    ("dcs_bf_simulate_input", c_int, [POINTER(CParams), _VP]),
    ("dcs_device_count", c_int, [POINTER(c_int)]),
    ("dcs_device_set", c_int, [c_int]),
    ("dcs_device_synchronize", c_int, []),
    ("dcs_device_name", c_int, [c_int, c_char_p, c_size_t]),
    ("dcs_malloc", c_int, [POINTER(_VP), c_size_t]),
    ("dcs_free", c_int, [_VP]),
    ("dcs_host_alloc", c_int, [POINTER(_VP), c_size_t]),
    ("dcs_host_free", c_int, [_VP]),
    ("dcs_memcpy_htod", c_int, [_VP, _VP, c_size_t, _VP]),
    ("dcs_memcpy_dtoh", c_int, [_VP, _VP, c_size_t, _VP]),
    ("dcs_memcpy_dtod", c_int, [_VP, _VP, c_size_t, _VP]),
    ("dcs_memcpy2d_dtoh", c_int, [_VP, c_size_t, _VP, c_size_t, c_size_t, c_size_t, _VP]),
    ("dcs_memset", c_int, [_VP, c_int, c_size_t, _VP]),
    ("dcs_stream_create", c_int, [POINTER(_VP)]),
    ("dcs_stream_destroy", c_int, [_VP]),
    ("dcs_stream_synchronize", c_int, [_VP]),
    ("dcs_event_create", c_int, [POINTER(_VP)]),
    ("dcs_event_destroy", c_int, [_VP]),
    ("dcs_event_record", c_int, [_VP, _VP]),
    ("dcs_event_synchronize", c_int, [_VP]),
    ("dcs_event_elapsed_ms", c_int, [_VP, _VP, POINTER(c_float)]),
    ("dcs_bf_create", c_int, [POINTER(CParams), POINTER(_VP)]),
    ("dcs_bf_destroy", c_int, [_VP]),
    ("dcs_bf_upload_delays", c_int, [_VP, _VP, _VP]),
    ("dcs_bf_set_delays_from_global", c_int, [_VP, _VP, c_uint32, c_uint32, _VP]),
    ("dcs_bf_generate", c_int, [_VP, c_int, c_int, c_uint64, c_uint32, _VP, c_size_t, _VP]),
    ("dcs_bf_generate_slab", c_int, [_VP, c_int, c_uint64, c_uint32, c_uint32, c_uint32, _VP, c_size_t, _VP]),
    ("dcs_bf_generate_dt", c_int, [_VP, c_int, c_int, POINTER(c_float), c_uint32, _VP, c_size_t, _VP]),
    ("dcs_bf_generate_slab_dt", c_int, [_VP, c_int, POINTER(c_float), c_uint32, c_uint32, c_uint32, _VP, c_size_t, _VP]),
    ("dcs_bf_generate_at", c_int, [_VP, c_int, c_int, POINTER(Timespec), POINTER(Timespec), c_uint32, _VP, c_size_t, _VP]),
    ("dcs_bf_generate_and_beamform_dt", c_int, [_VP, POINTER(c_float), c_uint32, _VP, c_size_t, _VP, c_size_t, _VP]),
    ("dcs_bf_beamform_accumulated", c_int, [_VP, c_uint64, c_uint32, _VP, c_size_t, _VP, c_size_t, _VP]),
    ("dcs_bf_beamform_accumulated_dt", c_int, [_VP, c_float, c_uint32, _VP, c_size_t, _VP, c_size_t, _VP]),
    ("dcs_bf_set_tuning", c_int, [_VP, _VP]),
    ("dcs_bf_autotune", c_int, [_VP, c_int, _VP, c_size_t, _VP, _VP]),
    ("dcs_bf_generate_and_beamform", c_int, [_VP, c_uint64, c_uint32, _VP, c_size_t, _VP, c_size_t, _VP]),
    ("dcs_bf_gpu_utilisation", c_int, [POINTER(CParams), c_float, POINTER(c_float)]),
    ("dcs_bf_stream_begin", c_int, [_VP, c_int, c_uint32, c_uint32, _VP, c_size_t, _VP, POINTER(_VP)]),
    ("dcs_bf_stream_tick", c_int, [_VP, c_uint64, _VP]),
    ("dcs_bf_stream_tick_dt", c_int, [_VP, c_float, _VP]),
    ("dcs_bf_stream_tick_at", c_int, [_VP, POINTER(Timespec), POINTER(Timespec), _VP]),
    ("dcs_bf_stream_tick_from_global", c_int, [_VP, c_uint64, _VP, c_uint32, c_uint32]),
    ("dcs_bf_stream_tick_dt_from_global", c_int, [_VP, c_float, _VP, c_uint32, c_uint32]),
    ("dcs_bf_stream_tick_at_from_global", c_int, [_VP, POINTER(Timespec), POINTER(Timespec), _VP, c_uint32, c_uint32]),
    ("dcs_bf_stream_end", c_int, [_VP]),
]

ABI_VERSION = 3  # include/dcs_beamformer.h: DCS_BF_ABI_VERSION

_LIB = None


def _lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m dc_sand_amd.build` "
                "(hipcc, gfx950). dc_sand_amd has no CPU fallback."
            )
        lib = ctypes.CDLL(str(LIB_PATH))
        for name, restype, argtypes in SIGNATURES:
            fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        if lib.dcs_abi_version() != ABI_VERSION:  # a stale build: struct dcs_bf_tuning changed between 2 and 3
            raise ImportError(f"{LIB_PATH} has ABI version {lib.dcs_abi_version()}, this package binds version {ABI_VERSION}: "
                              "rebuild it with `python -m dc_sand_amd.build --force`")
        _LIB = lib
    return _LIB


def lib() -> ctypes.CDLL:
    return _lib()


def check(status: int, where: str) -> None:
    if status != DCS_OK:
        raise DcsError(status, where)
