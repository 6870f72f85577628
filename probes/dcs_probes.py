"""ctypes binding of ``include/dcs_probes.h`` (probes/libdcs_probes.so).  Test and measurement
infrastructure: the product package never imports this.

The probes library also exports the whole ``dcs_bf_*`` API built with ``-DDCS_PROBES`` (the measurement knobs of
``struct dcs_probe_knobs`` honoured: :func:`set_knobs`); to drive it through the ordinary Python wrappers start the
process with ``DCS_LIB_PATH=probes/libdcs_probes.so``.
"""
from __future__ import annotations

import ctypes
from ctypes import POINTER, byref, c_float, c_int, c_size_t, c_uint32, c_uint64, c_void_p

import numpy as np

from .build import LIB, build

_VP = c_void_p
SIGNATURES = [
    ("dcs_probe_sincos", c_int, [c_int, _VP, c_size_t, _VP, _VP, _VP]),
    ("dcs_probe_fill", c_int, [_VP, c_size_t, c_int, _VP]),
    ("dcs_probe_copy", c_int, [_VP, _VP, c_size_t, c_int, c_int, _VP]),
    ("dcs_probe_one_store", c_int, [_VP, c_size_t, c_int, c_int, c_uint32, _VP]),
    ("dcs_probe_reduce", c_int, [_VP, c_size_t, POINTER(c_uint64), POINTER(c_float), _VP]),
    ("dcs_probe_mfma", c_int, [c_int, c_uint32, c_uint32, _VP, _VP]),
    ("dcs_probe_store_pattern", c_int, [_VP, c_uint32, c_uint32, c_uint32, c_uint32, c_int, c_int, c_int, c_uint32, _VP]),
    ("dcs_probe_set_knobs", c_int, [_VP, _VP]),
    ("dcs_probe_xcd_grouped", c_uint32, [c_uint32, c_uint32, c_uint32]),
]

# struct dcs_probe_knobs (include/dcs_probes.h), in order
KNOB_FIELDS = ("nomath", "pace", "fail_at_step", "bacc_probe", "bacc_rounds", "bacc_no_share", "bacc_plain", "bacc_wg_per_cu",
               "bacc_unstaged", "bacc_order", "bacc_nbt", "bacc_waves")

_LIB = None


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        if not LIB.exists():
            build()
        L = ctypes.CDLL(str(LIB))
        for name, restype, argtypes in SIGNATURES:
            fn = getattr(L, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _LIB = L
    return _LIB


def _check(status: int, where: str) -> None:
    if status != 0:
        raise RuntimeError(f"{where}: status {status}")


def _s(stream) -> c_void_p:
    if stream is None:
        return c_void_p(None)
    return c_void_p(int(getattr(stream, "handle", stream)))


def tensor_properties(d_ptr, nbytes: int, stream=None) -> tuple[int, float]:
    """(checksum, max | |z|^2 - 1 |) of an fp32 coefficient tensor on the device."""
    ck = c_uint64(0)
    dev = c_float(0.0)
    _check(lib().dcs_probe_reduce(c_void_p(int(d_ptr)), int(nbytes), byref(ck), byref(dev), _s(stream)), "dcs_probe_reduce")
    return int(ck.value), float(dev.value)


def sincos(which: int, d_x, n: int, d_sin, d_cos, stream=None) -> None:
    _check(lib().dcs_probe_sincos(int(which), c_void_p(int(d_x)), int(n), c_void_p(int(d_sin)), c_void_p(int(d_cos)), _s(stream)),
           "dcs_probe_sincos")


def fill(d_out, nbytes: int, nontemporal: int = 1, stream=None) -> None:
    _check(lib().dcs_probe_fill(c_void_p(int(d_out)), int(nbytes), int(nontemporal), _s(stream)), "dcs_probe_fill")


def copy(d_in, d_out, nbytes: int, store_mode: int = 1, per_thread: int = 1, stream=None) -> None:
    _check(lib().dcs_probe_copy(c_void_p(int(d_in)), c_void_p(int(d_out)), int(nbytes), int(store_mode), int(per_thread), _s(stream)),
           "dcs_probe_copy")


def store_pattern(d_out, rows, cols_kib, qb, rb, order=0, xcd_remap=0, store_mode=1, block_threads=256, stream=None) -> None:
    _check(lib().dcs_probe_store_pattern(c_void_p(int(d_out)), rows, cols_kib, qb, rb, order, xcd_remap, store_mode, block_threads,
                                         _s(stream)), "dcs_probe_store_pattern")


def one_store(d_out, nbytes, store_mode, stores_per_thread, row_bytes, stream=None) -> None:
    _check(lib().dcs_probe_one_store(c_void_p(int(d_out)), int(nbytes), int(store_mode), int(stores_per_thread), int(row_bytes),
                                     _s(stream)), "dcs_probe_one_store")


def mfma(which: int, blocks: int, iters: int, d_out, stream=None) -> None:
    _check(lib().dcs_probe_mfma(int(which), int(blocks), int(iters), c_void_p(int(d_out)), _s(stream)), "dcs_probe_mfma")


def set_knobs(gen, **knobs) -> None:
    """``dcs_probe_set_knobs`` on a :class:`dc_sand_amd.generator.SteeringCoefficientGenerator` whose context was made
    by THIS library (the process runs with ``DCS_LIB_PATH=probes/libdcs_probes.so``); no keyword = reset."""
    from dc_sand_amd import _lib as product

    if product.LIB_PATH.resolve() != LIB.resolve():
        raise RuntimeError("set_knobs needs the probes build behind the wrappers: run with DCS_LIB_PATH=probes/libdcs_probes.so")
    unknown = set(knobs) - set(KNOB_FIELDS)
    if unknown:
        raise TypeError(f"unknown knob(s) {sorted(unknown)}")
    if not knobs:
        _check(lib().dcs_probe_set_knobs(c_void_p(gen._h), c_void_p(None)), "dcs_probe_set_knobs")
        return
    k = (ctypes.c_int32 * len(KNOB_FIELDS))(*[int(knobs.get(f, 0)) for f in KNOB_FIELDS])
    _check(lib().dcs_probe_set_knobs(c_void_p(gen._h), ctypes.cast(k, c_void_p)), "dcs_probe_set_knobs")
