"""Device plumbing in the shape of the pycuda calls the reference's Python
example uses (``pycuda_example/vector_add.py:14-46``): ``mem_alloc``,
``pagelocked_empty``, ``memcpy_htod`` / ``memcpy_dtoh``, plus HIP events and
streams -- all through the C-ABI, no torch required.
"""
from __future__ import annotations

import ctypes
from ctypes import byref, c_float, c_int, c_void_p

import numpy as np

from . import _lib
from ._lib import check


def device_count() -> int:
    n = c_int(0)
    check(_lib.lib().dcs_device_count(byref(n)), "dcs_device_count")
    return n.value


def require_device() -> None:
    if device_count() < 1:
        raise RuntimeError("no HIP device visible: dc_sand_amd has no CPU fallback")


def set_device(i: int) -> None:
    check(_lib.lib().dcs_device_set(int(i)), "dcs_device_set")


def synchronize() -> None:
    check(_lib.lib().dcs_device_synchronize(), "dcs_device_synchronize")


def device_name(i: int = 0) -> str:
    buf = ctypes.create_string_buffer(256)
    check(_lib.lib().dcs_device_name(int(i), buf, 256), "dcs_device_name")
    return buf.value.decode()


class DeviceAllocation:
    """``cuda.mem_alloc`` (vector_add.py:15): owns ``nbytes`` of device memory."""

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        p = c_void_p()
        check(_lib.lib().dcs_malloc(byref(p), max(self.nbytes, 1)), "dcs_malloc")
        self.ptr = p.value

    def free(self) -> None:
        if self.ptr:
            _lib.lib().dcs_free(c_void_p(self.ptr))
            self.ptr = None

    def __int__(self) -> int:
        return int(self.ptr)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def mem_alloc(nbytes: int) -> DeviceAllocation:
    return DeviceAllocation(nbytes)


class _PinnedOwner:
    def __init__(self, nbytes: int):
        p = c_void_p()
        check(_lib.lib().dcs_host_alloc(byref(p), max(int(nbytes), 1)), "dcs_host_alloc")
        self.ptr = p.value

    def __del__(self):
        try:
            if self.ptr:
                _lib.lib().dcs_host_free(c_void_p(self.ptr))
                self.ptr = None
        except Exception:
            pass


def pagelocked_empty(shape, dtype) -> np.ndarray:
    """``cuda.pagelocked_empty`` (vector_add.py:14): a numpy array over pinned
    host memory (``cudaMallocHost``, BeamformerCoefficientTest.cu:73,77)."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) if np.ndim(shape) else int(shape)
    owner = _PinnedOwner(n * dtype.itemsize)
    buf = (ctypes.c_char * max(n * dtype.itemsize, 1)).from_address(owner.ptr)
    buf._dcs_owner = owner  # the allocation lives as long as any numpy view of buf
    return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)


def _host_ptr(a: np.ndarray) -> c_void_p:
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("host array must be C-contiguous")
    return c_void_p(a.ctypes.data)


def memcpy_htod(dst, src: np.ndarray, stream=None, sync: bool = True) -> None:
    """``cuda.memcpy_htod`` (vector_add.py:34)."""
    check(_lib.lib().dcs_memcpy_htod(c_void_p(int(dst)), _host_ptr(src), src.nbytes, _s(stream)), "dcs_memcpy_htod")
    if sync:
        stream_synchronize(stream)


def memcpy_dtoh(dst: np.ndarray, src, stream=None, sync: bool = True, nbytes: int | None = None) -> None:
    """``cuda.memcpy_dtoh`` (vector_add.py:46)."""
    n = dst.nbytes if nbytes is None else int(nbytes)
    check(_lib.lib().dcs_memcpy_dtoh(_host_ptr(dst), c_void_p(int(src)), n, _s(stream)), "dcs_memcpy_dtoh")
    if sync:
        stream_synchronize(stream)


def memcpy_dtod(dst, src, nbytes: int, stream=None) -> None:
    """Device-to-device copy, asynchronous on ``stream`` (``dcs_memcpy_dtod``)."""
    check(_lib.lib().dcs_memcpy_dtod(c_void_p(int(dst)), c_void_p(int(src)), int(nbytes), _s(stream)), "dcs_memcpy_dtod")


def memcpy2d_dtoh(dst: np.ndarray, dst_pitch: int, src, src_pitch: int, row_bytes: int, nrows: int, stream=None) -> None:
    check(
        _lib.lib().dcs_memcpy2d_dtoh(_host_ptr(dst), dst_pitch, c_void_p(int(src)), src_pitch, row_bytes, nrows, _s(stream)),
        "dcs_memcpy2d_dtoh",
    )
    stream_synchronize(stream)


def memset(dst, value: int, nbytes: int, stream=None) -> None:
    check(_lib.lib().dcs_memset(c_void_p(int(dst)), int(value), int(nbytes), _s(stream)), "dcs_memset")


def _s(stream) -> c_void_p:
    if stream is None:
        return c_void_p(None)
    if isinstance(stream, Stream):
        return c_void_p(stream.handle)
    return c_void_p(int(stream))


class Stream:
    def __init__(self):
        p = c_void_p()
        check(_lib.lib().dcs_stream_create(byref(p)), "dcs_stream_create")
        self.handle = p.value

    def synchronize(self) -> None:
        check(_lib.lib().dcs_stream_synchronize(c_void_p(self.handle)), "dcs_stream_synchronize")

    def __int__(self) -> int:
        return int(self.handle)

    def __del__(self):
        try:
            if self.handle:
                _lib.lib().dcs_stream_destroy(c_void_p(self.handle))
                self.handle = None
        except Exception:
            pass


def stream_synchronize(stream=None) -> None:
    check(_lib.lib().dcs_stream_synchronize(_s(stream)), "dcs_stream_synchronize")


class Event:
    """hipEvent pair-timing as ``common/UnitTest.cpp:9-14,34-53`` uses it."""

    def __init__(self):
        p = c_void_p()
        check(_lib.lib().dcs_event_create(byref(p)), "dcs_event_create")
        self.handle = p.value

    def record(self, stream=None) -> "Event":
        check(_lib.lib().dcs_event_record(c_void_p(self.handle), _s(stream)), "dcs_event_record")
        return self

    def synchronize(self) -> None:
        check(_lib.lib().dcs_event_synchronize(c_void_p(self.handle)), "dcs_event_synchronize")

    def elapsed_ms_since(self, start: "Event") -> float:
        ms = c_float(0.0)
        check(_lib.lib().dcs_event_elapsed_ms(c_void_p(start.handle), c_void_p(self.handle), byref(ms)), "dcs_event_elapsed_ms")
        return float(ms.value)

    def __del__(self):
        try:
            if self.handle:
                _lib.lib().dcs_event_destroy(c_void_p(self.handle))
                self.handle = None
        except Exception:
            pass
