"""Object wrapper over the hot-path entry points of the C-ABI.

``SteeringCoefficientGenerator`` owns one ``dcs_bf_context`` (device delay
table, double-buffered) for a fixed shape; output buffers belong to the caller
(a :class:`dc_sand_amd.device.DeviceAllocation`, a raw device pointer, or a
torch CUDA tensor's ``data_ptr()``).
"""
from __future__ import annotations

import ctypes
from ctypes import byref, c_float, c_size_t, c_void_p

import numpy as np

from . import _lib
from ._lib import B32, MULTIPLE_CHANNELS_AND_TIMESTAMPS, check
from .device import _s
from .parameters import BeamformerParameters, delay_vals_dtype


def delta_times(params: BeamformerParameters, t0: int, nt: int) -> np.ndarray:
    """fDeltaTime of the reference verifier for time indices [t0, t0+nt)
    (``BeamformerCoefficientTest.cu:299`` then ``:12-18``).  Host only."""
    out = np.empty(nt, dtype=np.float32)
    cp = params.to_c()
    check(
        _lib.lib().dcs_bf_delta_times(byref(cp), int(t0), int(nt), out.ctypes.data_as(ctypes.POINTER(c_float))),
        "dcs_bf_delta_times",
    )
    return out


def ts_diff(first, last) -> np.float32:
    """``ts_diff`` of the reference verifier (``BeamformerCoefficientTest.cu:12-18``); ``first`` / ``last`` are
    ``(tv_sec, tv_nsec)`` pairs.  Host only."""
    a, b = _lib.Timespec(int(first[0]), int(first[1])), _lib.Timespec(int(last[0]), int(last[1]))
    out = c_float(0.0)
    check(_lib.lib().dcs_bf_ts_diff(byref(a), byref(b), byref(out)), "dcs_bf_ts_diff")
    return np.float32(out.value)


def _dt_array(dt) -> np.ndarray:
    return np.ascontiguousarray(np.atleast_1d(np.asarray(dt, dtype=np.float32)))


def _timespecs(times):
    arr = (_lib.Timespec * len(times))()
    for i, (sec, nsec) in enumerate(times):
        arr[i].tv_sec, arr[i].tv_nsec = int(sec), int(nsec)
    return arr


def simulate_input(params: BeamformerParameters) -> np.ndarray:
    """The reference's linear-ramp table (``BeamformerCoefficientTest.cu:185-196``)."""
    out = np.empty(params.n_pairs, dtype=delay_vals_dtype)
    cp = params.to_c()
    check(_lib.lib().dcs_bf_simulate_input(byref(cp), c_void_p(out.ctypes.data)), "dcs_bf_simulate_input")
    return out


def output_bytes(params: BeamformerParameters, bitwidth: int, nt: int) -> int:
    n = c_size_t(0)
    cp = params.to_c()
    check(_lib.lib().dcs_bf_output_bytes(byref(cp), int(bitwidth), int(nt), byref(n)), "dcs_bf_output_bytes")
    return int(n.value)


def gpu_utilisation(params: BeamformerParameters, kernel_ms: float) -> tuple[float, float]:
    """``BeamformerCoeffTest::get_time`` model (``BeamformerCoefficientTest.cu:426-448``)."""
    out = (c_float * 2)()
    cp = params.to_c()
    check(_lib.lib().dcs_bf_gpu_utilisation(byref(cp), float(kernel_ms), out), "dcs_bf_gpu_utilisation")
    return float(out[0]), float(out[1])


class SteeringCoefficientGenerator:
    def __init__(self, params: BeamformerParameters):
        self.params = params
        self._cp = params.to_c()
        h = c_void_p()
        check(_lib.lib().dcs_bf_create(byref(self._cp), byref(h)), "dcs_bf_create")
        self._h = h.value
        self._host_table = None  # keeps the last uploaded table alive during the async copy

    # -- delay table ------------------------------------------------------
    def upload_delays(self, table: np.ndarray, stream=None) -> None:
        """transfer_HtoD (``BeamformerCoefficientTest.cu:207-216``)."""
        table = np.ascontiguousarray(table)
        if table.dtype != delay_vals_dtype or table.size != self.params.n_pairs:
            raise ValueError(f"delay table must be {self.params.n_pairs} x delay_vals_dtype")
        self._host_table = table
        check(_lib.lib().dcs_bf_upload_delays(c_void_p(self._h), c_void_p(table.ctypes.data), _s(stream)), "dcs_bf_upload_delays")

    def set_delays_from_global(self, d_global_table: int, nr_beams_total: int, beam_offset: int, stream=None) -> None:
        """Take beams [beam_offset, beam_offset+NR_BEAMS) of a device-resident
        global table [NR_STATIONS][nr_beams_total] (multi-GPU beam sharding)."""
        check(
            _lib.lib().dcs_bf_set_delays_from_global(
                c_void_p(self._h), c_void_p(int(d_global_table)), int(nr_beams_total), int(beam_offset), _s(stream)
            ),
            "dcs_bf_set_delays_from_global",
        )

    # -- generation -------------------------------------------------------
    def generate(self, d_out, out_bytes: int, t0: int = 0, nt: int = 1, kernel: int = MULTIPLE_CHANNELS_AND_TIMESTAMPS,
                 bitwidth: int = B32, stream=None) -> None:
        """run_kernel (``BeamformerCoefficientTest.cu:218-264``)."""
        check(
            _lib.lib().dcs_bf_generate(c_void_p(self._h), int(kernel), int(bitwidth), int(t0), int(nt), c_void_p(int(d_out)),
                                       int(out_bytes), _s(stream)),
            "dcs_bf_generate",
        )

    def generate_slab(self, d_out, out_bytes: int, c0: int, nc: int, t0: int = 0, nt: int = 1, bitwidth: int = B32,
                      stream=None) -> None:
        check(
            _lib.lib().dcs_bf_generate_slab(c_void_p(self._h), int(bitwidth), int(t0), int(nt), int(c0), int(nc),
                                            c_void_p(int(d_out)), int(out_bytes), _s(stream)),
            "dcs_bf_generate_slab",
        )

    def generate_dt(self, d_out, out_bytes: int, dt, kernel: int = MULTIPLE_CHANNELS_AND_TIMESTAMPS, bitwidth: int = B32,
                    stream=None) -> None:
        """As :meth:`generate` with fDeltaTime of every time step given by value (``dcs_bf_generate_dt``)."""
        a = _dt_array(dt)
        check(
            _lib.lib().dcs_bf_generate_dt(c_void_p(self._h), int(kernel), int(bitwidth), a.ctypes.data_as(ctypes.POINTER(c_float)),
                                          a.size, c_void_p(int(d_out)), int(out_bytes), _s(stream)),
            "dcs_bf_generate_dt",
        )

    def generate_slab_dt(self, d_out, out_bytes: int, c0: int, nc: int, dt, bitwidth: int = B32, stream=None) -> None:
        a = _dt_array(dt)
        check(
            _lib.lib().dcs_bf_generate_slab_dt(c_void_p(self._h), int(bitwidth), a.ctypes.data_as(ctypes.POINTER(c_float)), a.size,
                                               int(c0), int(nc), c_void_p(int(d_out)), int(out_bytes), _s(stream)),
            "dcs_bf_generate_slab_dt",
        )

    def generate_at(self, d_out, out_bytes: int, current_times, reference_time, kernel: int = MULTIPLE_CHANNELS_AND_TIMESTAMPS,
                    bitwidth: int = B32, stream=None) -> None:
        """The reference kernels' own time arguments (``struct timespec sCurrentTime, sRefTime``,
        ``BeamformerKernels.cuh:38-42``): one ``(tv_sec, tv_nsec)`` per time step and one reference."""
        cur = _timespecs(list(current_times))
        ref = _lib.Timespec(int(reference_time[0]), int(reference_time[1]))
        check(
            _lib.lib().dcs_bf_generate_at(c_void_p(self._h), int(kernel), int(bitwidth), cur, byref(ref), len(cur),
                                          c_void_p(int(d_out)), int(out_bytes), _s(stream)),
            "dcs_bf_generate_at",
        )

    def generate_and_beamform_dt(self, d_antenna, antenna_bytes: int, d_beams, beams_bytes: int, dt, stream=None) -> None:
        a = _dt_array(dt)
        check(
            _lib.lib().dcs_bf_generate_and_beamform_dt(c_void_p(self._h), a.ctypes.data_as(ctypes.POINTER(c_float)), a.size,
                                                       c_void_p(int(d_antenna)), int(antenna_bytes), c_void_p(int(d_beams)),
                                                       int(beams_bytes), _s(stream)),
            "dcs_bf_generate_and_beamform_dt",
        )

    def generate_and_beamform(self, d_antenna, antenna_bytes: int, d_beams, beams_bytes: int, t0: int = 0,
                              nt: int | None = None, stream=None) -> None:
        """Fused coefficient generation + beamforming (run_kernel's COMBINED branch,
        ``BeamformerCoefficientTest.cu:259-262``); the table is indexed [beam*A + antenna]."""
        nt = self.params.NR_SAMPLES_PER_CHANNEL if nt is None else nt
        check(
            _lib.lib().dcs_bf_generate_and_beamform(c_void_p(self._h), int(t0), int(nt), c_void_p(int(d_antenna)), int(antenna_bytes),
                                                    c_void_p(int(d_beams)), int(beams_bytes), _s(stream)),
            "dcs_bf_generate_and_beamform",
        )

    def beamform_accumulated(self, d_antenna, antenna_bytes: int, d_beams, beams_bytes: int, nt: int, t_coeff: int | None = None,
                             dt_coeff: float | None = None, stream=None) -> None:
        """Beamformer with coefficient reuse on the matrix cores (``dcs_bf_beamform_accumulated``): the coefficients of
        ONE time (time index ``t_coeff`` or fDeltaTime ``dt_coeff``) applied to ``nt`` samples; table indexed [beam*A + antenna]."""
        if (t_coeff is None) == (dt_coeff is None):
            raise ValueError("give exactly one of t_coeff / dt_coeff")
        if dt_coeff is None:
            check(
                _lib.lib().dcs_bf_beamform_accumulated(c_void_p(self._h), int(t_coeff), int(nt), c_void_p(int(d_antenna)), int(antenna_bytes),
                                                       c_void_p(int(d_beams)), int(beams_bytes), _s(stream)),
                "dcs_bf_beamform_accumulated",
            )
        else:
            check(
                _lib.lib().dcs_bf_beamform_accumulated_dt(c_void_p(self._h), float(np.float32(dt_coeff)), int(nt), c_void_p(int(d_antenna)),
                                                          int(antenna_bytes), c_void_p(int(d_beams)), int(beams_bytes), _s(stream)),
                "dcs_bf_beamform_accumulated_dt",
            )

    TUNING_FIELDS = ("form", "nontemporal", "chan_per_block", "tiles_per_block", "waves_per_block", "rows_per_wave",
                     "xcd_remap", "rows_same_tile", "math_mode", "wg_per_cu")
    _TUNING_DEFAULTS = (0, -1, 0, 0, 0, 0, -1, -1, 0, 0)

    def set_tuning(self, form: int = 0, nontemporal: int = -1, chan_per_block: int = 0, tiles_per_block: int = 0,
                   waves_per_block: int = 0, rows_per_wave: int = 0, xcd_remap: int = -1, rows_same_tile: int = -1,
                   math_mode: int = 0, wg_per_cu: int = 0) -> None:
        """``struct dcs_bf_tuning`` (ABI 3: ten ``int32_t``); ``set_tuning()`` restores the defaults.  The measurement
        knobs of ABI 2 (``probe_nomath`` / ``probe_pace``) are ``probes.dcs_probes.set_knobs`` of the probes build now."""
        vals = (form, nontemporal, chan_per_block, tiles_per_block, waves_per_block, rows_per_wave,
                xcd_remap, rows_same_tile, math_mode, wg_per_cu)
        if vals == self._TUNING_DEFAULTS:  # all defaults: NULL, which also forgets dcs_bf_autotune's result
            check(_lib.lib().dcs_bf_set_tuning(c_void_p(self._h), c_void_p(None)), "dcs_bf_set_tuning")
            return
        t = (ctypes.c_int32 * len(self.TUNING_FIELDS))(*vals)
        check(_lib.lib().dcs_bf_set_tuning(c_void_p(self._h), ctypes.cast(t, c_void_p)), "dcs_bf_set_tuning")

    def autotune(self, d_out, out_bytes: int, bitwidth: int = B32, stream=None) -> dict:
        """``dcs_bf_autotune``: time the tiled form's geometries on this device for this
        shape, keep the fastest for this context, return the chosen knobs."""
        t = (ctypes.c_int32 * len(self.TUNING_FIELDS))()
        check(
            _lib.lib().dcs_bf_autotune(c_void_p(self._h), int(bitwidth), c_void_p(int(d_out)), int(out_bytes), _s(stream),
                                       ctypes.cast(t, c_void_p)),
            "dcs_bf_autotune",
        )
        return dict(zip(self.TUNING_FIELDS, (int(v) for v in t)))

    def output_bytes(self, bitwidth: int = B32, nt: int = 1) -> int:
        return output_bytes(self.params, bitwidth, nt)

    # -- streaming (config 5) ---------------------------------------------
    def stream_begin(self, d_out, out_bytes: int, c0: int, nc: int, stream, bitwidth: int = B32) -> "CoefficientStream":
        h = c_void_p()
        check(
            _lib.lib().dcs_bf_stream_begin(c_void_p(self._h), int(bitwidth), int(c0), int(nc), c_void_p(int(d_out)), int(out_bytes),
                                           _s(stream), byref(h)),
            "dcs_bf_stream_begin",
        )
        return CoefficientStream(self, h.value)

    def close(self) -> None:
        if self._h:
            _lib.lib().dcs_bf_destroy(c_void_p(self._h))
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CoefficientStream:
    """hipGraph replay of one time step per tick (BASELINE config 5)."""

    def __init__(self, gen: SteeringCoefficientGenerator, handle: int):
        self._gen = gen
        self._h = handle

    def tick(self, t: int, new_table: np.ndarray | None = None) -> None:
        ptr = c_void_p(None)
        if new_table is not None:
            new_table = np.ascontiguousarray(new_table)
            if new_table.dtype != delay_vals_dtype or new_table.size != self._gen.params.n_pairs:
                raise ValueError("bad delay table")
            ptr = c_void_p(new_table.ctypes.data)
        check(_lib.lib().dcs_bf_stream_tick(c_void_p(self._h), int(t), ptr), "dcs_bf_stream_tick")

    def _table_ptr(self, new_table):
        if new_table is None:
            return c_void_p(None), None
        new_table = np.ascontiguousarray(new_table)
        if new_table.dtype != delay_vals_dtype or new_table.size != self._gen.params.n_pairs:
            raise ValueError("bad delay table")
        return c_void_p(new_table.ctypes.data), new_table

    def tick_dt(self, dt: float, new_table: np.ndarray | None = None) -> None:
        """A tick at model time ``dt`` seconds after the reference time (``dcs_bf_stream_tick_dt``)."""
        ptr, keep = self._table_ptr(new_table)
        check(_lib.lib().dcs_bf_stream_tick_dt(c_void_p(self._h), float(np.float32(dt)), ptr), "dcs_bf_stream_tick_dt")

    def tick_at(self, current_time, reference_time, new_table: np.ndarray | None = None) -> None:
        ptr, keep = self._table_ptr(new_table)
        cur = _lib.Timespec(int(current_time[0]), int(current_time[1]))
        ref = _lib.Timespec(int(reference_time[0]), int(reference_time[1]))
        check(_lib.lib().dcs_bf_stream_tick_at(c_void_p(self._h), byref(cur), byref(ref), ptr), "dcs_bf_stream_tick_at")

    # -- the same ticks with the new table already on the device (a global [NR_STATIONS][nr_beams_total] table, e.g.
    #    just broadcast by RCCL; this context owns beams [beam_offset, beam_offset + NR_BEAMS)): gathered by a node of
    #    the replayed graph, no host staging (``dcs_bf_stream_tick_*_from_global``)
    def _global_args(self, d_global_table, nr_beams_total, beam_offset):
        nb = self._gen.params.NR_BEAMS if nr_beams_total is None else nr_beams_total
        return c_void_p(int(d_global_table)), int(nb), int(beam_offset)

    def tick_from_global(self, t: int, d_global_table, nr_beams_total: int | None = None, beam_offset: int = 0) -> None:
        check(_lib.lib().dcs_bf_stream_tick_from_global(c_void_p(self._h), int(t), *self._global_args(d_global_table, nr_beams_total, beam_offset)),
              "dcs_bf_stream_tick_from_global")

    def tick_dt_from_global(self, dt: float, d_global_table, nr_beams_total: int | None = None, beam_offset: int = 0) -> None:
        check(_lib.lib().dcs_bf_stream_tick_dt_from_global(c_void_p(self._h), float(np.float32(dt)),
                                                           *self._global_args(d_global_table, nr_beams_total, beam_offset)),
              "dcs_bf_stream_tick_dt_from_global")

    def tick_at_from_global(self, current_time, reference_time, d_global_table, nr_beams_total: int | None = None,
                            beam_offset: int = 0) -> None:
        cur = _lib.Timespec(int(current_time[0]), int(current_time[1]))
        ref = _lib.Timespec(int(reference_time[0]), int(reference_time[1]))
        check(_lib.lib().dcs_bf_stream_tick_at_from_global(c_void_p(self._h), byref(cur), byref(ref),
                                                           *self._global_args(d_global_table, nr_beams_total, beam_offset)),
              "dcs_bf_stream_tick_at_from_global")

    def end(self) -> None:
        if self._h:
            _lib.lib().dcs_bf_stream_end(c_void_p(self._h))
            self._h = None

    def __del__(self):
        try:
            self.end()
        except Exception:
            pass
