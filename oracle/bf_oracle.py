"""ctypes loader of ``oracle/libbf_oracle.so`` (the C restatement of the
reference's CPU verifier) plus an independent numpy restatement of the same
lines, used to cross-check the C one.  TEST INFRASTRUCTURE ONLY.

PARITY STATUS: "parity unpinned" -- the reference stores no golden vectors for
this path and cannot be built here (no nvcc, no libcudart); see bf_oracle.h.
"""
from __future__ import annotations

import ctypes
import subprocess
from ctypes import POINTER, byref, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint16, c_uint32, c_uint64, c_void_p
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "libbf_oracle.so"

delay_vals_dtype = np.dtype(
    [("fDelay_s", "<f4"), ("fDelayRate_sps", "<f4"), ("fPhase_rad", "<f4"), ("fPhaseRate_radps", "<f4")]
)


class OracleParams(ctypes.Structure):
    _fields_ = [
        ("nr_channels", c_int32),
        ("nr_stations", c_int32),
        ("nr_beams", c_int32),
        ("sampling_period", c_float),
        ("fft_size", c_int32),
    ]


class Timespec(ctypes.Structure):
    _fields_ = [("tv_sec", ctypes.c_long), ("tv_nsec", ctypes.c_long)]


def build(force: bool = False) -> Path:
    src = [HERE / "bf_oracle.c", HERE / "bf_oracle.h", HERE / "Makefile"]
    if force or not LIB_PATH.exists() or any(s.stat().st_mtime > LIB_PATH.stat().st_mtime for s in src):
        res = subprocess.run(["make", "-C", str(HERE), "-B", "libbf_oracle.so"], capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"oracle build failed:\n{res.stdout}\n{res.stderr}")
    return LIB_PATH


_LIB = None


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        if not LIB_PATH.exists():
            build()
        L = ctypes.CDLL(str(LIB_PATH))
        P = POINTER(OracleParams)
        L.dcs_oracle_default_params.argtypes = [P]
        L.dcs_oracle_ts_diff.argtypes = [Timespec, Timespec]
        L.dcs_oracle_ts_diff.restype = c_float
        L.dcs_oracle_time_step_ns.argtypes = [P, c_size_t]
        L.dcs_oracle_time_step_ns.restype = ctypes.c_long
        L.dcs_oracle_time_step_ns_launch_loop.argtypes = [P, c_size_t]
        L.dcs_oracle_time_step_ns_launch_loop.restype = ctypes.c_long
        L.dcs_oracle_delta_time.argtypes = [P, c_size_t, Timespec]
        L.dcs_oracle_delta_time.restype = c_float
        L.dcs_oracle_simulate_input.argtypes = [P, c_void_p]
        L.dcs_oracle_generate.argtypes = [P, c_void_p, c_size_t, c_size_t, c_size_t, c_size_t, c_void_p]
        L.dcs_oracle_generate.restype = c_double
        L.dcs_oracle_generate_checksum.argtypes = [P, c_void_p, c_size_t, c_size_t, c_size_t, c_size_t, c_int, POINTER(c_uint64)]
        L.dcs_oracle_generate_checksum.restype = c_double
        L.dcs_oracle_generate_dt.argtypes = [P, c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p]
        L.dcs_oracle_generate_at.argtypes = [P, c_void_p, c_void_p, Timespec, c_size_t, c_size_t, c_size_t, c_void_p]
        L.dcs_oracle_beamform_dt.argtypes = [P, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]
        L.dcs_oracle_beamform_accumulated.argtypes = [P, c_void_p, c_float, c_size_t, c_void_p, c_void_p]
        L.dcs_oracle_beamform_accumulated_slab.argtypes = [P, c_void_p, c_float, c_size_t, c_size_t, c_size_t, c_void_p, c_void_p]
        L.dcs_oracle_beamform_accumulated_slab.restype = None
        L.dcs_oracle_beamform_slab.argtypes = [P, c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p, c_void_p]
        L.dcs_oracle_beamform_slab.restype = None
        L.dcs_oracle_compare_generated.argtypes = [P, c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p, c_int, c_int,
                                                   POINTER(c_uint64 * 4), POINTER(c_uint32), POINTER(c_int64)]
        L.dcs_oracle_compare_generated.restype = c_double
        L.dcs_oracle_compare_generated_f16.argtypes = L.dcs_oracle_compare_generated.argtypes
        L.dcs_oracle_compare_generated_f16.restype = c_double
        L.dcs_oracle_compare.argtypes = [c_void_p, c_void_p, c_size_t, c_float]
        L.dcs_oracle_compare.restype = c_int64
        L.dcs_oracle_ulp_diff.argtypes = [c_float, c_float]
        L.dcs_oracle_ulp_diff.restype = c_uint32
        L.dcs_oracle_max_ulp.argtypes = [c_void_p, c_void_p, c_size_t, c_uint32, POINTER(c_uint64), POINTER(c_int64)]
        L.dcs_oracle_max_ulp.restype = c_uint32
        L.dcs_oracle_device_variant_a3.argtypes = [P, c_void_p, c_size_t, c_size_t, c_void_p]
        L.dcs_oracle_simulate_antenna_data.argtypes = [c_void_p, c_size_t]
        L.dcs_oracle_beamform.argtypes = [P, c_void_p, c_size_t, c_void_p, c_void_p]
        L.dcs_oracle_set_trig_reading.argtypes = [c_int]
        L.dcs_oracle_get_trig_reading.restype = c_int
        L.dcs_oracle_f32_to_f16_rn.argtypes = [c_float]
        L.dcs_oracle_f32_to_f16_rn.restype = c_uint16
        _LIB = L
    return _LIB


def params(nr_channels=64, nr_stations=64, nr_beams=16, sampling_period=1e-7, fft_size=8192) -> OracleParams:
    return OracleParams(int(nr_channels), int(nr_stations), int(nr_beams), float(np.float32(sampling_period)), int(fft_size))


def params_from(bp) -> OracleParams:
    """From a dc_sand_amd.BeamformerParameters-like object (duck-typed)."""
    return params(bp.NR_CHANNELS, bp.NR_STATIONS, bp.NR_BEAMS, bp.SAMPLING_PERIOD, bp.FFT_SIZE)


def simulate_input(p: OracleParams) -> np.ndarray:
    out = np.empty(p.nr_stations * p.nr_beams, dtype=delay_vals_dtype)
    lib().dcs_oracle_simulate_input(byref(p), c_void_p(out.ctypes.data))
    return out


def delta_time(p: OracleParams, t: int, ref=(0, 0)) -> np.float32:
    return np.float32(lib().dcs_oracle_delta_time(byref(p), int(t), Timespec(*ref)))


def time_step_ns(p: OracleParams, t: int) -> int:
    return int(lib().dcs_oracle_time_step_ns(byref(p), int(t)))


def time_step_ns_launch_loop(p: OracleParams, t: int) -> int:
    return int(lib().dcs_oracle_time_step_ns_launch_loop(byref(p), int(t)))


DOUBLE_THEN_ROUND, FLOAT_LIBM = 0, 1


class trig_reading:
    """``with trig_reading(FLOAT_LIBM): ...`` -- evaluate the verifier's ``cos(fRotation)`` as the host
    libm's ``cosf`` (what nvcc's headers bind it to) instead of ``(float)cos((double)x)`` (the default,
    canonical reading of this repo's fixtures).  See the header comment of bf_oracle.c."""

    def __init__(self, reading: int):
        self.reading = int(reading)

    def __enter__(self):
        self.prev = lib().dcs_oracle_get_trig_reading()
        lib().dcs_oracle_set_trig_reading(self.reading)
        return self

    def __exit__(self, *exc):
        lib().dcs_oracle_set_trig_reading(self.prev)
        return False


def generate(p: OracleParams, delays: np.ndarray, t0=0, nt=1, c0=0, nc=None) -> np.ndarray:
    """Expected coefficients [nt][nc][A][B][2] (fp32)."""
    nc = p.nr_channels - c0 if nc is None else nc
    delays = np.ascontiguousarray(delays, dtype=delay_vals_dtype)
    assert delays.size == p.nr_stations * p.nr_beams
    out = np.empty((nt, nc, p.nr_stations, p.nr_beams, 2), dtype=np.float32)
    lib().dcs_oracle_generate(byref(p), c_void_p(delays.ctypes.data), t0, nt, c0, nc, c_void_p(out.ctypes.data))
    return out


def ts_diff(first, last) -> np.float32:
    """BeamformerCoefficientTest.cu:12-18; (tv_sec, tv_nsec) pairs."""
    return np.float32(lib().dcs_oracle_ts_diff(Timespec(*first), Timespec(*last)))


def generate_dt(p: OracleParams, delays: np.ndarray, dt, c0=0, nc=None) -> np.ndarray:
    """Expected coefficients [nt][nc][A][B][2] for fDeltaTime values given by the caller."""
    nc = p.nr_channels - c0 if nc is None else nc
    dt = np.ascontiguousarray(np.atleast_1d(np.asarray(dt, dtype=np.float32)))
    delays = np.ascontiguousarray(delays, dtype=delay_vals_dtype)
    assert delays.size == p.nr_stations * p.nr_beams
    out = np.empty((dt.size, nc, p.nr_stations, p.nr_beams, 2), dtype=np.float32)
    lib().dcs_oracle_generate_dt(byref(p), c_void_p(delays.ctypes.data), c_void_p(dt.ctypes.data), dt.size, c0, nc, c_void_p(out.ctypes.data))
    return out


def generate_at(p: OracleParams, delays: np.ndarray, current_times, reference_time, c0=0, nc=None) -> np.ndarray:
    """Expected coefficients for (current, reference) times as the reference kernels take them."""
    nc = p.nr_channels - c0 if nc is None else nc
    cur = (Timespec * len(current_times))(*[Timespec(int(s), int(ns)) for s, ns in current_times])
    delays = np.ascontiguousarray(delays, dtype=delay_vals_dtype)
    out = np.empty((len(current_times), nc, p.nr_stations, p.nr_beams, 2), dtype=np.float32)
    lib().dcs_oracle_generate_at(byref(p), c_void_p(delays.ctypes.data), cur, Timespec(int(reference_time[0]), int(reference_time[1])),
                                 len(current_times), c0, nc, c_void_p(out.ctypes.data))
    return out


def compare_generated(p: OracleParams, delays: np.ndarray, dt, c0: int, nc: int, got: np.ndarray, nthreads: int = 1, reading: int = 0):
    """Every element of ``got`` ([nt][nc][A][B][2] fp32 -- or uint16 binary16 patterns -- C-contiguous) against
    the verifier, generated on the fly over ``nthreads`` threads.  Returns dict(hist=[n0, n1, n2, n_more],
    max_ulp, first_over_1ulp, seconds)."""
    dt = np.ascontiguousarray(np.atleast_1d(np.asarray(dt, dtype=np.float32)))
    delays = np.ascontiguousarray(delays, dtype=delay_vals_dtype)
    assert got.dtype in (np.float32, np.uint16) and got.flags["C_CONTIGUOUS"]
    assert got.size == dt.size * nc * p.nr_stations * p.nr_beams * 2
    hist = (c_uint64 * 4)()
    mx = c_uint32(0)
    first = c_int64(-1)
    # uint16: the packed binary16 output; expectation RN-even(verifier's fp32), distances in binary16 ulps
    fn = lib().dcs_oracle_compare_generated if got.dtype == np.float32 else lib().dcs_oracle_compare_generated_f16
    secs = fn(byref(p), c_void_p(delays.ctypes.data), c_void_p(dt.ctypes.data), dt.size, int(c0), int(nc),
                                              c_void_p(got.ctypes.data), int(nthreads), int(reading), byref(hist), byref(mx), byref(first))
    return dict(hist=[int(v) for v in hist], max_ulp=int(mx.value), first_over_1ulp=int(first.value), seconds=float(secs))


def generate_checksum(p: OracleParams, delays: np.ndarray, t0=0, nt=1, c0=0, nc=None, nthreads=1):
    """(seconds, checksum): sum of the fp32 bit patterns mod 2^64."""
    nc = p.nr_channels - c0 if nc is None else nc
    delays = np.ascontiguousarray(delays, dtype=delay_vals_dtype)
    cks = c_uint64(0)
    secs = lib().dcs_oracle_generate_checksum(byref(p), c_void_p(delays.ctypes.data), t0, nt, c0, nc, int(nthreads), byref(cks))
    return float(secs), int(cks.value)


def checksum_of(arr: np.ndarray) -> int:
    """The same checksum over an fp32 array (for comparing a device tensor)."""
    return int(np.ascontiguousarray(arr, dtype=np.float32).view(np.uint32).astype(np.uint64).sum(dtype=np.uint64))


def compare(got: np.ndarray, expect: np.ndarray, tol: float) -> int:
    """First flat index with |got-expect| > tol, or -1 (BeamformerCoefficientTest.cu:348-357)."""
    got = np.ascontiguousarray(got, dtype=np.float32)
    expect = np.ascontiguousarray(expect, dtype=np.float32)
    assert got.size == expect.size
    return int(lib().dcs_oracle_compare(c_void_p(got.ctypes.data), c_void_p(expect.ctypes.data), got.size, float(tol)))


def max_ulp(got: np.ndarray, expect: np.ndarray, limit: int = 1):
    """(max ULP distance, count over limit, first index over limit)."""
    got = np.ascontiguousarray(got, dtype=np.float32)
    expect = np.ascontiguousarray(expect, dtype=np.float32)
    assert got.size == expect.size
    n_over = c_uint64(0)
    first = c_int64(-1)
    mx = lib().dcs_oracle_max_ulp(c_void_p(got.ctypes.data), c_void_p(expect.ctypes.data), got.size, int(limit), byref(n_over), byref(first))
    return int(mx), int(n_over.value), int(first.value)


def device_variant_a3(p: OracleParams, delays: np.ndarray, t0=0, nt=1) -> np.ndarray:
    delays = np.ascontiguousarray(delays, dtype=delay_vals_dtype)
    out = np.empty((nt, p.nr_channels, p.nr_stations, p.nr_beams, 2), dtype=np.float32)
    lib().dcs_oracle_device_variant_a3(byref(p), c_void_p(delays.ctypes.data), t0, nt, c_void_p(out.ctypes.data))
    return out


def simulate_antenna_data(p: OracleParams, nt: int) -> np.ndarray:
    """int8 [chan][nt/16][station][16][2], byte i = (int8)i (BeamformerCoefficientTest.cu:198-204)."""
    out = np.empty((p.nr_channels, nt // 16, p.nr_stations, 16, 2), dtype=np.int8)
    lib().dcs_oracle_simulate_antenna_data(c_void_p(out.ctypes.data), out.nbytes)
    return out


def beamform(p: OracleParams, delays_beam_major: np.ndarray, nt: int, antenna_data: np.ndarray) -> np.ndarray:
    """Expected beams float [chan][nt/16][beam][16][2]; delays indexed [b*A + a]."""
    assert nt % 16 == 0
    delays = np.ascontiguousarray(delays_beam_major, dtype=delay_vals_dtype)
    ant = np.ascontiguousarray(antenna_data, dtype=np.int8)
    assert ant.size == p.nr_channels * nt * p.nr_stations * 2
    out = np.empty((p.nr_channels, nt // 16, p.nr_beams, 16, 2), dtype=np.float32)
    lib().dcs_oracle_beamform(byref(p), c_void_p(delays.ctypes.data), nt, c_void_p(ant.ctypes.data), c_void_p(out.ctypes.data))
    return out


def beamform_dt(p: OracleParams, delays_beam_major: np.ndarray, dt, antenna_data: np.ndarray) -> np.ndarray:
    dt = np.ascontiguousarray(np.atleast_1d(np.asarray(dt, dtype=np.float32)))
    nt = dt.size
    assert nt % 16 == 0
    delays = np.ascontiguousarray(delays_beam_major, dtype=delay_vals_dtype)
    ant = np.ascontiguousarray(antenna_data, dtype=np.int8)
    assert ant.size == p.nr_channels * nt * p.nr_stations * 2
    out = np.empty((p.nr_channels, nt // 16, p.nr_beams, 16, 2), dtype=np.float32)
    lib().dcs_oracle_beamform_dt(byref(p), c_void_p(delays.ctypes.data), c_void_p(dt.ctypes.data), nt, c_void_p(ant.ctypes.data), c_void_p(out.ctypes.data))
    return out


def beamform_accumulated(p: OracleParams, delays_beam_major: np.ndarray, dt_coeff: float, nt: int, antenna_data: np.ndarray) -> np.ndarray:
    """Expected beams float [chan][nt/16][beam][16][2] with the coefficient of ONE fDeltaTime held for all nt samples."""
    assert nt % 16 == 0
    delays = np.ascontiguousarray(delays_beam_major, dtype=delay_vals_dtype)
    ant = np.ascontiguousarray(antenna_data, dtype=np.int8)
    assert ant.size == p.nr_channels * nt * p.nr_stations * 2
    out = np.empty((p.nr_channels, nt // 16, p.nr_beams, 16, 2), dtype=np.float32)
    lib().dcs_oracle_beamform_accumulated(byref(p), c_void_p(delays.ctypes.data), float(np.float32(dt_coeff)), nt, c_void_p(ant.ctypes.data),
                                          c_void_p(out.ctypes.data))
    return out


def beamform_slab(p: OracleParams, delays_beam_major: np.ndarray, nt: int, c0: int, nc: int, antenna_slab: np.ndarray, dt=None) -> np.ndarray:
    """Channels [c0, c0 + nc) of :func:`beamform` (``dt`` None: time indices 0 .. nt-1) / :func:`beamform_dt`."""
    assert nt % 16 == 0
    delays = np.ascontiguousarray(delays_beam_major, dtype=delay_vals_dtype)
    ant = np.ascontiguousarray(antenna_slab, dtype=np.int8)
    assert ant.size == nc * nt * p.nr_stations * 2 and c0 + nc <= p.nr_channels
    dtp = c_void_p(None)
    if dt is not None:
        dt = np.ascontiguousarray(np.atleast_1d(np.asarray(dt, dtype=np.float32)))
        assert dt.size == nt
        dtp = c_void_p(dt.ctypes.data)
    out = np.empty((nc, nt // 16, p.nr_beams, 16, 2), dtype=np.float32)
    lib().dcs_oracle_beamform_slab(byref(p), c_void_p(delays.ctypes.data), dtp, nt, int(c0), int(nc), c_void_p(ant.ctypes.data),
                                   c_void_p(out.ctypes.data))
    return out


def beamform_accumulated_slab(p: OracleParams, delays_beam_major: np.ndarray, dt_coeff: float, nt: int, c0: int, nc: int,
                              antenna_slab: np.ndarray) -> np.ndarray:
    """Channels [c0, c0 + nc) of :func:`beamform_accumulated`; ``antenna_slab`` holds those channels only."""
    assert nt % 16 == 0
    delays = np.ascontiguousarray(delays_beam_major, dtype=delay_vals_dtype)
    ant = np.ascontiguousarray(antenna_slab, dtype=np.int8)
    assert ant.size == nc * nt * p.nr_stations * 2 and c0 + nc <= p.nr_channels
    out = np.empty((nc, nt // 16, p.nr_beams, 16, 2), dtype=np.float32)
    lib().dcs_oracle_beamform_accumulated_slab(byref(p), c_void_p(delays.ctypes.data), float(np.float32(dt_coeff)), nt, int(c0), int(nc),
                                               c_void_p(ant.ctypes.data), c_void_p(out.ctypes.data))
    return out


def f32_to_f16_bits(x: np.ndarray) -> np.ndarray:
    """RN-even binary16 bit patterns of an fp32 array (C routine, element-wise)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    f = lib().dcs_oracle_f32_to_f16_rn
    return np.fromiter((f(float(v)) for v in x.ravel()), dtype=np.uint16, count=x.size).reshape(x.shape)


# ---------------------------------------------------------------------------
# Independent numpy restatement of BeamformerCoefficientTest.cu:294-337 (typed
# op by op with np.float32 / np.float64 scalars and arrays).  Slow; small cases.
# ---------------------------------------------------------------------------
def delta_time_numpy(p: OracleParams, t: int) -> np.float32:
    f = np.float32
    step = f(f(f(f(t) * f(p.sampling_period)) * f(1e9)) * f(p.fft_size))  # :299, all fp32
    step_ns = int(step)  # truncation to long
    return f(f(0.0) + f(f(step_ns) / f(1e9)))  # ts_diff :14-16


def generate_numpy(p: OracleParams, delays: np.ndarray, t0=0, nt=1, c0=0, nc=None) -> np.ndarray:
    f, d = np.float32, np.float64
    nc = p.nr_channels - c0 if nc is None else nc
    A, B, C = p.nr_stations, p.nr_beams, p.nr_channels
    Ts = f(p.sampling_period)
    pi_f = f(np.pi)
    denom = f(Ts * f(C))  # SAMPLING_PERIOD*NR_CHANNELS (fp32)
    delay = delays["fDelay_s"].astype(f)
    rate = delays["fDelayRate_sps"].astype(f)
    phase = delays["fPhase_rad"].astype(f)
    prate = delays["fPhaseRate_radps"].astype(f)
    out = np.empty((nt, nc, A * B, 2), dtype=f)
    for ti in range(nt):
        dt = delta_time_numpy(p, t0 + ti)
        dDelay = (rate * dt).astype(f)
        k = (rate + dDelay).astype(f)
        n2 = ((delay + dDelay).astype(f).astype(d) * d(C / 2.0) * d(pi_f) / d(denom)).astype(f)
        dPhase = (prate * dt).astype(f)
        phase0 = ((phase - n2).astype(f) + dPhase).astype(f)
        for ci in range(nc):
            c = f(c0 + ci)
            delayN = ((((k * c).astype(f)) * pi_f).astype(f) / denom).astype(f)
            rot = (delayN + phase0).astype(f)
            out[ti, ci, :, 0] = np.cos(rot.astype(d)).astype(f)
            out[ti, ci, :, 1] = np.sin(rot.astype(d)).astype(f)
    return out.reshape(nt, nc, A, B, 2)
