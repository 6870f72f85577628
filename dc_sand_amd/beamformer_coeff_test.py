"""Python mirror of ``BeamformerCoeffTest``
(``beamformer_coefficient_generator/BeamformerCoefficientTest.{hpp,cu}``): the
host driver of the steering-coefficient kernels, in the shape of the
reference's Python example (allocate -> fill -> HtoD -> launch -> DtoH ->
verify; ``pycuda_example/vector_add.py``), over the C-ABI.

The CPU verifier of the reference lives in this repo as *test infrastructure*
(``oracle/``) and is never imported here: the expected coefficients are
supplied by the caller as ``verifier(params, delays, nt) -> float32 ndarray``
(tests pass the oracle).  Without one, :meth:`verify_output` leaves the result
at 0 ("not run").
"""
from __future__ import annotations

import enum
from typing import Callable, Optional

import numpy as np

from . import _lib
from .device import mem_alloc, memcpy_dtoh, memcpy_htod, pagelocked_empty, require_device
from .generator import SteeringCoefficientGenerator, gpu_utilisation, output_bytes, simulate_input
from .parameters import BeamformerParameters, delay_vals_dtype
from .unit_test import UnitTest


class SteeringCoefficientKernel(enum.IntEnum):
    """``BeamformerCoefficientTest.hpp:19-25``."""

    NAIVE = _lib.NAIVE
    MULTIPLE_CHANNELS = _lib.MULTIPLE_CHANNELS
    MULTIPLE_CHANNELS_AND_TIMESTAMPS = _lib.MULTIPLE_CHANNELS_AND_TIMESTAMPS
    COMBINED_COEFF_GEN_AND_BEAMFORMER_SINGLE_CHANNEL = _lib.COMBINED_COEFF_GEN_AND_BEAMFORMER_SINGLE_CHANNEL


class SteeringCoefficientBitWidth(enum.IntEnum):
    """``BeamformerCoefficientTest.hpp:33-37``."""

    b16 = _lib.B16
    b32 = _lib.B32


Verifier = Callable[[BeamformerParameters, np.ndarray, int], np.ndarray]


class BeamformerCoeffTest(UnitTest):
    def __init__(
        self,
        fFloatingPointTolerance: float,
        eKernelOption: SteeringCoefficientKernel,
        eBitWidth: SteeringCoefficientBitWidth,
        params: Optional[BeamformerParameters] = None,
        verifier: Optional[Verifier] = None,
        verbose: bool = True,
    ):
        require_device()
        super().__init__()
        self.params = params or BeamformerParameters()
        self.m_fFloatingPointTolerance = float(fFloatingPointTolerance)
        self.m_eKernelOption = SteeringCoefficientKernel(eKernelOption)
        self.m_eBitWidth = SteeringCoefficientBitWidth(eBitWidth)
        self._verifier = verifier
        self._verbose = verbose
        self.max_ulp: Optional[int] = None
        p = self.params
        K, BW = SteeringCoefficientKernel, SteeringCoefficientBitWidth

        # BeamformerCoefficientTest.cu:40-50 -- the reference prints and throws
        if self.m_eKernelOption in (K.NAIVE, K.COMBINED_COEFF_GEN_AND_BEAMFORMER_SINGLE_CHANNEL) and self.m_eBitWidth == BW.b16:
            raise ValueError("This kernel does not support 16 bit steering coefficients.")
        if self.m_eKernelOption == K.COMBINED_COEFF_GEN_AND_BEAMFORMER_SINGLE_CHANNEL:
            raise NotImplementedError(
                "COMBINED_COEFF_GEN_AND_BEAMFORMER_SINGLE_CHANNEL is outside this build's hot path (SURVEY.md section 8 f1)"
            )

        # BeamformerCoefficientTest.cu:24-38
        self.m_ulSizeDelayValues = p.n_pairs * delay_vals_dtype.itemsize
        self.m_ulSizeSteeringCoefficients = output_bytes(p, int(self.m_eBitWidth), p.NR_SAMPLES_PER_CHANNEL)
        if verbose:
            print(f"{self.m_ulSizeSteeringCoefficients / 1e6:g} MB Allocated for steering coefficients")
            print(f"{self.m_ulSizeDelayValues / 1e6:g} MB Allocated for delay values")

        # BeamformerCoefficientTest.cu:73-87 -- pinned host + device buffers
        self.m_pHDelayValues = pagelocked_empty(p.n_pairs, delay_vals_dtype)
        host_dtype = np.float16 if self.m_eBitWidth == BW.b16 else np.float32
        self.m_pfHSteeringCoeffs = pagelocked_empty(self.m_ulSizeSteeringCoefficients // np.dtype(host_dtype).itemsize, host_dtype)
        self.m_pfDSteeringCoeffs = mem_alloc(self.m_ulSizeSteeringCoefficients)
        self._gen = SteeringCoefficientGenerator(p)  # owns m_pDDelayValues
        self.m_fGpuUtilisation_SingleTimeUnit = 0.0
        self.m_fGpuUtilisation_MultipleTimeUnits = 0.0

    # -- BeamformerCoefficientTest.cu:185-196 ---------------------------------
    def simulate_input(self) -> None:
        self.m_pHDelayValues[:] = simulate_input(self.params)

    # -- BeamformerCoefficientTest.cu:207-216 ---------------------------------
    def transfer_HtoD(self) -> None:
        self._gen.upload_delays(self.m_pHDelayValues)

    # -- BeamformerCoefficientTest.cu:218-264 ---------------------------------
    def run_kernel(self) -> None:
        self._gen.generate(
            self.m_pfDSteeringCoeffs,
            self.m_ulSizeSteeringCoefficients,
            t0=0,
            nt=self.params.NR_SAMPLES_PER_CHANNEL,
            kernel=int(self.m_eKernelOption),
            bitwidth=int(self.m_eBitWidth),
        )

    # -- BeamformerCoefficientTest.cu:266-276 ---------------------------------
    def transfer_DtoH(self) -> None:
        memcpy_dtoh(self.m_pfHSteeringCoeffs, self.m_pfDSteeringCoeffs)

    # -- BeamformerCoefficientTest.cu:278-361 ---------------------------------
    def verify_output(self) -> None:
        if self._verifier is None:
            if self._verbose:
                print("No verifier supplied - result left at 0 (not run)")
            return
        expect = np.asarray(self._verifier(self.params, self.m_pHDelayValues, self.params.NR_SAMPLES_PER_CHANNEL))
        got = self.m_pfHSteeringCoeffs
        if self.m_eBitWidth == SteeringCoefficientBitWidth.b16:
            # the reference skips this case and reports success (:282-287); here
            # the verifier must return the fp16 expectation
            expect = expect.astype(np.float16, copy=False).ravel()
            diff = np.abs(got.astype(np.float32) - expect.astype(np.float32))
        else:
            expect = expect.astype(np.float32, copy=False).ravel()
            diff = np.abs(got - expect)
            gi = got.view(np.int32).astype(np.int64)
            ei = expect.view(np.int32).astype(np.int64)
            gi = np.where(gi < 0, -(gi & 0x7FFFFFFF), gi)
            ei = np.where(ei < 0, -(ei & 0x7FFFFFFF), ei)
            self.max_ulp = int(np.max(np.abs(gi - ei))) if gi.size else 0
        bad = np.flatnonzero(~(diff <= self.m_fFloatingPointTolerance))  # NaN counts as a mismatch
        if bad.size:
            i = int(bad[0])
            print(f"Index: {i}. Generated Value: {got[i]}. Correct Value: {expect[i]}")
            self.m_iResult = -1
            return
        self.m_iResult = 1

    # -- BeamformerCoefficientTest.cu:422-454 ---------------------------------
    def get_time(self) -> float:
        single, multiple = gpu_utilisation(self.params, self.m_fKernelElapsedTime_ms)
        self.m_fGpuUtilisation_SingleTimeUnit = single
        self.m_fGpuUtilisation_MultipleTimeUnits = multiple
        p = self.params
        rate = np.float32(p.ADC_SAMPLE_RATE) / np.float32(p.FFT_SIZE)
        print(f"FFTs Per Second: {rate:g} Hz")
        print(f"Time to transfer {p.NR_SAMPLES_PER_CHANNEL} packets: {p.NR_SAMPLES_PER_CHANNEL / rate:g} s")
        print(f"Time to generate steering coefficients for {p.NR_SAMPLES_PER_CHANNEL} packets: {self.m_fKernelElapsedTime_ms / 1000.0:g} s\n")
        return super().get_time()

    def get_gpu_utilisation_per_single_time_unit(self) -> float:
        return self.m_fGpuUtilisation_SingleTimeUnit

    def get_gpu_utilisation_per_multiple_time_units(self) -> float:
        return self.m_fGpuUtilisation_MultipleTimeUnits
