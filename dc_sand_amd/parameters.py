"""Run-time mirror of ``beamformer_coefficient_generator/BeamformerParameters.h``.

The reference fixes every size at compile time (``BeamformerParameters.h:4-51``);
here the same names are fields of a dataclass whose defaults are the header's
values, so the reference configuration is ``BeamformerParameters()``.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import numpy as np

#: ``struct delay_vals`` (``BeamformerParameters.h:61-66``): 4 x fp32, 16 bytes,
#: one per (antenna, beam), indexed ``[antenna * NR_BEAMS + beam]``.
delay_vals_dtype = np.dtype(
    [
        ("fDelay_s", "<f4"),
        ("fDelayRate_sps", "<f4"),
        ("fPhase_rad", "<f4"),
        ("fPhaseRate_radps", "<f4"),
    ],
    align=False,
)
assert delay_vals_dtype.itemsize == 16

COMPLEXITY = 2  # BeamformerParameters.h:4


class CParams(ctypes.Structure):
    """``struct dcs_bf_params`` (include/dcs_beamformer.h)."""

    _fields_ = [
        ("nr_channels", ctypes.c_int32),
        ("nr_stations", ctypes.c_int32),
        ("nr_beams", ctypes.c_int32),
        ("nr_samples_per_channel", ctypes.c_int32),
        ("sampling_period", ctypes.c_float),
        ("fft_size", ctypes.c_int32),
        ("adc_sample_rate", ctypes.c_double),
        ("accumulations_before_new_coeffs", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


@dataclass(frozen=True)
class BeamformerParameters:
    """Field names follow the macros of ``BeamformerParameters.h:7-17``."""

    NR_CHANNELS: int = 64
    NR_STATIONS: int = 64
    NR_BEAMS: int = 16
    NR_SAMPLES_PER_CHANNEL: int = 256
    SAMPLING_PERIOD: float = 1e-7  # 1e-7f: rounded to fp32 at the C boundary
    FFT_SIZE: int = 8192
    ADC_SAMPLE_RATE: float = 1712e6
    ACCUMULATIONS_BEFORE_NEW_COEFFS: int = 256
    NR_POLARIZATIONS: int = 1

    @property
    def n_pairs(self) -> int:
        return self.NR_STATIONS * self.NR_BEAMS

    def coeffs_per_time_step(self) -> int:
        return self.NR_CHANNELS * self.n_pairs

    def output_shape(self, nt: int, bitwidth_b16: bool = False) -> tuple:
        """[t][c][a][b][re,im] (``BeamformerCoefficientTest.cu:331-333``)."""
        return (nt, self.NR_CHANNELS, self.NR_STATIONS, self.NR_BEAMS, COMPLEXITY)

    def with_beams(self, nr_beams: int) -> "BeamformerParameters":
        return BeamformerParameters(**{**self.__dict__, "NR_BEAMS": int(nr_beams)})

    def to_c(self) -> CParams:
        return CParams(
            self.NR_CHANNELS,
            self.NR_STATIONS,
            self.NR_BEAMS,
            self.NR_SAMPLES_PER_CHANNEL,
            float(np.float32(self.SAMPLING_PERIOD)),
            self.FFT_SIZE,
            self.ADC_SAMPLE_RATE,
            self.ACCUMULATIONS_BEFORE_NEW_COEFFS,
            0,
        )
