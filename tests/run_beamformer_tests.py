#!/usr/bin/env python3
"""The `runBeamformerTests.cpp` executable, over the C-ABI: runs the reference's
test sequence (same order, tolerances and result table; `runBeamformerTests.cpp:
10-82`) on the HIP kernels with the CPU oracle as `verify_output`'s expected
data.  Needs an MI355X.  Exit code 0 = all passed, 1 = first failure.

Unlike the reference (`BeamformerCoefficientTest.cu:282-287`) the 16-bit case IS
verified (RN-even of the fp32 expectation, tolerance 1e-3).
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from dc_sand_amd.beamformer_coeff_test import BeamformerCoeffTest, SteeringCoefficientBitWidth as BW, SteeringCoefficientKernel as K  # noqa: E402
from oracle import bf_oracle  # noqa: E402  (the checker; this file is test infrastructure)


def verifier(bp, delays, nt):
    return bf_oracle.generate(bf_oracle.params_from(bp), np.asarray(delays), 0, nt)


def beam_verifier(bp, delays, nt, ant):
    return bf_oracle.beamform(bf_oracle.params_from(bp), np.asarray(delays), nt, ant)


def main() -> int:
    results = [None] * 4
    cases = [
        (3, "Combined Steering Coeffs+Beamforming", 1e-1, K.COMBINED_COEFF_GEN_AND_BEAMFORMER_SINGLE_CHANNEL, BW.b32,
         "Testing with a a kernel that generates steering coefficients and performs beamforming."),
        (2, "Multiple Chans+Timestamps", 1e-4, K.MULTIPLE_CHANNELS_AND_TIMESTAMPS, BW.b32,
         "Testing with a single thread generating multiple steering coefficients (equal to the number of channels) per antenna-beam delay value.\n"
         "A single kernel generates multiple timestamps for a limited subset of delay values"),
        (1, "Multiple Channels", 1e-3, K.MULTIPLE_CHANNELS, BW.b16,
         "Testing with a single thread generating multiple steering coefficients per antenna-beam delay value for a single timestamp (16-bit output)."),
        (0, "Naive Implementation", 1e-4, K.NAIVE, BW.b32,
         "Testing with a single thread generating a single steering coefficient per antenna-beam-channel delay value"),
    ]
    for slot, name, tol, kern, bw, banner in cases:
        print(banner)
        t = BeamformerCoeffTest(tol, kern, bw, verifier=verifier, beam_verifier=beam_verifier)
        t.run_test()
        t.get_time()
        if t.get_result() != 1:
            print("Test failed, output data not generated correctly")
            return 1
        if t.max_ulp is not None:
            print(f"max ULP distance to the CPU verifier: {t.max_ulp}")
        if t.max_abs_diff is not None:
            print(f"max |beam - CPU verifier|: {t.max_abs_diff:g}")
        results[slot] = (name, t.get_gpu_utilisation_per_single_time_unit(), t.get_gpu_utilisation_per_multiple_time_units())
    print(f"{'Kernel Name':<50}{'GPU Utilisation':<20}{'GPU Utilisation':<20}")
    print(f"{'':<50}{'(1 Time Unit)':<20}{'(Many time Units)':<20}")
    for name, a, b in results:
        print(f"{name:<50}{a:<20.6g}{b:<20.6g}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
