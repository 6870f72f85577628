#!/usr/bin/env python3
"""The reference's Python host pattern (pycuda_example/vector_add.py:14-46: allocate -> fill -> copy in ->
launch -> copy out -> verify) for the steering-coefficient path, on an MI355X through the ctypes C-ABI.

    python examples/steering_coefficients.py [ant beams chan]

The verification at the end uses the CPU oracle (test infrastructure) exactly as the reference's harness
uses its CPU verifier; the product path itself never touches it."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dc_sand_amd import BeamformerParameters  # noqa: E402
from dc_sand_amd.device import Event, mem_alloc, memcpy_dtoh, pagelocked_empty, require_device, set_device  # noqa: E402
from dc_sand_amd.generator import SteeringCoefficientGenerator, simulate_input  # noqa: E402
from dc_sand_amd.parameters import delay_vals_dtype  # noqa: E402

A, B, C = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (64, 64, 4096)
require_device()
set_device(0)
p = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B)

table = pagelocked_empty(p.n_pairs, delay_vals_dtype)        # cuda.pagelocked_empty
table[:] = simulate_input(p)                                  # the reference's ramp (BeamformerCoefficientTest.cu:185-196)
gen = SteeringCoefficientGenerator(p)
gen.upload_delays(table)                                      # cuda.memcpy_htod
nbytes = gen.output_bytes(bitwidth=1, nt=1)
d_out = mem_alloc(nbytes)                                     # cuda.mem_alloc

start, stop = Event(), Event()
gen.generate(d_out, nbytes, t0=1, nt=1)                       # first launch of the process (cold)
start.record()
gen.generate(d_out, nbytes, t0=1, nt=1)                       # the launch
stop.record()
stop.synchronize()
ms = stop.elapsed_ms_since(start)

h_out = pagelocked_empty((1, C, A, B, 2), np.float32)
memcpy_dtoh(h_out, d_out)                                     # cuda.memcpy_dtoh
print(f"{A} ant x {B} beams x {C} chan: {p.coeffs_per_time_step() / ms / 1e6:.1f} Gcoeff/s "
      f"({nbytes / ms / 1e9:.2f} TB/s written), {ms * 1e3:.1f} us")

from oracle import bf_oracle  # noqa: E402  (verification only)

exp = bf_oracle.generate(bf_oracle.params_from(p), np.asarray(table), 1, 1)
mx, n_over, _ = bf_oracle.max_ulp(np.asarray(h_out), exp, 1)
print(f"max ULP distance to the CPU verifier: {mx} ({n_over} elements over 1 ULP)")
sys.exit(0 if n_over == 0 else 1)
