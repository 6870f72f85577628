"""Steady-state timing of the fused coefficient-generation + beamforming kernel
 against materialise-then-nothing (the coefficient generator alone)."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from dc_sand_amd import BeamformerParameters, device  # noqa: E402
from dc_sand_amd.generator import SteeringCoefficientGenerator, simulate_input  # noqa: E402
from explore import timeit  # noqa: E402

device.set_device(0)
for (A, B, C, nt) in ((64, 16, 64, 256), (64, 16, 4096, 256), (64, 64, 4096, 64), (64, 256, 4096, 16), (256, 64, 1024, 64)):
    bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B, NR_SAMPLES_PER_CHANNEL=nt)
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(simulate_input(bp))
    ant_bytes = A * C * nt * 2
    beam_bytes = B * C * nt * 8
    d_ant = device.mem_alloc(ant_bytes)
    device.memset(d_ant, 3, ant_bytes)
    d_beams = device.mem_alloc(beam_bytes)
    med, mn = timeit(lambda: g.generate_and_beamform(d_ant, ant_bytes, d_beams, beam_bytes, 0, nt), warm=3, reps=9)
    prods = A * B * C * nt
    line = f"{A}ant x {B}beam x {C}chan x {nt}t: fused {med * 1e3:.1f} us -> {prods / med / 1e6:.1f} Gcoeff-products/s"
    cbytes = g.output_bytes(1, nt)
    if cbytes <= 64 * 2**30:
        d_c = device.mem_alloc(cbytes)
        med2, _ = timeit(lambda: g.generate(d_c, cbytes, 0, nt), warm=3, reps=9)
        line += f" | coefficient tensor alone ({cbytes / 2**30:.2f} GiB) {med2 * 1e3:.1f} us -> {prods / med2 / 1e6:.1f} Gcoeff/s"
        d_c.free()
    print(line, flush=True)
    g.close()
