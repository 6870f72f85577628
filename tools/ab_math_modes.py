"""Interleaved A/B of the arithmetic forms (math_mode 0 = 3-op divide + low-degree polynomials where proven,
3 = 5-op divide + full polynomials) at config 3, one process."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dc_sand_amd import BeamformerParameters, device  # noqa: E402
from dc_sand_amd.generator import SteeringCoefficientGenerator, simulate_input  # noqa: E402

device.set_device(0)
bp = BeamformerParameters(NR_CHANNELS=32768, NR_STATIONS=64, NR_BEAMS=1024)
gen = SteeringCoefficientGenerator(bp)
gen.upload_delays(simulate_input(bp))
buf = device.mem_alloc(gen.output_bytes(1, 1))
for bw in (1, 0):
    nb = gen.output_bytes(bw, 1)
    for _ in range(15):
        gen.generate(buf, nb, t0=1, nt=1, bitwidth=bw)
    res = {0: [], 1: [], 2: [], 3: []}
    for rnd in range(6):
        for mode in (0, 3, 1, 2):
            gen.set_tuning(math_mode=mode)
            ts = []
            for _ in range(12):
                e0, e1 = device.Event(), device.Event()
                e0.record()
                gen.generate(buf, nb, t0=1, nt=1, bitwidth=bw)
                e1.record()
                e1.synchronize()
                ts.append(e1.elapsed_ms_since(e0))
            res[mode].append(float(np.median(ts[2:])))
    for mode in (0, 1, 2, 3):
        m = np.median(res[mode])
        print(f"bitwidth={'fp32' if bw else 'fp16'} math_mode={mode}: median of round medians {m:.4f} ms -> {bp.coeffs_per_time_step() / m / 1e6:.1f} Gcoeff/s   rounds {[round(x, 3) for x in res[mode]]}")
