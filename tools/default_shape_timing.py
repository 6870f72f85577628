"""The reference's default tensor (64 chan x 64 ant x 16 beams x 256 time steps = 134 MB) in ONE launch:
event time per call, warm, for the kernel-selection shapes of the harness."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dc_sand_amd import BeamformerParameters, device  # noqa: E402
from dc_sand_amd.generator import SteeringCoefficientGenerator, simulate_input  # noqa: E402

device.set_device(0)
bp = BeamformerParameters()
gen = SteeringCoefficientGenerator(bp)
gen.upload_delays(simulate_input(bp))
nt = 256
nb = gen.output_bytes(1, nt)
buf = device.mem_alloc(nb)
for cpb in (0, 4, 6, 8, 12, 16, 32, 64):
    gen.set_tuning(chan_per_block=cpb) if cpb else gen.set_tuning()
    for _ in range(10):
        gen.generate(buf, nb, t0=0, nt=nt)
    ts = []
    for _ in range(21):
        e0, e1 = device.Event(), device.Event()
        e0.record(); gen.generate(buf, nb, t0=0, nt=nt); e1.record(); e1.synchronize()
        ts.append(e1.elapsed_ms_since(e0))
    m = float(np.median(ts))
    print(f"chan_per_block={cpb or 'default'}: {m * 1e3:.1f} us per call = {nb / m / 1e9:.2f} TB/s", flush=True)
