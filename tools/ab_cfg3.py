"""Time the default fp32 launch at config 3 (interleavable A/B across library builds via DCS_LIB_PATH)."""
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
nbytes = gen.output_bytes(1, 1)
buf = device.mem_alloc(nbytes)
bw = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nb = gen.output_bytes(bw, 1)
for _ in range(15):
    gen.generate(buf, nb, t0=1, nt=1, bitwidth=bw)
device.synchronize()
ts = []
for _ in range(40):
    e0, e1 = device.Event(), device.Event()
    e0.record()
    gen.generate(buf, nb, t0=1, nt=1, bitwidth=bw)
    e1.record()
    e1.synchronize()
    ts.append(e1.elapsed_ms_since(e0))
print(f"bw={bw} median {np.median(ts):.4f} ms  min {np.min(ts):.4f}  -> {bp.coeffs_per_time_step() / np.median(ts) / 1e6:.1f} Gcoeff/s")
