"""A few launches of each hot kernel (fp32 / fp16 generator at config 3, fused beamformer) for
`rocprofv3 --pmc ...` passes (tools/run via: rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -- python3 tools/pmc_targets.py)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dc_sand_amd import BeamformerParameters, device  # noqa: E402
from dc_sand_amd.generator import SteeringCoefficientGenerator, simulate_input  # noqa: E402

device.set_device(0)
bp = BeamformerParameters(NR_CHANNELS=32768, NR_STATIONS=64, NR_BEAMS=1024)
g = SteeringCoefficientGenerator(bp)
g.upload_delays(simulate_input(bp))
nb = g.output_bytes(1, 1)
buf = device.mem_alloc(nb)
for _ in range(6):
    g.generate(buf, nb, t0=1, nt=1, bitwidth=1)
for _ in range(6):
    g.generate(buf, g.output_bytes(0, 1), t0=1, nt=1, bitwidth=0)
device.synchronize()
g.close()
A, B, C, nt = 64, 64, 4096, 64
bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B, NR_SAMPLES_PER_CHANNEL=nt)
g = SteeringCoefficientGenerator(bp)
g.upload_delays(simulate_input(bp))
d_ant = device.mem_alloc(A * C * nt * 2)
device.memset(d_ant, 3, A * C * nt * 2)
d_beams = device.mem_alloc(B * C * nt * 8)
for _ in range(6):
    g.generate_and_beamform(d_ant, A * C * nt * 2, d_beams, B * C * nt * 8, 0, nt)
device.synchronize()
