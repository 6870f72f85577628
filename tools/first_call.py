"""Latency of the first generate() of a process vs steady state (default shape, all 256 time steps)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dc_sand_amd import BeamformerParameters, device  # noqa: E402
from dc_sand_amd.generator import SteeringCoefficientGenerator, simulate_input  # noqa: E402

device.set_device(0)
bp = BeamformerParameters()
t0 = time.perf_counter()
g = SteeringCoefficientGenerator(bp)
g.upload_delays(simulate_input(bp))
nbytes = g.output_bytes(1, 256)
buf = device.mem_alloc(nbytes)
device.synchronize()
print(f"create+upload+alloc: {(time.perf_counter() - t0) * 1e3:.2f} ms")
for i in range(4):
    e0, e1 = device.Event(), device.Event()
    e0.record()
    g.generate(buf, nbytes, t0=0, nt=256)
    e1.record()
    e1.synchronize()
    print(f"call {i}: {e1.elapsed_ms_since(e0) * 1e3:.1f} us")
