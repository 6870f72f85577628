"""Does a long heavy run lower the sustained rate afterwards (thermal / power state)?  Default geometry, config 3:
rate over 50 launches right after start-up, then after 1, 2, 4 s of continuous generation, then after 2 s of idle."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dc_sand_amd import BeamformerParameters, device  # noqa: E402
from dc_sand_amd.generator import SteeringCoefficientGenerator, simulate_input  # noqa: E402

device.set_device(0)
bp = BeamformerParameters(NR_CHANNELS=32768, NR_STATIONS=64, NR_BEAMS=1024)
gen = SteeringCoefficientGenerator(bp)
gen.upload_delays(simulate_input(bp))
nb = gen.output_bytes(1, 1)
buf = device.mem_alloc(nb)


def rate(n=50):
    e0, e1 = device.Event(), device.Event()
    e0.record()
    for _ in range(n):
        gen.generate(buf, nb, t0=1, nt=1)
    e1.record(); e1.synchronize()
    return nb * n / e1.elapsed_ms_since(e0) / 1e9


def burn(seconds):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(40):
            gen.generate(buf, nb, t0=1, nt=1)
        device.synchronize()


for _ in range(20):
    gen.generate(buf, nb, t0=1, nt=1)
device.synchronize()
print(f"after 20 launches: {rate():.3f} TB/s  {rate():.3f}  {rate():.3f}", flush=True)
for s in (1, 2, 4, 8):
    burn(s)
    print(f"after {s} s more of continuous generation: {rate():.3f} TB/s  {rate():.3f}", flush=True)
time.sleep(2.0)
print(f"after 2 s idle: {rate():.3f} TB/s  {rate():.3f}  {rate():.3f}", flush=True)
