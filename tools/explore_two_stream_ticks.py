"""Config 5: do ticks issued alternately on TWO streams (into two slab buffers) sustain a shorter period than
back-to-back launches on one stream?  (The tail of one tick overlaps the ramp of the next.)"""
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
streams = [device.Stream(), device.Stream()]
for nc in (1024, 2048, 2560, 2816, 3072):
    nb = nc * bp.n_pairs * 8
    bufs = [device.mem_alloc(nb), device.mem_alloc(nb)]
    for mode in ("one stream", "two streams"):
        def run(n):
            for i in range(n):
                k = i % 2 if mode == "two streams" else 0
                gen.generate_slab(bufs[i % 2], nb, 0, nc, t0=i, nt=1, stream=streams[k].handle)
            for s in streams:
                s.synchronize()
        run(50)
        t0 = time.perf_counter()
        run(400)
        dt = (time.perf_counter() - t0) / 400
        print(f"{nc} channels, {mode}: {dt * 1e6:.1f} us per tick = {nb / dt / 1e12:.2f} TB/s", flush=True)
    for b in bufs:
        b.free()
