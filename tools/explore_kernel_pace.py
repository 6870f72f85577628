"""Real kernel: explicit pacing (sleep before each store) x channels per workgroup, config 3."""
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
nb = gen.output_bytes(1, 1)
buf = device.mem_alloc(nb)
for _ in range(15):
    gen.generate(buf, nb, t0=1, nt=1)
cands = [(cpb, pace) for cpb in (4, 8, 12, 16, 24, 32, 64) for pace in (0, 2, 4, 8, 12, 16, 24, 32)]
res = {c: [] for c in cands}
for rnd in range(2):
    for c in cands:
        gen.set_tuning(form=1, tiles_per_block=1, chan_per_block=c[0], nontemporal=1, pace=c[1])
        ts = []
        for _ in range(12):
            e0, e1 = device.Event(), device.Event()
            e0.record(); gen.generate(buf, nb, t0=1, nt=1); e1.record(); e1.synchronize()
            ts.append(e1.elapsed_ms_since(e0))
        res[c].append(float(np.median(ts[6:])))
rows = sorted(((np.median(v), c) for c, v in res.items()))
for m, c in rows[:14]:
    print(f"cpb={c[0]:2d} pace={c[1]:2d}: {m:.4f} ms -> {bp.coeffs_per_time_step() / m / 1e6:.1f} Gcoeff/s ({nb / m / 1e9:.2f} TB/s)")
print("by cpb (best pace):")
for cpb in (4, 8, 12, 16, 24, 32, 64):
    b = min(((np.median(res[(cpb, p)]), p) for p in (0, 2, 4, 8, 12, 16, 24, 32)))
    z = np.median(res[(cpb, 0)])
    print(f"  cpb={cpb:2d}: pace 0 -> {nb / z / 1e9:.2f} TB/s; best pace {b[1]} -> {nb / b[0] / 1e9:.2f} TB/s")
