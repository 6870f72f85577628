"""Rows form (no per-workgroup fp64 set-up): waves per workgroup x rows per wave x pace, config 3."""
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
cands = [(nw, rpw, st, pace) for nw in (4, 8) for rpw in (1, 2, 3, 4) for st in (1,) for pace in (0, 2, 4, 6, 8, 10, 12, 14, 16, 20, 24)]
res = {c: [] for c in cands}
for rnd in range(2):
    for c in cands:
        gen.set_tuning(form=2, waves_per_block=c[0], rows_per_wave=c[1], rows_same_tile=c[2], xcd_remap=0, nontemporal=1, pace=c[3])
        ts = []
        for _ in range(10):
            e0, e1 = device.Event(), device.Event()
            e0.record(); gen.generate(buf, nb, t0=1, nt=1); e1.record(); e1.synchronize()
            ts.append(e1.elapsed_ms_since(e0))
        res[c].append(float(np.median(ts[5:])))
rows = sorted(((np.median(v), c) for c, v in res.items()))
for m, c in rows[:12]:
    print(f"nw={c[0]} rpw={c[1]} same_tile={c[2]} pace={c[3]:2d}: {m:.4f} ms -> {bp.coeffs_per_time_step() / m / 1e6:.1f} Gcoeff/s ({nb / m / 1e9:.2f} TB/s)")
