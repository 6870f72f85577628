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
cands = [(x, tpb, cpb) for x in (0, 1) for tpb in (1, 2, 4) for cpb in (8, 10, 12, 13, 14, 15, 16, 18, 20, 24, 32)]
res = {c: [] for c in cands}
for rnd in range(3):
    for c in cands:
        gen.set_tuning(form=1, xcd_remap=c[0], tiles_per_block=c[1], chan_per_block=c[2], nontemporal=1)
        ts = []
        for _ in range(7):
            e0, e1 = device.Event(), device.Event()
            e0.record(); gen.generate(buf, nb, t0=1, nt=1); e1.record(); e1.synchronize()
            ts.append(e1.elapsed_ms_since(e0))
        res[c].append(float(np.median(ts[2:])))
rows = sorted(((np.median(v), c) for c, v in res.items()))
for m, c in rows[:16]:
    print(f"xcd={c[0]} tpb={c[1]} cpb={c[2]:2d}: {m:.4f} ms -> {bp.coeffs_per_time_step() / m / 1e6:.1f} Gcoeff/s ({nb / m / 1e9:.2f} TB/s)")
