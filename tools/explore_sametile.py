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
cands = [dict(form=2, waves_per_block=nw, rows_per_wave=rpw, rows_same_tile=1, xcd_remap=x, nontemporal=1) for nw in (4, 8, 16) for rpw in (1, 2, 3, 4) for x in (0, 1)]
cands += [dict(form=1, tiles_per_block=1, chan_per_block=c, nontemporal=1) for c in (12, 13, 14)]
cands += [dict(form=2, waves_per_block=4, rows_per_wave=2, xcd_remap=1, nontemporal=1)]
res = [[] for _ in cands]
for rnd in range(3):
    for i, c in enumerate(cands):
        gen.set_tuning(**c)
        ts = []
        for _ in range(7):
            e0, e1 = device.Event(), device.Event()
            e0.record(); gen.generate(buf, nb, t0=1, nt=1); e1.record(); e1.synchronize()
            ts.append(e1.elapsed_ms_since(e0))
        res[i].append(float(np.median(ts[2:])))
rows = sorted(((np.median(v), i) for i, v in enumerate(res)))
for m, i in rows[:14]:
    print(f"{m:.4f} ms -> {bp.coeffs_per_time_step() / m / 1e6:.1f} Gcoeff/s ({nb / m / 1e9:.2f} TB/s)  {cands[i]}")
