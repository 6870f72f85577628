"""Rows form with an occupancy limit (wg_per_cu) x (waves per workgroup, rows per wave), config 3."""
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
res = {}
import os
NWS = tuple(int(x) for x in os.environ.get("DCS_NWS", "4,8").split(","))
RPWS = tuple(int(x) for x in os.environ.get("DCS_RPWS", "2,3,4").split(","))
ST = int(os.environ.get("DCS_SAME_TILE", "-1"))
PACE = int(os.environ.get("DCS_PACE", "0"))
cands = [(nw, rpw, k) for nw in NWS for rpw in RPWS for k in (-1, 6, 5, 4, 3, 2)]
for rnd in range(2):
    for c in cands:
        gen.set_tuning(form=2, waves_per_block=c[0], rows_per_wave=c[1], wg_per_cu=c[2], rows_same_tile=ST, pace=PACE)
        for _ in range(8):
            gen.generate(buf, nb, t0=1, nt=1)
        ts = []
        for _ in range(9):
            e0, e1 = device.Event(), device.Event()
            e0.record(); gen.generate(buf, nb, t0=1, nt=1); e1.record(); e1.synchronize()
            ts.append(e1.elapsed_ms_since(e0))
        res.setdefault(c, []).append(float(np.median(ts)))
print(f"same_tile={ST} pace={PACE}")
for nw in NWS:
    for rpw in RPWS:
        print(f"nw={nw} rpw={rpw}: " + " ".join(f"k={k}:{nb / min(res[(nw, rpw, k)]) / 1e9:.2f}" for k in (-1, 6, 5, 4, 3, 2)), flush=True)
