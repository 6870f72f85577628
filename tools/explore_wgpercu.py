"""Occupancy-limited launches of the production generator (tuning knob wg_per_cu: unused dynamic LDS so that at
most k workgroups are resident per CU) x channels per workgroup, config 3.  Two interleaved rounds."""
import os
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
KS = tuple(int(x) for x in os.environ.get("DCS_KS", "0,7,6,5,4,3").split(","))
CPBS = tuple(int(x) for x in os.environ.get("DCS_CPBS", "8,10,12,14,16,20,24,32").split(","))
TPB = int(os.environ.get("DCS_TPB", "1"))
FORM = int(os.environ.get("DCS_FORM", "1"))
NT = int(os.environ.get("DCS_NT", "1"))
res = {}
for rnd in range(2):
    for k in KS:
        for cpb in CPBS:
            gen.set_tuning(form=FORM, tiles_per_block=TPB, chan_per_block=cpb, nontemporal=NT, wg_per_cu=k if k else -1)
            for _ in range(8):
                gen.generate(buf, nb, t0=1, nt=1)
            ts = []
            for _ in range(9):
                e0, e1 = device.Event(), device.Event()
                e0.record(); gen.generate(buf, nb, t0=1, nt=1); e1.record(); e1.synchronize()
                ts.append(e1.elapsed_ms_since(e0))
            res.setdefault((k, cpb), []).append(float(np.median(ts)))
print(f"tpb={TPB} form={FORM} nontemporal={NT}"); print("TB/s; rows = workgroups per CU (0 = unlimited), columns = channels per workgroup " + " ".join(f"{c:5d}" for c in CPBS))
for k in KS:
    print(f"k={k}: " + " ".join(f"{nb / float(np.min(res[(k, c)])) / 1e9:5.2f}" for c in CPBS), flush=True)
