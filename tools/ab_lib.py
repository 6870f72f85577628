"""One build's numbers for an A/B of two builds of the library on the SAME box:
    DCS_LIB_PATH=tools/ab/libdcs_old.so python tools/ab_lib.py ; python tools/ab_lib.py
fp32 generator (a few walks), fp16 generator, fused beamformer; median of 9 launches after a settle."""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dc_sand_amd import BeamformerParameters, device  # noqa: E402
from dc_sand_amd.generator import SteeringCoefficientGenerator, simulate_input  # noqa: E402

device.set_device(0)
tag = "old" if os.environ.get("DCS_LIB_PATH") else "new"


def med(fn, settle=8, reps=9):
    for _ in range(settle):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = device.Event(), device.Event()
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e1.elapsed_ms_since(e0))
    return float(np.median(ts))


bp = BeamformerParameters(NR_CHANNELS=32768, NR_STATIONS=64, NR_BEAMS=1024)
gen = SteeringCoefficientGenerator(bp)
gen.upload_delays(simulate_input(bp))
nb = gen.output_bytes(1, 1)
buf = device.mem_alloc(nb)
n = bp.coeffs_per_time_step()
out = []
for cpb in (11, 12, 14):
    gen.set_tuning(form=1, tiles_per_block=1, chan_per_block=cpb, nontemporal=1)
    m = med(lambda: gen.generate(buf, nb, t0=1, nt=1))
    out.append(f"fp32 cpb{cpb} {n / m / 1e6:.0f}")
gen.set_tuning()
for cpb in (64, 128, 256):
    gen.set_tuning(form=1, tiles_per_block=1, chan_per_block=cpb, nontemporal=1)
    m = med(lambda: gen.generate(buf, nb // 2, t0=1, nt=1, bitwidth=0))
    out.append(f"fp16 cpb{cpb} {n / m / 1e6:.0f}")
gen.close()
for (A, B, C, nt) in ((64, 64, 4096, 64), (256, 64, 1024, 64)):
    fp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B, NR_SAMPLES_PER_CHANNEL=nt)
    g = SteeringCoefficientGenerator(fp)
    g.upload_delays(simulate_input(fp))
    ab, bb = A * C * nt * 2, B * C * nt * 8
    d_ant, d_beams = device.mem_alloc(ab), device.mem_alloc(bb)
    device.memset(d_ant, 3, ab)
    m = med(lambda: g.generate_and_beamform(d_ant, ab, d_beams, bb, 0, nt))
    out.append(f"fused{A}x{B} {A * B * C * nt / m / 1e6:.0f}")
    g.close()
print(tag, "|", " | ".join(out), flush=True)
