"""Does dcs_bf_autotune find (nearly) the best geometry of an exhaustive sweep, for several shapes?"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from dc_sand_amd import BeamformerParameters, device  # noqa: E402
from dc_sand_amd.generator import SteeringCoefficientGenerator, simulate_input  # noqa: E402
from explore import timeit  # noqa: E402
device.set_device(0)
for name, A, B, C, nt in (("cfg2 64x64x4096", 64, 64, 4096, 1), ("mid 64x256x8192", 64, 256, 8192, 1), ("cfg3 64x1024x32768", 64, 1024, 32768, 1),
                          ("cfg4/GPU 256x512x32768", 256, 512, 32768, 1), ("narrow 16x16x32768", 16, 16, 32768, 1)):
    bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B)
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(simulate_input(bp))
    nb = g.output_bytes(1, nt)
    buf = device.mem_alloc(nb)
    n = bp.coeffs_per_time_step() * nt
    for _ in range(10):
        g.generate(buf, nb, t0=1, nt=nt)
    d_ms, _ = timeit(lambda: g.generate(buf, nb, t0=1, nt=nt), warm=12, reps=9)
    chosen = g.autotune(buf, nb)
    a_ms, _ = timeit(lambda: g.generate(buf, nb, t0=1, nt=nt), warm=12, reps=9)
    best = (1e9, None)
    for tpb, wpc in ((1, -1), (1, 7), (1, 6), (1, 5), (2, -1), (4, -1)):
        for cpb in (4, 8, 10, 11, 12, 13, 14, 15, 16, 18, 20, 24, 32):
            g.set_tuning(form=1, tiles_per_block=tpb, chan_per_block=cpb, nontemporal=1, wg_per_cu=wpc)
            ms, _ = timeit(lambda: g.generate(buf, nb, t0=1, nt=nt), warm=10, reps=7)
            if ms < best[0]:
                best = (ms, (tpb, cpb, wpc))
    print(f"{name}: default {n / d_ms / 1e6:.1f}  autotuned {n / a_ms / 1e6:.1f} (tpb={chosen['tiles_per_block']} cpb={chosen['chan_per_block']} wg_per_cu={chosen['wg_per_cu']})  "
          f"sweep best {n / best[0] / 1e6:.1f} {best[1]}  Gcoeff/s", flush=True)
    g.close(); buf.free()
