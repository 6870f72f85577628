#!/usr/bin/env python3
"""BASELINE configs[4] in twenty lines: a hipGraph-replayed coefficient stream at a 200 us model cadence whose delay
table changes on EVERY tick and never leaves the device -- the shape of the reference's kernels, which take the table as a
device pointer (BeamformerKernels.cuh:38-42), and of the multi-GPU design, where it lands by RCCL broadcast.

    python examples/streaming_ticks.py [ant beams chan slab_channels ticks]

Each tick regenerates the channel slab in place; the last tick is verified against the CPU oracle (test infrastructure)
the way the reference's harness verifies against its CPU loop."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dc_sand_amd import BeamformerParameters  # noqa: E402
from dc_sand_amd.device import Event, Stream, mem_alloc, memcpy_dtoh, memcpy_htod, require_device, set_device  # noqa: E402
from dc_sand_amd.generator import SteeringCoefficientGenerator, simulate_input  # noqa: E402

A, B, C, NC, TICKS = (int(x) for x in sys.argv[1:6]) if len(sys.argv) >= 6 else (64, 1024, 32768, 2560, 100)
require_device()
set_device(0)
p = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B)

# two delay models resident on the device (a real system: the buffers an RCCL broadcast fills)
models = [simulate_input(p), simulate_input(p)]
models[1]["fPhase_rad"] += np.float32(0.25)
d_models = [mem_alloc(m.nbytes) for m in models]
for d, m in zip(d_models, models):
    memcpy_htod(d, np.ascontiguousarray(m))

stream = Stream()
gen = SteeringCoefficientGenerator(p)
gen.upload_delays(models[0], stream=stream)
slab_bytes = NC * p.n_pairs * 8
d_out = mem_alloc(slab_bytes)
ticks = gen.stream_begin(d_out, slab_bytes, 0, NC, stream)    # instantiates the graphs

for k in range(8):                                            # warm-up
    ticks.tick_dt_from_global(k * 200e-6, d_models[k % 2])
start, stop = Event().record(stream), Event()
for k in range(8, 8 + TICKS):
    ticks.tick_dt_from_global(k * 200e-6, d_models[k % 2])    # gather node + generator, arguments rewritten, one replay
stop.record(stream)
stop.synchronize()
us = stop.elapsed_ms_since(start) / TICKS * 1e3
print(f"{A} ant x {B} beams, slab of {NC} channels ({slab_bytes / 1e9:.2f} GB) per tick, a new device table on every tick: "
      f"{us:.1f} us per tick = {slab_bytes / us / 1e6:.2f} TB/s ({'inside' if us <= 200 else 'OUTSIDE'} the 200 us cadence)")

k_last = 8 + TICKS - 1
rows = sorted({0, NC // 2, NC - 1})
host = np.empty((len(rows), A, B, 2), dtype=np.float32)
for i, r in enumerate(rows):
    memcpy_dtoh(host[i], int(d_out) + r * p.n_pairs * 8)
ticks.end()

from oracle import bf_oracle  # noqa: E402  (verification only)

worst = 0
for i, r in enumerate(rows):
    exp = bf_oracle.generate_dt(bf_oracle.params_from(p), models[k_last % 2], [np.float32(k_last * 200e-6)], r, 1)[0, 0]
    worst = max(worst, bf_oracle.max_ulp(host[i], exp, 1)[0])
print(f"last tick, channels {rows}: max ULP distance to the CPU verifier {worst}")
sys.exit(0 if worst <= 1 else 1)
