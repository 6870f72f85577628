#!/usr/bin/env python3
"""round 3: hipGraph replay (dcs_bf_stream_tick_dt) against plain launches (dcs_bf_generate_slab_dt) of the SAME slab at the
SAME explicit geometry, with and without the residency cap (dynamic LDS request) -- does the replayed kernel node run as the
plain launch does?  -> profiles/r03_streaming_config5.md"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tools.measure import SHAPES, make, per_launch_ms  # noqa: E402
from dc_sand_amd import device  # noqa: E402

bp, g, _, full, buf = make(SHAPES["cfg3"], 32)
stream = device.Stream()
for nc in (32768, 2304):
    nbytes = nc * bp.n_pairs * 8
    for cpb, wpc in ((8, -1), (8, 6), (11, 5), (11, -1), (12, -1), (12, 6)):
        g.set_tuning(chan_per_block=cpb, wg_per_cu=wpc, tiles_per_block=1)
        st = g.stream_begin(buf, nbytes, 0, nc, stream)
        k = [0]

        def tick():
            k[0] += 1
            st.tick_dt(k[0] * 200e-6)

        def plain():
            k[0] += 1
            g.generate_slab_dt(buf, nbytes, 0, nc, [k[0] * 200e-6], stream=stream)

        res = []
        for _ in range(2):
            res.append((per_launch_ms(plain, stream=stream, timed_ms=40), per_launch_ms(tick, stream=stream, timed_ms=40)))
        st.end()
        p, t = min(r[0] for r in res), min(r[1] for r in res)
        print(f"{nc} channels, chan_per_block={cpb} wg_per_cu={wpc}: plain {p * 1e3:.1f} us ({nbytes / p / 1e9:.2f} TB/s), graph replay {t * 1e3:.1f} us "
              f"({nbytes / t / 1e9:.2f} TB/s), replay/plain = {t / p:.4f}", flush=True)
g.close()
