"""BASELINE configs[4]: streaming at a 200 us update cadence with time-varying
delay polynomials, hipGraph-captured launch.  Reports (i) the achieved update
period of the FULL 64 x 1024 x 32768 tensor (it cannot meet 200 us: 16 GiB is
>= 2.15 ms at the 8 TB/s peak) and (ii) the largest channel slab whose update
period stays <= 200 us, for graph replay and for a plain launch.
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dc_sand_amd import BeamformerParameters, device  # noqa: E402
from dc_sand_amd.generator import SteeringCoefficientGenerator, simulate_input  # noqa: E402


def period_us(fn, stream, ticks=200, warm=20):
    for i in range(warm):
        fn(i)
    stream.synchronize()
    e0, e1 = device.Event(), device.Event()
    t0 = time.perf_counter()
    e0.record(stream)
    for i in range(ticks):
        fn(warm + i)
    e1.record(stream)
    e1.synchronize()
    wall = (time.perf_counter() - t0) / ticks * 1e6
    return e1.elapsed_ms_since(e0) / ticks * 1e3, wall


def main():
    device.require_device()
    device.set_device(0)
    bp = BeamformerParameters(NR_CHANNELS=32768, NR_STATIONS=64, NR_BEAMS=1024)
    gen = SteeringCoefficientGenerator(bp)
    table = simulate_input(bp)
    gen.upload_delays(table)
    stream = device.Stream()
    full = gen.output_bytes(1, 1)
    buf = device.mem_alloc(full)
    out = {"config": "64ant x 1024beam x 32768chan, fp32, one time step per tick", "cadence_target_us": 200.0, "slabs": []}

    def measure(nc, with_table_updates):
        nbytes = nc * bp.n_pairs * 8
        st = gen.stream_begin(buf, nbytes, 0, nc, stream)
        if with_table_updates:
            dev_us, wall_us = period_us(lambda i: st.tick(i, table if i % 16 == 0 else None), stream, ticks=100, warm=10)
        else:
            dev_us, wall_us = period_us(lambda i: st.tick(i), stream)
        st.end()
        plain_us, plain_wall = period_us(lambda i: gen.generate_slab(buf, nbytes, 0, nc, t0=i, nt=1, stream=stream), stream)
        return dict(channels=nc, bytes=nbytes, graph_period_us=dev_us, graph_wall_us=wall_us, plain_period_us=plain_us,
                    plain_wall_us=plain_wall, graph_TBps=nbytes / dev_us / 1e6, plain_TBps=nbytes / plain_us / 1e6)

    r = measure(bp.NR_CHANNELS, False)
    out["full_tensor"] = r
    print("full tensor:", json.dumps(r), flush=True)
    for nc in (256, 512, 1024, 1536, 2048, 2304, 2560, 2816, 3072, 4096):
        r = measure(nc, False)
        out["slabs"].append(r)
        print(json.dumps(r), flush=True)
    ok = [s for s in out["slabs"] if max(s["graph_period_us"], s["graph_wall_us"]) <= 200.0]
    out["largest_slab_at_200us_graph"] = max(ok, key=lambda s: s["channels"]) if ok else None
    okp = [s for s in out["slabs"] if max(s["plain_period_us"], s["plain_wall_us"]) <= 200.0]
    out["largest_slab_at_200us_plain"] = max(okp, key=lambda s: s["channels"]) if okp else None
    out["with_table_update_every_16_ticks_2048ch"] = measure(2048, True)
    print("SUMMARY", json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
