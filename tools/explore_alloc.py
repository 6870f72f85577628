"""Does the KIND of device allocation the caller hands in change the sustained write rate?
hipMalloc (coarse-grained, the default) vs hipExtMallocWithFlags fine-grained / uncached / contiguous.
Config 3, default geometry and a few neighbours, median of 9 launches each, two interleaved rounds."""
import ctypes
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dc_sand_amd import BeamformerParameters, device  # noqa: E402
from dc_sand_amd.generator import SteeringCoefficientGenerator, simulate_input  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipFree.argtypes = [ctypes.c_void_p]
import os
KINDS = {"hipMalloc": None, "finegrained": 0x1, "uncached": 0x3, "contiguous": 0x4}
if os.environ.get("DCS_KINDS"):
    KINDS = {k: KINDS[k] for k in os.environ["DCS_KINDS"].split(",")}
CPBS = tuple(int(x) for x in os.environ.get("DCS_CPBS", "12,13,14,16").split(","))
TPBS = tuple(int(x) for x in os.environ.get("DCS_TPBS", "1").split(","))

device.set_device(0)
bp = BeamformerParameters(NR_CHANNELS=32768, NR_STATIONS=64, NR_BEAMS=1024)
gen = SteeringCoefficientGenerator(bp)
gen.upload_delays(simulate_input(bp))
nb = gen.output_bytes(1, 1)


def alloc(kind):
    if KINDS[kind] is None:
        return device.mem_alloc(nb), None
    p = ctypes.c_void_p()
    rc = hip.hipExtMallocWithFlags(ctypes.byref(p), nb, KINDS[kind])
    if rc != 0:
        print(f"{kind}: hipExtMallocWithFlags -> {rc}", flush=True)
        return None, None
    return p.value, p


res = {}
for rnd in range(2):
    for kind in KINDS:
        buf, raw = alloc(kind)
        if buf is None:
            continue
        for tpb, cpb in ((t, c) for t in TPBS for c in CPBS):
            gen.set_tuning(form=1, tiles_per_block=tpb, chan_per_block=cpb, nontemporal=1)
            for _ in range(8):
                gen.generate(buf, nb, t0=1, nt=1)
            ts = []
            for _ in range(9):
                e0, e1 = device.Event(), device.Event()
                e0.record(); gen.generate(buf, nb, t0=1, nt=1); e1.record(); e1.synchronize()
                ts.append(e1.elapsed_ms_since(e0))
            res.setdefault((kind, tpb, cpb), []).append(float(np.median(ts)))
        device.synchronize()
        if raw is not None:
            hip.hipFree(raw)
        else:
            buf.free()
for (kind, tpb, cpb), v in res.items():
    m = float(np.median(v))
    print(f"{kind:12s} tpb={tpb} cpb={cpb:2d}: {m:.4f} ms -> {bp.coeffs_per_time_step() / m / 1e6:.1f} Gcoeff/s ({nb / m / 1e9:.2f} TB/s)", flush=True)
