"""Does limiting the CUs a stream may use (hipExtStreamCreateWithCUMask) change the sustained write rate?
Store-only probe (1 KiB x 4 rows per 256-thread workgroup, the best pattern of r01_store_patterns.md) and the real
generator at config 3 on streams with different CU masks."""
import ctypes
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dc_sand_amd import BeamformerParameters, _lib, device  # noqa: E402
from dc_sand_amd.generator import SteeringCoefficientGenerator, simulate_input  # noqa: E402

V = ctypes.c_void_p
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(V), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]

device.set_device(0)


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    s = V()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return s


def med(fn, stream, settle=6, reps=9):
    for _ in range(settle):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = device.Event(), device.Event()
        e0.record(stream); fn(); e1.record(stream); e1.synchronize()
        ts.append(e1.elapsed_ms_since(e0))
    return float(np.median(ts))


ALL = (1 << 256) - 1
masks = {
    "all 256": ALL,
    "low 192 bits": (1 << 192) - 1,
    "low 128 bits": (1 << 128) - 1,
    "even bits (128)": int("01" * 128, 2),
    "3 of 4 bits (192)": int("0111" * 64, 2),
    "7 of 8 bits (224)": int("01111111" * 32, 2),
    "bits with i%16 < 12 (192)": sum(1 << i for i in range(256) if i % 16 < 12),
    "bits with i%32 < 24 (192)": sum(1 << i for i in range(256) if i % 32 < 24),
}
rows, cols = 32768, 512
nbytes = rows * cols * 1024
buf = device.mem_alloc(nbytes)
bp = BeamformerParameters(NR_CHANNELS=32768, NR_STATIONS=64, NR_BEAMS=1024)
gen = SteeringCoefficientGenerator(bp)
gen.upload_delays(simulate_input(bp))
device.synchronize()
for name, bits in masks.items():
    s = masked_stream(bits)
    sv = s.value
    line = f"{name:28s}"
    m = med(lambda: _lib.check(_lib.lib().dcs_probe_store_pattern(V(int(buf)), rows, cols, 1, 4, 0, 0, 1, 256, s), "pat"), sv)
    line += f" probe {nbytes / m / 1e9:.2f} TB/s |"
    for cpb in (10, 12, 14, 16, 24):
        gen.set_tuning(form=1, tiles_per_block=1, chan_per_block=cpb, nontemporal=1)
        m = med(lambda: gen.generate(buf, nbytes, t0=1, nt=1, stream=sv), sv)
        line += f" cpb{cpb} {nbytes / m / 1e9:.2f}"
    print(line, flush=True)
