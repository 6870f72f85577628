"""Finer scan of the paced store probe (is there anything above 7.3 TB/s?)."""
import ctypes, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dc_sand_amd import _lib, device  # noqa: E402
V = ctypes.c_void_p
device.set_device(0)
nbytes = 16 * 2**30
buf = device.mem_alloc(nbytes)
def run(mode, spt, pace):
    ts = []
    for _ in range(12):
        e0, e1 = device.Event(), device.Event()
        e0.record()
        _lib.check(_lib.lib().dcs_probe_one_store(V(int(buf)), nbytes, mode | (pace << 8), spt, 512 * 1024, V(None)), "one")
        e1.record(); e1.synchronize()
        ts.append(e1.elapsed_ms_since(e0))
    return nbytes / float(np.median(ts[6:])) / 1e9
best = []
for spt in (2, 4, 8, 16):
    row = []
    for pace in (16, 18, 20, 22, 24, 26, 28, 30, 36, 44, 56, 72, 96, 128):
        v = run(1, spt, pace)
        row.append((pace, v)); best.append((v, spt, pace))
    print(f"spt={spt:2d}: " + " ".join(f"{p}:{v:.2f}" for p, v in row), flush=True)
best.sort(reverse=True)
print("TOP", [(round(v, 2), s, p) for v, s, p in best[:8]])
