"""Is there an XCD <-> HBM-address affinity?  Store probe with 4-KiB / 1-KiB rectangles whose column index is rotated
within groups of 8 (workgroup b runs on XCD b % 8 under round-robin dispatch)."""
import ctypes, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from dc_sand_amd import _lib, device  # noqa: E402
V = ctypes.c_void_p
device.set_device(0)
rows, cols = 32768, 512
nbytes = rows * cols * 1024
buf = device.mem_alloc(nbytes)
def run(qb, rb, rot, bt=256, nt=1):
    ts = []
    for _ in range(9):
        e0, e1 = device.Event(), device.Event()
        e0.record()
        _lib.check(_lib.lib().dcs_probe_store_pattern(V(int(buf)), rows, cols, qb, rb, 0, rot << 4, nt, bt, V(None)), "pat")
        e1.record(); e1.synchronize()
        ts.append(e1.elapsed_ms_since(e0))
    return float(np.median(ts[2:]))
for _ in range(5):
    run(4, 1, 0)
for (qb, rb, bt) in ((4, 1, 256), (1, 4, 256), (16, 1, 1024), (8, 1, 512), (2, 2, 256)):
    line = []
    for rnd in range(2):
        for rot in range(8):
            ms = run(qb, rb, rot, bt)
            line.append((rot, nbytes / ms / 1e9))
    by = {r: np.mean([v for rr, v in line if rr == r]) for r in range(8)}
    print(f"qb={qb} rb={rb} bt={bt}: " + "  ".join(f"rot{r}={by[r]:.2f}" for r in range(8)), flush=True)
