"""How much of the 7.1-7.3 TB/s 'ceiling' is per-wave overhead?  The leanest store kernels."""
import ctypes, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dc_sand_amd import _lib, device  # noqa: E402
V = ctypes.c_void_p
device.set_device(0)
nbytes = 16 * 2**30
buf = device.mem_alloc(nbytes)
def run(mode, spt):
    ts = []
    for _ in range(16):
        e0, e1 = device.Event(), device.Event()
        e0.record()
        _lib.check(_lib.lib().dcs_probe_one_store(V(int(buf)), nbytes, mode, spt, 512 * 1024, V(None)), "one")
        e1.record(); e1.synchronize()
        ts.append(e1.elapsed_ms_since(e0))
    return nbytes / float(np.median(ts[8:])) / 1e9
for rnd in range(2):
    for spt in (1, 2, 3, 4):
        print(f"stores/thread={spt}: plain {run(0, spt):.2f} TB/s   nt {run(1, spt):.2f} TB/s", flush=True)
