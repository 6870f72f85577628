"""Self-paced stores: a wave waits for its previous store's acknowledgement (s_waitcnt vmcnt(0)) before the next."""
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
    for _ in range(14):
        e0, e1 = device.Event(), device.Event()
        e0.record()
        _lib.check(_lib.lib().dcs_probe_one_store(V(int(buf)), nbytes, mode | (pace << 8), spt, 512 * 1024, V(None)), "one")
        e1.record(); e1.synchronize()
        ts.append(e1.elapsed_ms_since(e0))
    return nbytes / float(np.median(ts[7:])) / 1e9
for rnd in range(2):
    for spt in (2, 4, 8, 16, 64):
        print(f"stores/thread={spt:2d}: unpaced nt {run(1, spt, 0):.2f}  self-paced nt {run(1, spt, 0x8000):.2f}  self-paced plain {run(0, spt, 0x8000):.2f} TB/s", flush=True)
