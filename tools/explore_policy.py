"""Store probe: cache policy of the 16-byte store (plain / nt / sc1 / sc0 sc1 / sc1 nt) on the best patterns."""
import ctypes, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dc_sand_amd import _lib, device  # noqa: E402
V = ctypes.c_void_p
device.set_device(0)
rows, cols = 32768, 512
nbytes = rows * cols * 1024
buf = device.mem_alloc(nbytes)
names = {0: "plain", 1: "nt", 2: "sc1", 3: "sc0 sc1", 4: "sc1 nt"}
def run(qb, rb, mode, bt):
    ts = []
    for _ in range(14):
        e0, e1 = device.Event(), device.Event()
        e0.record()
        _lib.check(_lib.lib().dcs_probe_store_pattern(V(int(buf)), rows, cols, qb, rb, 0, 0, mode, bt, V(None)), "pat")
        e1.record(); e1.synchronize()
        ts.append(e1.elapsed_ms_since(e0))
    return nbytes / float(np.median(ts[6:])) / 1e9
for (qb, rb, bt) in ((4, 1, 256), (1, 4, 256), (1, 16, 256), (4, 2, 512), (1, 64, 256)):
    out = []
    for rnd in range(2):
        for mode in range(5):
            out.append((mode, run(qb, rb, mode, bt)))
    print(f"qb={qb} rb={rb} bt={bt}: " + "  ".join(f"{names[m]}={np.mean([v for mm, v in out if mm == m]):.2f}" for m in range(5)), flush=True)
