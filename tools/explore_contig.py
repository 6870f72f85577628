"""Store probe: a wave writing several ADJACENT 1-KiB chunks vs the wave-interleaved assignment."""
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
def run(qb, rb, flags, bt, nt=1):
    ts = []
    for _ in range(12):
        e0, e1 = device.Event(), device.Event()
        e0.record()
        _lib.check(_lib.lib().dcs_probe_store_pattern(V(int(buf)), rows, cols, qb, rb, 0, flags, nt, bt, V(None)), "pat")
        e1.record(); e1.synchronize()
        ts.append(e1.elapsed_ms_since(e0))
    return nbytes / float(np.median(ts[5:])) / 1e9
res = []
for bt in (64, 128, 256):
    nw = bt // 64
    for spw in (2, 3, 4, 8):
        for (qb, rb) in ((nw * spw, 1), (spw, nw), (nw, spw), (1, nw * spw)):
            for flags in (0, 2):
                res.append((run(qb, rb, flags, bt), bt, spw, qb, rb, flags))
                print(f"bt={bt} spw={spw} qb={qb} rb={rb} contiguous={flags >> 1}: {res[-1][0]:.2f} TB/s", flush=True)
res.sort(reverse=True)
print("TOP")
for r in res[:12]:
    print(f"  {r[0]:.2f} TB/s bt={r[1]} spw={r[2]} qb={r[3]} rb={r[4]} contiguous={r[5] >> 1}")
