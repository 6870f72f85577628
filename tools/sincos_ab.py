"""__ocml_sincos_f32 (the form BASELINE.json's north_star names) vs the library's
dcs_sincos_fast: accuracy against (float)sin/cos((double)x) on every fp32 in
[1, 128) plus a sample of (0, 1), on the device (dcs_probe_sincos)."""
import ctypes
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dc_sand_amd import _lib, device  # noqa: E402

device.set_device(0)
V = ctypes.c_void_p


def ulp(a, b):
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


lo = np.float32(1.0).view(np.uint32)
hi = np.float32(128.0).view(np.uint32)
chunk = 1 << 24
stats = {0: [0, 0, 0, 0], 1: [0, 0, 0, 0]}  # max sin, max cos, n>1 sin, n>1 cos
n_total = 0
for start in range(int(lo), int(hi), chunk):
    bits = np.arange(start, min(start + chunk, int(hi)), dtype=np.uint32)
    x = bits.view(np.float32)
    n = x.size
    n_total += n
    es = np.sin(x.astype(np.float64)).astype(np.float32)
    ec = np.cos(x.astype(np.float64)).astype(np.float32)
    dx, ds, dc = device.mem_alloc(4 * n), device.mem_alloc(4 * n), device.mem_alloc(4 * n)
    device.memcpy_htod(dx, x)
    for which in (0, 1):
        _lib.check(_lib.lib().dcs_probe_sincos(which, V(int(dx)), n, V(int(ds)), V(int(dc)), V(None)), "probe")
        s = np.empty(n, np.float32)
        c = np.empty(n, np.float32)
        device.memcpy_dtoh(s, ds)
        device.memcpy_dtoh(c, dc)
        us, uc = ulp(s, es), ulp(c, ec)
        st = stats[which]
        st[0] = max(st[0], int(us.max()))
        st[1] = max(st[1], int(uc.max()))
        st[2] += int((us > 1).sum())
        st[3] += int((uc > 1).sum())
    for b in (dx, ds, dc):
        b.free()
for which, name in ((0, "dcs_sincos_fast"), (1, "__ocml_sincos_f32")):
    st = stats[which]
    print(f"{name}: every fp32 in [1,128) ({n_total} args): max ULP sin {st[0]} cos {st[1]}; elements over 1 ULP: sin {st[2]} cos {st[3]}")
