"""Which write patterns does the MI355X memory system sustain?  Store-only
probe over a rows x cols-KiB matrix (default: 32768 rows x 512 KiB = 16 GiB,
the config-3 output)."""
import ctypes
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dc_sand_amd import _lib, device  # noqa: E402
from explore import timeit  # noqa: E402

V = ctypes.c_void_p


def main():
    device.require_device()
    device.set_device(0)
    rows, cols = 32768, 512
    nbytes = rows * cols * 1024
    buf = device.mem_alloc(nbytes)
    res = []
    combos = []
    for bt in (256, 512, 1024, 64):
        nw = bt // 64
        for spw in (1, 2, 4, 8):  # stores per wave
            chunks = nw * spw
            for qb in (1, 2, 4, 8, 16, 32, 64):
                if chunks % qb:
                    continue
                rb = chunks // qb
                for xcd in (0, 1):
                    for nt in (0, 1):
                        combos.append((qb, rb, 0, xcd, nt, bt))
    for (qb, rb, order, xcd, nt, bt) in combos:
        fn = lambda: _lib.check(_lib.lib().dcs_probe_store_pattern(V(int(buf)), rows, cols, qb, rb, order, xcd, nt, bt, V(None)), "pat")
        med, mn = timeit(fn, warm=1, reps=7)
        res.append((nbytes / med / 1e9, qb, rb, order, xcd, nt, bt, nbytes / mn / 1e9))
        print(f"bt={bt:4d} qb={qb:4d} rb={rb:5d} xcd={xcd} nt={nt}: med {med:.3f} ms {nbytes / med / 1e9:.2f} TB/s (best {nbytes / mn / 1e9:.2f})", flush=True)
    res.sort(reverse=True)
    print("TOP 25 (median)")
    for r in res[:25]:
        print(f"  {r[0]:.2f} TB/s (best {r[7]:.2f}) bt={r[6]} qb={r[1]} rb={r[2]} xcd={r[4]} nt={r[5]}")


if __name__ == "__main__":
    main()
