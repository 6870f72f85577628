"""Exploration of launch geometries on one MI355X: prints Gcoeff/s and TB/s for
the store-only skeleton (nomath), the real kernel, and a linear fill, at
BASELINE config 3 (64 ant x 1024 beam x 32768 chan = 16 GiB per time step).
Usage: python tools/explore.py [--chan 32768] [--quick]
"""
import argparse
import ctypes
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from dc_sand_amd import BeamformerParameters, _lib, device  # noqa: E402
from dc_sand_amd.generator import SteeringCoefficientGenerator, simulate_input  # noqa: E402


def timeit(fn, warm=2, reps=5):
    for _ in range(warm):
        fn()
    device.synchronize()
    times = []
    for _ in range(reps):
        e0, e1 = device.Event(), device.Event()
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        times.append(e1.elapsed_ms_since(e0))
    return float(np.median(times)), float(np.min(times))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ant", type=int, default=64)
    ap.add_argument("--beams", type=int, default=1024)
    ap.add_argument("--chan", type=int, default=32768)
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    device.require_device()
    device.set_device(0)
    print("device:", device.device_name(0), flush=True)
    bp = BeamformerParameters(NR_CHANNELS=args.chan, NR_STATIONS=args.ant, NR_BEAMS=args.beams)
    gen = SteeringCoefficientGenerator(bp)
    gen.upload_delays(simulate_input(bp))
    nbytes = gen.output_bytes(1, 1)
    ncoeff = bp.coeffs_per_time_step()
    buf = device.mem_alloc(nbytes)
    V = ctypes.c_void_p
    print(f"output {nbytes / 2**30:.2f} GiB, {ncoeff / 1e9:.3f} Gcoeff", flush=True)

    for nt_store in (0, 1):
        med, mn = timeit(lambda: _lib.check(_lib.lib().dcs_probe_fill(V(int(buf)), nbytes, nt_store, V(None)), "fill"))
        print(f"fill linear nt={nt_store}: med {med:.3f} ms  min {mn:.3f} ms  -> {nbytes / mn / 1e9:.2f} TB/s (min)", flush=True)
    # hipMemsetAsync as a second reference
    med, mn = timeit(lambda: device.memset(buf, 0, nbytes))
    print(f"hipMemsetAsync: med {med:.3f} ms min {mn:.3f} -> {nbytes / mn / 1e9:.2f} TB/s", flush=True)

    def run(label, **tuning):
        gen.set_tuning(**tuning)
        med, mn = timeit(lambda: gen.generate(buf, nbytes, t0=1, nt=1), warm=2, reps=7)
        print(
            f"{label}: med {med:.3f} ms min {mn:.3f} ms -> {ncoeff / med / 1e6:.1f} Gcoeff/s, {nbytes / med / 1e9:.2f} TB/s"
            f" ({nbytes / med / 1e9 / 8.0 * 100:.1f}% of 8 TB/s)",
            flush=True,
        )

    for nomath in (True, False):
        tag = "nomath" if nomath else "math  "
        for nw in (4, 8, 16):
            for rpw in (1, 2, 4):
                for xcd in (0, 1):
                    for nts in (0, 1):
                        run(f"{tag} rows nw={nw:2d} rpw={rpw} xcd={xcd} nt={nts}", form=2, waves_per_block=nw, rows_per_wave=rpw,
                            xcd_remap=xcd, nontemporal=nts, nomath=nomath)
        if not args.quick:
            for tpb in (1, 4):
                for cpb in (0, 16, 64):
                    for nts in (0, 1):
                        run(f"{tag} tiled tpb={tpb} cpb={cpb:4d} nt={nts}", form=1, tiles_per_block=tpb, chan_per_block=cpb,
                            nontemporal=nts, nomath=nomath)
    # fp16
    gen.set_tuning()
    nb16 = gen.output_bytes(0, 1)
    med, mn = timeit(lambda: gen.generate(buf, nb16, t0=1, nt=1, bitwidth=0))
    print(f"fp16 default: med {med:.3f} ms -> {ncoeff / med / 1e6:.1f} Gcoeff/s, {nb16 / med / 1e9:.2f} TB/s", flush=True)
    # reference launch shapes
    for kern, name in ((1, "MULTIPLE_CHANNELS"), (0, "NAIVE")):
        med, mn = timeit(lambda: gen.generate(buf, nbytes, t0=1, nt=1, kernel=kern), warm=1, reps=3)
        print(f"{name}: med {med:.3f} ms -> {ncoeff / med / 1e6:.1f} Gcoeff/s", flush=True)


if __name__ == "__main__":
    main()
