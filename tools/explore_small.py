"""Finer sweep of short-wave geometries (both forms, fp32 and fp16) at config 3."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from dc_sand_amd import BeamformerParameters, device  # noqa: E402
from dc_sand_amd.generator import SteeringCoefficientGenerator, simulate_input  # noqa: E402
from explore import timeit  # noqa: E402


def main():
    device.require_device()
    device.set_device(0)
    bp = BeamformerParameters(NR_CHANNELS=32768, NR_STATIONS=64, NR_BEAMS=1024)
    gen = SteeringCoefficientGenerator(bp)
    gen.upload_delays(simulate_input(bp))
    ncoeff = bp.coeffs_per_time_step()
    buf = device.mem_alloc(gen.output_bytes(1, 1))
    res = []

    import os
    math_mode = int(os.environ.get("DCS_MATH_MODE", "0"))
    only = os.environ.get("DCS_ONLY", "")

    def run(label, bitwidth, **tuning):
        if only and not label.startswith(only):
            return
        nbytes = gen.output_bytes(bitwidth, 1)
        gen.set_tuning(math_mode=math_mode, **tuning)
        med, mn = timeit(lambda: gen.generate(buf, nbytes, t0=1, nt=1, bitwidth=bitwidth), warm=2, reps=9)
        res.append((ncoeff / med / 1e6, label))
        print(f"{label}: med {med:.3f} ms min {mn:.3f} -> {ncoeff / med / 1e6:.1f} Gcoeff/s {nbytes / med / 1e9:.2f} TB/s", flush=True)

    for bw, name in ((1, "fp32"), (0, "fp16")):
        for tpb in (1, 2, 4):
            for cpb in (4, 8, 12, 16, 20, 24, 32):
                for nts in (0, 1):
                    run(f"{name} tiled tpb={tpb} cpb={cpb:3d} nt={nts}", bw, form=1, tiles_per_block=tpb, chan_per_block=cpb, nontemporal=nts)
        for nw in (4, 8, 16):
            for rpw in (1, 2, 4):
                for xcd in (0, 1):
                    run(f"{name} rows nw={nw:2d} rpw={rpw} xcd={xcd} nt=1", bw, form=2, waves_per_block=nw, rows_per_wave=rpw, xcd_remap=xcd, nontemporal=1)
    res.sort(reverse=True)
    print("TOP")
    for r in res[:30]:
        print(f"  {r[0]:.1f}  {r[1]}")


if __name__ == "__main__":
    main()
