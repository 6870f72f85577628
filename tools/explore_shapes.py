"""Is the default launch geometry near-best for other shapes?  Sweeps the tiled
and rows forms (fp32) for several (ant, beams, chan, nt) and prints the top
geometries plus where the library default lands."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from dc_sand_amd import BeamformerParameters, device  # noqa: E402
from dc_sand_amd.generator import SteeringCoefficientGenerator, simulate_input  # noqa: E402
from explore import timeit  # noqa: E402

SHAPES = [
    ("cfg2 64x64x4096 nt=1", 64, 64, 4096, 1),
    ("default 64x16x64 nt=256", 64, 16, 64, 256),
    ("mid 64x256x8192 nt=1", 64, 256, 8192, 1),
    ("cfg4/GPU 256x512x32768 nt=1", 256, 512, 32768, 1),
    ("cfg3 64x1024x32768 nt=1", 64, 1024, 32768, 1),
    ("narrow 16x16x32768 nt=4", 16, 16, 32768, 4),
]


def main():
    device.require_device()
    device.set_device(0)
    for name, A, B, C, nt in SHAPES:
        bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B)
        gen = SteeringCoefficientGenerator(bp)
        gen.upload_delays(simulate_input(bp))
        nbytes = gen.output_bytes(1, nt)
        ncoeff = bp.coeffs_per_time_step() * nt
        buf = device.mem_alloc(nbytes)
        res = []

        def run(label, **tuning):
            gen.set_tuning(**tuning)
            med, mn = timeit(lambda: gen.generate(buf, nbytes, t0=1, nt=nt), warm=2, reps=7)
            res.append((ncoeff / med / 1e6, label, med))

        run("DEFAULT")
        for tpb in (1, 2, 4):
            for cpb in (4, 8, 16, 32, 64):
                run(f"tiled tpb={tpb} cpb={cpb} nt=1", form=1, tiles_per_block=tpb, chan_per_block=cpb, nontemporal=1)
        for nw in (4, 8):
            for rpw in (1, 2, 4):
                run(f"rows nw={nw} rpw={rpw} xcd=1 nt=1", form=2, waves_per_block=nw, rows_per_wave=rpw, xcd_remap=1, nontemporal=1)
        res.sort(reverse=True)
        dflt = [r for r in res if r[1] == "DEFAULT"][0]
        print(f"== {name}: {nbytes / 2**20:.0f} MiB; default {dflt[0]:.1f} Gcoeff/s ({dflt[2] * 1e3:.1f} us) = {dflt[0] / res[0][0] * 100:.1f}% of best", flush=True)
        for r in res[:5]:
            print(f"   {r[0]:8.1f} Gcoeff/s {r[2] * 1e3:9.1f} us  {r[1]}", flush=True)
        gen.close()
        buf.free()


if __name__ == "__main__":
    main()
