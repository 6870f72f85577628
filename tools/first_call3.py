"""What is cold in the first launch of a process: the kernel code or the output buffer's pages?"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dc_sand_amd import BeamformerParameters, device  # noqa: E402
from dc_sand_amd.generator import SteeringCoefficientGenerator, simulate_input  # noqa: E402

mode = sys.argv[1]
device.set_device(0)
bp = BeamformerParameters()
g = SteeringCoefficientGenerator(bp)
g.upload_delays(simulate_input(bp))
nbytes = g.output_bytes(1, 256)
buf = device.mem_alloc(nbytes)
small = device.mem_alloc(g.output_bytes(1, 1))
device.synchronize()
if mode == "warm-kernel":  # same kernel symbol, another (small) buffer
    g.generate(small, small.nbytes, t0=0, nt=1)
    device.synchronize()
elif mode == "submit-after-alloc":  # any submission after the allocations, not touching the big buffer
    device.memset(small, 0, 256)
    device.synchronize()
elif mode == "warm-buffer":  # touch the big buffer with a fill, not with the kernel
    device.memset(buf, 0, nbytes)
    device.synchronize()
out = []
for i in range(3):
    e0, e1 = device.Event(), device.Event()
    e0.record(); g.generate(buf, nbytes, t0=0, nt=256); e1.record(); e1.synchronize()
    out.append(f"{e1.elapsed_ms_since(e0) * 1e3:.1f}")
print(f"{mode}: calls (us) {' '.join(out)}")
