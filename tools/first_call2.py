"""First-call latency by path: python tools/first_call2.py NT [wg_per_cu]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dc_sand_amd import BeamformerParameters, device  # noqa: E402
from dc_sand_amd.generator import SteeringCoefficientGenerator, simulate_input  # noqa: E402

nt = int(sys.argv[1])
device.set_device(0)
bp = BeamformerParameters()
g = SteeringCoefficientGenerator(bp)
if len(sys.argv) > 2:
    g.set_tuning(wg_per_cu=int(sys.argv[2]))
g.upload_delays(simulate_input(bp))
nbytes = g.output_bytes(1, nt)
buf = device.mem_alloc(nbytes)
device.synchronize()
out = []
for i in range(3):
    e0, e1 = device.Event(), device.Event()
    e0.record()
    g.generate(buf, nbytes, t0=0, nt=nt)
    e1.record()
    e1.synchronize()
    out.append(f"{e1.elapsed_ms_since(e0) * 1e3:.1f}")
print(f"nt={nt} wg_per_cu={sys.argv[2] if len(sys.argv) > 2 else 'default'}: calls (us) {' '.join(out)}")
