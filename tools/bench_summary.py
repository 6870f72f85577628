"""Print the timing fields of bench.py JSON lines: python tools/bench_summary.py file.json ..."""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        print(f, "unreadable:", e)
        continue
    r = d["roofline"]
    print(f"{f}: value {d['value']:.1f} {d['unit']} n_gpus {d['n_gpus']} geometry {d['config'].get('launch_geometry')} "
          f"ms/step {d['ms_per_step']:.4f} kernel_ms mean {r['kernel_ms']:.4f} median {r.get('kernel_ms_median', 0):.4f} "
          f"min {r.get('kernel_ms_min', 0):.4f} max {r.get('kernel_ms_max', 0):.4f} frac {r['frac']:.4f}")
