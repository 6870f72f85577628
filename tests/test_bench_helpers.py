"""bench.py pieces that run without a GPU: the cpu_baseline leg (oracle timed on a
bounded slab) and the committed PMC traffic file it reads."""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent


def test_cpu_baseline_leg_small_shape(oracle):
    sys.path.insert(0, str(ROOT))
    import bench
    from dc_sand_amd import BeamformerParameters

    bp = BeamformerParameters(NR_CHANNELS=256, NR_STATIONS=8, NR_BEAMS=16)
    table = oracle.simulate_input(oracle.params_from(bp))
    out = bench.cpu_baseline(bp, table, seconds=0.2)
    assert out["unit"] == "Gcoeff/s" and out["cores"] == 1 and out["kind"] == "port"
    assert out["value"] > 0 and "channels [0," in out["sample"]


def test_committed_pmc_traffic_matches_workload():
    sys.path.insert(0, str(ROOT))
    import bench

    d = json.loads((ROOT / "profiles" / "pmc_write_size.json").read_text())
    algo = 8 * 64 * 1024 * 32768
    assert d["algorithmic_bytes_per_launch"] == algo
    t, src = bench.pmc_traffic(algo)
    assert t is not None and 0.99 < t / algo < 1.05  # measured HBM bytes ~ algorithmic bytes
    assert "pmc_write_size.json" in src and "rocprofv3 --pmc" in src  # the JSON line says where the number comes from
    assert bench.pmc_traffic(algo // 2) == (None, None)  # another workload: no number is invented
