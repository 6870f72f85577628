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


def test_gpus_n_without_ranks_is_an_error_not_a_one_gpu_line():
    """bench.py's launch contract, the part that needs no GPU: ``--gpus 2`` without a launcher starts two rank
    processes; here neither finds a HIP device, so the command exits non-zero and prints NO JSON line (round 2's
    bench.py would have measured one GPU and printed ``n_gpus: 1``).  Also: a launcher environment whose WORLD_SIZE
    disagrees with --gpus is refused."""
    import os
    import subprocess

    import pytest
    import torch

    if torch.cuda.device_count() >= 1:
        pytest.skip("a GPU is present: tests/test_bench_contract.py covers this case there")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], capture_output=True,
                         text=True, env=env, timeout=600)
    assert res.returncode != 0 and res.stdout.strip() == ""
    assert "no result line" in res.stderr and "rank" in res.stderr
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8"], capture_output=True, text=True,
                         env=dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"), timeout=600)
    assert res.returncode != 0 and res.stdout.strip() == "" and "WORLD_SIZE=2" in res.stderr


def test_named_configs_are_the_baseline_shapes():
    sys.path.insert(0, str(ROOT))
    import bench

    assert bench.CONFIGS["cfg3"]["ant"] * bench.CONFIGS["cfg3"]["beams_per_gpu"] * bench.CONFIGS["cfg3"]["chan"] == 64 * 1024 * 32768
    c4 = bench.CONFIGS["cfg4"]
    assert (c4["ant"], c4["beams_per_gpu"] * 8, c4["chan"]) == (256, 4096, 32768)  # BASELINE configs[3] over 8 GPUs
