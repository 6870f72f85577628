"""bench.py's output contract on a real GPU: exactly ONE line on stdout, a JSON object with
the driver's fields plus ``roofline`` (and ``cpu_baseline`` at N = 1), on a small shape so the
two runs take seconds.  The second run drives the N > 1 control flow (process group, per-step
asynchronous broadcast one step ahead, double-buffered table, slice gather) over RCCL with a world of
one rank -- the only way to exercise it on a one-GPU box."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
SMALL = ["--ant", "16", "--beams-per-gpu", "64", "--chan", "2048", "--steps", "5", "--warmup", "2"]


def _free_port() -> str:
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


def _run(extra):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):  # a plain single-process run
        env.pop(k, None)
    env["MASTER_PORT"] = _free_port()
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), *SMALL, *extra], capture_output=True, text=True, env=env,
                         timeout=600, cwd=str(ROOT))
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one line, got {len(lines)}: {res.stdout[:500]}"
    return json.loads(lines[0])


def _common(d, n_gpus=1):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["unit"] == "Gcoeff/s" and d["n_gpus"] == n_gpus and d["steps"] == 5 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["traffic"] is None and r["traffic_source"] is None  # PMC traffic is committed for the headline workload only; nothing is invented
    pr = d["per_rank"]
    assert len(pr) == n_gpus and sorted(x["rank"] for x in pr) == list(range(n_gpus))
    for x in pr:
        assert x["kernel_ms"] > 0 and x["wall_ms_per_step"] > 0 and x["geometry"]
    assert "rccl_world_size" in d
    # value = coefficients of all ranks' steps / wall time; rank 0's kernel-only rate cannot be lower than its share
    assert 0 < d["value"] * 8 / n_gpus <= r["achieved"] * 1.001


def test_single_gpu_line_with_cpu_baseline():
    d = _run(["--cpu-seconds", "0.3", "--sustain-seconds", "0.4"])
    _common(d)
    assert d["config"]["collective"] == "none" and d["rccl_world_size"] is None
    am = d["also_measured"]
    su = am["sustained"]
    assert su["seconds"] >= 0.4 and su["value"] > 0 and su["launches"] >= 50 and 0 < su["frac_of_hbm_peak"] < 1
    st = am["streaming_cfg5"]
    assert st["cadence_target_us"] == 200.0 and st["full_tensor_period_us"] > 0
    assert st["largest_slab_at_200us"] is None or st["largest_slab_at_200us"]["period_us"] <= 200.0
    ev = st["new_table_every_tick"]
    assert ev["full_tensor_host_table_period_us"] > 0 and ev["full_tensor_device_table_period_us"] > 0
    assert am["fp16_output"]["math_mode"] == 0 and am["fp16_output_b16_arithmetic"]["math_mode"] == 4
    assert am["fused_generate_and_beamform"]["value"] > 0
    ba = am["beamform_accumulated"]
    assert len(ba) == 3 and all(x["value"] > 0 and 0 < x["frac_of_hbm_peak"] < 1 and x["unit"] == "T coefficient-products/s" for x in ba)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "Gcoeff/s" and c["value"] > 0 and c["sample"]
    assert c["gpu_vs_oracle_spot_check"]["over_1ulp"] == 0
    # every side measurement's last output is checked against the oracle too (in the cpu_baseline leg)
    xs = c["extras_vs_oracle"]
    assert len(xs) >= 9 and all(x["ok"] for x in xs) and c["extras_vs_oracle_all_ok"] is True, [x for x in xs if not x["ok"]]
    assert "error" not in am
    assert any("device memory" in x["item"] for x in xs) and any(x["item"].startswith("beamform_accumulated 256x") for x in xs)
    assert c["beamform_accumulated"]["value"] > 0 and c["beamform_accumulated"]["cores"] == 1


def test_collective_control_flow_over_rccl_world_of_one():
    d = _run(["--force-collective", "--no-cpu-baseline", "--check-all-ranks", "--no-extras"])
    _common(d)
    assert d["config"]["collective"] == "RCCL broadcast" and d["rccl_world_size"] == 1
    assert "cpu_baseline" not in d


def test_gpus_2_without_a_launcher_starts_two_ranks():
    """The launch contract (bench.py docstring): plain ``python bench.py --gpus 2`` -- the form of the driver's recorded
    N = 1 command with another N -- starts TWO rank processes itself and relays rank 0's line; it can never print an
    ``n_gpus: 1`` line for a ``--gpus 2`` request.  Rehearsed with both ranks on GPU 0 over gloo."""
    d = _run(["--gpus", "2", "--backend", "gloo", "--shared-device", "--check-all-ranks", "--no-cpu-baseline", "--no-extras"])
    _common(d, n_gpus=2)
    assert d["n_gpus"] == 2 and len(d["per_rank"]) == 2
    assert "REHEARSAL" in d["config"]["collective"]
    assert d["config"]["coeffs_per_step"] == 2 * 16 * 64 * 2048


def test_gpus_n_with_fewer_devices_prints_no_line_and_fails():
    """``--gpus 2`` over RCCL on a ONE-GPU box: rank 1 has no GPU; the command exits non-zero and stdout stays empty."""
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", *SMALL, "--no-cpu-baseline", "--no-extras"],
                         capture_output=True, text=True, env=env, timeout=600, cwd=str(ROOT))
    assert res.returncode != 0
    assert res.stdout.strip() == "", res.stdout[:300]
    assert "no result line" in res.stderr


def test_streaming_step_with_the_table_gathered_in_graph():
    """``--streaming``: each step is a hipGraph replay fed from the broadcast's device buffer (configs[3] + configs[4]),
    over RCCL with a world of one rank; the last step is spot-checked against the oracle by every rank."""
    d = _run(["--force-collective", "--streaming", "--no-cpu-baseline", "--check-all-ranks", "--no-extras"])
    _common(d)
    assert "dcs_bf_stream_tick_dt_from_global" in d["config"]["kernel"] and d["rccl_world_size"] == 1


def test_two_ranks_streaming_from_the_broadcast_buffer_over_gloo():
    """configs[3] + configs[4] with a REAL inter-process broadcast (gloo; both ranks on GPU 0: a rehearsal): every step a graph
    replay whose table is the buffer the broadcast has just filled, each rank gathering its own beam slice in-graph and
    checking its slab against the oracle; started by bench.py itself (no launcher)."""
    d = _run(["--gpus", "2", "--backend", "gloo", "--shared-device", "--streaming", "--check-all-ranks", "--no-cpu-baseline", "--no-extras"])
    _common(d, n_gpus=2)
    assert "dcs_bf_stream_tick_dt_from_global" in d["config"]["kernel"] and "gloo" in d["config"]["collective"]


def test_named_config_reaches_the_workload_field():
    d = _run(["--config", "cfg4", "--no-cpu-baseline", "--no-extras"])
    assert d["config"]["workload"].startswith("cfg4 (shape overridden)") and "configs[3]" in d["config"]["named_config"]


def test_two_ranks_sharing_the_gpu_over_gloo():
    """The N = 2 control flow with a real inter-process broadcast (gloo; both ranks on GPU 0 -- a rehearsal, never
    a result): launched exactly as the driver launches N > 1, every rank checks its own beam slab against the
    oracle, rank 0 alone prints the line."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), str(ROOT / "bench.py"), "--gpus", "2", *SMALL, "--backend", "gloo", "--shared-device",
           "--check-all-ranks", "--no-cpu-baseline", "--no-extras"]
    res = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=str(ROOT))
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout[:500]
    d = json.loads(lines[0])
    _common(d, n_gpus=2)
    assert "gloo" in d["config"]["collective"] and "REHEARSAL" in d["config"]["collective"] and d["rccl_world_size"] is None
    assert d["config"]["coeffs_per_step"] == 2 * 16 * 64 * 2048
