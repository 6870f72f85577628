import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: long CPU sweep")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; never imported by dc_sand_amd)."""
    from oracle import bf_oracle

    bf_oracle.build()
    return bf_oracle


@pytest.fixture(scope="session")
def dcs_lib():
    """The built C-ABI library (built on demand: hipcc cross-compiles without a GPU)."""
    from dc_sand_amd import build

    build.build()
    from dc_sand_amd import _lib

    return _lib.lib()


@pytest.fixture(scope="session")
def gpu(dcs_lib):
    from dc_sand_amd import device

    if device.device_count() < 1:
        pytest.fail("gpu-marked test on a machine without a HIP device (there is no CPU fallback)")
    device.set_device(0)
    return device


@pytest.fixture(scope="session")
def probes(gpu):
    """probes/libdcs_probes.so (include/dcs_probes.h): device sincos sweep and whole-tensor properties --
    measurement apparatus kept out of the product library."""
    from probes import build as pb, dcs_probes

    pb.build()
    return dcs_probes


def rand_table(n: int, seed: int = 0x5EED, Ts: float = 1e-7) -> np.ndarray:
    """Input set S of SURVEY.md section 8(d): seeded uniform delay polynomials."""
    from oracle.bf_oracle import delay_vals_dtype

    rng = np.random.default_rng(seed)
    d = np.empty(n, dtype=delay_vals_dtype)
    d["fDelay_s"] = rng.uniform(-Ts / 3, Ts / 3, n)
    d["fDelayRate_sps"] = rng.uniform(-2e-6, 2e-6, n)
    d["fPhase_rad"] = rng.uniform(-np.pi, np.pi, n)
    d["fPhaseRate_radps"] = rng.uniform(-3e-6, 3e-6, n)
    return d
