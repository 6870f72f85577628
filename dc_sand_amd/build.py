"""Build driver: compiles the gfx950 kernels and the C-ABI into
``dc_sand_amd/csrc/libdcs_beamformer.so`` with hipcc (in-tree, so the library
travels with the source tree).  ``python -m dc_sand_amd.build [--force]``.

Flags that are part of the numerical contract (DESIGN.md "numerics"):
  -ffp-contract=off   no fused multiply-add except where bf_math.h writes one;
  (default)           -fhip-fp32-correctly-rounded-divide-sqrt stays on;
  (default)           fp32 denormals are kept (no -fgpu-flush-denormals-to-zero).
Performance-only: -fno-slp-vectorize (scalar fp32 VALU ops instead of v_pk_*_f32).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB = CSRC / "libdcs_beamformer.so"
SOURCES = ["bf_kernels.hip", "bf_beamform_mfma.hip", "bf_capi.hip"]
HEADERS = ["bf_kernels.h", "bf_math.h", "bf_device.h", "../../include/dcs_beamformer.h"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def flags() -> list[str]:
    extra = os.environ.get("DCS_HIPCC_EXTRA", "").split()
    return [
        f"--offload-arch={ARCH}",
        "-O3",
        "-std=c++17",
        "-ffp-contract=off",
        "-fno-fast-math",
        "-fno-slp-vectorize",  # v_pk_*_f32 packing costs 1.3 % (fp32) / 6 % (fp16) here: profiles/r01_geometry_sweep.md
        "-fPIC",
        "-fvisibility=hidden",  # exported: exactly what include/dcs_beamformer.h declares
        "-Wall",
        "-Wextra",
        "-Wno-unused-parameter",
        *extra,
    ]


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = [CSRC / s for s in SOURCES + HEADERS] + [Path(__file__)]
    return any(d.resolve().stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), *flags(), "-shared", "-o", str(LIB), *[str(CSRC / s) for s in SOURCES]]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed ({res.returncode}):\n{res.stdout}\n{res.stderr}")
    if verbose and res.stderr.strip():
        print(res.stderr, file=sys.stderr)
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print(path)
