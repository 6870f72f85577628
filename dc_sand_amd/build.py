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


def compile_and_link(sources, extra_flags, out: Path, verbose: bool = False) -> None:
    """One ``hipcc -c`` per source, side by side (the translation units are independent: no -fgpu-rdc), then one
    link.  The objects live in a temporary directory; only ``out`` is left in the tree."""
    import tempfile
    from concurrent.futures import ThreadPoolExecutor

    hipcc = _hipcc()
    with tempfile.TemporaryDirectory(prefix="dcs_build_") as tmp:
        def one(src):
            obj = Path(tmp) / (Path(src).stem + ".o")
            cmd = [hipcc, *flags(), *extra_flags, "-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), flush=True)
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                raise RuntimeError(f"hipcc failed ({res.returncode}) on {src}:\n{res.stdout}\n{res.stderr}")
            if verbose and res.stderr.strip():
                print(res.stderr, file=sys.stderr)
            return obj

        with ThreadPoolExecutor(max_workers=min(4, len(sources))) as pool:
            objs = list(pool.map(one, sources))
        cmd = [hipcc, f"--offload-arch={ARCH}", "-fPIC", "-shared", "-o", str(out), *[str(o) for o in objs]]
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc link failed ({res.returncode}):\n{res.stdout}\n{res.stderr}")


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return LIB
    compile_and_link([CSRC / s for s in SOURCES], [], LIB, verbose)
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print(path)
