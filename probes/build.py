"""Builds ``probes/libdcs_probes.so``: the probe kernels (bf_probes.hip) plus a ``-DDCS_PROBES``
build of the product sources, with the flags of :mod:`dc_sand_amd.build` (the numerical contract is
the same).  ``python -m probes.build [--force]``.  Test / measurement infrastructure only."""
from __future__ import annotations

import subprocess
import sys
from pathlib import Path

from dc_sand_amd import build as product

HERE = Path(__file__).resolve().parent
LIB = HERE / "libdcs_probes.so"
SOURCES = [HERE / "bf_probes.hip", *[product.CSRC / s for s in product.SOURCES]]
DEPS = SOURCES + [product.CSRC / h for h in product.HEADERS] + [HERE.parent / "include" / "dcs_probes.h", Path(__file__)]


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(d.resolve().stat().st_mtime > t for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return LIB
    product.compile_and_link(SOURCES, ["-DDCS_PROBES"], LIB, verbose)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
