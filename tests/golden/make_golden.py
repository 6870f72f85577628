"""Writes the committed fixtures of tests/golden/ from the CPU oracle.

The reference stores no golden vectors for the steering-coefficient path and
cannot be compiled in this image (CUDA headers absent), so these vectors come
from this repo's restatement of its verifier ("parity unpinned"); they exist to
freeze that restatement and to give the GPU tests expected outputs that do not
depend on the oracle library at run time.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
sys.path.insert(0, str(HERE.parent))

from conftest import rand_table  # noqa: E402
from oracle import bf_oracle as o  # noqa: E402

T = [0, 1, 5, 7, 9, 18, 255]


def main():
    cases = []
    # config 1: 4 ant x 2 beam x 1024 chan, full tensor for the parity time steps (64 KiB each)
    C, A, B = 1024, 4, 2
    p = o.params(C, A, B)
    d = o.simulate_input(p)
    arrays = {"delays": d.view(np.float32).reshape(-1, 4)}
    slabs = []
    for t in T:
        key = f"t{t}"
        arrays[key] = o.generate(p, d, t, 1)
        slabs.append([t, 0, C, key])
    np.savez_compressed(HERE / "config1_4x2x1024.npz", **arrays)
    cases.append(dict(file="config1_4x2x1024.npz", C=C, A=A, B=B, input="ramp", slabs=slabs))

    # reference default shape (64 chan x 64 ant x 16 beams): t = 0, 9, 255 in full + checksums of others
    C, A, B = 64, 64, 16
    p = o.params(C, A, B)
    d = o.simulate_input(p)
    arrays = {"delays": d.view(np.float32).reshape(-1, 4)}
    slabs = []
    for t in (0, 9, 255):
        key = f"t{t}"
        arrays[key] = o.generate(p, d, t, 1)
        slabs.append([t, 0, C, key])
    cks = [[t, o.generate_checksum(p, d, t, 1, 0, C, 1)[1]] for t in range(0, 256, 17)]
    np.savez_compressed(HERE / "reference_default_64x16x64.npz", **arrays)
    cases.append(dict(file="reference_default_64x16x64.npz", C=C, A=A, B=B, input="ramp", slabs=slabs, checksums=cks))

    # config 2 (64 x 64 x 4096): sampled channel slabs + whole-step checksums, seeded input
    C, A, B = 4096, 64, 64
    p = o.params(C, A, B)
    d = rand_table(A * B, seed=0x5EED)
    arrays = {"delays": d.view(np.float32).reshape(-1, 4)}
    slabs = []
    for t in (1, 9):
        for c0 in (0, 2047, 4094):
            key = f"t{t}_c{c0}"
            arrays[key] = o.generate(p, d, t, 1, c0, 2)
            slabs.append([t, c0, 2, key])
    cks = [[t, o.generate_checksum(p, d, t, 1, 0, C, 4)[1]] for t in (1, 9)]
    np.savez_compressed(HERE / "config2_64x64x4096_seeded.npz", **arrays)
    cases.append(dict(file="config2_64x64x4096_seeded.npz", C=C, A=A, B=B, input="seeded", seed=0x5EED, slabs=slabs, checksums=cks))

    # The one fixture that comes from the REFERENCE ITSELF: its data-contract header compiled where it lies
    # (oracle/ref_contract.cpp, `make -C oracle ref`); regenerated only where the reference tree is present.
    contract = None
    ref_bin = HERE.parent.parent / "oracle" / "_ref" / "ref_contract"
    if Path("/root/reference/beamformer_coefficient_generator/BeamformerParameters.h").exists():
        import subprocess

        subprocess.run(["make", "-C", str(HERE.parent.parent / "oracle"), "ref"], check=True, capture_output=True)
        (HERE / "reference_contract.json").write_text(subprocess.run([str(ref_bin)], check=True, capture_output=True, text=True).stdout)
    if (HERE / "reference_contract.json").exists():
        contract = dict(file="reference_contract.json", pinned_by="the reference's own header, compiled (oracle/ref_contract.cpp)",
                        covers="struct delay_vals layout and the compile-time constants (SURVEY 8 row a0) -- not the arithmetic")
    (HERE / "manifest.json").write_text(json.dumps(dict(parity="unpinned", generator="tests/golden/make_golden.py", cases=cases,
                                                        reference_contract=contract), indent=1))
    for f in sorted(HERE.glob("*.npz")):
        print(f.name, f.stat().st_size)


if __name__ == "__main__":
    main()
