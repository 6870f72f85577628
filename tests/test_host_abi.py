"""The C-ABI library: it loads, exports every symbol include/dcs_beamformer.h
declares, and its host-only entry points agree with the oracle.  No GPU needed
(no compute call is made)."""
import ctypes
import re
from ctypes import byref
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared_functions():
    text = (ROOT / "include" / "dcs_beamformer.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(dcs_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_symbols_are_exported_and_bound(dcs_lib):
    from dc_sand_amd import _lib

    declared = _declared_functions()
    assert len(declared) >= 40
    bound = {name for name, _, _ in _lib.SIGNATURES}
    for name in declared:
        assert hasattr(dcs_lib, name), f"{name} declared in the header but not exported"
        assert name in bound, f"{name} has no ctypes signature in dc_sand_amd/_lib.py"
    assert bound <= set(declared)
    assert dcs_lib.dcs_abi_version() == 3


def test_product_library_exports_no_measurement_apparatus(dcs_lib):
    """The probes (store patterns, sincos sweep, tensor checksum) live in probes/libdcs_probes.so
    (include/dcs_probes.h), which also carries the -DDCS_PROBES build of the product sources; the product
    library exports none of them."""
    import subprocess

    from dc_sand_amd import _lib
    from probes import build as pb

    syms = subprocess.run(["nm", "-D", "--defined-only", str(_lib.LIB_PATH)], check=True, capture_output=True, text=True).stdout
    assert "probe" not in syms
    exported = {l.split()[-1] for l in syms.splitlines() if " T " in l}
    assert exported == set(_declared_functions()), exported ^ set(_declared_functions())
    plib = pb.build()
    psyms = subprocess.run(["nm", "-D", "--defined-only", str(plib)], check=True, capture_output=True, text=True).stdout
    pexp = {l.split()[-1] for l in psyms.splitlines() if " T " in l}
    text = re.sub(r"/\*.*?\*/", "", (ROOT / "include" / "dcs_probes.h").read_text(), flags=re.S)
    declared_probes = set(re.findall(r"\b(dcs_probe_[a-z0-9_]+)\s*\(", text))
    assert len(declared_probes) == 9 and declared_probes <= pexp
    assert "dcs_probe_set_knobs" in declared_probes and "dcs_probe_set_knobs" not in exported
    assert set(_declared_functions()) <= pexp  # the probes library is a superset build


def test_abi_3_structs_as_a_c_compiler_lays_them_out(dcs_lib, tmp_path):
    """ABI 3: ``struct dcs_bf_tuning`` is ten int32_t with NO measurement field (ABI 2 had ``probe_nomath`` /
    ``probe_pace``; they are ``struct dcs_probe_knobs`` of include/dcs_probes.h now), in the order the Python wrapper
    fills them; ``struct dcs_bf_params`` and ``struct dcs_delay_vals`` as the reference header has them.  Layout taken
    from gcc compiling the public header as plain C, not from this file's idea of it."""
    import subprocess

    from dc_sand_amd.generator import SteeringCoefficientGenerator
    from probes.dcs_probes import KNOB_FIELDS

    hdr = (ROOT / "include" / "dcs_beamformer.h").read_text()
    body = re.search(r"struct dcs_bf_tuning \{(.*?)\n\};", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"int32_t\s+([a-z_0-9]+);", body)
    assert tuple(fields) == SteeringCoefficientGenerator.TUNING_FIELDS and len(fields) == 10
    assert not any("probe" in f for f in fields) and "must be 0" not in hdr
    pbody = re.search(r"struct dcs_probe_knobs \{(.*?)\n\};", (ROOT / "include" / "dcs_probes.h").read_text(), re.S).group(1)
    pfields = re.findall(r"int32_t\s+([a-z_0-9]+);", re.sub(r"/\*.*?\*/", "", pbody, flags=re.S))
    assert tuple(pfields) == KNOB_FIELDS
    src = tmp_path / "layout.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "dcs_probes.h"\n'
        "int main(void) {\n"
        '  printf("%zu %zu %zu %zu\\n", sizeof(struct dcs_bf_tuning), sizeof(struct dcs_bf_params), sizeof(struct dcs_delay_vals), sizeof(struct dcs_probe_knobs));\n'
        + "".join(f'  printf("{f} %zu\\n", offsetof(struct dcs_bf_tuning, {f}));\n' for f in fields)
        + '  printf("abi %d\\n", DCS_BF_ABI_VERSION);\n  return 0;\n}\n'
    )
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Werror", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True,
                   capture_output=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    assert out[0] == f"40 40 16 {4 * len(KNOB_FIELDS)}"
    assert [l.split() for l in out[1:11]] == [[f, str(4 * i)] for i, f in enumerate(fields)]
    assert out[11] == "abi 3"
    # 49 entry points in ABI 2 + the three device-table stream ticks
    assert len(_declared_functions()) == 52


def test_xcd_grouped_workgroup_numbering_is_a_bijection():
    """The matrix-core beamformers renumber their workgroups so that the G workgroups sharing a channel's samples run on one
    XCD (bf_kernels.h: bf_xcd_grouped, the very function the kernels call, evaluated on the host through the probes library):
    for every grid size and group size tried it is a permutation of [0, total), the members of a whole group come from
    dispatch numbers with equal w % 8 (one XCD) and consecutive w / 8 (one after the other), the eight XCDs' q-th groups are
    eight NEIGHBOURING logical groups, and the tail that does not fill 8 G workgroups is left as dispatched."""
    from probes import dcs_probes

    f = dcs_probes.lib().dcs_probe_xcd_grouped
    for G in (1, 2, 3, 4, 5, 16, 64):
        for total in (1, 7, 8, 8 * G - 1, 8 * G, 8 * G + 1, 40 * G + 3, 4096, 4099, 16384 + 5 * G):
            logical = np.array([f(w, total, G) for w in range(total)], dtype=np.int64)
            assert np.array_equal(np.sort(logical), np.arange(total)), (G, total)
            full = total - total % (8 * G)
            assert np.array_equal(logical[full:], np.arange(full, total))
            if G == 1:
                assert np.array_equal(logical, np.arange(total))
                continue
            w_of = np.argsort(logical)  # logical number -> dispatch number
            for grp in range(0, full // G, max(1, full // G // 50)):
                ws = w_of[grp * G:(grp + 1) * G]
                assert len(set(ws % 8)) == 1 and np.array_equal(ws // 8, ws[0] // 8 + np.arange(G)), (G, total, grp)
            for q in range(0, full // (8 * G), max(1, full // (8 * G) // 20)):  # the XCDs' q-th groups: logical groups 8 q .. 8 q + 7
                groups = sorted({int(logical[(q * G) * 8 + x]) // G for x in range(8)})
                assert groups == list(range(8 * q, 8 * q + 8)), (G, total, q)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from dc_sand_amd import _lib

    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.lib()


def test_no_oracle_import_in_product():
    """The product path never touches oracle/ (or the host numerics lab)."""
    for f in (ROOT / "dc_sand_amd").rglob("*"):
        if f.suffix in (".py", ".hip", ".h", ".cpp"):
            txt = f.read_text()
            assert "from oracle" not in txt and "import oracle" not in txt and "bf_oracle" not in txt
            assert "numerics_lab" not in txt


def test_default_params_match_reference_header(dcs_lib):
    from dc_sand_amd.parameters import BeamformerParameters, CParams

    cp = CParams()
    assert dcs_lib.dcs_bf_default_params(byref(cp)) == 0
    ref = BeamformerParameters()  # BeamformerParameters.h:7-17
    assert (cp.nr_channels, cp.nr_stations, cp.nr_beams, cp.nr_samples_per_channel) == (64, 64, 16, 256)
    assert cp.sampling_period == np.float32(1e-7) and cp.fft_size == 8192
    assert cp.adc_sample_rate == 1712e6 and cp.accumulations_before_new_coeffs == 256
    assert ref.to_c().sampling_period == cp.sampling_period


def test_delta_times_and_simulate_input_equal_oracle(dcs_lib, oracle):
    """The product's own host arithmetic (used for every launch) equals the
    oracle's restatement of BeamformerCoefficientTest.cu:299,12-18,185-196."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import delta_times, simulate_input

    for bp in (BeamformerParameters(), BeamformerParameters(NR_CHANNELS=1024, NR_STATIONS=4, NR_BEAMS=2),
               BeamformerParameters(NR_CHANNELS=77, NR_STATIONS=5, NR_BEAMS=3, FFT_SIZE=4096, SAMPLING_PERIOD=2.5e-7)):
        op = oracle.params_from(bp)
        dts = delta_times(bp, 0, 1024)
        exp = np.array([oracle.delta_time(op, t) for t in range(1024)], dtype=np.float32)
        assert np.array_equal(dts.view(np.uint32), exp.view(np.uint32))
        assert np.array_equal(simulate_input(bp).view(np.uint32), oracle.simulate_input(op).view(np.uint32))
    big = delta_times(BeamformerParameters(), 10 ** 6, 4)
    exp = np.array([oracle.delta_time(oracle.params(), 10 ** 6 + i) for i in range(4)], dtype=np.float32)
    assert np.array_equal(big, exp)


def test_ts_diff_equals_the_oracles(dcs_lib, oracle):
    """dcs_bf_ts_diff (what dcs_bf_generate_at / dcs_bf_stream_tick_at feed the kernels) is the verifier's ts_diff,
    BeamformerCoefficientTest.cu:12-18, bit for bit -- seconds boundaries, un-normalised nanoseconds, negative
    differences, epoch-sized seconds."""
    from dc_sand_amd.generator import ts_diff

    rng = np.random.default_rng(12)
    cases = [((10, 0), (10, 819200)), ((10, 999_999_999), (11, 199)), ((10, 999_999_999), (10, 1_000_000_199)),
             ((5, 0), (3, 500_000_000)), ((1_700_000_000, 0), (1_700_000_001, 0)), ((0, 0), (0, 0))]
    for _ in range(2000):
        s0 = int(rng.integers(0, 1 << 25))
        cases.append(((s0, int(rng.integers(0, 10 ** 9))), (s0 + int(rng.integers(-3, 4)), int(rng.integers(0, 2 * 10 ** 9)))))
    for first, last in cases:
        a, b = ts_diff(first, last), oracle.ts_diff(first, last)
        assert a.view(np.uint32) == b.view(np.uint32), (first, last, a, b)
    out = ctypes.c_float()
    assert dcs_lib.dcs_bf_ts_diff(None, None, byref(out)) == -1


def test_output_bytes_and_utilisation(dcs_lib):
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import gpu_utilisation, output_bytes

    bp = BeamformerParameters()
    assert output_bytes(bp, 1, 256) == 256 * 64 * 64 * 16 * 2 * 4  # BeamformerCoefficientTest.cu:37
    assert output_bytes(bp, 0, 256) == 256 * 64 * 64 * 16 * 2 * 2  # :34
    big = BeamformerParameters(NR_CHANNELS=32768, NR_STATIONS=64, NR_BEAMS=1024)
    assert output_bytes(big, 1, 1) == 16 * 2 ** 30
    # BeamformerCoefficientTest.cu:426-430,447-448
    single, multiple = gpu_utilisation(bp, 10.0)
    rate = np.float32(1712e6) / np.float32(8192)
    exp = np.float32((10.0 / 1000.0) / (256 * (np.float32(1) / rate))) * 4
    assert abs(single - exp) <= 1e-6 * exp
    assert abs(multiple - exp / 256) <= 1e-6 * exp


def test_invalid_arguments_are_status_codes(dcs_lib):
    from dc_sand_amd import _lib
    from dc_sand_amd.parameters import BeamformerParameters

    n = ctypes.c_size_t()
    bad = BeamformerParameters(NR_CHANNELS=0).to_c()
    assert dcs_lib.dcs_bf_output_bytes(byref(bad), 1, 1, byref(n)) == _lib.DCS_ERR_INVALID_ARGUMENT
    good = BeamformerParameters().to_c()
    assert dcs_lib.dcs_bf_output_bytes(byref(good), 7, 1, byref(n)) == _lib.DCS_ERR_INVALID_ARGUMENT
    assert dcs_lib.dcs_bf_output_bytes(None, 1, 1, byref(n)) == _lib.DCS_ERR_INVALID_ARGUMENT
    util = (ctypes.c_float * 2)()
    for bad_kw in (dict(NR_SAMPLES_PER_CHANNEL=0), dict(ADC_SAMPLE_RATE=0.0), dict(ACCUMULATIONS_BEFORE_NEW_COEFFS=0)):
        badp = BeamformerParameters(**bad_kw).to_c()  # dcs_bf_gpu_utilisation divides by these
        assert dcs_lib.dcs_bf_gpu_utilisation(byref(badp), 1.0, util) == _lib.DCS_ERR_INVALID_ARGUMENT, bad_kw
    assert b"invalid argument" in dcs_lib.dcs_error_string(_lib.DCS_ERR_INVALID_ARGUMENT)
    assert b"16 bit" in dcs_lib.dcs_error_string(_lib.DCS_ERR_UNSUPPORTED) or b"mode" in dcs_lib.dcs_error_string(_lib.DCS_ERR_UNSUPPORTED)
    assert b"another device" in dcs_lib.dcs_error_string(_lib.DCS_ERR_WRONG_DEVICE)
    cnt = ctypes.c_int(-1)
    assert dcs_lib.dcs_device_count(byref(cnt)) == 0 and cnt.value >= 0
    if cnt.value == 0:  # no GPU here: creating a context is refused, not emulated
        h = ctypes.c_void_p()
        assert dcs_lib.dcs_bf_create(byref(good), byref(h)) == _lib.DCS_ERR_NO_DEVICE
        assert not h.value


def test_no_kernel_uses_scratch(dcs_lib, tmp_path):
    """Every kernel of the gfx950 code object has a zero private segment: scratch costs the first launch of
    a process 130-230 us (the reference's harness times exactly one, cold, launch) and has crept in twice
    through rolled loops over small per-lane arrays.  Read from the built library's own metadata."""
    import shutil
    import subprocess

    from dc_sand_amd import _lib

    llvm = Path("/opt/rocm/lib/llvm/bin")
    tools = [llvm / "llvm-objcopy", llvm / "clang-offload-bundler", llvm / "llvm-readelf"]
    if not all(t.exists() for t in tools):
        tools = [Path(p) for p in (shutil.which("llvm-objcopy"), shutil.which("clang-offload-bundler"), shutil.which("llvm-readelf")) if p]
    if len(tools) != 3:
        pytest.skip("LLVM binutils of the ROCm toolchain not found")
    from probes import build as pb

    for tag, so, least in (("product", _lib.LIB_PATH, 100), ("probes", pb.build(), 120)):
        fat = tmp_path / f"{tag}.fatbin"
        subprocess.run([str(tools[0]), f"--dump-section=.hip_fatbin={fat}", str(so)], check=True, capture_output=True)
        # one offload bundle per translation unit, concatenated (each padded): split at the magic
        blob, magic = fat.read_bytes(), b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        names, sizes = [], []
        for i, a in enumerate(starts):
            piece, co = tmp_path / f"{tag}.{i}.bundle", tmp_path / f"{tag}.{i}.co"
            piece.write_bytes(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            subprocess.run([str(tools[1]), "--unbundle", "--type=o", f"--input={piece}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            f"--output={co}"], check=True, capture_output=True)
            if co.stat().st_size == 0:
                continue  # a translation unit without device code (the C-ABI file)
            notes = subprocess.run([str(tools[2]), "--notes", str(co)], check=True, capture_output=True, text=True).stdout
            names += [l.split(":", 1)[1].strip() for l in notes.splitlines() if l.strip().startswith(".name:")]
            sizes += [int(l.split(":", 1)[1]) for l in notes.splitlines() if ".private_segment_fixed_size:" in l]
        assert len(sizes) > least, f"{tag}: metadata not found ({len(sizes)} kernels in {len(starts)} bundles)"
        bad = [(n, s) for n, s in zip(names[: len(sizes)], sizes) if s != 0]
        assert not any(s for s in sizes), f"{tag}: {sum(1 for s in sizes if s)} kernel(s) use scratch: {bad[:3]}"


def test_makefile_and_build_py_use_the_same_flags():
    """Two ways to build the library (``make`` / ``python -m dc_sand_amd.build``), one set of flags:
    -ffp-contract=off and friends are part of the numerical contract."""
    from dc_sand_amd import build

    mk = (Path(__file__).resolve().parent.parent / "Makefile").read_text()
    flags = re.search(r"HIPFLAGS := (.*?)\n\n", mk, re.S).group(1).replace("\\\n", " ").split()
    assert sorted(flags) == sorted(build.flags())
