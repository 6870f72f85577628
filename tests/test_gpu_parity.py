"""GPU parity: the HIP path, called through the C-ABI, against the CPU oracle.

Bar: every fp32 element within 1 ULP of the oracle (ordered-integer distance of
the bit patterns), and the reference's own rule |gpu - cpu| <= 1e-4
(runBeamformerTests.cpp:30,61).  The oracle's canonical reading of ``cos(float)`` is
double-then-round; configs 1-2 and the reference's default tensor are also held to
1 ULP of its float-libm reading (``_check_float_reading``).  Time indices 5, 7, 9, 18 are those where the
reference's three dt derivations disagree.
"""
import numpy as np
import pytest

from conftest import rand_table

pytestmark = pytest.mark.gpu

PARITY_T = [0, 1, 5, 7, 9, 18, 255]


def _gen(gpu, bp, table, t0, nt, kernel=2, bitwidth=1, c0=None, nc=None, tuning=None):
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    g = SteeringCoefficientGenerator(bp)
    if tuning:
        g.set_tuning(**tuning)
    g.upload_delays(table)
    eb = 4 if bitwidth == 0 else 8
    if c0 is None:
        nbytes = nt * bp.NR_CHANNELS * bp.n_pairs * eb
        shape = (nt, bp.NR_CHANNELS, bp.NR_STATIONS, bp.NR_BEAMS, 2)
    else:
        nbytes = nt * nc * bp.n_pairs * eb
        shape = (nt, nc, bp.NR_STATIONS, bp.NR_BEAMS, 2)
    buf = gpu.mem_alloc(nbytes + 256)
    gpu.memset(buf, 0xFF, nbytes + 256)  # NaN canary, also past the end
    if c0 is None:
        g.generate(buf, nbytes, t0=t0, nt=nt, kernel=kernel, bitwidth=bitwidth)
    else:
        g.generate_slab(buf, nbytes, c0, nc, t0=t0, nt=nt, bitwidth=bitwidth)
    host = np.empty(nbytes + 256, dtype=np.uint8)
    gpu.memcpy_dtoh(host, buf)
    assert np.all(host[nbytes:] == 0xFF), "wrote past the end of the output tensor"
    dt = np.float16 if bitwidth == 0 else np.float32
    out = host[:nbytes].view(dt).reshape(shape)
    g.close()
    buf.free()
    return out


def _check(oracle, got, exp, tol=1e-4):
    mx, n_over, first = oracle.max_ulp(got, exp, 1)
    assert n_over == 0, f"max ULP {mx}, {n_over} elements over 1 ULP, first flat index {first}"
    assert oracle.compare(got, exp, tol) == -1
    return mx


def _check_float_reading(oracle, got, op, table, t0, nt, limit=1):
    """The verifier's other reading (oracle/bf_oracle.c): ``cosf`` / ``sinf`` of the host libm, which is what
    nvcc's headers bind ``cos(fRotation)`` to.  Proven <= 1 ULP for every argument below 256 (and for the
    low-degree set below 512) against this glibc in tests/test_numerics.py; here on the tensor."""
    with oracle.trig_reading(oracle.FLOAT_LIBM):
        exp = oracle.generate(op, table, t0, nt)
    mx, n_over, first = oracle.max_ulp(got, exp, limit)
    assert n_over == 0, f"float-libm reading: max ULP {mx}, {n_over} elements over {limit} ULP, first flat index {first}"
    assert oracle.compare(got, exp, 1e-4) == -1
    return mx


def _compare_every_element(gpu, oracle, d_buf, op, table, dt, nc_total, n_pairs, readings=(0, 1), slab_bytes=1 << 30, half=False):
    """verify_output at full size (BeamformerCoefficientTest.cu:348-357 compares EVERY element): the device tensor
    [nc_total][n_pairs][2] fp32 of ONE time step comes back in <= 1 GiB slabs through a pinned buffer and each slab
    is compared with the verifier generated on the fly over all host cores (oracle.compare_generated).  Returns
    {reading: dict(hist, max_ulp, first_over_1ulp, seconds)} accumulated over the slabs.  ``half``: the packed binary16
    output, compared as bit patterns with RN-even(verifier's fp32), distances in binary16 ulps."""
    import os
    import time

    row = n_pairs * (4 if half else 8)
    per = max(1, slab_bytes // row)
    nthreads = max(1, min(64, len(os.sched_getaffinity(0))))
    pinned = gpu.pagelocked_empty(per * n_pairs * 2, np.uint16 if half else np.float32)
    tot = {r: dict(hist=[0, 0, 0, 0], max_ulp=0, first_over_1ulp=-1, seconds=0.0) for r in readings}
    t_copy = 0.0
    for c0 in range(0, nc_total, per):
        nc = min(per, nc_total - c0)
        view = pinned[: nc * n_pairs * 2]
        t0 = time.perf_counter()
        gpu.memcpy_dtoh(view, int(d_buf) + c0 * row)
        t_copy += time.perf_counter() - t0
        for r in readings:
            res = oracle.compare_generated(op, table, [dt], c0, nc, view, nthreads=nthreads, reading=r)
            acc = tot[r]
            acc["hist"] = [a + b for a, b in zip(acc["hist"], res["hist"])]
            acc["max_ulp"] = max(acc["max_ulp"], res["max_ulp"])
            if acc["first_over_1ulp"] < 0 and res["first_over_1ulp"] >= 0:
                acc["first_over_1ulp"] = c0 * n_pairs * 2 + res["first_over_1ulp"]
            acc["seconds"] += res["seconds"]
    tot["copy_seconds"] = t_copy
    tot["threads"] = nthreads
    return tot


@pytest.mark.parametrize("kernel", [0, 1, 2])
def test_config1_4ant_2beam_1024chan(gpu, oracle, kernel):
    """BASELINE configs[0]: 4 ant x 2 beam x 1024 chan, reference ramp input."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import simulate_input

    bp = BeamformerParameters(NR_CHANNELS=1024, NR_STATIONS=4, NR_BEAMS=2)
    table = simulate_input(bp)
    op = oracle.params_from(bp)
    assert np.array_equal(table.view(np.uint32), oracle.simulate_input(op).view(np.uint32))
    for t in PARITY_T:
        got = _gen(gpu, bp, table, t, 1, kernel=kernel)
        _check(oracle, got, oracle.generate(op, table, t, 1))
        _check_float_reading(oracle, got, op, table, t, 1)


@pytest.mark.parametrize("kernel", [0, 1, 2])
def test_reference_default_shape_all_256_time_steps(gpu, oracle, kernel):
    """BeamformerParameters.h defaults (64 chan, 64 ant, 16 beams, 256 samples):
    the tensor runBeamformerTests verifies, every element."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import simulate_input

    bp = BeamformerParameters()
    table = simulate_input(bp)
    op = oracle.params_from(bp)
    got = _gen(gpu, bp, table, 0, 256, kernel=kernel)
    _check(oracle, got, oracle.generate(op, table, 0, 256))
    _check_float_reading(oracle, got, op, table, 0, 256)


@pytest.mark.parametrize("seeded", [False, True])
def test_config2_64ant_64beam_4096chan(gpu, oracle, seeded):
    """BASELINE configs[1]: single launch, every element compared."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import simulate_input

    bp = BeamformerParameters(NR_CHANNELS=4096, NR_STATIONS=64, NR_BEAMS=64)
    table = rand_table(bp.n_pairs) if seeded else simulate_input(bp)
    op = oracle.params_from(bp)
    for t in (1, 9):
        got = _gen(gpu, bp, table, t, 1)
        _check(oracle, got, oracle.generate(op, table, t, 1))
        _check_float_reading(oracle, got, op, table, t, 1)


TUNINGS = [
    dict(form=1, tiles_per_block=1, nontemporal=1),
    dict(form=1, tiles_per_block=2, nontemporal=0),
    dict(form=1, tiles_per_block=4, chan_per_block=8, nontemporal=1),
    dict(form=1, tiles_per_block=1, chan_per_block=3),
    dict(form=1, tiles_per_block=4, chan_per_block=1000),
    dict(form=1, tiles_per_block=1, chan_per_block=13, xcd_remap=1),
    dict(form=1, tiles_per_block=1, chan_per_block=12, wg_per_cu=-1),
    dict(form=1, tiles_per_block=2, chan_per_block=5, wg_per_cu=2),
    dict(form=1, tiles_per_block=4, chan_per_block=9, wg_per_cu=7),
    dict(form=3),
    dict(form=3, tiles_per_block=2, chan_per_block=5, wg_per_cu=-1),
    dict(form=3, tiles_per_block=4, chan_per_block=3, wg_per_cu=7),
    dict(form=3, tiles_per_block=1, chan_per_block=16, nontemporal=0),
    dict(form=2),
    dict(form=2, waves_per_block=4, rows_per_wave=1, rows_same_tile=0),
    dict(form=2, waves_per_block=4, rows_per_wave=2, nontemporal=1, rows_same_tile=0),
    dict(form=2, waves_per_block=8, rows_per_wave=4, xcd_remap=1, rows_same_tile=0),
    dict(form=2, waves_per_block=16, rows_per_wave=1, xcd_remap=1, nontemporal=1, rows_same_tile=0),
    dict(form=2, waves_per_block=16, rows_per_wave=4, rows_same_tile=0),
    dict(form=2, waves_per_block=4, rows_per_wave=3, rows_same_tile=1),
    dict(form=2, waves_per_block=4, rows_per_wave=3, rows_same_tile=1, wg_per_cu=4),
    dict(form=2, waves_per_block=8, rows_per_wave=1, rows_same_tile=1, xcd_remap=0),
]


@pytest.mark.parametrize("tuning", TUNINGS, ids=lambda d: "-".join(f"{k}{v}" for k, v in d.items()))
@pytest.mark.parametrize("bitwidth", [1, 0])
def test_tunings_agree(gpu, oracle, tuning, bitwidth):
    """Every form and launch geometry gives the same bits (ragged channel
    blocks, partial tiles and odd row groups included)."""
    from dc_sand_amd import BeamformerParameters

    for (A, B, C) in ((7, 38, 301), (3, 5, 17), (64, 16, 64)):
        bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B)
        table = rand_table(bp.n_pairs, seed=7)
        op = oracle.params_from(bp)
        got = _gen(gpu, bp, table, 3, 3, tuning=tuning, bitwidth=bitwidth)
        exp = oracle.generate(op, table, 3, 3)
        if bitwidth == 1:
            _check(oracle, got, exp)
        else:
            base = _gen(gpu, bp, table, 3, 3, bitwidth=0)
            assert np.array_equal(got.view(np.uint16), base.view(np.uint16))


@pytest.mark.parametrize("A,B,C", [(1, 1, 1), (3, 5, 17), (5, 3, 64), (1, 129, 33), (64, 1, 5), (2, 257, 9)])
@pytest.mark.parametrize("kernel", [0, 1, 2])
def test_ragged_and_odd_shapes(gpu, oracle, A, B, C, kernel):
    """Odd pair counts (8-byte stores), partial tiles, single elements; the
    three kernel options cover the naive, tiled and rows forms."""
    from dc_sand_amd import BeamformerParameters

    bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B)
    table = rand_table(bp.n_pairs, seed=A * 1000 + B)
    op = oracle.params_from(bp)
    got = _gen(gpu, bp, table, 7, 3, kernel=kernel)
    _check(oracle, got, oracle.generate(op, table, 7, 3))


def test_channel_slab(gpu, oracle):
    from dc_sand_amd import BeamformerParameters

    bp = BeamformerParameters(NR_CHANNELS=32768, NR_STATIONS=4, NR_BEAMS=32)
    table = rand_table(bp.n_pairs, seed=11)
    op = oracle.params_from(bp)
    for c0, nc in ((0, 5), (32000, 768), (16383, 2)):
        got = _gen(gpu, bp, table, 18, 2, c0=c0, nc=nc)
        _check(oracle, got, oracle.generate(op, table, 18, 2, c0, nc))


def test_slow_path_large_rotation_and_extreme_rates(gpu, oracle):
    """Pairs outside the fast path's proven range (|rotation| >= 32000, tiny or
    huge rate terms, zero rate) take the IEEE-divide + fp64-sincos branch."""
    from dc_sand_amd import BeamformerParameters

    bp = BeamformerParameters(NR_CHANNELS=512, NR_STATIONS=2, NR_BEAMS=128)
    table = rand_table(bp.n_pairs, seed=3)
    table["fDelayRate_sps"][5] = 1e-2      # rotation up to ~3e5 rad
    table["fDelayRate_sps"][6] = 1e-30     # below dcs_div_const's range
    table["fDelayRate_sps"][7] = 0.0
    table["fPhase_rad"][130] = 5e4
    table["fDelayRate_sps"][200] = -3.0    # huge
    table["fDelay_s"][201] = 1.0
    op = oracle.params_from(bp)
    for t in (0, 9):
        for form in (1, 2, 3):
            got = _gen(gpu, bp, table, t, 1, tuning=dict(form=form))
            _check(oracle, got, oracle.generate(op, table, t, 1), tol=1e-4)


def test_fp16_output(gpu, oracle):
    """b16: packed half2, RN-even of the fp32 coefficient.  The reference never
    verifies this mode (BeamformerCoefficientTest.cu:282-287): parity unpinned;
    the expectation is RN-even(oracle fp32), tolerance 1 half-ULP."""
    from dc_sand_amd import BeamformerParameters

    for (A, B, C) in ((64, 16, 64), (3, 5, 17), (2, 130, 9)):
        bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B)
        table = rand_table(bp.n_pairs, seed=5)
        op = oracle.params_from(bp)
        got = _gen(gpu, bp, table, 5, 2, kernel=2, bitwidth=0)
        exp = oracle.generate(op, table, 5, 2).astype(np.float16)
        gi = got.view(np.int16).astype(np.int32)
        ei = exp.view(np.int16).astype(np.int32)
        gi = np.where(gi < 0, -(gi & 0x7FFF), gi)
        ei = np.where(ei < 0, -(ei & 0x7FFF), ei)
        assert np.max(np.abs(gi - ei)) <= 1
        # MULTIPLE_CHANNELS supports b16 as well (reference kernel a2)
        got2 = _gen(gpu, bp, table, 5, 2, kernel=1, bitwidth=0)
        assert np.array_equal(got.view(np.uint16), got2.view(np.uint16))


def test_error_behaviour(gpu):
    """NAIVE + b16 and COMBINED are refused like the reference's ctor throws
    (BeamformerCoefficientTest.cu:40-50); generate before upload is NOT_READY."""
    from dc_sand_amd import BeamformerParameters, _lib
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    bp = BeamformerParameters(NR_CHANNELS=8, NR_STATIONS=2, NR_BEAMS=2)
    g = SteeringCoefficientGenerator(bp)
    buf = gpu.mem_alloc(g.output_bytes(1, 1))
    with pytest.raises(_lib.DcsError) as e:
        g.generate(buf, buf.nbytes)
    assert e.value.status == _lib.DCS_ERR_NOT_READY
    g.upload_delays(rand_table(4))
    with pytest.raises(_lib.DcsError) as e:
        g.generate(buf, buf.nbytes, kernel=0, bitwidth=0)
    assert e.value.status == _lib.DCS_ERR_UNSUPPORTED
    with pytest.raises(_lib.DcsError) as e:
        g.generate(buf, buf.nbytes, kernel=3)  # the fused kernel has its own entry point
    assert e.value.status == _lib.DCS_ERR_UNSUPPORTED
    with pytest.raises(_lib.DcsError) as e:
        g.generate_and_beamform(buf, 8, buf, 8, t0=0, nt=8)  # not a multiple of 16
    assert e.value.status == _lib.DCS_ERR_INVALID_ARGUMENT
    with pytest.raises(_lib.DcsError) as e:
        g.generate(buf, buf.nbytes - 1)
    assert e.value.status == _lib.DCS_ERR_INVALID_ARGUMENT
    for bad in (dict(wg_per_cu=1), dict(wg_per_cu=8), dict(wg_per_cu=-2), dict(tiles_per_block=3), dict(form=4), dict(math_mode=-1)):
        with pytest.raises(_lib.DcsError) as e:
            g.set_tuning(**bad)
        assert e.value.status == _lib.DCS_ERR_INVALID_ARGUMENT, bad
    for probe_only in (dict(probe_pace=3), dict(probe_nomath=True)):  # ABI 2's measurement fields are gone from the struct (ABI 3)
        with pytest.raises(TypeError):
            g.set_tuning(**probe_only)
    g.close()


def test_beam_shard_gather(gpu, oracle):
    """set_delays_from_global: a context holding beams [off, off+B_loc) of a
    global [A][B_total] device table produces that column slab of the global
    tensor."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    A, B_total, C = 6, 40, 33
    glob = rand_table(A * B_total, seed=9)
    opg = oracle.params(C, A, B_total)
    full = oracle.generate(opg, glob, 2, 2)
    d_glob = gpu.mem_alloc(glob.nbytes)
    gpu.memcpy_htod(d_glob, glob)
    for off, bl in ((0, 10), (10, 10), (30, 10), (7, 33)):
        bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=bl)
        g = SteeringCoefficientGenerator(bp)
        g.set_delays_from_global(d_glob, B_total, off)
        nbytes = g.output_bytes(1, 2)
        buf = gpu.mem_alloc(nbytes)
        g.generate(buf, nbytes, t0=2, nt=2)
        out = np.empty((2, C, A, bl, 2), dtype=np.float32)
        gpu.memcpy_dtoh(out, buf)
        _check(oracle, out, np.ascontiguousarray(full[:, :, :, off:off + bl, :]))
        g.close()


def test_unit_test_harness_five_phases(gpu, oracle, capsys):
    """The BeamformerCoeffTest mirror: run_test() -> result 1, three timed
    phases, utilisation model; get_result() before run_test() is 0."""
    from dc_sand_amd.beamformer_coeff_test import BeamformerCoeffTest, SteeringCoefficientBitWidth as BW, SteeringCoefficientKernel as K

    def verifier(bp, delays, nt):
        return oracle.generate(oracle.params_from(bp), np.asarray(delays), 0, nt)

    for kern in (K.MULTIPLE_CHANNELS_AND_TIMESTAMPS, K.NAIVE):
        t = BeamformerCoeffTest(1e-4, kern, BW.b32, verifier=verifier, verbose=False)
        assert t.get_result() == 0
        t.run_test()
        assert t.get_result() == 1
        assert t.max_ulp is not None and t.max_ulp <= 1
        total = t.get_time()
        assert total > 0 and t.m_fKernelElapsedTime_ms > 0
        assert t.get_gpu_utilisation_per_single_time_unit() > 0
    with pytest.raises(ValueError):
        BeamformerCoeffTest(1e-4, K.NAIVE, BW.b16)


def test_sincos_probe_fast_path_matches_host_sweep(gpu, oracle, probes):
    """The device evaluates dcs_sincos_fast to the same bits as the host build
    swept exhaustively in test_numerics.py (sampled: 4M arguments)."""
    rng = np.random.default_rng(1)
    x = np.concatenate([
        rng.uniform(-100, 100, 1 << 21).astype(np.float32),
        rng.uniform(-32000, 32000, 1 << 20).astype(np.float32),
        (np.arange(1, 1 << 20, dtype=np.float32) * np.float32(np.pi / 2)),  # near multiples of pi/2
    ])
    n = x.size
    dx, ds, dc = gpu.mem_alloc(4 * n), gpu.mem_alloc(4 * n), gpu.mem_alloc(4 * n)
    gpu.memcpy_htod(dx, x)
    for which, limit in ((0, 1), (2, 1), (3, 1)):
        probes.sincos(which, dx, n, ds, dc)
        gpu.synchronize()
        s = np.empty(n, np.float32)
        c = np.empty(n, np.float32)
        gpu.memcpy_dtoh(s, ds)
        gpu.memcpy_dtoh(c, dc)
        # the fast path is only used (and only proven) below DCS_SINCOS_FAST_LIMIT
        sel = np.abs(x) < 32768.0 if which == 0 else (np.abs(x) < 512.0 if which == 3 else np.ones(n, dtype=bool))
        es = np.sin(x.astype(np.float64)).astype(np.float32)
        ec = np.cos(x.astype(np.float64)).astype(np.float32)
        assert oracle.max_ulp(s[sel], es[sel], limit)[1] == 0
        assert oracle.max_ulp(c[sel], ec[sel], limit)[1] == 0


def test_against_committed_golden_fixtures(gpu):
    """HIP path vs tests/golden/*.npz (written by tests/golden/make_golden.py):
    inputs and expected outputs come from the files alone."""
    import json
    from pathlib import Path

    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.parameters import delay_vals_dtype

    gold = Path(__file__).resolve().parent / "golden"
    man = json.loads((gold / "manifest.json").read_text())
    for case in man["cases"]:
        z = np.load(gold / case["file"], allow_pickle=False)
        bp = BeamformerParameters(NR_CHANNELS=case["C"], NR_STATIONS=case["A"], NR_BEAMS=case["B"])
        table = np.ascontiguousarray(z["delays"]).view(delay_vals_dtype).ravel()
        for t, c0, nc, key in case["slabs"]:
            got = _gen(gpu, bp, table, t, 1, c0=c0, nc=nc)
            exp = z[key]
            gi = got.view(np.int32).astype(np.int64).ravel()
            ei = exp.view(np.int32).astype(np.int64).ravel()
            gi = np.where(gi < 0, -(gi & 0x7FFFFFFF), gi)
            ei = np.where(ei < 0, -(ei & 0x7FFFFFFF), ei)
            assert np.max(np.abs(gi - ei)) <= 1, (case["file"], key)


def test_config3_full_size_every_element(gpu, oracle, probes, record_property):
    """BASELINE configs[2] at FULL size (64 x 1024 x 32768, 16 GiB on the GPU): EVERY one of the 2^32 floats is
    compared with the verifier, as the reference's verify_output does (BeamformerCoefficientTest.cu:348-357) --
    <= 1 ULP under the canonical reading of cos(float), and <= 1 ULP under the float-libm reading too
    (|fRotation| < 100 here, inside the range proven in tests/test_numerics.py), whose count of 1-ULP elements is
    reported.  Then the size-independent properties on the device copy: unit modulus everywhere and the two
    independent forms (tiled / rows) writing bit-identical tensors."""
    import time

    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator, delta_times

    bp = BeamformerParameters(NR_CHANNELS=32768, NR_STATIONS=64, NR_BEAMS=1024)
    op = oracle.params_from(bp)
    table = rand_table(bp.n_pairs)
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(table)
    nbytes = g.output_bytes(1, 1)
    assert nbytes == 16 * 2 ** 30
    buf = gpu.mem_alloc(nbytes)
    t = 9
    g.generate(buf, nbytes, t0=t, nt=1)
    gpu.synchronize()
    t0 = time.perf_counter()
    res = _compare_every_element(gpu, oracle, buf, op, table, delta_times(bp, t, 1)[0], bp.NR_CHANNELS, bp.n_pairs)
    wall = time.perf_counter() - t0
    n = bp.NR_CHANNELS * bp.n_pairs * 2
    for r in (0, 1):
        h = res[r]["hist"]
        assert sum(h) == n == 2 ** 32, "sampled fraction must be 1.0"
        assert h[2] == 0 and h[3] == 0 and res[r]["max_ulp"] <= 1, (r, res[r])
    summary = (f"config 3, all {n} floats: reading 0 (double-then-round) {res[0]['hist'][1]} at 1 ULP, 0 beyond; reading 1 (float libm) "
               f"{res[1]['hist'][1]} at 1 ULP, 0 beyond; {wall:.1f} s wall on {res['threads']} threads (D2H {res['copy_seconds']:.1f} s)")
    print(summary)
    record_property("config3_full_compare", summary)
    # whole-tensor properties on the device: unit modulus everywhere, and the two
    # independent forms (tiled / rows: different grids and index arithmetic, both
    # 64-bit) write bit-identical 16 GiB tensors
    ck1, dev1 = probes.tensor_properties(buf, nbytes)
    assert dev1 < 4e-7
    g.set_tuning(form=2)
    gpu.memset(buf, 0, nbytes)
    g.generate(buf, nbytes, t0=t, nt=1)
    ck2, dev2 = probes.tensor_properties(buf, nbytes)
    assert ck2 == ck1 and dev2 == dev1
    # small-case anchor of the checksum itself
    small = np.empty((4, bp.NR_STATIONS, bp.NR_BEAMS, 2), dtype=np.float32)
    gpu.memcpy_dtoh(small, buf)
    cks, _ = probes.tensor_properties(buf, small.nbytes)
    assert cks == oracle.checksum_of(small)
    g.close()
    buf.free()


@pytest.mark.parametrize("math_mode", [0, 4])
def test_config3_full_size_fp16_every_element(gpu, oracle, record_property, math_mode):
    """The packed binary16 output of config 3 at FULL size (2^32 halves, 8 GiB), EVERY element against
    RN-even(verifier's fp32) -- the check the reference never makes (BeamformerCoefficientTest.cu:282-287 skip the b16
    case).  Default arithmetic (the <= 1 ULP fp32 pair rounded once): a half differs from the expectation only where the
    fp32 value sits within its own last place of a binary16 rounding boundary -- at most 1 binary16 ulp, and rarely.  The
    opt-in b16 arithmetic form (math_mode 4): within 1 binary16 ulp everywhere (tests/test_numerics.py proves that per
    argument; this is the whole tensor)."""
    import time

    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd._lib import B16
    from dc_sand_amd.generator import SteeringCoefficientGenerator, delta_times

    bp = BeamformerParameters(NR_CHANNELS=32768, NR_STATIONS=64, NR_BEAMS=1024)
    op = oracle.params_from(bp)
    table = rand_table(bp.n_pairs)
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(table)
    if math_mode:
        g.set_tuning(math_mode=math_mode)
    nbytes = g.output_bytes(B16, 1)
    assert nbytes == 8 * 2 ** 30
    buf = gpu.mem_alloc(nbytes)
    t = 9
    g.generate(buf, nbytes, t0=t, nt=1, bitwidth=B16)
    gpu.synchronize()
    t0 = time.perf_counter()
    res = _compare_every_element(gpu, oracle, buf, op, table, delta_times(bp, t, 1)[0], bp.NR_CHANNELS, bp.n_pairs, readings=(0,), half=True)
    wall = time.perf_counter() - t0
    n = bp.NR_CHANNELS * bp.n_pairs * 2
    h = res[0]["hist"]
    assert sum(h) == n == 2 ** 32, "sampled fraction must be 1.0"
    assert h[2] == 0 and h[3] == 0 and res[0]["max_ulp"] <= 1, res[0]
    if math_mode == 0:
        assert h[1] < n // 1000  # double rounding only: a few in 10^4
    summary = (f"config 3 fp16 (math_mode {math_mode}), all {n} halves: {h[1]} at 1 binary16 ulp of RN16(verifier), 0 beyond; "
               f"{wall:.1f} s wall on {res['threads']} threads (D2H {res['copy_seconds']:.1f} s)")
    print(summary)
    record_property("config3_fp16_full_compare", summary)
    g.close()
    buf.free()


def test_streaming_graph_ticks_with_table_updates(gpu, oracle):
    """BASELINE configs[4] plumbing: hipGraph-captured launch replayed per tick
    with a new time index, and a new delay table landing between ticks
    (double-buffered); every tick's slab equals the oracle's."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    bp = BeamformerParameters(NR_CHANNELS=96, NR_STATIONS=8, NR_BEAMS=40)
    op = oracle.params_from(bp)
    c0, nc = 16, 64
    tables = [rand_table(bp.n_pairs, seed=s) for s in (31, 32, 33)]
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(tables[0])
    nbytes = nc * bp.n_pairs * 8
    buf = gpu.mem_alloc(nbytes)
    stream = gpu.Stream()
    st = g.stream_begin(buf, nbytes, c0, nc, stream)
    host = np.empty((1, nc, bp.NR_STATIONS, bp.NR_BEAMS, 2), dtype=np.float32)
    cur = 0
    for tick, t in enumerate([0, 1, 5, 7, 9, 18, 255, 256, 1000]):
        new = None
        if tick in (3, 4, 7):
            cur = (cur + 1) % 3
            new = tables[cur]
        st.tick(t, new)
        stream.synchronize()
        gpu.memcpy_dtoh(host, buf)
        exp = oracle.generate(op, tables[cur], t, 1, c0, nc)
        mx, n_over, first = oracle.max_ulp(host, exp, 1)
        assert n_over == 0, (tick, t, mx, first)
    st.end()
    g.close()


@pytest.mark.parametrize("A,B,C,nt,seeded", [(64, 16, 64, 256, False), (64, 16, 64, 32, True), (8, 4, 5, 16, True),
                                             (37, 21, 9, 48, True), (130, 3, 4, 16, True), (4, 40, 7, 32, True),
                                             (9, 5, 3, 16, True), (129, 2, 2, 16, True), (258, 2, 5, 16, True),
                                             (1, 1, 1, 16, True), (3, 17, 2, 32, True)])
def test_fused_coefficient_generation_and_beamforming(gpu, oracle, A, B, C, nt, seeded):
    """beams = sum over antennas (in order) of the element-wise product of
    coefficient and int8 sample, against the verifier restatement
    (BeamformerCoefficientTest.cu:363-414).  The reference's tolerance is 1e-1
    (runBeamformerTests.cpp:15); the summation order is the verifier's, so the only
    difference is each coefficient's <= 1 ULP: |diff| <= sum|sample| * 2^-23 ~ 2e-3
    at 64 antennas; asserted at 2e-5 * A (and at the reference's 1e-1)."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator, simulate_input

    bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B, NR_SAMPLES_PER_CHANNEL=nt)
    op = oracle.params_from(bp)
    table = rand_table(bp.n_pairs, seed=A + B) if seeded else simulate_input(bp)  # indexed [b*A + a] by this kernel
    ant = oracle.simulate_antenna_data(op, nt)
    if seeded:
        ant = np.random.default_rng(A).integers(-128, 128, size=ant.shape, dtype=np.int8)
    exp = oracle.beamform(op, table, nt, ant)
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(table)
    d_ant = gpu.mem_alloc(ant.nbytes)
    gpu.memcpy_htod(d_ant, ant)
    d_beams = gpu.mem_alloc(exp.nbytes + 64)
    gpu.memset(d_beams, 0xFF, exp.nbytes + 64)
    g.generate_and_beamform(d_ant, ant.nbytes, d_beams, exp.nbytes, t0=0, nt=nt)
    host = np.empty(exp.nbytes + 64, dtype=np.uint8)
    gpu.memcpy_dtoh(host, d_beams)
    assert np.all(host[exp.nbytes:] == 0xFF)
    got = host[:exp.nbytes].view(np.float32).reshape(exp.shape)
    diff = np.abs(got - exp)
    assert np.all(np.isfinite(got))
    assert diff.max() <= 2e-5 * A + 1e-6, diff.max()
    assert oracle.compare(got, exp, 1e-1) == -1
    g.close()


def test_fused_seeded_fuzz(gpu, oracle):
    """40 seeded random cases of the per-sample fused kernel: antennas 1..300 (several 128-antenna LDS chunks), beams
    1..70 (partial 16-beam groups), channels (odd counts: the two-channels-per-pass loop's tail), 1..6 sample blocks,
    a time offset, times by index or by value -- each within 2e-5 * A of the verifier (same summation order: the only
    difference is each coefficient's <= 1 ULP), nothing written outside the tensor."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator, delta_times

    rng = np.random.default_rng(20261006)
    for case in range(40):
        A = int(rng.choice([1, 2, 63, 64, 65, 127, 128, 129, 256, 257, 300])) if case % 2 else int(rng.integers(1, 200))
        B = int(rng.integers(1, 71))
        C = int(rng.integers(1, 8))
        nblk = int(rng.integers(1, 7))
        while A * B * C * nblk * 16 > 400000 and nblk > 1:
            nblk -= 1
        while A * B * C * nblk * 16 > 400000 and C > 1:
            C -= 1
        nt = 16 * nblk
        bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B, NR_SAMPLES_PER_CHANNEL=nt)
        op = oracle.params_from(bp)
        table = rand_table(bp.n_pairs, seed=5000 + case)
        ant = rng.integers(-128, 128, size=(C, nblk, A, 16, 2), dtype=np.int8)
        by_value = bool(rng.integers(0, 2))
        if by_value:
            dts = rng.uniform(0.0, 0.3, size=nt).astype(np.float32)
            exp = oracle.beamform_dt(op, table, dts, ant)
        else:
            exp = oracle.beamform(op, table, nt, ant)
        g = SteeringCoefficientGenerator(bp)
        g.upload_delays(table)
        d_ant = gpu.mem_alloc(ant.nbytes)
        gpu.memcpy_htod(d_ant, ant)
        d_beams = gpu.mem_alloc(exp.nbytes + 64)
        gpu.memset(d_beams, 0xFF, exp.nbytes + 64)
        if by_value:
            g.generate_and_beamform_dt(d_ant, ant.nbytes, d_beams, exp.nbytes, dts)
        else:
            g.generate_and_beamform(d_ant, ant.nbytes, d_beams, exp.nbytes, t0=0, nt=nt)
        host = np.empty(exp.nbytes + 64, dtype=np.uint8)
        gpu.memcpy_dtoh(host, d_beams)
        tag = (case, A, B, C, nt, by_value)
        assert np.all(host[exp.nbytes:] == 0xFF), tag
        got = host[:exp.nbytes].view(np.float32).reshape(exp.shape)
        assert np.all(np.isfinite(got)), tag
        assert np.abs(got - exp).max() <= 2e-5 * A + 1e-6, (tag, float(np.abs(got - exp).max()))
        g.close()
        d_ant.free()
        d_beams.free()


def test_fused_more_steps_than_ride_in_the_kernel_arguments_and_under_capture(gpu, oracle):
    """The per-sample fused kernel with 288 > 256 time steps (their fDeltaTime table is staged through pinned memory, two
    terms launches) and, with 256, captured into a hipGraph after a first plain call and replayed on new samples."""
    import ctypes

    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    A, B, C = 9, 20, 3
    rng = np.random.default_rng(11)
    for nt in (288, 256):
        bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B, NR_SAMPLES_PER_CHANNEL=nt)
        op = oracle.params_from(bp)
        table = rand_table(bp.n_pairs, seed=62)
        ant = rng.integers(-128, 128, size=(C, nt // 16, A, 16, 2), dtype=np.int8)
        exp = oracle.beamform(op, table, nt, ant)
        g = SteeringCoefficientGenerator(bp)
        g.upload_delays(table)
        d_ant = gpu.mem_alloc(ant.nbytes)
        gpu.memcpy_htod(d_ant, ant)
        d_beams = gpu.mem_alloc(exp.nbytes)
        s = gpu.Stream()
        g.generate_and_beamform(d_ant, ant.nbytes, d_beams, exp.nbytes, t0=0, nt=nt, stream=s.handle)
        s.synchronize()
        got = np.empty_like(exp)
        gpu.memcpy_dtoh(got, d_beams)
        assert np.abs(got - exp).max() <= 2e-5 * A + 1e-6
        if nt == 256:
            hip = ctypes.CDLL("libamdhip64.so")
            V = ctypes.c_void_p
            hip.hipStreamBeginCapture.argtypes = [V, ctypes.c_int]
            hip.hipStreamEndCapture.argtypes = [V, ctypes.POINTER(V)]
            hip.hipGraphInstantiate.argtypes = [ctypes.POINTER(V), V, V, V, ctypes.c_size_t]
            hip.hipGraphLaunch.argtypes = [V, V]
            hip.hipGraphExecDestroy.argtypes = [V]
            hip.hipGraphDestroy.argtypes = [V]
            assert hip.hipStreamBeginCapture(V(s.handle), 0) == 0
            g.generate_and_beamform(d_ant, ant.nbytes, d_beams, exp.nbytes, t0=0, nt=nt, stream=s.handle)
            graph = V()
            assert hip.hipStreamEndCapture(V(s.handle), ctypes.byref(graph)) == 0 and graph.value
            ex = V()
            assert hip.hipGraphInstantiate(ctypes.byref(ex), graph, None, None, 0) == 0
            for rep in range(2):
                ant2 = rng.integers(-128, 128, size=ant.shape, dtype=np.int8)
                gpu.memcpy_htod(d_ant, ant2, stream=s.handle)
                gpu.memset(d_beams, 0xFF, exp.nbytes, stream=s.handle)
                assert hip.hipGraphLaunch(ex, V(s.handle)) == 0
                s.synchronize()
                gpu.memcpy_dtoh(got, d_beams)
                assert np.abs(got - oracle.beamform(op, table, nt, ant2)).max() <= 2e-5 * A + 1e-6
            hip.hipGraphExecDestroy(ex)
            hip.hipGraphDestroy(graph)
        g.close()


def test_fused_harness_and_slow_path(gpu, oracle):
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.beamformer_coeff_test import BeamformerCoeffTest, SteeringCoefficientBitWidth as BW, SteeringCoefficientKernel as K
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    def beam_verifier(bp, delays, nt, ant):
        return oracle.beamform(oracle.params_from(bp), np.asarray(delays), nt, ant)

    t = BeamformerCoeffTest(1e-1, K.COMBINED_COEFF_GEN_AND_BEAMFORMER_SINGLE_CHANNEL, BW.b32, beam_verifier=beam_verifier, verbose=False)
    t.run_test()
    assert t.get_result() == 1 and t.max_abs_diff < 2e-3
    assert t.get_time() > 0
    with pytest.raises(ValueError):
        BeamformerCoeffTest(1e-1, K.COMBINED_COEFF_GEN_AND_BEAMFORMER_SINGLE_CHANNEL, BW.b16)
    # a pair outside the fast path's range sends its 16-sample blocks down the slow branch
    bp = BeamformerParameters(NR_CHANNELS=6, NR_STATIONS=9, NR_BEAMS=5, NR_SAMPLES_PER_CHANNEL=32)
    op = oracle.params_from(bp)
    table = rand_table(bp.n_pairs, seed=77)
    table["fDelayRate_sps"][11] = 1e-2
    table["fDelayRate_sps"][12] = 1e-30
    ant = np.random.default_rng(3).integers(-128, 128, size=(6, 2, 9, 16, 2), dtype=np.int8)
    exp = oracle.beamform(op, table, 32, ant)
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(table)
    d_ant = gpu.mem_alloc(ant.nbytes)
    gpu.memcpy_htod(d_ant, ant)
    d_beams = gpu.mem_alloc(exp.nbytes)
    g.generate_and_beamform(d_ant, ant.nbytes, d_beams, exp.nbytes, t0=0, nt=32)
    got = np.empty_like(exp)
    gpu.memcpy_dtoh(got, d_beams)
    assert np.abs(got - exp).max() <= 2e-4 * 9
    g.close()


@pytest.mark.parametrize("A,B,C", [(9, 16, 16384), (130, 16, 4096)])
def test_fused_slow_path_with_several_channels_per_pass(gpu, oracle, A, B, C):
    """The fused kernel's slow branch when a pass covers 4 channels (<= 64 antennas) or 2 (more: the
    antennas also cross the 128-antenna LDS chunk): enough channels that the launcher keeps 4 per
    workgroup, and slow-class pairs so that every 16-sample block takes the branch."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    nt = 16
    bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B, NR_SAMPLES_PER_CHANNEL=nt)
    op = oracle.params_from(bp)
    table = rand_table(bp.n_pairs, seed=A)
    table["fDelayRate_sps"][5] = 1e-2    # |fRotation| far beyond the fast path's range
    table["fDelayRate_sps"][A + 1] = 1e-30  # rate term outside the divide's proven range
    ant = np.random.default_rng(A).integers(-128, 128, size=(C, nt // 16, A, 16, 2), dtype=np.int8)
    exp = oracle.beamform(op, table, nt, ant)
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(table)
    d_ant = gpu.mem_alloc(ant.nbytes)
    gpu.memcpy_htod(d_ant, ant)
    d_beams = gpu.mem_alloc(exp.nbytes)
    g.generate_and_beamform(d_ant, ant.nbytes, d_beams, exp.nbytes, t0=0, nt=nt)
    got = np.empty_like(exp)
    gpu.memcpy_dtoh(got, d_beams)
    assert np.all(np.isfinite(got))
    assert np.abs(got - exp).max() <= 2e-4 * A
    g.close()


def test_config4_one_rank_shard_at_full_size_every_element(gpu, oracle, probes, record_property):
    """BASELINE configs[3]: 256 ant x 4096 beam x 32768 chan beam-sharded over 8 GPUs.  One rank's share at FULL size
    on this GPU (rank 3: beams [1536, 2048), 2^32 coefficients, 32 GiB): the slice is gathered on the device from the
    16 MiB global table, and EVERY one of its 2^33 floats is compared with the verifier's column slab (canonical
    reading; byte offsets beyond 2^32); the tiled and rows forms must agree on the whole tensor."""
    import time

    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator, delta_times
    from dc_sand_amd.sharding import beam_range, local_parameters, slice_table

    gp = BeamformerParameters(NR_CHANNELS=32768, NR_STATIONS=256, NR_BEAMS=4096)
    sh = beam_range(gp.NR_BEAMS, 8, 3)
    assert (sh.beam_lo, sh.beam_hi) == (1536, 2048)
    lp = local_parameters(gp, sh)
    glob = rand_table(gp.n_pairs, seed=44)
    assert glob.nbytes == 16 * 2 ** 20
    d_glob = gpu.mem_alloc(glob.nbytes)
    gpu.memcpy_htod(d_glob, glob)
    g = SteeringCoefficientGenerator(lp)
    g.set_delays_from_global(d_glob, gp.NR_BEAMS, sh.beam_lo)
    nbytes = g.output_bytes(1, 1)
    assert nbytes == 32 * 2 ** 30
    buf = gpu.mem_alloc(nbytes)
    t = 18
    g.generate(buf, nbytes, t0=t, nt=1)
    gpu.synchronize()
    local = slice_table(glob, gp, sh)
    op = oracle.params_from(lp)
    t0 = time.perf_counter()
    res = _compare_every_element(gpu, oracle, buf, op, local, delta_times(lp, t, 1)[0], lp.NR_CHANNELS, lp.n_pairs, readings=(0,))
    wall = time.perf_counter() - t0
    h = res[0]["hist"]
    assert sum(h) == 2 ** 33 and h[2] == 0 and h[3] == 0 and res[0]["max_ulp"] <= 1, res[0]
    summary = f"config 4 rank-3 shard, all {sum(h)} floats: {h[1]} at 1 ULP, 0 beyond; {wall:.1f} s wall on {res['threads']} threads"
    print(summary)
    record_property("config4_shard_full_compare", summary)
    ck, dev = probes.tensor_properties(buf, nbytes)
    assert dev < 4e-7
    g.set_tuning(form=2)
    gpu.memset(buf, 0, nbytes)
    g.generate(buf, nbytes, t0=t, nt=1)
    ck2, dev2 = probes.tensor_properties(buf, nbytes)
    assert (ck2, dev2) == (ck, dev)
    g.close()
    buf.free()


@pytest.mark.parametrize("script,args", [("steering_coefficients.py", ["16", "24", "512"]), ("streaming_ticks.py", ["16", "64", "2048", "256", "20"])])
def test_examples_run_and_verify_themselves(gpu, script, args):
    """examples/: the pycuda-shaped host pattern and the config-5 stream with device tables, on small shapes; each verifies
    its own output against the oracle and exits 0."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    res = subprocess.run([sys.executable, str(root / "examples" / script), *args], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    assert "max ULP distance to the CPU verifier" in res.stdout


def test_cpp_host_against_c_abi(gpu):
    """The reference's harness is C++: tests/cpp/run_beamformer_tests.cpp is the
    runBeamformerTests executable written against the C-ABI (plain g++, no hipcc
    on the host side), with the C oracle as verify_output's expected data."""
    import subprocess
    from pathlib import Path

    d = Path(__file__).resolve().parent / "cpp"
    res = subprocess.run(["make", "-C", str(d)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    run = subprocess.run([str(d / "run_beamformer_tests")], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    assert run.stdout.count("CPU took") == 3
    for name in ("Combined Steering Coeffs+Beamforming", "Multiple Chans+Timestamps", "Multiple Channels", "Naive Implementation"):
        assert name in run.stdout


@pytest.mark.parametrize("form", [1, 2, 3])
def test_arithmetic_forms_agree_and_class_boundaries(gpu, oracle, form):
    """The fast path has four arithmetic forms (3-op / 5-op divide x low- / full-degree
    polynomials), chosen per wave from a bound on |fRotation|; all must give the
    oracle's bits (<= 1 ULP) and each other's bits.  The table mixes pairs below 500 rad,
    between 500 and 32000 rad and beyond (slow path), so every class occurs, also
    inside one wave."""
    from dc_sand_amd import BeamformerParameters

    bp = BeamformerParameters(NR_CHANNELS=200, NR_STATIONS=3, NR_BEAMS=200)
    table = rand_table(bp.n_pairs, seed=13)
    table["fDelayRate_sps"][100:140] = 2.0e-5    # ~630 rad at the top channel: full-degree class
    table["fDelayRate_sps"][300:310] = -4.0e-4   # ~12600 rad
    table["fDelayRate_sps"][470] = 1.5e-3        # > 32000 rad: slow path
    table["fPhase_rad"][520:530] = 499.0         # just below / above the class boundary
    table["fPhase_rad"][530:540] = 501.0
    op = oracle.params_from(bp)
    exp = oracle.generate(op, table, 9, 2)
    outs = []
    for mode in (0, 1, 2, 3):
        got = _gen(gpu, bp, table, 9, 2, tuning=dict(form=form, math_mode=mode))
        _check(oracle, got, exp)
        outs.append(got.copy())
    for o in outs[1:]:
        # the forms are all within 1 ULP of the oracle; between each other they may
        # differ by the same 1 ULP only where the polynomial degree differs
        assert oracle.max_ulp(o, outs[0], 2)[1] == 0
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))  # divide form never changes a bit
    assert np.array_equal(outs[2].view(np.uint32), outs[3].view(np.uint32))


def test_autotune_keeps_results(gpu, oracle):
    """dcs_bf_autotune picks a launch geometry by measurement; whatever it picks, the
    bits are the oracle's."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    bp = BeamformerParameters(NR_CHANNELS=300, NR_STATIONS=16, NR_BEAMS=40)
    table = rand_table(bp.n_pairs, seed=55)
    op = oracle.params_from(bp)
    for bw in (1, 0):
        g = SteeringCoefficientGenerator(bp)
        g.upload_delays(table)
        nbytes = g.output_bytes(bw, 2)
        buf = gpu.mem_alloc(nbytes)
        import time

        t0 = time.perf_counter()
        chosen = g.autotune(buf, nbytes, bitwidth=bw)
        t_first = time.perf_counter() - t0
        assert chosen["form"] == 0 and chosen["tiles_per_block"] in (1, 2, 4) and chosen["chan_per_block"] >= 1
        t0 = time.perf_counter()
        again = g.autotune(buf, nbytes, bitwidth=bw)  # cached in the context per output width
        assert again == chosen and time.perf_counter() - t0 < max(0.05, 0.25 * t_first)
        g.generate(buf, nbytes, t0=5, nt=2, bitwidth=bw)
        exp = oracle.generate(op, table, 5, 2)
        if bw == 1:
            out = np.empty(exp.shape, dtype=np.float32)
            gpu.memcpy_dtoh(out, buf)
            _check(oracle, out, exp)
        else:
            out = np.empty(exp.shape, dtype=np.float16)
            gpu.memcpy_dtoh(out, buf)
            assert np.max(np.abs(out.astype(np.float32) - exp)) < 1e-3
        g.close()


def test_fp16_is_exactly_rne_of_the_fp32_output(gpu):
    """b16 output == RN-even(b32 output), bit for bit (the reference's
    __floats2half2_rn of the same fp32 value, BeamformerKernels.cu:113,182): the fp16
    mode adds one correctly rounded conversion and nothing else."""
    from dc_sand_amd import BeamformerParameters

    for (A, B, C) in ((64, 16, 64), (5, 7, 33), (2, 300, 40)):
        bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B)
        table = rand_table(bp.n_pairs, seed=C)
        f32 = _gen(gpu, bp, table, 9, 3, bitwidth=1)
        f16 = _gen(gpu, bp, table, 9, 3, bitwidth=0)
        assert np.array_equal(f16.view(np.uint16), f32.astype(np.float16).view(np.uint16))


@pytest.mark.parametrize("form", [1, 2])
def test_many_time_steps_cross_the_internal_chunking(gpu, oracle, form):
    """nt = 5000 > the 4096 fDeltaTime values staged per launch: the call is split into
    several launches internally; every time step must still be the oracle's."""
    from dc_sand_amd import BeamformerParameters

    bp = BeamformerParameters(NR_CHANNELS=5, NR_STATIONS=3, NR_BEAMS=6)
    table = rand_table(bp.n_pairs, seed=99)
    op = oracle.params_from(bp)
    got = _gen(gpu, bp, table, 250, 5000, tuning=dict(form=form))
    _check(oracle, got, oracle.generate(op, table, 250, 5000))


@pytest.mark.parametrize("form", [1, 2, 3])
@pytest.mark.parametrize("bitwidth", [1, 0])
def test_output_pointer_not_16_byte_aligned(gpu, oracle, form, bitwidth):
    """An output tensor that starts 8 (fp32) or 4 (fp16) bytes off a 16-byte boundary
    takes the per-pair store path; the bytes before and after it stay untouched."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    bp = BeamformerParameters(NR_CHANNELS=19, NR_STATIONS=4, NR_BEAMS=34)
    table = rand_table(bp.n_pairs, seed=5)
    g = SteeringCoefficientGenerator(bp)
    g.set_tuning(form=form)
    g.upload_delays(table)
    eb = 8 if bitwidth == 1 else 4
    nbytes = 2 * bp.NR_CHANNELS * bp.n_pairs * eb
    buf = gpu.mem_alloc(nbytes + 64)
    gpu.memset(buf, 0xFF, nbytes + 64)
    g.generate(int(buf) + eb, nbytes, t0=3, nt=2, bitwidth=bitwidth)
    host = np.empty(nbytes + 64, dtype=np.uint8)
    gpu.memcpy_dtoh(host, buf)
    assert np.all(host[:eb] == 0xFF) and np.all(host[eb + nbytes:] == 0xFF)
    exp = oracle.generate(oracle.params_from(bp), table, 3, 2)
    if bitwidth == 1:
        got = host[eb:eb + nbytes].copy().view(np.float32).reshape(exp.shape)
        _check(oracle, got, exp)
    else:
        got = host[eb:eb + nbytes].copy().view(np.float16).reshape(exp.shape)
        assert np.max(np.abs(got.astype(np.float32) - exp)) < 1e-3
    g.close()


def test_fused_with_time_offset(gpu, oracle):
    """generate_and_beamform(t0 = 32, nt = 32) equals the last two 16-sample blocks of the
    verifier's 64-sample tensor (the time index is absolute)."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    A, B, C = 12, 20, 6
    bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B, NR_SAMPLES_PER_CHANNEL=64)
    op = oracle.params_from(bp)
    table = rand_table(bp.n_pairs, seed=17)
    ant_full = np.random.default_rng(9).integers(-128, 128, size=(C, 4, A, 16, 2), dtype=np.int8)
    exp = oracle.beamform(op, table, 64, ant_full)[:, 2:4]
    ant = np.ascontiguousarray(ant_full[:, 2:4])
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(table)
    d_ant = gpu.mem_alloc(ant.nbytes)
    gpu.memcpy_htod(d_ant, ant)
    got = np.empty((C, 2, B, 16, 2), dtype=np.float32)
    d_beams = gpu.mem_alloc(got.nbytes)
    g.generate_and_beamform(d_ant, ant.nbytes, d_beams, got.nbytes, t0=32, nt=32)
    gpu.memcpy_dtoh(got, d_beams)
    assert np.abs(got - np.ascontiguousarray(exp)).max() <= 2e-5 * A + 1e-6
    g.close()


def test_config5_streaming_at_the_200us_slab(gpu, oracle):
    """BASELINE configs[4] at size: 64 x 1024 pairs x the 2560-channel slab that sustains the 200 us cadence
    (1.34 GB per tick), hipGraph replay, MODEL time advancing 200 us per tick (dcs_bf_stream_tick_dt, and
    dcs_bf_stream_tick_at across a seconds boundary), a new delay table landing between ticks; sampled rows of
    every tick against the oracle evaluated at the same fDeltaTime."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    bp = BeamformerParameters(NR_CHANNELS=32768, NR_STATIONS=64, NR_BEAMS=1024)
    op = oracle.params_from(bp)
    c0, nc = 4096, 2560
    tables = [rand_table(bp.n_pairs, seed=s) for s in (61, 62)]
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(tables[0])
    nbytes = nc * bp.n_pairs * 8
    buf = gpu.mem_alloc(nbytes)
    stream = gpu.Stream()
    st = g.stream_begin(buf, nbytes, c0, nc, stream)
    row = bp.n_pairs * 8
    host = np.empty((bp.NR_STATIONS, bp.NR_BEAMS, 2), dtype=np.float32)
    cur = 0
    ref = (777, 999_500_000)  # 0.5 ms before a seconds boundary: ticks 3.. lie beyond it
    for tick in range(1, 7):
        new = None
        if tick == 3:
            cur, new = 1, tables[1]
        if tick % 2:
            dt = np.float32(tick * 200e-6)
            st.tick_dt(dt, new)
        else:
            ns = ref[1] + tick * 200_000
            now = (ref[0] + ns // 10 ** 9, ns % 10 ** 9)  # a normalised wall-clock reading
            dt = oracle.ts_diff(ref, now)
            assert abs(float(dt) - tick * 200e-6) < 1e-7
            st.tick_at(now, ref, new)
        stream.synchronize()
        for cl in (0, 1279, nc - 1):
            gpu.memcpy_dtoh(host, int(buf) + cl * row)
            exp = oracle.generate_dt(op, tables[cur], [dt], c0 + cl, 1)
            assert oracle.max_ulp(host, exp, 1)[1] == 0, (tick, cl)
    st.tick(300)  # the time-index form still works on the same stream object
    stream.synchronize()
    gpu.memcpy_dtoh(host, int(buf))
    assert oracle.max_ulp(host, oracle.generate(op, tables[cur], 300, 1, c0, 1), 1)[1] == 0
    st.end()
    g.close()


def test_seeded_fuzz_of_shapes_slabs_time_ranges_and_geometries(gpu, oracle):
    """80 seeded random cases: shape (ragged tiles, odd pair counts), channel slab, time range (1..300 time
    steps: both sides of the 256 that travel in the kernel arguments), kernel selection, output width,
    launch geometry of either form (incl. wg_per_cu) and an output pointer 0 / 8 / 16 bytes off alignment --
    every fp32 case within 1 ULP of the oracle, every fp16 case the RN-even image of the fp32 run,
    canaries before and after the tensor intact."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    rng = np.random.default_rng(20261004)
    for case in range(80):
        A, B = int(rng.integers(1, 9)), int(rng.integers(1, 70))
        C = int(rng.integers(1, 80))
        nt = int(rng.choice([1, 2, 3, 17, 255, 256, 257, 300])) if case % 4 == 0 else int(rng.integers(1, 6))
        if nt > 16:
            A, B, C = min(A, 3), min(B, 9), min(C, 7)  # keep the oracle's work small
        t0 = int(rng.integers(0, 1000))
        bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B)
        table = rand_table(bp.n_pairs, seed=1000 + case)
        bitwidth = int(rng.integers(0, 2))
        form = int(rng.integers(1, 4))
        if form != 2:
            tuning = dict(form=form if case % 3 else 0, tiles_per_block=int(rng.choice([1, 2, 4])), chan_per_block=int(rng.integers(1, 40)),
                          nontemporal=int(rng.integers(0, 2)), wg_per_cu=int(rng.choice([-1, 0, 2, 5, 6, 7])))
        else:
            tuning = dict(form=2, waves_per_block=int(rng.choice([4, 8, 16])), rows_per_wave=int(rng.integers(1, 5)),
                          rows_same_tile=int(rng.integers(0, 2)), wg_per_cu=int(rng.choice([0, 3, 6])))
        use_slab = bool(rng.integers(0, 2))
        c0 = int(rng.integers(0, C)) if use_slab else 0
        nc = int(rng.integers(1, C - c0 + 1)) if use_slab else C
        kernel = 2 if (use_slab or form == 2) else int(rng.choice([1, 2]))  # (MULTIPLE_CHANNELS: the tiled form, one launch per time step)
        eb = 8 if bitwidth == 1 else 4
        off = int(rng.choice([0, eb, 16]))
        nbytes = nt * nc * bp.n_pairs * eb
        g = SteeringCoefficientGenerator(bp)
        g.set_tuning(**tuning)
        g.upload_delays(table)
        buf = gpu.mem_alloc(nbytes + 64)
        gpu.memset(buf, 0xFF, nbytes + 64)
        if use_slab:
            g.generate_slab(int(buf) + off, nbytes, c0, nc, t0=t0, nt=nt, bitwidth=bitwidth)
        else:
            g.generate(int(buf) + off, nbytes, t0=t0, nt=nt, kernel=kernel, bitwidth=bitwidth)
        host = np.empty(nbytes + 64, dtype=np.uint8)
        gpu.memcpy_dtoh(host, buf)
        tag = f"case {case}: {A}x{B}x{C} nt={nt} t0={t0} slab=({c0},{nc}) kernel={kernel} b{bitwidth} off={off} {tuning}"
        assert np.all(host[:off] == 0xFF) and np.all(host[off + nbytes:] == 0xFF), tag
        exp = oracle.generate(oracle.params_from(bp), table, t0, nt, c0, nc)
        if bitwidth == 1:
            got = host[off:off + nbytes].copy().view(np.float32).reshape(exp.shape)
            mx, n_over, first = oracle.max_ulp(got, exp, 1)
            assert n_over == 0, f"{tag}: max ULP {mx}, first {first}"
        else:
            got16 = host[off:off + nbytes].copy().view(np.uint16)
            g.set_tuning()
            buf32 = gpu.mem_alloc(nbytes * 2)
            if use_slab:
                g.generate_slab(buf32, nbytes * 2, c0, nc, t0=t0, nt=nt, bitwidth=1)
            else:
                g.generate(buf32, nbytes * 2, t0=t0, nt=nt, kernel=2, bitwidth=1)
            h32 = np.empty(nbytes * 2, dtype=np.uint8)
            gpu.memcpy_dtoh(h32, buf32)
            ref16 = oracle.f32_to_f16_bits(h32.view(np.float32))
            assert np.array_equal(got16, ref16.ravel()), tag
            buf32.free()
        g.close()
        buf.free()


@pytest.mark.parametrize("nt,kernel", [(1, 2), (256, 2), (16, 1), (3, 1), (16, 0)])
def test_generate_under_stream_capture(gpu, oracle, nt, kernel):
    """include/dcs_beamformer.h: "all device work is enqueued on the caller's stream so the calls can be
    captured in a hipGraph".  dcs_bf_generate under hipStreamBeginCapture / EndCapture (up to 256 time steps:
    their fDeltaTime values are kernel arguments; longer launches stage a table through pinned memory and
    are not capturable), instantiated and replayed twice: the replays write what a direct launch writes."""
    import ctypes

    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    hip = ctypes.CDLL("libamdhip64.so")
    V = ctypes.c_void_p
    hip.hipStreamBeginCapture.argtypes = [V, ctypes.c_int]
    hip.hipStreamEndCapture.argtypes = [V, ctypes.POINTER(V)]
    hip.hipGraphInstantiate.argtypes = [ctypes.POINTER(V), V, V, V, ctypes.c_size_t]
    hip.hipGraphLaunch.argtypes = [V, V]
    hip.hipGraphExecDestroy.argtypes = [V]
    hip.hipGraphDestroy.argtypes = [V]

    bp = BeamformerParameters(NR_CHANNELS=24, NR_STATIONS=5, NR_BEAMS=13)
    table = rand_table(bp.n_pairs, seed=31)
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(table)
    gpu.synchronize()
    nbytes = g.output_bytes(1, nt)
    buf = gpu.mem_alloc(nbytes)
    s = gpu.Stream()
    assert hip.hipStreamBeginCapture(V(s.handle), 0) == 0  # hipStreamCaptureModeGlobal
    g.generate(buf, nbytes, t0=7, nt=nt, stream=s.handle, kernel=kernel)  # (kernel 1 from 8 time steps on: a fork and a join over side streams)
    graph = V()
    assert hip.hipStreamEndCapture(V(s.handle), ctypes.byref(graph)) == 0 and graph.value
    ex = V()
    assert hip.hipGraphInstantiate(ctypes.byref(ex), graph, None, None, 0) == 0
    exp = oracle.generate(oracle.params_from(bp), table, 7, nt)
    for _ in range(2):
        gpu.memset(buf, 0xFF, nbytes, stream=s.handle)  # nothing was generated yet by the capture itself
        assert hip.hipGraphLaunch(ex, V(s.handle)) == 0
        s.synchronize()
        got = np.empty(exp.shape, dtype=np.float32)
        gpu.memcpy_dtoh(got, buf)
        _check(oracle, got, exp)
    hip.hipGraphExecDestroy(ex)
    hip.hipGraphDestroy(graph)
    g.close()
    buf.free()


# ---- arbitrary-time entry points (the reference kernels take struct timespec sCurrentTime, sRefTime:
# ---- BeamformerKernels.cuh:38-42, 81-86; the verifier's fDeltaTime is ts_diff, BeamformerCoefficientTest.cu:12-18, :320)
OFF_GRID_DT = [0.0, 200e-6, 400e-6, 1e-3 + 3e-7, 0.123456, 1.5, -2e-4, 86400.0]  # not multiples of 819.2 us; before the reference; a day


@pytest.mark.parametrize("kernel", [0, 1, 2])
def test_generate_dt_off_grid_times(gpu, oracle, kernel):
    """dcs_bf_generate_dt: fDeltaTime by value, every launch shape, against the verifier at the same fDeltaTime."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    bp = BeamformerParameters(NR_CHANNELS=24, NR_STATIONS=5, NR_BEAMS=13)
    op = oracle.params_from(bp)
    table = rand_table(bp.n_pairs, seed=71)
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(table)
    dts = np.array(OFF_GRID_DT, dtype=np.float32)
    nb = g.output_bytes(1, dts.size)
    buf = gpu.mem_alloc(nb)
    g.generate_dt(buf, nb, dts, kernel=kernel)
    got = np.empty((dts.size, bp.NR_CHANNELS, bp.NR_STATIONS, bp.NR_BEAMS, 2), dtype=np.float32)
    gpu.memcpy_dtoh(got, buf)
    exp = oracle.generate_dt(op, table, dts)
    _check(oracle, got, exp)
    assert not np.array_equal(got[1], got[2])
    if kernel != 0:  # b16 (NAIVE has no b16: BeamformerCoefficientTest.cu:40-44)
        nb16 = g.output_bytes(0, dts.size)
        g.generate_dt(buf, nb16, dts, kernel=kernel, bitwidth=0)
        h16 = np.empty(got.shape, dtype=np.float16)
        gpu.memcpy_dtoh(h16, buf)
        assert np.array_equal(h16.view(np.uint16), got.astype(np.float16).view(np.uint16))
    # a channel slab through dcs_bf_generate_slab_dt
    c0, nc = 7, 9
    nbs = dts.size * nc * bp.n_pairs * 8
    g.generate_slab_dt(buf, nbs, c0, nc, dts)
    gs = np.empty((dts.size, nc, bp.NR_STATIONS, bp.NR_BEAMS, 2), dtype=np.float32)
    gpu.memcpy_dtoh(gs, buf)
    _check(oracle, gs, np.ascontiguousarray(exp[:, c0:c0 + nc]))  # (NAIVE picks its polynomial degree per pair, the tiled form per wave: not bit-equal)
    g.close()


@pytest.mark.parametrize("form", [1, 2])
def test_generate_dt_more_steps_than_ride_in_the_kernel_arguments(gpu, oracle, form):
    """300 > 256 time steps 200 us apart (the staged fDeltaTime table), both forms; and 4100 > 4096 (two launches)."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    bp = BeamformerParameters(NR_CHANNELS=5, NR_STATIONS=3, NR_BEAMS=6)
    op = oracle.params_from(bp)
    table = rand_table(bp.n_pairs, seed=72)
    g = SteeringCoefficientGenerator(bp)
    g.set_tuning(form=form)
    g.upload_delays(table)
    for nt in (300, 4100):
        dts = (np.arange(nt, dtype=np.float64) * 200e-6).astype(np.float32)
        nb = g.output_bytes(1, nt)
        buf = gpu.mem_alloc(nb)
        g.generate_dt(buf, nb, dts)
        got = np.empty((nt, bp.NR_CHANNELS, bp.NR_STATIONS, bp.NR_BEAMS, 2), dtype=np.float32)
        gpu.memcpy_dtoh(got, buf)
        _check(oracle, got, oracle.generate_dt(op, table, dts))
        buf.free()
    g.close()


@pytest.mark.parametrize("kernel", [0, 1, 2])
def test_non_finite_delay_values_give_the_verifiers_nans(gpu, oracle, kernel):
    """Infinities and NaNs in the delay table (and a rate of 1e38): the coefficients of those pairs are NaN exactly where
    the verifier's are (cos(inf) = NaN, inf * 0 = NaN at channel 0 ...), every other element is within 1 ULP, nothing
    hangs.  The reference does not test this; the slow class (IEEE divide, fp64 sincos) is what these pairs fall into."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    bp = BeamformerParameters(NR_CHANNELS=6, NR_STATIONS=3, NR_BEAMS=7)
    op = oracle.params_from(bp)
    table = rand_table(bp.n_pairs, seed=5)
    table["fDelayRate_sps"][2] = np.inf
    table["fDelay_s"][5] = np.nan
    table["fPhase_rad"][9] = -np.inf
    table["fPhaseRate_radps"][13] = np.nan
    table["fDelayRate_sps"][17] = 1e38
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(table)
    dts = np.array([0.0, 0.37], dtype=np.float32)
    nb = g.output_bytes(1, dts.size)
    buf = gpu.mem_alloc(nb)
    g.generate_dt(buf, nb, dts, kernel=kernel)
    got = np.empty((dts.size, 6, 3, 7, 2), dtype=np.float32)
    gpu.memcpy_dtoh(got, buf)
    exp = oracle.generate_dt(op, table, dts)
    assert np.isnan(exp).sum() > 50
    assert np.array_equal(np.isnan(exp), np.isnan(got))
    fin = ~np.isnan(exp)
    mx, n_over, _ = oracle.max_ulp(np.where(fin, got, 0).astype(np.float32), np.where(fin, exp, 0).astype(np.float32), 1)
    assert n_over == 0, mx
    if kernel != 0:  # the packed binary16 output, both arithmetic forms: NaN in the same places, the rest RN-even / within 1 ulp
        nb16 = g.output_bytes(0, dts.size)
        for math_mode in (0, 4):
            g.set_tuning(math_mode=math_mode) if math_mode else g.set_tuning()
            g.generate_dt(buf, nb16, dts, kernel=kernel, bitwidth=0)
            h16 = np.empty(got.shape, dtype=np.float16)
            gpu.memcpy_dtoh(h16, buf)
            assert np.array_equal(np.isnan(h16), np.isnan(exp)), math_mode
            want = np.where(fin, exp, 0).astype(np.float16).view(np.uint16).astype(np.int32)
            have = np.where(fin, h16, 0).astype(np.float16).view(np.uint16).astype(np.int32)
            def ordered(u):
                return np.where(u & 0x8000, -(u & 0x7FFF), u & 0x7FFF)
            assert np.abs(ordered(have) - ordered(want)).max() <= 1, math_mode
    g.close()


def test_seeded_fuzz_of_the_sampling_period_fft_size_and_channel_count(gpu, oracle):
    """40 seeded random PARAMETER sets -- SAMPLING_PERIOD from 1e-9 to 1e-5 s, FFT_SIZE 256 .. 65536, channel counts that
    are not powers of two -- i.e. forty different divisors Ts * C for the divide by the launch constant (whose 3-operation
    form dcs_bf_create verifies per divisor) and forty different time grids: every launch shape by time index, fp32 within
    1 ULP of the verifier, fp16 the RN-even image of it."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    rng = np.random.default_rng(20261008)
    for case in range(40):
        Ts = float(10.0 ** rng.uniform(-9, -5))
        fft = int(2 ** rng.integers(8, 17))
        A, B = int(rng.integers(1, 6)), int(rng.integers(1, 30))
        C = int(rng.choice([1, 2, 3, 7, 48, 100, 257, 1000, 1023, 1025])) if case % 2 else int(rng.integers(1, 300))
        bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B, SAMPLING_PERIOD=Ts, FFT_SIZE=fft)
        op = oracle.params_from(bp)
        table = rand_table(bp.n_pairs, seed=9000 + case, Ts=Ts)
        nt = int(rng.integers(1, 4))
        t0 = int(rng.integers(0, 3000))
        kernel = int(rng.integers(0, 3))
        g = SteeringCoefficientGenerator(bp)
        g.upload_delays(table)
        nb = g.output_bytes(1, nt)
        buf = gpu.mem_alloc(nb)
        g.generate(buf, nb, t0=t0, nt=nt, kernel=kernel)
        got = np.empty((nt, C, A, B, 2), dtype=np.float32)
        gpu.memcpy_dtoh(got, buf)
        exp = oracle.generate(op, table, t0, nt)
        tag = (case, Ts, fft, A, B, C, nt, t0, kernel)
        mx, n_over, first = oracle.max_ulp(got, exp, 1)
        assert n_over == 0, (tag, mx, n_over, first)
        if kernel != 0:
            g.generate(buf, g.output_bytes(0, nt), t0=t0, nt=nt, kernel=kernel, bitwidth=0)
            h16 = np.empty(got.shape, dtype=np.float16)
            gpu.memcpy_dtoh(h16, buf)
            assert np.array_equal(h16.view(np.uint16), got.astype(np.float16).view(np.uint16)), tag
        g.close()
        buf.free()


def test_streaming_seeded_fuzz_of_tick_kinds_slabs_and_table_updates(gpu, oracle):
    """12 seeded streams (hipGraph replay, BASELINE configs[4]'s plumbing) x 8 ticks each: a random channel slab and
    output width per stream, every tick by time index, by fDeltaTime or by a (current, reference) pair, a new delay table
    landing with some of the ticks (double-buffered), the slab read back after every tick: fp32 within 1 ULP of the
    verifier at that tick's time and table, fp16 within one binary16 ulp of its RN-even image."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator, delta_times

    rng = np.random.default_rng(20261009)
    for case in range(12):
        A, B, C = int(rng.integers(1, 7)), int(rng.integers(1, 40)), int(rng.integers(2, 60))
        bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B)
        op = oracle.params_from(bp)
        table = rand_table(bp.n_pairs, seed=11000 + case)
        c0 = int(rng.integers(0, C))
        nc = int(rng.integers(1, C - c0 + 1))
        bw = int(rng.integers(0, 2))
        g = SteeringCoefficientGenerator(bp)
        st = gpu.Stream()
        g.upload_delays(table, stream=st)
        nbytes = nc * bp.n_pairs * (8 if bw == 1 else 4)
        buf = gpu.mem_alloc(nbytes)
        s_ = g.stream_begin(buf, nbytes, c0, nc, st, bitwidth=bw)
        for tick in range(8):
            new = None
            if rng.integers(0, 3) == 0:
                table = rand_table(bp.n_pairs, seed=12000 + 10 * case + tick)
                new = table
            kind = int(rng.integers(0, 3))
            if kind == 0:
                t = int(rng.integers(0, 5000))
                s_.tick(t, new_table=new)
                dt = delta_times(bp, t, 1)[0]
            elif kind == 1:
                dt = np.float32(rng.uniform(-2.0, 2.0))
                s_.tick_dt(float(dt), new_table=new)
            else:
                ref = (int(rng.integers(0, 10 ** 6)), int(rng.integers(0, 10 ** 9)))
                cur = (ref[0] + int(rng.integers(0, 3)), int(rng.integers(0, 10 ** 9)))
                s_.tick_at(cur, ref, new_table=new)
                dt = oracle.ts_diff(ref, cur)
            st.synchronize()
            exp = oracle.generate_dt(op, table, [dt], c0, nc)[0]
            tag = (case, tick, kind, A, B, C, c0, nc, bw, float(dt))
            if bw == 1:
                got = np.empty(exp.shape, dtype=np.float32)
                gpu.memcpy_dtoh(got, buf)
                mx, n_over, first = oracle.max_ulp(got, exp, 1)
                assert n_over == 0, (tag, mx, n_over, first)
            else:
                h16 = np.empty(exp.shape, dtype=np.float16)
                gpu.memcpy_dtoh(h16, buf)
                have = h16.view(np.uint16).astype(np.int32)
                want = exp.astype(np.float16).view(np.uint16).astype(np.int32)
                ordered = lambda u: np.where(u & 0x8000, -(u & 0x7FFF), u & 0x7FFF)
                assert np.abs(ordered(have) - ordered(want)).max() <= 1, tag
        s_.end()
        g.close()
        buf.free()


def test_two_contexts_on_two_streams_at_once(gpu, oracle):
    """The one-stream-at-a-time rule is per CONTEXT (include/dcs_beamformer.h, dcs_bf_create): two contexts, each on its
    own stream, enqueue launches of every kind turn by turn -- generators in both widths, a streaming graph, both
    beamformers -- without a host synchronisation in between, and each gets its own results."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator, delta_times

    shapes = [BeamformerParameters(NR_CHANNELS=300, NR_STATIONS=8, NR_BEAMS=40, NR_SAMPLES_PER_CHANNEL=32),
              BeamformerParameters(NR_CHANNELS=77, NR_STATIONS=66, NR_BEAMS=24, NR_SAMPLES_PER_CHANNEL=32)]
    ctx = []
    for i, bp in enumerate(shapes):
        g = SteeringCoefficientGenerator(bp)
        st = gpu.Stream()
        table = rand_table(bp.n_pairs, seed=400 + i)
        g.upload_delays(table, stream=st)
        nt = 3
        nb = g.output_bytes(1, nt)
        ant = np.random.default_rng(i).integers(-128, 128, size=(bp.NR_CHANNELS, 2, bp.NR_STATIONS, 16, 2), dtype=np.int8)
        d_ant = gpu.mem_alloc(ant.nbytes)
        gpu.memcpy_htod(d_ant, ant, stream=st)
        ctx.append(dict(bp=bp, g=g, st=st, table=table, nt=nt, nb=nb, out32=gpu.mem_alloc(nb), out16=gpu.mem_alloc(nb // 2), ant=ant, d_ant=d_ant,
                        acc=gpu.mem_alloc(bp.NR_BEAMS * bp.NR_CHANNELS * 32 * 8), fused=gpu.mem_alloc(bp.NR_BEAMS * bp.NR_CHANNELS * 32 * 8)))
    for rep in range(3):  # turn by turn, nothing waits
        for c in ctx:
            c["g"].generate(c["out32"], c["nb"], t0=5, nt=c["nt"], stream=c["st"])
        for c in ctx:
            c["g"].generate(c["out16"], c["nb"] // 2, t0=5, nt=c["nt"], bitwidth=0, stream=c["st"])
        for c in ctx:
            c["g"].beamform_accumulated(c["d_ant"], c["ant"].nbytes, c["acc"], c["bp"].NR_BEAMS * c["bp"].NR_CHANNELS * 32 * 8, 32, t_coeff=9, stream=c["st"])
        for c in ctx:
            c["g"].generate_and_beamform(c["d_ant"], c["ant"].nbytes, c["fused"], c["bp"].NR_BEAMS * c["bp"].NR_CHANNELS * 32 * 8, t0=0, nt=32, stream=c["st"])
    for c in ctx:
        c["st"].synchronize()
        bp, op = c["bp"], oracle.params_from(c["bp"])
        got = np.empty((c["nt"], bp.NR_CHANNELS, bp.NR_STATIONS, bp.NR_BEAMS, 2), dtype=np.float32)
        gpu.memcpy_dtoh(got, c["out32"])
        table_ab = c["table"]
        _check(oracle, got, oracle.generate(op, table_ab, 5, c["nt"]))
        h16 = np.empty(got.shape, dtype=np.float16)
        gpu.memcpy_dtoh(h16, c["out16"])
        assert np.array_equal(h16.view(np.uint16), got.astype(np.float16).view(np.uint16))
        exp_acc = oracle.beamform_accumulated(op, c["table"], delta_times(bp, 9, 1)[0], 32, c["ant"])
        acc = np.empty_like(exp_acc)
        gpu.memcpy_dtoh(acc, c["acc"])
        assert np.abs(acc - exp_acc).max() <= 4e-5 * bp.NR_STATIONS + 1e-6
        exp_f = oracle.beamform(op, c["table"], 32, c["ant"])
        fused = np.empty_like(exp_f)
        gpu.memcpy_dtoh(fused, c["fused"])
        assert np.abs(fused - exp_f).max() <= 2e-5 * bp.NR_STATIONS + 1e-6
        c["g"].close()


def test_generate_dt_seeded_fuzz_over_the_whole_time_and_rate_range(gpu, oracle):
    """50 seeded random cases of dcs_bf_generate_dt: fDeltaTime from 1e-7 s to 1e4 s of either sign (and 0), delay
    tables whose rates span nine decades on top of the usual ones -- so that waves land in every class (low-degree,
    full, slow: |fRotation| from 1e-3 to far beyond 32000) -- every launch shape, several time steps at once, fp32 within
    1 ULP of the verifier (the tolerance rule applies only where |fRotation| is small enough for 1e-4 to mean anything)
    and fp16 the RN-even image of the fp32 run."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    rng = np.random.default_rng(20261007)
    for case in range(50):
        A, B, C = int(rng.integers(1, 7)), int(rng.integers(1, 40)), int(rng.integers(1, 50))
        bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B)
        op = oracle.params_from(bp)
        table = rand_table(bp.n_pairs, seed=7000 + case)
        if case % 3 == 0:  # a few pairs with rates far outside the usual range
            k = rng.integers(0, bp.n_pairs, size=max(1, bp.n_pairs // 5))
            table["fDelayRate_sps"][k] = (10.0 ** rng.uniform(-12, -3, size=k.size)) * rng.choice([-1.0, 1.0], size=k.size)
            table["fPhaseRate_radps"][k] = (10.0 ** rng.uniform(-9, 2, size=k.size)) * rng.choice([-1.0, 1.0], size=k.size)
        nt = int(rng.integers(1, 6))
        dts = ((10.0 ** rng.uniform(-7, 4, size=nt)) * rng.choice([-1.0, 1.0], size=nt)).astype(np.float32)
        if case % 7 == 0:
            dts[0] = 0.0
        kernel = int(rng.integers(0, 3))
        g = SteeringCoefficientGenerator(bp)
        g.upload_delays(table)
        nb = g.output_bytes(1, nt)
        buf = gpu.mem_alloc(nb + 64)
        gpu.memset(buf, 0xFF, nb + 64)
        g.generate_dt(buf, nb, dts, kernel=kernel)
        host = np.empty(nb + 64, dtype=np.uint8)
        gpu.memcpy_dtoh(host, buf)
        tag = (case, A, B, C, nt, kernel, dts.tolist())
        assert np.all(host[nb:] == 0xFF), tag
        got = host[:nb].view(np.float32).reshape(nt, C, A, B, 2)
        exp = oracle.generate_dt(op, table, dts)
        mx, n_over, first = oracle.max_ulp(got, exp, 1)
        assert n_over == 0, (tag, mx, n_over, first)
        if kernel != 0:
            nb16 = g.output_bytes(0, nt)
            g.generate_dt(buf, nb16, dts, kernel=kernel, bitwidth=0)
            h16 = np.empty(got.shape, dtype=np.float16)
            gpu.memcpy_dtoh(h16, buf)
            assert np.array_equal(h16.view(np.uint16), got.astype(np.float16).view(np.uint16)), tag
        g.close()
        buf.free()


@pytest.mark.parametrize("kernel", [0, 1, 2])
def test_generate_at_timespec_pairs_across_a_seconds_boundary(gpu, oracle, kernel):
    """dcs_bf_generate_at: (current, reference) as the reference's kernels take them.  Times 200 us apart that cross a
    seconds boundary, given normalised (tv_sec + 1, small tv_nsec: a negative nanosecond difference) and
    un-normalised (tv_nsec >= 1e9, as the reference's own launch loop builds them, BeamformerCoefficientTest.cu:232-236)."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    bp = BeamformerParameters(NR_CHANNELS=17, NR_STATIONS=4, NR_BEAMS=9)
    op = oracle.params_from(bp)
    table = rand_table(bp.n_pairs, seed=73)
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(table)
    ref = (4242, 999_700_000)
    steps = [k * 200_000 for k in range(8)]  # ns; the boundary falls between k = 1 and k = 2
    unnorm = [(ref[0], ref[1] + st) for st in steps]
    norm = [(ref[0] + (ref[1] + st) // 10 ** 9, (ref[1] + st) % 10 ** 9) for st in steps]
    assert norm[2][0] == ref[0] + 1
    nb = g.output_bytes(1, len(steps))
    buf = gpu.mem_alloc(nb)
    outs = []
    for cur in (unnorm, norm):
        g.generate_at(buf, nb, cur, ref, kernel=kernel)
        got = np.empty((len(steps), bp.NR_CHANNELS, bp.NR_STATIONS, bp.NR_BEAMS, 2), dtype=np.float32)
        gpu.memcpy_dtoh(got, buf)
        _check(oracle, got, oracle.generate_at(op, table, cur, ref))
        outs.append(got)
    # the two spellings of the same instants give fDeltaTime values that agree to an fp32 rounding, not always bit for bit
    assert np.max(np.abs(outs[0] - outs[1])) < 1e-3
    g.close()


def test_fused_generate_and_beamform_dt(gpu, oracle):
    """dcs_bf_generate_and_beamform_dt: 32 samples 200 us apart."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    A, B, C, nt = 12, 20, 6, 32
    bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B, NR_SAMPLES_PER_CHANNEL=nt)
    op = oracle.params_from(bp)
    table = rand_table(bp.n_pairs, seed=74)
    ant = np.random.default_rng(10).integers(-128, 128, size=(C, nt // 16, A, 16, 2), dtype=np.int8)
    dts = (np.arange(nt, dtype=np.float64) * 200e-6 + 0.25).astype(np.float32)
    exp = oracle.beamform_dt(op, table, dts, ant)
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(table)
    d_ant = gpu.mem_alloc(ant.nbytes)
    gpu.memcpy_htod(d_ant, ant)
    got = np.empty_like(exp)
    d_beams = gpu.mem_alloc(got.nbytes)
    g.generate_and_beamform_dt(d_ant, ant.nbytes, d_beams, got.nbytes, dts)
    gpu.memcpy_dtoh(got, d_beams)
    assert np.abs(got - exp).max() <= 2e-5 * A + 1e-6
    g.close()


def test_b16_arithmetic_form_device_equals_host_sweep(gpu, probes):
    """The device evaluates dcs_sincos_half2 (math_mode bit 2) to the same bits as the host build swept exhaustively
    in tests/test_numerics.py -- including the compiler's v_fma_mix{lo,hi}_f16 fusion of the last fma with the
    conversion, which must round like fmaf-then-convert (8M arguments: a single-rounding fma would differ on ~2^-13
    of them)."""
    import ctypes
    import subprocess
    from pathlib import Path

    lab_dir = Path(__file__).resolve().parent / "numerics"
    assert subprocess.run(["make", "-C", str(lab_dir)], capture_output=True).returncode == 0
    L = ctypes.CDLL(str(lab_dir / "libnumerics_lab.so"))
    L.lab_sincos_half2.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    rng = np.random.default_rng(2)
    x = np.concatenate([
        rng.uniform(-100, 100, 1 << 22).astype(np.float32),
        rng.uniform(-500, 500, 1 << 21).astype(np.float32),
        rng.uniform(-32000, 32000, 1 << 21).astype(np.float32),
        (np.arange(1, 20000, dtype=np.float32) * np.float32(np.pi / 2)),
        np.arange(0x3F800000, 0x3F800000 + (1 << 21), dtype=np.uint32).view(np.float32),  # a dense run of [1, 1.25)
    ])
    n = x.size
    dx, dw = gpu.mem_alloc(4 * n), gpu.mem_alloc(4 * n)
    gpu.memcpy_htod(dx, x)
    probes.sincos(4, dx, n, dw, dw)
    gpu.synchronize()
    got = np.empty(n, np.uint32)
    gpu.memcpy_dtoh(got, dw)
    exp = np.empty(n, np.uint32)
    L.lab_sincos_half2(x.ctypes.data, n, exp.ctypes.data)
    bad = np.flatnonzero(got != exp)
    assert bad.size == 0, (bad.size, x[bad[:4]], got[bad[:4]], exp[bad[:4]])


@pytest.mark.parametrize("kernel", [1, 2])
def test_b16_arithmetic_form_in_the_generator(gpu, oracle, kernel):
    """math_mode = 4: b16 output from the binary16-sized arithmetic.  Every half within one binary16 ulp of
    RN16(oracle fp32) (the bar the default b16 form is held to) -- also for pairs between 500 and 32000 rad; workgroups
    holding a slow-class pair (|fRotation| >= 32000) keep the fp64 path and so reproduce the default output bit for bit;
    fp32 output is unaffected."""
    from dc_sand_amd import BeamformerParameters

    def ordered(h):
        i = h.view(np.int16).astype(np.int32)
        return np.where(i < 0, -(i & 0x7FFF), i)

    for (A, B, C) in ((64, 16, 64), (3, 5, 17), (2, 600, 40)):
        bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B)
        table = rand_table(bp.n_pairs, seed=C + 1)
        if A == 2:
            table["fPhase_rad"][0:100] = 2000.0     # full-degree class, still the b16 form
            table["fPhase_rad"][600] = 40000.0      # slow class: the 256-pair tile 512..767 keeps the fp64 path
        op = oracle.params_from(bp)
        exp = oracle.generate(op, table, 5, 3).astype(np.float16)
        dflt = _gen(gpu, bp, table, 5, 3, kernel=kernel, bitwidth=0)
        fast = _gen(gpu, bp, table, 5, 3, kernel=kernel, bitwidth=0, tuning=dict(math_mode=4))
        fast_tt = _gen(gpu, bp, table, 5, 3, kernel=kernel, bitwidth=0, tuning=dict(math_mode=4, form=3))  # terms from the pre-pass table
        assert np.array_equal(fast.view(np.uint16), fast_tt.view(np.uint16))
        assert np.max(np.abs(ordered(fast) - ordered(exp))) <= 1
        assert np.max(np.abs(ordered(dflt) - ordered(exp))) <= 1
        differ = np.mean(fast.view(np.uint16) != dflt.view(np.uint16))
        assert differ < 0.02, differ  # ~0.5 % of halves
        if A == 2:
            f = fast.reshape(3, C, bp.n_pairs, 2).view(np.uint16)
            d = dflt.reshape(3, C, bp.n_pairs, 2).view(np.uint16)
            assert np.array_equal(f[:, :, 512:768], d[:, :, 512:768])
            assert not np.array_equal(f[:, :, 0:256], d[:, :, 0:256])
        f32a = _gen(gpu, bp, table, 5, 3, kernel=kernel, bitwidth=1)
        f32b = _gen(gpu, bp, table, 5, 3, kernel=kernel, bitwidth=1, tuning=dict(math_mode=4))
        assert np.array_equal(f32a.view(np.uint32), f32b.view(np.uint32))


def test_autotune_measures_a_large_shape_once(gpu, oracle):
    """dcs_bf_autotune on a tensor large enough to be tuned (2 GiB: 64 x 256 x 16384): the first call measures (~1 s), the
    second returns the cached choice at once, dcs_bf_set_tuning(NULL) forgets it; whatever was chosen, sampled rows are
    the oracle's."""
    import time

    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    bp = BeamformerParameters(NR_CHANNELS=16384, NR_STATIONS=64, NR_BEAMS=256)
    op = oracle.params_from(bp)
    table = rand_table(bp.n_pairs, seed=91)
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(table)
    nbytes = g.output_bytes(1, 1)
    buf = gpu.mem_alloc(nbytes)
    t0 = time.perf_counter()
    chosen = g.autotune(buf, nbytes)
    t_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    assert g.autotune(buf, nbytes) == chosen
    t_second = time.perf_counter() - t0
    assert t_first > 0.2 and t_second < 0.02, (t_first, t_second)
    g.generate(buf, nbytes, t0=9, nt=1)
    row = bp.n_pairs * 8
    host = np.empty((bp.NR_STATIONS, bp.NR_BEAMS, 2), dtype=np.float32)
    for ch in (0, 4097, 16383):
        gpu.memcpy_dtoh(host, int(buf) + ch * row)
        assert oracle.max_ulp(host, oracle.generate(op, table, 9, 1, ch, 1), 1)[1] == 0
    # a 1 GiB slab of the same shape takes the OTHER kernel variant (terms computed per workgroup, not read from the pre-pass
    # table): its geometry is measured and cached separately, and the full tensor's stays cached beside it
    t0 = time.perf_counter()
    slab_choice = g.autotune(buf, nbytes // 2)
    assert time.perf_counter() - t0 > 0.2
    t0 = time.perf_counter()
    assert g.autotune(buf, nbytes) == chosen and g.autotune(buf, nbytes // 2) == slab_choice
    assert time.perf_counter() - t0 < 0.02
    g.generate_slab(buf, nbytes // 2, 100, bp.NR_CHANNELS // 2, t0=9, nt=1)
    gpu.memcpy_dtoh(host, int(buf) + 17 * row)
    assert oracle.max_ulp(host, oracle.generate(op, table, 9, 1, 117, 1), 1)[1] == 0
    g.set_tuning()  # forgets the cached choices
    t0 = time.perf_counter()
    g.autotune(buf, nbytes)
    assert time.perf_counter() - t0 > 0.2
    g.close()


# ---- beamformer with coefficient reuse on the matrix cores ("general version")
@pytest.mark.parametrize("math_mode", [0, 8])
@pytest.mark.parametrize("A,B,C,nt", [(64, 16, 64, 256), (64, 16, 5, 32), (64, 64, 7, 64), (64, 40, 3, 48), (8, 4, 5, 16), (37, 21, 9, 48),
                                       (130, 3, 4, 16), (4, 40, 7, 32), (9, 5, 3, 16), (129, 33, 2, 32), (256, 17, 2, 16), (1, 1, 1, 16),
                                       (66, 70, 2, 80), (128, 16, 3, 64), (192, 48, 2, 32), (200, 20, 2, 32), (64, 1024, 1, 32),
                                       (64, 32, 3, 112), (64, 24, 2, 272), (64, 16, 2, 592), (48, 16, 3, 48), (64, 64, 2, 272),
                                       (256, 64, 2, 272), (100, 20, 3, 112), (256, 16, 1, 1600), (65, 16, 2, 48),
                                       # enough workgroups for the XCD-grouped numbering to leave its identity tail (> 8 x the sharers):
                                       (64, 1024, 9, 32), (130, 20, 9, 32), (256, 64, 9, 16), (192, 48, 11, 48),
                                       # more than 64 beams at <= 64 antennas (several beam groups per channel, ragged last group)
                                       (64, 128, 3, 64), (48, 200, 2, 48), (64, 72, 2, 32), (33, 129, 2, 16), (64, 256, 2, 272)])
def test_beamform_accumulated_on_the_matrix_cores(gpu, oracle, A, B, C, nt, math_mode):
    """dcs_bf_beamform_accumulated: the coefficients of ONE time applied to nt samples as two real contractions over the
    antennas on the matrix cores, against the verifier's beamformer with the coefficient held
    (BeamformerCoefficientTest.cu:363-414).  Both forms: the default exact fixed-point contraction on the int8 pipe
    (24-bit coefficients as three signed digits; sums exact) and (math_mode 8) the fp32 fma chain on
    v_mfma_f32_16x16x4_f32.  The verifier multiplies and adds with separate roundings, and each coefficient is within
    1 ULP: |difference| <= 4e-5 * A (1 ulp of the coefficient + quantisation + the roundings of either side <= 3e-7 per unit of
    sample, samples <= 128: reached at A = 1, ten times less at 64 antennas); the reference's own tolerance is
    1e-1 (runBeamformerTests.cpp:15).  The fixed-point form is also held to its own, much tighter, bound against the
    sum in exact (fp64) arithmetic of the oracle's fp32 coefficients: 2.5e-7 * sum_a |sample_a| + 2e-7 * |sum|.
    Shapes cover every beam-tile count (1, 2, 4 per workgroup: coefficients shared by 4, 2, 1 waves), ragged antennas /
    beams / sample blocks, odd block counts and more than 16 blocks per (channel, beam group) (several workgroups), 1-4
    64-antenna chunks (whole and partial) and the 256-antenna limit -- above 64 antennas the kChain form (round 3: the
    coefficients in LDS, the integer sums chained through the chunks by the matrix instruction's accumulator), with fewer
    sample blocks than waves (waves that only make their chunk's coefficients), odd pair counts and several workgroups
    per channel."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator, delta_times, simulate_input

    bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B, NR_SAMPLES_PER_CHANNEL=nt)
    op = oracle.params_from(bp)
    seeded = (A, B, C, nt) != (64, 16, 64, 256)
    table = rand_table(bp.n_pairs, seed=A + B) if seeded else simulate_input(bp)  # indexed [b*A + a]
    ant = oracle.simulate_antenna_data(op, nt)
    if seeded:
        ant = np.random.default_rng(A).integers(-128, 128, size=ant.shape, dtype=np.int8)
    t_coeff = 9
    exp = oracle.beamform_accumulated(op, table, delta_times(bp, t_coeff, 1)[0], nt, ant)
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(table)
    if math_mode:
        g.set_tuning(math_mode=math_mode)
    d_ant = gpu.mem_alloc(ant.nbytes)
    gpu.memcpy_htod(d_ant, ant)
    d_beams = gpu.mem_alloc(exp.nbytes + 64)
    gpu.memset(d_beams, 0xFF, exp.nbytes + 64)
    g.beamform_accumulated(d_ant, ant.nbytes, d_beams, exp.nbytes, nt, t_coeff=t_coeff)
    host = np.empty(exp.nbytes + 64, dtype=np.uint8)
    gpu.memcpy_dtoh(host, d_beams)
    assert np.all(host[exp.nbytes:] == 0xFF)
    got = host[:exp.nbytes].view(np.float32).reshape(exp.shape)
    assert np.all(np.isfinite(got))
    diff = np.abs(got - exp)
    assert diff.max() <= 4e-5 * A + 1e-6, diff.max()
    assert oracle.compare(got, exp, 1e-1) == -1
    if math_mode == 0:
        # exact-arithmetic sum of the oracle's coefficients (table [b*A + a] -> the generator's [a*B + b])
        coef = oracle.generate_dt(op, np.ascontiguousarray(table.reshape(B, A).T).ravel(), delta_times(bp, t_coeff, 1)[0])[0].astype(np.float64)
        x = ant.reshape(C, nt // 16, A, 16, 2).astype(np.float64)
        exact = np.einsum("cabk,ctaik->ctbik", coef, x)  # [c][t/16][b][t%16][re, im]
        mag = np.abs(x).sum(axis=2)[:, :, None, :, :]   # sum_a |sample|, per (c, t/16, t%16, plane)
        assert np.all(np.abs(got - exact) <= 2.5e-7 * mag + 2e-7 * np.abs(exact) + 1e-30)
    # fDeltaTime by value gives the same bits
    g.beamform_accumulated(d_ant, ant.nbytes, d_beams, exp.nbytes, nt, dt_coeff=float(delta_times(bp, t_coeff, 1)[0]))
    got2 = np.empty(exp.shape, dtype=np.float32)
    gpu.memcpy_dtoh(got2, d_beams)
    assert np.array_equal(got2.view(np.uint32), got.view(np.uint32))
    g.close()


def test_beamform_accumulated_seeded_fuzz(gpu, oracle):
    """60 seeded random cases of the coefficient-reuse beamformer: antennas 1..256 (every form: staged, K-split with
    2-4 chunks, whole and partial), beams 1..90 (1, 2, 4 tiles per workgroup, partial tiles), channels, 1..40 sample
    blocks (odd counts, several workgroups per channel), either arithmetic form, time by index or by value -- each
    within 4e-5 * A of the verifier's loop with the coefficient held, the fixed-point form also within its own bound of
    the exact sum, nothing written outside the tensor."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator, delta_times

    rng = np.random.default_rng(20261005)
    for case in range(60):
        A = int(rng.choice([1, 3, 17, 63, 64, 65, 100, 128, 129, 191, 192, 200, 255, 256])) if case % 2 else int(rng.integers(1, 257))
        B = int(rng.integers(1, 91))
        C = int(rng.integers(1, 5))
        nblk = int(rng.integers(1, 41))
        if A * B * C * nblk > 600000:  # keep the oracle's work small
            nblk = max(1, 600000 // (A * B * C))
        nt = 16 * nblk
        math_mode = 8 if case % 5 == 0 else 0
        bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B, NR_SAMPLES_PER_CHANNEL=nt)
        op = oracle.params_from(bp)
        table = rand_table(bp.n_pairs, seed=3000 + case)  # indexed [b*A + a]
        ant = rng.integers(-128, 128, size=(C, nblk, A, 16, 2), dtype=np.int8)
        by_value = bool(rng.integers(0, 2))
        t_coeff = int(rng.integers(0, 2000))
        dt = np.float32(rng.uniform(0.0, 1.5)) if by_value else delta_times(bp, t_coeff, 1)[0]
        exp = oracle.beamform_accumulated(op, table, dt, nt, ant)
        g = SteeringCoefficientGenerator(bp)
        g.upload_delays(table)
        if math_mode:
            g.set_tuning(math_mode=math_mode)
        d_ant = gpu.mem_alloc(ant.nbytes)
        gpu.memcpy_htod(d_ant, ant)
        d_beams = gpu.mem_alloc(exp.nbytes + 64)
        gpu.memset(d_beams, 0xFF, exp.nbytes + 64)
        if by_value:
            g.beamform_accumulated(d_ant, ant.nbytes, d_beams, exp.nbytes, nt, dt_coeff=float(dt))
        else:
            g.beamform_accumulated(d_ant, ant.nbytes, d_beams, exp.nbytes, nt, t_coeff=t_coeff)
        host = np.empty(exp.nbytes + 64, dtype=np.uint8)
        gpu.memcpy_dtoh(host, d_beams)
        tag = (case, A, B, C, nt, math_mode, by_value)
        assert np.all(host[exp.nbytes:] == 0xFF), tag
        got = host[:exp.nbytes].view(np.float32).reshape(exp.shape)
        assert np.all(np.isfinite(got)), tag
        assert np.abs(got - exp).max() <= 4e-5 * A + 1e-6, (tag, float(np.abs(got - exp).max()))
        if math_mode == 0:
            coef = oracle.generate_dt(op, np.ascontiguousarray(table.reshape(B, A).T).ravel(), dt)[0].astype(np.float64)
            x = ant.astype(np.float64)
            exact = np.einsum("cabk,ctaik->ctbik", coef, x)
            mag = np.abs(x).sum(axis=2)[:, :, None, :, :]
            assert np.all(np.abs(got - exact) <= 2.5e-7 * mag + 2e-7 * np.abs(exact) + 1e-30), tag
        g.close()
        d_ant.free()
        d_beams.free()


@pytest.mark.parametrize("math_mode", [0, 8])
@pytest.mark.parametrize("A,B,C,nt", [(20, 18, 3, 32), (64, 16, 2, 64), (130, 20, 2, 32), (256, 5, 1, 16)])
def test_beamform_accumulated_non_finite_coefficients(gpu, oracle, A, B, C, nt, math_mode):
    """An infinite or NaN delay value makes that pair's coefficient NaN and, with it, every sample of its beam (the plane
    concerned) in the verifier's sum: NaN * 0 = NaN.  Both forms give NaN in exactly those places -- the fixed-point form,
    whose digits cannot hold a NaN, by marking the rows -- and the other beams are unaffected."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B, NR_SAMPLES_PER_CHANNEL=nt)
    op = oracle.params_from(bp)
    table = rand_table(bp.n_pairs, seed=5)  # [b*A + a]
    table["fDelayRate_sps"][2 * A + min(3, A - 1)] = np.inf       # beam 2
    table["fPhase_rad"][(B - 1) * A + A - 1] = np.nan              # the last beam, the last antenna
    ant = np.random.default_rng(1).integers(-128, 128, size=(C, nt // 16, A, 16, 2), dtype=np.int8)
    dt = np.float32(0.25)
    exp = oracle.beamform_accumulated(op, table, dt, nt, ant)
    assert np.isnan(exp).any() and not np.isnan(exp).all()
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(table)
    if math_mode:
        g.set_tuning(math_mode=math_mode)
    d_ant = gpu.mem_alloc(ant.nbytes)
    gpu.memcpy_htod(d_ant, ant)
    d_beams = gpu.mem_alloc(exp.nbytes)
    g.beamform_accumulated(d_ant, ant.nbytes, d_beams, exp.nbytes, nt, dt_coeff=float(dt))
    got = np.empty_like(exp)
    gpu.memcpy_dtoh(got, d_beams)
    assert np.array_equal(np.isnan(exp), np.isnan(got))
    fin = ~np.isnan(exp)
    assert np.abs(np.where(fin, got - exp, 0)).max() <= 4e-5 * A + 1e-6
    g.close()


@pytest.mark.parametrize("A,B", [(64, 16), (130, 20)])
def test_beamform_accumulated_under_stream_capture(gpu, oracle, A, B):
    """dcs_bf_beamform_accumulated is two kernel launches and nothing else once the context's terms table exists (its one
    fDeltaTime travels in the kernel arguments): captured into a hipGraph after a first, plain, call and replayed -- a
    real-time beamformer re-launching one graph per block of samples."""
    import ctypes

    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    hip = ctypes.CDLL("libamdhip64.so")
    V = ctypes.c_void_p
    hip.hipStreamBeginCapture.argtypes = [V, ctypes.c_int]
    hip.hipStreamEndCapture.argtypes = [V, ctypes.POINTER(V)]
    hip.hipGraphInstantiate.argtypes = [ctypes.POINTER(V), V, V, V, ctypes.c_size_t]
    hip.hipGraphLaunch.argtypes = [V, V]
    hip.hipGraphExecDestroy.argtypes = [V]
    hip.hipGraphDestroy.argtypes = [V]

    C, nt = 5, 48
    bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B, NR_SAMPLES_PER_CHANNEL=nt)
    op = oracle.params_from(bp)
    table = rand_table(bp.n_pairs, seed=61)
    rng = np.random.default_rng(7)
    ant = rng.integers(-128, 128, size=(C, nt // 16, A, 16, 2), dtype=np.int8)
    dt = np.float32(0.0421)
    exp = oracle.beamform_accumulated(op, table, dt, nt, ant)
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(table)
    d_ant = gpu.mem_alloc(ant.nbytes)
    gpu.memcpy_htod(d_ant, ant)
    d_beams = gpu.mem_alloc(exp.nbytes)
    s = gpu.Stream()
    g.beamform_accumulated(d_ant, ant.nbytes, d_beams, exp.nbytes, nt, dt_coeff=float(dt), stream=s.handle)  # allocates the terms table
    s.synchronize()
    assert hip.hipStreamBeginCapture(V(s.handle), 0) == 0
    g.beamform_accumulated(d_ant, ant.nbytes, d_beams, exp.nbytes, nt, dt_coeff=float(dt), stream=s.handle)
    graph = V()
    assert hip.hipStreamEndCapture(V(s.handle), ctypes.byref(graph)) == 0 and graph.value
    ex = V()
    assert hip.hipGraphInstantiate(ctypes.byref(ex), graph, None, None, 0) == 0
    for rep in range(3):
        ant2 = rng.integers(-128, 128, size=ant.shape, dtype=np.int8)  # new samples, same coefficients: what a replay is for
        gpu.memcpy_htod(d_ant, ant2, stream=s.handle)
        gpu.memset(d_beams, 0xFF, exp.nbytes, stream=s.handle)
        assert hip.hipGraphLaunch(ex, V(s.handle)) == 0
        s.synchronize()
        got = np.empty_like(exp)
        gpu.memcpy_dtoh(got, d_beams)
        assert np.abs(got - oracle.beamform_accumulated(op, table, dt, nt, ant2)).max() <= 4e-5 * A + 1e-6
    hip.hipGraphExecDestroy(ex)
    hip.hipGraphDestroy(graph)
    g.close()


def test_beamform_accumulated_slow_class_and_limits(gpu, oracle):
    """A pair outside the fast path's range sends the coefficient generation down the slow branch (IEEE divide, fp64
    sincos); more than 256 antennas and sample counts that are not whole 16-sample blocks are refused."""
    from dc_sand_amd import BeamformerParameters, _lib
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    bp = BeamformerParameters(NR_CHANNELS=6, NR_STATIONS=9, NR_BEAMS=5, NR_SAMPLES_PER_CHANNEL=32)
    op = oracle.params_from(bp)
    table = rand_table(bp.n_pairs, seed=77)
    table["fDelayRate_sps"][11] = 1e-2
    table["fDelayRate_sps"][12] = 1e-30
    ant = np.random.default_rng(3).integers(-128, 128, size=(6, 2, 9, 16, 2), dtype=np.int8)
    dt = np.float32(0.25)
    exp = oracle.beamform_accumulated(op, table, dt, 32, ant)
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(table)
    d_ant = gpu.mem_alloc(ant.nbytes)
    gpu.memcpy_htod(d_ant, ant)
    d_beams = gpu.mem_alloc(exp.nbytes)
    for math_mode in (0, 8):  # the fixed-point form and the fp32 chain
        g.set_tuning(math_mode=math_mode)
        gpu.memset(d_beams, 0xFF, exp.nbytes)
        g.beamform_accumulated(d_ant, ant.nbytes, d_beams, exp.nbytes, 32, dt_coeff=float(dt))
        got = np.empty_like(exp)
        gpu.memcpy_dtoh(got, d_beams)
        assert np.abs(got - exp).max() <= 2e-4 * 9
    g.set_tuning()
    with pytest.raises(_lib.DcsError) as e:
        g.beamform_accumulated(d_ant, ant.nbytes, d_beams, exp.nbytes, 24, t_coeff=0)
    assert e.value.status == _lib.DCS_ERR_INVALID_ARGUMENT
    g.close()
    big = SteeringCoefficientGenerator(BeamformerParameters(NR_CHANNELS=1, NR_STATIONS=257, NR_BEAMS=1))
    big.upload_delays(rand_table(257))
    with pytest.raises(_lib.DcsError) as e:
        big.beamform_accumulated(d_ant, 257 * 32, d_beams, 16 * 8, 16, t_coeff=0)
    assert e.value.status == _lib.DCS_ERR_UNSUPPORTED
    big.close()
