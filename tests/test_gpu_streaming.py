"""GPU parity of the streaming path (BASELINE configs[4]) beyond the single-node graph: the TWO-NODE graph (terms
pre-pass + generator, what slabs of >= 2 GiB and ``form = 3`` build -- the path ``bench.py``'s full-tensor
``streaming_cfg5`` figure runs on) and the ticks that take their new delay table from DEVICE memory
(``dcs_bf_stream_tick_*_from_global``: a gather node in the replayed graph; configs[3] + configs[4] composed).

Everything is compared with the CPU oracle evaluated at the tick's own fDeltaTime and table -- never with another GPU
launch.  Bar as in test_gpu_parity.py: fp32 within 1 ULP, binary16 within one binary16 ulp of RN-even(oracle).
"""
import numpy as np
import pytest

from conftest import rand_table

pytestmark = pytest.mark.gpu


def _ordered16(u):
    return np.where(u & 0x8000, -(u & 0x7FFF), u & 0x7FFF)


def _check_slab(oracle, gpu, buf, exp, bitwidth, tag):
    """exp: the oracle's fp32 slab; NaN exactly where the verifier has NaN, the rest within the bar."""
    fin = ~np.isnan(exp)
    if bitwidth == 1:
        got = np.empty(exp.shape, dtype=np.float32)
        gpu.memcpy_dtoh(got, buf)
        assert np.array_equal(np.isnan(got), ~fin), tag
        mx, n_over, first = oracle.max_ulp(np.where(fin, got, 0).astype(np.float32), np.where(fin, exp, 0).astype(np.float32), 1)
        assert n_over == 0, (tag, mx, n_over, first)
    else:
        h16 = np.empty(exp.shape, dtype=np.float16)
        gpu.memcpy_dtoh(h16, buf)
        assert np.array_equal(np.isnan(h16), ~fin), tag
        have = np.where(fin, h16, 0).astype(np.float16).view(np.uint16).astype(np.int32)
        want = np.where(fin, exp, 0).astype(np.float16).view(np.uint16).astype(np.int32)
        assert np.abs(_ordered16(have) - _ordered16(want)).max() <= 1, tag


def _tables(n_pairs, seeds, slow_in=()):
    """Seeded tables; those listed in ``slow_in`` carry slow-class pairs (|fRotation| far beyond 32000, a rate of
    1e38, an infinity, a NaN): with the terms table it is the PRE-PASS node that writes those tiles into d_out."""
    out = []
    for i, s in enumerate(seeds):
        t = rand_table(n_pairs, seed=s)
        if i in slow_in:
            t["fDelayRate_sps"][5 % n_pairs] = 1e-2
            t["fDelayRate_sps"][(n_pairs // 2 + 3) % n_pairs] = 1e38
            t["fPhase_rad"][(n_pairs - 2) % n_pairs] = np.inf
            t["fDelay_s"][(n_pairs // 3) % n_pairs] = np.nan
        out.append(t)
    return out


@pytest.mark.parametrize("bitwidth,math_mode", [(1, 0), (0, 0), (0, 4)])
def test_two_node_streaming_graph_every_tick_against_the_oracle(gpu, oracle, bitwidth, math_mode):
    """``form = 3`` before ``stream_begin`` builds the two-node graph (bf_terms_kernel -> tiled generator reading the
    terms table) at a small shape.  Ticks by time index, by fDeltaTime and by (current, reference); a new HOST table
    on some ticks (one of them with slow-class pairs, so the pre-pass node's own stores into d_out happen inside the
    graph) and a new DEVICE table on others; the whole slab after every tick against the oracle."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator, delta_times

    bp = BeamformerParameters(NR_CHANNELS=96, NR_STATIONS=8, NR_BEAMS=40)
    op = oracle.params_from(bp)
    c0, nc = 16, 64
    tables = _tables(bp.n_pairs, (131, 132, 133, 134), slow_in=(1, 3))
    d_tables = []
    for t in tables:
        d = gpu.mem_alloc(t.nbytes)
        gpu.memcpy_htod(d, t)
        d_tables.append(d)
    g = SteeringCoefficientGenerator(bp)
    g.set_tuning(form=3, math_mode=math_mode)
    stream = gpu.Stream()
    g.upload_delays(tables[0], stream=stream)
    nbytes = nc * bp.n_pairs * (8 if bitwidth == 1 else 4)
    buf = gpu.mem_alloc(nbytes)
    st = g.stream_begin(buf, nbytes, c0, nc, stream, bitwidth=bitwidth)
    cur = 0
    ref = (41, 999_800_000)
    #        kind  time                      table source
    plan = [("t", 0, None), ("t", 9, ("host", 1)), ("dt", 200e-6, None), ("at", (42, 100), ("dev", 2)), ("dt", 0.37, ("dev", 3)),
            ("t", 255, None), ("at", (41, 999_900_000), ("host", 0)), ("dt", -2e-4, ("dev", 1)), ("t", 1000, ("host", 2)),
            ("dt", 1.5, ("dev", 0)), ("dt", 400e-6, ("dev", 3)), ("t", 18, ("host", 1)), ("t", 7, None)]
    for tick, (kind, when, src) in enumerate(plan):
        host_tbl = None
        if src is not None:
            cur = src[1]
            host_tbl = tables[cur] if src[0] == "host" else None
        dev = src is not None and src[0] == "dev"
        if kind == "t":
            dt = delta_times(bp, when, 1)[0]
            st.tick_from_global(when, d_tables[cur]) if dev else st.tick(when, host_tbl)
        elif kind == "dt":
            dt = np.float32(when)
            st.tick_dt_from_global(when, d_tables[cur]) if dev else st.tick_dt(when, host_tbl)
        else:
            dt = oracle.ts_diff(ref, when)
            st.tick_at_from_global(when, ref, d_tables[cur]) if dev else st.tick_at(when, ref, host_tbl)
        stream.synchronize()
        exp = oracle.generate_dt(op, tables[cur], [dt], c0, nc)
        _check_slab(oracle, gpu, buf, exp, bitwidth, (tick, kind, when, src))
    st.end()
    g.close()
    buf.free()
    for d in d_tables:
        d.free()


def test_streaming_from_a_global_device_table_changing_every_tick(gpu, oracle):
    """configs[3] + configs[4] composed: the context owns beams [off, off + B_loc) of a GLOBAL [A][B_total] table that
    sits in device memory (where an RCCL broadcast lands it) and CHANGES ON EVERY TICK; the tick gathers its slice
    inside the replayed graph.  Both graph shapes (one node / two nodes behind the gather), both widths; each tick's
    slab against the oracle on that tick's slice.  Ticks are queued back to back in bursts (no host
    synchronisation between them) into separate output slabs, so the double-buffered table is exercised while
    earlier replays are still in flight."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    A, B_total, C = 6, 50, 40
    rng = np.random.default_rng(77)
    n_ticks = 9
    globs = [rand_table(A * B_total, seed=900 + k) for k in range(n_ticks)]
    globs[4]["fDelayRate_sps"][3 * B_total + 20] = 2e-2  # a slow-class pair inside the second shard below
    d_globs = []
    for t in globs:
        d = gpu.mem_alloc(t.nbytes)
        gpu.memcpy_htod(d, t)
        d_globs.append(d)
    for (off, bl), form, bw in (((0, 16), 0, 1), ((16, 34), 3, 1), ((7, 9), 3, 0), ((49, 1), 0, 0)):
        bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=bl)
        op = oracle.params_from(bp)
        g = SteeringCoefficientGenerator(bp)
        if form:
            g.set_tuning(form=form)
        stream = gpu.Stream()
        g.set_delays_from_global(d_globs[0], B_total, off, stream=stream)
        c0, nc = 3, 30
        slab = nc * bp.n_pairs * (8 if bw == 1 else 4)
        slab_pad = (slab + 255) & ~255
        buf = gpu.mem_alloc(slab_pad * n_ticks)
        streams = [g.stream_begin(int(buf) + k * slab_pad, slab, c0, nc, stream, bitwidth=bw) for k in range(n_ticks)]
        dts = [np.float32(k * 200e-6 + rng.uniform(0, 1e-5)) for k in range(n_ticks)]
        for k in range(n_ticks):  # one burst: nine graph launches, nine different tables, no synchronisation
            streams[k].tick_dt_from_global(float(dts[k]), d_globs[k], B_total, off)
        stream.synchronize()
        for k in range(n_ticks):
            local = np.ascontiguousarray(globs[k].reshape(A, B_total)[:, off:off + bl]).ravel()
            exp = oracle.generate_dt(op, local, [dts[k]], c0, nc)
            _check_slab(oracle, gpu, int(buf) + k * slab_pad, exp, bw, (off, bl, form, bw, k))
        # the argument checks of the device-table ticks
        from dc_sand_amd import _lib

        with pytest.raises(_lib.DcsError) as e:
            streams[0].tick_dt_from_global(0.0, d_globs[0], B_total, B_total - bl + 1)  # slice runs past the table
        assert e.value.status == _lib.DCS_ERR_OUT_OF_RANGE
        with pytest.raises(_lib.DcsError) as e:
            streams[0].tick_dt_from_global(0.0, int(d_globs[0]) + 4, B_total, off)  # not 16-byte aligned
        assert e.value.status == _lib.DCS_ERR_INVALID_ARGUMENT
        for s_ in streams:
            s_.end()
        g.close()
        buf.free()
    for d in d_globs:
        d.free()


def test_host_table_on_every_tick_never_reuses_a_staging_buffer_in_flight(gpu, oracle):
    """A new HOST table with every tick, twelve ticks queued back to back (more than the ring of four pinned staging
    buffers): the caller's array may be overwritten as soon as the tick call returns, and every slab still shows
    its own tick's table."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    bp = BeamformerParameters(NR_CHANNELS=48, NR_STATIONS=4, NR_BEAMS=96)
    op = oracle.params_from(bp)
    n_ticks = 12
    tables = [rand_table(bp.n_pairs, seed=500 + k) for k in range(n_ticks)]
    g = SteeringCoefficientGenerator(bp)
    stream = gpu.Stream()
    g.upload_delays(tables[0], stream=stream)
    slab = bp.NR_CHANNELS * bp.n_pairs * 8
    buf = gpu.mem_alloc(slab)
    keep = gpu.mem_alloc(slab * n_ticks)  # every tick's slab is copied aside on the same stream
    st = g.stream_begin(buf, slab, 0, bp.NR_CHANNELS, stream)
    scratch = np.empty_like(tables[0])
    for k in range(n_ticks):
        scratch[:] = tables[k]
        st.tick_dt(k * 200e-6, scratch)
        scratch["fDelay_s"][:] = np.nan  # the call has copied it: the caller's buffer is free again
        gpu.memcpy_dtod(int(keep) + k * slab, buf, slab, stream)
    stream.synchronize()
    for k in range(n_ticks):
        exp = oracle.generate_dt(op, tables[k], [np.float32(k * 200e-6)])
        _check_slab(oracle, gpu, int(keep) + k * slab, exp, 1, k)
    st.end()
    g.close()
    buf.free()
    keep.free()


def test_config5_full_tensor_two_node_graph_every_element(gpu, oracle, record_property):
    """BASELINE configs[4] at FULL size: ``stream_begin`` over the whole 64 x 1024 x 32768 tensor (16 GiB >= 2 GiB: the
    two-node graph, the code ``bench.py``'s ``streaming_cfg5.full_tensor_period_us`` times), three ticks at a 200 us
    model cadence -- the second with a new host table, the third with a new DEVICE table --, then EVERY one of the 2^32
    floats of the last tick against the verifier (<= 1 ULP, both readings of cos(float))."""
    import time

    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator
    from test_gpu_parity import _compare_every_element

    bp = BeamformerParameters(NR_CHANNELS=32768, NR_STATIONS=64, NR_BEAMS=1024)
    op = oracle.params_from(bp)
    tables = [rand_table(bp.n_pairs, seed=s) for s in (71, 72, 73)]
    d_tab = gpu.mem_alloc(tables[2].nbytes)
    gpu.memcpy_htod(d_tab, tables[2])
    g = SteeringCoefficientGenerator(bp)
    stream = gpu.Stream()
    g.upload_delays(tables[0], stream=stream)
    nbytes = g.output_bytes(1, 1)
    assert nbytes == 16 * 2 ** 30
    buf = gpu.mem_alloc(nbytes)
    st = g.stream_begin(buf, nbytes, 0, bp.NR_CHANNELS, stream)
    row = bp.n_pairs * 8
    host = np.empty((bp.NR_STATIONS, bp.NR_BEAMS, 2), dtype=np.float32)
    dts = [np.float32(k * 200e-6) for k in (1, 2, 3)]
    st.tick_dt(float(dts[0]))
    stream.synchronize()
    for c in (0, 12345, 32767):  # sampled rows of the earlier ticks, everything of the last
        gpu.memcpy_dtoh(host, int(buf) + c * row)
        assert oracle.max_ulp(host, oracle.generate_dt(op, tables[0], [dts[0]], c, 1), 1)[1] == 0, c
    st.tick_dt(float(dts[1]), tables[1])
    stream.synchronize()
    for c in (1, 20000, 32766):
        gpu.memcpy_dtoh(host, int(buf) + c * row)
        assert oracle.max_ulp(host, oracle.generate_dt(op, tables[1], [dts[1]], c, 1), 1)[1] == 0, c
    st.tick_dt_from_global(float(dts[2]), d_tab)
    stream.synchronize()
    t0 = time.perf_counter()
    res = _compare_every_element(gpu, oracle, buf, op, tables[2], dts[2], bp.NR_CHANNELS, bp.n_pairs)
    wall = time.perf_counter() - t0
    n = bp.NR_CHANNELS * bp.n_pairs * 2
    for r in (0, 1):
        h = res[r]["hist"]
        assert sum(h) == n == 2 ** 32
        assert h[2] == 0 and h[3] == 0 and res[r]["max_ulp"] <= 1, (r, res[r])
    summary = (f"config 5, full tensor through the two-node streaming graph, third tick (device table), all {n} floats: "
               f"{res[0]['hist'][1]} at 1 ULP, 0 beyond (float-libm reading: {res[1]['hist'][1]}, 0 beyond); {wall:.1f} s wall")
    print(summary)
    record_property("config5_full_compare", summary)
    st.end()
    g.close()
    buf.free()
    d_tab.free()
