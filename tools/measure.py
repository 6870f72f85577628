#!/usr/bin/env python3
"""tools/measure.py -- the measurement scripts behind profiles/ and DESIGN.md, one CLI (needs an MI355X).

Nothing here is on the product path: every subcommand drives the library through the same ctypes C-ABI
the tests use.  The store-only studies need the probes build (include/dcs_probes.h):

    python tools/measure.py geometry --shapes cfg2,mid,cfg3,cfg4,narrow [--bits 32|16] [--sweep]
        default geometry vs dcs_bf_autotune vs an exhaustive sweep, per shape      -> profiles/r02_autotune.md
    python tools/measure.py refshape
        the reference's default tensor (64 x 64 x 16 x 256 steps): one launch, and the per-time-step
        launch shapes NAIVE / MULTIPLE_CHANNELS (a1 / a2)
    python tools/measure.py fp16 [--modes 0,4]
        fp16 generator rate per arithmetic form                                     -> profiles/r02_fp16.md
    python tools/measure.py fused
        fused generate + beamform rate on several shapes                            -> profiles/r0N_fused.md
    python tools/measure.py stream
        BASELINE configs[4]: full-tensor period and the largest slab at <= 200 us    -> profiles/r0N_streaming_config5.md
    python tools/measure.py pmc
        a few launches of each hot kernel, for `rocprofv3 --pmc ... -- python3 tools/measure.py pmc`
    python tools/measure.py sustained [--seconds 6]
        back-to-back launches at config 3 for several seconds (rate per second)
    DCS_LIB_PATH=probes/libdcs_probes.so python tools/measure.py stores --kind pattern|lean|kernel ...
        store-only probes and the real kernel with dcs_probe_knobs nomath / pace        -> profiles/r01_store_patterns.md
    python tools/measure.py sincos
        device sweep of the sincos forms over every fp32 in [1, 128)                 -> profiles/r01_sincos_ab.md
"""
from __future__ import annotations

import argparse
import ctypes
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from dc_sand_amd import BeamformerParameters, device  # noqa: E402
from dc_sand_amd.generator import SteeringCoefficientGenerator, simulate_input  # noqa: E402

SHAPES = {  # name: (ant, beams, chan, nt)
    "ref": (64, 16, 64, 256),
    "cfg2": (64, 64, 4096, 1),
    "mid": (64, 256, 8192, 1),
    "cfg3": (64, 1024, 32768, 1),
    "cfg4": (256, 512, 32768, 1),
    "narrow": (16, 16, 32768, 1),
    "wide": (64, 4096, 2048, 1),
    "small": (16, 64, 2048, 1),
    "small2": (64, 16, 512, 1),
    "small3": (16, 64, 4096, 1),
    "wide2": (256, 1024, 1024, 1),
}


def per_launch_ms(fn, settle_ms=30.0, timed_ms=12.0, stream=None, max_n=4000):
    """ms per call of ``fn`` at steady state: settle ~settle_ms on this access pattern (the first launches
    after a change of pattern run 3-10 % slow), then ONE event pair around ~timed_ms worth of calls."""
    e0, e1 = device.Event(), device.Event()
    fn()
    e0.record(stream)
    fn()
    fn()
    e1.record(stream)
    e1.synchronize()
    one = max(e1.elapsed_ms_since(e0) / 2, 1e-3)
    for _ in range(int(min(max_n, max(4, settle_ms / one)))):
        fn()
    n = int(min(max_n, max(4, timed_ms / one)))
    e0.record(stream)
    for _ in range(n):
        fn()
    e1.record(stream)
    e1.synchronize()
    return e1.elapsed_ms_since(e0) / n


def make(shape, bits=32):
    A, B, C, nt = shape
    bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B)
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(simulate_input(bp))
    bw = 1 if bits == 32 else 0
    nb = g.output_bytes(bw, nt)
    buf = device.mem_alloc(nb)
    return bp, g, bw, nb, buf


def cmd_geometry(args):
    rows = []
    for name in args.shapes.split(","):
        shape = SHAPES[name]
        bp, g, bw, nb, buf = make(shape, args.bits)
        nt = shape[3]
        n = bp.coeffs_per_time_step() * nt
        run = lambda: g.generate(buf, nb, t0=1, nt=nt, bitwidth=bw)  # noqa: E731
        g.set_tuning()
        d = [per_launch_ms(run) for _ in range(2)]
        chosen = g.autotune(buf, nb, bitwidth=bw)
        a = [per_launch_ms(run) for _ in range(2)]
        g.set_tuning()
        d.append(per_launch_ms(run))
        g.set_tuning(**{k: chosen[k] for k in ("form", "tiles_per_block", "chan_per_block", "nontemporal", "wg_per_cu")})
        a.append(per_launch_ms(run))
        best = (1e9, None)
        table = []
        if args.sweep:
            cpbs = (4, 6, 8, 10, 11, 12, 13, 14, 16, 20, 24, 32) if args.bits == 32 else (16, 32, 64, 96, 128, 192, 256)
            for tpb, wpc in ((1, -1), (1, 7), (1, 6), (1, 5), (2, -1), (4, -1)):
                for cpb in cpbs:
                    g.set_tuning(form=args.form, tiles_per_block=tpb, chan_per_block=cpb, nontemporal=1, wg_per_cu=wpc)
                    ms = per_launch_ms(run, settle_ms=20.0, timed_ms=8.0)
                    table.append((ms, tpb, cpb, wpc))
                    if ms < best[0]:
                        best = (ms, (tpb, cpb, wpc))
            # second look at the five best (short trials rank neighbours only within their noise)
            top = sorted(table)[:5]
            best = (1e9, None)
            for _, tpb, cpb, wpc in top:
                g.set_tuning(form=args.form, tiles_per_block=tpb, chan_per_block=cpb, nontemporal=1, wg_per_cu=wpc)
                ms = min(per_launch_ms(run) for _ in range(2))
                if ms < best[0]:
                    best = (ms, (tpb, cpb, wpc))
        dm, am = min(d), min(a)
        line = (f"{name} {shape[0]}x{shape[1]}x{shape[2]} nt={nt} b{args.bits}: default {n / dm / 1e6:.1f}  autotuned {n / am / 1e6:.1f} "
                f"(tpb={chosen['tiles_per_block']} cpb={chosen['chan_per_block']} wg_per_cu={chosen['wg_per_cu']})")
        if best[1]:
            line += f"  sweep best {n / best[0] / 1e6:.1f} {best[1]}  default/best {best[0] / dm:.3f}  autotuned/default {dm / am:.3f}"
        print(line + "  Gcoeff/s", flush=True)
        if args.sweep and args.verbose:
            for ms, tpb, cpb, wpc in sorted(table)[:12]:
                print(f"    tpb={tpb} cpb={cpb:3d} wpc={wpc:2d}: {n / ms / 1e6:.1f}", flush=True)
        rows.append(dict(shape=name, dims=shape, bits=args.bits, default=n / dm / 1e6, autotuned=n / am / 1e6, chosen=chosen,
                         sweep_best=(n / best[0] / 1e6 if best[1] else None), sweep_best_geometry=best[1]))
        g.close()
        buf.free()
    print("JSON", json.dumps(rows), flush=True)


def cmd_refshape(args):
    """The tensor runBeamformerTests times (BeamformerParameters.h defaults): a3 in one launch, a1 / a2 as
    256 launches from the host loop (BeamformerCoefficientTest.cu:230-250); --sweep: a2 / a3 per geometry."""
    bp = BeamformerParameters()
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(simulate_input(bp))
    nt = 256
    nb = g.output_bytes(1, nt)
    buf = device.mem_alloc(nb)

    def t(kern, bw):
        nbb = g.output_bytes(bw, nt)
        return nbb, min(per_launch_ms(lambda: g.generate(buf, nbb, t0=0, nt=nt, kernel=kern, bitwidth=bw), settle_ms=20, timed_ms=40)
                        for _ in range(3))

    for kern, name in ((2, "MULTIPLE_CHANNELS_AND_TIMESTAMPS (1 launch)"), (1, "MULTIPLE_CHANNELS (256 launches)"), (0, "NAIVE (256 launches)")):
        for bw in ((1, 0) if kern else (1,)):
            nbb, ms = t(kern, bw)
            print(f"{name} b{32 if bw else 16}: {ms * 1e3:.1f} us per tensor = {ms * 1e3 / (nt if kern != 2 else 1):.2f} us per launch, "
                  f"{nbb / ms / 1e9:.2f} TB/s", flush=True)
    if args.sweep:
        for kern in (1, 2):
            for bw in (1, 0):
                for tpb in (1, 2, 4):
                    for cpb in (1, 2, 4, 8, 12, 16, 32, 64):
                        if cpb * tpb < 4:
                            continue
                        g.set_tuning(form=1, tiles_per_block=tpb, chan_per_block=cpb, wg_per_cu=-1)
                        nbb, ms = t(kern, bw)
                        print(f"  kernel={kern} b{32 if bw else 16} tpb={tpb} cpb={cpb:2d}: {ms * 1e3 / (nt if kern != 2 else 1):.2f} us per launch", flush=True)
    g.close()


def cmd_fp16(args):
    bp, g, _, _, buf = make(SHAPES["cfg3"], 32)
    n = bp.coeffs_per_time_step()
    bw = 1 if args.bits == 32 else 0
    nb16 = g.output_bytes(bw, 1)
    res = []
    for mode in [int(m) for m in args.modes.split(",")]:
        for tpb in [int(c) for c in args.tpb.split(",")]:
            for wpc in [int(c) for c in args.wpc.split(",")]:
                for cpb in [int(c) for c in args.cpb.split(",")]:
                    g.set_tuning(form=args.form, tiles_per_block=tpb, chan_per_block=cpb, nontemporal=1, math_mode=mode, wg_per_cu=wpc)
                    ms = min(per_launch_ms(lambda: g.generate(buf, nb16, t0=1, nt=1, bitwidth=bw)) for _ in range(2))
                    res.append((ms, mode, tpb, wpc, cpb))
                    print(f"fp16 math_mode={mode} tpb={tpb} cpb={cpb} wpc={wpc}: {ms:.4f} ms -> {n / ms / 1e6:.1f} Gcoeff/s = {nb16 / ms / 1e9:.2f} TB/s "
                          f"({nb16 / ms / 1e9 / 8 * 100:.1f} % of 8 TB/s)", flush=True)
    for mode in sorted({r[1] for r in res}):
        b = min(r for r in res if r[1] == mode)
        print(f"best math_mode={mode}: tpb={b[2]} cpb={b[4]} wpc={b[3]} -> {n / b[0] / 1e6:.1f} Gcoeff/s", flush=True)
    g.set_tuning(math_mode=int(args.modes.split(",")[-1]))
    ms = min(per_launch_ms(lambda: g.generate(buf, nb16, t0=1, nt=1, bitwidth=0)) for _ in range(2))
    print(f"library default geometry, math_mode={args.modes.split(',')[-1]}: {n / ms / 1e6:.1f} Gcoeff/s", flush=True)
    g.close()


def cmd_fused(args):
    for (A, B, C, nt) in ((64, 16, 64, 256), (64, 16, 4096, 256), (64, 64, 4096, 64), (64, 256, 4096, 16), (256, 64, 1024, 64)):
        bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B, NR_SAMPLES_PER_CHANNEL=nt)
        g = SteeringCoefficientGenerator(bp)
        g.upload_delays(simulate_input(bp))
        ab, bb = A * C * nt * 2, B * C * nt * 8
        d_ant, d_beams = device.mem_alloc(ab), device.mem_alloc(bb)
        device.memset(d_ant, 3, ab)
        ms = per_launch_ms(lambda: g.generate_and_beamform(d_ant, ab, d_beams, bb, 0, nt))
        prods = A * B * C * nt
        print(f"{A}ant x {B}beam x {C}chan x {nt}t: fused {ms * 1e3:.1f} us -> {prods / ms / 1e6:.1f} G coefficient-products/s", flush=True)
        g.close()


def cmd_bfacc(args):
    """Beamformer with coefficient reuse on the matrix cores: rate against its roofline (int8 samples in + fp32 beams
    out vs 8 TB/s; fp32 MFMA 2 * 2 * A * B flop per sample vs 155 TFLOP/s)."""
    shapes = ((64, 16, 64, 256), (64, 16, 4096, 256), (64, 16, 4096, 4096), (64, 64, 4096, 256), (64, 256, 1024, 256), (256, 64, 1024, 256),
              (64, 1024, 256, 256))
    if args.shape:
        shapes = (tuple(int(v) for v in args.shape.split("x")),)
    for (A, B, C, nt) in shapes:
        bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B, NR_SAMPLES_PER_CHANNEL=nt)
        g = SteeringCoefficientGenerator(bp)
        g.upload_delays(simulate_input(bp))
        ab, bb = A * C * nt * 2, B * C * nt * 8
        d_ant, d_beams = device.mem_alloc(ab), device.mem_alloc(bb)
        if args.random:  # noise-like samples (what a telescope delivers): a 32 MiB seeded block repeated
            blk = min(ab, 32 << 20)
            device.memcpy_htod(d_ant, np.random.default_rng(0xA17).integers(-128, 128, size=blk, dtype=np.int8))
            off = blk
            while off < ab:
                n = min(off, ab - off)
                device.memcpy_dtod(int(d_ant) + off, d_ant, n)
                off += n
            device.synchronize()
        else:
            device.memset(d_ant, 3, ab)
        for mode in (int(m) for m in args.modes.split(",")):
            g.set_tuning(math_mode=mode)
            ms = min(per_launch_ms(lambda: g.beamform_accumulated(d_ant, ab, d_beams, bb, nt, t_coeff=1)) for _ in range(2))
            flop = 4.0 * A * B * C * nt
            form = "fp32 chain" if mode & 8 else "int8 fixed point"
            print(f"{A}ant x {B}beam x {C}chan x {nt}samples [{form}]: {ms * 1e3:.1f} us -> {A * B * C * nt / ms / 1e9:.2f} T coefficient-products/s, "
                  f"{(ab + bb) / ms / 1e9:.2f} TB/s algorithmic ({(ab + bb) / ms / 1e9 / 8 * 100:.1f} % of 8 TB/s), {flop / ms / 1e9:.1f} TFLOP/s-equivalent "
                  f"({flop / ms / 1e9 / 155 * 100:.1f} % of the fp32 MFMA peak)", flush=True)
        g.close()
        d_ant.free()
        d_beams.free()


def cmd_copy(args):
    """Mixed read + write ceiling: device-to-device copies (lean kernel in address order; hipMemcpyDtoD)."""
    from probes import dcs_probes as pr
    from dc_sand_amd import _lib

    nbytes = 4 * 2 ** 30
    a, b = device.mem_alloc(nbytes), device.mem_alloc(nbytes)
    device.memset(a, 1, nbytes)
    for per in (1, 2, 4, 8):
        for mode in (0, 1):
            ms = min(per_launch_ms(lambda: pr.copy(a, b, nbytes, mode, per), timed_ms=30) for _ in range(2))
            print(f"copy kernel, {per} x 16 B per thread, {'nontemporal' if mode else 'plain'} stores: {2 * nbytes / ms / 1e9:.2f} TB/s (read + write)", flush=True)
    ms = min(per_launch_ms(lambda: _lib.check(_lib.lib().dcs_memcpy_dtod(ctypes.c_void_p(int(b)), ctypes.c_void_p(int(a)), nbytes, None), "dtod"),
                           timed_ms=30) for _ in range(2))
    print(f"hipMemcpyDtoDAsync: {2 * nbytes / ms / 1e9:.2f} TB/s (read + write)", flush=True)


def cmd_mfma(args):
    """fp32 matrix-core issue rate on register operands (what the coefficient-reuse beamformer is measured against)."""
    from probes import dcs_probes as pr

    out = device.mem_alloc(1 << 20)
    for which, name, flop in ((0, "v_mfma_f32_16x16x4_f32, 2 accumulators", 2048), (1, "v_mfma_f32_16x16x4_f32, 4 accumulators", 2048),
                              (2, "v_mfma_f32_32x32x2_f32, 2 accumulators", 4096), (3, "16x16x4 with int8->fp32 conversions between", 2048),
                              (4, "16x16x4, A operands from LDS + conversions (the beamformer k-step)", 2048)):
        for blocks in (256, 512, 1024, 2048):
            iters = 4096
            ms = min(per_launch_ms(lambda: pr.mfma(which, blocks, iters, out), settle_ms=20, timed_ms=20) for _ in range(2))
            n = blocks * 4 * iters * (16 if which != 2 else 8)
            print(f"{name}, {blocks} workgroups ({blocks * 4 / 1024:.0f} waves per SIMD): {n * flop / ms / 1e9:.1f} TFLOP/s", flush=True)


def cmd_stream(args):
    bp, g, _, full, buf = make(SHAPES["cfg3"], 32)
    table = simulate_input(bp)
    stream = device.Stream()
    step_s = args.model_step_us * 1e-6

    def period_us(fn, ticks=200, warm=20):
        for i in range(warm):
            fn(i)
        stream.synchronize()
        e0, e1 = device.Event(), device.Event()
        t0 = time.perf_counter()
        e0.record(stream)
        for i in range(ticks):
            fn(warm + i)
        e1.record(stream)
        e1.synchronize()
        return e1.elapsed_ms_since(e0) / ticks * 1e3, (time.perf_counter() - t0) / ticks * 1e6

    out = {"config": "64ant x 1024beam x 32768chan, fp32, one time step per tick; model time advances "
                     f"{args.model_step_us} us per tick (dcs_bf_stream_tick_dt)", "cadence_target_us": 200.0, "slabs": []}

    def measure(nc, with_updates=False):
        nbytes = nc * bp.n_pairs * 8
        st = g.stream_begin(buf, nbytes, 0, nc, stream)
        if with_updates:
            dev_us, wall_us = period_us(lambda i: st.tick_dt(i * step_s, table if i % 16 == 0 else None), ticks=100, warm=10)
        else:
            dev_us, wall_us = period_us(lambda i: st.tick_dt(i * step_s))
        st.end()
        p_us, p_wall = period_us(lambda i: g.generate_slab_dt(buf, nbytes, 0, nc, [i * step_s], stream=stream))
        return dict(channels=nc, bytes=nbytes, graph_period_us=dev_us, graph_wall_us=wall_us, plain_period_us=p_us, plain_wall_us=p_wall,
                    graph_TBps=nbytes / dev_us / 1e6, plain_TBps=nbytes / p_us / 1e6)

    out["full_tensor"] = measure(bp.NR_CHANNELS)
    print("full tensor:", json.dumps(out["full_tensor"]), flush=True)
    for nc in (256, 512, 1024, 1536, 2048, 2304, 2560, 2816, 3072, 4096):
        r = measure(nc)
        out["slabs"].append(r)
        print(json.dumps(r), flush=True)
    ok = [s for s in out["slabs"] if max(s["graph_period_us"], s["graph_wall_us"]) <= 200.0]
    out["largest_slab_at_200us_graph"] = max(ok, key=lambda s: s["channels"]) if ok else None
    okp = [s for s in out["slabs"] if max(s["plain_period_us"], s["plain_wall_us"]) <= 200.0]
    out["largest_slab_at_200us_plain"] = max(okp, key=lambda s: s["channels"]) if okp else None
    out["with_table_update_every_16_ticks_2048ch"] = measure(2048, True)
    print("SUMMARY", json.dumps(out), flush=True)
    g.close()


def cmd_pmc(args):
    bp, g, _, nb, buf = make(SHAPES["cfg3"], 32)
    for _ in range(6):
        g.generate(buf, nb, t0=1, nt=1, bitwidth=1)
    for mode in (0, 4):
        g.set_tuning(math_mode=mode)
        for _ in range(6):
            g.generate(buf, g.output_bytes(0, 1), t0=1, nt=1, bitwidth=0)
    device.synchronize()
    g.close()
    A, B, C, nt = 64, 64, 4096, 64
    bp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B, NR_SAMPLES_PER_CHANNEL=nt)
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(simulate_input(bp))
    d_ant = device.mem_alloc(A * C * nt * 2)
    device.memset(d_ant, 3, A * C * nt * 2)
    d_beams = device.mem_alloc(B * C * nt * 8)
    for _ in range(6):
        g.generate_and_beamform(d_ant, A * C * nt * 2, d_beams, B * C * nt * 8, 0, nt)
    device.synchronize()


def cmd_sustained(args):
    bp, g, bw, nb, buf = make(SHAPES["cfg3"], 32)
    n = bp.coeffs_per_time_step()
    if args.autotune:
        print("autotune:", g.autotune(buf, nb), flush=True)
    t_end = time.perf_counter() + args.seconds
    k = 0
    while time.perf_counter() < t_end:
        e0, e1 = device.Event(), device.Event()
        e0.record()
        for _ in range(100):
            g.generate(buf, nb, t0=1 + (k % 255), nt=1)
            k += 1
        e1.record()
        e1.synchronize()
        ms = e1.elapsed_ms_since(e0) / 100
        print(f"t={args.seconds - (t_end - time.perf_counter()):5.2f} s: {ms:.4f} ms -> {n / ms / 1e6:.1f} Gcoeff/s, {nb / ms / 1e9:.3f} TB/s", flush=True)
    g.close()


def cmd_stores(args):
    """Store-only probes (needs DCS_LIB_PATH=probes/libdcs_probes.so for --kind kernel)."""
    from probes import dcs_probes as pr

    nbytes = 16 * 2 ** 30
    buf = device.mem_alloc(nbytes)
    if args.kind == "pattern":  # rows x cols KiB matrix, workgroup rectangles of rb rows x qb KiB
        cols = args.cols_kib
        rows = nbytes // (cols * 1024)
        for qb in [int(v) for v in args.qb.split(",")]:
            for rb in [int(v) for v in args.rb.split(",")]:
                for mode in [int(v) for v in args.mode.split(",")]:
                    ms = per_launch_ms(lambda: pr.store_pattern(buf, rows, cols, qb, rb, args.order, args.xcd, mode, args.threads))
                    print(f"pattern qb={qb} rb={rb} order={args.order} xcd={args.xcd} mode={mode} threads={args.threads}: {nbytes / ms / 1e9:.2f} TB/s", flush=True)
    elif args.kind == "lean":  # no loop, no division: spt stores per thread, optional sleep before each
        for spt in [int(v) for v in args.spt.split(",")]:
            for pace in [int(v) for v in args.pace.split(",")]:
                for mode in [int(v) for v in args.mode.split(",")]:
                    ms = per_launch_ms(lambda: pr.one_store(buf, nbytes, mode | (pace << 8), spt, 512 * 1024))
                    print(f"lean stores/thread={spt} pace={pace} mode={mode}: {nbytes / ms / 1e9:.2f} TB/s", flush=True)
    else:  # the real kernel, with and without arithmetic, paced
        bp = BeamformerParameters(NR_CHANNELS=32768, NR_STATIONS=64, NR_BEAMS=1024)
        g = SteeringCoefficientGenerator(bp)
        g.upload_delays(simulate_input(bp))
        for cpb in [int(v) for v in args.cpb.split(",")]:
            for pace in [int(v) for v in args.pace.split(",")]:
                for nomath in (False, True):
                    g.set_tuning(form=1, tiles_per_block=1, chan_per_block=cpb, nontemporal=1)
                    pr.set_knobs(g, pace=pace, nomath=int(nomath))  # include/dcs_probes.h (the probes build behind the wrappers)
                    ms = per_launch_ms(lambda: g.generate(buf, nbytes, t0=1, nt=1))
                    print(f"kernel cpb={cpb} pace={pace} nomath={int(nomath)}: {nbytes / ms / 1e9:.2f} TB/s", flush=True)
        g.close()
    ms = per_launch_ms(lambda: device.memset(buf, 0, nbytes))
    print(f"hipMemsetAsync: {nbytes / ms / 1e9:.2f} TB/s", flush=True)


def cmd_sincos(args):
    """Every fp32 in [1, 128) through the device sincos forms against (float)sin((double)x)."""
    from probes import dcs_probes as pr

    x = np.arange(0x3F800000, 0x43000000, dtype=np.uint32).view(np.float32)
    n = x.size
    dx, ds, dc = device.mem_alloc(4 * n), device.mem_alloc(4 * n), device.mem_alloc(4 * n)
    device.memcpy_htod(dx, x)
    es = np.sin(x.astype(np.float64)).astype(np.float32).view(np.int32).astype(np.int64)
    ec = np.cos(x.astype(np.float64)).astype(np.float32).view(np.int32).astype(np.int64)
    for which, name in ((0, "library fast path (full polynomials)"), (3, "library fast path (low degree)"), (1, "__ocml_sincos_f32"), (2, "fp64 slow path")):
        pr.sincos(which, dx, n, ds, dc)
        device.synchronize()
        s, c = np.empty(n, np.float32), np.empty(n, np.float32)
        device.memcpy_dtoh(s, ds)
        device.memcpy_dtoh(c, dc)
        us = np.abs(np.where(s.view(np.int32) < 0, -(s.view(np.int32).astype(np.int64) & 0x7FFFFFFF), s.view(np.int32).astype(np.int64)) - np.where(es < 0, -(es & 0x7FFFFFFF), es))
        uc = np.abs(np.where(c.view(np.int32) < 0, -(c.view(np.int32).astype(np.int64) & 0x7FFFFFFF), c.view(np.int32).astype(np.int64)) - np.where(ec < 0, -(ec & 0x7FFFFFFF), ec))
        print(f"{name}: sin max {us.max()} ULP ({int((us > 1).sum())} over 1), cos max {uc.max()} ULP ({int((uc > 1).sum())} over 1)", flush=True)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    p = sub.add_parser("geometry")
    p.add_argument("--shapes", default="cfg2,mid,cfg3,cfg4,narrow")
    p.add_argument("--bits", type=int, default=32, choices=[16, 32])
    p.add_argument("--sweep", action="store_true")
    p.add_argument("--form", type=int, default=0, help="form of the sweep's explicit geometries (0 = library's choice, 1 / 3)")
    p.add_argument("--verbose", action="store_true")
    p = sub.add_parser("refshape")
    p.add_argument("--sweep", action="store_true")
    p = sub.add_parser("fp16")
    p.add_argument("--modes", default="0,4")
    p.add_argument("--cpb", default="64,128,256")
    p.add_argument("--wpc", default="-1")
    p.add_argument("--tpb", default="1")
    p.add_argument("--form", type=int, default=1, help="1 = per-workgroup terms, 3 = terms table, 0 = library's choice")
    p.add_argument("--bits", type=int, default=16, choices=[16, 32])
    sub.add_parser("fused")
    sub.add_parser("mfma")
    sub.add_parser("copy")
    p = sub.add_parser("bfacc")
    p.add_argument("--shape", default="", help="AxBxCxNT: one shape only (PMC passes)")
    p.add_argument("--modes", default="0,8", help="math_mode values: 0 = int8 fixed point, 8 = fp32 chain")
    p.add_argument("--random", action="store_true", help="noise-like int8 samples instead of a constant byte (the matrix pipe's power depends on the data)")
    p = sub.add_parser("stream")
    p.add_argument("--model-step-us", type=float, default=200.0)
    sub.add_parser("pmc")
    p = sub.add_parser("sustained")
    p.add_argument("--seconds", type=float, default=6.0)
    p.add_argument("--autotune", action="store_true")
    p = sub.add_parser("stores")
    p.add_argument("--kind", default="lean", choices=["pattern", "lean", "kernel"])
    p.add_argument("--qb", default="1")
    p.add_argument("--rb", default="4,8,16")
    p.add_argument("--order", type=int, default=0)
    p.add_argument("--xcd", type=int, default=0)
    p.add_argument("--mode", default="1")
    p.add_argument("--threads", type=int, default=256)
    p.add_argument("--cols-kib", type=int, default=512, help="row length of the pattern matrix in KiB (32: the beamformer's output at 256 beams)")
    p.add_argument("--spt", default="1,2,3,4,8")
    p.add_argument("--pace", default="0")
    p.add_argument("--cpb", default="8,12,16")
    sub.add_parser("sincos")
    args = ap.parse_args()
    device.require_device()
    device.set_device(0)
    print("device:", device.device_name(0), flush=True)
    {"geometry": cmd_geometry, "refshape": cmd_refshape, "fp16": cmd_fp16, "fused": cmd_fused, "mfma": cmd_mfma, "copy": cmd_copy, "bfacc": cmd_bfacc, "stream": cmd_stream, "pmc": cmd_pmc,
     "sustained": cmd_sustained, "stores": cmd_stores, "sincos": cmd_sincos}[args.cmd](args)


if __name__ == "__main__":
    main()
