#!/usr/bin/env python3
"""bench.py -- steering-coefficient throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: generate one time step of
the 64 ant x 1024 beam x 32768 chan coefficient tensor (16 GiB, fp32 complex)
per GPU from a delay table already resident in HBM.  With N > 1 (one rank per GPU) the BEAM axis is sharded:
the global table holds 1024*N beams, rank 0 broadcasts it over RCCL every step
(issued asynchronously one step ahead, double-buffered, so it overlaps the generation), each rank gathers its 1024-beam
slice and generates its own column slab -- no other collective (weak scaling).

LAUNCH CONTRACT.  ``--gpus N`` ALWAYS means N ranks, one per GPU:
  * under a launcher (``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``: RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in the environment) this process IS one of the N ranks; WORLD_SIZE != N is an error;
  * without one (plain ``python bench.py --gpus N``, N > 1) this process starts the N ranks itself -- fresh child
    processes, created BEFORE anything here touches torch or the GPU, rendezvous on 127.0.0.1 -- relays rank 0's one
    JSON line and exits non-zero, printing NO line, if any rank fails (fewer than N GPUs, a rank dying, a timeout).
  It never prints an ``n_gpus: 1`` line for a ``--gpus 8`` request.
``--config cfg3`` (default) is BASELINE configs[2], the metric's config; ``--config cfg4`` is configs[3]'s per-GPU
share (256 ant x 4096/8 = 512 beams per GPU x 32768 chan); the name is in ``config.workload``.

Rank 0 prints ONE JSON line.  ``roofline`` prices the dominant kernel against
the 8 TB/s HBM peak with its ALGORITHMIC bytes (8 B per coefficient written);
``cpu_baseline`` times the CPU oracle (the restated reference verifier, one
thread like the reference) on a bounded channel slab of the same workload.
``also_measured`` (N = 1) carries what the headline's 50-step burst does not show: the SUSTAINED rate over >= 5 s of
back-to-back steps, BASELINE configs[4] (streaming at a 200 us cadence: full-tensor period and the largest slab that
keeps the cadence, model time advancing 200 us per tick), the fp16 output modes and the fused beamformer.  At N > 1
``per_rank`` holds every rank's own kernel time and launch geometry, and ``rccl_world_size`` the communicator's size.
The oracle is loaded only in that cpu_baseline leg (where it is also used to
spot-check the last generated step), never inside the timed GPU region; with
--no-cpu-baseline bench.py does not touch oracle/ at all.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s (spec)
CONFIGS = {  # SURVEY.md numbering (1-based): cfg3 = BASELINE.json configs[2], cfg4 = configs[3]
    "cfg3": dict(ant=64, beams_per_gpu=1024, chan=32768, name="BASELINE configs[2]: 64 ant x 1024 beam x 32768 chan on one GPU"),
    "cfg4": dict(ant=256, beams_per_gpu=512, chan=32768, name="BASELINE configs[3]: 256 ant x 4096 beam x 32768 chan beam-sharded over 8 GPUs "
                                                              "(512 beams per GPU)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS),
                    help="named workload (SURVEY.md numbering): cfg3 = BASELINE configs[2], 64 ant x 1024 beams per GPU x 32768 chan "
                         "(the metric's config); cfg4 = BASELINE configs[3]'s per-GPU share, 256 ant x 512 beams per GPU (4096 over "
                         "8 GPUs) x 32768 chan.  --ant / --beams-per-gpu / --chan override single dimensions")
    ap.add_argument("--ant", type=int, default=None)
    ap.add_argument("--beams-per-gpu", type=int, default=None)
    ap.add_argument("--chan", type=int, default=None)
    ap.add_argument("--streaming", action="store_true",
                    help="each step is a hipGraph replay (BASELINE configs[4]'s launch) whose delay table comes from the "
                         "broadcast's DEVICE buffer through a gather node of the graph (dcs_bf_stream_tick_dt_from_global): "
                         "configs[3] and configs[4] composed; model time advances 200 us per step")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="--gpus N > 1 without a launcher: seconds the parent waits for its N rank processes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the fp16 / fused-beamformer side measurements (N = 1)")
    ap.add_argument("--per-step-events", action="store_true",
                    help="record a HIP event after every timed step and report the per-step median / min / max "
                         "(each event is a marker packet that costs the stream ~6 us, so it is off by default)")
    ap.add_argument("--no-autotune", action="store_true",
                    help="keep the library's default launch geometry instead of letting dcs_bf_autotune measure it in the "
                         "untimed set-up (its trial launches run under separate kernel symbols, template TAG = 1, so a "
                         "rocprofv3 --stats of this command still averages only the production launches)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline sample time")
    ap.add_argument("--output-alloc", default="torch", choices=["torch", "hip"],
                    help="who allocates the output tensor: torch's caching allocator (default) or hipMalloc through dc_sand_amd.device (measured: no difference)")
    ap.add_argument("--sustain-seconds", type=float, default=5.0, help="length of the sustained-rate side measurement (N = 1)")
    # rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks (never a result):
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL over xGMI (the product path)")
    ap.add_argument("--shared-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--check-all-ranks", action="store_true", help="every rank spot-checks its slab against the oracle")
    ap.add_argument("--force-collective", action="store_true",
                    help="rehearsal on one GPU: run the N > 1 control flow (process group, per-step broadcast on the side "
                         "stream, double-buffered table, slice gather) with a world of ONE rank over RCCL")
    args = ap.parse_args()
    named = CONFIGS[args.config]
    shape_overridden = any(v is not None for v in (args.ant, args.beams_per_gpu, args.chan))
    args.ant = named["ant"] if args.ant is None else args.ant
    args.beams_per_gpu = named["beams_per_gpu"] if args.beams_per_gpu is None else args.beams_per_gpu
    args.chan = named["chan"] if args.chan is None else args.chan
    args.config_name = args.config + (" (shape overridden)" if shape_overridden else "")
    return args


def launch_ranks(args) -> int:
    """``--gpus N`` (N > 1) without a launcher: start the N ranks as fresh child processes of this command line --
    the parent has made no torch / HIP call -- with the environment a launcher would give them, relay rank 0's one
    JSON line, and report failure (non-zero, no line) if any rank does not finish cleanly."""
    import socket
    import subprocess
    import threading

    N = args.gpus
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(N):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(N), LOCAL_WORLD_SIZE=str(N), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + args.launch_timeout
    failed = None
    while failed is None:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = f"rank {bad[0][0]} exited with status {bad[0][1]}"
        elif all(c == 0 for c in codes):
            break
        elif time.monotonic() > deadline:
            failed = f"the ranks did not finish within {args.launch_timeout:.0f} s"
        else:
            time.sleep(0.05)
    if failed is not None:
        for p in procs:  # exactly the processes started here
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        print(f"bench.py --gpus {N}: {failed}; no result line is printed (a run of fewer ranks is not a run of {N})", file=sys.stderr)
        return 1
    reader.join(timeout=10)
    lines = [l for l in (out0[0] if out0 else b"").decode().splitlines() if l.strip()]
    if len(lines) != 1:
        print(f"bench.py --gpus {N}: rank 0 printed {len(lines)} lines instead of one", file=sys.stderr)
        return 1
    d = json.loads(lines[0])
    if d.get("n_gpus") != N or len(d.get("per_rank", [])) != N:
        print(f"bench.py --gpus {N}: rank 0's line is not a {N}-rank result: n_gpus={d.get('n_gpus')}", file=sys.stderr)
        return 1
    sys.stdout.write(lines[0] + "\n")
    sys.stdout.flush()
    return 0


def cpu_baseline(bp, table, seconds: float) -> dict:
    """Time the CPU oracle (restated reference verifier loop, serial like the
    reference) on channels [0, nc) of the same workload, t = 1."""
    from oracle import bf_oracle as orc

    op = orc.params_from(bp)
    n_pairs = bp.n_pairs
    # The verifier as the reference's own toolchain builds it evaluates cosf / sinf (nvcc's headers bind the
    # unqualified cos(float) to the float overload: oracle/bf_oracle.c); that reading is ~2x faster on a CPU than
    # the double-then-round one and is the baseline reported; the other is timed beside it on a third of the sample.
    with orc.trig_reading(orc.FLOAT_LIBM):
        s, _ = orc.generate_checksum(op, table, 1, 1, 0, 16, 1)  # calibrate
        rate = 16 * n_pairs / max(s, 1e-9)
        nc = int(max(16, min(bp.NR_CHANNELS, rate * seconds / n_pairs)))
        s1, ck1 = orc.generate_checksum(op, table, 1, 1, 0, nc, 1)
    out = {
        "value": nc * n_pairs / s1 / 1e9,
        "unit": "Gcoeff/s",
        "cores": 1,
        "kind": "port",
        "sample": f"channels [0,{nc}) of {bp.NR_STATIONS}ant x {bp.NR_BEAMS}beam x {bp.NR_CHANNELS}chan, t=1, "
                  f"{nc * n_pairs / 1e6:.0f} Mcoeff in {s1:.2f} s, 1 thread (the reference verifier is serial), "
                  f"cos/sin as float libm calls (what the reference's nvcc build binds them to)",
    }
    nc_d = max(16, nc // 3)
    s_d, _ = orc.generate_checksum(op, table, 1, 1, 0, nc_d, 1)
    out["double_then_round_reading"] = {"value": nc_d * n_pairs / s_d / 1e9, "cores": 1,
                                        "sample": f"channels [0,{nc_d}) in {s_d:.2f} s, (float)cos((double)x)"}
    ncores = os.cpu_count() or 1
    if ncores > 1:
        nt_threads = min(ncores, 64)
        nc_mt = int(min(bp.NR_CHANNELS, max(nt_threads, nc * min(nt_threads, 8) // 3)))
        with orc.trig_reading(orc.FLOAT_LIBM):
            s2, _ = orc.generate_checksum(op, table, 1, 1, 0, nc_mt, nt_threads)
        out["all_cores"] = {"value": nc_mt * n_pairs / s2 / 1e9, "cores": nt_threads,
                            "sample": f"channels [0,{nc_mt}) in {s2:.2f} s"}
    # the verifier's beamformer with the coefficient of one time held for 256 samples (what
    # also_measured.beamform_accumulated runs on the GPU), serial, on a bounded sample of the 64 x 16 shape
    import time as _time

    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import simulate_input

    A, B, nt = 64, 16, 256
    C = max(1, int(min(8192, 6000 * seconds / 12.0)))  # ~1 s at the default --cpu-seconds
    fp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B, NR_SAMPLES_PER_CHANNEL=nt)
    fop = orc.params_from(fp)
    ant = orc.simulate_antenna_data(fop, nt)
    t0 = _time.perf_counter()
    with orc.trig_reading(orc.FLOAT_LIBM):
        orc.beamform_accumulated(fop, simulate_input(fp), np.float32(0.0008192), nt, ant)
    sb = _time.perf_counter() - t0
    out["beamform_accumulated"] = {"value": A * B * C * nt / sb / 1e12, "unit": "T coefficient-products/s", "cores": 1, "kind": "port",
                                   "sample": f"{A}ant x {B}beam x {C}chan x {nt}samples in {sb:.2f} s, 1 thread"}
    return out


def extras(gen, bp, out, out_bytes, sh, device, sustain_seconds: float, samples: list) -> dict:
    """Side measurements after the timed region (N = 1), same HIP-event method:
    * ``sustained``: >= 5 s of back-to-back steps of the headline workload at the geometry the headline ran with
      (the chip's write rate sags ~1-3 % over the first seconds of load; a real-time generator lives there);
    * ``streaming_cfg5``: BASELINE configs[4] -- hipGraph replay, one time step per tick, model time advancing
      200 us per tick (dcs_bf_stream_tick_dt): the full tensor's update period (it cannot meet 200 us: 16 GiB
      need >= 2.15 ms at the 8 TB/s peak) and the largest channel slab whose period stays <= 200 us; both also with a NEW
      DELAY TABLE ON EVERY TICK, from host memory (pinned ring + H2D copy) and from device memory (gather node in the graph);
    * ``fp16_output``: the b16 output mode, exact-RNE form and the opt-in b16 arithmetic form;
    * ``fused_generate_and_beamform`` (f1) on a 64 x 64 x 4096 x 64 problem;
    * ``beamform_accumulated``: the coefficient-reuse beamformer (256 samples per coefficient) at 16 and 256 beams, and the
      K-split form at 256 antennas.
    Nothing here touches oracle/: a few rows of each item's LAST output are copied to the host into ``samples`` and
    compared with the oracle later, in the cpu_baseline leg (``cpu_baseline.extras_vs_oracle``)."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import SteeringCoefficientGenerator, simulate_input

    def timed(fn, n=20, warm=5, settle_ms=30.0, timed_ms=12.0):
        """ms per call at steady state: at least `warm` calls AND ~settle_ms on this access pattern first (the first launches
        after a change of pattern run 3-10 % slow: DESIGN.md 5.1), then ONE event pair around max(n, ~timed_ms worth of) calls."""
        e0, e1 = device.Event(), device.Event()
        fn()
        e0.record(sh)
        fn()
        fn()
        e1.record(sh)
        e1.synchronize()
        one = max(e1.elapsed_ms_since(e0) / 2, 1e-3)
        for _ in range(int(min(4000, max(warm, settle_ms / one)))):
            fn()
        n = int(min(4000, max(n, timed_ms / one)))
        e0.record(sh)
        for _ in range(n):
            fn()
        e1.record(sh)
        e1.synchronize()
        return e1.elapsed_ms_since(e0) / n

    res = {}
    n_coeff = bp.coeffs_per_time_step()
    row_bytes = bp.n_pairs * 8

    def sample_rows(rows, row_b, dtype):
        host = np.empty((len(rows), bp.NR_STATIONS, bp.NR_BEAMS, 2), dtype=dtype)
        device.stream_synchronize(sh)
        for i, r in enumerate(rows):
            device.memcpy_dtoh(host[i], out.data_ptr() + r * row_b)
        return host

    # -- sustained: batches of 50 launches, the host one batch ahead of the device; per-second rates kept
    batch, k, total_ms, launches, per_sec = 50, 0, 0.0, 0, []
    ev = [device.Event().record(sh)]
    t_end = time.perf_counter() + sustain_seconds
    sec_ms, sec_n = 0.0, 0
    while True:
        for _ in range(batch):
            gen.generate(out.data_ptr(), out_bytes, t0=1 + (k % 255), nt=1, stream=sh)
            k += 1
        ev.append(device.Event().record(sh))
        if len(ev) >= 3:  # wait for the batch before the one just queued
            ev[-2].synchronize()
            ms = ev[-2].elapsed_ms_since(ev[-3])
            total_ms += ms
            launches += batch
            sec_ms += ms
            sec_n += batch
            if sec_ms >= 1000.0:
                per_sec.append(n_coeff * sec_n / sec_ms / 1e6)
                sec_ms, sec_n = 0.0, 0
        if time.perf_counter() >= t_end and launches > 0:
            break
    ev[-1].synchronize()
    total_ms += ev[-1].elapsed_ms_since(ev[-2])
    launches += batch
    res["sustained"] = {"value": n_coeff * launches / total_ms / 1e6, "unit": "Gcoeff/s", "seconds": total_ms / 1e3, "launches": launches,
                        "ms_per_step": total_ms / launches, "frac_of_hbm_peak": 8 * n_coeff * launches / total_ms / 1e6 / HBM_PEAK_GBPS,
                        "per_second": per_sec, "note": "back-to-back steps of the headline workload, same kernel and launch geometry"}

    # -- BASELINE configs[4]: streaming at a 200 us cadence
    tab = [np.ascontiguousarray(simulate_input(bp)), np.ascontiguousarray(simulate_input(bp))]
    tab[1]["fPhase_rad"] += np.float32(0.25)  # a second delay model, so that a table update is visible in the output
    d_tab = [device.mem_alloc(t.nbytes) for t in tab]
    for d, t in zip(d_tab, tab):
        device.memcpy_htod(d, t)

    def tick_period_us(nc, ticks=150, warm=15, table=None):
        """table: None = the table stays; "host" / "device" = a new table with EVERY tick, from host / device memory."""
        nbytes = nc * bp.n_pairs * 8
        gen.upload_delays(tab[0], stream=sh)  # every run starts from the headline's table
        st = gen.stream_begin(out.data_ptr(), nbytes, 0, nc, sh)

        def tick(i):
            if table == "host":
                st.tick_dt(i * 200e-6, tab[i % 2])
            elif table == "device":
                st.tick_dt_from_global(i * 200e-6, d_tab[i % 2])
            else:
                st.tick_dt(i * 200e-6)

        for i in range(warm):
            tick(i)
        device.stream_synchronize(sh)
        e0, e1 = device.Event().record(sh), device.Event()
        t0 = time.perf_counter()
        for i in range(ticks):
            tick(warm + i)
        e1.record(sh)
        e1.synchronize()
        wall = (time.perf_counter() - t0) / ticks * 1e6
        dev = e1.elapsed_ms_since(e0) / ticks * 1e3
        last = warm + ticks - 1
        rows = sorted({0, nc // 2, nc - 1})
        samples.append({"item": f"streaming_cfg5: {nc} channels, table {'unchanged' if table is None else 'new every tick from ' + table + ' memory'}",
                        "kind": "generate", "dt": float(np.float32(last * 200e-6)), "rows": rows,
                        "table": tab[last % 2] if table else None, "data": sample_rows(rows, row_bytes, np.float32)})
        st.end()
        return max(dev, wall), nbytes

    full_us, full_bytes = tick_period_us(bp.NR_CHANNELS, ticks=60, warm=6)
    full_host_us, _ = tick_period_us(bp.NR_CHANNELS, ticks=60, warm=6, table="host")
    full_dev_us, _ = tick_period_us(bp.NR_CHANNELS, ticks=60, warm=6, table="device")
    # slabs below 2 GiB run the generator's other variant (terms per workgroup): let the library measure ITS geometry too
    # (cached per variant; the full tensor's stays), as the headline does for the 16 GiB launch
    slab_tuning = gen.autotune(out.data_ptr(), min(out_bytes, 2560 * bp.n_pairs * 8), stream=sh)
    best = None
    for nc in (1536, 2048, 2304, 2432, 2560, 2688, 2816):
        if nc > bp.NR_CHANNELS:
            break
        us, nb = tick_period_us(nc)
        if us <= 200.0:
            best = {"channels": nc, "bytes_per_tick": nb, "period_us": us, "Mcoeff_per_tick": nc * bp.n_pairs / 1e6, "TBps": nb / us / 1e6}
    every = None
    if best is not None:  # the same slab with a new table on every tick
        h_us, _ = tick_period_us(best["channels"], table="host")
        d_us, _ = tick_period_us(best["channels"], table="device")
        every = {"channels": best["channels"], "host_table_period_us": h_us, "device_table_period_us": d_us,
                 "table_bytes": int(tab[0].nbytes)}
    res["streaming_cfg5"] = {"cadence_target_us": 200.0, "model_time_step_us": 200.0, "launch": "hipGraph replay, dcs_bf_stream_tick_dt",
                             "full_tensor_period_us": full_us, "full_tensor_TBps": full_bytes / full_us / 1e6,
                             "meets_200us_full_tensor": bool(full_us <= 200.0), "largest_slab_at_200us": best,
                             "slab_launch_geometry": {k: slab_tuning[k] for k in ("tiles_per_block", "chan_per_block", "wg_per_cu")},
                             "new_table_every_tick": {"full_tensor_host_table_period_us": full_host_us,
                                                      "full_tensor_device_table_period_us": full_dev_us, "slab_at_200us": every,
                                                      "note": "host: memcpy into a ring of 4 pinned buffers + H2D copy in front of the "
                                                              "replay (dcs_bf_stream_tick_dt); device: gather node inside the replayed "
                                                              "graph (dcs_bf_stream_tick_dt_from_global), no host staging"}}
    for d in d_tab:
        d.free()
    gen.upload_delays(tab[0], stream=sh)  # back to the headline's table

    # -- fp16 output, both arithmetic forms: at the library's default geometry, and -- like the headline -- at the geometry
    #    dcs_bf_autotune measures on this device (cached per form); `value` is the tuned rate, as for fp32
    nb16 = gen.output_bytes(0, 1)
    for key, mode in (("fp16_output", 0), ("fp16_output_b16_arithmetic", 4)):
        gen.set_tuning() if mode == 0 else gen.set_tuning(math_mode=mode)
        run16 = lambda: gen.generate(out.data_ptr(), nb16, t0=1, nt=1, bitwidth=0, stream=sh)  # noqa: E731
        ms_default = timed(run16, n=40, warm=20)
        tuned = gen.autotune(out.data_ptr(), nb16, bitwidth=0, stream=sh)
        ms = min(timed(run16, n=40, warm=20), ms_default)  # (the tuner keeps the default unless beaten by > 0.7 %)
        res[key] = {"value": n_coeff / ms / 1e6, "unit": "Gcoeff/s", "ms": ms, "hbm_GBps": nb16 / ms / 1e6,
                    "frac_of_hbm_peak": nb16 / ms / 1e6 / HBM_PEAK_GBPS, "math_mode": mode,
                    "launch_geometry": {k: tuned[k] for k in ("tiles_per_block", "chan_per_block", "wg_per_cu")},
                    "value_at_default_geometry": n_coeff / ms_default / 1e6,
                    "bound": "4 B written per coefficient; VALU issue under the 1400 W power cap (30 / 21 vector operations per coefficient: "
                             "profiles/r03_fp16.md)"}
        rows = sorted({0, bp.NR_CHANNELS // 2, bp.NR_CHANNELS - 1})
        samples.append({"item": key, "kind": "generate_f16", "t": 1, "rows": rows, "table": None,
                        "data": sample_rows(rows, row_bytes // 2, np.float16)})
    gen.set_tuning()

    def antenna_pattern(nbytes):
        """Device buffer of pseudo-random int8 samples: a 32 MiB seeded block repeated (D2D copies)."""
        blk = min(nbytes, 32 << 20)
        host = np.random.default_rng(0xA17).integers(-128, 128, size=blk, dtype=np.int8)
        d = device.mem_alloc(nbytes)
        device.memcpy_htod(d, host, stream=sh)
        off = blk
        while off < nbytes:
            n = min(off, nbytes - off)  # doubling
            device.memcpy_dtod(int(d) + off, d, n, sh)
            off += n
        device.stream_synchronize(sh)
        return d, host

    A, B, C, nt = 64, 64, 4096, 64
    fp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B, NR_SAMPLES_PER_CHANNEL=nt)
    g = SteeringCoefficientGenerator(fp)
    g.upload_delays(simulate_input(fp), stream=sh)
    ant_bytes, beam_bytes = A * C * nt * 2, B * C * nt * 8
    d_ant, ant_host = antenna_pattern(ant_bytes)
    d_beams = device.mem_alloc(beam_bytes)
    ms = timed(lambda: g.generate_and_beamform(d_ant, ant_bytes, d_beams, beam_bytes, 0, nt, stream=sh))
    res["fused_generate_and_beamform"] = {"value": A * B * C * nt / ms / 1e6, "unit": "G coefficient-products/s", "ms": ms,
                                          "shape": f"{A}ant x {B}beam x {C}chan x {nt}samples", "bound": "fp32 VALU (no coefficient reaches HBM)"}
    device.stream_synchronize(sh)
    nchk = 2
    got = np.empty((nchk, nt // 16, B, 16, 2), dtype=np.float32)
    device.memcpy_dtoh(got, d_beams)
    samples.append({"item": "fused_generate_and_beamform", "kind": "beamform", "shape": (A, B, C, nt), "nc": nchk,
                    "ant": ant_host[: nchk * nt * A * 2].copy(), "data": got})
    g.close()
    d_ant.free()
    d_beams.free()
    # -- the same beamformer with the coefficients of ONE time reused for 256 samples (ACCUMULATIONS_BEFORE_NEW_COEFFS,
    #    BeamformerParameters.h:17): exact fixed-point contraction on the int8 matrix pipe; roofline = HBM
    res["beamform_accumulated"] = []
    for (A, B, C, nt) in ((64, 16, 32768, 256), (64, 256, 4096, 256), (256, 64, 4096, 256)):
        fp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B, NR_SAMPLES_PER_CHANNEL=nt)
        g = SteeringCoefficientGenerator(fp)
        g.upload_delays(simulate_input(fp), stream=sh)
        ant_bytes, beam_bytes = A * C * nt * 2, B * C * nt * 8
        d_ant, ant_host = antenna_pattern(ant_bytes)
        d_beams = device.mem_alloc(beam_bytes)
        ms = timed(lambda: g.beamform_accumulated(d_ant, ant_bytes, d_beams, beam_bytes, nt, t_coeff=1, stream=sh), n=40, warm=20)
        res["beamform_accumulated"].append({
            "value": A * B * C * nt / ms / 1e9, "unit": "T coefficient-products/s", "ms": ms, "shape": f"{A}ant x {B}beam x {C}chan x {nt}samples",
            "algorithmic_bytes": ant_bytes + beam_bytes, "hbm_GBps": (ant_bytes + beam_bytes) / ms / 1e6,
            "frac_of_hbm_peak": (ant_bytes + beam_bytes) / ms / 1e6 / HBM_PEAK_GBPS,
            "bound": "HBM (2 B per antenna and sample in, 8 B per beam and sample out); v_mfma_i32_16x16x64_i8 on 24-bit fixed-point coefficients"})
        device.stream_synchronize(sh)
        nchk = 2
        got = np.empty((nchk, nt // 16, B, 16, 2), dtype=np.float32)
        device.memcpy_dtoh(got, d_beams)
        samples.append({"item": f"beamform_accumulated {A}x{B}x{C}x{nt}", "kind": "beamform_accumulated", "shape": (A, B, C, nt), "nc": nchk,
                        "ant": ant_host[: nchk * nt * A * 2].copy(), "data": got})
        g.close()
        d_ant.free()
        d_beams.free()
    return res


def check_extras_against_oracle(bp, table, samples: list) -> list:
    """The rows ``extras`` copied back, against the oracle (part of the cpu_baseline leg: the only place where bench.py
    touches oracle/).  fp32 generator rows: ULP distance (bar 1); fp16 rows: binary16 ulps from RN-even(oracle) (bar 1);
    beamformers: largest |difference| against the verifier's loop over the first channels, bar 2e-5 / 4e-5 x antennas
    (tests/test_gpu_parity.py, include/dcs_beamformer.h; the reference's own tolerance is 1e-1)."""
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.generator import delta_times, simulate_input
    from oracle import bf_oracle as orc

    op = orc.params_from(bp)
    out = []
    for s in samples:
        if s["kind"] in ("generate", "generate_f16"):
            tbl = table if s["table"] is None else s["table"]
            dt = [np.float32(s["dt"])] if "dt" in s else delta_times(bp, s["t"], 1)
            worst = 0
            for i, r in enumerate(s["rows"]):
                exp = orc.generate_dt(op, tbl, dt, r, 1)[0, 0]
                if s["kind"] == "generate":
                    worst = max(worst, orc.max_ulp(s["data"][i], exp, 1)[0])
                else:
                    have = s["data"][i].view(np.uint16).astype(np.int32)
                    want = exp.astype(np.float16).view(np.uint16).astype(np.int32)
                    o = lambda u: np.where(u & 0x8000, -(u & 0x7FFF), u & 0x7FFF)  # noqa: E731
                    worst = max(worst, int(np.abs(o(have) - o(want)).max()))
            out.append({"item": s["item"], "rows_checked": len(s["rows"]), "max_ulp": int(worst), "bar": 1, "ok": bool(worst <= 1),
                        "unit": "fp32 ULP" if s["kind"] == "generate" else "binary16 ulp of RN-even(oracle)"})
        else:
            A, B, C, nt = s["shape"]
            fp = BeamformerParameters(NR_CHANNELS=C, NR_STATIONS=A, NR_BEAMS=B, NR_SAMPLES_PER_CHANNEL=nt)
            fop = orc.params_from(fp)
            ftab = simulate_input(fp)
            ant = s["ant"].reshape(s["nc"], nt // 16, A, 16, 2)
            if s["kind"] == "beamform":
                exp = orc.beamform_slab(fop, ftab, nt, 0, s["nc"], ant)
            else:
                exp = orc.beamform_accumulated_slab(fop, ftab, delta_times(fp, 1, 1)[0], nt, 0, s["nc"], ant)
            err = float(np.abs(s["data"].astype(np.float64) - exp.astype(np.float64)).max())
            # per-sample kernel: the verifier's own summation order, <= 1 ULP per coefficient (tests: 2e-5 x antennas);
            # coefficient-reuse kernel: include/dcs_beamformer.h's worst-case bound, 4e-5 x antennas
            bar = (2e-5 if s["kind"] == "beamform" else 4e-5) * A + 1e-6
            out.append({"item": s["item"], "channels_checked": s["nc"], "max_abs_err": err, "bar": bar, "ok": bool(err <= bar),
                        "largest_expected_magnitude": float(np.abs(exp).max())})
    return out


def pmc_traffic(bytes_algo: int):
    """(HBM bytes per launch, source) from the committed rocprofv3 --pmc passes of this same workload
    (profiles/pmc_write_size.json: WRITE_SIZE and FETCH_SIZE in separate passes of this command), or (None, None).
    The counters cannot be read from inside an un-profiled run, so this is that committed measurement, labelled."""
    f = ROOT / "profiles" / "pmc_write_size.json"
    try:
        d = json.loads(f.read_text())
        if int(d.get("algorithmic_bytes_per_launch", -1)) == int(bytes_algo):
            src = (f"profiles/pmc_write_size.json: rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE (separate passes) of `bench.py` in {d.get('round', 'r01')}, "
                   f"kernel {d.get('kernel', '?')[:80]}, launch geometry {d.get('launch_geometry', 'library default')}; every store is a whole 128-B line, "
                   "so the bytes do not depend on the geometry")
            return float(d["hbm_bytes_per_launch"]), src
    except Exception:
        pass
    return None, None


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # no launcher: this process only starts the ranks and relays rank 0's line (no torch, no HIP call here)
        raise SystemExit(launch_ranks(args))
    # stdout carries exactly ONE line, the JSON: anything libraries print there on the way
    # (RCCL's version banner, for one) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # the launch contract (module docstring): --gpus N is N ranks, or no result at all
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU, N ranks for --gpus N")
    N = args.gpus

    # the host driver of this pool only supports dmabuf IPC; without this RCCL's cross-process buffer
    # registration fails (hipIpcGetMemHandle: invalid argument).  Already exported on the boxes; kept here
    # so that the N > 1 path does not depend on the caller's environment.  Must precede HIP initialisation.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch

    if args.shared_device:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():  # (counting devices does not initialise the GPU)
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank}, but this machine shows {torch.cuda.device_count()} HIP device(s) "
                         "(no CPU fallback; --shared-device is a rehearsal flag, never a result)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    use_dist = N > 1 or args.force_collective
    if use_dist:
        import torch.distributed as dist  # noqa: F811

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world <= 1 and "RANK" not in os.environ:  # --force-collective without a launcher
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ["RANK"], os.environ["WORLD_SIZE"] = "0", "1"
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")

    from dc_sand_amd import BeamformerParameters, device
    from dc_sand_amd.generator import SteeringCoefficientGenerator, simulate_input
    from dc_sand_amd.parameters import delay_vals_dtype

    device.set_device(local_rank)
    B_total = args.beams_per_gpu * N
    bp_global = BeamformerParameters(NR_CHANNELS=args.chan, NR_STATIONS=args.ant, NR_BEAMS=B_total)
    bp = bp_global.with_beams(args.beams_per_gpu)
    beam_off = rank * args.beams_per_gpu
    gen = SteeringCoefficientGenerator(bp)
    out_bytes = gen.output_bytes(1, 1)
    if args.output_alloc == "torch":
        out = torch.empty(out_bytes, dtype=torch.uint8, device="cuda")
    else:  # plain hipMalloc through the library's device helpers (the pycuda-shaped host of the reference allocates so)
        class _HipBuffer:
            def __init__(self, nbytes):
                self._mem = device.mem_alloc(nbytes)

            def data_ptr(self):
                return int(self._mem)

        out = _HipBuffer(out_bytes)
    # one explicit non-blocking stream for everything (not the legacy null stream, which synchronises
    # implicitly with every blocking stream a library may have created); it is also torch's current
    # stream, so the process group orders its collectives against it
    main_stream = torch.cuda.Stream()
    torch.cuda.set_stream(main_stream)
    sh = main_stream.cuda_stream  # hipStream_t the kernels are launched on

    # the global delay table: the reference's ramp recipe (simulate_input) over all
    # A x B_total pairs, resident in HBM; two buffers so the next step's broadcast
    # overlaps this step's generation
    table_host = simulate_input(bp_global) if rank == 0 else np.zeros(bp_global.n_pairs, dtype=delay_vals_dtype)
    tbl = [torch.from_numpy(table_host.view(np.uint8).copy()).cuda() for _ in range(2)]

    works = [None, None]

    def prefetch(k: int):
        """Broadcast step k's table into buffer k % 2, overlapping the generation that is launched next.
        The collective is issued from the main stream's position (the process group's own stream waits
        until the main stream has got here, i.e. until the previous reader of this buffer, the slice
        gather of step k - 2, has run) and is NOT waited for here: step k waits for it.  (A side stream
        with explicit ready / freed events does the same with more queue packets: 0.3 % slower.)"""
        if use_dist:
            works[k % 2] = dist.broadcast(tbl[k % 2], src=0, async_op=True)

    coeff_stream = None  # --streaming: one hipGraph over the whole tensor, replayed per step

    def step(k: int):
        b = k % 2
        if use_dist:
            works[b].wait()  # the main stream waits for the broadcast; the host does not block
        if coeff_stream is not None:
            # BASELINE configs[3] + configs[4]: the broadcast's device buffer feeds the replayed graph directly (gather node,
            # then pre-pass + generator); model time advances 200 us per step
            coeff_stream.tick_dt_from_global(k * 200e-6, tbl[b].data_ptr(), B_total, beam_off)
            prefetch(k + 1)
            return
        gen.set_delays_from_global(tbl[b].data_ptr(), B_total, beam_off, stream=sh)
        prefetch(k + 1)
        gen.generate(out.data_ptr(), out_bytes, t0=1 + (k % 255), nt=1, stream=sh)

    if use_dist:
        # untimed set-up: the first barrier / all-reduce / broadcast of a process group sets up its
        # channels (tens of ms with the GPU idle); do that first, not between warm-up and timing
        dist.barrier()
        dist.all_reduce(torch.zeros(1, dtype=torch.float64, device="cuda"), op=dist.ReduceOp.MAX)
        dist.broadcast(tbl[1], src=0)
        torch.cuda.synchronize()

    # bring the device out of idle (clock ramp) with plain fills of the output buffer, so
    # that the W warm-up steps and the timed steps all run the kernel at steady state
    for _ in range(24):
        device.memset(out.data_ptr(), 0, out_bytes, stream=sh)
    torch.cuda.synchronize()

    # untimed set-up: let the library measure its launch geometries on this device for
    # this shape (dcs_bf_autotune; every geometry gives the same bits)
    tuning = None
    gen.upload_delays(np.ascontiguousarray(simulate_input(bp)), stream=sh)
    if not args.no_autotune:
        tuning = gen.autotune(out.data_ptr(), out_bytes, stream=sh)
    if use_dist:  # the ranks' tuners take different times: line the ranks up before the last, busy, set-up stage
        torch.cuda.synchronize()
        dist.barrier()
    # ... and settle on the geometry in use: after idle or a change of access pattern the first
    # ~20 ms of launches run 3-10 % (at worst 30 %) slower, whatever W the caller asked for
    one = device.Event().record(sh)
    gen.generate(out.data_ptr(), out_bytes, t0=1, nt=1, stream=sh)
    two = device.Event().record(sh)
    two.synchronize()
    for _ in range(max(4, min(400, int(40.0 / max(two.elapsed_ms_since(one), 1e-3))))):
        gen.generate(out.data_ptr(), out_bytes, t0=1, nt=1, stream=sh)

    if args.streaming:
        coeff_stream = gen.stream_begin(out.data_ptr(), out_bytes, 0, args.chan, sh)
    prefetch(0)
    for k in range(args.warmup):
        step(k)
    torch.cuda.synchronize()
    t_bar = time.perf_counter()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    if use_dist and rank == 0:
        print(f"barrier before the timed region: {(time.perf_counter() - t_bar) * 1e3:.3f} ms", file=sys.stderr)

    e0, e1 = device.Event(), device.Event()
    per_step_events = args.per_step_events or bool(os.environ.get("DCS_BENCH_DUMP_STEPS"))
    marks = [device.Event() for _ in range(args.steps)] if per_step_events else []  # the distribution, not only the mean
    t_start = time.perf_counter()
    e0.record(sh)
    for i, k in enumerate(range(args.warmup, args.warmup + args.steps)):
        step(k)
        if per_step_events:
            marks[i].record(sh)
    e1.record(sh)
    t_issued = time.perf_counter()
    torch.cuda.synchronize()
    if rank == 0 and os.environ.get("DCS_BENCH_DUMP_STEPS"):
        print(f"host issued the {args.steps} timed steps in {(t_issued - t_start) * 1e3:.2f} ms "
              f"(GPU finished {(time.perf_counter() - t_start) * 1e3:.2f} ms after the start)", file=sys.stderr)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    elapsed_local = elapsed
    ev_ms = e1.elapsed_ms_since(e0)

    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    coeffs_per_gpu_step = bp.coeffs_per_time_step()
    total_coeffs = coeffs_per_gpu_step * N * args.steps
    value = total_coeffs / elapsed / 1e9

    # every rank's own numbers (each rank tunes and times independently): a scaling efficiency below 1 can then be
    # attributed to the slowest GPU, to the collective or to launch skew
    geom_keys = ("tiles_per_block", "chan_per_block", "nontemporal", "wg_per_cu")
    mine = {"rank": rank, "kernel_ms": ev_ms / args.steps, "wall_ms_per_step": elapsed_local / args.steps * 1e3,
            "geometry": ({k: tuning[k] for k in geom_keys} if tuning else "library defaults")}
    per_rank = [mine]
    comm_world = 1
    if use_dist:
        comm_world = dist.get_world_size()
        gathered = [None] * comm_world
        dist.all_gather_object(gathered, mine)
        per_rank = gathered

    def spot_check():
        """First channels of the last generated step of THIS rank's slab vs the oracle
        (outside the timed region; part of the cpu_baseline leg / the rehearsal flag)."""
        from oracle import bf_oracle as orc

        k_last = args.warmup + args.steps - 1
        from dc_sand_amd.generator import delta_times

        dt_last = [np.float32(k_last * 200e-6)] if args.streaming else delta_times(bp, 1 + (k_last % 255), 1)
        nchk = min(4, args.chan)
        host = np.empty((nchk, args.ant, args.beams_per_gpu, 2), dtype=np.float32)
        device.memcpy_dtoh(host, out.data_ptr(), nbytes=host.nbytes)
        full = simulate_input(bp_global).reshape(args.ant, B_total)
        local = np.ascontiguousarray(full[:, beam_off:beam_off + args.beams_per_gpu]).ravel()
        exp = orc.generate_dt(orc.params_from(bp), local, dt_last, 0, nchk)
        mx, n_over, _ = orc.max_ulp(host, exp, 1)
        with orc.trig_reading(orc.FLOAT_LIBM):  # the verifier's other reading (oracle/bf_oracle.c)
            mx_f, _, _ = orc.max_ulp(host, orc.generate_dt(orc.params_from(bp), local, dt_last, 0, nchk), 1)
        return int(mx), int(n_over), int(mx_f)

    if args.check_all_ranks:
        mx, n_over, _ = spot_check()
        assert n_over == 0, f"rank {rank}: {n_over} elements over 1 ULP (max {mx})"

    result = None
    if rank == 0:
        # dominant kernel: the tiled generator; its duration = the HIP-event span on
        # its own stream over the timed region / launches (one launch per step; the
        # 16-KiB-per-row slice gather is the only other kernel there)
        kern_ms = ev_ms / args.steps
        per_step = np.diff([0.0] + [m.elapsed_ms_since(e0) for m in marks]) if per_step_events else np.array([kern_ms])
        if os.environ.get("DCS_BENCH_DUMP_STEPS"):
            print("per-step ms:", " ".join(f"{x:.3f}" for x in per_step), file=sys.stderr)
        algo_bytes = 8 * coeffs_per_gpu_step
        achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
        traffic, traffic_source = pmc_traffic(algo_bytes)
        result = {
            "metric": f"Gcoeff/s (complex weights) {args.ant}ant x {args.beams_per_gpu}beam x {args.chan}chan per GPU",
            "value": value,
            "unit": "Gcoeff/s",
            "n_gpus": N,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.config_name}: {args.ant}ant x {B_total}beam x {args.chan}chan, 1 time step per step, "
                            f"beam-sharded {args.beams_per_gpu} beams/GPU" + (f", {args.backend} bcast of the delay table each step" if use_dist else ""),
                "named_config": CONFIGS[args.config]["name"] + (" -- shape overridden on the command line" if "overridden" in args.config_name else ""),
                "coeffs_per_step": coeffs_per_gpu_step * N,
                "output_bytes_per_gpu_step": out_bytes,
                "kernel": ("MULTIPLE_CHANNELS_AND_TIMESTAMPS (tiled form; launches of >= 2 GiB read the pairs' terms from a pre-pass table)"
                           + ("; each step a hipGraph replay with the table gathered in-graph from the broadcast's device buffer "
                              "(dcs_bf_stream_tick_dt_from_global)" if args.streaming else "")),
                "launch_geometry": ({k: tuning[k] for k in ("tiles_per_block", "chan_per_block", "nontemporal", "wg_per_cu")} if tuning
                                    else "library defaults"),
                "collective": ("none" if not use_dist else ("RCCL broadcast" if args.backend == "nccl" else "gloo broadcast (REHEARSAL, not a result)")),
            },
            "per_rank": per_rank,
            "rccl_world_size": (comm_world if (use_dist and args.backend == "nccl") else None),
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel_ms": kern_ms,
                **({"kernel_ms_median": float(np.median(per_step)), "kernel_ms_min": float(np.min(per_step)),
                    "kernel_ms_max": float(np.max(per_step))} if per_step_events else {}),
                "algorithmic_bytes_per_launch": algo_bytes,
            },
        }
        check = None
        if N == 1 and not args.no_cpu_baseline:
            check = spot_check()  # before anything else rewrites the output buffer
        samples = []
        if N == 1 and not args.no_extras:
            # the side measurements must not be able to take the headline line down with them: a failure there is REPORTED
            # (also_measured.error / extras_vs_oracle_all_ok), the line is still printed
            try:
                result["also_measured"] = extras(gen, bp, out, out_bytes, sh, device, args.sustain_seconds, samples)
            except Exception as e:  # noqa: BLE001
                import traceback

                traceback.print_exc()
                result["also_measured"] = {"error": f"{type(e).__name__}: {e}"}
        if N == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(bp, np.ascontiguousarray(table_host), args.cpu_seconds)
            if samples:
                try:
                    chk = check_extras_against_oracle(bp, np.ascontiguousarray(table_host), samples)
                except Exception as e:  # noqa: BLE001
                    chk = [{"item": "check_extras_against_oracle", "ok": False, "error": f"{type(e).__name__}: {e}"}]
                result["cpu_baseline"]["extras_vs_oracle"] = chk
                result["cpu_baseline"]["extras_vs_oracle_all_ok"] = bool(all(c["ok"] for c in chk))
                for c in chk:
                    if not c["ok"]:
                        print(f"bench.py: side measurement differs from the oracle: {c}", file=sys.stderr)
            result["cpu_baseline"]["gpu_vs_oracle_spot_check"] = {"max_ulp": check[0], "over_1ulp": check[1],
                                                                  "max_ulp_float_libm_reading": check[2],
                                                                  "sample": "first 4 channels of the last timed step"}
            assert check[1] == 0, "GPU output of the timed region differs from the oracle by more than 1 ULP"
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(result) + "\n").encode())

    if coeff_stream is not None:
        coeff_stream.end()
    gen.close()
    if use_dist:
        for w in works:  # the last prefetched broadcast has no consumer
            if w is not None:
                w.wait()
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
