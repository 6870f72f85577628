"""Error paths of the C-ABI on the GPU: a call that cannot be part of a stream capture says so
with a status BEFORE it enqueues anything -- the capture stays valid --, and a failure in the middle of a per-time-step
launch loop leaves the side streams joined and the caller's stream usable."""
import ctypes
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import rand_table

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _hip():
    hip = ctypes.CDLL("libamdhip64.so")
    V = ctypes.c_void_p
    hip.hipStreamBeginCapture.argtypes = [V, ctypes.c_int]
    hip.hipStreamEndCapture.argtypes = [V, ctypes.POINTER(V)]
    hip.hipGraphInstantiate.argtypes = [ctypes.POINTER(V), V, V, V, ctypes.c_size_t]
    hip.hipGraphLaunch.argtypes = [V, V]
    hip.hipGraphExecDestroy.argtypes = [V]
    hip.hipGraphDestroy.argtypes = [V]
    return hip, V


def test_non_capturable_calls_are_refused_up_front_and_the_capture_survives(gpu, oracle):
    """On a capturing stream: ``generate_dt`` of 300 time steps (their fDeltaTime values would be staged through pinned
    memory behind an event wait), the rows form with > 1 time step, a first-ever ``beamform_accumulated`` and
    ``generate_and_beamform`` (their terms table would be allocated), ``autotune`` and ``stream_begin`` all return
    DCS_ERR_UNSUPPORTED; capturable calls made in the SAME capture afterwards are recorded, the capture ends cleanly
    and its replay writes what the oracle says."""
    from dc_sand_amd import BeamformerParameters, _lib
    from dc_sand_amd.generator import SteeringCoefficientGenerator

    hip, V = _hip()
    bp = BeamformerParameters(NR_CHANNELS=24, NR_STATIONS=5, NR_BEAMS=13)
    op = oracle.params_from(bp)
    table = rand_table(bp.n_pairs, seed=31)
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(table)
    g_rows = SteeringCoefficientGenerator(bp)
    g_rows.set_tuning(form=2)
    g_rows.upload_delays(table)
    gpu.synchronize()
    nbytes = g.output_bytes(1, 300)
    buf = gpu.mem_alloc(nbytes)
    ant = gpu.mem_alloc(bp.NR_STATIONS * bp.NR_CHANNELS * 16 * 2)
    beams = gpu.mem_alloc(bp.NR_BEAMS * bp.NR_CHANNELS * 16 * 8)
    gpu.memset(ant, 1, ant.nbytes)
    s = gpu.Stream()
    dts = np.linspace(0.0, 0.3, 300).astype(np.float32)
    assert hip.hipStreamBeginCapture(V(s.handle), 0) == 0
    refused = [
        lambda: g.generate_dt(buf, nbytes, dts, stream=s),
        lambda: g.generate(buf, nbytes, t0=0, nt=300, stream=s),
        lambda: g_rows.generate(buf, nbytes, t0=0, nt=2, stream=s),
        lambda: g.beamform_accumulated(ant, ant.nbytes, beams, beams.nbytes, 16, t_coeff=1, stream=s),
        lambda: g.generate_and_beamform(ant, ant.nbytes, beams, beams.nbytes, 0, 16, stream=s),
        lambda: g.autotune(buf, nbytes, stream=s),
        lambda: g.stream_begin(buf, nbytes, 0, bp.NR_CHANNELS, s),
    ]
    for i, call in enumerate(refused):
        with pytest.raises(_lib.DcsError) as e:
            call()
        assert e.value.status == _lib.DCS_ERR_UNSUPPORTED, (i, e.value.status)
    # the capture is still alive: these ARE capturable (up to 256 time steps ride in the kernel arguments)
    g.generate(buf, nbytes, t0=7, nt=256, stream=s)
    graph = V()
    rc = hip.hipStreamEndCapture(V(s.handle), ctypes.byref(graph))
    assert rc == 0 and graph.value, rc
    ex = V()
    assert hip.hipGraphInstantiate(ctypes.byref(ex), graph, None, None, 0) == 0
    gpu.memset(buf, 0xFF, nbytes, stream=s)
    assert hip.hipGraphLaunch(ex, V(s.handle)) == 0
    s.synchronize()
    exp = oracle.generate(op, table, 7, 256)
    got = np.empty(exp.shape, dtype=np.float32)
    gpu.memcpy_dtoh(got, buf)
    assert oracle.max_ulp(got, exp, 1)[1] == 0
    hip.hipGraphExecDestroy(ex)
    hip.hipGraphDestroy(graph)
    # outside a capture every one of the refused calls works (and the beamformers become capturable after their first call)
    g.generate_dt(buf, nbytes, dts, stream=s)
    s.synchronize()
    gpu.memcpy_dtoh(got, buf, nbytes=got.nbytes)
    assert oracle.max_ulp(got, oracle.generate_dt(op, table, dts[:256]), 1)[1] == 0
    g.beamform_accumulated(ant, ant.nbytes, beams, beams.nbytes, 16, t_coeff=1, stream=s)
    s.synchronize()
    assert hip.hipStreamBeginCapture(V(s.handle), 0) == 0
    g.beamform_accumulated(ant, ant.nbytes, beams, beams.nbytes, 16, t_coeff=1, stream=s)
    assert hip.hipStreamEndCapture(V(s.handle), ctypes.byref(graph)) == 0 and graph.value
    hip.hipGraphDestroy(graph)
    g.close()
    g_rows.close()
    for b in (buf, ant, beams):
        b.free()


@pytest.mark.parametrize("kernel", [0, 1])
def test_time_index_out_of_range_mid_loop_launches_nothing(gpu, kernel):
    """A per-time-step loop whose LATER time indices overflow the verifier's nanosecond step (BCT.cu:299: the fp32
    product no longer fits a long) returns DCS_ERR_OUT_OF_RANGE before the first launch and before the fork: the
    output buffer is untouched and the stream carries on."""
    from dc_sand_amd import BeamformerParameters, _lib
    from dc_sand_amd.generator import SteeringCoefficientGenerator, delta_times

    bp = BeamformerParameters(NR_CHANNELS=8, NR_STATIONS=2, NR_BEAMS=3)
    # the largest valid time index: the step in ns is t * 1e-7f * 1e9f * 8192 (fp32) and must stay below 9.2e18
    t_edge = int(9.2e18 / (1e-7 * 1e9 * 8192))
    while True:
        try:
            delta_times(bp, t_edge, 1)
            break
        except _lib.DcsError:
            t_edge -= 1 << 20
    lo, hi = t_edge, t_edge + (1 << 22)
    while hi - lo > 1:  # first failing index
        mid = (lo + hi) // 2
        try:
            delta_times(bp, mid, 1)
            lo = mid
        except _lib.DcsError:
            hi = mid
    nt = 16
    t0 = hi - 5  # steps 0..4 are valid, 5.. are not
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(rand_table(bp.n_pairs))
    nbytes = g.output_bytes(1, nt)
    buf = gpu.mem_alloc(nbytes)
    s = gpu.Stream()
    gpu.memset(buf, 0xFF, nbytes, stream=s)
    with pytest.raises(_lib.DcsError) as e:
        g.generate(buf, nbytes, t0=t0, nt=nt, kernel=kernel, stream=s)
    assert e.value.status == _lib.DCS_ERR_OUT_OF_RANGE
    s.synchronize()
    host = np.empty(nbytes, dtype=np.uint8)
    gpu.memcpy_dtoh(host, buf)
    assert np.all(host == 0xFF), "a launch was made before the range check"
    g.generate(buf, nbytes, t0=0, nt=nt, kernel=kernel, stream=s)  # the stream is fine
    s.synchronize()
    g.close()
    buf.free()


def test_launch_failure_in_the_fan_out_loop_joins_the_side_streams(gpu):
    """The injected-failure rehearsal (tests/helpers/fanout_failure.py) in a child process that loads the -DDCS_PROBES
    build of the SAME sources behind the ordinary wrappers: status returned, side streams joined, capture endable,
    stream usable afterwards."""
    from probes import build as pb

    env = dict(os.environ, DCS_LIB_PATH=str(pb.build()))
    res = subprocess.run([sys.executable, str(ROOT / "tests" / "helpers" / "fanout_failure.py")], env=env, capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0 and res.stdout.strip().endswith("OK"), res.stdout[-2000:] + res.stderr[-4000:]
