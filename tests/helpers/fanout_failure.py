"""Run by tests/test_gpu_errors.py in a child process with DCS_LIB_PATH=probes/libdcs_probes.so: the -DDCS_PROBES build
of the product sources, whose ``dcs_probe_knobs.fail_at_step`` makes ONE launch of a per-time-step loop report
hipErrorLaunchFailure without being enqueued.  What must hold then (include/dcs_beamformer.h):

* outside a capture: the call returns that status, the side streams are joined (the caller's stream is ordered behind
  everything that WAS launched), and the caller's stream stays usable -- the next call's tensor equals the oracle's;
* inside a capture: the call returns the status, and the capture can still be ENDED (no unjoined fork); the graph it
  yields replays the launches that were made.
Prints "OK" on success.
"""
import ctypes
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from conftest import rand_table  # noqa: E402
from dc_sand_amd import BeamformerParameters, _lib, device  # noqa: E402
from dc_sand_amd.generator import SteeringCoefficientGenerator  # noqa: E402
from oracle import bf_oracle as oracle  # noqa: E402
from probes import dcs_probes  # noqa: E402

HIP_ERROR_LAUNCH_FAILURE = 719


def main():
    assert "libdcs_probes" in str(_lib.LIB_PATH), "run with DCS_LIB_PATH=probes/libdcs_probes.so"
    oracle.build()
    device.set_device(0)
    bp = BeamformerParameters(NR_CHANNELS=24, NR_STATIONS=5, NR_BEAMS=13)
    op = oracle.params_from(bp)
    table = rand_table(bp.n_pairs, seed=31)
    nt = 32
    g = SteeringCoefficientGenerator(bp)
    g.upload_delays(table)
    device.synchronize()
    nbytes = g.output_bytes(1, nt)
    buf = device.mem_alloc(nbytes)
    s = device.Stream()
    exp = oracle.generate(op, table, 3, nt)
    step = exp[0].nbytes

    for kernel in (1, 0):  # MULTIPLE_CHANNELS (fans out over side streams from 8 steps on), NAIVE (caller's stream)
        fail_at = 21
        # ---- outside a capture
        dcs_probes.set_knobs(g, fail_at_step=fail_at)
        device.memset(buf, 0xFF, nbytes, stream=s)
        try:
            g.generate(buf, nbytes, t0=3, nt=nt, kernel=kernel, stream=s)
            raise AssertionError("the injected failure was not reported")
        except _lib.DcsError as e:
            assert e.status == HIP_ERROR_LAUNCH_FAILURE, e.status
        # joined: a D2H copy on the CALLER's stream sees every step launched before the failure
        got = np.empty(exp.shape, dtype=np.float32)
        device.memcpy_dtoh(got, buf, stream=s)
        done = fail_at - 1
        assert oracle.max_ulp(got[:done], exp[:done], 1)[1] == 0, "steps launched before the failure are not complete behind the join"
        assert np.all(np.isnan(got[done:])), "steps at and after the failure must not have been launched"
        # the stream is usable: the same call without the fault gives the whole tensor
        dcs_probes.set_knobs(g)
        g.generate(buf, nbytes, t0=3, nt=nt, kernel=kernel, stream=s)
        device.memcpy_dtoh(got, buf, stream=s)
        assert oracle.max_ulp(got, exp, 1)[1] == 0

        # ---- inside a capture
        hip = ctypes.CDLL("libamdhip64.so")
        V = ctypes.c_void_p
        hip.hipStreamBeginCapture.argtypes = [V, ctypes.c_int]
        hip.hipStreamEndCapture.argtypes = [V, ctypes.POINTER(V)]
        hip.hipGraphInstantiate.argtypes = [ctypes.POINTER(V), V, V, V, ctypes.c_size_t]
        hip.hipGraphLaunch.argtypes = [V, V]
        hip.hipGraphExecDestroy.argtypes = [V]
        hip.hipGraphDestroy.argtypes = [V]
        dcs_probes.set_knobs(g, fail_at_step=fail_at)
        assert hip.hipStreamBeginCapture(V(s.handle), 0) == 0
        try:
            g.generate(buf, nbytes, t0=3, nt=nt, kernel=kernel, stream=s)
            raise AssertionError("the injected failure was not reported (capture)")
        except _lib.DcsError as e:
            assert e.status == HIP_ERROR_LAUNCH_FAILURE, e.status
        graph = V()
        rc = hip.hipStreamEndCapture(V(s.handle), ctypes.byref(graph))
        assert rc == 0 and graph.value, f"hipStreamEndCapture after the failed call: {rc} (an unjoined fork?)"
        ex = V()
        assert hip.hipGraphInstantiate(ctypes.byref(ex), graph, None, None, 0) == 0
        device.memset(buf, 0xFF, nbytes, stream=s)
        assert hip.hipGraphLaunch(ex, V(s.handle)) == 0
        device.memcpy_dtoh(got, buf, stream=s)
        assert oracle.max_ulp(got[:done], exp[:done], 1)[1] == 0
        assert np.all(np.isnan(got[done:]))
        hip.hipGraphExecDestroy(ex)
        hip.hipGraphDestroy(graph)
        dcs_probes.set_knobs(g)
    g.close()
    buf.free()
    print("OK")


if __name__ == "__main__":
    main()
