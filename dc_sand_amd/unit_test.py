"""Python mirror of ``common/UnitTest.{hpp,cpp}``: the five-phase, event-timed
test harness every dc_sand kernel experiment derives from.

Same contract as the reference: a non-virtual :meth:`run_test` calls the five
overridable phases in fixed order (``UnitTest.cpp:28-59``), timing the three
device phases with events; :meth:`get_result` is tri-state (1 pass / -1 fail /
0 not run, with a stderr warning, ``UnitTest.cpp:61-74``); :meth:`get_time`
prints the timings, names the limiting phase and returns HtoD+kernel+DtoH in
ms (``UnitTest.cpp:77-112``).
"""
from __future__ import annotations

import sys

from .device import Event


class UnitTest:
    def __init__(self, stream=None):
        # UnitTest.cpp:7-15 -- result starts at 0 ("not run"), six events
        self.m_iResult = 0
        self.m_fKernelElapsedTime_ms = 0.0
        self.m_fHtoDElapsedTime_ms = 0.0
        self.m_fDtoHElapsedTime_ms = 0.0
        self._stream = stream
        self._ev = [Event() for _ in range(6)]

    # -- the five phases (pure virtual in the reference, UnitTest.hpp:33-45) --
    def simulate_input(self) -> None:
        raise NotImplementedError

    def transfer_HtoD(self) -> None:
        raise NotImplementedError

    def run_kernel(self) -> None:
        raise NotImplementedError

    def transfer_DtoH(self) -> None:
        raise NotImplementedError

    def verify_output(self) -> None:
        raise NotImplementedError

    # -- UnitTest.cpp:28-59 ---------------------------------------------------
    def run_test(self) -> None:
        e = self._ev
        self.simulate_input()

        e[0].record(self._stream)
        self.transfer_HtoD()
        e[1].record(self._stream)
        e[1].synchronize()
        self.m_fHtoDElapsedTime_ms = e[1].elapsed_ms_since(e[0])

        e[2].record(self._stream)
        self.run_kernel()
        e[3].record(self._stream)
        e[3].synchronize()
        self.m_fKernelElapsedTime_ms = e[3].elapsed_ms_since(e[2])

        e[4].record(self._stream)
        self.transfer_DtoH()
        e[5].record(self._stream)
        e[5].synchronize()
        self.m_fDtoHElapsedTime_ms = e[5].elapsed_ms_since(e[4])

        self.verify_output()

    # -- UnitTest.cpp:61-74 ---------------------------------------------------
    def get_result(self) -> int:
        if not self.m_iResult:
            print("UnitTest hasn't been run yet!", file=sys.stderr)
        return self.m_iResult

    # -- UnitTest.cpp:77-112 --------------------------------------------------
    def get_time(self) -> float:
        h, k, d = self.m_fHtoDElapsedTime_ms, self.m_fKernelElapsedTime_ms, self.m_fDtoHElapsedTime_ms
        print(f"HtoD:\t\t{h:g} ms")
        print(f"Kernel:\t\t{k:g} ms")
        print(f"DtoH:\t\t{d:g} ms\n")
        if h > k and h > d:
            print("Host to device transfer is the limiting factor.")
        elif d > k and d > h:
            print("Device to host transfer is the limiting factor.")
        elif k > h and k > d:
            print("Kernel execution is the limiting factor.")
        denom = h if h > d else d
        ratio = k / denom if denom > 0 else float("inf")
        print(f"GPU Utilisation: {ratio * 100.0:g}%")
        return h + k + d
