"""TEST INFRASTRUCTURE ONLY -- what stands in for a captured hipGraph when the kernels run on the lane emulator.

`ShardedTrainStep(graph_factory=EmuGraph)`: a "capture" records the host function that issues the launches, a replay runs
it again -- through the emulator library's `hipGraphLaunch` shim when the native step driver launches it (csrc/step_driver.hip,
compiled unchanged), so the driver's program walk, its communicators and the order of their collectives are the code that
runs.  Like a real graph it keeps reading the tensors it was recorded over (the slot's static buffers): state that does not
travel through them shows up as a wrong result, which is the point."""
import ctypes as C

from torcheasyrec_amd import _lib

_CB = C.CFUNCTYPE(None, C.c_void_p)


class EmuGraph:
    def __init__(self, body):
        self._body, self.error, self.launches = body, None, 0
        self._cb = _CB(self._run)  # (kept alive with the object: the C side holds the raw pointer)
        L = _lib.lib()
        L.tzr_emu_graph_create.restype = C.c_int
        L.tzr_emu_graph_create.argtypes = [_CB, C.c_void_p, C.POINTER(C.c_void_p)]
        h = C.c_void_p()
        assert L.tzr_emu_graph_create(self._cb, None, C.byref(h)) == 0
        self._h = h

    def _run(self, _arg):
        try:  # (an exception cannot cross the C frames of the driver: kept, re-raised by `check`)
            self.launches += 1
            self._body()
        except BaseException as e:  # noqa: BLE001
            self.error = e

    def check(self):
        if self.error is not None:
            e, self.error = self.error, None
            raise e

    def replay(self):
        self._run(None)
        self.check()

    def raw_cuda_graph_exec(self) -> int:
        return self._h.value
