"""Profiling hooks: roctx ranges around every HIP launcher of libdfe_hip.so and the reference's wall-clock ``Profiler``
(core/visualize/profiler.py:6-32: ``report_process`` / ``report_all`` print the time since the last mark after a device
synchronisation; the reference's ``pdb.set_trace()`` at the end of ``report_all`` is not kept).

``enable()`` (train.py ``--profile``; ``DFE_PROFILE=1``) wraps each C-ABI entry point in a roctx range named after it, so a
``rocprofv3 --marker-trace --kernel-trace`` run attributes kernels to launchers; ``range(name)`` marks host-side sections
(forward / backward / optimiser).  Off by default: no wrapper, no overhead."""
import contextlib
import ctypes
import os
import time

_STATE = {"on": False, "roctx": None}


def _load_roctx():
    if _STATE["roctx"] is None:
        for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
            try:
                lib = ctypes.CDLL(name)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                lib.roctxRangePushA.restype = ctypes.c_int
                lib.roctxRangePop.restype = ctypes.c_int
                _STATE["roctx"] = lib
                break
            except OSError:
                continue
        else:
            _STATE["roctx"] = False          # no roctx library on this machine: ranges are no-ops, the Profiler still works
    return _STATE["roctx"]


def enabled():
    return _STATE["on"]


def enable():
    """Wrap every entry point of libdfe_hip.so in a roctx range (idempotent)."""
    if _STATE["on"]:
        return
    from . import _lib
    rx = _load_roctx()
    lib = _lib.get_lib()
    if rx:
        for name in _lib.header_symbols():
            fn = getattr(lib, name)
            if getattr(fn, "_dfe_wrapped", False) or name in ("dfe_abi_version", "dfe_error_string"):
                continue

            def wrapper(*a, _fn=fn, _tag=name.encode()):
                rx.roctxRangePushA(_tag)
                try:
                    return _fn(*a)
                finally:
                    rx.roctxRangePop()
            wrapper._dfe_wrapped = True
            setattr(lib, name, wrapper)
    _STATE["on"] = True


@contextlib.contextmanager
def range(name):
    """A named host-side section (a roctx range when profiling is on and the library exists; free otherwise)."""
    rx = _STATE["roctx"] if _STATE["on"] else None
    if rx:
        rx.roctxRangePushA(str(name).encode())
    try:
        yield
    finally:
        if rx:
            rx.roctxRangePop()


class Profiler(object):
    """core/visualize/profiler.py:6-32 with the same method names: wall time between marks, device synchronised at each."""

    def __init__(self, silent=False):
        import torch
        self.silent = silent
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.start = time.time()
        self.cache_time = self.start

    def reset(self, silent=None):
        self.__init__(silent=self.silent if silent is None else silent)

    def _now(self):
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        return time.time()

    def report_process(self, process_name):
        if self.silent:
            return None
        now = self._now()
        print("{0}\t: {1:.4f}".format(process_name, now - self.cache_time))
        self.cache_time = now
        return now

    def report_all(self, whole_process_name):
        if self.silent:
            return None
        now = self._now()
        print("{0}\t: {1:.4f}".format(whole_process_name, now - self.start))
        return now


if os.environ.get("DFE_PROFILE", "0") == "1":
    try:
        enable()
    except Exception:      # the library is not built yet: train.py / bench.py fail loudly on their own
        pass
