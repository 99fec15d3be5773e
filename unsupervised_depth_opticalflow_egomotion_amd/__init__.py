"""MI355X-native photometric-warping loss stack + model API of
jianfenglihg/Unsupervised_depth_OpticalFlow_egomotion."""
__all__ = ["get_model", "Model_geometry", "Model_depth", "Model_flow", "set_align_corners", "HW_QUEUES"]


def _kfd_is_open():
    """True once this process has opened /dev/kfd, i.e. once the HIP / HSA runtime has started -- ``torch.cuda.is_available()``
    does that without setting ``torch.cuda.is_initialized()``."""
    import os
    try:
        for fd in os.listdir("/proc/self/fd"):
            try:
                if os.readlink("/proc/self/fd/" + fd) == "/dev/kfd":
                    return True
            except OSError:
                pass
    except OSError:
        pass
    return False


def _initial_env(name):
    """The variable as the process was STARTED with (/proc/self/environ), whatever os.environ says now."""
    try:
        with open("/proc/self/environ", "rb") as f:
            for item in f.read().split(b"\0"):
                if item.startswith(name.encode() + b"="):
                    return item.split(b"=", 1)[1].decode()
    except OSError:
        pass
    return None


def _hw_queue_limit(started=None):
    """The HIP runtime deals a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The training step runs
    three network branches on three streams; once a process group's streams exist as well, two normal-priority branches end up
    sharing one queue and run one after the other (23.2 ms per step instead of 19.0, profiles/r06_hw_queues.txt).  Eight
    queues keep them apart.  The runtime reads the variable when it starts (its first call -- ``torch.cuda.is_available()`` is
    one), so: not started yet -> a value the user set is left alone, otherwise it becomes 8; already started -> nothing can be
    changed, and what the runtime saw is the value the process was started with (a later ``os.environ`` assignment is NOT
    believed: measured 23.4 ms).  Returns the value in effect; models._side_streams picks its stream priorities from it."""
    import os
    import sys
    name = "GPU_MAX_HW_QUEUES"
    if started is None:
        t = sys.modules.get("torch")
        started = (t is not None and t.cuda.is_initialized()) or _kfd_is_open()
    if started:
        seen = _initial_env(name)
        return int(seen) if seen else 4
    pre = os.environ.get(name)
    if pre is not None:
        return int(pre)
    os.environ[name] = "8"
    return 8


HW_QUEUES = _hw_queue_limit()


def __getattr__(name):   # lazy: importing the package must not need torch.cuda or the built library
    if name in ("get_model", "Model_geometry", "Model_depth", "Model_flow"):
        from . import models
        return getattr(models, name)
    if name == "set_align_corners":
        from .ops import set_align_corners
        return set_align_corners
    raise AttributeError(name)
