"""MI355X-native photometric-warping loss stack + model API of
jianfenglihg/Unsupervised_depth_OpticalFlow_egomotion."""
__all__ = ["get_model", "Model_geometry", "Model_depth", "Model_flow", "set_align_corners", "HW_QUEUES"]


def _hw_queue_limit():
    """The HIP runtime deals a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The training step runs
    three network branches on three streams; once a process group's streams exist as well, two normal-priority branches end up
    sharing one queue and run one after the other (23.2 ms per step instead of 19.0, profiles/r06_hw_queues.txt).  Eight
    queues keep them apart.  The runtime reads the variable when it initialises, so this only helps when the package is
    imported before the first HIP call; a value the user exported is left alone, and the value in effect is returned
    (models._side_streams picks its stream priorities from it)."""
    import os
    import sys
    pre = os.environ.get("GPU_MAX_HW_QUEUES")
    if pre is not None:
        return int(pre)
    t = sys.modules.get("torch")
    if t is not None and t.cuda.is_initialized():
        return 4
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
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
