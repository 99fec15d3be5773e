"""MI355X-native photometric-warping loss stack + model API of
jianfenglihg/Unsupervised_depth_OpticalFlow_egomotion."""
__all__ = ["get_model", "Model_geometry", "Model_depth", "Model_flow", "set_align_corners"]


def __getattr__(name):   # lazy: importing the package must not need torch.cuda or the built library
    if name in ("get_model", "Model_geometry", "Model_depth", "Model_flow"):
        from . import models
        return getattr(models, name)
    if name == "set_align_corners":
        from .ops import set_align_corners
        return set_align_corners
    raise AttributeError(name)
