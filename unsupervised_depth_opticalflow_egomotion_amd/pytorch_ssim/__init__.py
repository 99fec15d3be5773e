"""Reference-signature shim for core/networks/pytorch_ssim/ssim.py:4-19 (HIP k_ssim_fwd/bwd)."""
from .. import ops


def SSIM(x, y):
    """Per-channel SSIM map with a 3x3 zero-padded box window (divisor always 9)."""
    return ops.SSIMFn.apply(x, y)
