from unsupervised_depth_opticalflow_egomotion_amd.pytorch_ssim import SSIM  # noqa: F401
