from unsupervised_depth_opticalflow_egomotion_amd.structures import *  # noqa: F401,F403
from unsupervised_depth_opticalflow_egomotion_amd.networks import (  # noqa: F401
    Depth_Model, PoseCNN, FeaturePyramid, PWC_tf)
