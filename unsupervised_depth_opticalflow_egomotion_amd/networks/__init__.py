"""PyTorch-ROCm networks of the joint model (MIOpen convolutions; not hand-written HIP).

State-dict keys match the reference exactly (checkpoints, staged pre-training and the ``--fix_*``
substring matching of train.py:36-80 depend on them): ``depth_net.*``, ``pose_net.*``, ``fpyramid.*``,
``pwc_model.*``."""
from .depth_model import Depth_Model
from .pose_cnn import PoseCNN
from .feature_pyramid import FeaturePyramid
from .pwc_tf import PWC_tf
