from unsupervised_depth_opticalflow_egomotion_amd.evaluation import (  # noqa: F401
    eval_flow_avg, calculate_error_rate, eval_depth, compute_errors, resize_flow_like_cv2)
from unsupervised_depth_opticalflow_egomotion_amd.kitti_io import load_gt_flow_kitti, load_gt_mask  # noqa: F401
