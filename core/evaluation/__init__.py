from unsupervised_depth_opticalflow_egomotion_amd.evaluation import (  # noqa: F401
    eval_flow_avg, calculate_error_rate, eval_depth, compute_errors)
