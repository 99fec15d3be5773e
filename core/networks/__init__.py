from unsupervised_depth_opticalflow_egomotion_amd.models import (  # noqa: F401
    get_model, Model_geometry, Model_depth, Model_flow)
