import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The parity suite compares masks for EQUALITY with goldens whose cos / sin came from the golden host's libm: its
    # synthetic poses are conditioned (synthetic.robust_pose) unless a test asks for raw ones (condition_pose=False:
    # tests/test_hip_loss_stack.py::test_unconditioned_poses_*).  The library default is raw poses.
    from unsupervised_depth_opticalflow_egomotion_amd import synthetic
    synthetic.CONDITION_POSE = True


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")
