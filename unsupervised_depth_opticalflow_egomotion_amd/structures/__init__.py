"""Mirror of core/networks/structures/__init__.py:1-12 (the names the reference star-imports)."""
from .net_utils import conv, deconv, warp_flow
from .inverse_warp import (inverse_warp2, calculate_rigid_flow, compute_essential_matrix,
                           compute_projection_matrix, pose_vec2mat, euler2mat)
