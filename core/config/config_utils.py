"""loss_pack key -> YAML weight (reference core/config/config_utils.py:3-22)."""

_KEYS = {
    "loss_flow_pixel": "w_flow_pixel", "loss_flow_ssim": "w_flow_ssim", "loss_flow_smooth": "w_flow_smooth",
    "loss_flow_consis": "w_flow_consis", "loss_depth_pixel": "w_depth_pixel", "loss_depth_ssim": "w_depth_ssim",
    "loss_depth_smooth": "w_depth_smooth", "loss_depth_consis": "w_depth_consis",
    "loss_depth_flow_consis": "w_depth_flow_consis", "loss_epipolar": "w_epipolar", "loss_triangle": "w_triangle",
    "loss_pnp": "w_pnp", "loss_eight_point": "w_8point",
}


def generate_loss_weights_dict(cfg):
    return {loss: getattr(cfg, attr) for loss, attr in _KEYS.items()}
