from .config_utils import generate_loss_weights_dict  # noqa: F401
