from .utils import load_state_dict  # noqa: F401
from .model_manager import ModelManager  # noqa: F401
