"""`QwenImagePipeline`: the upstream Qwen-Image(-Edit) pipeline = QwenImagePhysicPipeline without the
visual-thinking adapter / special tokens (reference: pipelines/qwen_image.py)."""
from .qwen_image_physical import QwenImagePhysicPipeline, ModelConfig  # noqa: F401


class QwenImagePipeline(QwenImagePhysicPipeline):
    use_special_tokens = False
