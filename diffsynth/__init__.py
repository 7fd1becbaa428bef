"""Import-compatible façade of the parts of `diffsynth` (DiffSynth-Studio) that PhysicEdit's inference
scripts touch (scripts/inference/validate.py:17-18), backed by the MI355X-native hot path in
`physicedit_amd`.  Drop this directory in place of `DiffSynth-Studio/diffsynth` (validate.py puts
`<checkout>/DiffSynth-Studio` first on sys.path) or put the repo root ahead of it on PYTHONPATH.

Only the denoising hot path is reimplemented; the prompt prologue (Qwen2.5-VL text encoder, tokenizer /
processor, physical-reasoning generation) is third-party `transformers` code in the reference and is
plugged in through `pipe.prompt_encoder`."""
from .models.utils import load_state_dict  # noqa: F401
from .models.model_manager import ModelManager  # noqa: F401
from .pipelines.qwen_image_physical import QwenImagePhysicPipeline, ModelConfig  # noqa: F401
from .pipelines.qwen_image import QwenImagePipeline  # noqa: F401
from .schedulers.flow_match import FlowMatchScheduler  # noqa: F401
