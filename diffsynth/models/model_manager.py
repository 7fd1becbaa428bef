"""Minimal `ModelManager` (reference: models/model_manager.py:350-384): loads checkpoint files and files
them under the names the pipeline fetches.  Like the reference it recognises a model by the md5 of its sorted
key/shape list (models/utils.py:148-182 against the table configs/model_config.py:21-24); reduced-depth
checkpoints (tests, synthetic benchmarks), whose fingerprint is in no table, fall back to a characteristic key."""
from typing import Dict, List, Union

import torch

from .utils import LazyStateDict, hash_state_dict_keys, load_state_dict

# fingerprints of the official checkpoints' layouts (configs/model_config.py:21-24)
_HASHES = {
    "0319a1cb19835fb510907dd3367c95ff": "qwen_image_dit",
    "8004730443f55db63092006dd9f7110e": "qwen_image_text_encoder",
    "ed4ea5824d55ec3107b09815e318123a": "qwen_image_vae",
}

_SIGNATURES = (
    ("qwen_image_dit", "transformer_blocks.0.img_mod.1.weight"),
    ("qwen_image_vae", "decoder.up_blocks.0.resnets.0.conv1.weight"),
    ("qwen_image_blockwise_controlnet", "controlnet_blocks.0.x_rms.weight"),
    ("qwen_image_text_encoder", "model.language_model.layers.0.self_attn.q_proj.weight"),
    ("qwen_image_text_encoder", "model.layers.0.self_attn.q_proj.weight"),
)


def detect_model_name(state_dict: Dict[str, torch.Tensor]) -> str:
    name = _HASHES.get(hash_state_dict_keys(state_dict, with_shape=True))
    if name is not None:
        return name
    for name, key in _SIGNATURES:
        if key in state_dict:
            return name
    return "unknown"


class ModelManager:
    def __init__(self, torch_dtype=torch.bfloat16, device="cpu"):
        self.torch_dtype = torch_dtype
        self.device = device
        self.model: List[Dict[str, torch.Tensor]] = []
        self.model_name: List[str] = []
        self.model_path: List[Union[str, List[str]]] = []

    def load_model(self, file_path: Union[str, List[str]], device=None, torch_dtype=None):
        paths = file_path if isinstance(file_path, (list, tuple)) else [file_path]
        if all(str(p).endswith(".safetensors") for p in paths):
            # lazily: the consumer (DiT engine, VAE, text encoder) pulls each tensor once, straight onto its device
            sd = LazyStateDict(list(paths), torch_dtype=torch_dtype or self.torch_dtype, device="cpu")
        else:
            sd = {}
            for p in sorted(paths):
                sd.update(load_state_dict(p, torch_dtype=torch_dtype or self.torch_dtype, device="cpu"))
        self.model.append(sd)
        self.model_name.append(detect_model_name(sd))
        self.model_path.append(file_path)

    def load_models(self, file_path_list, **kw):
        for p in file_path_list:
            self.load_model(p, **kw)

    def fetch_model(self, model_name, file_path=None, require_model_path=False, index=None):
        """`index="all"` returns every loaded model of that name as a list (the reference fetches its block-wise ControlNets
        this way, qwen_image_physical.py:521); otherwise the first match or None."""
        found = [((sd, path) if require_model_path else sd) for sd, name, path in zip(self.model, self.model_name, self.model_path)
                 if name == model_name and (file_path is None or path == file_path)]
        if index == "all":
            return found
        return found[0] if found else None
