"""Checkpoint readers with the reference's signatures (DiffSynth-Studio/diffsynth/models/utils.py:65-88)."""
import hashlib
import os

import torch


def load_state_dict(file_path, torch_dtype=None, device="cpu"):
    """`load_state_dict(file_path, torch_dtype=None, device="cpu") -> dict[str, Tensor]`; `.safetensors` or torch pickle."""
    if file_path.endswith(".safetensors"):
        from safetensors import safe_open
        out = {}
        with safe_open(file_path, framework="pt", device=str(device)) as f:
            for k in f.keys():
                t = f.get_tensor(k)
                out[k] = t.to(torch_dtype) if torch_dtype is not None else t
        return out
    sd = torch.load(file_path, map_location=device, weights_only=True)
    if torch_dtype is not None:
        sd = {k: (v.to(torch_dtype) if isinstance(v, torch.Tensor) else v) for k, v in sd.items()}
    return sd


def load_state_dict_from_folder(file_path, torch_dtype=None):
    sd = {}
    for name in sorted(os.listdir(file_path)):
        if name.rsplit(".", 1)[-1] in ("safetensors", "bin", "ckpt", "pth", "pt"):
            sd.update(load_state_dict(os.path.join(file_path, name), torch_dtype=torch_dtype))
    return sd


def hash_state_dict_keys(state_dict, with_shape=True) -> str:
    """The reference's model fingerprint (models/utils.py:148-182): every tensor contributes the strings "key:d0_d1_..."
    (when with_shape) and "key", nested dicts contribute "key|<their own string>"; the sorted list is joined with ","
    and md5-hashed.  `ModelManager` matches it against the detector table (configs/model_config.py:21-24)."""
    def keys_string(sd):
        items = []
        for key, value in sd.items():
            if not isinstance(key, str):
                continue
            if isinstance(value, torch.Tensor):
                if with_shape:
                    items.append(key + ":" + "_".join(str(d) for d in value.shape))
                items.append(key)
            elif isinstance(value, dict):
                items.append(key + "|" + keys_string(value))
        return ",".join(sorted(items))
    return hashlib.md5(keys_string(state_dict).encode("utf-8")).hexdigest()
