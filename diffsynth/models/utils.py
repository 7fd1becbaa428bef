"""Checkpoint readers with the reference's signatures (DiffSynth-Studio/diffsynth/models/utils.py:65-88)."""
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
