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


class LazyStateDict:
    """Read-only state-dict view over one or more `.safetensors` shards that reads a tensor only when it is asked for, and
    then straight onto `device` (safetensors memory-maps the file: with device="cuda" a tensor goes page cache -> HBM without
    a host copy of the 41 GB checkpoint ever existing).  Names and shapes come from the shard headers, so model detection
    (hash_state_dict_keys) costs no tensor I/O.  `ModelManager` hands these out; `QwenImageDiTEngine` ingests from them."""

    def __init__(self, paths, torch_dtype=None, device="cpu"):
        from safetensors import safe_open
        self.paths = sorted(paths) if isinstance(paths, (list, tuple)) else [paths]
        self.torch_dtype = torch_dtype
        self.device = str(device)
        self._where, self._shapes = {}, {}
        for p in self.paths:
            with safe_open(p, framework="pt", device="cpu") as f:
                for k in f.keys():
                    self._where[k] = p
                    self._shapes[k] = tuple(f.get_slice(k).get_shape())
        self._open = {}

    def to_device(self, device):
        device = str(device)
        if device != self.device:
            self._open.clear()
            self.device = device
        return self

    def _file(self, path):
        from safetensors import safe_open
        if path not in self._open:
            self._open[path] = safe_open(path, framework="pt", device=self.device)
        return self._open[path]

    def close(self):
        self._open.clear()

    def shape_of(self, key):
        return self._shapes[key]

    def __getitem__(self, key):
        t = self._file(self._where[key]).get_tensor(key)
        return t.to(self.torch_dtype) if self.torch_dtype is not None and t.dtype != self.torch_dtype else t

    def get(self, key, default=None):
        return self[key] if key in self._where else default

    def __contains__(self, key):
        return key in self._where

    def __iter__(self):
        return iter(self._where)

    def __len__(self):
        return len(self._where)

    def keys(self):
        return self._where.keys()

    def items(self):
        for k in self._where:
            yield k, self[k]

    def values(self):
        for k in self._where:
            yield self[k]

    def dtype_of_first(self):
        """dtype the tensors are handed out in (what the pipeline calls the checkpoint's stored dtype)."""
        return self[next(iter(self._where))].dtype


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
        if isinstance(sd, LazyStateDict):          # names + shapes from the shard headers: no tensor is read
            for key in sd.keys():
                if with_shape:
                    items.append(key + ":" + "_".join(str(d) for d in sd.shape_of(key)))
                items.append(key)
            return ",".join(sorted(items))
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
