"""DINOv2 (with registers) feature extractor of the training-time prior on the library's kernels.

Host mirror of the reference's `Dinov2withNorm` (DiffSynth-Studio/diffsynth/pipelines/dinov2.py:8-31: transformers'
`Dinov2WithRegistersModel` with the final LayerNorm's affine removed, the CLS token and the 4 register tokens dropped from the output),
called by `QwenImageUnit_PhysicalVisualEmbedder.process` (pipelines/qwen_image_physical.py:1060-1090) on the key frames and on the
source image.  State-dict keys are transformers' (`embeddings.*`, `encoder.layer.{i}.*`), so a `Dinov2WithRegistersModel` checkpoint
loads as it is.

Every Linear is a `pe_gemm_bf16` launch -- the patch embedding (Conv2d k = stride = patch) as a GEMM over unfolded patches, q/k/v as
ONE GEMM over the stacked weights, LayerScale + residual (`x + lambda * y`) and the exact-erf GELU in the epilogues --, the LayerNorms are
`pe_layernorm_affine`, the self-attention (heads of 64) is `pe_sdpa_heads64` (scaled_dot_product_attention's numerics).  What
happens on the host is data movement (patch unfolding, token concatenation) and, once per input size, the bicubic resampling of
the position table (a parameter, `interpolate_pos_encoding`).  Forward only; bf16 roundings where eager PyTorch has them."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from . import _lib, ops

BF = torch.bfloat16


class Dinov2WithNorm:
    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda", patch_size: int = 14, num_heads: int = 12,
                 layer_norm_eps: float = 1e-6, normalize: bool = True):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.PeError("Dinov2WithNorm needs a HIP device: physicedit_amd has no CPU path")
        sd = {k[len("encoder_model."):] if k.startswith("encoder_model.") else k: v for k, v in state_dict.items()}
        p = self.p = {k: v.to(device=self.device, dtype=BF).contiguous() for k, v in sd.items() if v.is_floating_point()}
        self.patch, self.heads, self.eps = int(patch_size), int(num_heads), float(layer_norm_eps)
        need = ("embeddings.patch_embeddings.projection.weight", "embeddings.register_tokens", "embeddings.position_embeddings",
                "encoder.layer.0.mlp.fc1.weight", "encoder.layer.0.attention.attention.query.weight")
        missing = [k for k in need if k not in p]
        if missing:
            hint = " (a SwiGLU-MLP checkpoint: mlp.weights_in / weights_out are not supported)" if any("mlp.weights_in" in k for k in p) else ""
            raise _lib.PeError(f"Dinov2WithNorm: not a DINOv2-with-registers state dict with fc1 / fc2 MLPs, missing {missing}{hint}")
        w = p["embeddings.patch_embeddings.projection.weight"]                      # [hidden, C, patch, patch]
        self.hidden = w.shape[0]
        if self.hidden != self.heads * 64:
            raise _lib.PeError(f"Dinov2WithNorm: hidden {self.hidden} / {self.heads} heads: the attention kernel has heads of 64")
        k_raw = w.shape[1] * w.shape[2] * w.shape[3]
        self.k_raw, self.k_pad = k_raw, (k_raw + 63) // 64 * 64                     # GEMM K granule; zero columns add exact zeros
        self.w_patch = torch.zeros((self.hidden, self.k_pad), dtype=BF, device=self.device)
        self.w_patch[:, :k_raw] = w.reshape(self.hidden, k_raw)
        self.depth = 1 + max(int(k.split(".")[2]) for k in p if k.startswith("encoder.layer."))
        self.n_reg = p["embeddings.register_tokens"].shape[1]
        # q, k, v as one Linear: [3 * hidden, hidden] (query first: the kernel takes q and the [k | v] columns separately)
        self.qkv = []
        for i in range(self.depth):
            a = f"encoder.layer.{i}.attention.attention."
            self.qkv.append((torch.cat([p[a + "query.weight"], p[a + "key.weight"], p[a + "value.weight"]]).contiguous(),
                             torch.cat([p[a + "query.bias"], p[a + "key.bias"], p[a + "value.bias"]]).contiguous()))
        self.normalize = normalize
        self.ones = torch.ones((self.hidden,), dtype=BF, device=self.device)
        self.zeros = torch.zeros((self.hidden,), dtype=BF, device=self.device)
        self._pos_cache: Dict[tuple, torch.Tensor] = {}

    @classmethod
    def from_transformers(cls, model, device="cuda", normalize: bool = True) -> "Dinov2WithNorm":
        cfg = model.config
        return cls(model.state_dict(), device=device, patch_size=cfg.patch_size, num_heads=cfg.num_attention_heads,
                   layer_norm_eps=cfg.layer_norm_eps, normalize=normalize)

    def _pos(self, gh: int, gw: int) -> torch.Tensor:
        """interpolate_pos_encoding: the stored table when the grid matches, else its patch part resampled (bicubic, antialias, fp32)."""
        key = (gh, gw)
        if key not in self._pos_cache:
            pos = self.p["embeddings.position_embeddings"]                          # [1, 1 + n0, hidden]
            n0 = pos.shape[1] - 1
            if gh * gw == n0 and gh == gw:
                out = pos[0]
            else:
                s0 = int(n0 ** 0.5)
                grid = pos[:, 1:].reshape(1, s0, s0, self.hidden).permute(0, 3, 1, 2).to(torch.float32)
                grid = F.interpolate(grid, size=(gh, gw), mode="bicubic", align_corners=False, antialias=True).to(BF)
                out = torch.cat((pos[0, :1], grid.permute(0, 2, 3, 1).reshape(-1, self.hidden)), dim=0)
            self._pos_cache[key] = out.contiguous()
        return self._pos_cache[key]

    @torch.no_grad()
    def forward(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """pixel_values [B, 3, H, W] (ImageNet-normalised, any float dtype) -> patch features [B, (H/p)*(W/p), hidden] bf16
        (dinov2.py:27-31: last_hidden_state without the CLS and register tokens)."""
        p, P = self.p, self.patch
        x = pixel_values.to(device=self.device, dtype=BF)
        B, C, Hh, Ww = x.shape
        gh, gw = Hh // P, Ww // P
        n = gh * gw
        # Conv2d(k = stride = patch) == Linear over unfolded patches; column order (c, dy, dx) = the conv weight's
        cols = x[:, :, :gh * P, :gw * P].reshape(B, C, gh, P, gw, P).permute(0, 2, 4, 1, 3, 5).reshape(B * n, self.k_raw)
        a = torch.zeros((B * n, self.k_pad), dtype=BF, device=self.device)
        a[:, :self.k_raw] = cols
        emb = ops.gemm(a, self.w_patch, p["embeddings.patch_embeddings.projection.bias"]).reshape(B, n, self.hidden)
        pos = self._pos(gh, gw)
        outs = []
        T = 1 + self.n_reg + n
        for b in range(B):
            tok = torch.cat((p["embeddings.cls_token"][0], emb[b]), dim=0).contiguous()             # [1 + n, hidden]
            tok = ops.add_(tok, pos)                                                                # + position table (bf16 add)
            h = torch.cat((tok[:1], p["embeddings.register_tokens"][0], tok[1:]), dim=0).contiguous()   # registers behind CLS
            for i in range(self.depth):
                L = f"encoder.layer.{i}."
                y = ops.layernorm_affine(h, p[L + "norm1.weight"], p[L + "norm1.bias"], eps=self.eps)
                qkv = ops.gemm(y, *self.qkv[i])                                                     # [T, 3 * hidden]
                att = ops.sdpa_heads64(qkv[:, :self.hidden].contiguous(), qkv[:, self.hidden:].contiguous(), self.heads)
                h = ops.gemm(att, p[L + "attention.output.dense.weight"], p[L + "attention.output.dense.bias"], "gate_res",
                             gate=p[L + "layer_scale1.lambda1"], res=h)                             # h + lambda1 * dense(att)
                y = ops.layernorm_affine(h, p[L + "norm2.weight"], p[L + "norm2.bias"], eps=self.eps)
                y = ops.gemm(y, p[L + "mlp.fc1.weight"], p[L + "mlp.fc1.bias"], "gelu_erf")
                h = ops.gemm(y, p[L + "mlp.fc2.weight"], p[L + "mlp.fc2.bias"], "gate_res", gate=p[L + "layer_scale2.lambda1"], res=h)
            if self.normalize:
                h = ops.layernorm_affine(h, self.ones, self.zeros, eps=self.eps)                    # LayerNorm without affine
            else:
                h = ops.layernorm_affine(h, p["layernorm.weight"], p["layernorm.bias"], eps=self.eps)
            assert h.shape[0] == T
            outs.append(h[1 + self.n_reg:])
        return torch.stack(outs)

    __call__ = forward
