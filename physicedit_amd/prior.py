"""Training-time prior of PhysicEdit after DINOv2 / the VAE have run: host mirror of `QwenImageUnit_PhysicalVisualEmbedder.process`
(pipelines/qwen_image_physical.py:1071-1118) with the reference's modules `PerceiverResampler` (pipelines/helpers.py:66-109),
`PerceiverAttention` (:21-64), `FeedForward` (:8-19) and `VisualThinkingAdapter` (:111-120) -- SURVEY.md section 8 row f2.

Forward only (the library has no backward): it produces `pseudo_special_emb_dino` / `pseudo_special_emb_vae`, the targets that
`model_fn_qwen_image(is_train=True, ...)` turns into the special-token loss.  Every Linear is a `pe_gemm_bf16` launch (residuals and
the exact-erf GELU ride in its epilogue), the LayerNorms and the attention core are the two small kernels `pe_layernorm_affine` /
`pe_perceiver_attention`, element-wise sums are `pe_add_bf16`; what happens on the host is data movement (slices, concatenation,
broadcast of the frame embedding).  State-dict keys are the reference's (`dino_resampler.*`, `dino_time_embed.weight`,
`dino_resampler_adapter.net.*`, and the `vae_` twins)."""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from . import _lib, ops

BF = torch.bfloat16


class PerceiverResampler:
    def __init__(self, sd: Dict[str, torch.Tensor], device="cuda", heads: int = 8):
        self.device = torch.device(device)
        self.p = {k: v.to(device=self.device, dtype=BF).contiguous() for k, v in sd.items()}
        self.dim = self.p["latents"].shape[1]
        self.heads = heads
        self.depth = 1 + max(int(k.split(".")[1]) for k in self.p if k.startswith("layers."))
        self.ones = torch.ones((self.dim,), dtype=BF, device=self.device)        # gate of the residual epilogue: res + 1 * y

    def forward(self, media: torch.Tensor) -> torch.Tensor:
        """media [n, dim] -> [num_latents, dim]"""
        p = self.p
        n = media.shape[0]
        if n > p["pos_emb.weight"].shape[0]:
            raise _lib.PeError(f"PerceiverResampler: {n} media tokens, pos_emb holds {p['pos_emb.weight'].shape[0]}")
        x = ops.add_(media.to(device=self.device, dtype=BF).contiguous().clone(), p["pos_emb.weight"][:n])
        latents = p["latents"].clone()
        for i in range(self.depth):
            a, f = f"layers.{i}.0.", f"layers.{i}.1.net."
            xm = ops.layernorm_affine(x, p[a + "norm_media.weight"], p[a + "norm_media.bias"])
            lt = ops.layernorm_affine(latents, p[a + "norm_latents.weight"], p[a + "norm_latents.bias"])
            q = ops.gemm(lt, p[a + "to_q.weight"])
            kv = ops.gemm(torch.cat((xm, lt), dim=0), p[a + "to_kv.weight"])
            o = ops.perceiver_attention(q, kv, self.heads)
            latents = ops.gemm(o, p[a + "to_out.weight"], None, "gate_res", gate=self.ones, res=latents)
            h = ops.layernorm_affine(latents, p[f + "0.weight"], p[f + "0.bias"])
            h = ops.gemm(h, p[f + "1.weight"], p[f + "1.bias"], "gelu_erf")
            latents = ops.gemm(h, p[f + "3.weight"], p[f + "3.bias"], "gate_res", gate=self.ones, res=latents)
        return ops.layernorm_affine(latents, p["norm.weight"], p["norm.bias"])


class VisualThinkingAdapter:
    """helpers.py:111-120: Linear(in, 3 * out) -> GELU -> Linear(3 * out, out)"""

    def __init__(self, sd: Dict[str, torch.Tensor], device="cuda"):
        self.p = {k: v.to(device=device, dtype=BF).contiguous() for k, v in sd.items()}

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        h = ops.gemm(x, self.p["net.0.weight"], self.p["net.0.bias"], "gelu_erf")
        return ops.gemm(h, self.p["net.2.weight"], self.p["net.2.bias"])


def _sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


class PhysicalVisualPrior:
    """The part of QwenImageUnit_PhysicalVisualEmbedder.process that follows the two encoders (:1071-1118)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda"):
        self.device = torch.device(device)
        self.dino_resampler = PerceiverResampler(_sub(state_dict, "dino_resampler."), device)
        self.vae_resampler = PerceiverResampler(_sub(state_dict, "vae_resampler."), device)
        self.dino_adapter = VisualThinkingAdapter(_sub(state_dict, "dino_resampler_adapter."), device)
        self.vae_adapter = VisualThinkingAdapter(_sub(state_dict, "vae_resampler_adapter."), device)
        self.dino_time = state_dict["dino_time_embed.weight"].to(device=self.device, dtype=BF).contiguous()
        self.vae_time = state_dict["vae_time_embed.weight"].to(device=self.device, dtype=BF).contiguous()

    def _frames(self, feats: torch.Tensor, time_weight: torch.Tensor) -> torch.Tensor:
        """[B, L, H] + time_embed(arange(B))[:, None] -> [(B L), H]   (:1072-1074, :1100-1103)"""
        B, L, H = feats.shape
        if B > time_weight.shape[0]:
            raise _lib.PeError(f"{B} key frames, the frame embedding has {time_weight.shape[0]} rows")
        x = feats.to(device=self.device, dtype=BF).contiguous().clone()
        ops.add_(x, time_weight[:B, None, :].expand(B, L, H).contiguous())
        return x.reshape(B * L, H)

    def __call__(self, dino_middle: torch.Tensor, dino_source: torch.Tensor, latents_middle: torch.Tensor,
                 latents_source: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """DINOv2 patch features of the key frames [B, L, 768] / of the source image [1, L, 768], VAE latents [B, 16, h, w] /
        [1, 16, h, w] -> (pseudo_special_emb_dino, pseudo_special_emb_vae), each [1, 64, 3584]."""
        d_mid = self.dino_adapter.forward(self.dino_resampler.forward(self._frames(dino_middle, self.dino_time)))
        src = dino_source.to(device=self.device, dtype=BF).reshape(-1, dino_source.shape[-1]).contiguous()
        d_src = self.dino_adapter.forward(self.dino_resampler.forward(src))
        pat = lambda z: torch.stack([ops.patchify(f.to(device=self.device, dtype=BF).contiguous()) for f in z])     # [B, hw/4, 64]
        v_mid = self.vae_adapter.forward(self.vae_resampler.forward(self._frames(pat(latents_middle), self.vae_time)))
        v_src = self.vae_adapter.forward(self.vae_resampler.forward(pat(latents_source).reshape(-1, 64)))
        return ops.add_(d_mid, d_src, -1.0).unsqueeze(0), ops.add_(v_mid, v_src, -1.0).unsqueeze(0)
