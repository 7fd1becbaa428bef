"""Block-wise ControlNet for the Qwen-Image DiT: host mirror of the reference's
`QwenImageBlockWiseControlNet` (models/qwen_image_controlnet.py:29-57), `QwenImageBlockwiseMultiControlNet`
(pipelines/qwen_image_physical.py:157-180) and `ControlNetInput` (pipelines/flux_image_new.py:6-13).

The weights stay torch tensors on the device; the per-block pointer table goes to `pe_dit_forward` inside the call struct
(`pe_control_input`), which applies every ACTIVE input after each transformer block.  Which inputs are active at a step is host
logic here, exactly the reference's progress gate.  No tensor math on the host: `preprocess` is `pe_patchify` + `pe_gemm_bf16`."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib, ops
from ._lib import ControlNetBlock

BF = torch.bfloat16
DIM = 3072


@dataclass
class ControlNetInput:
    """pipelines/flux_image_new.py:6-13 (same fields and defaults)."""
    controlnet_id: int = 0
    scale: float = 1.0
    start: float = 1.0
    end: float = 0.0
    image: object = None
    inpaint_mask: object = None
    processor_id: Optional[str] = None


class QwenImageBlockWiseControlNet:
    """state dict (reference key names: `img_in.*`, `controlnet_blocks.i.{x_rms,y_rms}.weight`, `controlnet_blocks.i.{input,output}_proj.*`)
    -> resident bf16 tensors + the pointer table of `pe_controlnet_block`s.  The inpaint variant (additional_in_dim = 4,
    detected from img_in's width like the reference's state-dict converter does by hash) takes 17-channel conditioning latents."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda"):
        self.device = torch.device(device)
        sd = {k: v.to(device=self.device, dtype=BF).contiguous() for k, v in state_dict.items()}
        w = sd["img_in.weight"]
        if w.shape[0] != DIM or w.shape[1] % 4 != 0:
            raise _lib.PeError(f"controlnet img_in.weight has shape {tuple(w.shape)}")
        self.in_dim = w.shape[1]                                   # 64, or 68 for the inpaint ControlNet
        kp = (self.in_dim + 63) // 64 * 64                         # GEMM K granularity: zero columns change nothing
        self.img_in_w = torch.zeros((DIM, kp), dtype=BF, device=self.device)
        self.img_in_w[:, :self.in_dim] = w
        self.img_in_b = sd["img_in.bias"]
        self.num_layers = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("controlnet_blocks."))
        self.params = sd
        self._blocks = (ControlNetBlock * self.num_layers)()
        for i in range(self.num_layers):
            p = f"controlnet_blocks.{i}."
            b = self._blocks[i]
            b.x_rms_w, b.y_rms_w = sd[p + "x_rms.weight"].data_ptr(), sd[p + "y_rms.weight"].data_ptr()
            b.in_w, b.in_b = sd[p + "input_proj.weight"].data_ptr(), sd[p + "input_proj.bias"].data_ptr()
            b.out_w, b.out_b = sd[p + "output_proj.weight"].data_ptr(), sd[p + "output_proj.bias"].data_ptr()

    @property
    def block_table(self):
        return self._blocks

    def process_controlnet_conditioning(self, conditioning_latents: torch.Tensor) -> torch.Tensor:
        """img_in(patchify(latents)) -> [S0, 3072]  (QwenImageBlockwiseMultiControlNet.preprocess :167-168)."""
        lat = conditioning_latents.to(device=self.device, dtype=BF).contiguous()
        if lat.dim() == 4:
            if lat.shape[0] != 1:
                raise _lib.PeError("controlnet conditioning: batch size 1 only")
            lat = lat[0]
        if lat.shape[0] * 4 != self.in_dim:
            raise _lib.PeError(f"controlnet conditioning has {lat.shape[0]} channels, this ControlNet takes {self.in_dim // 4}")
        tokens = ops.patchify(lat)                                 # [S0, C*4]
        if tokens.shape[1] != self.img_in_w.shape[1]:
            padded = torch.zeros((tokens.shape[0], self.img_in_w.shape[1]), dtype=BF, device=self.device)
            padded[:, :tokens.shape[1]] = tokens
            tokens = padded
        return ops.gemm(tokens, self.img_in_w, self.img_in_b)


class QwenImageBlockwiseMultiControlNet:
    """pipelines/qwen_image_physical.py:157-180: a list of ControlNets addressed by `ControlNetInput.controlnet_id`."""

    def __init__(self, models):
        self.models: List[QwenImageBlockWiseControlNet] = list(models) if isinstance(models, (list, tuple)) else [models]

    def preprocess(self, controlnet_inputs: Sequence[ControlNetInput], conditionings: Sequence[torch.Tensor], **kwargs):
        return [self.models[ci.controlnet_id].process_controlnet_conditioning(c) for ci, c in zip(controlnet_inputs, conditionings)]

    @staticmethod
    def is_active(controlnet_input: ControlNetInput, progress_id: int, num_inference_steps: int) -> bool:
        """the gate of blockwise_forward (:175-177)"""
        progress = (num_inference_steps - 1 - progress_id) / max(num_inference_steps - 1, 1)
        return not (progress > controlnet_input.start + 1e-4 or progress < controlnet_input.end - 1e-4)

    def active_controls(self, controlnet_inputs, processed_conditionings, progress_id: int, num_inference_steps: int):
        """[(ControlNet, processed conditioning [S0,3072], scale)] of the inputs that contribute at this step"""
        out = []
        for ci, cond in zip(controlnet_inputs, processed_conditionings):
            if self.is_active(ci, progress_id, num_inference_steps):
                out.append((self.models[ci.controlnet_id], cond, float(ci.scale)))
        return out
