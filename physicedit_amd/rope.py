"""Host-side 3-axis RoPE tables of the Qwen-Image DiT, as real cos/sin fp32 tables.

Same values as QwenEmbedRope(theta=10000, axes_dim=[16,56,56], scale_rope=True).forward
(DiffSynth-Studio/diffsynth/models/qwen_image_dit.py:60-165): per image `idx` the frame axis uses
position idx (16 dims), height/width use centred positions -ceil(n/2)..floor(n/2)-1 (56+56 dims);
text tokens sit on the diagonal starting at max(h//2, w//2) over all images.  Built once per
geometry on the host and cached on the device (the reference caches per "{idx}_{h}_{w}" too).
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import torch

AXES_DIM = (16, 56, 56)
THETA = 10000


def _angles(index: torch.Tensor, dim: int) -> torch.Tensor:
    inv = 1.0 / torch.pow(THETA, torch.arange(0, dim, 2).to(torch.float32).div(dim))
    return torch.outer(index, inv)  # int64 x fp32 -> fp32, as rope_params does (:86-89)


def rope_angles(img_shapes: Sequence[Tuple[int, int, int]], txt_len: int, sampling: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """sampling = QwenEmbedRope.forward_sampling (:168-226, `edit_rope_interpolation=True`): an image idx > 0 whose grid differs from
    image 0's takes its height / width entries from image 0's table at linspace(0, n0 - 1, n).long() -- it spans the same coordinate
    range as the target image instead of its own centred one -- and keeps its own frame position.  (The reference caches tables per
    "{idx}_{h}_{w}" inside the module, so its result can depend on earlier calls with other geometries; this is the fresh-module value.)"""
    vid = []
    max_vid_index = 0
    h0, w0 = (img_shapes[0][1], img_shapes[0][2]) if len(img_shapes) else (0, 0)
    for idx, (frame, height, width) in enumerate(img_shapes):
        a_f = _angles(torch.arange(idx, idx + frame), AXES_DIM[0])
        a_h = _angles(torch.arange(height) - (height - height // 2), AXES_DIM[1])
        a_w = _angles(torch.arange(width) - (width - width // 2), AXES_DIM[2])
        if sampling and idx > 0 and (height, width) != (h0, w0):
            a_h = _angles(torch.arange(h0) - (h0 - h0 // 2), AXES_DIM[1])[torch.linspace(0, h0 - 1, height).long()]
            a_w = _angles(torch.arange(w0) - (w0 - w0 // 2), AXES_DIM[2])[torch.linspace(0, w0 - 1, width).long()]
        ang = torch.cat([
            a_f[:, None, None, :].expand(frame, height, width, -1),
            a_h[None, :, None, :].expand(frame, height, width, -1),
            a_w[None, None, :, :].expand(frame, height, width, -1)], dim=-1)
        vid.append(ang.reshape(frame * height * width, -1))
        max_vid_index = max(max_vid_index, height // 2, width // 2)
    t_idx = torch.arange(max_vid_index, max_vid_index + txt_len)
    txt = torch.cat([_angles(t_idx, d) for d in AXES_DIM], dim=1)
    return torch.cat(vid, dim=0), txt


def rope_cos_sin(img_shapes: Sequence[Tuple[int, int, int]], txt_len: int, sampling: bool = False):
    """(cos_img, sin_img, cos_txt, sin_txt) fp32 CPU tensors.  Uses torch.polar like rope_params
    (:90) -- its cos/sin differ from torch.cos/torch.sin by 1 fp32 ulp on ~5% of the entries."""
    a_img, a_txt = rope_angles(img_shapes, txt_len, sampling)
    p_img = torch.polar(torch.ones_like(a_img), a_img)
    p_txt = torch.polar(torch.ones_like(a_txt), a_txt)
    return (p_img.real.contiguous(), p_img.imag.contiguous(), p_txt.real.contiguous(), p_txt.imag.contiguous())


class RopeCache:
    def __init__(self, device):
        self.device = device
        self._cache: Dict[tuple, tuple] = {}

    def get_segments(self, img_shapes: Sequence[Tuple[int, int, int]], txt_lens: Sequence[int]):
        """EliGen (QwenImageDiT.process_entity_masks, qwen_image_dit.py:439-446): several prompts in the text stream, each with
        the text positions of a prompt that stands alone -> (cos_img, sin_img, cos_txt, sin_txt), text tables concatenated."""
        key = (tuple(tuple(s) for s in img_shapes), tuple(int(n) for n in txt_lens))
        if key not in self._cache:
            parts = [self.get(img_shapes, n) for n in txt_lens]
            self._cache[key] = (parts[0][0], parts[0][1], torch.cat([p[2] for p in parts]).contiguous(),
                                torch.cat([p[3] for p in parts]).contiguous())
        return self._cache[key]

    def get(self, img_shapes: Sequence[Tuple[int, int, int]], txt_len: int, sampling: bool = False):
        """-> (cos_img, sin_img, cos_txt, sin_txt) fp32 device tensors [S_img,64] / [T,64]."""
        key = (tuple(tuple(s) for s in img_shapes), int(txt_len), bool(sampling))
        if key not in self._cache:
            self._cache[key] = tuple(t.to(self.device) for t in rope_cos_sin(img_shapes, txt_len, sampling))
        return self._cache[key]
