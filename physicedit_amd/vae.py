"""QwenImageVAE encode/decode on the HIP kernels (single frame, B = 1).

API mirror of DiffSynth-Studio/diffsynth/models/qwen_image_vae.py: `QwenImageVAE.encode(x, **kwargs)`
(:706-717) and `.decode(x, **kwargs)` (:719-729) take/return NCHW bf16 tensors and swallow the
`tiled/tile_size/tile_stride/device` kwargs exactly like the reference does.
Weights keep the reference's state-dict names.  The launch sequence is the C-ABI composite pe_vae_encode / pe_vae_decode (csrc/vae_graph.hip) over the
kernels of csrc/vae.hip (implicit-GEMM NHWC conv, channel RMS-norm+SiLU, D=384 attention); the pipeline's uint8 image
I/O (preprocess_image / vae_output_to_image) is fused into its first / last kernel.

Weight repacking at load (one time): conv weights [Cout,Cin,3,3,3] -> last temporal tap ->
[Cout_p][3*3][Cin_p] bf16 with both channel counts zero-padded to a multiple of 32.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

import ctypes as C

from . import _lib
from ._lib import VaeConv, VaeMid, VaeRes, VaeWeights, check, lib, stream_ptr

BF = torch.bfloat16

_VAE_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
             0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
_VAE_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
            3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]


def _pad32(c: int) -> int:
    return (c + 31) // 32 * 32


class _Conv:
    __slots__ = ("w", "b", "cin", "cout", "cin_p", "cout_p", "k")

    def __init__(self, weight: torch.Tensor, bias: torch.Tensor, device):
        if weight.dim() == 5:               # causal conv3d: only the last temporal tap sees data at T=1
            weight = weight[:, :, -1]
        cout, cin, kh, kw = weight.shape
        assert kh == kw and kh in (1, 3)
        self.k = kh
        self.cin, self.cout = cin, cout
        self.cin_p, self.cout_p = _pad32(cin), _pad32(cout)
        w = torch.zeros((self.cout_p, kh * kw, self.cin_p), dtype=BF)
        w[:cout, :, :cin] = weight.to(BF).permute(0, 2, 3, 1).reshape(cout, kh * kw, cin)
        b = torch.zeros((self.cout_p,), dtype=BF)
        b[:cout] = bias.to(BF)
        self.w = w.contiguous().to(device)
        self.b = b.to(device)


class QwenImageVAE:
    """Thin caller of the C-ABI composites pe_vae_encode / pe_vae_decode: this class only repacks the weights once
    (load time) and owns the workspace; the launch sequence lives in csrc/vae_graph.hip."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.PeError("QwenImageVAE needs a HIP device: physicedit_amd has no CPU path")
        lib()
        self.convs: Dict[str, _Conv] = {}
        self.gammas: Dict[str, torch.Tensor] = {}
        for k, v in state_dict.items():
            if k.endswith(".weight") and "time_conv" not in k:      # time_conv is never entered at T=1 (:259-299)
                name = k[: -len(".weight")]
                self.convs[name] = _Conv(v, state_dict[name + ".bias"], self.device)
            elif k.endswith(".gamma"):
                g = v.reshape(-1).to(BF)
                gp = torch.zeros((_pad32(g.numel()),), dtype=BF)
                gp[: g.numel()] = g
                self.gammas[k[: -len(".gamma")]] = gp.to(self.device)
        mean = torch.tensor(_VAE_MEAN).view(1, 16, 1, 1, 1)
        std = 1 / torch.tensor(_VAE_STD).view(1, 16, 1, 1, 1)          # (:703-704)
        self.mean = mean.to(BF).reshape(-1).to(self.device)           # `.to(dtype=x.dtype)` (:713,723)
        self.std = std.to(BF).reshape(-1).to(self.device)
        self.zero = torch.zeros((256,), dtype=BF, device=self.device)
        self.z_dim = 16
        self._handle = C.c_void_p()
        self._ws: Optional[torch.Tensor] = None
        self._create()

    # ------------------------------------------------------------------------------------------
    def _cv(self, dst: VaeConv, name: str):
        c = self.convs[name]
        dst.w, dst.b, dst.cin_p, dst.cout_p, dst.ksize = c.w.data_ptr(), c.b.data_ptr(), c.cin_p, c.cout_p, c.k

    def _rs(self, dst: VaeRes, p: str):
        self._cv(dst.conv1, p + "conv1")
        self._cv(dst.conv2, p + "conv2")
        if (p + "conv_shortcut") in self.convs:
            self._cv(dst.shortcut, p + "conv_shortcut")
        dst.norm1_g, dst.norm2_g = self.gammas[p + "norm1"].data_ptr(), self.gammas[p + "norm2"].data_ptr()

    def _md(self, dst: VaeMid, p: str):
        self._rs(dst.res0, p + "resnets.0.")
        dst.attn.norm_g = self.gammas[p + "attentions.0.norm"].data_ptr()
        self._cv(dst.attn.to_qkv, p + "attentions.0.to_qkv")
        self._cv(dst.attn.proj, p + "attentions.0.proj")
        self._rs(dst.res1, p + "resnets.1.")

    def _create(self):
        w = VaeWeights()
        self._cv(w.enc_conv_in, "encoder.conv_in")
        idx, r, d = 0, 0, 0
        for i in range(4):                                   # Encoder3d.down_blocks is a flat list (:376-396)
            for _ in range(2):
                self._rs(w.enc_res[r], f"encoder.down_blocks.{idx}.")
                r += 1
                idx += 1
            if i != 3:
                self._cv(w.enc_down[d], f"encoder.down_blocks.{idx}.resample.1")
                d += 1
                idx += 1
        self._md(w.enc_mid, "encoder.mid_block.")
        w.enc_norm_out_g = self.gammas["encoder.norm_out"].data_ptr()
        self._cv(w.enc_conv_out, "encoder.conv_out")
        self._cv(w.quant_conv, "quant_conv")
        self._cv(w.post_quant_conv, "post_quant_conv")
        self._cv(w.dec_conv_in, "decoder.conv_in")
        self._md(w.dec_mid, "decoder.mid_block.")
        for i in range(4):
            for j in range(3):
                self._rs(w.dec_res[i * 3 + j], f"decoder.up_blocks.{i}.resnets.{j}.")
            if i != 3:
                self._cv(w.dec_up[i], f"decoder.up_blocks.{i}.upsamplers.0.resample.1")
        w.dec_norm_out_g = self.gammas["decoder.norm_out"].data_ptr()
        self._cv(w.dec_conv_out, "decoder.conv_out")
        w.mean, w.inv_std, w.zero_page = self.mean.data_ptr(), self.std.data_ptr(), self.zero.data_ptr()
        check(lib().pe_vae_create(C.byref(w), C.byref(self._handle)), "pe_vae_create")

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                lib().pe_vae_destroy(self._handle)
        except Exception:
            pass

    def _workspace(self, H: int, W: int):
        n = lib().pe_vae_workspace_bytes(H, W)
        if self._ws is None or self._ws.numel() < n + 256:
            self._ws = None
            self._ws = torch.empty((n + 256,), dtype=torch.uint8, device=self.device)
        return (self._ws.data_ptr() + 255) // 256 * 256, n

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
        """[1,3,H,W] bf16 in [-1,1] -> normalised latents [1,16,H/8,W/8].  A uint8 [H,W,3] device tensor is accepted too:
        BasePipeline.preprocess_image is then applied inside the first kernel (same bf16 rounding points)."""
        if x.dtype == torch.uint8:
            assert x.dim() == 3 and x.shape[2] == 3 and x.is_cuda
            x = x.contiguous()
            H, W, fmt, squeeze = x.shape[0], x.shape[1], _lib.IMAGE_U8_HWC, False
        else:
            squeeze = x.dim() == 5
            if squeeze:
                assert x.shape[2] == 1
                x = x[:, :, 0]
            assert x.shape[0] == 1 and x.shape[1] == 3 and x.is_cuda
            x = x.to(BF).contiguous()
            H, W, fmt = x.shape[2], x.shape[3], _lib.IMAGE_BF16_NCHW
        assert H % 8 == 0 and W % 8 == 0
        out = torch.empty((1, 16, H // 8, W // 8), dtype=BF, device=self.device)
        ws, n = self._workspace(H, W)
        check(lib().pe_vae_encode(self._handle, x.data_ptr(), fmt, H, W, out.data_ptr(), ws, n, stream_ptr()), "pe_vae_encode")
        return out.unsqueeze(2) if squeeze else out

    @torch.no_grad()
    def decode(self, x: torch.Tensor, output_u8: bool = False, **kwargs) -> torch.Tensor:
        """normalised latents [1,16,h,w] -> image [1,3,8h,8w] bf16; output_u8: [8h,8w,3] uint8 with
        BasePipeline.vae_output_to_image's map applied inside the last kernel."""
        squeeze = x.dim() == 5
        if squeeze:
            assert x.shape[2] == 1
            x = x[:, :, 0]
        assert x.shape[0] == 1 and x.shape[1] == 16 and x.is_cuda
        x = x.to(BF).contiguous()
        h, w = x.shape[2], x.shape[3]
        ws, n = self._workspace(h * 8, w * 8)
        if output_u8:
            out = torch.empty((h * 8, w * 8, 3), dtype=torch.uint8, device=self.device)
            check(lib().pe_vae_decode(self._handle, x.data_ptr(), h, w, out.data_ptr(), _lib.IMAGE_U8_HWC, ws, n, stream_ptr()),
                  "pe_vae_decode")
            return out
        out = torch.empty((1, 3, h * 8, w * 8), dtype=BF, device=self.device)
        check(lib().pe_vae_decode(self._handle, x.data_ptr(), h, w, out.data_ptr(), _lib.IMAGE_BF16_NCHW, ws, n, stream_ptr()),
              "pe_vae_decode")
        return out.unsqueeze(2) if squeeze else out


def preprocess_image(img_u8_hwc, device, dtype=BF) -> torch.Tensor:
    """BasePipeline.preprocess_image (utils/__init__.py:60-66): uint8 HWC -> [1,3,H,W] in [-1,1], the cast
    to `dtype` happening BEFORE the affine map, which therefore rounds in bf16."""
    import numpy as np
    image = torch.Tensor(np.array(img_u8_hwc, dtype=np.float32))
    image = image.to(dtype=dtype, device=device)
    image = image * ((1 - (-1)) / 255) + (-1)
    return image.permute(2, 0, 1).unsqueeze(0).contiguous()


def vae_output_to_u8(vae_output: torch.Tensor) -> torch.Tensor:
    """BasePipeline.vae_output_to_image (utils/__init__.py:76-83) up to the PIL wrap: HWC uint8 (truncation)."""
    x = vae_output.mean(dim=0).permute(1, 2, 0)
    image = ((x - (-1)) * (255 / (1 - (-1)))).clip(0, 255)
    return image.to(device="cpu", dtype=torch.uint8)
