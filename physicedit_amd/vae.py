"""QwenImageVAE encode/decode on the HIP kernels (single frame, B = 1).

API mirror of DiffSynth-Studio/diffsynth/models/qwen_image_vae.py: `QwenImageVAE.encode(x, **kwargs)`
(:706-717) and `.decode(x, **kwargs)` (:719-729) take/return NCHW bf16 tensors and swallow the
`tiled/tile_size/tile_stride/device` kwargs exactly like the reference does.
Weights keep the reference's state-dict names.  The graph below is host sequencing only; every op is
a kernel from csrc/vae.hip (implicit-GEMM NHWC conv, channel RMS-norm+SiLU, D=384 attention).

Weight repacking at load (one time): conv weights [Cout,Cin,3,3,3] -> last temporal tap ->
[Cout_p][3*3][Cin_p] bf16 with both channel counts zero-padded to a multiple of 32.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _lib
from ._lib import check, lib, stream_ptr

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

    # ------------------------------------------------------------------------------------------
    def _buf(self, npix: int, cp: int) -> torch.Tensor:
        return torch.empty((npix, cp), dtype=BF, device=self.device)

    def _conv(self, name, x, H, W, res=None, stride=1, upsample=False):
        c = self.convs[name]
        assert x.shape[1] == c.cin_p, (name, x.shape, c.cin_p)
        if stride == 2:
            Ho, Wo = H // 2, W // 2
        elif upsample:
            Ho, Wo = H * 2, W * 2
        else:
            Ho, Wo = H, W
        out = self._buf(Ho * Wo, c.cout_p)
        check(lib().pe_conv2d_nhwc(x.data_ptr(), c.w.data_ptr(), c.b.data_ptr(),
                                   None if res is None else res.data_ptr(), out.data_ptr(), self.zero.data_ptr(),
                                   H, W, c.cin_p, c.cout_p, c.k, stride, 1 if upsample else 0, stream_ptr()),
              "pe_conv2d_nhwc")
        return out, Ho, Wo

    def _norm(self, name, x, C, silu=True):
        out = torch.empty_like(x)
        if x.shape[1] != C:
            out.zero_()
        check(lib().pe_vae_rmsnorm(x.data_ptr(), self.gammas[name].data_ptr(), out.data_ptr(), x.shape[0], C,
                                   x.shape[1], 1 if silu else 0, stream_ptr()), "pe_vae_rmsnorm")
        return out

    def _res(self, p, x, H, W):
        """QwenImageResidualBlock.forward (:112-152)."""
        cin = self.convs[p + "conv1"].cin
        cout = self.convs[p + "conv1"].cout
        h = x
        if (p + "conv_shortcut") in self.convs:
            h, _, _ = self._conv(p + "conv_shortcut", x, H, W)
        y = self._norm(p + "norm1", x, cin)
        y, _, _ = self._conv(p + "conv1", y, H, W)
        y = self._norm(p + "norm2", y, cout)
        y, _, _ = self._conv(p + "conv2", y, H, W, res=h)
        return y

    def _attn(self, p, x, H, W):
        """QwenImageAttentionBlock.forward (:173-198)."""
        C = 384
        N = H * W
        y = self._norm(p + "norm", x, C, silu=False)
        qkv, _, _ = self._conv(p + "to_qkv", y, H, W)                  # [N,1152]
        vt = torch.empty((C * ((N + 31) // 32 * 32),), dtype=BF, device=self.device)
        o = self._buf(N, C)
        check(lib().pe_vae_attention(qkv.data_ptr(), vt.data_ptr(), o.data_ptr(), N, stream_ptr()), "pe_vae_attention")
        y, _, _ = self._conv(p + "proj", o, H, W, res=x)               # x + identity (:198)
        return y

    def _mid(self, p, x, H, W):
        x = self._res(p + "resnets.0.", x, H, W)
        x = self._attn(p + "attentions.0.", x, H, W)
        return self._res(p + "resnets.1.", x, H, W)

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
        """[1,3,H,W] bf16 in [-1,1] -> normalised latents [1,16,H/8,W/8]."""
        squeeze = x.dim() == 5
        if squeeze:
            assert x.shape[2] == 1
            x = x[:, :, 0]
        assert x.shape[0] == 1 and x.shape[1] == 3 and x.is_cuda
        x = x.to(BF).contiguous()
        H, W = x.shape[2], x.shape[3]
        assert H % 8 == 0 and W % 8 == 0
        a = self._buf(H * W, 32)
        check(lib().pe_nchw_to_nhwc(x.data_ptr(), a.data_ptr(), 3, H * W, 32, 0, None, None, stream_ptr()), "pe_nchw_to_nhwc")
        a, _, _ = self._conv("encoder.conv_in", a, H, W)
        idx = 0
        for i in range(4):
            for _ in range(2):
                a = self._res(f"encoder.down_blocks.{idx}.", a, H, W)
                idx += 1
            if i != 3:
                a, H, W = self._conv(f"encoder.down_blocks.{idx}.resample.1", a, H, W, stride=2)
                idx += 1
        a = self._mid("encoder.mid_block.", a, H, W)
        a = self._norm("encoder.norm_out", a, 384)
        a, _, _ = self._conv("encoder.conv_out", a, H, W)
        a, _, _ = self._conv("quant_conv", a, H, W)
        out = torch.empty((1, 16, H, W), dtype=BF, device=self.device)
        check(lib().pe_nhwc_to_nchw(a.data_ptr(), out.data_ptr(), 16, H * W, a.shape[1], 2, self.mean.data_ptr(),
                                    self.std.data_ptr(), stream_ptr()), "pe_nhwc_to_nchw")
        return out.unsqueeze(2) if squeeze else out

    @torch.no_grad()
    def decode(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
        """normalised latents [1,16,h,w] -> image [1,3,8h,8w] bf16."""
        squeeze = x.dim() == 5
        if squeeze:
            assert x.shape[2] == 1
            x = x[:, :, 0]
        assert x.shape[0] == 1 and x.shape[1] == 16 and x.is_cuda
        x = x.to(BF).contiguous()
        H, W = x.shape[2], x.shape[3]
        a = self._buf(H * W, 32)
        check(lib().pe_nchw_to_nhwc(x.data_ptr(), a.data_ptr(), 16, H * W, 32, 1, self.mean.data_ptr(),
                                    self.std.data_ptr(), stream_ptr()), "pe_nchw_to_nhwc")
        a, _, _ = self._conv("post_quant_conv", a, H, W)
        a, _, _ = self._conv("decoder.conv_in", a, H, W)
        a = self._mid("decoder.mid_block.", a, H, W)
        for i in range(4):
            for j in range(3):
                a = self._res(f"decoder.up_blocks.{i}.resnets.{j}.", a, H, W)
            if i != 3:
                # nearest-exact 2x (exact copy, :213-214) fused into the following conv's gather
                a, H, W = self._conv(f"decoder.up_blocks.{i}.upsamplers.0.resample.1", a, H, W, upsample=True)
        a = self._norm("decoder.norm_out", a, 96)
        a, _, _ = self._conv("decoder.conv_out", a, H, W)
        out = torch.empty((1, 3, H, W), dtype=BF, device=self.device)
        check(lib().pe_nhwc_to_nchw(a.data_ptr(), out.data_ptr(), 3, H * W, a.shape[1], 0, None, None, stream_ptr()),
              "pe_nhwc_to_nchw")
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
