"""Synthetic weights and inputs for the PhysicEdit hot path (no network: there are no checkpoints).

The state-dict LAYOUT (key names and shapes) is the reference's:
  DiT      DiffSynth-Studio/diffsynth/models/qwen_image_dit.py:404-430 (+ block :319-353)
  VAE      DiffSynth-Studio/diffsynth/models/qwen_image_vae.py:640-665
  adapter  DiffSynth-Studio/diffsynth/pipelines/helpers.py:123-140
and is checked key-by-key against `tests/golden/layout_keys.json` (dumped from the imported
reference by tests/golden/make_golden.py).

The VALUES are ours: every tensor is drawn from its own `torch.Generator` seeded by
(seed, crc32(key)), so any subset of tensors can be regenerated on any machine without
materialising the rest, and the oracle / the HIP path / the imported reference all see
identical weights without shipping them.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Iterable, List, Tuple

import torch

Shape = Tuple[int, ...]

DIT_DIM = 3072
DIT_HEADS = 24
DIT_HEAD_DIM = 128
DIT_FF = 12288
TXT_DIM = 3584
PATCH_DIM = 64
ADAPTER_HIDDEN = 10752
SPECIAL_TOKEN_NUM = 64


# --------------------------------------------------------------------------------------
# layouts
# --------------------------------------------------------------------------------------
def dit_block_layout(i: int) -> List[Tuple[str, Shape]]:
    p = f"transformer_blocks.{i}."
    d, ff = DIT_DIM, DIT_FF
    out: List[Tuple[str, Shape]] = []

    def lin(name, o, k):
        out.append((p + name + ".weight", (o, k)))
        out.append((p + name + ".bias", (o,)))

    lin("img_mod.1", 6 * d, d)
    for n in ("to_q", "to_k", "to_v"):
        lin("attn." + n, d, d)
    out.append((p + "attn.norm_q.weight", (DIT_HEAD_DIM,)))
    out.append((p + "attn.norm_k.weight", (DIT_HEAD_DIM,)))
    for n in ("add_q_proj", "add_k_proj", "add_v_proj"):
        lin("attn." + n, d, d)
    out.append((p + "attn.norm_added_q.weight", (DIT_HEAD_DIM,)))
    out.append((p + "attn.norm_added_k.weight", (DIT_HEAD_DIM,)))
    lin("attn.to_out.0", d, d)
    lin("attn.to_add_out", d, d)
    lin("img_mlp.net.0.proj", ff, d)
    lin("img_mlp.net.2", d, ff)
    lin("txt_mod.1", 6 * d, d)
    lin("txt_mlp.net.0.proj", ff, d)
    lin("txt_mlp.net.2", d, ff)
    return out


def dit_layout(num_layers: int) -> List[Tuple[str, Shape]]:
    d = DIT_DIM
    out: List[Tuple[str, Shape]] = [
        ("time_text_embed.timestep_embedder.linear_1.weight", (d, 256)),
        ("time_text_embed.timestep_embedder.linear_1.bias", (d,)),
        ("time_text_embed.timestep_embedder.linear_2.weight", (d, d)),
        ("time_text_embed.timestep_embedder.linear_2.bias", (d,)),
        ("txt_norm.weight", (TXT_DIM,)),
        ("img_in.weight", (d, PATCH_DIM)),
        ("img_in.bias", (d,)),
        ("txt_in.weight", (d, TXT_DIM)),
        ("txt_in.bias", (d,)),
    ]
    for i in range(num_layers):
        out += dit_block_layout(i)
    out += [
        ("norm_out.linear.weight", (2 * d, d)),
        ("norm_out.linear.bias", (2 * d,)),
        ("proj_out.weight", (PATCH_DIM, d)),
        ("proj_out.bias", (PATCH_DIM,)),
    ]
    return out


def controlnet_layout(num_layers: int, additional_in_dim: int = 0) -> List[Tuple[str, Shape]]:
    """QwenImageBlockWiseControlNet(num_layers, in_dim=64, additional_in_dim, dim=3072) -- models/qwen_image_controlnet.py:6-57
    (additional_in_dim = 4 is the inpaint variant: the conditioning latents carry one mask channel, x 2 x 2 patch)."""
    d = DIT_DIM
    out: List[Tuple[str, Shape]] = [("img_in.weight", (d, PATCH_DIM + additional_in_dim)), ("img_in.bias", (d,))]
    for i in range(num_layers):
        p = f"controlnet_blocks.{i}."
        out += [(p + "x_rms.weight", (d,)), (p + "y_rms.weight", (d,)),
                (p + "input_proj.weight", (d, d)), (p + "input_proj.bias", (d,)),
                (p + "output_proj.weight", (d, d)), (p + "output_proj.bias", (d,))]
    return out


def adapter_layout() -> List[Tuple[str, Shape]]:
    out: List[Tuple[str, Shape]] = []
    for head in ("head_dino", "head_vae"):
        out += [
            (f"{head}.0.weight", (ADAPTER_HIDDEN, TXT_DIM)),
            (f"{head}.0.bias", (ADAPTER_HIDDEN,)),
            (f"{head}.2.weight", (TXT_DIM, ADAPTER_HIDDEN)),
            (f"{head}.2.bias", (TXT_DIM,)),
        ]
    return out


def _resampler_layout(prefix: str, dim: int, max_media: int, depth: int = 2, heads: int = 8, dim_head: int = 64,
                      num_latents: int = 64) -> List[Tuple[str, Shape]]:
    inner = heads * dim_head
    out: List[Tuple[str, Shape]] = [(prefix + "latents", (num_latents, dim)), (prefix + "pos_emb.weight", (max_media, dim))]
    for i in range(depth):
        a, f = f"{prefix}layers.{i}.0.", f"{prefix}layers.{i}.1.net."
        out += [(a + "norm_media.weight", (dim,)), (a + "norm_media.bias", (dim,)), (a + "norm_latents.weight", (dim,)),
                (a + "norm_latents.bias", (dim,)), (a + "to_q.weight", (inner, dim)), (a + "to_kv.weight", (2 * inner, dim)),
                (a + "to_out.weight", (dim, inner)),
                (f + "0.weight", (dim,)), (f + "0.bias", (dim,)), (f + "1.weight", (4 * dim, dim)), (f + "1.bias", (4 * dim,)),
                (f + "3.weight", (dim, 4 * dim)), (f + "3.bias", (dim,))]
    out += [(prefix + "norm.weight", (dim,)), (prefix + "norm.bias", (dim,))]
    return out


def prior_layout() -> List[Tuple[str, Shape]]:
    """The training-time prior modules of QwenImagePhysicPipeline (pipelines/qwen_image_physical.py:205-219): two
    PerceiverResamplers (pipelines/helpers.py:66-109; dim 768 over DINOv2 patch tokens, dim 64 over patchified VAE latents; 64 latents,
    depth 2, 8 heads of 64), their frame-index embeddings and the two VisualThinkingAdapters (helpers.py:111-120) into the text width."""
    out = _resampler_layout("dino_resampler.", 768, 4096) + [("dino_time_embed.weight", (6, 768))]
    out += _resampler_layout("vae_resampler.", 64, 10240) + [("vae_time_embed.weight", (6, 64))]
    for name, dim in (("dino_resampler_adapter", 768), ("vae_resampler_adapter", 64)):
        out += [(f"{name}.net.0.weight", (3 * TXT_DIM, dim)), (f"{name}.net.0.bias", (3 * TXT_DIM,)),
                (f"{name}.net.2.weight", (TXT_DIM, 3 * TXT_DIM)), (f"{name}.net.2.bias", (TXT_DIM,))]
    return out


def dino_layout(hidden: int = 768, num_layers: int = 12, mlp_ratio: int = 4, patch: int = 14, image_size: int = 224,
                num_register_tokens: int = 4, final_norm: bool = False) -> List[Tuple[str, Shape]]:
    """transformers' Dinov2WithRegistersModel state dict (the encoder inside the reference's Dinov2withNorm, pipelines/dinov2.py:8-31;
    `final_norm`: the affine of the last LayerNorm, which Dinov2withNorm(normalize=True) removes)."""
    n = (image_size // patch) ** 2
    out: List[Tuple[str, Shape]] = [
        ("embeddings.cls_token", (1, 1, hidden)), ("embeddings.mask_token", (1, hidden)),
        ("embeddings.register_tokens", (1, num_register_tokens, hidden)), ("embeddings.position_embeddings", (1, n + 1, hidden)),
        ("embeddings.patch_embeddings.projection.weight", (hidden, 3, patch, patch)),
        ("embeddings.patch_embeddings.projection.bias", (hidden,))]
    for i in range(num_layers):
        L = f"encoder.layer.{i}."
        out += [(L + "norm1.weight", (hidden,)), (L + "norm1.bias", (hidden,))]
        for nm in ("query", "key", "value"):
            out += [(L + f"attention.attention.{nm}.weight", (hidden, hidden)), (L + f"attention.attention.{nm}.bias", (hidden,))]
        out += [(L + "attention.output.dense.weight", (hidden, hidden)), (L + "attention.output.dense.bias", (hidden,)),
                (L + "layer_scale1.lambda1", (hidden,)), (L + "norm2.weight", (hidden,)), (L + "norm2.bias", (hidden,)),
                (L + "mlp.fc1.weight", (mlp_ratio * hidden, hidden)), (L + "mlp.fc1.bias", (mlp_ratio * hidden,)),
                (L + "mlp.fc2.weight", (hidden, mlp_ratio * hidden)), (L + "mlp.fc2.bias", (hidden,)),
                (L + "layer_scale2.lambda1", (hidden,))]
    if final_norm:
        out += [("layernorm.weight", (hidden,)), ("layernorm.bias", (hidden,))]
    return out


def _vae_res(prefix: str, cin: int, cout: int) -> List[Tuple[str, Shape]]:
    out: List[Tuple[str, Shape]] = [
        (prefix + "norm1.gamma", (cin, 1, 1, 1)),
        (prefix + "conv1.weight", (cout, cin, 3, 3, 3)),
        (prefix + "conv1.bias", (cout,)),
        (prefix + "norm2.gamma", (cout, 1, 1, 1)),
        (prefix + "conv2.weight", (cout, cout, 3, 3, 3)),
        (prefix + "conv2.bias", (cout,)),
    ]
    if cin != cout:
        out += [
            (prefix + "conv_shortcut.weight", (cout, cin, 1, 1, 1)),
            (prefix + "conv_shortcut.bias", (cout,)),
        ]
    return out


def _vae_mid(prefix: str, c: int) -> List[Tuple[str, Shape]]:
    out: List[Tuple[str, Shape]] = [
        (prefix + "attentions.0.norm.gamma", (c, 1, 1)),
        (prefix + "attentions.0.to_qkv.weight", (3 * c, c, 1, 1)),
        (prefix + "attentions.0.to_qkv.bias", (3 * c,)),
        (prefix + "attentions.0.proj.weight", (c, c, 1, 1)),
        (prefix + "attentions.0.proj.bias", (c,)),
    ]
    out += _vae_res(prefix + "resnets.0.", c, c)
    out += _vae_res(prefix + "resnets.1.", c, c)
    return out


def vae_layout() -> List[Tuple[str, Shape]]:
    """QwenImageVAE(base_dim=96, z_dim=16, dim_mult=[1,2,4,4], num_res_blocks=2,
    temperal_downsample=[False,True,True]) -- qwen_image_vae.py:640-665."""
    out: List[Tuple[str, Shape]] = [
        ("encoder.conv_in.weight", (96, 3, 3, 3, 3)),
        ("encoder.conv_in.bias", (96,)),
    ]
    dims = [96, 96, 192, 384, 384]
    temporal = [False, True, True]
    idx = 0
    for i in range(4):
        cin, cout = dims[i], dims[i + 1]
        for _ in range(2):
            out += _vae_res(f"encoder.down_blocks.{idx}.", cin, cout)
            cin = cout
            idx += 1
        if i != 3:
            out += [
                (f"encoder.down_blocks.{idx}.resample.1.weight", (cout, cout, 3, 3)),
                (f"encoder.down_blocks.{idx}.resample.1.bias", (cout,)),
            ]
            if temporal[i]:
                out += [
                    (f"encoder.down_blocks.{idx}.time_conv.weight", (cout, cout, 3, 1, 1)),
                    (f"encoder.down_blocks.{idx}.time_conv.bias", (cout,)),
                ]
            idx += 1
    out += _vae_mid("encoder.mid_block.", 384)
    out += [
        ("encoder.norm_out.gamma", (384, 1, 1, 1)),
        ("encoder.conv_out.weight", (32, 384, 3, 3, 3)),
        ("encoder.conv_out.bias", (32,)),
        ("quant_conv.weight", (32, 32, 1, 1, 1)),
        ("quant_conv.bias", (32,)),
        ("post_quant_conv.weight", (16, 16, 1, 1, 1)),
        ("post_quant_conv.bias", (16,)),
        ("decoder.conv_in.weight", (384, 16, 3, 3, 3)),
        ("decoder.conv_in.bias", (384,)),
    ]
    out += _vae_mid("decoder.mid_block.", 384)
    # decoder dims = [384, 384, 384, 192, 96]; in_dim halves for i>0 (qwen_image_vae.py:570-573)
    ddims = [384, 384, 384, 192, 96]
    tup = [True, True, False]  # temperal_upsample = reversed(temperal_downsample)
    for i in range(4):
        cin, cout = ddims[i], ddims[i + 1]
        if i > 0:
            cin = cin // 2
        for j in range(3):
            out += _vae_res(f"decoder.up_blocks.{i}.resnets.{j}.", cin, cout)
            cin = cout
        if i != 3:
            out += [
                (f"decoder.up_blocks.{i}.upsamplers.0.resample.1.weight", (cout // 2, cout, 3, 3)),
                (f"decoder.up_blocks.{i}.upsamplers.0.resample.1.bias", (cout // 2,)),
            ]
            if tup[i]:
                out += [
                    (f"decoder.up_blocks.{i}.upsamplers.0.time_conv.weight", (cout * 2, cout, 3, 1, 1)),
                    (f"decoder.up_blocks.{i}.upsamplers.0.time_conv.bias", (cout * 2,)),
                ]
    out += [
        ("decoder.norm_out.gamma", (96, 1, 1, 1)),
        ("decoder.conv_out.weight", (3, 96, 3, 3, 3)),
        ("decoder.conv_out.bias", (3,)),
    ]
    return out


# --------------------------------------------------------------------------------------
# values
# --------------------------------------------------------------------------------------
def _gen(seed: int, key: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1000003 + zlib.crc32(key.encode("utf-8"))) & 0x7FFFFFFFFFFFFFFF)
    return g


def make_tensor(seed: int, key: str, shape: Shape, dtype=torch.bfloat16,
                fan_in_hint: int | None = None) -> torch.Tensor:
    """One synthetic tensor. Linear/conv weights and biases ~ U(-1/sqrt(fan_in), +1/sqrt(fan_in))
    (PyTorch's default Linear/Conv init family); norm gains ~ 1 + 0.1 U(-1,1)."""
    g = _gen(seed, key)
    u = torch.rand(shape, generator=g, dtype=torch.float32) * 2.0 - 1.0
    leaf = key.rsplit(".", 1)[-1]
    is_norm = leaf == "gamma" or (leaf == "weight" and len(shape) == 1)
    if is_norm:
        t = 1.0 + 0.1 * u
    else:
        if len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
        else:
            fan_in = fan_in_hint or shape[0]
        t = u * (1.0 / math.sqrt(fan_in))
    return t.to(dtype)


def make_state_dict(layout: Iterable[Tuple[str, Shape]], seed: int, dtype=torch.bfloat16,
                    device: str | torch.device = "cpu") -> Dict[str, torch.Tensor]:
    layout = list(layout)
    fan = {k[: -len(".weight")]: (math.prod(s[1:]) if len(s) >= 2 else None)
           for k, s in layout if k.endswith(".weight")}
    sd: Dict[str, torch.Tensor] = {}
    for key, shape in layout:
        hint = None
        if key.endswith(".bias"):
            hint = fan.get(key[: -len(".bias")])
        sd[key] = make_tensor(seed, key, shape, dtype=dtype, fan_in_hint=hint).to(device)
    return sd


def make_state_dict_device(layout: Iterable[Tuple[str, Shape]], seed: int, device,
                           dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    """Same distribution family, drawn directly with the DEVICE generator (used only for the
    full 60-layer benchmark model: 40.9 GB would take minutes through the CPU generator).
    Values differ from `make_state_dict` and from box to box only in the sense that no host can regenerate them: the 60-layer parity
    tests use `make_state_dict_hashed` (same bits on host and device), bench.py keeps this one."""
    layout = list(layout)
    fan = {k[: -len(".weight")]: (math.prod(s[1:]) if len(s) >= 2 else None)
           for k, s in layout if k.endswith(".weight")}
    g = torch.Generator(device=device)
    sd: Dict[str, torch.Tensor] = {}
    for key, shape in layout:
        g.manual_seed((seed * 1000003 + zlib.crc32(key.encode("utf-8"))) & 0x7FFFFFFFFFFFFFFF)
        u = torch.rand(shape, generator=g, dtype=torch.float32, device=device) * 2.0 - 1.0
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "gamma" or (leaf == "weight" and len(shape) == 1):
            t = 1.0 + 0.1 * u
        else:
            if len(shape) >= 2:
                fan_in = math.prod(shape[1:])
            else:
                fan_in = fan.get(key[: -len(".bias")]) or shape[0]
            t = u * (1.0 / math.sqrt(fan_in))
        sd[key] = t.to(dtype)
        del u, t
    return sd


# --------------------------------------------------------------------------------------
# counter-based values: the SAME bits on any device (fixture G21-G23: the reference runs the 60-layer model in the build container,
# the GPU box regenerates the 41 GB there instead of shipping them)
# --------------------------------------------------------------------------------------
def hash_uniform(key32: int, numel: int, device, chunk: int = 1 << 20) -> torch.Tensor:
    """fp32 U[-1, 1) on a 2^-23 grid: element i = murmur3's 32-bit finaliser of (i * 0x9E3779B1 + key32) mod 2^32, top 24 bits.
    Integer arithmetic in int64 with explicit masks, no device RNG.  The products (a value below 2^32 times a 32-bit constant) can exceed
    2^63: they WRAP modulo 2^64, which torch's int64 multiply does on the host and on the GPU alike (two's complement, no trap), and the
    mask that follows keeps the low 32 bits -- the same function bit for bit on both; tests/test_gpu_parity_configs.py compares the
    device-generated weights' checksums with the host's, and G21-G24 depend on it.  Chunks keep the temporaries cache-resident on the host (65 s for the 60-layer DiT on 8 cores)."""
    out = torch.empty(numel, dtype=torch.float32, device=device)
    n0 = min(numel, chunk)
    h = torch.empty(n0, dtype=torch.int64, device=device)
    t = torch.empty(n0, dtype=torch.int64, device=device)
    for s in range(0, numel, chunk):
        e = min(numel, s + chunk)
        hh, tt = h[:e - s], t[:e - s]
        torch.arange(s, e, dtype=torch.int64, device=device, out=hh)
        hh.mul_(0x9E3779B1).add_(int(key32) & 0xFFFFFFFF).bitwise_and_(0xFFFFFFFF)
        torch.bitwise_right_shift(hh, 16, out=tt); hh.bitwise_xor_(tt)
        hh.mul_(0x85EBCA6B).bitwise_and_(0xFFFFFFFF)
        torch.bitwise_right_shift(hh, 13, out=tt); hh.bitwise_xor_(tt)
        hh.mul_(0xC2B2AE35).bitwise_and_(0xFFFFFFFF)
        torch.bitwise_right_shift(hh, 16, out=tt); hh.bitwise_xor_(tt)
        hh.bitwise_right_shift_(8)                       # 24 bits: exact in fp32
        o = out[s:e]
        o.copy_(hh)
        o.mul_(2.0 ** -23).sub_(1.0)                     # both exact
    return out


def make_state_dict_hashed(layout: Iterable[Tuple[str, Shape]], seed: int, device="cpu", dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    """`make_state_dict`'s distribution family (U(+-1/sqrt(fan_in)) weights and biases, 1 + 0.1 U norm gains) from `hash_uniform`:
    identical tensors whether drawn on the host or on the GPU (tests/test_gpu_parity_configs.py regenerates on the device the model
    whose REFERENCE outputs tests/golden/make_golden.py stored).  The scale is an fp32 value on both sides, each op rounds once."""
    layout = list(layout)
    fan = {k[: -len(".weight")]: (math.prod(s[1:]) if len(s) >= 2 else None)
           for k, s in layout if k.endswith(".weight")}
    sd: Dict[str, torch.Tensor] = {}
    for key, shape in layout:
        key32 = (seed * 1000003 + zlib.crc32(key.encode("utf-8"))) & 0xFFFFFFFF
        u = hash_uniform(key32, math.prod(shape), device)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "gamma" or (leaf == "weight" and len(shape) == 1):
            u.mul_(float(torch.tensor(0.1, dtype=torch.float32))).add_(1.0)
        else:
            fan_in = math.prod(shape[1:]) if len(shape) >= 2 else (fan.get(key[: -len(".bias")]) or shape[0])
            u.mul_(float(torch.tensor(1.0 / math.sqrt(fan_in), dtype=torch.float32)))
        sd[key] = u.to(dtype).view(shape)
        del u
    return sd


def make_lora(seed: int, num_layers: int, rank: int, targets: Iterable[str] | None = None,
              std: float = 0.02, dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    """Checkpoint-format LoRA (scripts/train/train_multigpu.sh:30-31 target list; key format
    `transformer_blocks.i.<target>.lora_{A,B}.default.weight`)."""
    shapes = dict(dit_block_layout(0))
    if targets is None:
        targets = LORA_TARGETS
    sd: Dict[str, torch.Tensor] = {}
    for i in range(num_layers):
        for t in targets:
            out_f, in_f = shapes[f"transformer_blocks.0.{t}.weight"]
            ka = f"transformer_blocks.{i}.{t}.lora_A.default.weight"
            kb = f"transformer_blocks.{i}.{t}.lora_B.default.weight"
            sd[ka] = (torch.randn((rank, in_f), generator=_gen(seed, ka)) * std).to(dtype)
            sd[kb] = (torch.randn((out_f, rank), generator=_gen(seed, kb)) * std).to(dtype)
    return sd


LORA_TARGETS = (
    "attn.to_q", "attn.to_k", "attn.to_v",
    "attn.add_q_proj", "attn.add_k_proj", "attn.add_v_proj",
    "attn.to_out.0", "attn.to_add_out",
    "img_mlp.net.2", "img_mod.1", "txt_mlp.net.2", "txt_mod.1",
)


# --------------------------------------------------------------------------------------
# synthetic inputs
# --------------------------------------------------------------------------------------
def make_prompt_emb(seed: int, T: int, dtype=torch.bfloat16) -> torch.Tensor:
    g = torch.Generator("cpu").manual_seed(seed)
    return torch.randn((1, T, TXT_DIM), generator=g, dtype=torch.float32).to(dtype)


def make_special_token_mask(T: int, n_special: int = SPECIAL_TOKEN_NUM, tail: int = 6) -> torch.Tensor:
    """`n_special` contiguous positions ending `tail` tokens before T."""
    m = torch.zeros((1, T), dtype=torch.bool)
    m[0, T - tail - n_special: T - tail] = True
    return m


def make_noise(seed: int, height: int, width: int, dtype=torch.bfloat16) -> torch.Tensor:
    """BasePipeline.generate_noise (utils/__init__.py:119-124) as NoiseInitializer calls it
    (qwen_image_physical.py:688): CPU generator, drawn directly in `dtype`."""
    g = torch.Generator("cpu").manual_seed(seed)
    return torch.randn((1, 16, height // 8, width // 8), generator=g, device="cpu", dtype=dtype)


def make_edit_image_u8(height: int, width: int, seed: int = 0):
    import numpy as np
    return (np.random.RandomState(seed).rand(height, width, 3) * 255).astype("uint8")
