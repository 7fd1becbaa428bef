"""CPU ORACLE for the PhysicEdit denoising hot path.  TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module.  The product path (physicedit_amd/*) never does and fails loudly when the HIP library is
missing.

What it is: an independent, functional restatement (plain functions over a state-dict, no
nn.Module tree) of the reference's algorithm for the path named in BASELINE.json `north_star`:

    flow-match scheduler -> [adapter on 64 special tokens -> DiT forward] x {posi, nega}
    -> CFG combine -> Euler step, x N steps -> VAE decode         (+ VAE encode of the edit image)

It is written with torch CPU ops because the reference's arithmetic IS "bf16 tensors through
torch CPU ops": every rounding boundary is reproduced by using the same op
at the same place, in bf16.  Each function cites the reference file:line it follows
(paths relative to /root/reference/DiffSynth-Studio/diffsynth/).

Pinning: the reference ships no tests and no golden vectors.  This oracle is
pinned against outputs of the reference itself, imported in the build container by
`tests/golden/make_golden.py`, which wrote the fixtures `tests/golden/*.safetensors`;
`tests/test_oracle_golden.py` replays them (CPU, `-m "not gpu"`).  "Bit-exact" is a statement about ONE host code path: torch's CPU
bf16 matmul is oneDNN's, which picks its kernel by the host's ISA (AMX, avx512_bf16 or the AVX-512 jit GEMM sum a K dimension in
different orders) and partitions by thread count, so the reference run on another host differs from itself by an ulp here and there.
The fixtures (round 6) were written, and the tests run, with that path pinned: ONEDNN_MAX_CPU_ISA=AVX512_CORE_VNNI, 8 threads
(`tests/conftest.py`; each fixture records it under `host_math`).

One optional branch cannot be pinned by CPU fixtures: the e4m3 "FP8 computation" Linear (`fp8_linear`, `fp8_quantize_rows`,
`to_fp8_state_dict` below; BASELINE.json configs[2]).  The reference's implementation ends in torch._scaled_mm, which does
not run on CPU with per-row scales, so no golden vector can be generated for it in the build container.  It is pinned on the
GPU instead, against the very call the reference makes: tests/test_gpu_fp8.py compares pe_quantize_rows_e4m3 with torch's own
device ops (bit-exact) and pe_gemm_e4m3 with torch._scaled_mm on the same operands (bit-identical on every tested shape,
profiles/r02_parity.json); this file's restatement of _scaled_mm (exact sum) is only the CPU-side cross-check.
Everything else in this file is pinned bit-exact by the G1..G12 fixtures.

`dtype=torch.float32` runs the same graph in fp32 (used for the "distance to fp32 truth" parity
bound in tests; never a reference behaviour).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# ======================================================================================
# scheduler   (schedulers/flow_match.py)
# ======================================================================================
class FlowMatchTables:
    """FlowMatchScheduler as QwenImagePhysicPipeline configures it
    (pipelines/qwen_image_physical.py:192: sigma_min=0, sigma_max=1, extra_one_step=True,
    exponential_shift=True, exponential_shift_mu=0.8, shift_terminal=0.02)."""

    def __init__(self, num_inference_steps: int = 100, dynamic_shift_len: Optional[int] = None,
                 denoising_strength: float = 1.0, exponential_shift_mu: Optional[float] = None):
        self.set_timesteps(num_inference_steps, dynamic_shift_len, denoising_strength, exponential_shift_mu)

    @staticmethod
    def calculate_shift(image_seq_len, base_seq_len=256, max_seq_len=8192, base_shift=0.5, max_shift=0.9):
        # flow_match.py:114-125
        m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
        b = base_shift - m * base_seq_len
        return image_seq_len * m + b

    def set_timesteps(self, num_inference_steps, dynamic_shift_len=None, denoising_strength=1.0,
                      exponential_shift_mu=None):
        # flow_match.py:34-69 with the pipeline's constructor flags folded in
        sigma_min, sigma_max = 0.0, 1.0
        sigma_start = sigma_min + (sigma_max - sigma_min) * denoising_strength
        sigmas = torch.linspace(sigma_start, sigma_min, num_inference_steps + 1)[:-1]  # extra_one_step
        if exponential_shift_mu is not None:
            mu = exponential_shift_mu
        elif dynamic_shift_len is not None:
            mu = self.calculate_shift(dynamic_shift_len)
        else:
            mu = 0.8
        sigmas = math.exp(mu) / (math.exp(mu) + (1 / sigmas - 1))  # exponential_shift
        one_minus_z = 1 - sigmas  # shift_terminal = 0.02
        scale_factor = one_minus_z[-1] / (1 - 0.02)
        sigmas = 1 - (one_minus_z / scale_factor)
        self.mu = mu
        self.sigmas = sigmas
        self.timesteps = sigmas * 1000

    def add_noise(self, original_samples, noise, progress_id=0):
        # flow_match.py:94-100
        sigma = self.sigmas[progress_id]
        return (1 - sigma) * original_samples + sigma * noise

    def return_to_timestep(self, progress_id, sample, sample_stablized):
        # flow_match.py:85-91
        return (sample - sample_stablized) / self.sigmas[progress_id]

    def step(self, model_output, progress_id, sample):
        # utils/__init__.py:150-156 -> flow_match.py:72-82 (argmin over |timesteps - t| == progress_id)
        sigma = self.sigmas[progress_id]
        if progress_id + 1 >= len(self.timesteps):
            sigma_ = 0
        else:
            sigma_ = self.sigmas[progress_id + 1]
        return sample + model_output * (sigma_ - sigma)


def adapter_t_range() -> Tuple[float, float]:
    """t_min/t_max handed to VisualThinkingDualAdapter (qwen_image_physical.py:225): min/max of the
    DEFAULT 100-step, mu=0.8 timetable built in the scheduler constructor."""
    tab = FlowMatchTables(100)
    return tab.timesteps.min().item(), tab.timesteps.max().item()


# ======================================================================================
# small ops   (models/utils.py, models/qwen_image_dit.py)
# ======================================================================================
def rmsnorm(x: torch.Tensor, weight: Optional[torch.Tensor], eps: float = 1e-6) -> torch.Tensor:
    # models/utils.py:250-257
    input_dtype = x.dtype
    variance = x.to(torch.float32).square().mean(-1, keepdim=True)
    x = x * torch.rsqrt(variance + eps)
    x = x.to(input_dtype)
    if weight is not None:
        x = x * weight
    return x


def timestep_sinusoid(timestep: torch.Tensor) -> torch.Tensor:
    """get_timestep_embedding(models/utils.py:189-216) as TimestepEmbeddings(256, 3072, scale=1000,
    align_dtype_to_timestep=True, flip_sin_to_cos=True, downscale_freq_shift=0) calls it (:274-293).
    `timestep` is already t/1000 in the pipeline dtype, shape [B]."""
    half_dim = 128
    exponent = -math.log(10000) * torch.arange(start=0, end=half_dim, dtype=torch.float32)
    exponent = exponent / (half_dim - 0)
    emb = torch.exp(exponent)
    emb = emb.to(timestep.dtype)  # align_dtype_to_timestep
    emb = timestep[:, None].float() * emb[None, :]
    emb = 1000 * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    emb = torch.cat([emb[:, half_dim:], emb[:, :half_dim]], dim=-1)  # flip_sin_to_cos
    return emb


def time_text_embed(sd: SD, timestep: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    # models/utils.py:290-293 + _DiffusersCompatibleTimestepProj :260-271
    p = "time_text_embed.timestep_embedder."
    x = timestep_sinusoid(timestep).to(dtype)
    x = _linear(sd, p + "linear_1", x)
    x = F.silu(x)
    x = _linear(sd, p + "linear_2", x)
    return x


_AXES_DIM = (16, 56, 56)
_THETA = 10000


def _rope_params(index: torch.Tensor, dim: int) -> torch.Tensor:
    # qwen_image_dit.py:80-91
    freqs = torch.outer(index, 1.0 / torch.pow(_THETA, torch.arange(0, dim, 2).to(torch.float32).div(dim)))
    return torch.polar(torch.ones_like(freqs), freqs)


def rope_tables(img_shapes: Sequence[Tuple[int, int, int]], txt_len: int, sampling: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """QwenEmbedRope.forward (qwen_image_dit.py:123-165) with scale_rope=True, axes (16,56,56); sampling=True: forward_sampling
    (:168-226) on a FRESH module (its per-"{idx}_{h}_{w}" cache makes later calls history dependent): an image idx > 0 whose
    grid differs from image 0's is sampled from image 0's table at linspace(0, n0 - 1, n).long(), frame entries its own.
    Returns complex64 (vid_freqs [S_img,64], txt_freqs [T,64])."""
    vid = []
    max_vid_index = 0
    for idx, (frame, height, width) in enumerate(img_shapes):
        if sampling and idx > 0 and (height, width) != tuple(img_shapes[0][1:]):
            frame_0, height_0, width_0 = img_shapes[0]
            spatial_0 = vid[0].reshape(frame_0, height_0, width_0, -1)
            h_grid, w_grid = torch.meshgrid(torch.linspace(0, height_0 - 1, height).long(),
                                            torch.linspace(0, width_0 - 1, width).long(), indexing="ij")
            sampled = spatial_0[:, h_grid, w_grid, :]
            fr = _rope_params(torch.arange(idx, idx + frame), _AXES_DIM[0])
            sampled[:, :, :, :fr.shape[-1]] = fr.view(frame, 1, 1, -1).expand(frame, height, width, -1)
            vid.append(sampled.reshape(frame * height * width, -1).clone())
            max_vid_index = max(height // 2, width // 2, max_vid_index)
            continue
        fr = _rope_params(torch.arange(idx, idx + frame), _AXES_DIM[0])  # pos_freqs[0][idx:idx+frame]
        h_idx = torch.cat([torch.arange(-(height - height // 2), 0), torch.arange(0, height // 2)])
        w_idx = torch.cat([torch.arange(-(width - width // 2), 0), torch.arange(0, width // 2)])
        fh = _rope_params(h_idx, _AXES_DIM[1])
        fw = _rope_params(w_idx, _AXES_DIM[2])
        freqs = torch.cat([
            fr.view(frame, 1, 1, -1).expand(frame, height, width, -1),
            fh.view(1, height, 1, -1).expand(frame, height, width, -1),
            fw.view(1, 1, width, -1).expand(frame, height, width, -1),
        ], dim=-1).reshape(frame * height * width, -1)
        vid.append(freqs)
        max_vid_index = max(height // 2, width // 2, max_vid_index)
    t_idx = torch.arange(max_vid_index, max_vid_index + txt_len)
    txt = torch.cat([_rope_params(t_idx, d) for d in _AXES_DIM], dim=1)
    return torch.cat(vid, dim=0).contiguous(), txt.contiguous()


def apply_rope(x: torch.Tensor, freqs_cis: torch.Tensor) -> torch.Tensor:
    # qwen_image_dit.py:51-57 ; x [B,H,S,D], freqs [S,D/2] complex
    x_rotated = torch.view_as_complex(x.float().reshape(*x.shape[:-1], -1, 2))
    x_out = torch.view_as_real(x_rotated * freqs_cis).flatten(3)
    return x_out.type_as(x)


def patchify(latents: torch.Tensor) -> torch.Tensor:
    # "B C (H P) (W Q) -> B (H W) (C P Q)", P=Q=2 (qwen_image_physical.py:1344)
    B, C, H2, W2 = latents.shape
    H, W = H2 // 2, W2 // 2
    x = latents.reshape(B, C, H, 2, W, 2).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B, H * W, C * 4)


def unpatchify(image: torch.Tensor, H: int, W: int) -> torch.Tensor:
    # "B (H W) (C P Q) -> B C (H P) (W Q)" (qwen_image_physical.py:1402)
    B = image.shape[0]
    x = image.reshape(B, H, W, 16, 2, 2).permute(0, 3, 1, 4, 2, 5)
    return x.reshape(B, 16, H * 2, W * 2)


# ======================================================================================
# DiT block   (models/qwen_image_dit.py:247-401)
# ======================================================================================
def _modulate(x, mod_params):
    # qwen_image_dit.py:355-357
    shift, scale, gate = mod_params.chunk(3, dim=-1)
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1), gate.unsqueeze(1)


def _heads(x: torch.Tensor, h: int = 24) -> torch.Tensor:
    B, S, HD = x.shape
    return x.reshape(B, S, h, HD // h).permute(0, 2, 1, 3)


def flash_attention_fp8(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, p_dtype=torch.float8_e4m3fn, kv_tile: Optional[int] = None,
                        lazy_tau_log2: float = 0.0, lazy_sum_limit: Optional[float] = None, row_sum_quantised: bool = False) -> torch.Tensor:
    """qwen_image_flash_attention(enable_fp8_attention=True), qwen_image_dit.py:24-35: q, k, v [B, H, S, D] bf16 are divided by their
    global standard deviations (torch.std: unbiased, over the whole tensor, a bf16 scalar), cast to float8_e4m3fn, handed to
    FlashAttention-3 with softmax_scale = q_std * k_std / sqrt(D), and the output (bf16) is multiplied by v_std.

    PARITY UNPINNED for the kernel itself: flash_attn_interface (FA3, Hopper) cannot run here or in the reference's CPU path, so
    what it does INSIDE -- in particular how it quantises P for the second matmul -- is restated from its published design: both
    matmuls on e4m3 operands with fp32 accumulation, softmax statistics in fp32, P cast to e4m3 (p_dtype=None: P kept in fp32,
    the upper bound on what any such kernel can reach).  kv_tile=None quantises P = exp(s - row max) against the FINAL row max; a
    flash kernel cannot know that yet -- it quantises each KV tile's P against the RUNNING max and rescales the accumulator when the
    max moves -- so kv_tile=n restates that online form with n-key tiles (FA3's own tile size is not observable; the library's is
    64).  lazy_tau_log2 = T > 0 restates the LAZY online form (the library's default kernel; FlashAttention-4 publishes the same
    trick with T = 8): the reference value m is raised to a tile's maximum only when that exceeds m by more than a factor 2^T,
    so P = exp(s - m) may reach 2^T (256 for T = 8, below e4m3's 448) and the accumulator is rescaled on a few tiles only.
    lazy_sum_limit = L restates the library's max-free fast path (attn_fp8_variant 2): a row keeps its reference m as long as its P of
    the tile, exp(s - m), sum to L at most (L = 448, the largest e4m3: then no single P saturates); a row over the limit (always on
    the first tile: m = -inf) moves m to the tile's maximum and recomputes its P.
    row_sum_quantised=True restates attn_fp8_variant 4: the row sum l adds up the e4m3 P the numerator multiplies (the library takes it
    from the matrix pipe, ones . P) instead of the fp32 exp values -- not the published FlashAttention-3 form, the library's opt-in.
    All these forms differ by e4m3 rounding noise of P only (same size, different rounding points).  Everything outside the kernel
    (the three std, the two casts, the scale, the output product and its roundings) is the reference's own arithmetic.
    -> [B, H, S, D] bf16."""
    origin = q.dtype
    q_std, k_std, v_std = q.std(), k.std(), v.std()
    q8, k8, v8 = (q / q_std).to(torch.float8_e4m3fn), (k / k_std).to(torch.float8_e4m3fn), (v / v_std).to(torch.float8_e4m3fn)
    scale = float(q_std * k_std / math.sqrt(q.size(-1)))                 # a bf16 tensor product, then the python float FA3 receives
    s = torch.matmul(q8.float(), k8.float().transpose(-1, -2)) * scale   # e4m3 products are exact in fp32; fp32 accumulation
    pr = torch.softmax(s, dim=-1)                                        # FA3: the row max gives P <= 1 before the cast
    if p_dtype is not None and kv_tile:
        S_ = s.shape[-1]
        m = torch.full(s.shape[:-1] + (1,), float("-inf"))
        l = torch.zeros(s.shape[:-1] + (1,))
        acc = torch.zeros(s.shape[:-1] + (v8.shape[-1],))
        vf = v8.float()
        for t0 in range(0, S_, kv_tile):
            st = s[..., t0:t0 + kv_tile]
            m_tile = st.amax(dim=-1, keepdim=True)
            # T = 0: the running maximum.  (m = -inf on the first tile: always raised)
            if lazy_sum_limit is not None:
                over = ~(torch.exp(st - m).sum(dim=-1, keepdim=True) <= lazy_sum_limit)          # NaN / inf (m = -inf) count as over
                m_new = torch.where(over, m_tile, m)
            else:
                m_new = torch.where(m_tile - m > lazy_tau_log2 * math.log(2.0), m_tile, m)
            alpha = torch.exp(m - m_new)
            e = torch.exp(st - m_new)
            l = l * alpha + (e.to(p_dtype).float() if row_sum_quantised else e).sum(dim=-1, keepdim=True)
            acc = acc * alpha + torch.matmul(e.to(p_dtype).float(), vf[..., t0:t0 + kv_tile, :])
            m = m_new
        x = acc / l
    elif p_dtype is not None:
        # the kernel quantises the UN-normalised exp(s - max) and divides the accumulated output by the fp32 row sum afterwards
        m = s.amax(dim=-1, keepdim=True)
        e = torch.exp(s - m)
        x = torch.matmul(e.to(p_dtype).float(), v8.float()) / e.sum(dim=-1, keepdim=True)
    else:
        x = torch.matmul(pr, v8.float())
    return x.to(origin) * v_std


def joint_attention(sd: SD, p: str, image, text, rope, attention_mask=None, fp8_attention: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    # QwenDoubleStreamAttention.forward, qwen_image_dit.py:274-316
    a = p + "attn."
    img_q = _linear(sd, a + "to_q", image)
    img_k = _linear(sd, a + "to_k", image)
    img_v = _linear(sd, a + "to_v", image)
    txt_q = _linear(sd, a + "add_q_proj", text)
    txt_k = _linear(sd, a + "add_k_proj", text)
    txt_v = _linear(sd, a + "add_v_proj", text)
    seq_txt = txt_q.shape[1]
    img_q, img_k, img_v = _heads(img_q), _heads(img_k), _heads(img_v)
    txt_q, txt_k, txt_v = _heads(txt_q), _heads(txt_k), _heads(txt_v)
    img_q, img_k = rmsnorm(img_q, sd[a + "norm_q.weight"]), rmsnorm(img_k, sd[a + "norm_k.weight"])
    txt_q, txt_k = rmsnorm(txt_q, sd[a + "norm_added_q.weight"]), rmsnorm(txt_k, sd[a + "norm_added_k.weight"])
    img_freqs, txt_freqs = rope
    img_q, img_k = apply_rope(img_q, img_freqs), apply_rope(img_k, img_freqs)
    txt_q, txt_k = apply_rope(txt_q, txt_freqs), apply_rope(txt_k, txt_freqs)
    q = torch.cat([txt_q, img_q], dim=2)
    k = torch.cat([txt_k, img_k], dim=2)
    v = torch.cat([txt_v, img_v], dim=2)
    # qwen_image_flash_attention, SDPA branch (:37-38); CPU has no FA3.  fp8_attention: the FA3 e4m3 branch (:24-35) as restated above
    if fp8_attention and attention_mask is None:
        x = flash_attention_fp8(q, k, v)
    else:
        x = F.scaled_dot_product_attention(q, k, v, attn_mask=attention_mask)
    B, H, S, D = x.shape
    x = x.permute(0, 2, 1, 3).reshape(B, S, H * D).to(q.dtype)
    txt_o, img_o = x[:, :seq_txt, :], x[:, seq_txt:, :]
    img_o = _linear(sd, a + "to_out.0", img_o)
    txt_o = _linear(sd, a + "to_add_out", txt_o)
    return img_o, txt_o


def feed_forward(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    # QwenFeedForward + ApproximateGELU, qwen_image_dit.py:42-49,228-245
    x = _linear(sd, p + "net.0.proj", x)
    x = x * torch.sigmoid(1.702 * x)
    return _linear(sd, p + "net.2", x)


def block_forward(sd: SD, i: int, image, text, temb, rope, attention_mask=None, fp8_attention: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    # QwenImageTransformerBlock.forward, qwen_image_dit.py:359-401.  Returns (text, image).
    p = f"transformer_blocks.{i}."
    D = image.shape[-1]
    st = F.silu(temb)
    img_mod = _linear(sd, p + "img_mod.1", st)
    txt_mod = _linear(sd, p + "txt_mod.1", st)
    img_mod_attn, img_mod_mlp = img_mod.chunk(2, dim=-1)
    txt_mod_attn, txt_mod_mlp = txt_mod.chunk(2, dim=-1)

    img_m, img_gate = _modulate(F.layer_norm(image, (D,), eps=1e-6), img_mod_attn)
    txt_m, txt_gate = _modulate(F.layer_norm(text, (D,), eps=1e-6), txt_mod_attn)
    img_attn, txt_attn = joint_attention(sd, p, img_m, txt_m, rope, attention_mask, fp8_attention)
    image = image + img_gate * img_attn
    text = text + txt_gate * txt_attn

    img_m2, img_gate2 = _modulate(F.layer_norm(image, (D,), eps=1e-6), img_mod_mlp)
    txt_m2, txt_gate2 = _modulate(F.layer_norm(text, (D,), eps=1e-6), txt_mod_mlp)
    image = image + img_gate2 * feed_forward(sd, p + "img_mlp.", img_m2)
    text = text + txt_gate2 * feed_forward(sd, p + "txt_mlp.", txt_m2)
    return text, image


# ======================================================================================
# adapter   (pipelines/helpers.py:123-164)
# ======================================================================================
def adapter_alpha(timestep: torch.Tensor, t_min: float, t_max: float) -> torch.Tensor:
    # helpers.py:142-150
    alpha = (timestep - t_min) / (t_max - t_min + 1e-6)
    return alpha.clamp(0.0, 1.0).view(-1, 1, 1)


def adapter_forward(ad: SD, x: torch.Tensor, timestep: torch.Tensor, t_min: float, t_max: float):
    # VisualThinkingDualAdapter.forward, helpers.py:152-164
    def head(n):
        h = F.linear(x, ad[n + ".0.weight"], ad[n + ".0.bias"])
        h = F.gelu(h)
        return F.linear(h, ad[n + ".2.weight"], ad[n + ".2.bias"])
    pred_dino, pred_vae = head("head_dino"), head("head_vae")
    alpha_view = adapter_alpha(timestep, t_min, t_max).type_as(pred_dino)
    mixed = alpha_view * pred_dino + (1 - alpha_view) * pred_vae
    return mixed, pred_dino, pred_vae


# ---- training-time prior (QwenImageUnit_PhysicalVisualEmbedder.process, :1060-1118, after DINOv2 / the VAE have run) ------------------
def perceiver_resampler(sd: SD, p: str, x: torch.Tensor, heads: int = 8) -> torch.Tensor:
    """PerceiverResampler.forward (pipelines/helpers.py:92-109) with PerceiverAttention (:21-64) and FeedForward (:8-19); x [1, n, dim]."""
    dim = x.shape[-1]
    n = x.shape[1]
    latents = sd[p + "latents"].unsqueeze(0)
    x = x + sd[p + "pos_emb.weight"][torch.arange(n)]
    i = 0
    while f"{p}layers.{i}.0.to_q.weight" in sd:
        a, f = f"{p}layers.{i}.0.", f"{p}layers.{i}.1.net."
        xm = F.layer_norm(x, (dim,), sd[a + "norm_media.weight"], sd[a + "norm_media.bias"])
        lt = F.layer_norm(latents, (dim,), sd[a + "norm_latents.weight"], sd[a + "norm_latents.bias"])
        q = F.linear(lt, sd[a + "to_q.weight"])
        k, v = F.linear(torch.cat((xm, lt), dim=1), sd[a + "to_kv.weight"]).chunk(2, dim=-1)
        split = lambda t: t.view(1, t.shape[1], heads, -1).permute(0, 2, 1, 3)
        q, k, v = split(q), split(k), split(v)
        dots = torch.einsum("b h i d, b h j d -> b h i j", q, k) * (q.shape[-1] ** -0.5)
        dots = dots - dots.amax(dim=-1, keepdim=True).detach()
        attn = dots.softmax(dim=-1)
        out = torch.einsum("b h i j, b h j d -> b h i d", attn, v)
        out = out.permute(0, 2, 1, 3).reshape(1, out.shape[2], -1)
        latents = latents + F.linear(out, sd[a + "to_out.weight"])
        h = F.layer_norm(latents, (dim,), sd[f + "0.weight"], sd[f + "0.bias"])
        h = F.gelu(F.linear(h, sd[f + "1.weight"], sd[f + "1.bias"]))
        latents = latents + F.linear(h, sd[f + "3.weight"], sd[f + "3.bias"])
        i += 1
    return F.layer_norm(latents, (dim,), sd[p + "norm.weight"], sd[p + "norm.bias"])


def resampler_adapter(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    # VisualThinkingAdapter.forward, helpers.py:111-120
    return F.linear(F.gelu(F.linear(x, sd[p + "net.0.weight"], sd[p + "net.0.bias"])), sd[p + "net.2.weight"], sd[p + "net.2.bias"])


def dinov2_features(sd: SD, pixel_values: torch.Tensor, num_heads: int, patch: int = 14, eps: float = 1e-6,
                    num_register_tokens: int = 4) -> torch.Tensor:
    """Dinov2withNorm.forward (pipelines/dinov2.py:27-34): transformers' Dinov2WithRegistersModel -- patch embedding, CLS token,
    position table (resampled when the patch grid differs from the stored one: `interpolate_pos_encoding`, bicubic + antialias in
    fp32), register tokens behind the CLS token, `num_layers` pre-norm blocks (q / k / v Linears, scaled_dot_product_attention,
    output Linear, LayerScale, residual; LayerNorm, Linear - GELU(erf) - Linear, LayerScale, residual) -- then the final LayerNorm
    WITHOUT affine (normalize=True, :21-24) and the 1 + 4 CLS / register tokens dropped (:30-31).  The algorithm lives in a
    third-party dependency of the reference (transformers 5.x, `models/dinov2_with_registers/modeling_dinov2_with_registers.py`);
    pinned on the outputs of the reference's own class, fixture G19.  sd: transformers' key names."""
    x = pixel_values.to(sd["embeddings.patch_embeddings.projection.weight"].dtype)
    B, _, Hh, Ww = x.shape
    emb = F.conv2d(x, sd["embeddings.patch_embeddings.projection.weight"], sd["embeddings.patch_embeddings.projection.bias"],
                   stride=patch).flatten(2).transpose(1, 2)
    hidden = emb.shape[-1]
    emb = torch.cat((sd["embeddings.cls_token"].expand(B, -1, -1), emb), dim=1)
    pos = sd["embeddings.position_embeddings"]
    n, n0 = emb.shape[1] - 1, pos.shape[1] - 1
    if not (n == n0 and Hh == Ww):
        s0 = int(n0 ** 0.5)
        grid = pos[:, 1:].reshape(1, s0, s0, hidden).permute(0, 3, 1, 2)
        grid = F.interpolate(grid.to(torch.float32), size=(Hh // patch, Ww // patch), mode="bicubic", align_corners=False,
                             antialias=True).to(pos.dtype)
        pos = torch.cat((pos[:, 0].unsqueeze(0), grid.permute(0, 2, 3, 1).reshape(1, -1, hidden)), dim=1)
    h = emb + pos
    h = torch.cat((h[:, :1], sd["embeddings.register_tokens"].expand(B, -1, -1), h[:, 1:]), dim=1)
    i = 0
    while f"encoder.layer.{i}.norm1.weight" in sd:
        L = f"encoder.layer.{i}."
        y = F.layer_norm(h, (hidden,), sd[L + "norm1.weight"], sd[L + "norm1.bias"], eps)
        hd = lambda t: t.view(B, -1, num_heads, hidden // num_heads).transpose(1, 2)
        k = hd(F.linear(y, sd[L + "attention.attention.key.weight"], sd[L + "attention.attention.key.bias"]))
        v = hd(F.linear(y, sd[L + "attention.attention.value.weight"], sd[L + "attention.attention.value.bias"]))
        q = hd(F.linear(y, sd[L + "attention.attention.query.weight"], sd[L + "attention.attention.query.bias"]))
        a = F.scaled_dot_product_attention(q, k, v, scale=(hidden // num_heads) ** -0.5).transpose(1, 2).reshape(B, -1, hidden)
        a = F.linear(a, sd[L + "attention.output.dense.weight"], sd[L + "attention.output.dense.bias"])
        h = a * sd[L + "layer_scale1.lambda1"] + h
        y = F.layer_norm(h, (hidden,), sd[L + "norm2.weight"], sd[L + "norm2.bias"], eps)
        y = F.linear(F.gelu(F.linear(y, sd[L + "mlp.fc1.weight"], sd[L + "mlp.fc1.bias"])), sd[L + "mlp.fc2.weight"], sd[L + "mlp.fc2.bias"])
        h = y * sd[L + "layer_scale2.lambda1"] + h
        i += 1
    h = F.layer_norm(h, (hidden,), None, None, eps)
    return h[:, 1 + num_register_tokens:]


def visual_prior(sd: SD, dino_middle: torch.Tensor, dino_source: torch.Tensor, lat_middle: torch.Tensor, lat_source: torch.Tensor):
    """QwenImageUnit_PhysicalVisualEmbedder.process (:1071-1118) from the DINOv2 patch features of the key frames [B, L, 768] / of the
    source image [1, L, 768] and their VAE latents [B, 16, h, w] / [1, 16, h, w] on: frame-index embedding, frames concatenated
    along the sequence, resampler + adapter, middle - source.  -> (pseudo_special_emb_dino, pseudo_special_emb_vae) [1, 64, 3584]."""
    B = dino_middle.shape[0]
    dm = dino_middle + sd["dino_time_embed.weight"][torch.arange(B)].unsqueeze(1)
    dm = dm.reshape(1, -1, dm.shape[-1])
    d_mid = resampler_adapter(sd, "dino_resampler_adapter.", perceiver_resampler(sd, "dino_resampler.", dm))
    d_src = resampler_adapter(sd, "dino_resampler_adapter.", perceiver_resampler(sd, "dino_resampler.", dino_source.reshape(1, -1, 768)))
    pat = lambda z: z.reshape(z.shape[0], 16, z.shape[2] // 2, 2, z.shape[3] // 2, 2).permute(0, 2, 4, 1, 3, 5).reshape(z.shape[0], -1, 64)
    vm = pat(lat_middle) + sd["vae_time_embed.weight"][torch.arange(B)].unsqueeze(1)
    vm = vm.reshape(1, -1, 64)
    v_mid = resampler_adapter(sd, "vae_resampler_adapter.", perceiver_resampler(sd, "vae_resampler.", vm))
    v_src = resampler_adapter(sd, "vae_resampler_adapter.", perceiver_resampler(sd, "vae_resampler.", pat(lat_source).reshape(1, -1, 64)))
    return d_mid - d_src, v_mid - v_src


def adapter_get_loss(pred_dino, pred_vae, gt_dino, gt_vae, timestep, t_min: float, t_max: float, epsilon: float = 0.1):
    # VisualThinkingDualAdapter.get_loss, helpers.py:166-183
    alpha = adapter_alpha(timestep, t_min, t_max).type_as(pred_dino)
    loss_dino = F.mse_loss(pred_dino, gt_dino, reduction="none").mean(dim=[1, 2])
    loss_vae = F.mse_loss(pred_vae, gt_vae, reduction="none").mean(dim=[1, 2])
    w = alpha.squeeze()
    weight_dino = w + epsilon
    weight_vae = (1 - w) + epsilon
    total_weight = weight_dino + weight_vae
    weight_dino = weight_dino / total_weight
    weight_vae = weight_vae / total_weight
    return (weight_dino * loss_dino + weight_vae * loss_vae).mean()


# ======================================================================================
# model_fn   (pipelines/qwen_image_physical.py:1302-1403)
# ======================================================================================
def num_layers_of(sd: SD) -> int:
    n = 0
    while f"transformer_blocks.{n}.img_mod.1.weight" in sd:
        n += 1
    return n


# ---- EliGen entity control (QwenImageDiT.process_entity_masks, models/qwen_image_dit.py:433-498) --------------------------------
def process_entity_masks(sd: SD, latents, prompt_emb, entity_prompt_emb, entity_masks, height, width, image_tokens, img_shapes):
    """-> (text [1, sum(T_i) + T, 3072] in the order entity prompts ..., global prompt; (img_freqs, txt_freqs) with the text
    positions RESTARTING for every prompt; additive attention mask [1, 1, S, S] over the joint order [prompts ..., image]:
    a prompt sees the image tokens of its region (any mask pixel in the token's 2 x 2 patch; the same region in every image of
    the sequence) and itself, the global prompt sees every image token, image tokens see each other.)
    `entity_masks` [1, N, 1, H/8, W/8] with values in {0, 1} (QwenImageUnit_EntityControl.preprocess_masks, :1157-1163);
    every prompt_emb_mask is all ones at B = 1 (get_prompt_emb pads to the longest of ONE prompt), so lengths = shapes."""
    all_emb = list(entity_prompt_emb) + [prompt_emb]
    text = torch.cat([_linear(sd, "txt_in", rmsnorm(e, sd["txt_norm.weight"])) for e in all_emb], dim=1)
    seq_lens = [e.shape[1] for e in all_emb]
    img_f, txt_f = rope_tables(img_shapes, seq_lens[-1])
    txt_f = torch.cat([rope_tables(img_shapes, n)[1] for n in seq_lens[:-1]] + [txt_f], dim=0)
    masks = entity_masks.repeat(1, 1, latents.shape[1], 1, 1)
    masks = [masks[:, i] for i in range(masks.shape[1])]
    masks.append(torch.ones_like(masks[0]))
    N = len(masks)
    n_img = image_tokens.shape[1]
    total = sum(seq_lens) + n_img
    allowed = torch.ones((1, total, total), dtype=torch.bool)
    image_start = sum(seq_lens)
    cum = [0]
    for n in seq_lens:
        cum.append(cum[-1] + n)
    for i in range(N):
        im = patchify(masks[i]).sum(dim=-1) > 0                              # [1, S0]
        im = im.unsqueeze(1).repeat(1, seq_lens[i], 1).repeat(1, 1, n_img // im.shape[-1])
        allowed[:, cum[i]:cum[i + 1], image_start:] = im
        allowed[:, image_start:, cum[i]:cum[i + 1]] = im.transpose(1, 2)
    for i in range(N):
        for j in range(N):
            if i != j:
                allowed[:, cum[i]:cum[i + 1], cum[j]:cum[j + 1]] = False
    mask = allowed.float()
    mask[mask == 0] = float("-inf")
    mask[mask == 1] = 0
    return text, (img_f, txt_f), mask.to(latents.dtype).unsqueeze(1)


# ---- block-wise ControlNet (models/qwen_image_controlnet.py, pipelines/qwen_image_physical.py:157-180, :1373-1396) ----------
def controlnet_preprocess(cs: SD, conditioning_latents: torch.Tensor) -> torch.Tensor:
    """QwenImageBlockwiseMultiControlNet.preprocess for one input (:164-170): patchify ("B C (H P) (W Q) -> B (H W) (C P Q)",
    C = 16, or 17 with the inpaint mask channel) + QwenImageBlockWiseControlNet.img_in."""
    B, Cc, H, W = conditioning_latents.shape
    x = conditioning_latents.reshape(B, Cc, H // 2, 2, W // 2, 2).permute(0, 2, 4, 1, 3, 5).reshape(B, (H // 2) * (W // 2), Cc * 4)
    return F.linear(x, cs["img_in.weight"], cs["img_in.bias"])


def controlnet_block(cs: SD, block_id: int, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """BlockWiseControlBlock.forward (qwen_image_controlnet.py:16-21): rms(x) + rms(y) -> Linear -> exact-erf GELU -> Linear."""
    p = f"controlnet_blocks.{block_id}."
    x, y = rmsnorm(x, cs[p + "x_rms.weight"]), rmsnorm(y, cs[p + "y_rms.weight"])
    x = F.linear(x + y, cs[p + "input_proj.weight"], cs[p + "input_proj.bias"])
    x = F.gelu(x)
    return F.linear(x, cs[p + "output_proj.weight"], cs[p + "output_proj.bias"])


def controlnet_active(progress_id: int, num_inference_steps: int, start: float, end: float) -> bool:
    """the gate of QwenImageBlockwiseMultiControlNet.blockwise_forward (:175-177); ControlNetInput defaults start=1, end=0"""
    progress = (num_inference_steps - 1 - progress_id) / max(num_inference_steps - 1, 1)
    return not (progress > start + 1e-4 or progress < end - 1e-4)


def controlnet_mask_on_image(image_u8_hwc, mask_u8_hwc, dtype=torch.bfloat16):
    """QwenImageUnit_BlockwiseControlNet.apply_controlnet_mask_on_image (:1218-1222) for a mask already resized to the image:
    pixels whose mask mean (taken over all three channels of preprocess_image's [-1,1] values) is > 0 become black."""
    import numpy as np
    m = preprocess_image(mask_u8_hwc, dtype).mean(dim=[0, 1])
    out = np.array(image_u8_hwc).copy()
    out[(m > 0).numpy()] = 0
    return out


def controlnet_mask_on_latents(latents: torch.Tensor, mask_u8_hwc, dtype=torch.bfloat16) -> torch.Tensor:
    """apply_controlnet_mask_on_latents (:1211-1216): channel 17 = 1 - nearest-resized((mask + 1) / 2 averaged over RGB)."""
    m = (preprocess_image(mask_u8_hwc, dtype) + 1) / 2
    m = m.mean(dim=1, keepdim=True)
    m = 1 - torch.nn.functional.interpolate(m, size=latents.shape[-2:])
    return torch.concat([latents, m], dim=1)


def model_fn(sd: SD, ad: Optional[SD], latents: torch.Tensor, timestep: torch.Tensor,
             prompt_emb: torch.Tensor, special_token_mask: Optional[torch.Tensor],
             height: int, width: int, edit_latents=None,
             t_min: float = 20.0, t_max: float = 1000.0, controlnets=None, progress_id: int = 0,
             num_inference_steps: int = 1, entity_prompt_emb=None, entity_masks=None, capture: Optional[dict] = None,
             edit_rope_interpolation: bool = False, pseudo_special_emb=None, enable_fp8_attention: bool = False) -> torch.Tensor:
    """One DiT forward at inference (is_train=False).  MUTATES `prompt_emb` IN PLACE on the
    special-token rows exactly as the reference does (:1336).
    `controlnets`: list of dicts {"sd": controlnet state dict, "conditioning": latents [1,16|17,h8,w8], "scale", "start", "end"}
    (one per ControlNetInput; the unit that VAE-encodes the control image is outside this function)."""
    if special_token_mask is not None:
        special = prompt_emb[special_token_mask].view(prompt_emb.shape[0], -1, prompt_emb.size(-1))
        special, dino_pred, vae_pred = adapter_forward(ad, special, timestep, t_min, t_max)
        prompt_emb[special_token_mask] = special.reshape(-1, prompt_emb.size(-1))
        if pseudo_special_emb is not None and capture is not None:      # is_train=True (:1337-1338); (gt_dino, gt_vae)
            capture["special_token_loss"] = adapter_get_loss(dino_pred, vae_pred, pseudo_special_emb[0], pseudo_special_emb[1],
                                                             timestep, t_min, t_max)

    img_shapes = [(latents.shape[0], latents.shape[2] // 2, latents.shape[3] // 2)]
    T = prompt_emb.shape[1]
    timestep = timestep / 1000
    image = patchify(latents)
    image_seq_len = image.shape[1]
    if edit_latents is not None:
        edit_list = edit_latents if isinstance(edit_latents, (list, tuple)) else [edit_latents]
        img_shapes += [(e.shape[0], e.shape[2] // 2, e.shape[3] // 2) for e in edit_list]
        image = torch.cat([image] + [patchify(e) for e in edit_list], dim=1)

    image = _linear(sd, "img_in", image)
    conditioning = time_text_embed(sd, timestep, image.dtype)
    attention_mask = None
    if entity_prompt_emb is not None:                      # EliGen (:1360-1364)
        text, (vid_f, txt_f), attention_mask = process_entity_masks(sd, latents, prompt_emb, entity_prompt_emb, entity_masks,
                                                                    height, width, image, img_shapes)
    else:
        text = _linear(sd, "txt_in", rmsnorm(prompt_emb, sd["txt_norm.weight"]))
        vid_f, txt_f = rope_tables(img_shapes, T, sampling=edit_rope_interpolation)      # :1367-1370

    processed = [controlnet_preprocess(c["sd"], c["conditioning"]) for c in controlnets] if controlnets else None

    for i in range(num_layers_of(sd)):
        text, image = block_forward(sd, i, image, text, conditioning, (vid_f, txt_f), attention_mask, enable_fp8_attention)
        if processed is not None:                          # :1389-1396
            image_slice = image[:, :image_seq_len].clone()
            res = 0
            for c, cond in zip(controlnets, processed):
                if not controlnet_active(progress_id, num_inference_steps, c.get("start", 1.0), c.get("end", 0.0)):
                    continue
                res = res + controlnet_block(c["sd"], i, image_slice, cond) * c.get("scale", 1.0)
            image[:, :image_seq_len] = image_slice + res

    if capture is not None:                                # tests: the residual streams after the last block
        capture["image"], capture["text"] = image.clone(), text.clone()
    # AdaLayerNorm(single=True), models/utils.py:304-309
    emb = _linear(sd, "norm_out.linear", F.silu(conditioning))
    scale, shift = emb.unsqueeze(1).chunk(2, dim=2)
    image = F.layer_norm(image, (image.shape[-1],), eps=1e-6) * (1 + scale) + shift
    image = _linear(sd, "proj_out", image)
    image = image[:, :image_seq_len]
    return unpatchify(image, height // 16, width // 16)


def denoise_loop(sd: SD, ad: Optional[SD], noise: torch.Tensor, prompt_emb_posi: torch.Tensor,
                 prompt_emb_nega: Optional[torch.Tensor], mask_posi, mask_nega,
                 height: int, width: int, num_inference_steps: int, cfg_scale: float = 4.0,
                 edit_latents=None, dtype=torch.bfloat16, controlnets=None, denoising_strength: float = 1.0,
                 input_latents=None, inpaint_mask=None) -> torch.Tensor:
    """QwenImagePhysicPipeline.__call__ lines 600 + 644-661 (loop only; prologue outputs are the
    arguments).  prompt_emb_* are cloned once here and then mutated across steps like the
    reference's `inputs_posi` / `inputs_nega` dict entries.  `noise` is the loop's first `latents` (for an image-to-image run the
    caller has applied `add_noise`); input_latents + inpaint_mask: the blend of BasePipeline.step (utils/__init__.py:146-156)."""
    tab = FlowMatchTables(num_inference_steps, dynamic_shift_len=(height // 16) * (width // 16),
                          denoising_strength=denoising_strength)
    t_min, t_max = adapter_t_range()
    latents = noise.clone()
    pe_p = prompt_emb_posi.clone()
    pe_n = prompt_emb_nega.clone() if prompt_emb_nega is not None else None
    for progress_id, timestep in enumerate(tab.timesteps):
        t = timestep.unsqueeze(0).to(dtype=dtype)
        kw = dict(controlnets=controlnets, progress_id=progress_id, num_inference_steps=num_inference_steps)
        pred = model_fn(sd, ad, latents, t, pe_p, mask_posi, height, width, edit_latents, t_min, t_max, **kw)
        if cfg_scale != 1.0:
            pred_n = model_fn(sd, ad, latents, t, pe_n, mask_nega, height, width, edit_latents, t_min, t_max, **kw)
            pred = pred_n + cfg_scale * (pred - pred_n)
        if inpaint_mask is not None:
            expected = tab.return_to_timestep(progress_id, latents, input_latents)
            pred = expected * (1 - inpaint_mask) + pred * inpaint_mask      # blend_with_mask(base, addition, mask)
        latents = tab.step(pred, progress_id, latents)
    return latents


# ======================================================================================
# LoRA merge   (lora/__init__.py:28-45)
# ======================================================================================
def lora_merge(sd: SD, lora: SD, alpha: float = 1.0, dtype=torch.bfloat16) -> int:
    """GeneralLoRALoader.load on the DiT state-dict, in place.  Returns #tensors updated."""
    n = 0
    for key in list(lora.keys()):
        if ".lora_B." not in key:
            continue
        keys = key.split(".")
        if len(keys) > keys.index("lora_B") + 2:
            keys.pop(keys.index("lora_B") + 1)
        keys.pop(keys.index("lora_B"))
        if keys[0] == "diffusion_model":
            keys.pop(0)
        keys.pop(-1)
        target = ".".join(keys) + ".weight"
        if target not in sd:
            continue
        up = lora[key].to(dtype=dtype)
        down = lora[key.replace(".lora_B.", ".lora_A.")].to(dtype=dtype)
        sd[target] = sd[target].to(dtype=dtype) + alpha * torch.mm(up, down)
        n += 1
    return n


def hot_lora_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                    loras: Sequence[Tuple[torch.Tensor, torch.Tensor]]) -> torch.Tensor:
    """AutoWrappedLinear.forward with hot-loaded LoRA (vram_management/layers.py:166-181; filled by
    load_lora(hotload=True), qwen_image_physical.py:265-272, which stores lora_A * alpha):
    out = linear(x); for each (A, B): out = out + x @ A.T @ B.T  -- every op rounds to the tensor dtype."""
    out = F.linear(x, weight, bias)
    for lora_A, lora_B in loras:
        out = out + x @ lora_A.T @ lora_B.T
    return out


class HotLoraSD(dict):
    """A DiT state-dict whose targeted Linear layers additionally carry hot-loaded LoRA pairs:
    `sd.hot[name] = [(A, B), ...]` with name like 'transformer_blocks.0.attn.to_q'."""
    hot: Dict[str, List[Tuple[torch.Tensor, torch.Tensor]]]


def attach_hot_lora(sd: SD, lora: SD, alpha: float = 1.0) -> "HotLoraSD":
    out = HotLoraSD(sd)
    # a second hot load APPENDS to the module's lists (qwen_image_physical.py:270-271)
    out.hot = {k: list(v) for k, v in getattr(sd, "hot", {}).items()}
    for key in lora:
        if ".lora_A." not in key:
            continue
        name = key.split(".lora_A.")[0]
        kb = key.replace(".lora_A.", ".lora_B.")
        if name + ".weight" in sd and kb in lora:
            out.hot.setdefault(name, []).append((lora[key] * alpha, lora[kb]))
    return out


_orig_linear = F.linear


def _linear(sd, name: str, x: torch.Tensor) -> torch.Tensor:
    """Linear `name` of the state-dict (AutoWrappedLinear.forward, vram_management/layers.py:152-186): the e4m3 path
    when the layer is stored in float8_e4m3fn, then its hot LoRA pairs if the dict carries any."""
    w, b = sd[name + ".weight"], sd.get(name + ".bias")
    hot = getattr(sd, "hot", None)
    if w.dtype == torch.float8_e4m3fn:
        out = fp8_linear(x, w, b)
        if hot and name in hot:
            for lora_A, lora_B in hot[name]:
                out = out + x @ lora_A.T @ lora_B.T
        return out
    if hot and name in hot:
        return hot_lora_linear(x, w, b, hot[name])
    return _orig_linear(x, w, b)


# ======================================================================================
# e4m3 ("FP8 computation") Linear -- not pinnable on CPU; pinned on the GPU against torch._scaled_mm (tests/test_gpu_fp8.py)
# ======================================================================================
# AutoWrappedLinear.fp8_linear (vram_management/layers.py:115-151) is reached when the DiT is STORED in
# float8_e4m3fn (ModelConfig(offload_dtype=torch.float8_e4m3fn)) and enable_vram_management(
# enable_dit_fp8_computation=True) wraps every torch.nn.Linear with computation_dtype = that stored dtype
# (pipelines/qwen_image_physical.py:440-496).  Its matmul is torch._scaled_mm, which has no CPU implementation for
# per-row scales, so the reference cannot produce golden vectors for it in the build container: this restatement has no
# CPU fixture (the HIP kernels are compared with torch._scaled_mm itself on the GPU box).  What it follows:
#   * the quantisation arithmetic literally, with the GPU semantics of `bf16_tensor / python_float`
#     (ATen BinaryDivTrueKernel: x * (1/448) in fp32, one rounding to bf16) because the reference can only run
#     this path on a GPU;
#   * for _scaled_mm its documented contract: out = ((A @ B) * scale_a * scale_b + bias).to(out_dtype) with the
#     product accumulated in fp32.  Products of e4m3 values are exact in fp32; the sum here is taken in float64 so
#     the oracle carries no accumulation-order noise of its own.
def fp8_quantize_rows(x2: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """layers.py:126-137 -> (xq float8_e4m3fn [M,K], scale_a fp32 [M,1])."""
    x_max = torch.max(torch.abs(x2), dim=-1, keepdim=True).values
    inv = torch.tensor(1.0, dtype=torch.float32) / torch.tensor(448.0, dtype=torch.float32)
    scale_a = torch.clamp((x_max.float() * inv).to(x2.dtype), min=1.0).float()
    xq = (x2 / (scale_a + 1e-8)).to(torch.float8_e4m3fn)
    return xq, scale_a


def fp8_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    shp = x.shape
    x2 = x.reshape(-1, shp[-1])
    xq, scale_a = fp8_quantize_rows(x2)
    acc = (xq.to(torch.float64) @ weight.to(torch.float64).T).float()
    y = acc * scale_a
    if bias is not None:
        y = y + bias.to(torch.bfloat16).float()
    return y.to(x.dtype).reshape(shp[:-1] + (weight.shape[0],))


def to_fp8_state_dict(sd: SD) -> SD:
    """The DiT state-dict as `load_state_dict(..., torch_dtype=float8_e4m3fn)` leaves it: every parameter cast to
    e4m3fn.  Linear weights and biases stay e4m3 (fp8_linear consumes them); everything else (RMSNorm weights) is
    what AutoWrappedModule hands to the bf16 computation: bf16(e4m3(w))."""
    linear = {k[:-len(".weight")] for k, v in sd.items() if k.endswith(".weight") and v.dim() == 2}
    out = type(sd)(sd) if isinstance(sd, HotLoraSD) else dict(sd)
    if isinstance(sd, HotLoraSD):
        out.hot = sd.hot
    for k, v in sd.items():
        q = v.to(torch.float8_e4m3fn)
        out[k] = q if k.rsplit(".", 1)[0] in linear else q.to(v.dtype)
    return out


# ======================================================================================
# VAE   (models/qwen_image_vae.py) -- single frame, feat_cache=None
# ======================================================================================
_VAE_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
             0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
_VAE_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
            3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]


# `VAE_CONV_MODE`: "3d" (default) evaluates the causal Conv3d literally like the reference --
# bit-exact with it; "2d" evaluates the mathematically identical 2-D conv with the last temporal
# tap (1/3 of the MACs; what the HIP path computes).  The two differ only by fp32 accumulation
# order inside the conv primitive (tests/test_oracle_golden.py::test_G7_vae quantifies it).
VAE_CONV_MODE = "3d"


def _conv3(vs: SD, name: str, x: torch.Tensor) -> torch.Tensor:
    """QwenImageCausalConv3d at T=1 without cache (qwen_image_vae.py:40-50): pad
    (W:1,1  H:1,1  T:2,0) with zeros then Conv3d.  The two causal pad frames are zeros, so only
    temporal tap 2 contributes -> equals a 2-D conv with weight[:, :, 2].
    1x1x1 kernels have no padding and a single tap."""
    w = vs[name + ".weight"]
    if VAE_CONV_MODE == "3d" and w.dim() == 5:
        x5 = x.unsqueeze(2)
        if w.shape[-1] == 3:
            x5 = F.pad(x5, (1, 1, 1, 1, 2, 0))
        return F.conv3d(x5, w, vs[name + ".bias"]).squeeze(2)
    w2 = w[:, :, -1] if w.dim() == 5 else w
    pad = 1 if w2.shape[-1] == 3 else 0
    return F.conv2d(x, w2, vs[name + ".bias"], padding=pad)


def _rms_norm_c(x: torch.Tensor, gamma: torch.Tensor) -> torch.Tensor:
    # QwenImageRMS_norm.forward (:76-77), channel_first, bias=False -> "+ 0.0"
    C = x.shape[1]
    return F.normalize(x, dim=1) * (C ** 0.5) * gamma.reshape(1, C, 1, 1) + 0.0


def _res_block(vs: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    # QwenImageResidualBlock.forward (:112-152), feat_cache None
    h = _conv3(vs, p + "conv_shortcut", x) if (p + "conv_shortcut.weight") in vs else x
    x = F.silu(_rms_norm_c(x, vs[p + "norm1.gamma"]))
    x = _conv3(vs, p + "conv1", x)
    x = F.silu(_rms_norm_c(x, vs[p + "norm2.gamma"]))
    x = _conv3(vs, p + "conv2", x)
    return x + h


def _attn_block(vs: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    # QwenImageAttentionBlock.forward (:173-198), T=1
    identity = x
    B, C, H, W = x.shape
    x = _rms_norm_c(x, vs[p + "norm.gamma"])
    qkv = F.conv2d(x, vs[p + "to_qkv.weight"], vs[p + "to_qkv.bias"])
    qkv = qkv.reshape(B, 1, C * 3, -1).permute(0, 1, 3, 2).contiguous()
    q, k, v = qkv.chunk(3, dim=-1)
    x = F.scaled_dot_product_attention(q, k, v)
    x = x.squeeze(1).permute(0, 2, 1).reshape(B, C, H, W)
    x = F.conv2d(x, vs[p + "proj.weight"], vs[p + "proj.bias"])
    return x + identity


def _mid_block(vs: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    x = _res_block(vs, p + "resnets.0.", x)
    x = _attn_block(vs, p + "attentions.0.", x)
    return _res_block(vs, p + "resnets.1.", x)


def vae_encode(vs: SD, x: torch.Tensor) -> torch.Tensor:
    """QwenImageVAE.encode (:706-717) + QwenImageEncoder3d.forward (:411-448) on [B,3,H,W]."""
    x = _conv3(vs, "encoder.conv_in", x)
    idx = 0
    for i in range(4):
        for _ in range(2):
            x = _res_block(vs, f"encoder.down_blocks.{idx}.", x)
            idx += 1
        if i != 3:
            # QwenImageResample downsample2d/3d without cache (:248-252, :285-287): ZeroPad2d((0,1,0,1)) + stride-2 conv
            p = f"encoder.down_blocks.{idx}.resample.1"
            x = F.conv2d(F.pad(x, (0, 1, 0, 1)), vs[p + ".weight"], vs[p + ".bias"], stride=2)
            idx += 1
    x = _mid_block(vs, "encoder.mid_block.", x)
    x = F.silu(_rms_norm_c(x, vs["encoder.norm_out.gamma"]))
    x = _conv3(vs, "encoder.conv_out", x)
    x = _conv3(vs, "quant_conv", x)
    x = x[:, :16]
    mean = torch.tensor(_VAE_MEAN).view(1, 16, 1, 1).to(dtype=x.dtype)
    std = (1 / torch.tensor(_VAE_STD).view(1, 16, 1, 1)).to(dtype=x.dtype)
    return (x - mean) * std


def vae_decode(vs: SD, x: torch.Tensor) -> torch.Tensor:
    """QwenImageVAE.decode (:719-729) + QwenImageDecoder3d.forward (:601-636) on [B,16,h,w]."""
    mean = torch.tensor(_VAE_MEAN).view(1, 16, 1, 1).to(dtype=x.dtype)
    std = (1 / torch.tensor(_VAE_STD).view(1, 16, 1, 1)).to(dtype=x.dtype)
    x = x / std + mean
    x = _conv3(vs, "post_quant_conv", x)
    x = _conv3(vs, "decoder.conv_in", x)
    x = _mid_block(vs, "decoder.mid_block.", x)
    for i in range(4):
        for j in range(3):
            x = _res_block(vs, f"decoder.up_blocks.{i}.resnets.{j}.", x)
        if i != 3:
            # QwenImageUpsample: nearest-exact 2x computed in fp32 then cast back (:213-214), + Conv2d(C, C/2, 3, pad 1)
            x = F.interpolate(x.float(), scale_factor=(2.0, 2.0), mode="nearest-exact").type_as(x)
            p = f"decoder.up_blocks.{i}.upsamplers.0.resample.1"
            x = F.conv2d(x, vs[p + ".weight"], vs[p + ".bias"], padding=1)
    x = F.silu(_rms_norm_c(x, vs["decoder.norm_out.gamma"]))
    return _conv3(vs, "decoder.conv_out", x)


# ======================================================================================
# image <-> tensor   (utils/__init__.py:60-83)
# ======================================================================================
def preprocess_image(img_u8_hwc, dtype=torch.bfloat16, min_value=-1, max_value=1) -> torch.Tensor:
    import numpy as np
    image = torch.Tensor(np.array(img_u8_hwc, dtype=np.float32))
    image = image.to(dtype=dtype)
    image = image * ((max_value - min_value) / 255) + min_value
    return image.permute(2, 0, 1).unsqueeze(0)


def inpaint_mask_plane(mask_rgb_u8_hwc, dtype=torch.bfloat16) -> torch.Tensor:
    """QwenImageUnit_Inpaint.process (:720-729) without the optional torchvision blur, for a mask already converted to RGB and
    resized to (W/8, H/8): [0, 1] values, mean over the three channels -> [1, 1, H/8, W/8]."""
    return preprocess_image(mask_rgb_u8_hwc, dtype, min_value=0, max_value=1).mean(dim=1, keepdim=True)


def vae_output_to_u8(vae_output: torch.Tensor) -> torch.Tensor:
    # reduce "B C H W -> H W C" mean, scale, clip, truncate to uint8
    x = vae_output.mean(dim=0).permute(1, 2, 0)
    image = ((x - (-1)) * (255 / (1 - (-1)))).clip(0, 255)
    return image.to(device="cpu", dtype=torch.uint8)
