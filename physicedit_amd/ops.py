"""torch-tensor wrappers over the granular C-ABI operators (one HIP kernel launch each).

Tensors must be contiguous bf16 CUDA(HIP) tensors unless stated; nothing here computes on the
host.  These are what the `-m gpu` parity tests call, and what `physicedit_amd.dit` is made of.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import check, lib, stream_ptr

BF = torch.bfloat16
EPI = {"bias": 0, "gelu_sigmoid": 1, "gelu_erf": 2, "gate_res": 3, "silu": 5}


def _chk(t: torch.Tensor, name: str, dtype=BF):
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a CUDA/HIP tensor (physicedit_amd has no CPU path)")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous tensor")
    return t


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def gemm(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: str = "bias",
         gate: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
         out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = epilogue(x @ w.T + bias); x [M,K], w [N,K] (nn.Linear layout)."""
    _chk(x, "x"), _chk(w, "w")
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K
    if out is None:
        out = torch.empty((M, N), dtype=BF, device=x.device)
    check(lib().pe_gemm_bf16(EPI[epilogue], x.data_ptr(), K, w.data_ptr(), _ptr(bias), out.data_ptr(), N, M, N, K,
                             _ptr(gate), _ptr(res), N if res is not None else 0, stream_ptr()), "pe_gemm_bf16")
    return out


def hot_lora_linear(x, w, bias, lora_a, lora_b, epilogue: str = "bias", gate=None, res=None) -> torch.Tensor:
    """AutoWrappedLinear with one hot-loaded LoRA pair (vram_management/layers.py:173-181), reference rounding:
    y1 = linear(x); t = x @ A.T; out = epilogue(y1 + t @ B.T).  lora_a [r,K], lora_b [N,r]; r is padded to 64."""
    r = lora_a.shape[0]
    rp = (r + 63) // 64 * 64
    a = torch.zeros((rp, x.shape[1]), dtype=BF, device=x.device); a[:r] = lora_a
    b = torch.zeros((w.shape[0], rp), dtype=BF, device=x.device); b[:, :r] = lora_b
    t = gemm(x, a)
    y1 = gemm(x, w, bias)
    M, N = y1.shape
    out = torch.empty_like(y1)
    check(lib().pe_gemm_bf16_pre(EPI[epilogue], t.data_ptr(), rp, b.data_ptr(), None, y1.data_ptr(), N, out.data_ptr(), N,
                                 M, N, rp, _ptr(gate), _ptr(res), N if res is not None else 0, stream_ptr()),
          "pe_gemm_bf16_pre")
    return out


F8 = torch.float8_e4m3fn


def quantize_rows_e4m3(x: torch.Tensor, Kp: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """fp8_linear's activation quantisation (vram_management/layers.py:126-137): -> (xq e4m3 [M,Kp], scale_a fp32 [M])."""
    _chk(x, "x")
    M, K = x.shape
    if Kp is None:
        Kp = (K + 127) // 128 * 128
    xq = torch.empty((M, Kp), dtype=F8, device=x.device)
    scale = torch.empty((M,), dtype=torch.float32, device=x.device)
    check(lib().pe_quantize_rows_e4m3(x.data_ptr(), K, M, K, xq.data_ptr(), Kp, scale.data_ptr(), stream_ptr()),
          "pe_quantize_rows_e4m3")
    return xq, scale


def gemm_e4m3(xq: torch.Tensor, scale_a: torch.Tensor, wq: torch.Tensor, bias: Optional[torch.Tensor] = None,
              epilogue: str = "bias", gate=None, res=None, pre=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = epilogue(bf16((xq @ wq.T) * scale_a[:,None] + bias)); xq [M,K], wq [N,K] e4m3fn, K % 128 == 0."""
    _chk(xq, "xq", F8), _chk(wq, "wq", F8), _chk(scale_a, "scale_a", torch.float32)
    M, K = xq.shape
    N = wq.shape[0]
    assert wq.shape[1] == K
    if out is None:
        out = torch.empty((M, N), dtype=BF, device=xq.device)
    check(lib().pe_gemm_e4m3(EPI[epilogue], xq.data_ptr(), K, scale_a.data_ptr(), wq.data_ptr(), _ptr(bias), _ptr(pre),
                             N if pre is not None else 0, out.data_ptr(), N, M, N, K, _ptr(gate), _ptr(res),
                             N if res is not None else 0, stream_ptr()), "pe_gemm_e4m3")
    return out


def gemm_e4m3_gelu_q8(xq: torch.Tensor, scale_a: torch.Tensor, wq: torch.Tensor, bias: Optional[torch.Tensor] = None):
    """MLP-up in e4m3 mode with the next Linear's row quantisation fused into the epilogue: -> (out bf16 [M,N], q e4m3 [M,N],
    scale fp32 [M]); (q, scale) == quantize_rows_e4m3(out) bit for bit."""
    _chk(xq, "xq", F8), _chk(wq, "wq", F8), _chk(scale_a, "scale_a", torch.float32)
    M, K = xq.shape
    N = wq.shape[0]
    out = torch.empty((M, N), dtype=BF, device=xq.device)
    q = torch.empty((M, N), dtype=F8, device=xq.device)
    sc = torch.empty((M,), dtype=torch.float32, device=xq.device)
    flags = torch.zeros((M,), dtype=torch.int32, device=xq.device)
    check(lib().pe_gemm_e4m3_gelu_q8(xq.data_ptr(), K, scale_a.data_ptr(), wq.data_ptr(), _ptr(bias), out.data_ptr(), N, q.data_ptr(),
                                     sc.data_ptr(), flags.data_ptr(), M, N, K, stream_ptr()), "pe_gemm_e4m3_gelu_q8")
    return out, q, sc, flags


def fp8_linear(x: torch.Tensor, wq: torch.Tensor, bias: Optional[torch.Tensor], epilogue: str = "bias", gate=None,
               res=None) -> torch.Tensor:
    """AutoWrappedLinear.fp8_linear: quantise rows, e4m3 GEMM.  wq [N,Kp] e4m3fn (zero padded beyond x.shape[1])."""
    xq, sa = quantize_rows_e4m3(x, wq.shape[1])
    return gemm_e4m3(xq, sa, wq, bias, epilogue, gate, res)


def s_pad_of(S: int) -> int:
    return (S + 63) // 64 * 64


def alloc_qkv(H: int, S: int, device) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    sp = s_pad_of(S)
    q = torch.zeros((H, sp, 128), dtype=BF, device=device)
    k = torch.zeros((H, sp, 128), dtype=BF, device=device)
    vt = torch.zeros((H, 128, sp), dtype=BF, device=device)
    return q, k, vt


def attn_q_prescale() -> float:
    """what Q must be multiplied by (in fp32, before its rounding to bf16) for flash_attn(q_prescaled=True): scale * log2(e) for the
    default attention kernel, 1.0 for the variants that take a plain Q"""
    return float(lib().pe_attn_q_prescale(1.0 / math.sqrt(128.0)))


def qkv_rmsnorm_rope(x, wqkv, bqkv, norm_q_w, norm_k_w, rope_cos, rope_sin, q, k, vt, seq_off: int, q_scale: float = 0.0) -> None:
    """q_scale != 0: Q is stored as bf16(rope(q) * q_scale) (pe_qkv_rmsnorm_rope_scaled; pair it with flash_attn(q_prescaled=True))"""
    _chk(x, "x"), _chk(wqkv, "wqkv")
    _chk(rope_cos, "rope_cos", torch.float32), _chk(rope_sin, "rope_sin", torch.float32)
    M, K = x.shape
    H = wqkv.shape[0] // 384
    if q_scale != 0.0:
        check(lib().pe_qkv_rmsnorm_rope_scaled(x.data_ptr(), K, wqkv.data_ptr(), _ptr(bqkv), M, H, K, norm_q_w.data_ptr(),
                                                norm_k_w.data_ptr(), rope_cos.data_ptr(), rope_sin.data_ptr(), q.data_ptr(),
                                                k.data_ptr(), vt.data_ptr(), seq_off, q.shape[1], q_scale, stream_ptr()),
              "pe_qkv_rmsnorm_rope_scaled")
        return
    check(lib().pe_qkv_rmsnorm_rope(x.data_ptr(), K, wqkv.data_ptr(), _ptr(bqkv), M, H, K, norm_q_w.data_ptr(),
                                     norm_k_w.data_ptr(), rope_cos.data_ptr(), rope_sin.data_ptr(), q.data_ptr(),
                                     k.data_ptr(), vt.data_ptr(), seq_off, q.shape[1], stream_ptr()),
          "pe_qkv_rmsnorm_rope")


_PERM16 = [(j & 3) | ((j & 4) << 1) | ((j & 8) >> 1) for j in range(16)]


def vt_positions(S_pad: int, device) -> torch.Tensor:
    """pos[s] = column of token s inside a Vt row (perm16 inside aligned 16-groups)."""
    s = torch.arange(S_pad, device=device)
    perm = torch.tensor(_PERM16, device=device)
    return (s & ~15) | perm[s & 15]


def pack_vt(v: torch.Tensor, S_pad: int) -> torch.Tensor:
    """[H,S,128] logical V -> the library's Vt layout [H,128,S_pad] (test helper)."""
    H, S, Dh = v.shape
    vt = torch.zeros((H, Dh, S_pad), dtype=v.dtype, device=v.device)
    vt[:, :, vt_positions(S_pad, v.device)[:S]] = v.transpose(1, 2)
    return vt


def unpack_vt(vt: torch.Tensor, S: int) -> torch.Tensor:
    return vt[:, :, vt_positions(vt.shape[2], vt.device)[:S]].transpose(1, 2).contiguous()


def flash_attn(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, S: int,
               out: Optional[torch.Tensor] = None, workspace: bool = True, token_words: Optional[torch.Tensor] = None,
               n_img: int = 0, q_prescaled: bool = False) -> torch.Tensor:
    """q,k [H,S_pad,128], vt [H,128,S_pad] -> [S, H*128].  token_words (int32 [S_pad], zero beyond S) + n_img: the EliGen mask,
    tokens a, b attend iff token_words[a] & token_words[b] != 0 (rows [0, n_img) are image tokens).  q_prescaled: q already
    carries attn_q_prescale() (pe_flash_attn_prescaled: the default kernel's own form)."""
    _chk(q, "q"), _chk(k, "k"), _chk(vt, "vt")
    H, sp, _ = q.shape
    if out is None:
        out = torch.empty((S, H * 128), dtype=BF, device=q.device)
    ws, nbytes = None, 0
    if workspace:
        nbytes = lib().pe_flash_attn_workspace_bytes(H, S)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=q.device)
    if token_words is not None:
        _chk(token_words, "token_words", torch.int32)
        assert token_words.numel() == sp and not q_prescaled
        check(lib().pe_flash_attn_masked(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), H, S, sp, H * 128,
                                         1.0 / math.sqrt(128.0), _ptr(ws), nbytes, token_words.data_ptr(), n_img, stream_ptr()),
              "pe_flash_attn_masked")
        return out
    fn, name = (lib().pe_flash_attn_prescaled, "pe_flash_attn_prescaled") if q_prescaled else (lib().pe_flash_attn, "pe_flash_attn")
    check(fn(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), H, S, sp, H * 128, 1.0 / math.sqrt(128.0), _ptr(ws), nbytes,
             stream_ptr()), name)
    return out


def ln_modulate(x, shift_a, scale_a, rows_a: Optional[int] = None, shift_b=None, scale_b=None, eps: float = 1e-6,
                out=None) -> torch.Tensor:
    _chk(x, "x")
    rows, dim = x.shape
    if out is None:
        out = torch.empty_like(x)
    if rows_a is None:
        rows_a = rows
    check(lib().pe_ln_modulate(x.data_ptr(), out.data_ptr(), rows, dim, rows_a, shift_a.data_ptr(), scale_a.data_ptr(),
                               _ptr(shift_b), _ptr(scale_b), eps, stream_ptr()), "pe_ln_modulate")
    return out


def ln_modulate_e4m3(x, shift_a, scale_a, rows_a: Optional[int] = None, shift_b=None, scale_b=None, eps: float = 1e-6,
                     want_bf16: bool = True):
    """ln_modulate + fp8_linear's row quantisation in one kernel -> (bf16 out or None, e4m3 [rows,dim], scale fp32 [rows])."""
    _chk(x, "x")
    rows, dim = x.shape
    out = torch.empty_like(x) if want_bf16 else None
    q = torch.empty((rows, dim), dtype=F8, device=x.device)
    sc = torch.empty((rows,), dtype=torch.float32, device=x.device)
    check(lib().pe_ln_modulate_e4m3(x.data_ptr(), _ptr(out), q.data_ptr(), sc.data_ptr(), rows, dim,
                                    rows if rows_a is None else rows_a, shift_a.data_ptr(), scale_a.data_ptr(),
                                    _ptr(shift_b), _ptr(scale_b), eps, stream_ptr()), "pe_ln_modulate_e4m3")
    return out, q, sc


def rmsnorm(x, w, eps: float = 1e-6) -> torch.Tensor:
    _chk(x, "x"), _chk(w, "w")
    out = torch.empty_like(x)
    check(lib().pe_rmsnorm(x.data_ptr(), w.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], eps, stream_ptr()),
          "pe_rmsnorm")
    return out


def gemv(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.Linear on one row: x [K] (any shape with K elements), w [N,K] -> [N]; with `res` [N]: res + linear(x), both rounded."""
    _chk(x, "x"), _chk(w, "w")
    N, K = w.shape
    assert x.numel() == K
    out = torch.empty((N,), dtype=BF, device=x.device)
    if res is not None:
        _chk(res, "res")
        assert res.numel() == N
        check(lib().pe_gemv_res_bf16(x.data_ptr(), w.data_ptr(), _ptr(bias), res.data_ptr(), out.data_ptr(), N, K, stream_ptr()),
              "pe_gemv_res_bf16")
        return out
    check(lib().pe_gemv_bf16(x.data_ptr(), w.data_ptr(), _ptr(bias), out.data_ptr(), N, K, stream_ptr()), "pe_gemv_bf16")
    return out


def gemv_swiglu(x: torch.Tensor, w_gate: torch.Tensor, w_up: torch.Tensor) -> torch.Tensor:
    """silu(w_gate @ x) * (w_up @ x) on one row, with nn.Linear / nn.SiLU roundings: x [K], w_* [N,K] -> [N]."""
    _chk(x, "x"), _chk(w_gate, "w_gate"), _chk(w_up, "w_up")
    N, K = w_gate.shape
    assert x.numel() == K and tuple(w_up.shape) == (N, K)
    out = torch.empty((N,), dtype=BF, device=x.device)
    check(lib().pe_gemv_swiglu_bf16(x.data_ptr(), w_gate.data_ptr(), w_up.data_ptr(), out.data_ptr(), N, K, stream_ptr()),
          "pe_gemv_swiglu_bf16")
    return out


def decode_qkv_rope(x, wq, bq, wk, bk, wv, bv, cos_sel, sin_sel):
    """q / k / v projections of one row + rotary embedding of the q and k heads (128-wide): -> q [Hq,128], k, v [Hkv,128]."""
    for t, n in ((x, "x"), (wq, "wq"), (wk, "wk"), (wv, "wv"), (cos_sel, "cos"), (sin_sel, "sin")):
        _chk(t, n)
    K = wq.shape[1]
    hq, hkv = wq.shape[0] // 128, wk.shape[0] // 128
    assert x.numel() == K and cos_sel.numel() == 128 and sin_sel.numel() == 128 and wv.shape == wk.shape
    q = torch.empty((hq, 128), dtype=BF, device=x.device)
    k = torch.empty((hkv, 128), dtype=BF, device=x.device)
    v = torch.empty((hkv, 128), dtype=BF, device=x.device)
    check(lib().pe_decode_qkv_rope(x.data_ptr(), wq.data_ptr(), _ptr(bq), wk.data_ptr(), _ptr(bk), wv.data_ptr(), _ptr(bv),
                                   cos_sel.data_ptr(), sin_sel.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), hq, hkv, K,
                                   stream_ptr()), "pe_decode_qkv_rope")
    return q, k, v


def decode_attention(q, k_cache, v_cache, scale: float) -> torch.Tensor:
    """one query [Hq,128] against a cache [Hkv,L,128] (GQA) -> [Hq*128]."""
    _chk(q, "q"), _chk(k_cache, "k_cache"), _chk(v_cache, "v_cache")
    hq, (hkv, L, _) = q.shape[0], k_cache.shape
    out = torch.empty((hq * 128,), dtype=BF, device=q.device)
    check(lib().pe_decode_attention(q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), out.data_ptr(), hq, hkv, L, scale,
                                    stream_ptr()), "pe_decode_attention")
    return out


def dual_rmsnorm_add(x, wx, y, wy, eps: float = 1e-6) -> torch.Tensor:
    """BlockWiseControlBlock input: bf16(RMSNorm(x; wx) + RMSNorm(y; wy)), rows of 3072."""
    _chk(x, "x"), _chk(y, "y"), _chk(wx, "wx"), _chk(wy, "wy")
    out = torch.empty_like(x)
    check(lib().pe_dual_rmsnorm_add(x.data_ptr(), wx.data_ptr(), y.data_ptr(), wy.data_ptr(), out.data_ptr(), x.shape[0],
                                    x.shape[1], eps, stream_ptr()), "pe_dual_rmsnorm_add")
    return out


def patchify(latents: torch.Tensor) -> torch.Tensor:
    _chk(latents, "latents")
    C_, H2, W2 = latents.shape[-3:]
    out = torch.empty(((H2 // 2) * (W2 // 2), C_ * 4), dtype=BF, device=latents.device)
    check(lib().pe_patchify(latents.data_ptr(), out.data_ptr(), C_, H2, W2, stream_ptr()), "pe_patchify")
    return out


def unpatchify(tokens: torch.Tensor, C_: int, H2: int, W2: int) -> torch.Tensor:
    _chk(tokens, "tokens")
    out = torch.empty((1, C_, H2, W2), dtype=BF, device=tokens.device)
    check(lib().pe_unpatchify(tokens.data_ptr(), out.data_ptr(), C_, H2, W2, stream_ptr()), "pe_unpatchify")
    return out


def cfg_euler_step(posi, nega, latents, cfg_scale: float, dsigma: float, out=None, input_latents=None, inpaint_mask=None,
                   sigma: float = 0.0) -> torch.Tensor:
    """CFG combination + Euler update; with `inpaint_mask` ([1,1,H,W] or [H,W] bf16) and `input_latents` also the inpainting blend
    of BasePipeline.step (utils/__init__.py:146-156), `sigma` = sigmas[progress_id]."""
    _chk(posi, "posi"), _chk(latents, "latents")
    if out is None:
        out = torch.empty_like(latents)
    use_cfg = 1 if (nega is not None and cfg_scale != 1.0) else 0
    if inpaint_mask is not None:
        _chk(inpaint_mask, "inpaint_mask"), _chk(input_latents, "input_latents")
        plane = inpaint_mask.numel()
        if input_latents.shape != latents.shape or plane != latents.shape[-1] * latents.shape[-2]:
            raise _lib.PeError(f"inpaint: input_latents {tuple(input_latents.shape)} / mask {tuple(inpaint_mask.shape)} do not match "
                          f"latents {tuple(latents.shape)}")
        check(lib().pe_cfg_inpaint_euler_step(posi.data_ptr(), _ptr(nega) if use_cfg else None, latents.data_ptr(),
                                              input_latents.data_ptr(), inpaint_mask.data_ptr(), out.data_ptr(), latents.numel(),
                                              plane, float(cfg_scale), use_cfg, float(sigma), float(dsigma), stream_ptr()),
              "pe_cfg_inpaint_euler_step")
        return out
    check(lib().pe_cfg_euler_step(posi.data_ptr(), _ptr(nega) if use_cfg else None, latents.data_ptr(), out.data_ptr(),
                                  latents.numel(), float(cfg_scale), use_cfg, float(dsigma), stream_ptr()),
          "pe_cfg_euler_step")
    return out


# ---- the decode step in graph-capturable form (device-side step counter; include/physicedit_amd.h "pe_decode_step_*") -----------
def _chk_i32(t, name):
    if t.dtype != torch.int32 or not t.is_cuda or not t.is_contiguous():
        raise _lib.PeError(f"{name}: need a contiguous int32 device tensor")


def decode_step_qkv(x, wq, bq, wk, bk, wv, bv, cos_table, sin_table, k_cache, v_cache, step, base_len: int, norm_w=None,
                    eps: float = 1e-6) -> torch.Tensor:
    """q / k / v projections + rotary embedding of the token at sequence position base_len + *step; k, v go into row base_len + *step
    of k_cache / v_cache [n_kv, cache_len, 128]; returns q [n_q * 128].  norm_w: the input is RMSNorm(x) * norm_w (fused)."""
    for t, n in ((x, "x"), (wq, "wq"), (wk, "wk"), (wv, "wv"), (cos_table, "cos_table"), (sin_table, "sin_table"),
                 (k_cache, "k_cache"), (v_cache, "v_cache")):
        _chk(t, n)
    _chk_i32(step, "step")
    K = x.numel()
    hq, hkv, cache_len = wq.shape[0] // 128, k_cache.shape[0], k_cache.shape[1]
    assert wk.shape[0] == hkv * 128 and k_cache.shape == v_cache.shape and k_cache.shape[2] == 128 and cos_table.shape[-1] == 128
    q = torch.empty((hq * 128,), dtype=BF, device=x.device)
    check(lib().pe_decode_step_qkv(x.data_ptr(), wq.data_ptr(), _ptr(bq), wk.data_ptr(), _ptr(bk), wv.data_ptr(), _ptr(bv),
                                   cos_table.data_ptr(), sin_table.data_ptr(), q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
                                   hq, hkv, K, step.data_ptr(), int(base_len), cache_len, _ptr(norm_w), float(eps), stream_ptr()),
          "pe_decode_step_qkv")
    return q


def decode_attention_workspace(hq: int, cache_len: int, device) -> torch.Tensor:
    """fp32 scratch of the split single-query attention (pe_decode_step_attention_split): one per decode stream"""
    return torch.empty((lib().pe_decode_attention_workspace_bytes(hq, cache_len) // 4,), dtype=torch.float32, device=device)


def decode_step_attention(q, k_cache, v_cache, step, base_len: int, scale: float, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """workspace (decode_attention_workspace): the three-launch form over 16 x as many work-groups, bit-identical to the one-launch form"""
    _chk(q, "q"), _chk(k_cache, "k_cache"), _chk(v_cache, "v_cache"), _chk_i32(step, "step")
    hq, hkv, cache_len = q.numel() // 128, k_cache.shape[0], k_cache.shape[1]
    out = torch.empty((hq * 128,), dtype=BF, device=q.device)
    if workspace is not None:
        check(lib().pe_decode_step_attention_split(q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), out.data_ptr(), hq, hkv,
                                                   step.data_ptr(), int(base_len), cache_len, float(scale), workspace.data_ptr(),
                                                   workspace.numel() * 4, stream_ptr()), "pe_decode_step_attention_split")
        return out
    check(lib().pe_decode_step_attention(q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), out.data_ptr(), hq, hkv,
                                         step.data_ptr(), int(base_len), cache_len, float(scale), stream_ptr()),
          "pe_decode_step_attention")
    return out


def decode_layer_scratch(hq: int, cache_len: int, ff: int, device) -> torch.Tensor:
    """zeroed scratch of pe_decode_layer (barrier counters + error flag, q / attention / hidden rows, the split attention's workspace);
    one per decode stream, shared by all layers (they run one after the other)"""
    n = int(lib().pe_decode_layer_scratch_bytes(hq, cache_len, ff))
    buf = torch.zeros((n + 256,), dtype=torch.uint8, device=device)
    off = (-buf.data_ptr()) % 256
    return buf[off:off + n]


def decode_layer_weights(q_w, q_b, k_w, k_b, v_w, v_b, o_w, gate_w, up_w, down_w, input_norm_w, input_norm_eps, post_norm_w, post_norm_eps):
    """-> the C struct of one decoder layer's operands (borrowed device pointers: keep the tensors alive)"""
    for t, n in ((q_w, "q_w"), (k_w, "k_w"), (v_w, "v_w"), (o_w, "o_w"), (gate_w, "gate_w"), (up_w, "up_w"), (down_w, "down_w"),
                 (input_norm_w, "input_norm_w"), (post_norm_w, "post_norm_w")):
        _chk(t, n)
    w = _lib.DecodeLayerWeights()
    w.q_w, w.q_b, w.k_w, w.k_b, w.v_w, w.v_b = q_w.data_ptr(), _ptr(q_b), k_w.data_ptr(), _ptr(k_b), v_w.data_ptr(), _ptr(v_b)
    w.o_w, w.gate_w, w.up_w, w.down_w = o_w.data_ptr(), gate_w.data_ptr(), up_w.data_ptr(), down_w.data_ptr()
    w.input_norm_w, w.post_norm_w = input_norm_w.data_ptr(), post_norm_w.data_ptr()
    w.input_norm_eps, w.post_norm_eps = float(input_norm_eps), float(post_norm_eps)
    w.n_q_heads, w.n_kv_heads, w.ff = q_w.shape[0] // 128, k_w.shape[0] // 128, gate_w.shape[0]
    return w


def decode_layer(w, x, x_out, cos_table, sin_table, k_cache, v_cache, step, base_len: int, scale: float, scratch) -> torch.Tensor:
    """one decoder layer of the decode step in ONE launch (pe_decode_layer): x [3584] -> x_out [3584]; k_cache / v_cache
    [n_kv, cache_len, 128] get row base_len + *step.  Same bits as decode_step_qkv + decode_step_attention + gemv(res) +
    gemv_swiglu_norm + gemv(res)."""
    _chk(x, "x"), _chk(x_out, "x_out"), _chk(k_cache, "k_cache"), _chk(v_cache, "v_cache"), _chk_i32(step, "step")
    check(lib().pe_decode_layer(C.byref(w), x.data_ptr(), x_out.data_ptr(), cos_table.data_ptr(), sin_table.data_ptr(), k_cache.data_ptr(),
                                v_cache.data_ptr(), step.data_ptr(), int(base_len), k_cache.shape[1], float(scale), scratch.data_ptr(),
                                scratch.numel(), stream_ptr()), "pe_decode_layer")
    return x_out


def decode_layer_error(scratch) -> int:
    """0, or 1 + the index of the grid barrier that timed out in some pe_decode_layer launch on this scratch (synchronises)"""
    return int(scratch[0:4].view(torch.int32).item())


def decode_embed(table, token) -> torch.Tensor:
    _chk(table, "table"), _chk_i32(token, "token")
    x = torch.empty((table.shape[1],), dtype=BF, device=table.device)
    check(lib().pe_decode_embed(table.data_ptr(), token.data_ptr(), x.data_ptr(), table.shape[1], table.shape[0], stream_ptr()),
          "pe_decode_embed")
    return x


def decode_argmax(logits, token, out_ids, step) -> None:
    _chk(logits, "logits"), _chk_i32(token, "token"), _chk_i32(out_ids, "out_ids"), _chk_i32(step, "step")
    check(lib().pe_decode_argmax(logits.data_ptr(), logits.numel(), token.data_ptr(), out_ids.data_ptr(), step.data_ptr(),
                                 out_ids.numel(), stream_ptr()), "pe_decode_argmax")


def gemv_norm(x, norm_w, eps: float, w, bias=None) -> torch.Tensor:
    """linear(RMSNorm(x) * norm_w) on one row of width 3584, the norm fused into the staging of x."""
    _chk(x, "x"), _chk(w, "w"), _chk(norm_w, "norm_w")
    N, K = w.shape
    assert x.numel() == K == norm_w.numel()
    out = torch.empty((N,), dtype=BF, device=x.device)
    check(lib().pe_gemv_norm_bf16(x.data_ptr(), norm_w.data_ptr(), float(eps), w.data_ptr(), _ptr(bias), out.data_ptr(), N, K,
                                  stream_ptr()), "pe_gemv_norm_bf16")
    return out


def gemv_swiglu_norm(x, norm_w, eps: float, wg, wu) -> torch.Tensor:
    _chk(x, "x"), _chk(wg, "wg"), _chk(wu, "wu"), _chk(norm_w, "norm_w")
    N, K = wg.shape
    assert x.numel() == K == norm_w.numel() and wu.shape == wg.shape
    out = torch.empty((N,), dtype=BF, device=x.device)
    check(lib().pe_gemv_swiglu_norm_bf16(x.data_ptr(), norm_w.data_ptr(), float(eps), wg.data_ptr(), wu.data_ptr(), out.data_ptr(),
                                         N, K, stream_ptr()), "pe_gemv_swiglu_norm_bf16")
    return out


# ---- small operators of the training-time prior (row f2) ------------------------------------------------------------------------
def add_(x: torch.Tensor, y: torch.Tensor, sign: float = 1.0) -> torch.Tensor:
    """x = bf16(x + sign * y) in place (same shape, numel % 8 == 0)."""
    _chk(x, "x"), _chk(y, "y")
    assert x.shape == y.shape
    check(lib().pe_add_bf16(x.data_ptr(), y.data_ptr(), x.numel(), float(sign), stream_ptr()), "pe_add_bf16")
    return x


def layernorm_affine(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    _chk(x, "x"), _chk(w, "w"), _chk(b, "b")
    rows, dim = x.shape
    out = torch.empty_like(x)
    check(lib().pe_layernorm_affine(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), rows, dim, float(eps), stream_ptr()),
          "pe_layernorm_affine")
    return out


def perceiver_attention(q: torch.Tensor, kv: torch.Tensor, heads: int = 8) -> torch.Tensor:
    """q [nq, heads*64], kv [nk, 2*heads*64] -> [nq, heads*64] (PerceiverAttention core, helpers.py:52-62)."""
    _chk(q, "q"), _chk(kv, "kv")
    nq, nk = q.shape[0], kv.shape[0]
    assert q.shape[1] == heads * 64 and kv.shape[1] == 2 * heads * 64
    out = torch.empty_like(q)
    check(lib().pe_perceiver_attention(q.data_ptr(), kv.data_ptr(), out.data_ptr(), nq, nk, heads, 64 ** -0.5, stream_ptr()),
          "pe_perceiver_attention")
    return out


def sdpa_heads64(q: torch.Tensor, kv: torch.Tensor, heads: int) -> torch.Tensor:
    """q [nq, heads*64], kv [nk, 2*heads*64] (keys first) -> [nq, heads*64] with scaled_dot_product_attention's numerics
    (DINOv2 self-attention, pipelines/dinov2.py:8-31)."""
    _chk(q, "q"), _chk(kv, "kv")
    nq, nk = q.shape[0], kv.shape[0]
    assert q.shape[1] == heads * 64 and kv.shape[1] == 2 * heads * 64
    out = torch.empty_like(q)
    check(lib().pe_sdpa_heads64(q.data_ptr(), kv.data_ptr(), out.data_ptr(), nq, nk, heads, 64 ** -0.5, stream_ptr()), "pe_sdpa_heads64")
    return out
