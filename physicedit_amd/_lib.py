"""ctypes binding of libphysicedit_amd.so (include/physicedit_amd.h).

There is NO fallback: if the library is not built, `lib()` raises.  Nothing here imports the oracle.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
# PE_LIB_PATH: an experiment knob (tools/microbench/attn_knobs.sh A/B-tests differently generated builds); product code never sets it
LIB_PATH = os.environ.get("PE_LIB_PATH") or os.path.join(HERE, "libphysicedit_amd.so")

c_void_p, c_int, c_float, c_size_t = C.c_void_p, C.c_int, C.c_float, C.c_size_t


class PeError(RuntimeError):
    pass


class DitBlockWeights(C.Structure):
    _fields_ = [(n, c_void_p) for n in (
        "img_mod_w", "img_mod_b", "img_qkv_w", "img_qkv_b", "norm_q_w", "norm_k_w",
        "img_out_w", "img_out_b", "img_mlp_up_w", "img_mlp_up_b", "img_mlp_down_w", "img_mlp_down_b",
        "txt_mod_w", "txt_mod_b", "txt_qkv_w", "txt_qkv_b", "norm_added_q_w", "norm_added_k_w",
        "txt_out_w", "txt_out_b", "txt_mlp_up_w", "txt_mlp_up_b", "txt_mlp_down_w", "txt_mlp_down_b")]


class DitBlockLora(C.Structure):
    _fields_ = [(n, c_void_p) for n in (
        "img_qkv_a", "img_qkv_b", "img_out_a", "img_out_b", "img_down_a", "img_down_b", "img_mod_a", "img_mod_b",
        "txt_qkv_a", "txt_qkv_b", "txt_out_a", "txt_out_b", "txt_down_a", "txt_down_b", "txt_mod_a", "txt_mod_b")]


class DitWeights(C.Structure):
    _fields_ = [("num_layers", c_int)] + [(n, c_void_p) for n in (
        "time_w1", "time_b1", "time_w2", "time_b2", "txt_norm_w", "img_in_w", "img_in_b",
        "txt_in_w", "txt_in_b", "norm_out_w", "norm_out_b", "proj_out_w", "proj_out_b")] + [
        ("blocks", C.POINTER(DitBlockWeights)), ("weights_e4m3", c_int)]


class AdapterWeights(C.Structure):
    _fields_ = [(n, c_void_p) for n in (
        "dino_w0", "dino_b0", "dino_w2", "dino_b2", "vae_w0", "vae_b0", "vae_w2", "vae_b2")]


class DecodeLayerWeights(C.Structure):
    """pe_decode_layer_weights (include/physicedit_amd.h)"""
    _fields_ = [(n, c_void_p) for n in ("q_w", "q_b", "k_w", "k_b", "v_w", "v_b", "o_w", "gate_w", "up_w", "down_w", "input_norm_w",
                                        "post_norm_w")] + [("input_norm_eps", C.c_float), ("post_norm_eps", C.c_float),
                                                           ("n_q_heads", c_int), ("n_kv_heads", c_int), ("ff", c_int)]


class VaeConv(C.Structure):
    _fields_ = [("w", c_void_p), ("b", c_void_p), ("cin_p", c_int), ("cout_p", c_int), ("ksize", c_int)]


class VaeRes(C.Structure):
    _fields_ = [("conv1", VaeConv), ("conv2", VaeConv), ("shortcut", VaeConv), ("norm1_g", c_void_p), ("norm2_g", c_void_p)]


class VaeAttn(C.Structure):
    _fields_ = [("norm_g", c_void_p), ("to_qkv", VaeConv), ("proj", VaeConv)]


class VaeMid(C.Structure):
    _fields_ = [("res0", VaeRes), ("attn", VaeAttn), ("res1", VaeRes)]


class VaeWeights(C.Structure):
    _fields_ = [("enc_conv_in", VaeConv), ("enc_res", VaeRes * 8), ("enc_down", VaeConv * 3), ("enc_mid", VaeMid),
                ("enc_norm_out_g", c_void_p), ("enc_conv_out", VaeConv), ("quant_conv", VaeConv),
                ("post_quant_conv", VaeConv), ("dec_conv_in", VaeConv), ("dec_mid", VaeMid), ("dec_res", VaeRes * 12),
                ("dec_up", VaeConv * 3), ("dec_norm_out_g", c_void_p), ("dec_conv_out", VaeConv),
                ("mean", c_void_p), ("inv_std", c_void_p), ("zero_page", c_void_p)]


IMAGE_BF16_NCHW, IMAGE_U8_HWC = 0, 1


class ControlNetBlock(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("x_rms_w", "y_rms_w", "in_w", "in_b", "out_w", "out_b")]


class ControlInput(C.Structure):
    _fields_ = [("blocks", C.POINTER(ControlNetBlock)), ("conditioning", c_void_p), ("scale", c_float)]


class DitCall(C.Structure):
    _fields_ = [
        ("latents", c_void_p), ("h8", c_int), ("w8", c_int),
        ("n_edit", c_int), ("edit_latents", c_void_p * 4), ("edit_h8", c_int * 4), ("edit_w8", c_int * 4),
        ("prompt_emb", c_void_p), ("T", c_int),
        ("special_idx", c_void_p), ("n_special", c_int),
        ("alpha", c_float), ("one_minus_alpha", c_float),
        ("rope_cos_img", c_void_p), ("rope_sin_img", c_void_p),
        ("rope_cos_txt", c_void_p), ("rope_sin_txt", c_void_p),
        ("step", c_int), ("noise_pred", c_void_p),
        ("n_control", c_int), ("control", ControlInput * 4),
        ("attn_words", c_void_p),
        ("fp8_attention", c_int),
    ]


# name -> (restype, argtypes); every symbol include/physicedit_amd.h declares
ABI_VERSION = 8   # include/physicedit_amd.h: bumped on any signature / struct change

SIGNATURES = {
    "pe_last_error": (C.c_char_p, []),
    "pe_abi_version": (c_int, []),
    "pe_build_id": (C.c_char_p, []),
    "pe_debug_set": (c_int, [C.c_char_p, c_int]),
    "pe_debug_set_ptr": (c_int, [C.c_char_p, c_void_p]),
    "pe_gemm_bf16": (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                             c_void_p, c_void_p, c_int, c_void_p]),
    "pe_gemm_bf16_pre": (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                                 c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "pe_lora_merge": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_float, c_void_p]),
    "pe_ln_modulate_e4m3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_float, c_void_p]),
    "pe_quantize_rows_e4m3": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "pe_gemm_e4m3": (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
                             c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "pe_qkv_rmsnorm_rope": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "pe_flash_attn": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p,
                              c_size_t, c_void_p]),
    "pe_gemm_workspace_bytes": (c_size_t, []),
    "pe_gemm_stash_bytes": (c_size_t, []),
    "pe_attn_q_prescale": (c_float, [c_float]),
    "pe_qkv_rmsnorm_rope_scaled": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "pe_flash_attn_prescaled": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p,
                                        c_size_t, c_void_p]),
    "pe_flash_attn_fp8_scratch_bytes": (c_size_t, [c_int, c_int]),
    "pe_flash_attn_fp8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p,
                                  c_size_t, c_void_p]),
    "pe_flash_attn_masked": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p,
                                     c_size_t, c_void_p, c_int, c_void_p]),
    "pe_flash_attn_workspace_bytes": (c_size_t, [c_int, c_int]),
    "pe_ln_modulate": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_float, c_void_p]),
    "pe_rmsnorm": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "pe_gemv_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "pe_decode_qkv_rope": (c_int, [c_void_p] * 12 + [c_int, c_int, c_int, c_void_p]),
    "pe_decode_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "pe_decode_step_qkv": (c_int, [c_void_p] * 12 + [c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_float, c_void_p]),
    "pe_gemv_norm_bf16": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "pe_gemv_swiglu_norm_bf16": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "pe_decode_step_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_float, c_void_p]),
    "pe_decode_attention_workspace_bytes": (c_size_t, [c_int, c_int]),
    "pe_decode_step_attention_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_float, c_void_p,
                                               c_size_t, c_void_p]),
    "pe_decode_layer_scratch_bytes": (c_size_t, [c_int, c_int, c_int]),
    "pe_decode_layer": (c_int, [C.POINTER(DecodeLayerWeights), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                c_int, C.c_float, c_void_p, c_size_t, c_void_p]),
    "pe_decode_embed": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "pe_decode_argmax": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "pe_gemv_res_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "pe_gemv_swiglu_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "pe_dual_rmsnorm_add": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "pe_patchify": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "pe_unpatchify": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "pe_add_bf16": (c_int, [c_void_p, c_void_p, c_size_t, c_float, c_void_p]),
    "pe_layernorm_affine": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "pe_perceiver_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "pe_sdpa_heads64": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "pe_mfma_probe": (c_int, [c_void_p, c_void_p, c_int, c_int, C.POINTER(C.c_double), c_void_p]),
    "pe_gemm_mix_probe": (c_int, [c_int, c_void_p, c_size_t, c_void_p, c_int, c_int, C.POINTER(C.c_double), c_void_p]),
    "pe_cfg_euler_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_int, c_float, c_void_p]),
    "pe_cfg_inpaint_euler_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_float,
                                          c_int, c_float, c_float, c_void_p]),
    "pe_dit_create": (c_int, [C.POINTER(DitWeights), C.POINTER(AdapterWeights), C.POINTER(c_void_p)]),
    "pe_gemm_e4m3_gelu_q8": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                     c_int, c_int, c_int, c_void_p]),
    "pe_dit_destroy": (None, [c_void_p]),
    "pe_dit_set_hot_lora": (c_int, [c_void_p, C.POINTER(DitBlockLora), c_int]),
    "pe_dit_add_hot_lora": (c_int, [c_void_p, C.POINTER(DitBlockLora), c_int]),
    "pe_dit_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "pe_dit_bind_workspace": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_void_p]),
    "pe_dit_prepare": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "pe_dit_forward": (c_int, [c_void_p, C.POINTER(DitCall), c_void_p]),
    "pe_dit_special_token_mse": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "pe_dit_debug_ptr": (c_void_p, [c_void_p, C.c_char_p]),
    "pe_conv2d_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                               c_int, c_int, c_int, c_void_p]),
    "pe_vae_rmsnorm": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "pe_nchw_to_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "pe_nhwc_to_nchw": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "pe_vae_attention_scratch_bytes": (c_size_t, [c_int]),
    "pe_vae_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "pe_vae_create": (c_int, [C.POINTER(VaeWeights), C.POINTER(c_void_p)]),
    "pe_vae_destroy": (None, [c_void_p]),
    "pe_vae_workspace_bytes": (c_size_t, [c_int, c_int]),
    "pe_vae_encode": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pe_vae_decode": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "pe_adapter_workspace_bytes": (c_size_t, [c_int]),
    "pe_adapter_forward": (c_int, [C.POINTER(AdapterWeights), c_void_p, c_int, c_float, c_float, c_void_p, c_void_p, c_size_t,
                                   c_void_p]),
    "pe_attn_mix_probe": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, C.POINTER(C.c_double), c_void_p]),
    "pe_profile_enable": (c_int, [c_int, c_int]),
    "pe_profile_disable": (None, []),
    "pe_profile_read": (c_int, [c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_double),
                                C.POINTER(C.c_double)]),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PeError(
                f"{LIB_PATH} is missing: build it with `python -m physicedit_amd.build` "
                "(hipcc --offload-arch=gfx950).  physicedit_amd has no CPU/PyTorch fallback.")
        # torch first: the process must hold ONE HIP runtime.  torch ships its own libamdhip64 (same soname as
        # /opt/rocm's); loaded first, the dynamic linker binds this library to it too.  The other order leaves torch on
        # a second runtime and every HIP call made here fails with "no ROCm-capable device is detected".
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        # experiment knobs (see pe_debug_set in include/physicedit_amd.h), e.g. PE_DEBUG="attn_variant=0,gemm_variant=17"
        for item in filter(None, os.environ.get("PE_DEBUG", "").split(",")):
            key, _, val = item.partition("=")
            if handle.pe_debug_set(key.strip().encode(), int(val)) != 0:
                raise PeError(f"PE_DEBUG: unknown knob {key!r}")
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().pe_last_error().decode("utf-8", "replace")
        raise PeError(f"{what or 'physicedit_amd'} failed (rc={rc}): {msg}")


def stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
