"""Prompt prologue of QwenImagePhysicPipeline: host Python over HF `transformers` under PyTorch-ROCm, as the north star
specifies (the Qwen2.5-VL text encoder is third-party code; nothing here is a HIP kernel of this repository).

Behaviour follows the reference's two units at inference (is_train=False):
  * QwenImageUnit_PhysicalVerbalEmbedder   DiffSynth-Studio/diffsynth/pipelines/qwen_image_physical.py:837-990
      chat-template prompt (system = SYSTEM_PROMPT_SAMPLE, user = "Edit Instruction:", prompt, "Edit Image:", image)
      -> text_encoder.generate(max_new_tokens=1000) -> JSON -> "\\n<key>: <value>" lines (raw text if it is not JSON)
  * QwenImageUnit_PromptEmbedder           :732-835
      prompt (+ physical text) in one of three templates -> hidden_states[-1] of the text encoder -> rows under the
      attention mask, minus the first `drop_idx` template tokens -> prompt_emb [1,T,3584]; with one edit image the 64
      `<imgN>` tokens between <begin_of_img> / <end_of_img> give special_token_mask [1,T]
The template strings and the system prompt are DATA the checkpoint was trained with; they are restated here verbatim.

The nega branch of the verbal embedder also generates text in the reference, but PromptEmbedder never reads it
(input_params_nega has no `physical_txt`, :736-737): it is skipped here.
"""
from __future__ import annotations

import json
import math
import os
import collections
import hashlib
from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch
from PIL import Image

SPECIAL_TOKEN_NUM = 64     # qwen_image_physical.py:35

# qwen_image_physical.py:764-765, :773-781, :806-809 (templates) and :136-155 (system prompt)
T2I_TEMPLATE = ("<|im_start|>system\nDescribe the image by detailing the color, shape, size, texture, quantity, text, spatial "
                "relationships of the objects and background:<|im_end|>\n<|im_start|>user\n{}<|im_end|>\n<|im_start|>assistant\n")
T2I_DROP = 34
EDIT_TEMPLATE = ("<|im_start|>system\nDescribe the key features of the input image (color, shape, size, texture, objects, background), "
                 "then explain how the user's text instruction should alter or modify the image. Generate a new image that meets the "
                 "user's requirements while maintaining consistency with the original input where appropriate.<|im_end|>\n"
                 "<|im_start|>user\n<|vision_start|><|image_pad|><|vision_end|>{}<|im_end|>\n<|im_start|>assistant\n")
EDIT_MULTI_TEMPLATE = ("<|im_start|>system\nDescribe the key features of the input image (color, shape, size, texture, objects, "
                       "background), then explain how the user's text instruction should alter or modify the image. Generate a new "
                       "image that meets the user's requirements while maintaining consistency with the original input where "
                       "appropriate.<|im_end|>\n<|im_start|>user\n{}<|im_end|>\n<|im_start|>assistant\n")
EDIT_DROP = 64
PICTURE_TEMPLATE = "Picture {}: <|vision_start|><|image_pad|><|vision_end|>"
SYSTEM_PROMPT_SAMPLE = """
You are a physics-aware visual editing assistant.
You will receive an "Edit Instruction" and an "Edit Image".
Your task is to generate a detailed description of the edit operations required to transform the image according to the instruction, ensuring all changes strictly follow physical laws.

INPUTS:
- Edit Instruction: The desired modification.
- Edit Image: The visual starting point.

REQUIREMENTS:
1. Physical Plausibility: All operations must respect physics (like gravity, inertia, material properties, light transport, collision, etc.).
2. Mechanism of Change: Describe *how* the change occurs visually (e.g., "The vase tilts and falls due to gravity," not just "The vase is on the floor").
3. Material Consistency: Ensure materials behave correctly (liquids flow, solids rigid/deform, cloth wrinkles).

OUTPUT FORMAT:
Return STRICT JSON ONLY:
{
  "middle_transition_prompt": "A multi-clause paragraph describing the step-by-step physical operations and visual transition."
}
""".strip()

ACCEPTED_FIELD_SETS = (("Reasoning",), ("physical_reasoning", "middle_transition_prompt", "final_state_prompt"),
                       ("middle_transition_prompt",))


def special_tokens() -> List[str]:
    """Tokens from_pretrained adds to the processor's tokenizer (:529-536)."""
    return ["<begin_of_img>", "<end_of_img>"] + [f"<img{i}>" for i in range(SPECIAL_TOKEN_NUM)]


def resize_for_vl(image: Image.Image, target_area: int = 384 * 384) -> Image.Image:
    """calculate_dimensions + resize_image (:749-759): ~target_area pixels, sides rounded to /32, PIL's default filter."""
    ratio = image.size[0] / image.size[1]
    width = math.sqrt(target_area * ratio)
    height = width / ratio
    return image.resize((round(width / 32) * 32, round(height / 32) * 32))


def parse_generation_response(response: str) -> Dict[str, str]:
    """What `_parse_generation_response` accepts (:875-907): the text between the first "{" and the last "}" must be a JSON object
    whose known fields (the union of ACCEPTED_FIELD_SETS; null counts as absent, anything else must be a string) are EXACTLY one
    of the accepted sets.  Returns the stripped strings in the declared order of that set -- the reference walks a Python `set`,
    so for the three-field form its order changes with the interpreter's hash seed; a fixed order is one of its outcomes.
    ValueError for everything else (the caller then keeps the raw text, :868-871)."""
    lo, hi = response.find("{"), response.rfind("}")
    if not 0 <= lo < hi:
        raise ValueError("the generated text holds no JSON object")
    try:
        obj = json.loads(response[lo:hi + 1])
    except json.JSONDecodeError as err:
        raise ValueError("the braces of the generated text do not enclose valid JSON") from err
    if not isinstance(obj, dict):
        raise ValueError("the generated JSON is not an object")
    known = {name for names in ACCEPTED_FIELD_SETS for name in names}
    present = {name: obj[name] for name in obj if name in known and obj[name] is not None}
    wrong = [name for name, value in present.items() if not isinstance(value, str)]
    if wrong:
        raise ValueError(f"non-string value for {wrong[0]!r} ({type(present[wrong[0]]).__name__})")
    for names in ACCEPTED_FIELD_SETS:
        if set(names) == set(present):
            return {name: present[name].strip() for name in names}
    raise ValueError(f"fields {sorted(present)} are none of the accepted sets {ACCEPTED_FIELD_SETS}")


class MiniQwen2VLProcessor:
    """The three things the prologue needs from `transformers.Qwen2VLProcessor`, without its torchvision dependency
    (transformers >= 4.5x wants a video processor, which needs torchvision; this ROCm image has none):
    `__call__(text=, images=, padding=, return_tensors=)`, `apply_chat_template(...)` and `.tokenizer`.
    Image patches come from transformers' own Qwen2VLImageProcessor (its PIL backend when torchvision is absent)."""

    image_token = "<|image_pad|>"

    def __init__(self, image_processor, tokenizer, chat_template: str):
        self.image_processor = image_processor
        self.tokenizer = tokenizer
        self.chat_template = chat_template

    def __call__(self, text, images=None, padding=True, return_tensors="pt", **kw):
        from transformers.feature_extraction_utils import BatchFeature
        text = [text] if isinstance(text, str) else list(text)
        data = {}
        if images is not None:
            images = [images] if isinstance(images, Image.Image) else list(images)
            img = self.image_processor(images=images, return_tensors=return_tensors)
            grid = img["image_grid_thw"]
            merge = int(self.image_processor.merge_size) ** 2
            idx = 0
            for i in range(len(text)):
                while self.image_token in text[i]:
                    n = int(grid[idx].prod()) // merge
                    text[i] = text[i].replace(self.image_token, "<|placeholder|>" * n, 1)
                    idx += 1
                text[i] = text[i].replace("<|placeholder|>", self.image_token)
            data.update({"pixel_values": img["pixel_values"], "image_grid_thw": grid})
        enc = self.tokenizer(text, padding=padding, return_tensors=return_tensors)
        return BatchFeature(data={**enc, **data})

    def apply_chat_template(self, messages, tokenize=False, add_generation_prompt=True, add_vision_id=False, **kw):
        assert not tokenize
        from jinja2.sandbox import ImmutableSandboxedEnvironment
        env = ImmutableSandboxedEnvironment(trim_blocks=True, lstrip_blocks=True)
        return env.from_string(self.chat_template).render(messages=messages, add_generation_prompt=add_generation_prompt,
                                                          add_vision_id=add_vision_id)


def load_processor(processor_path: str, tokenizer_path: Optional[str] = None):
    """Qwen2VLProcessor.from_pretrained(processor_path) (:525-527) when transformers can build it; else the minimal
    processor above from the same directory's preprocessor_config.json / chat_template.jinja and the tokenizer found in
    `processor_path` (or `tokenizer_path` when the processor directory ships without vocab/merges)."""
    try:
        from transformers import Qwen2VLProcessor
        return Qwen2VLProcessor.from_pretrained(processor_path)
    except Exception:
        pass
    from transformers import Qwen2Tokenizer, Qwen2VLImageProcessor
    tok_dir = processor_path if os.path.exists(os.path.join(processor_path, "vocab.json")) else tokenizer_path
    if tok_dir is None:
        raise FileNotFoundError(f"{processor_path} has no vocab.json and no tokenizer path was given")
    tokenizer = Qwen2Tokenizer.from_pretrained(tok_dir)
    cfg = {}
    cfg_path = os.path.join(processor_path, "preprocessor_config.json")
    if os.path.exists(cfg_path):
        raw = json.load(open(cfg_path))
        keep = ("do_resize", "do_rescale", "do_normalize", "image_mean", "image_std", "min_pixels", "max_pixels", "patch_size",
                "temporal_patch_size", "merge_size", "do_convert_rgb", "resample", "rescale_factor", "size")
        cfg = {k: raw[k] for k in keep if k in raw and raw[k] is not None}
    template = None
    for name in ("chat_template.jinja", "chat_template.json"):
        pth = os.path.join(processor_path, name)
        if os.path.exists(pth):
            template = open(pth).read()
            if name.endswith(".json"):
                template = json.loads(template)["chat_template"]
            break
    if template is None:
        template = getattr(tokenizer, "chat_template", None)
    return MiniQwen2VLProcessor(Qwen2VLImageProcessor(**cfg), tokenizer, template)


# Qwen2.5-VL-7B architecture of the Qwen-Image text encoder (models/qwen_image_text_encoder_withdecode.py:8-147): data
TEXT_ENCODER_CONFIG = dict(
    text_config=dict(hidden_size=3584, intermediate_size=18944, num_hidden_layers=28, num_attention_heads=28, num_key_value_heads=4,
                     vocab_size=152064, max_position_embeddings=128000, rms_norm_eps=1e-6, rope_theta=1000000.0, hidden_act="silu",
                     rope_scaling={"type": "default", "rope_type": "default", "mrope_section": [16, 24, 24]},
                     tie_word_embeddings=False, use_sliding_window=False, sliding_window=None, max_window_layers=28,
                     attention_dropout=0.0, bos_token_id=151643, eos_token_id=151645, pad_token_id=151645),
    vision_config=dict(depth=32, hidden_size=1280, intermediate_size=3420, num_heads=16, out_hidden_size=3584, patch_size=14,
                       spatial_merge_size=2, temporal_patch_size=2, window_size=112, fullatt_block_indexes=[7, 15, 23, 31],
                       in_chans=3, hidden_act="silu", tokens_per_second=2),
    image_token_id=151655, video_token_id=151656, vision_start_token_id=151652, vision_end_token_id=151653,
    bos_token_id=151643, eos_token_id=151645, tie_word_embeddings=False)


def convert_text_encoder_keys(state_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Checkpoint layout -> transformers' Qwen2_5_VLForConditionalGeneration layout
    (QwenImageTextEncoderStateDictConverter.from_diffusers, qwen_image_text_encoder_withdecode.py:286-297)."""
    out = {}
    for k, v in state_dict.items():
        if k.startswith("visual."):
            k = "model." + k
        elif k.startswith("model.language_model."):
            pass
        elif k.startswith("model."):
            k = k.replace("model.", "model.language_model.", 1)
        out[k] = v
    return out


def build_text_encoder(state_dict: Dict[str, torch.Tensor], device, torch_dtype=torch.bfloat16, config: Optional[dict] = None):
    """transformers' Qwen2_5_VLForConditionalGeneration, created directly on `device` in `torch_dtype` (16.6 GB for the
    7B model: nothing next to 288 GB of HBM) and filled from the checkpoint tensors."""
    import contextlib
    from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration
    cfg = Qwen2_5_VLConfig(**(config or TEXT_ENCODER_CONFIG))
    try:
        from transformers.modeling_utils import no_init_weights       # skip the random init: every tensor is overwritten
        guard = no_init_weights()
    except Exception:
        guard = contextlib.nullcontext()
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch_dtype)
    try:
        with guard, torch.device(device):
            model = Qwen2_5_VLForConditionalGeneration(cfg)
    finally:
        torch.set_default_dtype(old)
    missing, unexpected = model.load_state_dict(convert_text_encoder_keys(state_dict), strict=False)
    missing = [k for k in missing if "rotary_emb" not in k and "inv_freq" not in k]
    if missing:
        raise RuntimeError(f"text encoder checkpoint misses {len(missing)} tensors, e.g. {missing[:3]}")
    return model.eval()


def decode_internals_compatible() -> Tuple[bool, str]:
    """The q_len = 1 replacements of Qwen2_5_VLAttention.forward / Qwen2_5_VLDecoderLayer.forward below depend on transformers
    internals: keyword names, the attention returning `(attn_output, attn_weights)`, the decoder layer returning the bare hidden
    state.  transformers is not pinned by this repo (tested: 5.x), so the contract is checked where it is installed: version
    range, signatures, and the return statements of the stock methods.  Anything else -> the two classes are left alone and
    decoding runs through the stock modules (the row-wise Linear / RMSNorm / MLP replacements do not depend on any of this)."""
    import inspect
    import re
    import transformers
    try:
        ver = tuple(int(x) for x in re.findall(r"\d+", transformers.__version__)[:2])
        from transformers.models.qwen2_5_vl import modeling_qwen2_5_vl as M
        if not (5, 0) <= ver < (6, 0) and not os.environ.get("PE_PROLOGUE_ANY_TRANSFORMERS"):
            return False, f"transformers {transformers.__version__} is outside the tested range 5.x"
        need = {"Qwen2_5_VLDecoderLayer": ("hidden_states", "attention_mask", "position_ids", "past_key_values", "use_cache",
                                           "position_embeddings"),
                "Qwen2_5_VLAttention": ("hidden_states", "attention_mask", "position_ids", "past_key_values", "position_embeddings")}
        tail = {"Qwen2_5_VLDecoderLayer": r"return\s+hidden_states\s*$", "Qwen2_5_VLAttention": r"return\s+attn_output\s*,\s*attn_weights\s*$"}
        for name, params in need.items():
            fwd = getattr(M, name).forward
            have = inspect.signature(fwd).parameters
            miss = [p for p in params if p not in have]
            if miss:
                return False, f"{name}.forward has no parameter {miss[0]!r}"
            if not re.search(tail[name], inspect.getsource(fwd).rstrip()):
                return False, f"{name}.forward no longer ends with the return structure the replacement mirrors"
    except Exception as e:          # missing module, unreadable source, ...
        return False, f"{type(e).__name__}: {e}"
    return True, ""


def accelerate_decode(model) -> int:
    """Autoregressive decoding applies every nn.Linear of the language model to ONE row: 15 GB of weights per generated token, an
    HBM-bound GEMV.  On the GPU those calls are routed to the library's pe_gemv_bf16 (fp32 accumulation, one rounding, like
    nn.Linear); everything with more than one row (prefill, the vision tower) stays on torch.nn.functional.linear.  The language
    model's RMSNorms go to pe_rmsnorm (one launch instead of five).  The
    summation order differs from the BLAS GEMV's, as it does between BLAS versions: greedy decoding may break a near-tie the
    other way.  Returns the number of wrapped modules."""
    import torch.nn.functional as F
    from physicedit_amd import ops
    n = 0
    rope_cache = {"cos": None, "cs": None, "sn": None}      # the section-selected rotary row of the current step, shared by all layers
    internals_ok, why = decode_internals_compatible()
    if not internals_ok:
        print(f"[prompt_prologue] fused decode attention / layer not installed ({why}); single-row Linears, RMSNorms and MLPs only")
    for m in model.modules():
        if (isinstance(m, torch.nn.Linear) and m.weight.is_cuda and m.weight.dtype == torch.bfloat16 and m.weight.is_contiguous()
                and m.in_features % 8 == 0 and m.in_features <= 32768):
            def forward(x, _m=m):
                if x.numel() == _m.in_features and x.dtype == torch.bfloat16 and x.is_contiguous():
                    return ops.gemv(x, _m.weight, _m.bias).view(*x.shape[:-1], _m.out_features)
                return F.linear(x, _m.weight, _m.bias)
            m.forward = forward
            n += 1
        elif (type(m).__name__.endswith("RMSNorm") and hasattr(m, "variance_epsilon") and tuple(m.weight.shape) == (3584,)
              and m.weight.is_cuda and m.weight.dtype == torch.bfloat16):
            # the language model's RMSNorm (fp32 normalise -> bf16 -> x weight, the same roundings as pe_rmsnorm) is five
            # element-wise launches in eager PyTorch; the decode step is launch-bound (~840 launches per token)
            def norm_forward(x, _m=m, _orig=m.forward):
                if x.dtype == torch.bfloat16 and x.shape[-1] == 3584 and x.is_contiguous() and x.is_cuda:
                    return ops.rmsnorm(x.reshape(-1, 3584), _m.weight, float(_m.variance_epsilon)).view(x.shape)
                return _orig(x)
            m.forward = norm_forward
            n += 1
        elif (type(m).__name__ == "Qwen2MLP" and type(getattr(m, "act_fn", None)).__name__ in ("SiLU", "SiLUActivation")
              and m.gate_proj.bias is None and m.up_proj.bias is None and m.gate_proj.weight.is_cuda
              and m.gate_proj.weight.dtype == torch.bfloat16 and m.gate_proj.in_features % 8 == 0):
            # act_fn(gate_proj(x)) * up_proj(x) on one row: two GEMVs + SiLU + product in one launch instead of four
            def mlp_forward(x, _m=m, _orig=m.forward):
                if x.numel() == _m.gate_proj.in_features and x.dtype == torch.bfloat16 and x.is_contiguous():
                    hdn = ops.gemv_swiglu(x, _m.gate_proj.weight, _m.up_proj.weight).view(*x.shape[:-1], -1)
                    return _m.down_proj(hdn)
                return _orig(x)
            m.forward = mlp_forward
            n += 1
        elif not internals_ok:
            continue
        elif (type(m).__name__ == "Qwen2_5_VLAttention" and getattr(m, "head_dim", 0) == 128 and m.q_proj.weight.is_cuda
              and m.q_proj.weight.dtype == torch.bfloat16 and m.q_proj.in_features % 8 == 0):
            n += _patch_decode_attention(m, ops, rope_cache)
        elif type(m).__name__ == "Qwen2_5_VLDecoderLayer":
            n += _patch_decode_layer(m, ops)
    return n


class GraphDecoder:
    """Greedy `generate` of the text encoder (qwen_image_physical.py:859-873: `text_encoder.generate(**model_inputs,
    max_new_tokens=1000)`, greedy by the model's generation config) with the whole decode step -- embedding row, 28 decoder layers
    of 6 launches (q/k/v + rotary + cache append with the input norm fused, attention, o_proj + residual, gate/up + SiLU with the
    post-attention norm fused, down_proj + residual), lm_head with the final norm fused, arg-max -- captured ONCE per cache-capacity bucket (round 4: the
    capture outlives the call) in a hipGraph and replayed per token: no Python, no
    launch latency, no Cache object in the loop.  The prefill stays on transformers; its DynamicCache is copied into static planes
    [layer][kv_head][prompt + max_new_tokens][128] that the q/k/v launch appends to (pe_decode_step_*: everything that changes per
    token is read from device memory).  Same kernels, same order, same roundings as `accelerate_decode`, so the tokens are the ones
    the patched `generate()` produces.  EOS is tested on the host every `chunk` tokens; the overshoot is discarded."""

    MAX_CACHE_ROWS = 15360          # rows of the static K / V planes pe_decode_step_attention can walk

    def __init__(self, model, chunk: int = 16):
        from physicedit_amd import ops
        self.ops, self.model, self.chunk = ops, model, chunk
        lm = model.model.language_model
        self.lm = lm
        self.layers = list(lm.layers)
        at = self.layers[0].self_attn
        w = at.q_proj.weight
        if not (w.is_cuda and w.dtype == torch.bfloat16 and getattr(at, "head_dim", 0) == 128 and lm.embed_tokens.weight.dtype == w.dtype):
            raise ValueError("GraphDecoder needs a bf16 CUDA language model with 128-wide heads")
        try:
            section = list(at.config.rope_parameters["mrope_section"])
        except Exception:
            section = list(getattr(at, "rope_scaling", {}).get("mrope_section", []))
        if sum(section) * 2 != 128:
            raise ValueError("GraphDecoder: unexpected mrope sections")
        self.sel = torch.tensor([i % 3 for i, n_ in enumerate(section * 2) for _ in range(n_)], device=w.device)
        self.hkv = at.k_proj.out_features // 128
        for ly in self.layers:
            if ly.self_attn.o_proj.bias is not None or ly.mlp.down_proj.bias is not None or ly.mlp.gate_proj.bias is not None:
                raise ValueError("GraphDecoder: unexpected biases in o_proj / the MLP")
        # round 6, OPT-IN (PE_DECODE_LAYER_KERNEL=1): one launch per decoder layer (pe_decode_layer: a persistent grid walks the eight launches'
        # work items with grid-wide barriers in between; same bits, so the same tokens -- and 71 - 109 tok/s against 298: a software grid
        # barrier costs 40 - 67 us on the 8-XCD chip (agent-scope atomics serialise at the memory side), ten times the launch boundary it
        # replaces.  Without its barriers (wrong results) the same kernel runs at 362 tok/s: profiles/r06_prologue_notes.md.)
        self.layer_kernel = os.environ.get("PE_DECODE_LAYER_KERNEL", "0") == "1"
        self._layer_w = None        # per layer: the C struct of its operands (addresses: rebuilt when the weights' fingerprint changes)
        self.split_attention = os.environ.get("PE_DECODE_SPLIT_ATTENTION", "1") != "0"  # pe_decode_step_attention_split (448 work-groups per launch) instead of the 28-work-group launch; same bits
        self._static = {}           # capacity bucket -> static planes, tables, counters and the captured decode step
        self.captures = 0           # graphs captured so far (tests: calls of one bucket share one)

    def _weights_fingerprint(self):
        """Address, dtype and shape of every tensor the captured launches take by pointer.  A captured hipGraph bakes them in: after
        .to() / offload + reload / a LoRA merge that re-allocates / a dtype change, a replay would read freed or stale memory without
        any error, so generate() compares this with the fingerprint taken at capture time and re-captures on a mismatch."""
        ts = [self.lm.embed_tokens.weight, self.lm.norm.weight, self.model.lm_head.weight, self.model.lm_head.bias]
        for ly in self.layers:
            at, mlp = ly.self_attn, ly.mlp
            ts += [at.q_proj.weight, at.q_proj.bias, at.k_proj.weight, at.k_proj.bias, at.v_proj.weight, at.v_proj.bias, at.o_proj.weight,
                   mlp.gate_proj.weight, mlp.up_proj.weight, mlp.down_proj.weight, ly.input_layernorm.weight, ly.post_attention_layernorm.weight]
        return tuple((t.data_ptr(), t.dtype, tuple(t.shape)) if t is not None else None for t in ts)

    def reset(self):
        """Drop the captured graph and its static planes (28 KiB per cache row) now instead of at the next call.  Nothing has to call this
        for correctness: generate() notices moved / re-allocated weights by itself (_weights_fingerprint) and captures again."""
        self._static = {}

    def _step(self, st):
        """one decode step on the current stream (captured); st: dict of static device tensors"""
        ops = self.ops
        x = ops.decode_embed(self.lm.embed_tokens.weight, st["token"])
        if self.layer_kernel:
            for l, ly in enumerate(self.layers):
                x = ops.decode_layer(st["layer_w"][l], x, st["xbuf"][l & 1], st["cos"], st["sin"], st["kc"][l], st["vc"][l], st["step"], st["base"],
                                     float(ly.self_attn.scaling), st["layer_scratch"])
        for l, ly in enumerate(() if self.layer_kernel else self.layers):
            at, mlp = ly.self_attn, ly.mlp
            # the two RMSNorms ride in the staging of the launches they feed (bit-identical to pe_rmsnorm): 6 launches per layer
            q = ops.decode_step_qkv(x, at.q_proj.weight, at.q_proj.bias, at.k_proj.weight, at.k_proj.bias, at.v_proj.weight,
                                    at.v_proj.bias, st["cos"], st["sin"], st["kc"][l], st["vc"][l], st["step"], st["base"],
                                    norm_w=ly.input_layernorm.weight, eps=float(ly.input_layernorm.variance_epsilon))
            a = ops.decode_step_attention(q, st["kc"][l], st["vc"][l], st["step"], st["base"], float(at.scaling), workspace=st.get("attn_ws"))
            h1 = ops.gemv(a, at.o_proj.weight, None, res=x)
            hid = ops.gemv_swiglu_norm(h1, ly.post_attention_layernorm.weight, float(ly.post_attention_layernorm.variance_epsilon),
                                       mlp.gate_proj.weight, mlp.up_proj.weight)
            x = ops.gemv(hid, mlp.down_proj.weight, None, res=h1)
        logits = ops.gemv_norm(x, self.lm.norm.weight, float(self.lm.norm.variance_epsilon), self.model.lm_head.weight,
                               self.model.lm_head.bias)
        ops.decode_argmax(logits, st["token"], st["out_ids"], st["step"])

    @torch.no_grad()
    def generate(self, max_new_tokens: int = 1000, **model_inputs) -> torch.Tensor:
        """-> [1, prompt + generated] token ids, like GenerationMixin.generate for a batch of one (EOS included when it came)."""
        model, dev = self.model, self.lm.embed_tokens.weight.device
        ids = model_inputs["input_ids"]
        if ids.shape[0] != 1:
            raise ValueError("GraphDecoder.generate: batch of one")
        Lp = ids.shape[1]
        gen = model.generation_config
        eos = gen.eos_token_id
        eos = set() if eos is None else set(eos if isinstance(eos, (list, tuple)) else [eos])
        min_new = int(getattr(gen, "min_new_tokens", 0) or 0)
        cap = Lp + max_new_tokens
        if cap > self.MAX_CACHE_ROWS:
            raise ValueError(f"GraphDecoder.generate: prompt + max_new_tokens exceeds the decode attention's {self.MAX_CACHE_ROWS}-row cache")
        # ---- prefill on transformers (one pass over the prompt; image tokens, mrope positions, rope_deltas are its business)
        try:
            out = model(**model_inputs, use_cache=True, logits_to_keep=1)
        except TypeError:
            out = model(**model_inputs, use_cache=True)
        first = int(out.logits[0, -1].float().argmax())
        past = out.past_key_values
        L = len(self.layers)
        # ---- static state, one set per capacity bucket, kept across calls WITH its captured graph: nothing the graph's launches take
        # as an argument depends on the prompt (the cache row of generated token t is step = Lp + t, read from the device counter;
        # `base` is 0; the rotary tables are indexed by that same step), so a later call of the same bucket only refills the planes'
        # first Lp rows, the table rows [Lp, Lp + max_new_tokens) and the counter, and replays
        bucket = min(self.MAX_CACHE_ROWS, (cap + 1023) // 1024 * 1024)
        st = self._static.get(bucket)
        fp = self._weights_fingerprint()
        if st is not None and st.get("weights") != fp:
            st = None                                      # the weights moved since the capture: those launches point at the old ones
        if st is None:
            st = {"kc": torch.zeros((L, self.hkv, bucket, 128), dtype=torch.bfloat16, device=dev), "base": 0, "cap": bucket,
                  "cos": torch.zeros((bucket, 128), dtype=torch.bfloat16, device=dev),
                  "token": torch.zeros(1, dtype=torch.int32, device=dev), "step": torch.zeros(1, dtype=torch.int32, device=dev),
                  "out_ids": torch.zeros(bucket, dtype=torch.int32, device=dev), "graph": None, "weights": fp}
            st["vc"] = torch.zeros_like(st["kc"])
            if self.split_attention:     # scratch of the three-launch single-query attention (layers run one after the other: one buffer)
                st["attn_ws"] = self.ops.decode_attention_workspace(len(self.layers[0].self_attn.q_proj.weight) // 128, bucket, dev)
            st["sin"] = torch.zeros_like(st["cos"])
            if self.layer_kernel:
                at0, mlp0 = self.layers[0].self_attn, self.layers[0].mlp
                st["layer_scratch"] = self.ops.decode_layer_scratch(at0.q_proj.out_features // 128, bucket, mlp0.gate_proj.out_features, dev)
                st["xbuf"] = torch.zeros((2, self.lm.embed_tokens.weight.shape[1]), dtype=torch.bfloat16, device=dev)
                st["layer_w"] = [self.ops.decode_layer_weights(
                    ly.self_attn.q_proj.weight, ly.self_attn.q_proj.bias, ly.self_attn.k_proj.weight, ly.self_attn.k_proj.bias,
                    ly.self_attn.v_proj.weight, ly.self_attn.v_proj.bias, ly.self_attn.o_proj.weight, ly.mlp.gate_proj.weight, ly.mlp.up_proj.weight,
                    ly.mlp.down_proj.weight, ly.input_layernorm.weight, ly.input_layernorm.variance_epsilon, ly.post_attention_layernorm.weight,
                    ly.post_attention_layernorm.variance_epsilon) for ly in self.layers]
            self._static = {bucket: st}                    # one bucket alive at a time (28 KiB per row of capacity and plane pair)
        kc, vc = st["kc"], st["vc"]
        for l in range(L):
            if hasattr(past, "layers"):                    # transformers >= 4.54: DynamicCache.layers[l].keys / .values
                k, v = past.layers[l].keys, past.layers[l].values
            elif hasattr(past, "key_cache"):
                k, v = past.key_cache[l], past.value_cache[l]
            else:
                k, v = past[l][0], past[l][1]
            kc[l, :, :Lp].copy_(k[0])
            vc[l, :, :Lp].copy_(v[0])
        del out, past
        # ---- rotary tables of the decode positions: position of generated token t = prompt length + t + rope_delta on all 3 axes
        delta = getattr(model.model, "rope_deltas", None)
        delta = int(delta.reshape(-1)[0]) if delta is not None else 0
        pos = (torch.arange(Lp, Lp + max_new_tokens, device=dev) + delta).view(1, 1, -1).expand(3, 1, -1)
        cos, sin = self.lm.rotary_emb(kc.new_zeros(1), pos)               # [3, 1, T, 128]
        ar = torch.arange(128, device=dev)
        st["cos"][Lp:Lp + max_new_tokens].copy_(cos[:, 0].permute(1, 0, 2)[:, self.sel, ar].to(torch.bfloat16))
        st["sin"][Lp:Lp + max_new_tokens].copy_(sin[:, 0].permute(1, 0, 2)[:, self.sel, ar].to(torch.bfloat16))
        tokens = [first]
        n_steps = max_new_tokens - 1                       # the prefill produced the first new token
        if n_steps > 0 and not (first in eos and min_new <= 1):
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                if st["graph"] is None:
                    st["step"].fill_(Lp)
                    st["token"].fill_(first)
                    self._step(st)                         # warm-up outside the capture (lazy module loads, allocator)
                    side.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, stream=side):
                        self._step(st)
                    st["graph"] = graph
                    self.captures += 1
                graph = st["graph"]
                st["step"].fill_(Lp)
                st["token"].fill_(first)
                done = 0
                while done < n_steps:
                    n = min(self.chunk, n_steps - done)
                    for _ in range(n):
                        graph.replay()
                    new = st["out_ids"][Lp + done:Lp + done + n].tolist()            # synchronises
                    if self.layer_kernel and self.ops.decode_layer_error(st["layer_scratch"]):
                        raise RuntimeError("pe_decode_layer: a grid barrier timed out (the decode grid was not fully resident); "
                                           "set PE_DECODE_LAYER_KERNEL=0")
                    done += n
                    stop = next((i for i, t in enumerate(new) if t in eos and len(tokens) + i + 1 >= max(min_new, 1)), None)
                    if stop is not None:
                        tokens += new[:stop + 1]
                        break
                    tokens += new
            torch.cuda.current_stream(dev).wait_stream(side)
        return torch.cat([ids, torch.tensor([tokens], dtype=ids.dtype, device=ids.device)], dim=1)


def _patch_decode_attention(m, ops, shared=None) -> int:
    """q_len = 1 path of Qwen2_5_VLAttention.forward (transformers): q / k / v projections + multimodal rotary embedding in two
    launches (pe_decode_qkv_rope) instead of ~25 element-wise ones, the cache update left to transformers' Cache object, then one
    GQA-aware single-query attention launch (pe_decode_attention) instead of repeat_kv copies + SDPA + transposes, then o_proj
    (already a pe_gemv_bf16).  Anything else (prefill, batches, attention weights requested) takes the original forward."""
    orig = m.forward
    try:
        section = list(m.config.rope_parameters["mrope_section"])
    except Exception:
        section = list(getattr(m, "rope_scaling", {}).get("mrope_section", []))
    if sum(section) * 2 != 128:
        return 0
    sel = torch.tensor([i % 3 for i, n_ in enumerate(section * 2) for _ in range(n_)], device=m.q_proj.weight.device)
    ar = torch.arange(128, device=sel.device)
    last = shared if shared is not None else {"cos": None, "cs": None, "sn": None}
    hq, hkv = m.q_proj.out_features // 128, m.k_proj.out_features // 128

    def forward(hidden_states, attention_mask=None, position_ids=None, past_key_values=None, output_attentions=False,
                use_cache=False, position_embeddings=None, **kwargs):
        if (hidden_states.shape[0] != 1 or hidden_states.shape[1] != 1 or past_key_values is None or output_attentions
                or position_embeddings is None or hidden_states.dtype != torch.bfloat16 or not hidden_states.is_contiguous()):
            return orig(hidden_states, attention_mask=attention_mask, position_ids=position_ids, past_key_values=past_key_values,
                        output_attentions=output_attentions, use_cache=use_cache, position_embeddings=position_embeddings, **kwargs)
        cos, sin = position_embeddings
        return m.o_proj(core(hidden_states, past_key_values, cos, sin).view(1, 1, hq * 128)), None

    def core(hidden_states, past_key_values, cos, sin):
        """attention output of the one new token BEFORE o_proj, [Hq * 128]"""
        if last["cos"] is not cos:                 # the same (cos, sin) pair reaches every layer of one step.  The tensor itself is
            last["cos"] = cos                      # kept (not its id()): a freed tensor's id is reused by the next step's
            last["cs"] = cos[:, 0, 0, :][sel, ar].to(torch.bfloat16).contiguous()
            last["sn"] = sin[:, 0, 0, :][sel, ar].to(torch.bfloat16).contiguous()
        q, k, v = ops.decode_qkv_rope(hidden_states, m.q_proj.weight, m.q_proj.bias, m.k_proj.weight, m.k_proj.bias,
                                      m.v_proj.weight, m.v_proj.bias, last["cs"], last["sn"])
        kc, vc = past_key_values.update(k.view(1, hkv, 1, 128), v.view(1, hkv, 1, 128), m.layer_idx)
        if not (kc.is_contiguous() and vc.is_contiguous()):
            kc, vc = kc.contiguous(), vc.contiguous()
        return ops.decode_attention(q, kc[0], vc[0], float(m.scaling))

    m.forward = forward
    m._pe_decode_core = core
    return 1


def _patch_decode_layer(layer, ops) -> int:
    """q_len = 1 path of Qwen2_5_VLDecoderLayer.forward: the two `residual + sublayer(x)` additions ride in the epilogue of the
    o_proj / down_proj GEMVs (pe_gemv_res_bf16: the Linear's output and the sum are rounded separately, as in eager PyTorch), so a
    layer is 8 launches: norm, q/k/v + rotary, 2 x cache append (transformers), attention, o_proj + residual, norm, gate/up + SiLU,
    down_proj + residual."""
    orig = layer.forward
    attn, mlp = layer.self_attn, layer.mlp
    if not (type(mlp).__name__ == "Qwen2MLP" and mlp.down_proj.bias is None and mlp.gate_proj.bias is None and mlp.up_proj.bias is None
            and getattr(attn, "head_dim", 0) == 128 and attn.o_proj.weight.is_cuda and attn.o_proj.weight.dtype == torch.bfloat16):
        return 0

    def forward(hidden_states, attention_mask=None, position_ids=None, past_key_values=None, use_cache=False,
                position_embeddings=None, **kwargs):
        core = getattr(attn, "_pe_decode_core", None)
        if (core is None or hidden_states.shape[0] != 1 or hidden_states.shape[1] != 1 or past_key_values is None
                or position_embeddings is None or hidden_states.dtype != torch.bfloat16 or not hidden_states.is_contiguous()
                or kwargs.get("output_attentions")):
            return orig(hidden_states, attention_mask=attention_mask, position_ids=position_ids, past_key_values=past_key_values,
                        use_cache=use_cache, position_embeddings=position_embeddings, **kwargs)
        cos, sin = position_embeddings
        a = core(layer.input_layernorm(hidden_states), past_key_values, cos, sin)
        h1 = ops.gemv(a, attn.o_proj.weight, attn.o_proj.bias, res=hidden_states.view(-1)).view(1, 1, -1)
        hid = ops.gemv_swiglu(layer.post_attention_layernorm(h1), mlp.gate_proj.weight, mlp.up_proj.weight)
        return ops.gemv(hid, mlp.down_proj.weight, None, res=h1.view(-1)).view(1, 1, -1)

    layer.forward = forward
    return 1


class PromptPrologue:
    """callable installed as `pipe.prompt_encoder`:
    (pipe, prompt=, negative_prompt=, edit_image=, cfg=, have_text_reasoning=) -> (posi, nega) dicts."""

    def __init__(self, text_encoder, processor, tokenizer=None, device="cuda", torch_dtype=torch.bfloat16, decode_gemv: bool = True,
                 decode_graph: bool = True):
        self.text_encoder = text_encoder
        self.decode_gemv = 0
        self.graph_decoder = None
        if decode_gemv and text_encoder is not None and torch.device(device).type == "cuda":
            self.decode_gemv = accelerate_decode(text_encoder)
            if decode_graph:
                try:
                    self.graph_decoder = GraphDecoder(text_encoder)
                except (ValueError, AttributeError) as e:            # another architecture / dtype: transformers' generate()
                    print(f"[prompt_prologue] captured decode step not available ({e}); using generate()")
        self.processor = processor
        self.tokenizer = tokenizer if tokenizer is not None else processor.tokenizer
        self.device = torch.device(device)
        self.torch_dtype = torch_dtype
        tok = processor.tokenizer
        if tok.convert_tokens_to_ids("<begin_of_img>") in (None, getattr(tok, "unk_token_id", None)):
            tok.add_special_tokens({"additional_special_tokens": special_tokens()})      # (:529-536)
        self.boi_token_id = tok.convert_tokens_to_ids("<begin_of_img>")
        self.eoi_token_id = tok.convert_tokens_to_ids("<end_of_img>")
        self.last_physical_txt: Optional[str] = None
        # (prompt, negative prompt, image bytes, cfg, reasoning) -> results of __call__.  The prologue is deterministic (greedy
        # decoding, no dropout), and its greedy decode can take as long as the whole denoising loop (profiles/r02_prologue.json),
        # so repeated edits of one (image, instruction) pair -- other seeds, step counts, sizes -- skip it.  0 disables.
        self.cache_size = 8
        self._cache: "collections.OrderedDict" = collections.OrderedDict()

    @staticmethod
    def _image_key(image) -> tuple:
        if image is None:
            return ()
        images = [image] if isinstance(image, Image.Image) else list(image)
        return tuple((im.size, im.mode, hashlib.sha1(im.tobytes()).hexdigest()) for im in images)

    # ---- text encoder calls -------------------------------------------------------------------------------
    def _last_hidden(self, model_inputs) -> torch.Tensor:
        """text_encoder.edit_forward(...)[-1] (qwen_image_text_encoder_withdecode.py:188-275): the inner model with
        output_hidden_states=True, last entry."""
        keys = ("input_ids", "attention_mask", "pixel_values", "image_grid_thw")
        kwargs = {k: model_inputs[k] for k in keys if k in model_inputs}
        out = self.text_encoder.model(**kwargs, output_attentions=False, output_hidden_states=True, return_dict=True)
        return out.hidden_states[-1]

    @staticmethod
    def _masked_rows(hidden_states: torch.Tensor, mask: torch.Tensor) -> List[torch.Tensor]:
        """extract_masked_hidden (:742-748)."""
        bool_mask = mask.bool()[:, :hidden_states.shape[1]]
        lengths = bool_mask.sum(dim=1)
        return list(torch.split(hidden_states[bool_mask], lengths.tolist(), dim=0))

    # ---- PhysicalVerbalEmbedder at inference (:943-967, :859-873) -------------------------------------------
    @torch.no_grad()
    def generate_ids(self, model_inputs, max_new_tokens: int):
        """`text_encoder.generate(**model_inputs, max_new_tokens=...)`: through the captured decode step when the encoder qualifies
        (bf16 on the GPU, greedy), through transformers otherwise."""
        gen = self.text_encoder.generation_config
        plain_greedy = (not getattr(gen, "do_sample", False) and (getattr(gen, "num_beams", 1) or 1) == 1
                        and (getattr(gen, "repetition_penalty", None) or 1.0) == 1.0
                        and not getattr(gen, "no_repeat_ngram_size", 0) and not getattr(gen, "bad_words_ids", None)
                        and not getattr(gen, "suppress_tokens", None) and not getattr(gen, "forced_eos_token_id", None))
        ids = model_inputs["input_ids"]
        if (self.graph_decoder is not None and plain_greedy and ids.shape[0] == 1
                and ids.shape[1] + max_new_tokens <= GraphDecoder.MAX_CACHE_ROWS):        # longer: the reference's own path
            return self.graph_decoder.generate(max_new_tokens=max_new_tokens, **model_inputs)
        return self.text_encoder.generate(**model_inputs, max_new_tokens=max_new_tokens)

    def physical_text(self, prompt: str, edit_image: Image.Image) -> str:
        messages = [
            {"role": "system", "content": SYSTEM_PROMPT_SAMPLE},
            {"role": "user", "content": [{"type": "input_text", "text": "Edit Instruction:"}, {"type": "input_text", "text": prompt},
                                         {"type": "input_text", "text": "Edit Image:"}, {"type": "image"}]},
        ]
        text = self.processor.apply_chat_template(messages, tokenize=False, add_generation_prompt=True, add_vision_id=True)
        model_inputs = self.processor(text=[text], images=resize_for_vl(edit_image), padding=True, return_tensors="pt").to(self.device)
        decoded_ids = self.generate_ids(model_inputs, max_new_tokens=1000)
        trimmed = [out_ids[len(in_ids):] for in_ids, out_ids in zip(model_inputs["input_ids"], decoded_ids)]
        decoded = self.tokenizer.batch_decode(trimmed, skip_special_tokens=True, clean_up_tokenization_spaces=False)[0]
        try:
            fields = parse_generation_response(decoded)
        except ValueError:
            return decoded
        return "".join(f"\n{k}: {v}" for k, v in fields.items())

    # ---- PromptEmbedder (:761-835) ----------------------------------------------------------------------------
    @torch.no_grad()
    def embed_entity(self, prompt: str) -> Dict[str, Optional[torch.Tensor]]:
        """QwenImageUnit_EntityControl.get_prompt_emb (:1136-1153): the text-only template, truncated at 1024 + 34 tokens."""
        return self.embed(prompt, None, None, max_length=1024 + T2I_DROP)

    def embed(self, prompt: str, edit_image=None, physical_txt: Optional[str] = None,
              max_length: int = 4096 + T2I_DROP) -> Dict[str, Optional[torch.Tensor]]:
        if physical_txt is not None:
            prompt = prompt + physical_txt
        special_token_mask = None
        if edit_image is None:
            txt = [T2I_TEMPLATE.format(prompt)]
            drop = T2I_DROP
            mi = self.tokenizer(txt, max_length=max_length, padding=True, truncation=True, return_tensors="pt").to(self.device)
            if mi["input_ids"].shape[1] >= 1024:
                print(f"Warning!!! QwenImage model was trained on prompts up to 512 tokens. Current prompt requires "
                      f"{mi['input_ids'].shape[1] - drop} tokens, which may lead to unpredictable behavior.")
        elif isinstance(edit_image, Image.Image):
            suffix = "\n<begin_of_img>" + "".join(f"<img{i}>" for i in range(SPECIAL_TOKEN_NUM)) + "<end_of_img><|im_end|>"
            txt = [EDIT_TEMPLATE.format(prompt + suffix)]
            drop = EDIT_DROP
            mi = self.processor(text=txt, images=resize_for_vl(edit_image), padding=True, return_tensors="pt").to(self.device)
            ids = mi["input_ids"]
            boi_pos = torch.where(ids == self.boi_token_id)[1]
            eoi_pos = torch.where(ids == self.eoi_token_id)[1]
            special_token_mask = torch.zeros_like(mi["attention_mask"], dtype=torch.bool)
            special_token_mask[:, int(boi_pos[0]) + 1:int(eoi_pos[0])] = True
            special_token_mask = special_token_mask[:, drop:]
        else:
            images = [resize_for_vl(im) for im in edit_image]
            pictures = "".join(PICTURE_TEMPLATE.format(i + 1) for i in range(len(images)))
            txt = [EDIT_MULTI_TEMPLATE.format(pictures + prompt)]
            drop = EDIT_DROP
            mi = self.processor(text=txt, images=images, padding=True, return_tensors="pt").to(self.device)
        hidden = self._last_hidden(mi)
        rows = [e[drop:] for e in self._masked_rows(hidden, mi["attention_mask"])]
        max_len = max(e.size(0) for e in rows)
        prompt_emb = torch.stack([torch.cat([u, u.new_zeros(max_len - u.size(0), u.size(1))]) for u in rows])
        emb_mask = torch.stack([torch.cat([torch.ones(e.size(0), dtype=torch.long, device=e.device),
                                           torch.zeros(max_len - e.size(0), dtype=torch.long, device=e.device)]) for e in rows])
        return {"prompt_emb": prompt_emb.to(dtype=self.torch_dtype, device=self.device), "prompt_emb_mask": emb_mask,
                "special_token_mask": special_token_mask}

    # ---- the unit runner's separate-CFG protocol (utils/__init__.py:247-283) ----------------------------------
    def __call__(self, pipe=None, prompt: str = "", negative_prompt: str = "", edit_image=None, cfg: bool = True,
                 have_text_reasoning: bool = True, physical_txt: Optional[str] = None) -> Tuple[Dict, Dict]:
        """`physical_txt`: reasoning text supplied by the caller instead of generated (PhysicalVerbalEmbedder.process, :976-983:
        the annotated-triplet branch, taken when the rules / key frames / input image of a dataset sample are all given)."""
        clone = lambda d: {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in d.items()}   # callers mutate prompt_emb
        key = (prompt, negative_prompt, self._image_key(edit_image), bool(cfg), bool(have_text_reasoning), physical_txt)
        given = physical_txt
        if self.cache_size and key in self._cache:
            self._cache.move_to_end(key)
            posi, nega, self.last_physical_txt = self._cache[key]
            return clone(posi), clone(nega)
        physical_txt = None
        if have_text_reasoning:
            if given is not None:
                physical_txt = given
            else:
                if not isinstance(edit_image, Image.Image):
                    raise ValueError("have_text_reasoning=True needs one edit image (encode_physical_prompt_sample, :943-967)")
                physical_txt = self.physical_text(prompt, edit_image)
        self.last_physical_txt = physical_txt
        posi = self.embed(prompt, edit_image, physical_txt)
        nega = self.embed(negative_prompt, edit_image, None) if cfg else dict(posi)
        if self.cache_size:
            self._cache[key] = (clone(posi), clone(nega), physical_txt)
            while len(self._cache) > self.cache_size:
                self._cache.popitem(last=False)
        return posi, nega
