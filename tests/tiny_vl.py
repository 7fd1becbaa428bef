"""Test helper: a SYNTHETIC stand-in for the Qwen2.5-VL prompt stack, small enough for CPU tests and buildable anywhere
(no checkpoint, no vocabulary files):
  * a byte-level Qwen2Tokenizer whose vocabulary is the 256 byte symbols plus Qwen's control tokens (no merges: one token
    per byte), written to a temp directory at test time;
  * the repo's MiniQwen2VLProcessor over transformers' Qwen2VLImageProcessor with the Qwen-Image-Edit pre-processing
    constants (preprocessor_config.json: data);
  * a randomly initialised (seeded) Qwen2_5_VLForConditionalGeneration with 2 layers.
Used by tests/golden/make_golden.py (through the reference's units) and by the prologue tests (through the repo's)."""
import json
import os

import torch

CONTROL_TOKENS = ["<|endoftext|>", "<|im_start|>", "<|im_end|>", "<|object_ref_start|>", "<|object_ref_end|>", "<|box_start|>",
                  "<|box_end|>", "<|quad_start|>", "<|quad_end|>", "<|vision_start|>", "<|vision_end|>", "<|vision_pad|>",
                  "<|image_pad|>", "<|video_pad|>"]

CHAT_TEMPLATE = (
    "{% set image_count = namespace(value=0) %}{% set video_count = namespace(value=0) %}{% for message in messages %}"
    "{% if loop.first and message['role'] != 'system' %}<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n{% endif %}"
    "<|im_start|>{{ message['role'] }}\n{% if message['content'] is string %}{{ message['content'] }}<|im_end|>\n{% else %}"
    "{% for content in message['content'] %}{% if content['type'] == 'image' or 'image' in content or 'image_url' in content %}"
    "{% set image_count.value = image_count.value + 1 %}{% if add_vision_id %}Picture {{ image_count.value }}: {% endif %}"
    "<|vision_start|><|image_pad|><|vision_end|>{% elif content['type'] == 'video' or 'video' in content %}"
    "{% set video_count.value = video_count.value + 1 %}{% if add_vision_id %}Video {{ video_count.value }}: {% endif %}"
    "<|vision_start|><|video_pad|><|vision_end|>{% elif 'text' in content %}{{ content['text'] }}{% endif %}{% endfor %}<|im_end|>\n"
    "{% endif %}{% endfor %}{% if add_generation_prompt %}<|im_start|>assistant\n{% endif %}")


def prologue_module():
    """The repo's diffsynth/pipelines/prompt_prologue.py loaded BY PATH: in tests/golden/make_golden.py the name `diffsynth`
    is the reference's package."""
    import importlib.util
    import sys
    name = "pe_prompt_prologue"
    if name not in sys.modules:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        spec = importlib.util.spec_from_file_location(name, os.path.join(root, "diffsynth", "pipelines", "prompt_prologue.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    return sys.modules[name]


def _bytes_to_unicode():
    """GPT-2's printable stand-ins for the 256 byte values (the alphabet of every byte-level BPE vocabulary)."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def make_tokenizer(tmpdir: str):
    from transformers import Qwen2Tokenizer
    os.makedirs(tmpdir, exist_ok=True)
    vocab = {ch: i for i, ch in enumerate(_bytes_to_unicode().values())}
    with open(os.path.join(tmpdir, "vocab.json"), "w", encoding="utf-8") as f:
        json.dump(vocab, f, ensure_ascii=False)
    with open(os.path.join(tmpdir, "merges.txt"), "w", encoding="utf-8") as f:
        f.write("#version: 0.2\n")
    tok = Qwen2Tokenizer(os.path.join(tmpdir, "vocab.json"), os.path.join(tmpdir, "merges.txt"), unk_token="<|endoftext|>",
                         eos_token="<|im_end|>", pad_token="<|endoftext|>")
    tok.add_special_tokens({"additional_special_tokens": CONTROL_TOKENS})
    return tok


def make_processor(tmpdir: str):
    from transformers import Qwen2VLImageProcessor
    MiniQwen2VLProcessor = prologue_module().MiniQwen2VLProcessor
    ip = Qwen2VLImageProcessor(do_resize=True, do_rescale=True, do_normalize=True, do_convert_rgb=True, resample=3,
                               image_mean=[0.48145466, 0.4578275, 0.40821073], image_std=[0.26862954, 0.26130258, 0.27577711],
                               min_pixels=3136, max_pixels=12845056, patch_size=14, temporal_patch_size=2, merge_size=2)
    return MiniQwen2VLProcessor(ip, make_tokenizer(tmpdir), CHAT_TEMPLATE)


def make_text_encoder(tokenizer, hidden: int = 64, seed: int = 1234, extra_vocab: int = 80):
    """2-layer Qwen2.5-VL with the token ids of `tokenizer`; `extra_vocab` leaves room for the 66 tokens the pipeline adds."""
    from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration
    tid = tokenizer.convert_tokens_to_ids
    heads = max(hidden // 16, 2)
    cfg = Qwen2_5_VLConfig(
        text_config=dict(hidden_size=hidden, intermediate_size=2 * hidden, num_hidden_layers=2, num_attention_heads=heads,
                         num_key_value_heads=2, vocab_size=len(tokenizer) + extra_vocab, max_position_embeddings=32768,
                         rms_norm_eps=1e-6, rope_theta=1000000.0, tie_word_embeddings=False,
                         rope_scaling={"type": "default", "rope_type": "default", "mrope_section": [2, 3, 3]},
                         bos_token_id=tid("<|endoftext|>"), eos_token_id=tid("<|im_end|>"), pad_token_id=tid("<|endoftext|>")),
        vision_config=dict(depth=2, hidden_size=32, intermediate_size=64, num_heads=2, out_hidden_size=hidden, patch_size=14,
                           spatial_merge_size=2, temporal_patch_size=2, window_size=112, fullatt_block_indexes=[1], in_chans=3),
        image_token_id=tid("<|image_pad|>"), video_token_id=tid("<|video_pad|>"), vision_start_token_id=tid("<|vision_start|>"),
        vision_end_token_id=tid("<|vision_end|>"), bos_token_id=tid("<|endoftext|>"), eos_token_id=tid("<|im_end|>"),
        pad_token_id=tid("<|endoftext|>"))
    torch.manual_seed(seed)
    model = Qwen2_5_VLForConditionalGeneration(cfg).eval()
    model.generation_config.do_sample = False            # greedy: the fixture must be reproducible
    model.generation_config.eos_token_id = tid("<|im_end|>")
    model.generation_config.pad_token_id = tid("<|endoftext|>")
    return model


def make_image(w: int, h: int, seed: int):
    import numpy as np
    from PIL import Image
    return Image.fromarray((np.random.RandomState(seed).rand(h, w, 3) * 255).astype("uint8"))
