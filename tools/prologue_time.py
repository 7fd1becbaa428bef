#!/usr/bin/env python
"""Times the prompt prologue at its REAL size on the GPU, next to the denoising loop it feeds:
the Qwen2.5-VL-7B text encoder of Qwen-Image-Edit (diffsynth/pipelines/prompt_prologue.py: stock `transformers` code on
PyTorch-ROCm, not part of the HIP library), random weights, synthetic byte-level tokenizer (tests/tiny_vl.py; a real BPE
vocabulary would give ~4x fewer tokens for the same text, which is why the prompt below is sized in TOKENS).

    python tools/prologue_time.py [--decode-tokens 128] > profiles/r02_prologue.json

Reports: model build time, the two embedding passes of an edit (positive prompt + generated physical text; negative prompt), and
the greedy decode rate of `generate` (the reference asks for up to max_new_tokens=1000, qwen_image_physical.py:868); with
random weights EOS never comes, so the rate is measured over a fixed number of new tokens and the 1000-token worst case is
extrapolated."""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--decode-tokens", type=int, default=128)
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--no-gemv", action="store_true", help="leave the single-row Linears of the decode on torch's BLAS GEMV")
    ap.add_argument("--quick", action="store_true", help="what bench.py runs: embedding passes, prefill, captured decode step only")
    args = ap.parse_args()
    print(json.dumps(measure(args.decode_tokens, args.layers, args.no_gemv, args.quick), indent=1))


def streamed_bytes_per_token(model):
    """weight bytes one decode step reads: every Linear of the language model's layers + lm_head (the embedding table contributes one
    row, the vision tower nothing)"""
    n = 0
    for name, p in model.named_parameters():
        if "visual" in name or "embed_tokens" in name:
            continue
        n += p.numel() * p.element_size()
    return n


def measure(decode_tokens=128, layers=28, no_gemv=False, quick=False):
    """-> dict (see the module docstring).  quick: skips the stock-transformers generate() timings (bench.py's `prologue` block)."""
    import types
    args = types.SimpleNamespace(decode_tokens=decode_tokens, layers=layers, no_gemv=no_gemv)
    import torch
    import tiny_vl
    from diffsynth.pipelines import prompt_prologue as pp

    dev = "cuda"
    tmp = tempfile.mkdtemp(prefix="pe_prologue_")
    processor = tiny_vl.make_processor(tmp)
    tok = processor.tokenizer
    tok.add_special_tokens({"additional_special_tokens": pp.special_tokens()})
    tid = tok.convert_tokens_to_ids
    cfg = json.loads(json.dumps(pp.TEXT_ENCODER_CONFIG))
    cfg["text_config"]["num_hidden_layers"] = args.layers
    for key, name in (("image_token_id", "<|image_pad|>"), ("video_token_id", "<|video_pad|>"),
                      ("vision_start_token_id", "<|vision_start|>"), ("vision_end_token_id", "<|vision_end|>")):
        cfg[key] = tid(name)
    for scope in (cfg, cfg["text_config"]):
        scope["bos_token_id"], scope["eos_token_id"] = tid("<|endoftext|>"), tid("<|im_end|>")
    cfg["text_config"]["pad_token_id"] = tid("<|endoftext|>")

    from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration
    t0 = time.perf_counter()
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device(dev):
            model = Qwen2_5_VLForConditionalGeneration(Qwen2_5_VLConfig(**cfg))
    finally:
        torch.set_default_dtype(old)
    g = torch.Generator(device=dev).manual_seed(0)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() >= 2:
                p.normal_(0.0, 0.02, generator=g)
            elif "norm" in name or name.endswith("ln_q.weight"):
                p.fill_(1.0)
            else:
                p.zero_()
    model.eval()
    model.generation_config.do_sample = False
    model.generation_config.eos_token_id = None          # random weights: decode a fixed number of tokens
    model.generation_config.pad_token_id = tid("<|endoftext|>")
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    n_params = sum(p.numel() for p in model.parameters())

    image = tiny_vl.make_image(1024, 1024, 0)
    # greedy tokens of the UNPATCHED model first (the patch below replaces forwards in place): a sanity check of the decode kernels
    check_text = processor.apply_chat_template([{"role": "user", "content": [{"type": "input_text", "text": "what happens next?"},
                                                                             {"type": "image"}]}], tokenize=False,
                                               add_generation_prompt=True, add_vision_id=True)
    check_in = processor(text=[check_text], images=pp.resize_for_vl(image), padding=True, return_tensors="pt").to(dev)
    with torch.no_grad():
        stock_tokens = model.generate(**check_in, max_new_tokens=48, min_new_tokens=48)[0, check_in["input_ids"].shape[1]:].tolist()
    prologue = pp.PromptPrologue(model, processor, device=dev, decode_gemv=not args.no_gemv)
    with torch.no_grad():
        fast_tokens = model.generate(**check_in, max_new_tokens=48, min_new_tokens=48)[0, check_in["input_ids"].shape[1]:].tolist()
    same_prefix = next((i for i, (a, b) in enumerate(zip(stock_tokens, fast_tokens)) if a != b), len(stock_tokens))
    prompt = "push the red ball off the table " * 3                       # ~100 byte-level tokens
    physical = "\nReasoning: " + "the ball rolls to the edge, tips over and falls under gravity. " * 4    # ~270 tokens

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
        return sorted(ts)[len(ts) // 2], out

    t_posi, posi = timed(lambda: prologue.embed(prompt, image, physical))
    t_nega, nega = timed(lambda: prologue.embed("", image, None))

    # the generate() call of physical_text(), with a fixed number of new tokens
    messages = [{"role": "system", "content": pp.SYSTEM_PROMPT_SAMPLE},
                {"role": "user", "content": [{"type": "input_text", "text": "Edit Instruction:"}, {"type": "input_text", "text": prompt},
                                             {"type": "input_text", "text": "Edit Image:"}, {"type": "image"}]}]
    text = processor.apply_chat_template(messages, tokenize=False, add_generation_prompt=True, add_vision_id=True)
    mi = processor(text=[text], images=pp.resize_for_vl(image), padding=True, return_tensors="pt").to(dev)
    n_new = args.decode_tokens

    def gen(n):
        with torch.no_grad():
            return model.generate(**mi, max_new_tokens=n, min_new_tokens=n)

    if quick:
        t_gen1 = t_genn = rate = None
        out = None
    else:
        t_gen1, _ = timed(lambda: gen(1), reps=2)             # prefill + 1 token
        t_genn, out = timed(lambda: gen(n_new), reps=2)
        rate = (n_new - 1) / max(t_genn - t_gen1, 1e-9)
    bytes_tok = streamed_bytes_per_token(model)
    graph = None
    if prologue.graph_decoder is not None:
        # the path physical_text() takes: prefill on transformers, then the captured decode step (GraphDecoder)
        def ggen(n):
            return prologue.generate_ids(mi, n)
        tg1, _ = timed(lambda: ggen(1), reps=2)
        tgn, gout = timed(lambda: ggen(n_new), reps=2)
        grate = (n_new - 1) / max(tgn - tg1, 1e-9)
        same = None if out is None else int((gout[0, -n_new:] == out[0, -n_new:]).sum())
        import hashlib
        graph = {"prefill_plus_1_token_seconds": round(tg1, 4), "seconds": round(tgn, 3), "decode_tokens_per_second": round(grate, 1),
                 "weight_bytes_streamed_per_token": bytes_tok, "achieved_TBps": round(bytes_tok * grate / 1e12, 3),
                 "frac_of_hbm_peak_8TBps": round(bytes_tok * grate / 8e12, 4),
                 "includes": "graph capture (once per (cache length, max_new_tokens)) + KV-cache copy into the static planes",
                 "tokens_identical_to_generate()": None if same is None else f"{same} of {n_new}",
                 "token_ids_sha1": hashlib.sha1(gout[0, -n_new:].cpu().numpy().tobytes()).hexdigest()[:16],
                 "extrapolated_seconds_for_1000_new_tokens": round(tg1 + 999 / grate, 1)}
    res = {
        "what": "prompt prologue at real size (Qwen2.5-VL-7B architecture, random weights), stock transformers on PyTorch-ROCm",
        "parameters_billion": round(n_params / 1e9, 3), "layers": args.layers, "build_seconds": round(build_s, 2),
        "single_row_linears_on_pe_gemv_bf16": prologue.decode_gemv,
        "greedy_tokens_identical_to_unpatched_model": f"{same_prefix} of {len(stock_tokens)} (random weights: near-uniform logits, the "
                                                      "hardest case for a tie-break)",
        "embed_positive": {"tokens_after_drop": int(posi["prompt_emb"].shape[1]), "seconds": round(t_posi, 4)},
        "embed_negative": {"tokens_after_drop": int(nega["prompt_emb"].shape[1]), "seconds": round(t_nega, 4)},
        "generate": None if quick else {"prompt_tokens": int(mi["input_ids"].shape[1]), "prefill_plus_1_token_seconds": round(t_gen1, 4),
                                        "new_tokens_timed": n_new, "seconds": round(t_genn, 3), "decode_tokens_per_second": round(rate, 1),
                                        "extrapolated_seconds_for_1000_new_tokens": round(t_gen1 + 999 / rate, 1)},
        "prompt_tokens": int(mi["input_ids"].shape[1]), "new_tokens_timed": n_new,
        "generate_captured_decode_step": graph,
        "torch": torch.__version__, "device": torch.cuda.get_device_name(0),
    }
    del model, prologue
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    main()
