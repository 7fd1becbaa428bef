"""The reference's scripts/inference/validate.py, UNCHANGED, through this repo's `diffsynth` façade -- on the CPU, with the
three GPU pieces (DiT engine, VAE, denoise loop) replaced by recording stand-ins, so that everything validate.py itself
touches is exercised for real: ModelConfig resolution of the checkpoint / tokenizer / processor paths, model detection,
text-encoder construction, the prompt prologue (tiny synthetic Qwen2.5-VL, tests/tiny_vl.py), load_finetuned_into_pipe
(LoRA keys -> load_lora, `pipe.*` keys -> load_state_dict(strict=False)), the __call__ keyword flow and the saved image.
The script text is read from /root/reference at test time (never copied); the -m gpu twin of this test
(tests/test_gpu_facade.py) runs the same flow on the real kernels without the script."""
import json
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image
from safetensors.torch import save_file

import tiny_vl
from physicedit_amd import synth

VALIDATE = "/root/reference/scripts/inference/validate.py"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BF = torch.bfloat16


def write_model_tree(base: str, hidden: int = 3584):
    """base/Qwen/... laid out like the real download: DiT (1 layer), VAE, tiny text encoder, tokenizer/, processor/."""
    def d(*p):
        path = os.path.join(base, *p)
        os.makedirs(path, exist_ok=True)
        return path
    save_file(synth.make_state_dict(synth.dit_layout(1), 1234), os.path.join(d("Qwen", "Qwen-Image-Edit-2509", "transformer"),
                                                                             "diffusion_pytorch_model-00001-of-00001.safetensors"))
    save_file(synth.make_state_dict(synth.vae_layout(), 77), os.path.join(d("Qwen", "Qwen-Image", "vae"), "diffusion_pytorch_model.safetensors"))
    tok_dir = d("Qwen", "Qwen-Image", "tokenizer")
    tok = tiny_vl.make_tokenizer(tok_dir)
    tok.save_pretrained(tok_dir)
    proc_dir = d("Qwen", "Qwen-Image-Edit", "processor")
    with open(os.path.join(proc_dir, "preprocessor_config.json"), "w") as f:
        json.dump({"do_resize": True, "do_rescale": True, "do_normalize": True, "do_convert_rgb": True, "resample": 3,
                   "image_mean": [0.48145466, 0.4578275, 0.40821073], "image_std": [0.26862954, 0.26130258, 0.27577711],
                   "min_pixels": 3136, "max_pixels": 12845056, "patch_size": 14, "temporal_patch_size": 2, "merge_size": 2}, f)
    with open(os.path.join(proc_dir, "chat_template.jinja"), "w") as f:
        f.write(tiny_vl.CHAT_TEMPLATE)
    te = tiny_vl.make_text_encoder(tok, hidden=hidden, extra_vocab=80)
    sd = {}
    for k, v in te.state_dict().items():          # transformers layout -> checkpoint layout (inverse of the loader's map)
        if k.startswith("model.language_model."):
            k = "model." + k[len("model.language_model."):]
        elif k.startswith("model.visual."):
            k = k[len("model."):]
        sd[k] = v.to(BF).contiguous()
    save_file(sd, os.path.join(d("Qwen", "Qwen-Image", "text_encoder"), "model-00001-of-00001.safetensors"))
    cfg = te.config.to_dict()
    return {"text_config": cfg["text_config"], "vision_config": cfg["vision_config"],
            **{k: cfg[k] for k in ("image_token_id", "video_token_id", "vision_start_token_id", "vision_end_token_id")}}


def write_finetuned_ckpt(path: str):
    """LoRA keys + `pipe.visual_thinking_adapter.*` (+ one training-only key), as train_physicedit.py saves them."""
    full = dict(synth.make_lora(4321, 1, 4))
    for k, v in synth.make_state_dict(synth.adapter_layout(), 4321).items():
        full["pipe.visual_thinking_adapter." + k] = v
    full["pipe.dino_time_embed.weight"] = torch.zeros((6, 768), dtype=BF)
    save_file({k: v.contiguous() for k, v in full.items()}, path)


class Recorder:
    calls = []


@pytest.mark.skipif(not os.path.exists(VALIDATE), reason="reference checkout not present")
def test_reference_validate_py_runs_unchanged_through_the_facade(tmp_path, monkeypatch):
    import diffsynth.pipelines.qwen_image_physical as Q
    te_cfg = write_model_tree(str(tmp_path / "models"))
    ckpt = str(tmp_path / "finetuned.safetensors")
    write_finetuned_ckpt(ckpt)
    img_path, out_path = str(tmp_path / "in.png"), str(tmp_path / "out" / "edited.png")
    Image.fromarray(synth.make_edit_image_u8(96, 160, 5)).save(img_path)       # H=96, W=160: exercises the ratio maths

    class FakeEngine:          # stands in for physicedit_amd.dit.QwenImageDiTEngine (needs a GPU)
        def __init__(self, sd, ad, device):
            Recorder.calls.append(("engine", len(sd), None if ad is None else len(ad)))
            self.fp8 = False

        def load_lora(self, sd, alpha=1.0, hotload=False):
            Recorder.calls.append(("load_lora", len(sd), alpha, hotload))
            return len(sd) // 2

        def set_adapter(self, ad):
            Recorder.calls.append(("set_adapter", len(ad)))

    class FakeVAE:
        def __init__(self, sd, device):
            Recorder.calls.append(("vae", len(sd)))

        def encode(self, x, **kw):
            Recorder.calls.append(("vae.encode", tuple(x.shape), str(x.dtype)))
            return torch.zeros((1, 16, x.shape[0] // 8, x.shape[1] // 8), dtype=BF)

        def decode(self, lat, output_u8=False, **kw):
            Recorder.calls.append(("vae.decode", tuple(lat.shape), output_u8))
            return torch.full((lat.shape[2] * 8, lat.shape[3] * 8, 3), 128, dtype=torch.uint8)

    class FakeLoop:
        def __init__(self, dit, dual_stream=False, cfg_pair=None):
            self.scheduler = None

        def __call__(self, latents, pe_p, pe_n, m_p, m_n, height, width, **kw):
            Recorder.calls.append(("loop", tuple(latents.shape), tuple(pe_p.shape), tuple(pe_n.shape), int(m_p.sum()), int(m_n.sum()),
                                   height, width, kw["num_inference_steps"], kw["cfg_scale"], [tuple(e.shape) for e in kw["edit_latents"]]))
            return latents

    # the reference asks the text encoder for up to 1000 new tokens (:965) and a randomly initialised tiny model never emits EOS:
    # 1000 CPU decode steps (~100 s) that say nothing about the flow under test.  Token-level parity of the prologue is G12's job.
    import diffsynth.pipelines.prompt_prologue as PP
    orig_generate_ids = PP.PromptPrologue.generate_ids
    monkeypatch.setattr(PP.PromptPrologue, "generate_ids",
                        lambda self, model_inputs, max_new_tokens: orig_generate_ids(self, model_inputs, min(max_new_tokens, 24)))
    monkeypatch.setattr(Q, "QwenImageDiTEngine", FakeEngine)
    monkeypatch.setattr(Q, "QwenImageVAE", FakeVAE)
    monkeypatch.setattr(Q, "DenoiseLoop", FakeLoop)
    monkeypatch.setattr(Q.QwenImagePhysicPipeline, "text_encoder_config", te_cfg)
    orig_init = Q.QwenImagePhysicPipeline.__init__
    monkeypatch.setattr(Q.QwenImagePhysicPipeline, "__init__",      # validate.py hard-codes device="cuda" (:96)
                        lambda self, device="cuda", torch_dtype=BF, dinov2_path=None: orig_init(self, "cpu", torch_dtype, dinov2_path))
    # deployment layout: <tree>/scripts/inference/validate.py next to <tree>/DiffSynth-Studio/diffsynth (this repo's façade)
    tree = tmp_path / "tree"
    (tree / "scripts" / "inference").mkdir(parents=True)
    os.symlink(ROOT, str(tree / "DiffSynth-Studio"))
    fake_file = str(tree / "scripts" / "inference" / "validate.py")
    src = open(VALIDATE).read()
    monkeypatch.setattr(sys, "argv", ["validate.py", "--prompt", "make the cup fall off the table", "--image_path", img_path,
                                      "--save_path", out_path, "--base_model_path", str(tmp_path / "models"), "--dinov2_path", "unused",
                                      "--lora_path", ckpt, "--seed", "3", "--num_inference_steps", "4"])
    Recorder.calls.clear()
    ns = {"__name__": "__main__", "__file__": fake_file}
    exec(compile(src, fake_file, "exec"), ns)          # the reference's script, byte for byte
    assert os.path.exists(out_path)
    out = Image.open(out_path)
    kinds = [c[0] for c in Recorder.calls]
    assert kinds.count("engine") >= 1 and "vae" in kinds and "load_lora" in kinds
    lora = [c for c in Recorder.calls if c[0] == "load_lora"][0]
    assert lora[1] == 24 and lora[2] == 1.0 and lora[3] is False           # 12 targets x (A, B), merged (validate.py:52)
    assert kinds.count("engine") == 1                                      # the 41 GB engine is built ONCE ...
    assert [c for c in Recorder.calls if c[0] == "set_adapter"] == [("set_adapter", 8)]   # ... load_state_dict only re-binds the adapter
    loop = [c for c in Recorder.calls if c[0] == "loop"][0]
    # resize_image: 160x96 -> area ~1024^2, /32: width 1312, height 800 (validate.py:20-31)
    assert (loop[6], loop[7]) == (800, 1312) and out.size == (1312, 800)
    assert loop[1] == (1, 16, 100, 164) and loop[8] == 4 and loop[9] == 4.0
    assert loop[2][2] == 3584 and loop[3][2] == 3584 and loop[2][1] > loop[3][1]   # posi prompt + physical text is longer than ""
    assert loop[4] == 64 and loop[5] == 64                                  # 64 special tokens in both branches
    assert loop[10] == [(1, 16, 100, 164)]                                  # edit image auto-resized to the same area
    enc = [c for c in Recorder.calls if c[0] == "vae.encode"][0]
    assert enc[2] == "torch.uint8"                                          # fused uint8 path
