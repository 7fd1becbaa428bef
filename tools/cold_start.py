#!/usr/bin/env python
"""Cold start of the hot path: from_pretrained on a synthetic, sharded checkpoint of the real size (60-layer DiT = 40.9 GB bf16 in 9
shards like the official `transformer/diffusion_pytorch_model-0000x-of-00009.safetensors`, plus the 254 MB VAE), then
validate.py's load_finetuned_into_pipe (720 rank-128 LoRA merges + adapter re-binding).  Prints one JSON object.

    python tools/cold_start.py [--layers 60] [--dir /tmp/pe_ckpt] [--keep]

Reference path being replaced: ModelManager.load_model (models/model_manager.py:350-384), GeneralLoRALoader.load
(lora/__init__.py:28-45)."""
import argparse
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=60)
    ap.add_argument("--shards", type=int, default=9)
    ap.add_argument("--dir", default="/tmp/pe_ckpt")
    ap.add_argument("--keep", action="store_true")
    a = ap.parse_args()
    import torch
    from safetensors.torch import save_file
    from physicedit_amd import synth
    from diffsynth.pipelines.qwen_image_physical import ModelConfig, QwenImagePhysicPipeline
    dev = torch.device("cuda", 0)
    out = {"layers": a.layers, "shards": a.shards}
    tdir = os.path.join(a.dir, "Qwen", "Qwen-Image-Edit-2509", "transformer")
    vdir = os.path.join(a.dir, "Qwen", "Qwen-Image", "vae")
    os.makedirs(tdir, exist_ok=True)
    os.makedirs(vdir, exist_ok=True)
    # ---- write the synthetic checkpoint (not timed as part of the cold start)
    t0 = time.perf_counter()
    layout = list(synth.dit_layout(a.layers))
    per = (len(layout) + a.shards - 1) // a.shards
    nbytes = 0
    for s in range(a.shards):
        part = layout[s * per:(s + 1) * per]
        if not part:
            continue
        sd = {k: v.cpu() for k, v in synth.make_state_dict_device(part, 1234, dev).items()}
        nbytes += sum(v.numel() * v.element_size() for v in sd.values())
        save_file(sd, os.path.join(tdir, f"diffusion_pytorch_model-{s + 1:05d}-of-{a.shards:05d}.safetensors"))
        del sd
    save_file(synth.make_state_dict(synth.vae_layout(), 77), os.path.join(vdir, "diffusion_pytorch_model.safetensors"))
    out["checkpoint_gb"] = nbytes / 1e9
    out["write_seconds"] = time.perf_counter() - t0
    torch.cuda.synchronize()

    def load():
        t = time.perf_counter()
        pipe = QwenImagePhysicPipeline.from_pretrained(
            torch_dtype=torch.bfloat16, device="cuda",
            model_configs=[ModelConfig(model_id="Qwen/Qwen-Image-Edit-2509", origin_file_pattern="transformer/diffusion_pytorch_model*.safetensors", local_model_path=a.dir),
                           ModelConfig(model_id="Qwen/Qwen-Image", origin_file_pattern="vae/diffusion_pytorch_model.safetensors", local_model_path=a.dir)],
            dinov2_path=None)
        torch.cuda.synchronize()
        return pipe, time.perf_counter() - t

    # page cache as the writer left it (warm), then -- if the box lets us -- after dropping it (cold from disk)
    pipe, out["from_pretrained_seconds_page_cache_warm"] = load()
    out["resident_gib"] = torch.cuda.memory_allocated() / 2 ** 30
    dropped = False
    try:
        os.sync()
        with open("/proc/sys/vm/drop_caches", "w") as f:
            f.write("3\n")
        dropped = True
    except OSError:
        pass
    if dropped:
        del pipe
        torch.cuda.empty_cache()
        pipe, out["from_pretrained_seconds_page_cache_dropped"] = load()
    out["ingest_gb_per_s_warm"] = out["checkpoint_gb"] / out["from_pretrained_seconds_page_cache_warm"]
    # ---- validate.py:load_finetuned_into_pipe: LoRA merge of 12 targets x layers at rank 128, adapter state
    # (the synthetic LoRA is generated and uploaded BEFORE the clock starts: a real run reads it from one safetensors file)
    base = {k: v.to(dev) for k, v in synth.make_lora(4321, 1, 128).items()}
    loras = [{k.replace("transformer_blocks.0.", f"transformer_blocks.{i}."): v for k, v in base.items()} for i in range(a.layers)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = sum(pipe.dit.load_lora(lora) for lora in loras)
    torch.cuda.synchronize()
    out["lora_merge_tensors"] = n
    out["lora_merge_seconds"] = time.perf_counter() - t0
    ad = synth.make_state_dict(synth.adapter_layout(), 4321)
    t0 = time.perf_counter()
    pipe.load_state_dict({"visual_thinking_adapter." + k: v for k, v in ad.items()}, strict=False)
    torch.cuda.synchronize()
    out["adapter_load_state_dict_seconds"] = time.perf_counter() - t0
    print(json.dumps(out))
    if not a.keep:
        shutil.rmtree(a.dir, ignore_errors=True)


if __name__ == "__main__":
    main()
