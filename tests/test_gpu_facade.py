"""`-m gpu`: BASELINE.json configs[0] (c1) through the `diffsynth` façade exactly as
scripts/inference/validate.py drives it -- from_pretrained on safetensors files, load_lora(pipe.dit, ...),
load_state_dict(strict=False) of the adapter, pipe(prompt, edit_image=, seed=, num_inference_steps=4, ...) --
with a stub prompt encoder (the text-encoder prologue is outside the hot path), against the oracle pipeline."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

import oracle.physicedit_oracle as O
from physicedit_amd import synth

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def test_c1_validate_style_run(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from safetensors.torch import save_file
    from diffsynth import load_state_dict
    from diffsynth.pipelines.qwen_image_physical import QwenImagePhysicPipeline, ModelConfig

    H = W = 512
    steps, T, nsp = 4, 128, 64
    dit_sd = synth.make_state_dict(synth.dit_layout(2), 1234)
    vae_sd = synth.make_state_dict(synth.vae_layout(), 77)
    ad_sd = synth.make_state_dict(synth.adapter_layout(), 4321)
    lora = synth.make_lora(4321, 2, 4)
    base = tmp_path / "base"
    (base / "Qwen/Qwen-Image-Edit-2509/transformer").mkdir(parents=True)
    (base / "Qwen/Qwen-Image/vae").mkdir(parents=True)
    keys = sorted(dit_sd)
    half = len(keys) // 2   # two shards, like the real checkpoint
    save_file({k: dit_sd[k] for k in keys[:half]}, str(base / "Qwen/Qwen-Image-Edit-2509/transformer/diffusion_pytorch_model-00001-of-00002.safetensors"))
    save_file({k: dit_sd[k] for k in keys[half:]}, str(base / "Qwen/Qwen-Image-Edit-2509/transformer/diffusion_pytorch_model-00002-of-00002.safetensors"))
    save_file(vae_sd, str(base / "Qwen/Qwen-Image/vae/diffusion_pytorch_model.safetensors"))
    ckpt = {**lora, **{"pipe.visual_thinking_adapter." + k: v for k, v in ad_sd.items()},
            "pipe.dino_time_embed.weight": torch.zeros((6, 768), dtype=BF)}
    save_file(ckpt, str(tmp_path / "finetuned.safetensors"))

    pipe = QwenImagePhysicPipeline.from_pretrained(
        torch_dtype=torch.bfloat16, device="cuda",
        model_configs=[
            ModelConfig(model_id="Qwen/Qwen-Image-Edit-2509", origin_file_pattern="transformer/diffusion_pytorch_model*.safetensors", local_model_path=str(base)),
            ModelConfig(model_id="Qwen/Qwen-Image", origin_file_pattern="vae/diffusion_pytorch_model.safetensors", local_model_path=str(base)),
        ], dinov2_path=None)
    # validate.py:load_finetuned_into_pipe
    full = load_state_dict(str(tmp_path / "finetuned.safetensors"))
    lora_state = {k: v for k, v in full.items() if "lora_A" in k or "lora_B" in k}
    pipe.load_lora(pipe.dit, state_dict=lora_state)
    rest = {k[len("pipe."):]: v for k, v in full.items() if k not in lora_state and k.startswith("pipe.")}
    pipe.load_state_dict(rest, strict=False)

    pe = synth.make_prompt_emb(7, T)
    mask = synth.make_special_token_mask(T, nsp)

    def stub_prompt_encoder(p, prompt, negative_prompt, edit_image, cfg, have_text_reasoning=True):
        return {"prompt_emb": pe.clone(), "special_token_mask": mask}, None
    pipe.prompt_encoder = stub_prompt_encoder

    img_u8 = synth.make_edit_image_u8(H, W, 0)
    out = pipe("turn the ice into water", edit_image=Image.fromarray(img_u8), seed=0, num_inference_steps=steps,
               height=H, width=W, cfg_scale=1.0, is_train=False, edit_image_auto_resize=False)
    assert isinstance(out, Image.Image) and out.size == (W, H)
    got = torch.from_numpy(np.array(out)).float()

    # ---- oracle pipeline on the same inputs (2-D conv form of the VAE: same math, 1/3 of the CPU time)
    assert O.lora_merge(dit_sd, lora) == 24
    O.VAE_CONV_MODE = "2d"
    try:
        edit_lat = O.vae_encode(vae_sd, O.preprocess_image(img_u8))
        lat = O.denoise_loop(dit_sd, ad_sd, synth.make_noise(0, H, W), pe, None, mask, None, H, W, steps, cfg_scale=1.0,
                             edit_latents=edit_lat)
        ref = O.vae_output_to_u8(O.vae_decode(vae_sd, lat)).float()
    finally:
        O.VAE_CONV_MODE = "3d"
    dl = (pipe.last_latents.float().cpu() - lat.float()).abs()
    d = (got - ref).abs()
    print(f"[parity] c1 façade run: latents max|d| {dl.max().item():.4e} mean|d| {dl.mean().item():.4e}; "
          f"uint8 image max|d| {d.max().item():.0f} mean|d| {d.mean().item():.3f} identical {(d == 0).float().mean().item()*100:.1f}%")
    assert dl.mean().item() <= 5e-3 and dl.max().item() <= 0.125
    assert d.mean().item() <= 0.6 and d.max().item() <= 12


def test_c3_fp8_computation_through_the_facade(tmp_path):
    """BASELINE.json configs[2] the way a reference user switches it on: the DiT checkpoint loaded with
    ModelConfig(offload_dtype=torch.float8_e4m3fn) + pipe.enable_vram_management(enable_dit_fp8_computation=True)
    (qwen_image_physical.py:440-496), hot-loaded LoRA, adapter; vs the oracle on the e4m3 state-dict
    (the e4m3 Linear itself is pinned against torch._scaled_mm in tests/test_gpu_fp8.py)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from safetensors.torch import save_file
    from diffsynth.pipelines.qwen_image_physical import QwenImagePhysicPipeline, ModelConfig

    H = W = 256
    steps, T, nsp = 2, 64, 16
    dit_sd = synth.make_state_dict(synth.dit_layout(2), 1234)
    vae_sd = synth.make_state_dict(synth.vae_layout(), 77)
    ad_sd = synth.make_state_dict(synth.adapter_layout(), 4321)
    lora = synth.make_lora(4321, 2, 8, std=0.05)
    base = tmp_path / "base"
    (base / "Qwen/Qwen-Image-Edit-2509/transformer").mkdir(parents=True)
    (base / "Qwen/Qwen-Image/vae").mkdir(parents=True)
    save_file(dit_sd, str(base / "Qwen/Qwen-Image-Edit-2509/transformer/diffusion_pytorch_model.safetensors"))
    save_file(vae_sd, str(base / "Qwen/Qwen-Image/vae/diffusion_pytorch_model.safetensors"))
    pipe = QwenImagePhysicPipeline.from_pretrained(
        torch_dtype=torch.bfloat16, device="cuda",
        model_configs=[
            ModelConfig(model_id="Qwen/Qwen-Image-Edit-2509", origin_file_pattern="transformer/diffusion_pytorch_model*.safetensors",
                        local_model_path=str(base), offload_dtype=torch.float8_e4m3fn),
            ModelConfig(model_id="Qwen/Qwen-Image", origin_file_pattern="vae/diffusion_pytorch_model.safetensors", local_model_path=str(base)),
        ], dinov2_path=None)
    assert not pipe.dit.fp8                                   # stored in e4m3, still computing in bf16
    pipe.enable_vram_management(enable_dit_fp8_computation=True)
    assert pipe.dit.fp8
    pipe.load_lora(pipe.dit, state_dict=lora, hotload=True)
    pipe.load_state_dict({"visual_thinking_adapter." + k: v for k, v in ad_sd.items()}, strict=False)
    assert pipe.dit.fp8 and pipe.dit.params["transformer_blocks.1.img_mlp.net.2.weight"].dtype == torch.float8_e4m3fn

    pe = synth.make_prompt_emb(7, T)
    mask = synth.make_special_token_mask(T, nsp)
    pipe.prompt_encoder = lambda p, prompt, negative_prompt, edit_image, cfg, have_text_reasoning=True: (
        {"prompt_emb": pe.clone(), "special_token_mask": mask}, None)
    img_u8 = synth.make_edit_image_u8(H, W, 0)
    out = pipe("let the candle burn down", edit_image=Image.fromarray(img_u8), seed=0, num_inference_steps=steps,
               height=H, width=W, cfg_scale=1.0, is_train=False, edit_image_auto_resize=False)
    assert isinstance(out, Image.Image) and out.size == (W, H)

    O.VAE_CONV_MODE = "2d"
    try:
        edit_lat = O.vae_encode(vae_sd, O.preprocess_image(img_u8))
        sd_hot = O.attach_hot_lora(dit_sd, lora)
        lat8 = O.denoise_loop(O.to_fp8_state_dict(sd_hot), ad_sd, synth.make_noise(0, H, W), pe.clone(), None, mask, None, H, W,
                              steps, cfg_scale=1.0, edit_latents=edit_lat)
        lat16 = O.denoise_loop(sd_hot, ad_sd, synth.make_noise(0, H, W), pe.clone(), None, mask, None, H, W, steps,
                               cfg_scale=1.0, edit_latents=edit_lat)
    finally:
        O.VAE_CONV_MODE = "3d"
    dl = (pipe.last_latents.float().cpu() - lat8.float()).abs()
    mode = (lat8.float() - lat16.float()).abs()
    print(f"[parity] c3 façade run (e4m3): latents vs oracle-e4m3 max|d| {dl.max().item():.4e} mean|d| {dl.mean().item():.4e}; "
          f"mode effect (oracle e4m3 vs bf16) mean|d| {mode.mean().item():.4e}")
    assert torch.isfinite(pipe.last_latents.float()).all()
    assert dl.mean().item() <= 0.25 * mode.mean().item()


def test_validate_flow_with_real_prologue(tmp_path):
    """validate.py's whole argument flow on the REAL kernels, prompt prologue included (no stub): from_pretrained with the five
    ModelConfigs of validate.py:97-125 (DiT, text encoder, VAE, tokenizer/, processor/), load_finetuned_into_pipe, then
    pipe(prompt, edit_image=, seed=, num_inference_steps=4, height=, width=, is_train=False) with its defaults (CFG 4.0,
    have_text_reasoning=True, auto-resized edit image).  The text encoder is the synthetic tiny Qwen2.5-VL of tests/tiny_vl.py
    (3584 wide).  The hot path is then checked against the oracle fed with the prologue's own embeddings."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from test_validate_flow_host import write_finetuned_ckpt, write_model_tree
    from diffsynth import load_state_dict
    import diffsynth.pipelines.qwen_image_physical as Q
    base = str(tmp_path / "models")
    te_cfg = write_model_tree(base)
    ckpt = str(tmp_path / "finetuned.safetensors")
    write_finetuned_ckpt(ckpt)
    Q.QwenImagePhysicPipeline.text_encoder_config = te_cfg
    try:
        pipe = Q.QwenImagePhysicPipeline.from_pretrained(
            torch_dtype=torch.bfloat16, device="cuda",
            model_configs=[
                Q.ModelConfig(model_id="Qwen/Qwen-Image-Edit-2509", origin_file_pattern="transformer/diffusion_pytorch_model*.safetensors", local_model_path=base),
                Q.ModelConfig(model_id="Qwen/Qwen-Image", origin_file_pattern="text_encoder/model*.safetensors", local_model_path=base),
                Q.ModelConfig(model_id="Qwen/Qwen-Image", origin_file_pattern="vae/diffusion_pytorch_model.safetensors", local_model_path=base),
            ],
            tokenizer_config=Q.ModelConfig(model_id="Qwen/Qwen-Image", origin_file_pattern="tokenizer/", local_model_path=base),
            processor_config=Q.ModelConfig(model_id="Qwen/Qwen-Image-Edit", origin_file_pattern="processor/", local_model_path=base),
            dinov2_path="unused")
    finally:
        Q.QwenImagePhysicPipeline.text_encoder_config = None
    assert pipe.prompt_encoder is not None and pipe.text_encoder is not None
    full = load_state_dict(ckpt)
    lora_state = {k: v for k, v in full.items() if "lora_A" in k or "lora_B" in k}
    pipe.load_lora(pipe.dit, state_dict=lora_state)
    pipe.load_state_dict({k[len("pipe."):]: v for k, v in full.items() if k not in lora_state and k.startswith("pipe.")}, strict=False)

    img = Image.fromarray(synth.make_edit_image_u8(128, 128, 5))
    prompt = "make the cup fall off the table"
    out = pipe(prompt, edit_image=img, seed=3, num_inference_steps=2, height=256, width=256, is_train=False)
    assert isinstance(out, Image.Image) and out.size == (256, 256)
    assert torch.isfinite(pipe.last_latents.float()).all()
    # the hot path vs the oracle on the prologue's own outputs (edit image auto-resized to 1024x1024 -> 4096 edit tokens)
    posi, nega = pipe.prompt_encoder(pipe, prompt=prompt, negative_prompt="", edit_image=Q.QwenImagePhysicPipeline._auto_resize(img),
                                     cfg=True, have_text_reasoning=True)
    assert posi["prompt_emb"].shape[2] == 3584 and int(posi["special_token_mask"].sum()) == 64
    dit_sd = synth.make_state_dict(synth.dit_layout(1), 1234)
    assert O.lora_merge(dit_sd, synth.make_lora(4321, 1, 4)) == 12
    ad_sd = synth.make_state_dict(synth.adapter_layout(), 4321)
    O.VAE_CONV_MODE = "2d"
    try:
        edit_u8 = np.array(Q.QwenImagePhysicPipeline._auto_resize(img))
        edit_lat = O.vae_encode(synth.make_state_dict(synth.vae_layout(), 77), O.preprocess_image(edit_u8))
    finally:
        O.VAE_CONV_MODE = "3d"
    lat = O.denoise_loop(dit_sd, ad_sd, synth.make_noise(3, 256, 256), posi["prompt_emb"].cpu(), nega["prompt_emb"].cpu(),
                         posi["special_token_mask"].cpu(), nega["special_token_mask"].cpu(), 256, 256, 2, cfg_scale=4.0,
                         edit_latents=edit_lat)
    dl = (pipe.last_latents.float().cpu() - lat.float()).abs()
    print(f"[parity] validate flow with the real prologue: latents max|d| {dl.max().item():.4e} mean|d| {dl.mean().item():.4e}")
    assert dl.mean().item() <= 5e-3 and dl.max().item() <= 0.125


def test_controlnet_and_eligen_through_the_facade(tmp_path):
    """The two secondary features of the same __call__ the way a reference user reaches them:
    a block-wise ControlNet checkpoint next to the DiT in from_pretrained (found by its key layout, qwen_image_physical.py:521),
    `blockwise_controlnet_inputs=[ControlNetInput(image=, inpaint_mask=, scale=)]`, `eligen_entity_prompts` / `eligen_entity_masks`
    -- against the oracle loop fed with the same conditioning latents / entity embeddings."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from safetensors.torch import save_file
    from diffsynth.pipelines.qwen_image_physical import ControlNetInput, ModelConfig, QwenImagePhysicPipeline

    H = W = 256
    steps, T, nsp = 3, 48, 16
    dit_sd = synth.make_state_dict(synth.dit_layout(2), 1234)
    vae_sd = synth.make_state_dict(synth.vae_layout(), 77)
    ad_sd = synth.make_state_dict(synth.adapter_layout(), 4321)
    cn_sd = synth.make_state_dict(synth.controlnet_layout(2, 4), 556)          # the inpaint layout (17-channel conditioning)
    base = tmp_path / "base"
    for sub in ("dit", "vae", "controlnet"):
        (base / "m" / sub).mkdir(parents=True)
    save_file(dit_sd, str(base / "m/dit/model.safetensors"))
    save_file(vae_sd, str(base / "m/vae/model.safetensors"))
    save_file(cn_sd, str(base / "m/controlnet/model.safetensors"))
    pipe = QwenImagePhysicPipeline.from_pretrained(
        torch_dtype=torch.bfloat16, device="cuda",
        model_configs=[ModelConfig(model_id="m", origin_file_pattern=f"{sub}/model.safetensors", local_model_path=str(base))
                       for sub in ("dit", "vae", "controlnet")], dinov2_path=None)
    assert pipe.blockwise_controlnet is not None and len(pipe.blockwise_controlnet.models) == 1
    pipe.load_state_dict({"visual_thinking_adapter." + k: v for k, v in ad_sd.items()}, strict=False)

    pe_p, mask_p = synth.make_prompt_emb(7, T), synth.make_special_token_mask(T, nsp)
    pe_n, mask_n = synth.make_prompt_emb(8, 24), synth.make_special_token_mask(24, nsp)
    ent_emb = {"a red ball": synth.make_prompt_emb(31, 10), "a glass table": synth.make_prompt_emb(32, 14)}

    class StubPrologue:                                    # the text encoder is not what this test is about
        def __call__(self, p, prompt, negative_prompt, edit_image, cfg, have_text_reasoning=True):
            return ({"prompt_emb": pe_p.clone(), "special_token_mask": mask_p}, {"prompt_emb": pe_n.clone(), "special_token_mask": mask_n})

        def embed_entity(self, prompt):
            return {"prompt_emb": ent_emb[prompt].clone()}
    pipe.prompt_encoder = StubPrologue()

    rs = np.random.RandomState(3)
    ctl_img = (rs.rand(H, W, 3) * 255).astype("uint8")
    inpaint = np.zeros((H, W, 3), dtype="uint8"); inpaint[64:160, 80:200] = 255
    ent_masks = []
    for y0, y1, x0, x1 in ((16, 120, 24, 140), (100, 240, 90, 250)):
        m = np.zeros((H, W, 3), dtype="uint8"); m[y0:y1, x0:x1] = 255
        ent_masks.append(m)
    cin = [ControlNetInput(controlnet_id=0, scale=0.6, start=1.0, end=0.4, image=Image.fromarray(ctl_img), inpaint_mask=Image.fromarray(inpaint))]
    out = pipe("put the ball on the table", seed=0, num_inference_steps=steps, height=H, width=W, cfg_scale=3.0, is_train=False,
               blockwise_controlnet_inputs=cin, eligen_entity_prompts=list(ent_emb), eligen_entity_masks=[Image.fromarray(m) for m in ent_masks],
               eligen_enable_on_negative=True)
    assert isinstance(out, Image.Image) and out.size == (W, H)

    # ---- oracle on the same inputs
    O.VAE_CONV_MODE = "2d"
    try:
        masked_img = O.controlnet_mask_on_image(ctl_img, inpaint)
        cond = O.controlnet_mask_on_latents(O.vae_encode(vae_sd, O.preprocess_image(masked_img)), inpaint)
    finally:
        O.VAE_CONV_MODE = "3d"
    emask = torch.stack([(O.preprocess_image(np.array(Image.fromarray(m).resize((W // 8, H // 8), resample=Image.NEAREST)))
                          .mean(dim=1, keepdim=True) > 0).to(BF)[0] for m in ent_masks]).unsqueeze(0)
    ents = list(ent_emb.values())
    tab = O.FlowMatchTables(steps, dynamic_shift_len=(H // 16) * (W // 16))
    t_min, t_max = O.adapter_t_range()
    ctl = [{"sd": cn_sd, "conditioning": cond, "scale": 0.6, "start": 1.0, "end": 0.4}]
    lat = synth.make_noise(0, H, W)
    pp, pn = pe_p.clone(), pe_n.clone()
    for pid, timestep in enumerate(tab.timesteps):
        t = timestep.unsqueeze(0).to(BF)
        kw = dict(controlnets=ctl, progress_id=pid, num_inference_steps=steps, entity_masks=emask)
        a = O.model_fn(dit_sd, ad_sd, lat, t, pp, mask_p, H, W, None, t_min, t_max, entity_prompt_emb=ents, **kw)
        b = O.model_fn(dit_sd, ad_sd, lat, t, pn, mask_n, H, W, None, t_min, t_max, entity_prompt_emb=[pn, pn], **kw)     # the SAME tensor, as :1177 hands it out
        lat = tab.step(b + 3.0 * (a - b), pid, lat)
    dl = (pipe.last_latents.float().cpu() - lat.float()).abs()
    print(f"[parity] facade controlnet + eligen: latents max|d| {dl.max().item():.4e} mean|d| {dl.mean().item():.4e} "
          f"(|latents| mean {lat.float().abs().mean().item():.3f})")
    assert dl.mean().item() <= 8e-3 and dl.max().item() <= 0.25


def test_inpaint_through_the_facade(tmp_path):
    """`input_image` + `denoising_strength` + `inpaint_mask` of the same __call__ (InputImageEmbedder :693-711, QwenImageUnit_Inpaint
    :714-729, BasePipeline.step utils/__init__.py:150-156) against the oracle loop on the same image, mask and prompt embeddings."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from safetensors.torch import save_file
    from diffsynth.pipelines.qwen_image_physical import ModelConfig, QwenImagePhysicPipeline

    H = W = 256
    steps, T, nsp, strength = 3, 48, 16, 0.6
    dit_sd = synth.make_state_dict(synth.dit_layout(2), 1234)
    vae_sd = synth.make_state_dict(synth.vae_layout(), 77)
    ad_sd = synth.make_state_dict(synth.adapter_layout(), 4321)
    base = tmp_path / "base"
    for sub in ("dit", "vae"):
        (base / "m" / sub).mkdir(parents=True)
    save_file(dit_sd, str(base / "m/dit/model.safetensors"))
    save_file(vae_sd, str(base / "m/vae/model.safetensors"))
    pipe = QwenImagePhysicPipeline.from_pretrained(
        torch_dtype=torch.bfloat16, device="cuda",
        model_configs=[ModelConfig(model_id="m", origin_file_pattern=f"{sub}/model.safetensors", local_model_path=str(base))
                       for sub in ("dit", "vae")], dinov2_path=None)
    pipe.load_state_dict({"visual_thinking_adapter." + k: v for k, v in ad_sd.items()}, strict=False)
    pe_p, mask_p = synth.make_prompt_emb(7, T), synth.make_special_token_mask(T, nsp)
    pe_n, mask_n = synth.make_prompt_emb(8, 24), synth.make_special_token_mask(24, nsp)

    class StubPrologue:
        def __call__(self, p, prompt, negative_prompt, edit_image, cfg, have_text_reasoning=True):
            return ({"prompt_emb": pe_p.clone(), "special_token_mask": mask_p}, {"prompt_emb": pe_n.clone(), "special_token_mask": mask_n})
    pipe.prompt_encoder = StubPrologue()

    img = synth.make_edit_image_u8(H, W, seed=21)
    yy, xx = np.mgrid[0:H, 0:W]
    m_u8 = (np.clip(1.5 - np.hypot(yy - 120, xx - 140) / 50.0, 0, 1) * 255).astype("uint8")
    out = pipe("fill the hole", seed=0, num_inference_steps=steps, height=H, width=W, cfg_scale=3.0, is_train=False,
               input_image=Image.fromarray(img), denoising_strength=strength, inpaint_mask=Image.fromarray(m_u8, mode="L"))
    assert isinstance(out, Image.Image) and out.size == (W, H)

    O.VAE_CONV_MODE = "2d"
    try:
        x0 = O.vae_encode(vae_sd, O.preprocess_image(img))
    finally:
        O.VAE_CONV_MODE = "3d"
    mask = O.inpaint_mask_plane(np.array(Image.fromarray(m_u8, mode="L").convert("RGB").resize((W // 8, H // 8))))
    tab = O.FlowMatchTables(steps, dynamic_shift_len=(H // 16) * (W // 16), denoising_strength=strength)
    start = tab.add_noise(x0, synth.make_noise(0, H, W), 0).to(BF)
    lat = O.denoise_loop(dit_sd, ad_sd, start, pe_p, pe_n, mask_p, mask_n, H, W, steps, cfg_scale=3.0, denoising_strength=strength,
                         input_latents=x0, inpaint_mask=mask)
    dl = (pipe.last_latents.float().cpu() - lat.float()).abs()
    print(f"[parity] facade inpaint: latents max|d| {dl.max().item():.4e} mean|d| {dl.mean().item():.4e} "
          f"(|latents| mean {lat.float().abs().mean().item():.3f})")
    assert dl.mean().item() <= 8e-3 and dl.max().item() <= 0.25


def test_accelerated_decode_matches_transformers():
    """accelerate_decode (prompt_prologue.py) swaps the decode step's single-row Linears, gated MLP, RMSNorms, q/k/v + rotary
    embedding and single-query attention for library kernels.  Two copies of a 4-layer model at the REAL widths (hidden 3584, 28 / 4
    heads of 128, MLP 18944), same seeded weights: the next-token logits after a prefill + 6 decode steps must agree to bf16 noise,
    far below the spread of the logits."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import copy
    from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration
    from diffsynth.pipelines import prompt_prologue as PP
    cfg = copy.deepcopy(PP.TEXT_ENCODER_CONFIG)
    cfg["text_config"].update(num_hidden_layers=4, vocab_size=4096, bos_token_id=1, eos_token_id=2, pad_token_id=0)
    cfg["vision_config"].update(depth=1, fullatt_block_indexes=[0])
    cfg.update(image_token_id=10, video_token_id=11, vision_start_token_id=12, vision_end_token_id=13, bos_token_id=1, eos_token_id=2)

    def build():
        old = torch.get_default_dtype()
        torch.set_default_dtype(torch.bfloat16)
        try:
            with torch.device("cuda"):
                m = Qwen2_5_VLForConditionalGeneration(Qwen2_5_VLConfig(**cfg))
        finally:
            torch.set_default_dtype(old)
        g = torch.Generator(device="cuda").manual_seed(5)
        with torch.no_grad():
            for name, p in m.named_parameters():
                if p.dim() >= 2:
                    p.normal_(0.0, 0.05, generator=g)
                elif "norm" in name:
                    p.fill_(1.0)
                else:
                    p.normal_(0.0, 0.02, generator=g)
        return m.eval()
    stock, fast = build(), build()
    assert PP.accelerate_decode(fast) >= 4 * (7 + 2 + 1 + 1)        # per layer: 7 Linears, 2 norms, the MLP, the attention
    ids = torch.randint(20, 4000, (1, 37), generator=torch.Generator().manual_seed(3)).cuda()

    def run(m):
        with torch.no_grad():
            out = m(input_ids=ids, use_cache=True)
            cache, logits = out.past_key_values, out.logits[:, -1]
            tok = ids[:, -1:]
            for step in range(6):                                   # teacher-forced: both models see the same tokens
                tok = (tok * 7 + 13 + step) % 3900 + 20
                out = m(input_ids=tok, past_key_values=cache, use_cache=True)
                cache, logits = out.past_key_values, out.logits[:, -1]
        return logits.float()
    a, b = run(stock), run(fast)
    spread = a.std().item()
    err = (a - b).abs().max().item()
    print(f"[parity] accelerated decode: max |dlogit| {err:.4e}, logit std {spread:.4e}, top-1 equal {bool(a.argmax() == b.argmax())}")
    assert torch.isfinite(b).all() and err <= 0.05 * spread


def test_graph_decoder_matches_generate():
    """GraphDecoder (the decode step captured in a hipGraph, KV cache in static planes, device-side step counter) against
    transformers' generate() on the same accelerated model: same kernels in the same order, so the greedy tokens must be IDENTICAL,
    with and without an EOS in the middle; text-only prompt (rope_delta 0) at the real widths, 4 layers."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import copy
    from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration
    from diffsynth.pipelines import prompt_prologue as PP
    cfg = copy.deepcopy(PP.TEXT_ENCODER_CONFIG)
    cfg["text_config"].update(num_hidden_layers=4, vocab_size=4096, bos_token_id=1, eos_token_id=2, pad_token_id=0)
    cfg["vision_config"].update(depth=1, fullatt_block_indexes=[0])
    cfg.update(image_token_id=10, video_token_id=11, vision_start_token_id=12, vision_end_token_id=13, bos_token_id=1, eos_token_id=2)
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device("cuda"):
            m = Qwen2_5_VLForConditionalGeneration(Qwen2_5_VLConfig(**cfg))
    finally:
        torch.set_default_dtype(old)
    g = torch.Generator(device="cuda").manual_seed(5)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if p.dim() >= 2:
                p.normal_(0.0, 0.05, generator=g)
            elif "norm" in name:
                p.fill_(1.0)
            else:
                p.normal_(0.0, 0.02, generator=g)
    m.eval()
    m.generation_config.do_sample = False
    m.generation_config.pad_token_id = 0
    m.generation_config.eos_token_id = None
    assert PP.accelerate_decode(m) > 0
    dec = PP.GraphDecoder(m, chunk=8)
    ids = torch.randint(20, 4000, (1, 37), generator=torch.Generator().manual_seed(3)).cuda()
    inputs = {"input_ids": ids, "attention_mask": torch.ones_like(ids)}
    with torch.no_grad():
        ref = m.generate(**inputs, max_new_tokens=40, min_new_tokens=40)
    got = dec.generate(max_new_tokens=40, **inputs)
    n_same = int((ref[0, 37:] == got[0, 37:]).sum()) if ref.shape == got.shape else -1
    print(f"[parity] graph decoder: {n_same} of 40 greedy tokens identical to generate(); shapes {tuple(ref.shape)} {tuple(got.shape)}")
    assert ref.shape == got.shape and torch.equal(ref, got)
    # an EOS in the middle: both must stop right after it (the chunked host-side test discards the overshoot)
    eos = int(ref[0, 37 + 13])
    first = (ref[0, 37:] == eos).nonzero()[0, 0].item()
    m.generation_config.eos_token_id = eos
    with torch.no_grad():
        ref2 = m.generate(**inputs, max_new_tokens=40)
    got2 = dec.generate(max_new_tokens=40, **inputs)
    assert ref2.shape[1] == 37 + first + 1 and torch.equal(ref2, got2)
    # another prompt length and token budget in the same capacity bucket: the captured step is REUSED (nothing in it depends on the
    # prompt: cache row and rotary row come from the device-side step counter), the tokens are still generate()'s
    ids3 = ids[:, :29].contiguous()
    m.generation_config.eos_token_id = None
    with torch.no_grad():
        ref3 = m.generate(input_ids=ids3, attention_mask=torch.ones_like(ids3), max_new_tokens=12, min_new_tokens=12)
    got3 = dec.generate(max_new_tokens=12, input_ids=ids3, attention_mask=torch.ones_like(ids3))
    assert torch.equal(ref3, got3)
    assert dec.captures == 1
    # a prompt that needs the next bucket: one more capture, stale rows of the first bucket's planes play no part
    ids4 = torch.randint(20, 4000, (1, 1030), generator=torch.Generator().manual_seed(4)).cuda()
    with torch.no_grad():
        ref4 = m.generate(input_ids=ids4, attention_mask=torch.ones_like(ids4), max_new_tokens=10, min_new_tokens=10)
    got4 = dec.generate(max_new_tokens=10, input_ids=ids4, attention_mask=torch.ones_like(ids4))
    assert torch.equal(ref4, got4) and dec.captures == 2
    got5 = dec.generate(max_new_tokens=40, **inputs)               # back to a short prompt: its bucket was dropped, captured again
    assert torch.equal(ref, got5)
    # a weight that MOVES between two calls (offload + reload, .to(), a LoRA merge that re-allocates).  The captured launches
    # hold the old address: the decoder must notice (fingerprint of the addresses), capture again and still produce generate()'s tokens --
    # here the old storage is overwritten with garbage first, so a stale replay could not pass by luck.
    n_cap = dec.captures
    lyr = m.model.language_model.layers[1]
    old_w = lyr.mlp.down_proj.weight.data
    lyr.mlp.down_proj.weight.data = old_w.clone()
    old_w.normal_(0.0, 5.0)
    got6 = dec.generate(max_new_tokens=40, **inputs)
    assert dec.captures == n_cap + 1 and torch.equal(ref, got6)
    got7 = dec.generate(max_new_tokens=40, **inputs)               # nothing moved: no further capture
    assert dec.captures == n_cap + 1 and torch.equal(ref, got7)
    dec.reset()                                                    # manual release: the next call builds planes and graph anew
    assert dec._static == {}
    assert torch.equal(ref, dec.generate(max_new_tokens=40, **inputs)) and dec.captures == n_cap + 2


def test_is_train_runs_the_visual_prior_through_the_facade(tmp_path):
    """`pipe(..., is_train=True, middle_key_frames=[...])` (the reference's default flag): the training-time unit
    QwenImageUnit_PhysicalVisualEmbedder (:992-1120) runs -- DINOv2 through transformers (a 2-layer random one of the real width here),
    everything after it on the library -- its targets match the oracle fed with the same encoder features, and the image is the
    one is_train=False produces (the reference discards the loss in __call__, :653)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from safetensors.torch import save_file
    from transformers import Dinov2WithRegistersConfig, Dinov2WithRegistersModel
    from diffsynth.pipelines.qwen_image_physical import ModelConfig, QwenImagePhysicPipeline

    H = W = 128
    steps, T, nsp = 2, 40, 16
    dit_sd = synth.make_state_dict(synth.dit_layout(2), 1234)
    vae_sd = synth.make_state_dict(synth.vae_layout(), 77)
    ad_sd = synth.make_state_dict(synth.adapter_layout(), 4321)
    prior_sd = synth.make_state_dict(synth.prior_layout(), 1818)
    base = tmp_path / "base"
    for sub in ("dit", "vae"):
        (base / "m" / sub).mkdir(parents=True)
    save_file(dit_sd, str(base / "m/dit/model.safetensors"))
    save_file(vae_sd, str(base / "m/vae/model.safetensors"))
    pipe = QwenImagePhysicPipeline.from_pretrained(
        torch_dtype=torch.bfloat16, device="cuda",
        model_configs=[ModelConfig(model_id="m", origin_file_pattern=f"{sub}/model.safetensors", local_model_path=str(base))
                       for sub in ("dit", "vae")], dinov2_path=None)
    pipe.load_state_dict({**{"visual_thinking_adapter." + k: v for k, v in ad_sd.items()}, **prior_sd}, strict=False)
    torch.manual_seed(3)
    enc = Dinov2WithRegistersModel(Dinov2WithRegistersConfig(hidden_size=768, num_hidden_layers=2, num_attention_heads=12, image_size=224,
                                                             patch_size=14, num_register_tokens=4))
    enc.layernorm.elementwise_affine, enc.layernorm.weight, enc.layernorm.bias = False, None, None
    pipe.dinov2 = enc.to(device="cuda", dtype=torch.bfloat16).eval().requires_grad_(False)
    pe_p, mask_p = synth.make_prompt_emb(7, T), synth.make_special_token_mask(T, nsp)
    pe_n, mask_n = synth.make_prompt_emb(8, 24), synth.make_special_token_mask(24, nsp)

    class StubPrologue:
        def __call__(self, p, prompt, negative_prompt, edit_image, cfg, have_text_reasoning=True, **kw):
            return ({"prompt_emb": pe_p.clone(), "special_token_mask": mask_p}, {"prompt_emb": pe_n.clone(), "special_token_mask": mask_n})
    pipe.prompt_encoder = StubPrologue()
    edit = Image.fromarray(synth.make_edit_image_u8(H, W, seed=31))
    frames = [Image.fromarray(synth.make_edit_image_u8(96, 160, seed=40 + i)) for i in range(3)]
    kw = dict(seed=0, num_inference_steps=steps, height=H, width=W, cfg_scale=3.0, edit_image=edit, edit_image_auto_resize=False,
              have_text_reasoning=False)
    with pytest.raises(Exception):
        pipe("tilt the glass", is_train=True, **kw)                      # no key frames
    torch.manual_seed(5)
    pipe("tilt the glass", is_train=True, middle_key_frames=frames, **kw)
    lat_train = pipe.last_latents.clone()
    pd, pv = pipe.last_pseudo_special_emb
    pipe("tilt the glass", is_train=False, **kw)
    assert torch.equal(lat_train, pipe.last_latents)
    # ---- oracle on the same encoder features (same crops: the same global RNG draws)
    torch.manual_seed(5)
    with torch.no_grad():
        dino_mid = pipe._dino_features(pipe.dino_input_preprocess(frames)).cpu()
        dino_src = pipe._dino_features(pipe.dino_input_preprocess([edit])).cpu()
    assert dino_mid.shape == (3, 256, 768)
    O.VAE_CONV_MODE = "2d"
    try:
        lat_mid = torch.cat([O.vae_encode(vae_sd, O.preprocess_image(np.array(f))) for f in frames])
        lat_src = O.vae_encode(vae_sd, O.preprocess_image(np.array(edit)))
    finally:
        O.VAE_CONV_MODE = "3d"
    rd, rv = O.visual_prior(prior_sd, dino_mid, dino_src, lat_mid, lat_src)
    for name, got, ref in (("pseudo_special_emb_dino", pd, rd), ("pseudo_special_emb_vae", pv, rv)):
        dd = (got.float().cpu() - ref.float()).abs()
        print(f"[parity] facade {name}: max|d| {dd.max().item():.3e} mean|d| {dd.mean().item():.3e} (|ref| mean {ref.float().abs().mean().item():.3e})")
        assert got.shape == (1, 64, 3584) and dd.mean().item() <= 0.06 * ref.float().abs().mean().item()   # a difference of two close vectors: 2.8 % measured
