"""CPU tests of the `diffsynth` façade's host plumbing (no kernel launches)."""
import inspect
import os

import pytest
import torch

BF = torch.bfloat16

# keyword surface of the reference's QwenImagePhysicPipeline.__call__ (qwen_image_physical.py:545-597)
REFERENCE_CALL_KWARGS = {
    "prompt": inspect.Parameter.empty, "negative_prompt": "", "cfg_scale": 4.0, "input_image": None,
    "denoising_strength": 1.0, "inpaint_mask": None, "inpaint_blur_size": None, "inpaint_blur_sigma": None,
    "height": 1328, "width": 1328, "seed": None, "rand_device": "cpu", "num_inference_steps": 30,
    "exponential_shift_mu": None, "blockwise_controlnet_inputs": None, "eligen_entity_prompts": None,
    "eligen_entity_masks": None, "eligen_enable_on_negative": False, "edit_image": None,
    "edit_image_auto_resize": True, "edit_rope_interpolation": False, "context_image": None,
    "enable_fp8_attention": False, "tiled": False, "tile_size": 128, "tile_stride": 64,
    "supported_rules": None, "contradicted_rules": None, "middle_key_frames": None, "stitched_image": None,
    "state": None, "transition": None, "triplet": None, "is_train": True, "have_text_reasoning": True,
}


def test_validate_py_imports_resolve():
    from diffsynth import load_state_dict  # noqa: F401  (validate.py:17)
    from diffsynth.pipelines.qwen_image_physical import QwenImagePhysicPipeline, ModelConfig  # noqa: F401  (:18)
    from diffsynth.pipelines.qwen_image import QwenImagePipeline  # noqa: F401
    from diffsynth import ModelManager, FlowMatchScheduler  # noqa: F401


def test_call_signature_matches_reference():
    from diffsynth.pipelines.qwen_image_physical import QwenImagePhysicPipeline
    sig = inspect.signature(QwenImagePhysicPipeline.__call__)
    for name, default in REFERENCE_CALL_KWARGS.items():
        assert name in sig.parameters, name
        assert sig.parameters[name].default == default or (default is inspect.Parameter.empty), name
    fp = inspect.signature(QwenImagePhysicPipeline.from_pretrained)
    assert list(fp.parameters) == ["torch_dtype", "device", "model_configs", "tokenizer_config", "processor_config", "dinov2_path"]
    ll = inspect.signature(QwenImagePhysicPipeline.load_lora)
    assert list(ll.parameters) == ["self", "module", "lora_config", "alpha", "hotload", "state_dict"]


def test_model_config_resolves_local_files(tmp_path):
    from diffsynth.pipelines.qwen_image_physical import ModelConfig
    d = tmp_path / "Qwen" / "Qwen-Image" / "vae"
    d.mkdir(parents=True)
    (d / "diffusion_pytorch_model.safetensors").write_bytes(b"")
    c = ModelConfig(model_id="Qwen/Qwen-Image", origin_file_pattern="vae/diffusion_pytorch_model.safetensors", local_model_path=str(tmp_path))
    c.download_if_necessary()
    assert c.path == str(d / "diffusion_pytorch_model.safetensors")
    t = tmp_path / "Qwen" / "Qwen-Image" / "tokenizer"
    t.mkdir()
    c = ModelConfig(model_id="Qwen/Qwen-Image", origin_file_pattern="tokenizer/", local_model_path=str(tmp_path))
    c.download_if_necessary()
    assert c.path == os.path.join(str(tmp_path), "Qwen/Qwen-Image", "tokenizer/")
    with pytest.raises(ValueError):
        ModelConfig().download_if_necessary()
    with pytest.raises(FileNotFoundError):
        ModelConfig(model_id="Qwen/None", origin_file_pattern="x*.safetensors", local_model_path=str(tmp_path)).download_if_necessary()


def test_load_state_dict_and_model_detection(tmp_path):
    from safetensors.torch import save_file
    from diffsynth import load_state_dict, ModelManager
    sd = {"transformer_blocks.0.img_mod.1.weight": torch.zeros((4, 4), dtype=BF), "x": torch.ones((2,))}
    p = str(tmp_path / "a.safetensors")
    save_file(sd, p)
    got = load_state_dict(p)
    assert set(got) == set(sd) and got["x"].dtype == torch.float32
    assert load_state_dict(p, torch_dtype=BF)["x"].dtype == BF
    mm = ModelManager()
    mm.load_model(p)
    assert mm.fetch_model("qwen_image_dit") is not None and mm.fetch_model("qwen_image_vae") is None


def test_pipeline_refuses_cpu_and_missing_pieces():
    from physicedit_amd._lib import PeError
    from diffsynth.pipelines.qwen_image_physical import QwenImagePhysicPipeline
    pipe = QwenImagePhysicPipeline(device="cuda")
    with pytest.raises(PeError):
        pipe("x", is_train=False)              # no weights loaded
    with pytest.raises(PeError):
        QwenImagePhysicPipeline(device="cuda", torch_dtype=torch.float16)
    assert pipe.check_resize_height_width(500, 512) == (512, 512)
    n1 = pipe.generate_noise((1, 16, 8, 8), seed=3, rand_torch_dtype=BF, device="cpu")
    from physicedit_amd import synth
    assert torch.equal(n1, synth.make_noise(3, 64, 64))


def test_layout_fingerprints_match_the_reference_detector_table():
    """The state-dict layouts the library is built around (physicedit_amd.synth: names + shapes) must BE the official
    checkpoints' layouts: their md5 fingerprints (the reference's hash_state_dict_keys, models/utils.py:148-182) equal
    the entries of the reference's detector table (configs/model_config.py:21,24)."""
    import torch
    from diffsynth.models.utils import hash_state_dict_keys
    from diffsynth.models.model_manager import detect_model_name
    from physicedit_amd import synth
    dit = {k: torch.empty(shape, device="meta") for k, shape in synth.dit_layout(60)}
    vae = {k: torch.empty(shape, device="meta") for k, shape in synth.vae_layout()}
    assert len(dit) == 1933
    assert hash_state_dict_keys(dit) == "0319a1cb19835fb510907dd3367c95ff"
    assert hash_state_dict_keys(vae) == "ed4ea5824d55ec3107b09815e318123a"
    assert detect_model_name(dit) == "qwen_image_dit" and detect_model_name(vae) == "qwen_image_vae"
    small = {k: torch.empty(shape, device="meta") for k, shape in synth.dit_layout(2)}
    assert detect_model_name(small) == "qwen_image_dit"          # reduced depth: key-signature fallback


def test_fp8_attention_flag_reaches_the_engine():
    """`enable_fp8_attention` is a real switch since round 4 (pe_flash_attn_fp8: the branch the reference takes where FlashAttention-3
    exists, qwen_image_dit.py:24-35): model_fn_qwen_image hands it to the engine's forward, DenoiseLoop to both CFG branches."""
    import inspect
    import physicedit_amd.dit as D
    import physicedit_amd.pipeline as P
    assert "enable_fp8_attention" in inspect.signature(D.QwenImageDiTEngine.forward).parameters
    assert "enable_fp8_attention" in inspect.signature(P.DenoiseLoop.__call__).parameters
    seen = {}

    class Eng:
        def forward(self, *a, **kw):
            seen.update(kw)
            return torch.zeros(1)
    D.model_fn_qwen_image(dit=Eng(), latents=torch.zeros(1, 16, 8, 8), timestep=torch.tensor([500.0]), prompt_emb=torch.zeros(1, 4, 3584),
                          is_train=False, enable_fp8_attention=True)
    assert seen["enable_fp8_attention"] is True
    assert not hasattr(D, "_warn_fp8_attention_once")


def test_G20_dino_preprocess_resize_and_crop():
    """The DINOv2 input preprocessing (qwen_image_physical.py:1043-1057) on fixed crop offsets against tests/golden G20: shorter edge
    to int(1.5 * 224) by the torchvision size rule, PIL bicubic, crop, ToTensor, ImageNet Normalize.  (torchvision's RandomCrop draw
    itself cannot be pinned: random by construction, and torchvision is not installed.)"""
    import os
    import sys
    import types
    from safetensors.torch import load_file
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import tiny_vl
    from diffsynth.pipelines.qwen_image_physical import QwenImagePhysicPipeline
    g = load_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G20_dino_preprocess.safetensors"))
    ns = types.SimpleNamespace(dino_input_size=224, device="cpu")
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    for name, (w, h, top, left) in {"landscape": (300, 200, 40, 100), "portrait": (180, 260, 77, 5)}.items():
        x = QwenImagePhysicPipeline.dino_input_preprocess(ns, [tiny_vl.make_image(w, h, 3)], crop_offsets=[(top, left)])
        assert x.shape == (1, 3, 224, 224)
        u8 = ((x[0].float() * std + mean) * 255).round().clamp(0, 255).to(torch.uint8).permute(1, 2, 0)
        assert torch.equal(u8, g[f"{name}.crop_u8"]), name
        assert torch.allclose(x[0].float()[:, ::16, ::16], g[f"{name}.normalized_sample"], atol=1e-6), name
    # an offset outside the resized frame is refused, and without offsets two calls draw different crops of the right size
    with pytest.raises(ValueError):
        QwenImagePhysicPipeline.dino_input_preprocess(ns, [tiny_vl.make_image(300, 200, 3)], crop_offsets=[(200, 0)])
    a = QwenImagePhysicPipeline.dino_input_preprocess(ns, [tiny_vl.make_image(300, 200, 3)])
    assert a.shape == (1, 3, 224, 224)
