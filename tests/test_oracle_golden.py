"""Pin oracle/physicedit_oracle.py against outputs of the imported reference (tests/golden/*,
written by tests/golden/make_golden.py).  CPU only.  Bit-exact unless a test states otherwise."""
import json
import os

import numpy as np
import pytest
import torch

import oracle.physicedit_oracle as O
from physicedit_amd import synth

BF = torch.bfloat16
HERE = os.path.dirname(os.path.abspath(__file__))


def assert_same(a, b, what=""):
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert a.dtype == b.dtype, (what, a.dtype, b.dtype)
    if not torch.equal(a, b):
        d = (a.float() - b.float()).abs()
        raise AssertionError(f"{what}: {int((d > 0).sum())}/{d.numel()} differ, max {d.max().item():.3e}")


def test_layout_matches_reference():
    with open(os.path.join(HERE, "golden", "layout_keys.json")) as f:
        ref = json.load(f)
    assert [[k, list(s)] for k, s in synth.dit_layout(1)] == ref["dit_1layer"]
    assert [[k, list(s)] for k, s in synth.vae_layout()] == ref["vae"]
    assert [[k, list(s)] for k, s in synth.adapter_layout()] == ref["adapter"]


def test_G1_scheduler(golden):
    g = golden("G1_scheduler")
    assert_same(O.FlowMatchTables(100).timesteps, g["default_timesteps"], "default timesteps")
    for steps, S0 in ((4, 1024), (4, 64), (40, 4096), (50, 6889)):
        tab = O.FlowMatchTables(steps, dynamic_shift_len=S0)
        assert_same(tab.sigmas, g[f"sigmas_{steps}_{S0}"], "sigmas")
        assert_same(tab.timesteps, g[f"timesteps_{steps}_{S0}"], "timesteps")
        assert_same(tab.timesteps.to(BF), g[f"timesteps_bf16_{steps}_{S0}"], "timesteps bf16")
        gen = torch.Generator().manual_seed(11)
        x = torch.randn((1, 16, 8, 8), generator=gen).to(BF)
        v = torch.randn((1, 16, 8, 8), generator=gen).to(BF)
        outs = torch.stack([tab.step(v, i, x) for i in range(steps)])
        assert_same(outs, g[f"step_{steps}_{S0}"], "euler step")
    tab = O.FlowMatchTables(40, dynamic_shift_len=4096)
    assert abs(tab.mu - 0.693548) < 1e-6
    assert abs(tab.timesteps[-1].item() - 20.0) < 1e-3


def test_G2_time_embed(golden):
    g = golden("G2_time_embed")
    sd = synth.make_state_dict([kv for kv in synth.dit_layout(0) if kv[0].startswith("time_text_embed")], 1234)
    ts = O.FlowMatchTables(40, dynamic_shift_len=4096).timesteps.to(BF)
    sin = torch.cat([O.timestep_sinusoid(ts[i:i + 1] / 1000) for i in range(40)])
    assert_same(sin, g["sinusoid_f32"], "sinusoid")
    temb = torch.cat([O.time_text_embed(sd, ts[i:i + 1] / 1000, BF) for i in range(40)])
    assert_same(temb, g["temb"], "temb")
    # batched == per-row (the HIP path hoists all timesteps into one GEMM)
    temb_b = O.time_text_embed(sd, ts / 1000, BF)
    assert (temb_b.float() - temb.float()).abs().max() <= 2 ** -6


def test_G3_norm_rope(golden):
    g = golden("G3_norm_rope")
    assert_same(O.rmsnorm(g["rms_in"], g["rms_w"]), g["rms_out"], "rmsnorm128")
    vid, txt = O.rope_tables([(1, 8, 8), (1, 6, 10)], 37)
    assert_same(vid.real.contiguous(), g["vid_re"], "vid re")
    assert_same(vid.imag.contiguous(), g["vid_im"], "vid im")
    assert_same(txt.real.contiguous(), g["txt_re"], "txt re")
    assert_same(txt.imag.contiguous(), g["txt_im"], "txt im")
    assert_same(O.apply_rope(g["rms_in"], txt), g["rope_out"], "rope apply")
    vid2, txt2 = O.rope_tables([(1, 64, 64), (1, 64, 64)], 512)
    assert_same(torch.view_as_real(vid2[[0, 63, 64, 4095, 4096, 8191]]).contiguous(), g["vid64_rows"], "vid64 rows")
    assert_same(torch.view_as_real(txt2[[0, 1, 511]]).contiguous(), g["txt64_rows"], "txt64 rows")
    s = torch.stack([vid2.real.double().sum(), vid2.imag.double().sum(),
                     (vid2.real.double() * torch.arange(vid2.shape[0]).double()[:, None]).sum()])
    assert torch.allclose(s, g["vid64_sum"], rtol=0, atol=1e-6)
    w = synth.make_tensor(5, "txt_norm.weight", (3584,))
    assert_same(O.rmsnorm(g["rms3584_in"], w), g["rms3584_out"], "rmsnorm3584")


def _block_inputs(S_img, T, seed):
    g = torch.Generator().manual_seed(seed)
    image = torch.randn((1, S_img, 3072), generator=g).to(BF)
    text = torch.randn((1, T, 3072), generator=g).to(BF)
    temb = (torch.randn((1, 3072), generator=g) * 0.5).to(BF)
    return image, text, temb


def test_G4_block(golden):
    g = golden("G4_block")
    sd = synth.make_state_dict(synth.dit_block_layout(0), 1234)
    image, text, temb = _block_inputs(128, 40, 44)
    rope = O.rope_tables([(1, 8, 8), (1, 8, 8)], 40)
    text_o, image_o = O.block_forward(sd, 0, image, text, temb, rope)
    assert_same(text_o, g["text_out"], "block text")
    assert_same(image_o, g["image_out"], "block image")
    # the same graph in fp32 reproduces the reference's fp32 run (used as the parity yardstick)
    sd32 = {k: v.float() for k, v in sd.items()}
    t32, i32 = O.block_forward(sd32, 0, image.float(), text.float(), temb.float(), rope)
    assert (t32 - g["text_out_f32"]).abs().max() < 1e-4
    assert (i32 - g["image_out_f32"]).abs().max() < 1e-4


def _model_fn_inputs(h, w, T, n_special, seed):
    noise = synth.make_noise(seed, h, w)
    g = torch.Generator().manual_seed(seed + 100)
    edit = torch.randn((1, 16, h // 8, w // 8), generator=g).to(BF)
    pe = synth.make_prompt_emb(seed + 7, T)
    mask = synth.make_special_token_mask(T, n_special)
    return noise, edit, pe, mask


def test_G5_model_fn_inplace_quirk(golden):
    g = golden("G5_model_fn")
    sd = synth.make_state_dict(synth.dit_layout(2), 1234)
    ad = synth.make_state_dict(synth.adapter_layout(), 4321)
    t_min, t_max = O.adapter_t_range()
    noise, edit, pe, mask = _model_fn_inputs(256, 256, 48, 16, 0)
    pe_run = pe.clone()
    for call, tval in enumerate((986.96, 749.27)):
        t = torch.tensor([tval]).to(BF)
        lat = O.model_fn(sd, ad, noise, t, pe_run, mask, 256, 256, edit, t_min, t_max)
        assert_same(pe_run, g[f"prompt_emb_after_call{call}"], f"prompt_emb after call {call}")
        assert_same(lat, g[f"latents_call{call}"], f"latents call {call}")
    # non-special rows never change; special rows change on every call
    m = mask[0]
    assert torch.equal(pe_run[0, ~m], pe[0, ~m])
    assert not torch.equal(g["prompt_emb_after_call0"][0, m], g["prompt_emb_after_call1"][0, m])
    lat = O.model_fn(sd, None, noise, torch.tensor([500.0]).to(BF), pe.clone(), None, 256, 256, None)
    assert_same(lat, g["latents_plain"], "plain model_fn")


def _controlnet_case():
    """inputs of fixture G13 (tests/golden/make_golden.py::G13_controlnet): shared with the GPU test"""
    sd = synth.make_state_dict(synth.dit_layout(2), 1234)
    nets = [synth.make_state_dict(synth.controlnet_layout(2, add), seed) for seed, add in ((555, 0), (556, 4))]
    noise, edit, pe, _ = _model_fn_inputs(128, 128, 24, 0, 3)
    return sd, nets, noise, edit, pe


def test_G13_controlnet(golden):
    """Block-wise ControlNet hook of model_fn (qwen_image_physical.py:1373-1396), two ControlNets with scales and a progress
    gate, against the reference's own model_fn + QwenImageBlockwiseMultiControlNet."""
    g = golden("G13_controlnet")
    sd, nets, noise, edit, pe = _controlnet_case()
    ctl = [{"sd": nets[0], "conditioning": g["conditioning0"], "scale": 0.7},
           {"sd": nets[1], "conditioning": g["conditioning1"], "scale": 0.5, "start": 1.0, "end": 0.5}]
    assert_same(O.controlnet_preprocess(nets[0], g["conditioning0"]), g["processed0"], "controlnet img_in(patchify(cond))")
    for pid, tval in ((0, 986.96), (3, 300.0)):
        lat = O.model_fn(sd, None, noise, torch.tensor([tval]).to(BF), pe.clone(), None, 128, 128, edit,
                         controlnets=ctl, progress_id=pid, num_inference_steps=4)
        assert_same(lat, g[f"latents_progress{pid}"], f"model_fn + 2 controlnets, progress_id {pid}")
    assert O.controlnet_active(0, 4, 1.0, 0.5) and not O.controlnet_active(3, 4, 1.0, 0.5)
    lat = O.model_fn(sd, None, noise, torch.tensor([500.0]).to(BF), pe.clone(), None, 128, 128, None,
                     controlnets=ctl[:1], progress_id=1, num_inference_steps=4)
    assert_same(lat, g["latents_single"], "model_fn + 1 controlnet")


def _eligen_inputs(h, w, seed):
    """inputs of fixture G14 (tests/golden/make_golden.py::_eligen_inputs): shared with the GPU test"""
    ents = [synth.make_prompt_emb(seed + 20 + i, T) for i, T in enumerate((12, 20, 8))]
    m = torch.zeros((1, 3, 1, h // 8, w // 8), dtype=BF)
    m[0, 0, 0, 1:7, 2:9] = 1
    m[0, 1, 0, 5:14, 6:15] = 1
    m[0, 1, 0, 3, 3] = 1
    return ents, m


def test_G14_eligen(golden):
    """EliGen entity control (QwenImageDiT.process_entity_masks, qwen_image_dit.py:433-498) through model_fn against the
    reference: region mask, per-prompt RoPE restart, interplay with the adapter's in-place special-token update."""
    g = golden("G14_eligen")
    sd = synth.make_state_dict(synth.dit_layout(2), 1234)
    ad = synth.make_state_dict(synth.adapter_layout(), 4321)
    t_min, t_max = O.adapter_t_range()
    noise, edit, pe, mask = _model_fn_inputs(128, 128, 40, 16, 5)
    ents, emask = _eligen_inputs(128, 128, 5)
    assert torch.equal(emask, g["entity_masks"])
    text, (img_f, txt_f), am = O.process_entity_masks(sd, noise, pe, ents, emask, 128, 128, torch.zeros((1, 128, 64), dtype=BF),
                                                      [(1, 8, 8), (1, 8, 8)])
    assert torch.equal((am[0, 0] == 0).to(torch.uint8), g["attention_allowed"])
    assert torch.equal(txt_f.real.contiguous(), g["txt_rotary_real"]) and torch.equal(txt_f.imag.contiguous(), g["txt_rotary_imag"])
    pe_run = pe.clone()
    for call, tval in enumerate((986.96, 600.0)):
        lat = O.model_fn(sd, ad, noise, torch.tensor([tval]).to(BF), pe_run, mask, 128, 128, edit, t_min, t_max,
                         entity_prompt_emb=ents, entity_masks=emask)
        assert_same(pe_run, g[f"prompt_emb_after_call{call}"], f"eligen prompt_emb after call {call}")
        assert_same(lat, g[f"latents_call{call}"], f"eligen latents call {call}")
    lat = O.model_fn(sd, None, noise, torch.tensor([500.0]).to(BF), pe.clone(), None, 128, 128, None,
                     entity_prompt_emb=ents[:2], entity_masks=emask[:, :2])
    assert_same(lat, g["latents_plain"], "eligen, no adapter, no edit image")


def test_G13_controlnet_unit_helpers(golden):
    """QwenImageUnit_BlockwiseControlNet's inpaint helpers (:1211-1222): the oracle's restatement AND the product's host code
    (diffsynth/pipelines/qwen_image_physical.py: pure torch / numpy / PIL, no kernels involved) against the reference unit."""
    import numpy as np
    from PIL import Image
    from diffsynth.pipelines.qwen_image_physical import QwenImagePhysicPipeline
    g = golden("G13_controlnet")
    mask, image = g["unit_mask"].numpy(), g["unit_image"].numpy()
    assert_same(O.controlnet_mask_on_latents(g["unit_latents_in"], mask), g["unit_latents_out"], "oracle mask on latents")
    resized = np.array(Image.fromarray(mask).resize((image.shape[1], image.shape[0])))
    assert np.array_equal(O.controlnet_mask_on_image(image, resized), g["unit_image_out"].numpy())
    pipe = QwenImagePhysicPipeline.__new__(QwenImagePhysicPipeline)
    pipe.device, pipe.torch_dtype = torch.device("cpu"), BF
    assert_same(pipe.apply_controlnet_mask_on_latents(g["unit_latents_in"], Image.fromarray(mask)), g["unit_latents_out"],
                "facade mask on latents")
    out = pipe.apply_controlnet_mask_on_image(Image.fromarray(image), Image.fromarray(mask))
    assert np.array_equal(np.array(out), g["unit_image_out"].numpy())


@pytest.mark.parametrize("cfg", [1.0, 4.0])
def test_G6_loop(golden, cfg):
    g = golden("G6_loop")
    sd = synth.make_state_dict(synth.dit_layout(2), 1234)
    ad = synth.make_state_dict(synth.adapter_layout(), 4321)
    noise, edit, pe_p, mask_p = _model_fn_inputs(128, 128, 40, 16, 0)
    pe_n = synth.make_prompt_emb(8, 24)
    mask_n = synth.make_special_token_mask(24, 16)
    lat = O.denoise_loop(sd, ad, noise, pe_p, pe_n, mask_p, mask_n, 128, 128, 4, cfg_scale=cfg, edit_latents=edit)
    assert_same(lat, g[f"latents_cfg{cfg}_step3"], f"loop cfg {cfg}")


def test_G15_inpaint(golden):
    """image-to-image + inpainting: the mask unit (:714-729), add_noise (:708) and BasePipeline.step's blend (utils/__init__.py:
    146-156) -- the oracle's restatement and the facade's host-side mask preparation, bit for bit against the reference."""
    from PIL import Image
    from diffsynth.pipelines.qwen_image_physical import QwenImagePhysicPipeline
    g, meta = golden("G15_inpaint", with_meta=True)
    h = w = meta["h"]
    mask = O.inpaint_mask_plane(g["mask_rgb_u8"].numpy())
    assert_same(mask, g["mask"], "oracle inpaint mask")
    assert 0.05 < ((mask > 0) & (mask < 1)).float().mean().item()          # the fixture has fractional mask values
    pipe = QwenImagePhysicPipeline.__new__(QwenImagePhysicPipeline)
    pipe.device, pipe.torch_dtype = torch.device("cpu"), BF
    yy, xx = np.mgrid[0:h, 0:w]
    m_u8 = (np.clip(1.4 - np.hypot(yy - 70, xx - 50) / 30.0, 0, 1) * 255).astype("uint8")      # the generator's mask image
    assert_same(pipe.preprocess_inpaint_mask(Image.fromarray(m_u8, mode="L"), h, w), g["mask"], "facade inpaint mask")
    blurred = pipe.preprocess_inpaint_mask(Image.fromarray(m_u8, mode="L"), h, w, 2, 1.5)
    assert blurred.shape == mask.shape and abs(blurred.float().mean().item() - mask.float().mean().item()) < 0.02
    # The blur is torchvision's GaussianBlur in the reference (:726-727), and torchvision is not in this image: no fixture.  An
    # INDEPENDENT implementation of the same published algorithm (normalised exp(-x^2 / 2 sigma^2) taps over 2 * size + 1 pixels,
    # mirror padding without edge repeat) in float64 cross-checks the restatement -- parity with torchvision itself stays unpinned.
    from scipy import ndimage
    pipe.torch_dtype = torch.float32
    for size, sigma in ((2, 1.5), (4, 2.0), (1, 0.8)):
        plain = pipe.preprocess_inpaint_mask(Image.fromarray(m_u8, mode="L"), h, w).double().numpy()[0, 0]
        got = pipe.preprocess_inpaint_mask(Image.fromarray(m_u8, mode="L"), h, w, size, sigma).double().numpy()[0, 0]
        want = ndimage.gaussian_filter(plain, sigma=sigma, mode="mirror", truncate=(size + 0.25) / sigma)   # radius = int(truncate * sigma + 0.5) = size
        assert np.abs(got - want).max() < 2e-6, (size, sigma, np.abs(got - want).max())
    pipe.torch_dtype = BF
    sd = synth.make_state_dict(synth.dit_layout(2), 1234)
    ad = synth.make_state_dict(synth.adapter_layout(), 4321)
    noise, edit, pe_p, mask_p = _model_fn_inputs(h, w, 40, 16, 0)
    pe_n = synth.make_prompt_emb(8, 24)
    mask_n = synth.make_special_token_mask(24, 16)
    x0 = (torch.randn((1, 16, h // 8, w // 8), generator=torch.Generator().manual_seed(meta["x0_seed"])) * 0.7).to(BF)
    tab = O.FlowMatchTables(meta["steps"], dynamic_shift_len=(h // 16) * (w // 16), denoising_strength=meta["denoising_strength"])
    start = tab.add_noise(x0, noise, 0)
    assert_same(start, g["latents_start"], "add_noise")
    lat = O.denoise_loop(sd, ad, start, pe_p, pe_n, mask_p, mask_n, h, w, meta["steps"], cfg_scale=meta["cfg"], edit_latents=edit,
                         denoising_strength=meta["denoising_strength"], input_latents=x0, inpaint_mask=mask)
    assert_same(lat, g[f"latents_step{meta['steps'] - 1}"], "inpaint loop")


def test_G16_rope_sampling(golden):
    """`edit_rope_interpolation=True` (QwenEmbedRope.forward_sampling): oracle tables and the product's host-side tables (rope.py)
    bit-exact against a fresh reference module; oracle model_fn bit-exact with the flag, and the flag changes the result."""
    from physicedit_amd import rope as R
    g, meta = golden("G16_rope_sampling", with_meta=True)
    shapes = [(1, 8, 8), (1, 6, 10), (1, 8, 8), (1, 11, 5)]
    vid, txt = O.rope_tables(shapes, 37, sampling=True)
    assert torch.equal(vid.real, g["vid_re"]) and torch.equal(vid.imag, g["vid_im"])
    assert torch.equal(txt.real, g["txt_re"]) and torch.equal(txt.imag, g["txt_im"])
    ci, si, ct, st = R.rope_cos_sin(shapes, 37, sampling=True)
    assert torch.equal(ci, g["vid_re"]) and torch.equal(si, g["vid_im"]) and torch.equal(ct, g["txt_re"]) and torch.equal(st, g["txt_im"])
    plain = R.rope_cos_sin(shapes, 37)[0]
    assert torch.equal(plain[:64], ci[:64]) and not torch.equal(plain[64:124], ci[64:124]) and torch.equal(plain[124:188], ci[124:188])
    sd = synth.make_state_dict(synth.dit_layout(2), 1234)
    ad = synth.make_state_dict(synth.adapter_layout(), 4321)
    noise, _, pe, mask = _model_fn_inputs(meta["h"], meta["w"], meta["T"], meta["n_special"], 0)
    edit = torch.randn((1, 16, meta["edit_h"] // 8, meta["edit_w"] // 8), generator=torch.Generator().manual_seed(meta["edit_seed"])).to(BF)
    t = torch.tensor([meta["timestep"]]).to(BF)
    t_min, t_max = O.adapter_t_range()
    lat = O.model_fn(sd, ad, noise, t, pe.clone(), mask, meta["h"], meta["w"], edit, t_min, t_max, edit_rope_interpolation=True)
    assert_same(lat, g["latents"], "model_fn with sampled rope")
    lat0 = O.model_fn(sd, ad, noise, t, pe.clone(), mask, meta["h"], meta["w"], edit, t_min, t_max)
    assert_same(lat0, g["latents_plain"], "model_fn, same inputs, regular rope")
    assert not torch.equal(g["latents"], g["latents_plain"])


def test_G17_special_token_loss(golden):
    """model_fn's is_train=True branch (:1337-1338 -> VisualThinkingDualAdapter.get_loss): the oracle's loss value, bit for bit."""
    g, meta = golden("G17_special_token_loss", with_meta=True)
    sd = synth.make_state_dict(synth.dit_layout(2), 1234)
    ad = synth.make_state_dict(synth.adapter_layout(), 4321)
    noise, edit, pe, mask = _model_fn_inputs(meta["h"], meta["w"], meta["T"], meta["n_special"], 0)
    gen = torch.Generator().manual_seed(meta["gt_seed"])
    gt_d = (torch.randn((1, meta["n_special"], 3584), generator=gen) * 0.5).to(BF)
    gt_v = (torch.randn((1, meta["n_special"], 3584), generator=gen) * 0.5).to(BF)
    t_min, t_max = O.adapter_t_range()
    for i, tval in enumerate(meta["timesteps"]):
        cap = {}
        lat = O.model_fn(sd, ad, noise, torch.tensor([tval]).to(BF), pe.clone(), mask, meta["h"], meta["w"], edit, t_min, t_max,
                         pseudo_special_emb=(gt_d, gt_v), capture=cap)
        assert_same(lat, g[f"latents_{i}"], f"latents t={tval}")
        assert torch.equal(cap["special_token_loss"].reshape(1), g[f"loss_{i}"]), (tval, cap["special_token_loss"], g[f"loss_{i}"])
    assert len({float(g[f"loss_{i}"]) for i in range(3)}) == 3         # the weighting moves with the timestep


def _prior_inputs(meta):
    g = torch.Generator().manual_seed(meta["inputs_seed"])
    B = meta["frames"]
    return (torch.randn((B, 256, 768), generator=g).to(BF), torch.randn((1, 256, 768), generator=g).to(BF),
            torch.randn((B, 16, meta["lat_h"], meta["lat_w"]), generator=g).to(BF),
            torch.randn((1, 16, meta["lat_h"], meta["lat_w"]), generator=g).to(BF))


def test_G18_visual_prior(golden):
    """The training-time prior after DINOv2 / the VAE (PhysicalVisualEmbedder.process :1071-1118: frame embeddings, two Perceiver
    resamplers, two adapters, middle - source) against the reference's own modules, bit for bit."""
    g, meta = golden("G18_visual_prior", with_meta=True)
    sd = synth.make_state_dict(synth.prior_layout(), meta["weights_seed"])
    dino_mid, dino_src, lat_mid, lat_src = _prior_inputs(meta)
    B = meta["frames"]
    dm = (dino_mid + sd["dino_time_embed.weight"][torch.arange(B)].unsqueeze(1)).reshape(1, -1, 768)
    assert_same(O.perceiver_resampler(sd, "dino_resampler.", dm), g["dino_resampled_middle"], "dino resampler")
    pd, pv = O.visual_prior(sd, dino_mid, dino_src, lat_mid, lat_src)
    assert_same(pd, g["pseudo_dino"], "pseudo_special_emb_dino")
    assert_same(pv, g["pseudo_vae"], "pseudo_special_emb_vae")


def test_G19_dinov2(golden):
    """DINOv2 with registers as the reference's Dinov2withNorm runs it (pipelines/dinov2.py:8-31, a random 2-layer instance in bf16):
    patch features at the configured size and at another one (resampled position table), bit for bit."""
    g, meta = golden("G19_dinov2", with_meta=True)
    sd = synth.make_state_dict(synth.dino_layout(meta["hidden"], meta["layers"], 4, meta["patch"], meta["image_size"]), meta["weights_seed"])
    gen = torch.Generator().manual_seed(meta["inputs_seed"])
    x224 = torch.randn((2, 3, 224, 224), generator=gen)
    x168 = torch.randn((1, 3, 168, 112), generator=gen)
    assert_same(O.dinov2_features(sd, x224.to(BF), meta["heads"], meta["patch"]), g["feat_224"], "dinov2 224x224")
    assert_same(O.dinov2_features(sd, x168.to(BF), meta["heads"], meta["patch"]), g["feat_168x112"], "dinov2 168x112 (interpolated positions)")
    sd32 = {k: v.float() for k, v in sd.items()}
    d = (O.dinov2_features(sd32, x224.to(BF).float(), meta["heads"], meta["patch"]) - g["feat_224_fp32"]).abs().max().item()
    assert d <= 1e-5, d


def test_G7_vae(golden):
    g = golden("G7_vae")
    vs = synth.make_state_dict(synth.vae_layout(), 77)
    # the restatement's premise: causal conv3d at T=1 == conv2d with the last temporal tap
    d = (g["conv3d_ref"].float() - g["conv2d_lasttap"].float()).abs()
    assert d.max() <= 2 ** -7 * g["conv3d_ref"].float().abs().max()
    for R in (64, 96):
        x = O.preprocess_image(synth.make_edit_image_u8(R, R, seed=R))
        gen = torch.Generator().manual_seed(R)
        lat = torch.randn((1, 16, R // 8, R // 8), generator=gen).to(BF)
        O.VAE_CONV_MODE = "3d"      # literal causal conv3d: bit-exact with the reference
        z, y = O.vae_encode(vs, x), O.vae_decode(vs, lat)
        assert_same(z, g[f"enc_{R}"], f"vae enc {R}")
        assert_same(y, g[f"dec_{R}"], f"vae dec {R}")
        try:
            O.VAE_CONV_MODE = "2d"  # last-tap 2-D conv: same math, other accumulation order
            z2, y2 = O.vae_encode(vs, x), O.vae_decode(vs, lat)
        finally:
            O.VAE_CONV_MODE = "3d"
        for got, ref, what in ((z2, z, "enc"), (y2, y, "dec")):
            diff = (got.float() - ref.float()).abs()
            tol = 2 ** -6 * ref.float().abs().clamp_min(1.0)   # <= 2 bf16 ulp of max(|ref|,1)
            assert (diff <= tol).all(), (what, R, diff.max().item())
    gen = torch.Generator().manual_seed(9)
    _ = torch.randn((1, 96, 1, 24, 24), generator=gen)
    xn = (torch.randn((1, 96, 1, 8, 8), generator=gen).to(BF) * 3)[:, :, 0]
    assert_same(xn.contiguous(), g["rmsnorm_in"], "rmsnorm in")
    assert_same(O._rms_norm_c(xn, vs["decoder.norm_out.gamma"]), g["rmsnorm_out"], "vae rmsnorm")


def test_G8_lora(golden):
    g, meta = golden("G8_lora", with_meta=True)
    sd = synth.make_state_dict(synth.dit_layout(1), 1234)
    lora = synth.make_lora(4321, 1, meta["rank"])
    before = sd["transformer_blocks.0.img_mlp.net.0.proj.weight"].clone()
    n = O.lora_merge(sd, lora, alpha=1.0)
    assert n == 12
    for t in synth.LORA_TARGETS:
        k = f"transformer_blocks.0.{t}.weight"
        assert_same(sd[k][:64, :256].contiguous(), g[k + ".head"], k)
        assert sd[k].double().sum().item() == g[k + ".sum"].item()
    assert torch.equal(sd["transformer_blocks.0.img_mlp.net.0.proj.weight"], before)


def test_G9_adapter(golden):
    g = golden("G9_adapter")
    ad = synth.make_state_dict(synth.adapter_layout(), 4321)
    t_min, t_max = O.adapter_t_range()
    assert g["t_range"][0].item() == t_min and g["t_range"][1].item() == t_max
    gen = torch.Generator().manual_seed(9)
    x = torch.randn((1, 64, 3584), generator=gen).to(BF)
    for tv in (1000.0, 748.0, 300.0, 20.0):
        t = torch.tensor([tv]).to(BF)
        mixed, d, v = O.adapter_forward(ad, x, t, t_min, t_max)
        assert_same(mixed, g[f"mixed_{int(tv)}"], f"mixed {tv}")
        assert O.adapter_alpha(t, t_min, t_max).float().item() == g[f"alpha_{int(tv)}"].item()
    assert_same(d, g["dino"], "dino head")
    assert_same(v, g["vae"], "vae head")


def test_G10_image(golden):
    g = golden("G10_image")
    ramp = (np.arange(16 * 16 * 3) % 256).astype("uint8").reshape(16, 16, 3)
    assert_same(O.preprocess_image(ramp), g["pre"], "preprocess")
    assert_same(O.vae_output_to_u8(g["post_in"]), g["post_u8"], "postprocess")


def test_G11_hot_lora(golden):
    """Runtime LoRA (hotload=True), vram_management/layers.py:166-181: one Linear and a whole block."""
    g, meta = golden("G11_hot_lora", with_meta=True)
    sd = O.attach_hot_lora(synth.make_state_dict(synth.dit_block_layout(0), 1234), synth.make_lora(4321, 1, meta["rank"]))
    assert len(sd.hot) == 12
    gen = torch.Generator().manual_seed(55)
    x = torch.randn((1, 70, 3072), generator=gen).to(BF)
    assert_same(O._linear(sd, "transformer_blocks.0.attn.to_q", x), g["linear_out"], "hot-lora linear")
    image, text, temb = _block_inputs(128, 40, 44)
    rope = O.rope_tables([(1, 8, 8), (1, 8, 8)], 40)
    text_o, image_o = O.block_forward(sd, 0, image, text, temb, rope)
    assert_same(text_o, g["text_out"], "hot-lora block text")
    assert_same(image_o, g["image_out"], "hot-lora block image")
    # and it is NOT the same arithmetic as merging the LoRA into the weights
    merged = synth.make_state_dict(synth.dit_block_layout(0), 1234)
    O.lora_merge(merged, synth.make_lora(4321, 1, meta["rank"]))
    _, image_m = O.block_forward(merged, 0, image, text, temb, rope)
    assert not torch.equal(image_m, image_o)


def test_fp8_attention_restatements_agree():
    """`flash_attention_fp8` restates what the e4m3 attention kernels do inside (FlashAttention-3 cannot run here: parity of P's quantisation is
    unpinned, see the oracle's header): the online forms the kernels are pinned to -- running maximum, the lazily raised reference, the
    sum-limited fast path, and the row sums of the quantised P (attn_fp8_variant 0 / 1 + 3 / 2 / 4) -- must all sit at the same distance from
    the fp32 attention as the form that knows the final maximum, and differ from it by a fraction of that distance: they move the rounding
    points of P, nothing else."""
    g = torch.Generator().manual_seed(77)
    q, k, v = ((torch.randn((1, 2, 333, 128), generator=g) * s).to(torch.bfloat16) for s in (1.3, 0.9, 2.0))
    truth = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float())
    base = O.flash_attention_fp8(q, k, v)
    rms = lambda a, b: float((a.float() - b.float()).pow(2).mean().sqrt())
    e0 = rms(base, truth)
    assert 0 < e0 < 0.05
    forms = {"running max": dict(kv_tile=64), "lazy 2^8": dict(kv_tile=64, lazy_tau_log2=8.0), "sum limit": dict(kv_tile=64, lazy_sum_limit=448.0),
             "lazy 2^8, quantised row sums": dict(kv_tile=64, lazy_tau_log2=8.0, row_sum_quantised=True)}
    for name, kw in forms.items():
        out = O.flash_attention_fp8(q, k, v, **kw)
        assert torch.isfinite(out.float()).all(), name
        assert rms(out, truth) <= 1.1 * e0, (name, rms(out, truth), e0)
        assert rms(out, base) <= 0.9 * e0, (name, rms(out, base), e0)
    # without quantising P every online form is the plain softmax
    assert rms(O.flash_attention_fp8(q, k, v, p_dtype=None), truth) <= e0
