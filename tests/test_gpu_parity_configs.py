"""`-m gpu` end-to-end parity at the depth and geometry of the BASELINE configs:

  * configs[1]/[3]: the FULL 60-layer DiT (+ adapter) against fixtures written by the REFERENCE ITSELF in the build container
    (tests/golden/make_golden.py G21: `model_fn_qwen_image`, `qwen_image_physical.py:1302-1403`, on the same counter-based weights
    `synth.make_state_dict_hashed` regenerates on the GPU bit for bit): G21 = the headline geometry (1024x1024 + 1024x1024 edit, S = 8704),
    G22 = 512x512 + 512x512 (S = 2208) with an fp32 evaluation of the same graph for the fp32-distance criterion, G23 = two CFG-4 steps
    of the loop (`:644-661`), G24 = configs[4]'s per-GPU geometry (S = 11497).  No host oracle pass runs for them: nothing to skip
    when the host is slow.  VAE decode at 1024 x 1024 against the oracle;
  * configs[4]: one layer at the 1328x1328 geometry (83x83 noise tokens + 64x64 edit tokens, T = 512) against the oracle.

Each test records frac(|d| <= 1e-3), max |d| and the fp32-distance ratio (tests/parity_record.py)."""
import pytest
import torch

import oracle.physicedit_oracle as O
from parity_record import parity_stats, record
from physicedit_amd import synth

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


class HostView:
    """Read-only state-dict view for the oracle over weights that live on the GPU: a tensor is copied to the host when the oracle
    first asks for it and kept (bf16; widened to fp32 per access when asked), so the seven oracle passes of this module over the
    60-layer model copy its 41 GB once instead of seven times.  `cache`: a dict shared by the views of one state dict."""

    def __init__(self, dev_sd, dtype=None, cache=None):
        self.sd, self.dtype, self.cache = dev_sd, dtype, cache

    def __contains__(self, k):
        return k in self.sd

    def __getitem__(self, k):
        if self.cache is not None:
            t = self.cache.get(k)
            if t is None:
                t = self.cache[k] = self.sd[k].cpu()
        else:
            t = self.sd[k].cpu()
        return t.to(self.dtype) if self.dtype is not None else t

    def get(self, k, default=None):
        return self[k] if k in self.sd else default

    def keys(self):
        return self.sd.keys()


def _inputs(h, w, eh, ew, T, nsp, seed):
    noise = synth.make_noise(seed, h, w)
    g = torch.Generator().manual_seed(seed + 100)
    edit = torch.randn((1, 16, eh // 8, ew // 8), generator=g).to(BF)
    pe = synth.make_prompt_emb(seed + 7, T)
    mask = synth.make_special_token_mask(T, nsp)
    return noise, edit, pe, mask


_MODEL60 = {}


@pytest.fixture(scope="module", autouse=True)
def _release_model60():
    yield
    _MODEL60.clear()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


def model60():
    """the 60-layer DiT (+ adapter) of this module's tests: generated on the device and wrapped in an engine ONCE (41 GB), from the
    counter-based generator -- the same bits the build container drew on its host for the reference's fixtures (checked against it in
    tests/test_cabi_host.py::test_hashed_weights_checksum and test_hashed_weights_same_on_device below); `host`: a host cache for the
    opt-in oracle passes"""
    if not _MODEL60:
        from physicedit_amd.dit import QwenImageDiTEngine
        dev = torch.device("cuda")
        sd_dev = synth.make_state_dict_hashed(synth.dit_layout(60), 1234, dev)
        ad = synth.make_state_dict(synth.adapter_layout(), 4321)
        _MODEL60.update(sd=sd_dev, ad=ad, eng=QwenImageDiTEngine(sd_dev, ad, device=dev), host={})
    return _MODEL60["sd"], _MODEL60["ad"], _MODEL60["eng"], _MODEL60["host"]


def test_hashed_weights_same_on_device():
    """the premise of the G21-G24 fixtures: `synth.make_state_dict_hashed` draws the same bits on the GPU as on a host"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    lay = [kv for kv in synth.dit_layout(1) if "img_mlp.net.0" in kv[0] or "norm_q" in kv[0] or "to_q" in kv[0] or kv[0].startswith("proj_out")]
    a = synth.make_state_dict_hashed(lay, 1234, "cpu")
    b = synth.make_state_dict_hashed(lay, 1234, torch.device("cuda"))
    assert len(a) >= 7
    for k in a:
        assert torch.equal(a[k], b[k].cpu()), k


def _forward60(HW, EH, T, nsp, variants, fp8=False):
    """one `model_fn` call of the 60-layer engine at the first timestep of the 40-step schedule per attention variant"""
    from physicedit_amd._lib import lib
    from physicedit_amd.dit import special_indices
    from physicedit_amd.scheduler import qwen_image_scheduler
    dev = torch.device("cuda")
    sd_dev, ad, eng, host = model60()
    noise, edit, pe, mask = _inputs(HW, HW, EH, EH, T, nsp, 0)
    sch = qwen_image_scheduler()
    sch.set_timesteps(40, dynamic_shift_len=(HW // 16) * (HW // 16))
    t = sch.timesteps[0:1].to(BF)
    got, special = {}, {}
    try:
        for variant in variants:
            assert lib().pe_debug_set(b"attn_variant", variant) == 0
            pe_d = pe.cuda().clone()
            got[variant] = eng.forward(noise.cuda(), t, pe_d, special_indices(mask, dev), edit.cuda()).clone()
            special[variant] = pe_d[mask.cuda()].cpu()
        if fp8:      # the e4m3 attention branch at depth: reported next to the bf16 kernels' distance to fp32
            got["fp8"] = eng.forward(noise.cuda(), t, pe.cuda().clone(), special_indices(mask, dev), edit.cuda(),
                                     enable_fp8_attention=True).clone()
        torch.cuda.synchronize()
    finally:
        lib().pe_debug_set(b"attn_variant", 5)
    return got, special, (noise, edit, pe, mask, t)


_NAMES = {5: "attention variant 5 = default", 4: "attention variant 4", 0: "attention variant 0"}


def _check_vs_fp32(st):
    # as close to the fp32 evaluation of the same graph as the reference's own bf16 run is
    for s_ in st.values():
        assert s_["fp32_distance_ratio"] <= 1.25, s_
        assert s_["max_to_fp32_hip"] <= 1.5 * s_["max_to_fp32_reference_bf16"] + 1e-3, s_
    # the decision rule for the default attention kernel: not measurably further from fp32 than the textbook update
    if 0 in st:
        for v in st:
            assert st[v]["fp32_distance_ratio"] <= st[0]["fp32_distance_ratio"] * 1.02 + 1e-3, (st[v], st[0])


def _depth_meets_length(HW, T, nsp, case, edit_hw=None, config="configs[1]", fp32=True):
    """OPT-IN form (PE_PARITY_FULL=1): the same comparison against host ORACLE passes in bf16 and fp32 (minutes of CPU), for geometries
    without an fp32 fixture.  Returns the parity records by variant."""
    import os
    sd_dev, ad, eng, host = model60()
    EH = HW if edit_hw is None else edit_hw
    variants = (5, 4, 0) if fp32 else (5,)
    got, _, (noise, edit, pe, mask, t) = _forward60(HW, EH, T, nsp, variants, fp8=fp32)
    t_min, t_max = O.adapter_t_range()
    threads = torch.get_num_threads()
    ref32 = None
    try:
        torch.set_num_threads(max(threads, min(32, os.cpu_count() or 16)))
        ref = O.model_fn(HostView(sd_dev, cache=host), ad, noise, t, pe.clone(), mask, HW, HW, edit, t_min, t_max)
        if fp32:
            # the fp32 pass is bound by element-wise traffic on the host: more threads than oneDNN's bf16 GEMMs like
            torch.set_num_threads(max(threads, min(64, os.cpu_count() or 16)))
            ad32 = {k: v.float() for k, v in ad.items()}
            ref32 = O.model_fn(HostView(sd_dev, torch.float32, cache=host), ad32, noise.float(), t.float(), pe.clone().float(), mask, HW, HW,
                               edit.float(), t_min, t_max)
    finally:
        torch.set_num_threads(threads)
    st = {v: record(config, f"{case} [{_NAMES[v]}] [host oracle]", got[v], ref, ref32) for v in variants}
    assert torch.isfinite(ref.float()).all() and all(torch.isfinite(g.float()).all() for g in got.values())
    if "fp8" in got:
        st8 = record(config, f"{case} [enable_fp8_attention: e4m3 attention vs the bf16-attention references] [host oracle]", got["fp8"], ref, ref32)
        assert st8["fp32_distance_ratio"] <= 6.0, st8
    if not fp32:
        assert st[5]["mean_abs_diff"] <= 2e-3 and st[5]["max_ulp"] <= 6.0, st[5]
        return st
    _check_vs_fp32(st)
    return st


def test_60_layers_depth_meets_length(golden):
    """Depth AND sequence length together: the FULL 60-layer DiT + adapter on a 512x512 target with a 512x512 edit image (S_img = 2048)
    and T = 160 with 16 special tokens: S = 2208 -> 9 query blocks and 35 KV tiles per head in the flash kernel (multi-tile, split-KV
    leftovers), 9 M tiles per GEMM (several rounds of work-groups: the persistent schedule walks tiles).  One `model_fn` call at the first
    timestep of the 40-step schedule (t = 1000: the adapter's in-place update of the special rows included) against fixture G22: the
    REFERENCE's own bf16 output (`qwen_image_physical.py:1302-1403`; the oracle was bit-identical to it at this depth when the fixture
    was written) and an fp32 evaluation of the same graph.  THREE attention variants: the default (5: lazy max, scale and max folded
    out of the softmax stream) and round 3's (4) must be as close to the fp32 evaluation as the reference's own bf16 run, and no further
    from it than the textbook kernel (0) -- the repo's criterion for choosing a default (profiles/r03_attention_notes.md,
    r04_attention_notes.md).  The e4m3 attention branch is reported against the same pair."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    fx = golden("G22_60_layers_s2208")
    ref, ref32 = fx["latents"], fx["latents_fp32"]
    got, special, _ = _forward60(512, 512, 160, 16, (5, 4, 0), fp8=True)
    case = "60 layers, 512x512 + 512x512 edit (S = 2208), T 160, one model_fn call (first of 40 steps) vs the REFERENCE (G22)"
    st = {v: record("configs[1]", f"{case} [{_NAMES[v]}]", got[v], ref, ref32) for v in (5, 4, 0)}
    assert all(torch.isfinite(g.float()).all() for g in got.values())
    _check_vs_fp32(st)
    st8 = record("configs[1]", f"{case} [enable_fp8_attention: e4m3 attention vs the bf16-attention references]", got["fp8"], ref, ref32)
    assert st8["fp32_distance_ratio"] <= 6.0, st8
    # the adapter's in-place update of the special rows (adapter GEMMs only: shallow), against the reference's
    sp = parity_stats(special[5], fx["special_after"])
    assert sp["max_ulp"] <= 4.0 and sp["frac_bit_identical"] >= 0.9, sp


def test_60_layers_headline_geometry(golden):
    """The same at BASELINE configs[1]'s own geometry: 1024x1024 target + 1024x1024 edit image, T = 512 with 64 special tokens,
    S = 8704 (34 query blocks x 136 KV tiles per head; 1632 / 1224 / 408-tile GEMM launches), against fixture G21: the output of the
    REFERENCE's `model_fn_qwen_image` itself on the same 60-layer weights (176 s of the build container's CPU; no host pass here, so
    the test never skips).  Element-wise it is held to what 60 layers of bf16 allow (mean |d| <= 2e-3, <= 6 ulp: two bf16 runs of the
    graph that sum in different orders differ by that -- the reference's own run is 2e-2 rms from an fp32 evaluation); when the fixture
    carries the fp32 evaluation (PE_G21_FP32=1 at generation) the fp32-distance criterion is applied too.  PE_PARITY_FULL=1 adds the
    host-oracle form with all attention variants."""
    import os
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    fx = golden("G21_60_layers_headline")
    ref, ref32 = fx["latents"], fx.get("latents_fp32")
    got, special, _ = _forward60(1024, 1024, 512, 64, (5,))
    case = "60 layers, 1024x1024 + 1024x1024 edit (S = 8704), T 512, one model_fn call (first of 40 steps) vs the REFERENCE (G21)"
    st = record("configs[1]", f"{case} [{_NAMES[5]}]", got[5], ref, ref32)
    assert torch.isfinite(got[5].float()).all()
    assert st["mean_abs_diff"] <= 2e-3 and st["max_ulp"] <= 8.0, st      # measured 5 - 6 ulp: two bf16 runs of 60 layers (the reference on 6 or 8 host threads differs from itself by as much)
    if ref32 is not None:
        _check_vs_fp32({5: st})
    sp = parity_stats(special[5], fx["special_after"])
    assert sp["max_ulp"] <= 4.0 and sp["frac_bit_identical"] >= 0.9, sp
    if os.environ.get("PE_PARITY_FULL") == "1":
        _depth_meets_length(1024, 512, 64, "60 layers, 1024x1024 + 1024x1024 edit (S = 8704), T 512, one model_fn call (first of 40 steps)")


def test_60_layers_two_cfg_steps(golden):
    """Multi-step at depth: TWO CFG-4 steps of the 60-layer loop at S = 672 / 592 (256x256 target + 256x256 edit image, T_pos = 160
    and T_neg = 80, 16 special tokens each) through DenoiseLoop's default form (two streams), against fixture G23: the REFERENCE's own
    loop (`qwen_image_physical.py:644-661`, four `model_fn_qwen_image` forwards + `step`) on the same weights: the adapter's in-place
    accumulation on the special rows across steps (the second step's prompt embeddings are the first step's outputs), the CFG combine
    and the Euler update, at full depth.  Bound: a 2-step schedule moves the latents by 0.5 pred per step and CFG 4 weighs the two
    forwards' errors by 4 and 3, so the single-call mean |d| of 1.65e-3 becomes ~1e-2 at the end; twice that is the limit."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from physicedit_amd.pipeline import DenoiseLoop
    fx = golden("G23_60_layers_two_cfg_steps")
    sd_dev, ad, eng, host = model60()
    noise, edit, pe_p, mask_p = _inputs(256, 256, 256, 256, 160, 16, 3)
    pe_n = synth.make_prompt_emb(11, 80)
    mask_n = synth.make_special_token_mask(80, 16)
    loop = DenoiseLoop(eng, dual_stream=True)
    pp, pn = pe_p.cuda().clone(), pe_n.cuda().clone()
    got = loop(noise.cuda(), pp, pn, mask_p, mask_n, 256, 256, num_inference_steps=2, cfg_scale=4.0, edit_latents=edit.cuda())
    torch.cuda.synchronize()
    ref = fx["latents_step1"]
    st = record("configs[1]", "60 layers, 256x256 + 256x256 edit, TWO CFG-4 steps of the loop (two streams) vs the REFERENCE's loop (G23)",
                got, ref, fx["latents_step1_fp32"])
    assert torch.isfinite(got.float()).all()
    # round 6: the fixture carries an fp32 evaluation of the same two steps, so "as good as the reference's own bf16 run" is shown for the
    # loop too (rms distance to fp32 within 1.25 x the reference's), and the element-wise bound is 1.5 x / 1.7 x what was measured
    # (mean 9.3e-3, max 0.070) instead of 2 x / 3.5 x
    assert st["fp32_distance_ratio"] <= 1.25, st
    assert st["mean_abs_diff"] <= 1.4e-2 and st["max_abs_diff"] <= 0.12, st
    # the loop owns its prompt embeddings like the reference's `inputs_posi` / `inputs_nega` entries: the adapter rewrote the special
    # rows (twice), nothing else -- and to the reference's values
    mp, mn = mask_p[0].bool(), mask_n[0].bool()
    assert torch.equal(pp[0].cpu()[~mp], pe_p[0][~mp]) and torch.equal(pn[0].cpu()[~mn], pe_n[0][~mn])
    assert not torch.equal(pp[0].cpu()[mp], pe_p[0][mp]) and not torch.equal(pn[0].cpu()[mn], pe_n[0][mn])
    for name, rows, want in (("posi", pp[0].cpu()[mp], fx["special_posi_after"]), ("nega", pn[0].cpu()[mn], fx["special_nega_after"])):
        sp = parity_stats(rows, want)
        assert sp["max_ulp"] <= 6.0 and sp["frac_bit_identical"] >= 0.7, (name, sp)


def test_60_layers_40_step_loop(golden):
    """The FULL loop at depth: 40 CFG-4 steps x 60 layers through DenoiseLoop's default form (two streams) at 256x256 + a 256x256 edit
    image (S = 672 / 592; G23's inputs), against fixture G25: the REFERENCE's own loop (`qwen_image_physical.py:644-661`, 80
    `model_fn_qwen_image` forwards) -- forty in-place applications of the adapter to each branch's special rows (`:1333-1336`), the
    dynamic-shift schedule with its terminal step (`flow_match.py:72-82`), 80 forwards of error accumulation -- and an fp32 evaluation
    of the same 40-step graph.  After 80 bf16 forwards two correct bf16 runs differ element-wise by far more than 1e-3 (the
    reference's own bf16 run is that far from fp32), so the criterion is the distance to fp32: rms(hip - fp32) <= 1.25 x
    rms(reference_bf16 - fp32) at the end AND along the trajectory.  The first steps, where the runs have not drifted yet, are also
    held element-wise: latents within 16 ulp, both branches' special rows within 4 ulp of the reference's."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from physicedit_amd.pipeline import DenoiseLoop
    fx = golden("G25_40_steps_60_layers")
    sd_dev, ad, eng, host = model60()
    noise, edit, pe_p, mask_p = _inputs(256, 256, 256, 256, 160, 16, 3)
    pe_n = synth.make_prompt_emb(11, 80)
    mask_n = synth.make_special_token_mask(80, 16)
    mp, mn = mask_p[0].bool().cuda(), mask_n[0].bool().cuda()
    loop = DenoiseLoop(eng, dual_stream=True)
    pp, pn = pe_p.cuda().clone(), pe_n.cuda().clone()
    traj, specials = {}, {}

    def keep(i, lat):
        traj[i] = lat.clone()
        if i < 4:
            specials[i] = (pp[0][mp].clone(), pn[0][mn].clone())
    got = loop(noise.cuda(), pp, pn, mask_p, mask_n, 256, 256, num_inference_steps=40, cfg_scale=4.0, edit_latents=edit.cuda(), on_step=keep)
    torch.cuda.synchronize()
    assert torch.isfinite(got.float()).all() and torch.equal(got, traj[39])
    ratios = {}
    for i in (0, 1, 3, 9, 19, 29, 39):
        st = parity_stats(traj[i], fx[f"latents_step{i}"], fx[f"latents_step{i}_fp32"])
        ratios[i] = st["fp32_distance_ratio"]
        print(f"[parity] 40-step loop, after step {i + 1}: rms to fp32 hip {st['rms_to_fp32_hip']:.4e} reference-bf16 "
              f"{st['rms_to_fp32_reference_bf16']:.4e} (ratio {ratios[i]:.3f}); vs reference mean|d| {st['mean_abs_diff']:.3e} max {st['max_ulp']:.1f} ulp")
        if i < 4:
            assert st["max_ulp"] <= 16.0, (i, st)
    st = record("configs[1]", "60 layers, 256x256 + 256x256 edit, the FULL 40-step CFG-4 loop (two streams) vs the REFERENCE's loop (G25)",
                got, fx["latents_step39"], fx["latents_step39_fp32"], trajectory_fp32_distance_ratio={str(k): v for k, v in ratios.items()})
    assert max(ratios.values()) <= 1.25, ratios
    for i in range(4):
        for name, rows, want in (("posi", specials[i][0], fx[f"special_posi_step{i}"]), ("nega", specials[i][1], fx[f"special_nega_step{i}"])):
            sp = parity_stats(rows, want)
            assert sp["max_ulp"] <= 4.0, (i, name, sp)
    # after step 40: every special row has been through the adapter forty times on both sides
    for name, rows, want, want32 in (("posi", pp[0][mp], fx["special_posi_after"], fx["special_posi_after_fp32"]),
                                     ("nega", pn[0][mn], fx["special_nega_after"], fx["special_nega_after_fp32"])):
        sp = parity_stats(rows, want, want32)
        print(f"[parity] 40-step loop, {name} special rows after step 40: max {sp['max_ulp']:.1f} ulp, fp32-distance ratio {sp['fp32_distance_ratio']:.3f}")
        assert sp["fp32_distance_ratio"] <= 1.25, (name, sp)


def test_60_layers_configs4_geometry(golden):
    """And at BASELINE configs[4]'s per-GPU geometry: 1328x1328 target (83 x 83 = 6889 noise tokens, odd) + the 1024x1024 edit image,
    T = 512: S = 11497, 60 layers, against fixture G24 (the REFERENCE's `model_fn_qwen_image` on the same weights).  PE_PARITY_FULL=1
    adds the host-oracle form with the fp32 pass (~15 minutes of CPU; numbers in profiles/r03_parity.json)."""
    import os
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if not os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G24_60_layers_configs4.safetensors")):
        pytest.skip("fixture G24 not generated (tests/golden/make_golden.py G21 with PE_G21_PARTS=24)")
    fx = golden("G24_60_layers_configs4")
    got, special, _ = _forward60(1328, 1024, 512, 64, (5,))
    case = "60 layers, 1328x1328 + 1024x1024 edit (S = 11497), T 512, one model_fn call (first of 40 steps) vs the REFERENCE (G24)"
    st = record("configs[4]", f"{case} [{_NAMES[5]}]", got[5], fx["latents"])
    assert torch.isfinite(got[5].float()).all()
    assert st["mean_abs_diff"] <= 2e-3 and st["max_ulp"] <= 8.0, st      # measured 5 - 6 ulp: two bf16 runs of 60 layers (the reference on 6 or 8 host threads differs from itself by as much)
    sp = parity_stats(special[5], fx["special_after"])
    assert sp["max_ulp"] <= 4.0 and sp["frac_bit_identical"] >= 0.9, sp
    if os.environ.get("PE_PARITY_FULL") == "1":
        _depth_meets_length(1328, 512, 64, "60 layers, 1328x1328 + 1024x1024 edit (S = 11497), T 512, one model_fn call (first of 40 steps)",
                            edit_hw=1024, config="configs[4]")


def test_vae_decode_1024_vs_oracle():
    """VAE decode at the headline size (1024 x 1024: 128 x 128 latents, 16384-token mid-block attention, every upsampling stage at
    its real extent) against the oracle's 2-D form (`qwen_image_vae.py:719-729`; the causal conv3d with one frame is the 2-D
    convolution with its last temporal tap, test_oracle_golden.py pins that equivalence), in bf16 and fp32."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from physicedit_amd.vae import QwenImageVAE
    torch.set_num_threads(max(torch.get_num_threads(), 16))
    vs = synth.make_state_dict(synth.vae_layout(), 77)
    v = QwenImageVAE(vs, device="cuda")
    gen = torch.Generator().manual_seed(1024)
    lat = torch.randn((1, 16, 128, 128), generator=gen).to(BF)
    got = v.decode(lat.cuda())
    got_u8 = v.decode(lat.cuda(), output_u8=True)
    O.VAE_CONV_MODE = "2d"
    try:
        ref = O.vae_decode(vs, lat)
        ref32 = O.vae_decode({k: t.float() for k, t in vs.items()}, lat.float())
    finally:
        O.VAE_CONV_MODE = "3d"
    st = record("configs[1]", "VAE decode 1024x1024 vs oracle (2-D form)", got, ref, ref32)
    assert got.shape == (1, 3, 1024, 1024) and torch.isfinite(got.float()).all()
    assert st["fp32_distance_ratio"] <= 1.3, st
    assert st["max_abs_diff"] <= 10.0 * st["rms_to_fp32_reference_bf16"] + 1e-3, st
    ref_u8 = O.vae_output_to_u8(ref)
    du8 = (got_u8.cpu().float() - ref_u8.float()).abs()
    print(f"[parity] vae.decode 1024^2 uint8 image: mean |d| {du8.mean().item():.3f} max {du8.max().item():.0f} "
          f"identical {(du8 == 0).float().mean().item()*100:.1f}%")
    assert du8.mean().item() <= 0.6


def test_configs4_geometry_one_layer():
    """BASELINE configs[4] geometry: 1328x1328 -> 83x83 = 6889 noise tokens (odd, not a multiple of 16) + a 1024x1024
    edit image (4096 tokens), T = 512 with 64 special tokens: S = 11497.  One layer so the oracle finishes in a minute."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from physicedit_amd.dit import QwenImageDiTEngine, special_indices
    sd = synth.make_state_dict(synth.dit_layout(1), 1234)
    ad = synth.make_state_dict(synth.adapter_layout(), 4321)
    t_min, t_max = O.adapter_t_range()
    noise, edit, pe, mask = _inputs(1328, 1328, 1024, 1024, 512, 64, 21)
    t = torch.tensor([940.0]).to(BF)
    pe_ref = pe.clone()
    ref = O.model_fn(sd, ad, noise, t, pe_ref, mask, 1328, 1328, edit, t_min, t_max)
    ref32 = O.model_fn({k: v.float() for k, v in sd.items()}, {k: v.float() for k, v in ad.items()}, noise.float(), t.float(),
                       pe.clone().float(), mask, 1328, 1328, edit.float(), t_min, t_max)
    eng = QwenImageDiTEngine(sd, ad, device="cuda")
    pe_run = pe.cuda().clone()
    got = eng.forward(noise.cuda(), t, pe_run, special_indices(mask, "cuda"), edit.cuda())
    st = record("configs[4]", "1328x1328 geometry (S = 11497), 1 layer, one model_fn call", got, ref, ref32)
    assert torch.equal(pe_run[0, ~mask[0].cuda()].cpu(), pe[0, ~mask[0]])
    assert st["fp32_distance_ratio"] <= 1.25 and st["max_ulp"] <= 16.0 and st["mean_abs_diff"] <= 4e-3, st
