"""`-m gpu` parity of the VAE kernels (csrc/vae.hip) against torch-CPU ops / the oracle / reference
golden vectors (tests/golden/G7_vae, generated from the imported reference)."""
import pytest
import torch
import torch.nn.functional as F

import oracle.physicedit_oracle as O
from physicedit_amd import synth
from test_gpu_kernels import report, rnd, ulps

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def vae_mod():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from physicedit_amd import vae
    from physicedit_amd._lib import lib
    lib()
    return vae


def _nhwc(x_nchw, cp):
    _, C, H, W = x_nchw.shape
    out = torch.zeros((H * W, cp), dtype=BF)
    out[:, :C] = x_nchw[0].permute(1, 2, 0).reshape(H * W, C)
    return out.cuda()


def _run_conv(vae_mod, x, w, b, stride=1, upsample=False, res=None):
    from physicedit_amd._lib import check, lib, stream_ptr
    conv = vae_mod._Conv(w, b, "cuda")
    _, C, H, W = x.shape
    xin = _nhwc(x, conv.cin_p)
    Ho, Wo = (H // 2, W // 2) if stride == 2 else ((H * 2, W * 2) if upsample else (H, W))
    out = torch.empty((Ho * Wo, conv.cout_p), dtype=BF, device="cuda")
    zero = torch.zeros((256,), dtype=BF, device="cuda")
    r = None if res is None else _nhwc(res, conv.cout_p)
    check(lib().pe_conv2d_nhwc(xin.data_ptr(), conv.w.data_ptr(), conv.b.data_ptr(), None if r is None else r.data_ptr(),
                               out.data_ptr(), zero.data_ptr(), H, W, conv.cin_p, conv.cout_p, conv.k, stride,
                               1 if upsample else 0, stream_ptr()))
    got = out[:, :conv.cout].reshape(Ho, Wo, conv.cout).permute(2, 0, 1)[None]
    assert torch.count_nonzero(out[:, conv.cout:]).item() == 0       # pad channels stay zero
    return got


@pytest.mark.parametrize("cin,cout,H,W", [(96, 96, 24, 40), (192, 384, 16, 16), (3, 96, 32, 32), (384, 32, 9, 13),
                                          (16, 384, 8, 8), (96, 3, 40, 24)])
def test_conv3x3(vae_mod, cin, cout, H, W):
    x = rnd((1, cin, H, W), 1)
    w = rnd((cout, cin, 3, 3), 2, (cin * 9) ** -0.5)
    b = rnd((cout,), 3, 0.1)
    ref = F.conv2d(x, w, b, padding=1)
    report(f"conv3x3 {cin}->{cout} {H}x{W}", _run_conv(vae_mod, x, w, b), ref, 1.01, 0.03)


def test_conv_variants(vae_mod):
    x = rnd((1, 96, 16, 24), 4)
    w = rnd((96, 96, 3, 3), 5, (96 * 9) ** -0.5)
    b = rnd((96,), 6, 0.1)
    # stride 2 with ZeroPad2d((0,1,0,1))  (qwen_image_vae.py:249)
    ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
    report("conv3x3 stride2", _run_conv(vae_mod, x, w, b, stride=2), ref, 1.01, 0.03)
    # nearest-exact 2x upsample fused into the gather (:240)
    up = F.interpolate(x.float(), scale_factor=(2.0, 2.0), mode="nearest-exact").to(BF)
    w2 = rnd((64, 96, 3, 3), 7, (96 * 9) ** -0.5)[:48]
    ref = F.conv2d(up, w2, b[:48], padding=1)
    report("conv3x3 upsample2x", _run_conv(vae_mod, x, w2, b[:48], upsample=True), ref, 1.01, 0.03)
    # residual add after the conv's own rounding (:152)
    res = rnd((1, 96, 16, 24), 8)
    ref = F.conv2d(x, w, b, padding=1) + res
    report("conv3x3 + residual", _run_conv(vae_mod, x, w, b, res=res), ref, 1.01, 0.03)
    # 1x1 from the 5-D causal weight (last temporal tap)
    w5 = rnd((192, 96, 1, 1, 1), 9, 96 ** -0.5)
    ref = F.conv2d(x, w5[:, :, 0], b.repeat(2), padding=0)
    report("conv1x1", _run_conv(vae_mod, x, w5, b.repeat(2)), ref, 1.01, 0.03)


@pytest.mark.parametrize("C", [96, 192, 384])
def test_vae_rmsnorm(vae_mod, C):
    from physicedit_amd._lib import check, lib, stream_ptr
    x = rnd((1, C, 13, 7), 10 + C, 3.0)
    gamma = synth.make_tensor(5, "norm.gamma", (C, 1, 1))
    for silu in (0, 1):
        ref = O._rms_norm_c(x, gamma)
        if silu:
            ref = F.silu(ref)
        xin = _nhwc(x, C)
        out = torch.empty_like(xin)
        check(lib().pe_vae_rmsnorm(xin.data_ptr(), gamma.reshape(-1).cuda().data_ptr(), out.data_ptr(), 13 * 7, C, C, silu,
                                   stream_ptr()))
        got = out.reshape(13, 7, C).permute(2, 0, 1)[None]
        report(f"vae_rmsnorm C={C} silu={silu}", got, ref, 1.01, 0.01)


@pytest.mark.parametrize("N", [64, 100, 1000])
def test_vae_attention(vae_mod, N):
    from physicedit_amd._lib import check, lib, stream_ptr
    qkv = rnd((N, 1152), 20 + N, 0.7)
    q, k, v = qkv[None, None].chunk(3, dim=-1)
    ref = F.scaled_dot_product_attention(q, k, v)[0, 0]
    ref32 = F.scaled_dot_product_attention(q.float(), k.float(), v.float())[0, 0]
    vt = torch.empty((int(lib().pe_vae_attention_scratch_bytes(N)) + 256,), dtype=torch.uint8, device="cuda")
    out = torch.empty((N, 384), dtype=BF, device="cuda")
    check(lib().pe_vae_attention(qkv.cuda().data_ptr(), (vt.data_ptr() + 255) // 256 * 256, out.data_ptr(), N, stream_ptr()))
    report(f"vae_attention N={N}", out, ref, 3.01, 0.50)   # P is bf16 in both; summation order differs (round 6: the keys of a query block are split over up to 4 work-groups)
    e_gpu = (out.float().cpu() - ref32).pow(2).mean().sqrt().item()
    e_cpu = (ref.float() - ref32).pow(2).mean().sqrt().item()
    print(f"[parity] vae_attention N={N}: rms err vs fp32 truth  hip {e_gpu:.3e}  cpu-bf16-sdpa {e_cpu:.3e}")
    assert e_gpu <= 1.5 * e_cpu + 1e-6


def test_vae_encode_decode_G7(vae_mod, golden):
    """Full encoder / decoder vs the REFERENCE's outputs (conv3d path) and vs the oracle's 2-D form."""
    g = golden("G7_vae")
    vs = synth.make_state_dict(synth.vae_layout(), 77)
    v = vae_mod.QwenImageVAE(vs, device="cuda")
    for R in (64, 96):
        x = O.preprocess_image(synth.make_edit_image_u8(R, R, seed=R))
        gen = torch.Generator().manual_seed(R)
        lat = torch.randn((1, 16, R // 8, R // 8), generator=gen).to(BF)
        z = v.encode(x.cuda())
        y = v.decode(lat.cuda())
        sd32 = {k: t.float() for k, t in vs.items()}
        O.VAE_CONV_MODE = "2d"
        try:
            z32, y32 = O.vae_encode(sd32, x.float()), O.vae_decode(sd32, lat.float())
        finally:
            O.VAE_CONV_MODE = "3d"
        for name, got, ref, ref32 in (("enc", z, g[f"enc_{R}"], z32), ("dec", y, g[f"dec_{R}"], y32)):
            d = (got.float().cpu() - ref.float()).abs()
            u = ulps(got, ref)
            e_hip = (got.float().cpu() - ref32).pow(2).mean().sqrt().item()
            e_ref = (ref.float() - ref32).pow(2).mean().sqrt().item()
            print(f"[parity] vae.{name} {R}x{R}: max|d| {d.max().item():.4e} mean|d| {d.mean().item():.4e} exact "
                  f"{(d == 0).float().mean().item()*100:.1f}% max {u.max().item():.1f} ulp; rms to fp32 run: hip {e_hip:.3e} "
                  f"reference-bf16 {e_ref:.3e}")
            assert torch.isfinite(got.float()).all()
            # 30-37 conv layers deep, one-ulp flips decorrelate element-wise: the criterion is the distance
            # to the fp32 run (must match the reference's own) plus a bound on the worst element
            assert e_hip <= 1.3 * e_ref + 1e-4
            assert d.max().item() <= 10.0 * e_ref + 1e-3


def test_image_io(vae_mod, golden):
    g = golden("G10_image")
    import numpy as np
    ramp = (np.arange(16 * 16 * 3) % 256).astype("uint8").reshape(16, 16, 3)
    assert torch.equal(vae_mod.preprocess_image(ramp, "cuda").cpu(), g["pre"])
    assert torch.equal(vae_mod.vae_output_to_u8(g["post_in"].cuda()), g["post_u8"])


def test_vae_fused_u8_io(vae_mod):
    """pe_vae_encode(PE_IMAGE_U8_HWC) / pe_vae_decode(PE_IMAGE_U8_HWC): BasePipeline.preprocess_image /
    vae_output_to_image fused into the composite's first / last kernel must be bit-identical to the stand-alone maps
    (themselves pinned on the reference's outputs, G10)."""
    vs = dict(synth.make_state_dict(synth.vae_layout(), 77))
    vs["decoder.conv_out.bias"] = torch.tensor([4.0, -4.0, 0.0]).to(BF)   # channel 0 clips at 255, channel 1 at 0
    v = vae_mod.QwenImageVAE(vs, device="cuda")
    u8 = synth.make_edit_image_u8(96, 64, seed=3)                      # H != W on purpose
    z_ref = v.encode(vae_mod.preprocess_image(u8, "cuda"))
    z_u8 = v.encode(torch.from_numpy(u8).cuda())
    assert torch.equal(z_ref, z_u8)
    gen = torch.Generator().manual_seed(9)
    lat = torch.randn((1, 16, 12, 8), generator=gen).to(BF).cuda()
    img = v.decode(lat)
    got = v.decode(lat, output_u8=True)
    ref = vae_mod.vae_output_to_u8(img)
    assert got.shape == (96, 64, 3) and got.dtype == torch.uint8
    assert torch.equal(got.cpu(), ref)
    assert (ref == 0).any() and (ref == 255).any()


def test_adapter_forward_cabi(golden):
    """pe_adapter_forward (VisualThinkingDualAdapter.forward, helpers.py:152-164) vs the reference's outputs (G9)."""
    import ctypes as C
    from physicedit_amd import _lib
    from physicedit_amd._lib import AdapterWeights, check, lib, stream_ptr
    from physicedit_amd.scheduler import adapter_alpha
    g = golden("G9_adapter")
    ad = {k: t.cuda() for k, t in synth.make_state_dict(synth.adapter_layout(), 4321).items()}
    a = AdapterWeights()
    a.dino_w0, a.dino_b0 = ad["head_dino.0.weight"].data_ptr(), ad["head_dino.0.bias"].data_ptr()
    a.dino_w2, a.dino_b2 = ad["head_dino.2.weight"].data_ptr(), ad["head_dino.2.bias"].data_ptr()
    a.vae_w0, a.vae_b0 = ad["head_vae.0.weight"].data_ptr(), ad["head_vae.0.bias"].data_ptr()
    a.vae_w2, a.vae_b2 = ad["head_vae.2.weight"].data_ptr(), ad["head_vae.2.bias"].data_ptr()
    gen = torch.Generator().manual_seed(9)
    x = torch.randn((1, 64, 3584), generator=gen).to(BF)[0].cuda().contiguous()
    n = lib().pe_adapter_workspace_bytes(64)
    ws = torch.empty((n,), dtype=torch.uint8, device="cuda")
    t_min, t_max = O.adapter_t_range()
    for tv in (1000.0, 748.0, 20.0):
        al, om = adapter_alpha(torch.tensor([tv]).to(BF), t_min, t_max)
        out = torch.empty_like(x)
        check(lib().pe_adapter_forward(C.byref(a), x.data_ptr(), 64, al, om, out.data_ptr(), ws.data_ptr(), n, stream_ptr()),
              "pe_adapter_forward")
        report(f"pe_adapter_forward t={tv}", out, g[f"mixed_{int(tv)}"][0], 3.0, 0.08)
