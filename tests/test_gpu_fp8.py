"""`-m gpu` tests of the e4m3 ("FP8 computation") Linear path: row quantiser and e4m3 GEMM through the C-ABI against the
oracle's restatement of AutoWrappedLinear.fp8_linear (vram_management/layers.py:115-151).

torch._scaled_mm has no CPU implementation with per-row scales, so the reference cannot emit golden vectors for the matmul
in the build container; it is pinned HERE, on the device, against the call the reference makes: pe_gemm_e4m3 must equal
torch._scaled_mm on the same e4m3 operands and scales (measured bit-identical, test_gemm_e4m3_vs_torch_scaled_mm).  The
quantiser is pure element-wise arithmetic and is compared BIT-EXACTLY, both with the oracle and with torch's own device ops.
"""
import pytest
import torch
import torch.nn.functional as F

import oracle.physicedit_oracle as O
from test_gpu_kernels import BF, report, rnd, ulps

pytestmark = pytest.mark.gpu
F8 = torch.float8_e4m3fn


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from physicedit_amd import ops as _ops
    from physicedit_amd._lib import lib
    lib()
    return _ops


def _acts(M, K, seed):
    """activations with the features that matter: rows above and below the 448 clamp, an all-zero row, denormal-range
    values after scaling, exact 448 multiples."""
    x = rnd((M, K), seed, 3.0).float()
    x[0] = 0.0
    if M > 1:
        x[1] *= 400.0                    # x_max far above 448 -> scale > 1
    if M > 2:
        x[2] *= 2.0 ** -9                # lands in e4m3's denormal range
    if M > 3:
        x[3, 0] = 448.0
    if M > 4:
        x[4, 5] = -30000.0
    return x.to(BF)


@pytest.mark.parametrize("M,K", [(5, 64), (33, 256), (300, 3072), (272, 3584), (130, 12288), (7, 8)])
def test_quantize_rows_bit_exact(ops, M, K):
    x = _acts(M, K, 11)
    xq_ref, sa_ref = O.fp8_quantize_rows(x)
    xq, sa = ops.quantize_rows_e4m3(x.cuda())
    Kp = xq.shape[1]
    assert Kp % 128 == 0 and Kp >= K
    assert torch.equal(sa.cpu(), sa_ref.reshape(-1)), "scale_a"
    got = xq.view(torch.uint8).cpu()
    assert torch.equal(got[:, :K], xq_ref.view(torch.uint8)), "e4m3 bytes"
    assert int(got[:, K:].max().item() if Kp > K else 0) == 0, "pad columns must be zero"
    # the same arithmetic with torch's device ops (what the reference executes on the GPU)
    xg = x.cuda()
    x_max = torch.max(torch.abs(xg), dim=-1, keepdim=True).values
    scale_t = torch.clamp(x_max / 448.0, min=1.0).float()
    xq_t = (xg / (scale_t + 1e-8)).to(F8)
    assert torch.equal(scale_t.reshape(-1), sa), "scale_a vs torch device ops"
    assert torch.equal(xq_t.view(torch.uint8), xq.view(torch.uint8)[:, :K]), "bytes vs torch device ops"


def test_ln_modulate_e4m3_fused_is_bit_identical(ops):
    """the fused LayerNorm-modulate + quantiser == pe_ln_modulate followed by pe_quantize_rows_e4m3, bit for bit."""
    rows, dim, rows_a = 301, 3072, 200
    x = _acts(rows, dim, 41).cuda()
    sh_a, sc_a, sh_b, sc_b = (rnd((dim,), 42 + i, 0.5).cuda() for i in range(4))
    ref = ops.ln_modulate(x, sh_a, sc_a, rows_a, sh_b, sc_b)
    q_ref, s_ref = ops.quantize_rows_e4m3(ref)
    out, q, s = ops.ln_modulate_e4m3(x, sh_a, sc_a, rows_a, sh_b, sc_b)
    assert torch.equal(out, ref)
    assert torch.equal(s, s_ref)
    assert torch.equal(q.view(torch.uint8), q_ref.view(torch.uint8))
    none, q2, s2 = ops.ln_modulate_e4m3(x, sh_a, sc_a, rows_a, sh_b, sc_b, want_bf16=False)
    assert none is None and torch.equal(q2.view(torch.uint8), q.view(torch.uint8)) and torch.equal(s2, s)


@pytest.mark.parametrize("M,N,K,gain", [(300, 1024, 512, 1.0), (520, 12288, 3072, 1.0), (300, 1024, 512, 400.0), (2100, 3072, 256, 60.0)])
def test_gemm_e4m3_gelu_fused_quantisation_is_bit_identical(ops, M, N, K, gain):
    """The MLP-up epilogue also emits the MLP-down Linear's e4m3 operand (pe_gemm_e4m3_gelu_q8): bf16 output identical to the plain
    GELU epilogue, (e4m3 bytes, scales) identical to pe_quantize_rows_e4m3 of that output.  gain > 1 scales the weights so that
    rows exceed 447 (scale > 1): those take the flagged re-quantisation pass, the others the direct one -- both in one call."""
    x = _acts(M, K, 61)
    w8 = rnd((N, K), 62, K ** -0.5 * gain).to(F8)
    if gain > 1.0:
        w8.view(torch.uint8)[N // 2:] = (rnd((N - N // 2, K), 63, K ** -0.5).to(F8)).view(torch.uint8)    # only half the columns are hot
    b = rnd((N,), 64, 0.1)
    xq, sa = ops.quantize_rows_e4m3(x.cuda())
    ref = ops.gemm_e4m3(xq, sa, w8.cuda(), b.cuda(), "gelu_sigmoid")
    q_ref, s_ref = ops.quantize_rows_e4m3(ref)
    out, q, s, flags = ops.gemm_e4m3_gelu_q8(xq, sa, w8.cuda(), b.cuda())
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert torch.equal(s, s_ref)
    assert torch.equal(q.view(torch.uint8), q_ref.view(torch.uint8))
    assert int(flags.abs().sum().item()) == 0                       # lowered again for the next launch
    n_scaled = int((s_ref > 1.0).sum().item())
    print(f"[parity] fused GELU quantisation {M}x{N}x{K} gain {gain}: {n_scaled} of {M} rows with scale > 1")
    assert 0 < n_scaled < M              # both paths taken in the same call (_acts plants one huge row even at gain 1)


def test_e4m3_conversion_all_values(ops):
    """every e4m3 value and every rounding midpoint between neighbours goes through the hardware conversion."""
    vals = torch.arange(0, 256, dtype=torch.uint8).view(F8).float()
    vals = vals[torch.isfinite(vals)]
    v = torch.sort(vals).values
    mids = (v[1:] + v[:-1]) / 2
    pts = torch.cat([v, mids, torch.nextafter(mids, mids + 1), torch.nextafter(mids, mids - 1)])
    pts = pts[pts.abs() <= 448].to(BF).float().unique()          # bf16-representable probes, no scaling (max <= 448)
    K = (pts.numel() + 7) // 8 * 8
    x = torch.zeros((1, K), dtype=BF)
    x[0, :pts.numel()] = pts.to(BF)
    xq, sa = ops.quantize_rows_e4m3(x.cuda())
    assert sa.item() == 1.0
    assert torch.equal(xq.view(torch.uint8).cpu()[:, :K], x.float().to(F8).view(torch.uint8))


def test_gemm_e4m3_exact_small_integers(ops):
    """small integers: every product and partial sum is exact in any accumulator -> catches layout errors exactly."""
    M, N, K = 70, 264, 256
    g = torch.Generator().manual_seed(5)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-2, 3, (N, K), generator=g).float()
    sa = torch.ones((M,), dtype=torch.float32)
    out = ops.gemm_e4m3(a.to(F8).cuda(), sa.cuda(), w.to(F8).cuda())
    assert torch.equal(out.float().cpu(), (a @ w.t()).to(BF).float())


@pytest.mark.parametrize("M,N,K", [(300, 3072, 3072), (257, 264, 128), (40, 18432, 3072), (1, 3072, 256),
                                   (520, 12288, 3072), (272, 3072, 12288)])
def test_fp8_linear(ops, M, N, K):
    x = _acts(M, K, 21)
    w8 = rnd((N, K), 22, K ** -0.5).to(F8)
    b8 = rnd((N,), 23, 0.1).to(F8)
    ref = O.fp8_linear(x, w8, b8)
    out = ops.fp8_linear(x.cuda(), w8.cuda(), b8.to(BF).cuda())
    # The block-scaled MFMA aligns the 64 products of a block to the largest one before adding (measured: ~2^-14 of
    # the largest term is dropped), where the oracle's sum is exact.  Bound: 2 bf16 ulp at operand scale, <= 6 % of
    # the outputs off by one.
    report(f"fp8_linear {M}x{N}x{K}", out, ref, max_ulp=2.01, max_frac=0.06)


def test_fp8_linear_epilogues(ops):
    M, N, K = 300, 1024, 512
    x = _acts(M, K, 31)
    w8, b8 = rnd((N, K), 32, K ** -0.5).to(F8), rnd((N,), 33, 0.1).to(F8)
    gate, res = rnd((N,), 34), rnd((M, N), 35)
    y = O.fp8_linear(x, w8, b8)
    xc, wc, bc = x.cuda(), w8.cuda(), b8.to(BF).cuda()
    report("fp8 gelu_sigmoid", ops.fp8_linear(xc, wc, bc, "gelu_sigmoid"), y * torch.sigmoid(1.702 * y), 2.01, 0.06)
    report("fp8 silu", ops.fp8_linear(xc, wc, bc, "silu"), F.silu(y), 2.01, 0.06)
    report("fp8 gate_res", ops.fp8_linear(xc, wc, bc, "gate_res", gate=gate.cuda(), res=res.cuda()), res + gate * y,
           2.01, 0.06)


# ------------------------------------------------------------------------------------------------
# the DiT composite in e4m3 mode (pe_dit_weights.weights_e4m3)
# ------------------------------------------------------------------------------------------------
def _inputs(h, w, T, n_special, seed):
    from physicedit_amd import synth
    noise = synth.make_noise(seed, h, w)
    g = torch.Generator().manual_seed(seed + 100)
    edit = torch.randn((1, 16, h // 8, w // 8), generator=g).to(BF)
    return noise, edit, synth.make_prompt_emb(seed + 7, T), synth.make_special_token_mask(T, n_special)


def _dist(name, got, ref):
    d = (got.float().cpu() - ref.float().cpu()).abs()
    print(f"[parity] {name}: max|d| {d.max().item():.4e} mean|d| {d.mean().item():.4e} "
          f"exact {(d == 0).float().mean().item()*100:.2f}%")
    return d


@pytest.mark.parametrize("hot_lora", [False, True])
def test_model_fn_e4m3(hot_lora):
    """2-layer full-width DiT with every Linear on the e4m3 path (time MLP, modulation, img_in with its K=64 padded
    to 128, txt_in, QKV, out, MLP, norm_out, proj_out) and the adapter in bf16, vs the oracle run on the e4m3
    state-dict.  The yardstick is the effect of the mode itself: |oracle_e4m3 - oracle_bf16|."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from physicedit_amd import synth
    from physicedit_amd.dit import QwenImageDiTEngine, special_indices
    sd = synth.make_state_dict(synth.dit_layout(2), 1234)
    ad = synth.make_state_dict(synth.adapter_layout(), 4321)
    lora = synth.make_lora(4321, 2, 16, std=0.05) if hot_lora else None
    t_min, t_max = O.adapter_t_range()
    noise, edit, pe, mask = _inputs(128, 128, 40, 8, 3)
    t = torch.tensor([700.0]).to(BF)
    sd_ref = O.attach_hot_lora(sd, lora) if hot_lora else sd
    ref_bf16 = O.model_fn(sd_ref, ad, noise, t, pe.clone(), mask, 128, 128, edit, t_min, t_max)
    ref = O.model_fn(O.to_fp8_state_dict(sd_ref), ad, noise, t, pe.clone(), mask, 128, 128, edit, t_min, t_max)
    eng = QwenImageDiTEngine(sd, ad, device="cuda")
    if hot_lora:
        assert eng.load_lora(lora, hotload=True) == 24
    eng.enable_fp8_computation()
    assert eng.fp8 and eng.params["img_in.weight"].dtype == F8 and eng.params["img_in.weight"].shape == (3072, 128)
    got = eng.forward(noise.cuda(), t, pe.cuda().clone(), special_indices(mask, "cuda"), edit.cuda())
    mode = _dist(f"e4m3 mode effect (oracle e4m3 vs oracle bf16){' +hot LoRA' if hot_lora else ''}", ref, ref_bf16)
    d = _dist(f"model_fn e4m3 vs oracle e4m3{' +hot LoRA' if hot_lora else ''}", got, ref)
    assert torch.isfinite(got.float()).all()
    # parity error must be well inside the mode's own quantisation effect
    assert d.mean().item() <= 0.25 * mode.mean().item(), (d.mean().item(), mode.mean().item())
    assert d.max().item() <= mode.max().item()


def test_model_fn_e4m3_with_bf16_controlnet():
    """"FP8 computation" + block-wise ControlNet: the reference gives the fp8 computation dtype to the DiT only, the ControlNet's
    Linears keep the pipeline dtype (qwen_image_physical.py:478-493) -- e4m3 block Linears, bf16 hook -- vs the oracle on the e4m3 DiT
    state dict with the bf16 ControlNet."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from physicedit_amd import synth
    from physicedit_amd.controlnet import QwenImageBlockWiseControlNet
    from physicedit_amd.dit import QwenImageDiTEngine, special_indices
    sd = synth.make_state_dict(synth.dit_layout(2), 1234)
    ad = synth.make_state_dict(synth.adapter_layout(), 4321)
    cn_sd = synth.make_state_dict(synth.controlnet_layout(2, 0), 555)
    t_min, t_max = O.adapter_t_range()
    noise, edit, pe, mask = _inputs(128, 128, 40, 8, 3)
    cond = torch.randn((1, 16, 16, 16), generator=torch.Generator().manual_seed(9)).to(BF)
    t = torch.tensor([700.0]).to(BF)
    ctl = [{"sd": cn_sd, "conditioning": cond, "scale": 0.8, "start": 1.0, "end": 0.0}]
    ref_plain = O.model_fn(O.to_fp8_state_dict(sd), ad, noise, t, pe.clone(), mask, 128, 128, edit, t_min, t_max)
    ref = O.model_fn(O.to_fp8_state_dict(sd), ad, noise, t, pe.clone(), mask, 128, 128, edit, t_min, t_max, controlnets=ctl)
    eng = QwenImageDiTEngine(sd, ad, device="cuda")
    eng.enable_fp8_computation()
    net = QwenImageBlockWiseControlNet(cn_sd, device="cuda")
    got = eng.forward(noise.cuda(), t, pe.cuda().clone(), special_indices(mask, "cuda"), edit.cuda(),
                      controls=[(net, net.process_controlnet_conditioning(cond), 0.8)])
    effect = _dist("e4m3 + controlnet: effect of the ControlNet (oracle with vs without)", ref, ref_plain)
    d = _dist("e4m3 + controlnet: model_fn vs oracle", got, ref)
    assert torch.isfinite(got.float()).all()
    assert d.mean().item() <= 0.25 * effect.mean().item() and d.max().item() <= effect.max().item()


def test_loop_e4m3_dual_stream_bit_identical():
    """e4m3 mode through the sampler loop (fork()ed second context shares the e4m3 weights): finite output, and
    dual-stream == single-stream bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from physicedit_amd import synth
    from physicedit_amd.dit import QwenImageDiTEngine
    from physicedit_amd.pipeline import DenoiseLoop
    sd = synth.make_state_dict(synth.dit_layout(2), 1234)
    ad = synth.make_state_dict(synth.adapter_layout(), 4321)
    eng = QwenImageDiTEngine(sd, ad, device="cuda")
    eng.enable_fp8_computation()
    noise, edit, pe, mask = _inputs(128, 128, 40, 8, 5)
    _, _, pe_n, mask_n = _inputs(128, 128, 24, 8, 6)
    outs = []
    for dual in (False, True):
        loop = DenoiseLoop(eng, dual_stream=dual)
        outs.append(loop(noise, pe.cuda().clone(), pe_n.cuda().clone(), mask, mask_n, 128, 128, num_inference_steps=3,
                         cfg_scale=4.0, edit_latents=edit.cuda()).clone())
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0], outs[1])


def test_model_fn_e4m3_full_shape_one_layer():
    """configs[2] geometry through the e4m3 composite (S_img = 8192, T = 512, 64 special tokens), one layer: every
    full-size e4m3 launch shape incl. the fused LayerNorm->e4m3 producer, vs the oracle on the e4m3 state-dict."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from physicedit_amd import synth
    from physicedit_amd.dit import QwenImageDiTEngine, special_indices
    sd = synth.make_state_dict(synth.dit_layout(1), 1234)
    ad = synth.make_state_dict(synth.adapter_layout(), 4321)
    t_min, t_max = O.adapter_t_range()
    noise, edit, pe, mask = _inputs(1024, 1024, 512, 64, 11)
    t = torch.tensor([860.0]).to(BF)
    ref16 = O.model_fn(sd, ad, noise, t, pe.clone(), mask, 1024, 1024, edit, t_min, t_max)
    ref = O.model_fn(O.to_fp8_state_dict(sd), ad, noise, t, pe.clone(), mask, 1024, 1024, edit, t_min, t_max)
    eng = QwenImageDiTEngine(sd, ad, device="cuda")
    eng.enable_fp8_computation()
    got = eng.forward(noise.cuda(), t, pe.cuda().clone(), special_indices(mask, "cuda"), edit.cuda())
    mode = _dist("full shape: e4m3 mode effect (oracle e4m3 vs oracle bf16)", ref, ref16)
    d = _dist("full shape: model_fn e4m3 vs oracle e4m3", got, ref)
    assert torch.isfinite(got.float()).all()
    assert d.mean().item() <= 0.25 * mode.mean().item() and d.max().item() <= mode.max().item()


@pytest.mark.parametrize("M,N,K", [(300, 3072, 3072), (520, 12288, 3072), (272, 3072, 12288), (8704, 3072, 3072)])
def test_gemm_e4m3_vs_torch_scaled_mm(ops, M, N, K):
    """Pin the e4m3 GEMM on what the REFERENCE executes on a GPU: torch._scaled_mm with per-row scale_a, unit scale_b,
    bf16 bias and bf16 output -- the exact call of AutoWrappedLinear.fp8_linear (vram_management/layers.py:141-148) --
    on the same e4m3 operands.  Both sides accumulate e4m3 products (exact in fp32) in some order on the device; the
    comparison is element-wise in bf16 ulps and the histogram goes into the parity record (configs[2])."""
    from parity_record import record
    x = _acts(M, K, 31)
    w8 = rnd((N, K), 32, K ** -0.5).to(F8).cuda()
    bias = rnd((N,), 33, 0.1).to(F8).to(BF).cuda()
    xq, sa = ops.quantize_rows_e4m3(x.cuda())
    assert xq.shape[1] == K
    try:
        ref = torch._scaled_mm(xq, w8.t(), scale_a=sa.reshape(M, 1), scale_b=torch.ones((1, N), device="cuda"), bias=bias,
                               out_dtype=BF)
    except Exception as e:            # rowwise-scaled fp8 GEMM not available in this torch/hipBLASLt build
        pytest.skip(f"torch._scaled_mm unavailable here: {type(e).__name__}: {str(e)[:120]}")
    out = ops.gemm_e4m3(xq, sa, w8, bias)
    u = ulps(out, ref)
    hist = {f"<= {k} ulp": float((u <= k).float().mean()) for k in (0, 1, 2)}
    st = record("configs[2]", f"pe_gemm_e4m3 vs torch._scaled_mm {M}x{N}x{K}", out, ref, ulp_histogram=hist)
    exact = O.fp8_linear(x, w8.cpu(), bias.cpu().to(F8))          # the oracle's exact (f64-summed) restatement
    st_t = record("configs[2]", f"torch._scaled_mm vs exact-sum oracle {M}x{N}x{K}", ref, exact)
    st_h = record("configs[2]", f"pe_gemm_e4m3 vs exact-sum oracle {M}x{N}x{K}", out, exact)
    assert u.max().item() <= 2.01 and hist["<= 0 ulp"] >= 0.97, (u.max().item(), hist)
    # and neither side is further from the exact sum than the other by more than a hair
    assert st_h["mean_abs_diff"] <= 1.5 * st_t["mean_abs_diff"] + 1e-6
