"""`-m gpu` parity tests: every HIP kernel, called through the C-ABI, against the CPU oracle on the
same seeded inputs.  Tolerances are in bf16 ulps of the reference value and are written next to each
assert; the only legitimate source of difference is fp32 summation order (and device exp/rsqrt
rounding), which flips a small fraction of bf16 roundings by one ulp.
"""
import functools
import math

import pytest
import torch
import torch.nn.functional as F

import oracle.physicedit_oracle as O
from physicedit_amd import synth

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from physicedit_amd import ops as _ops
    from physicedit_amd._lib import lib
    lib()  # raises (does not skip) if the HIP library is missing: no silent fallback
    return _ops


def ulps(got: torch.Tensor, ref: torch.Tensor, floor=None) -> torch.Tensor:
    """|got-ref| in bf16 ulps of max(|ref|, floor); floor defaults to rms(ref), the magnitude of the
    operands the last add/sub worked on (results of a cancellation are judged at operand scale)."""
    g, r = got.float().cpu(), ref.float().cpu()
    if floor is None:
        floor = max(r.pow(2).mean().sqrt().item(), 2.0 ** -20)
    _, e = torch.frexp(r.abs().clamp_min(floor))
    return (g - r).abs() / torch.exp2(e.float() - 8)


def report(name, got, ref, max_ulp, max_frac, floor=None):
    u = ulps(got, ref, floor)
    frac = (u > 0).float().mean().item()
    mx = u.max().item()
    print(f"[parity] {name}: mismatching {frac*100:.3f}% max {mx:.2f} ulp  (limits {max_frac*100:.1f}% / {max_ulp} ulp)")
    assert torch.isfinite(got.float()).all(), name
    assert mx <= max_ulp, (name, mx)
    assert frac <= max_frac, (name, frac)


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(BF)


# ------------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------------
def test_gemm_identity_asymmetric(ops):
    """A = I against an asymmetric W: catches operand/row-column transposes exactly (no rounding)."""
    K = 256
    A = torch.eye(K, dtype=BF)
    W = (torch.arange(320 * K, dtype=torch.float32).reshape(320, K) % 251 - 125).to(BF)  # exact in bf16
    out = ops.gemm(A.cuda(), W.cuda())
    assert torch.equal(out.cpu(), W.t().contiguous())


@pytest.mark.parametrize("M,N,K", [(300, 3072, 3072), (257, 264, 64), (40, 18432, 3072), (64, 64, 3072),
                                   (1, 3072, 256), (520, 12288, 3072), (272, 3072, 12288)])
def test_gemm_bias(ops, M, N, K):
    x, w, b = rnd((M, K), 1), rnd((N, K), 2, K ** -0.5), rnd((N,), 3, 0.1)
    ref = F.linear(x, w, b)
    out = ops.gemm(x.cuda(), w.cuda(), b.cuda())
    report(f"gemm_bias {M}x{N}x{K}", out, ref, max_ulp=1.01, max_frac=0.03)
    out2 = ops.gemm(x.cuda(), w.cuda(), None)
    report(f"gemm_nobias {M}x{N}x{K}", out2, F.linear(x, w), max_ulp=1.01, max_frac=0.03)


def test_gemm_epilogues(ops):
    M, N, K = 300, 1024, 512
    x, w, b = rnd((M, K), 4), rnd((N, K), 5, K ** -0.5), rnd((N,), 6, 0.1)
    y = F.linear(x, w, b)
    xc, wc, bc = x.cuda(), w.cuda(), b.cuda()
    # ApproximateGELU (qwen_image_dit.py:47-49)
    report("gemm+gelu_sigmoid", ops.gemm(xc, wc, bc, "gelu_sigmoid"), y * torch.sigmoid(1.702 * y), 2.01, 0.05)
    # nn.GELU() (helpers.py:127-131)
    report("gemm+gelu_erf", ops.gemm(xc, wc, bc, "gelu_erf"), F.gelu(y), 2.01, 0.05)
    report("gemm+silu", ops.gemm(xc, wc, bc, "silu"), F.silu(y), 2.01, 0.05)
    # gated residual (qwen_image_dit.py:386): image + gate * out
    gate, res = rnd((N,), 7, 0.5), rnd((M, N), 8)
    ref = res + gate.unsqueeze(0) * y
    report("gemm+gate_res", ops.gemm(xc, wc, bc, "gate_res", gate=gate.cuda(), res=res.cuda()), ref, 2.01, 0.05)
    # in place (res aliases out), as the block uses it
    r2 = res.cuda().clone()
    ops.gemm(xc, wc, bc, "gate_res", gate=gate.cuda(), res=r2, out=r2)
    report("gemm+gate_res in place", r2, ref, 2.01, 0.05)


@pytest.mark.parametrize("M,N,K", [(2100, 1288, 192), (2048, 3072, 3072), (300, 18432, 256), (8704, 12288, 3072)])
def test_gemm_persistent_schedule_bit_identical(ops, M, N, K):
    """Schedule 17 (persistent work-groups, next tile's first K tiles prefetched by the main loop's tail, two-pass epilogue in the
    two free LDS regions) performs schedule 15's arithmetic in the same order: bit-identical outputs for every epilogue, with a
    grid of 8 work-groups (every work-group walks many tiles, ragged ones and -- in the grouped QKV launches of the composite
    tests -- problem boundaries included) and with the production grid (one per CU; the last shape has 6.4 rounds of tiles)."""
    from physicedit_amd._lib import lib
    x, w, b = rnd((M, K), 1).cuda(), rnd((N, K), 2, K ** -0.5).cuda(), rnd((N,), 3, 0.1).cuda()
    gate, res = rnd((N,), 7, 0.5).cuda(), rnd((M, N), 8).cuda()
    try:
        for wgs in ((8, 0) if M < 8000 else (0,)):
            assert lib().pe_debug_set(b"gemm_persist_wgs", wgs) == 0
            for epi in ("bias", "gelu_sigmoid", "gate_res"):
                kw = dict(gate=gate, res=res) if epi == "gate_res" else {}
                assert lib().pe_debug_set(b"gemm_variant", 15) == 0
                a = ops.gemm(x, w, b, epi, **kw)
                assert lib().pe_debug_set(b"gemm_variant", 17) == 0
                for _ in range(3):                                   # + a short race screen
                    c = ops.gemm(x, w, b, epi, **kw)
                    assert torch.equal(a, c), (wgs, epi)
    finally:
        lib().pe_debug_set(b"gemm_variant", 0)
        lib().pe_debug_set(b"gemm_persist_wgs", 0)


@pytest.mark.parametrize("var", [21, 22])
def test_gemm_round5_schedules_bit_identical(ops, var):
    """Round 5's two opt-in main loops against schedule 15: 21 = the persistent ping-pong with ONE hand-off per K tile (32-MFMA slots),
    22 = four waves, one per SIMD, 128 x 128 per wave (gemm4.hip: the tiling whose attainable ceiling the probe ladder puts 7 % higher).
    Same K order per output element, so bit-identical: every epilogue incl. QKV (RMSNorm + RoPE + transposed V) and e4m3 operands,
    ragged / tiny / multi-round shapes, a short race screen."""
    from physicedit_amd._lib import lib
    try:
        # the four-wave kernel keeps the 32 x 32 x 16 MFMA (another summation order inside a K tile than the 16 x 16 x 32 the eight-wave
        # schedules use since round 5): its reference is schedule 15 in that shape
        assert lib().pe_debug_set(b"gemm_mfma16", 0 if var == 22 else 3) == 0
        for (M, N, K) in ((300, 3072, 3072), (257, 264, 64), (272, 3072, 12288), (1, 3072, 256), (8704, 3072, 3072), (2100, 12288, 3072)):
            x, w, b = rnd((M, K), 1).cuda(), rnd((N, K), 2, K ** -0.5).cuda(), rnd((N,), 3, 0.1).cuda()
            gate, res = rnd((N,), 7, 0.5).cuda(), rnd((M, N), 8).cuda()
            for epi in ("bias", "gelu_sigmoid", "gelu_erf", "silu", "gate_res"):
                kw = dict(gate=gate, res=res) if epi == "gate_res" else {}
                assert lib().pe_debug_set(b"gemm_variant", 15) == 0
                a = ops.gemm(x, w, b, epi, **kw)
                assert lib().pe_debug_set(b"gemm_variant", var) == 0
                for _ in range(3):
                    assert torch.equal(a, ops.gemm(x, w, b, epi, **kw)), (M, N, K, epi)
        for (M, seq_off) in ((300, 0), (2300, 0), (37, 135)):
            x, w, b = rnd((M, 3072), 11).cuda(), rnd((9216, 3072), 12, 3072 ** -0.5).cuda(), rnd((9216,), 13, 0.1).cuda()
            nq, nk = synth.make_tensor(5, "norm_q.weight", (128,)).cuda(), synth.make_tensor(5, "norm_k.weight", (128,)).cuda()
            _, txt = O.rope_tables([(1, 8, 8)], M)
            cos, sin = txt.real.contiguous().cuda(), txt.imag.contiguous().cuda()
            outs = {}
            for v in (15, var):
                assert lib().pe_debug_set(b"gemm_variant", v) == 0
                q, k, vt = ops.alloc_qkv(24, seq_off + M, "cuda")
                ops.qkv_rmsnorm_rope(x, w, b, nq, nk, cos, sin, q, k, vt, seq_off, q_scale=0.1275)
                outs[v] = (q, k, vt)
            assert all(torch.equal(u, v) for u, v in zip(outs[15], outs[var])), (M, seq_off)
        for (M, N, K) in ((300, 3072, 3072), (2300, 3072, 12288), (272, 256, 128)):
            x, w8, b = rnd((M, K), 21).cuda(), rnd((N, K), 22, K ** -0.5).cuda().to(torch.float8_e4m3fn), rnd((N,), 23, 0.1).cuda()
            xq, sc = ops.quantize_rows_e4m3(x)
            for epi in ("bias", "gelu_sigmoid"):
                assert lib().pe_debug_set(b"gemm_variant", 15) == 0
                a = ops.gemm_e4m3(xq, sc, w8, b, epi)
                assert lib().pe_debug_set(b"gemm_variant", var) == 0
                assert torch.equal(a, ops.gemm_e4m3(xq, sc, w8, b, epi)), (M, N, K, epi)
    finally:
        lib().pe_debug_set(b"gemm_variant", 0)
        lib().pe_debug_set(b"gemm_mfma16", 3)


def test_gemm_mfma_shapes_agree(ops):
    """The bf16 GEMM's two MFMA shapes (round 5: v_mfma_f32_16x16x32_bf16 by default, 32x32x16 behind "gemm_mfma16" = 0) accumulate a K tile's
    products in different orders: not bit-identical, but the same distance from the fp64 result -- and for each shape schedules 15 and 17 are
    bit-identical with each other, LDS and direct epilogue alike."""
    from physicedit_amd._lib import lib
    try:
        for (M, N, K) in ((2300, 3072, 3072), (300, 3072, 12288), (257, 264, 64)):
            x, w, b = rnd((M, K), 1).cuda(), rnd((N, K), 2, K ** -0.5).cuda(), rnd((N,), 3, 0.1).cuda()
            gate, res = rnd((N,), 7, 0.5).cuda(), rnd((M, N), 8).cuda()
            ref = (x.double() @ w.double().T + b.double())
            for epi in ("bias", "gate_res", "gelu_sigmoid"):
                kw = dict(gate=gate, res=res) if epi == "gate_res" else {}
                outs = {}
                for shape in (1, 0):
                    assert lib().pe_debug_set(b"gemm_mfma16", shape) == 0
                    for var in (15, 17) + ((21,) if shape else ()):
                        assert lib().pe_debug_set(b"gemm_variant", var) == 0
                        for direct in (1, 0):
                            assert lib().pe_debug_set(b"gemm_direct_epilogue", direct) == 0
                            o = ops.gemm(x, w, b, epi, **kw)
                            assert torch.equal(outs.setdefault(shape, o), o), (M, N, K, epi, shape, var, direct)
                if epi == "bias":
                    e16 = (outs[1].double() - ref).abs().mean().item()
                    e32 = (outs[0].double() - ref).abs().mean().item()
                    assert abs(e16 - e32) <= 0.02 * e32, (e16, e32)          # both are the bf16 rounding of the same fp32-accurate sum
                d = (outs[1].float() - outs[0].float()).abs()
                frac = (d == 0).float().mean().item()
                assert frac >= 0.97, (M, N, K, epi, frac)                     # they differ where the fp32 sums straddle a bf16 rounding boundary
                assert d.max().item() <= 2.0 ** -6 * max(1.0, outs[0].float().abs().max().item()), (M, N, K, epi)
        # e4m3 operands: v_mfma_scale_f32_16x16x128_f8f6f4 against 32x32x64
        for (M, N, K) in ((2300, 3072, 3072), (300, 3072, 12288), (272, 256, 128)):
            x, w8, b = rnd((M, K), 21).cuda(), rnd((N, K), 22, K ** -0.5).cuda().to(torch.float8_e4m3fn), rnd((N,), 23, 0.1).cuda()
            xq, sc = ops.quantize_rows_e4m3(x)
            for epi in ("bias", "gelu_sigmoid"):
                outs = {}
                for shape in (3, 1):      # bit 1: e4m3 on the 16 x 16 x 128 MFMA (default), clear: 32 x 32 x 64
                    assert lib().pe_debug_set(b"gemm_mfma16", shape) == 0
                    for var in (15, 17) + ((21,) if shape == 3 else ()):
                        assert lib().pe_debug_set(b"gemm_variant", var) == 0
                        for direct in (1, 0):
                            assert lib().pe_debug_set(b"gemm_direct_epilogue", direct) == 0
                            o = ops.gemm_e4m3(xq, sc, w8, b, epi)
                            assert torch.equal(outs.setdefault(shape, o), o), (M, N, K, epi, shape, var, direct)
                d = (outs[3].float() - outs[1].float()).abs()
                assert (d == 0).float().mean().item() >= 0.97, (M, N, K, epi)
                assert d.max().item() <= 2.0 ** -6 * max(1.0, outs[1].float().abs().max().item()), (M, N, K, epi)
    finally:
        lib().pe_debug_set(b"gemm_variant", 0)
        lib().pe_debug_set(b"gemm_mfma16", 3)
        lib().pe_debug_set(b"gemm_direct_epilogue", 1)


@pytest.mark.parametrize("var", [15, 17, 22])
def test_gemm_direct_epilogue_bit_identical(ops, var):
    """Round 5: complete tiles of the GELU and gate x y + residual epilogues go accumulators -> v_permlane32_swap -> 16-byte stores without
    the LDS round trip (gemm_tile.h gemm_epilogue_direct; knob gemm_direct_epilogue, default 1).  Same operations and roundings per element
    as the LDS form: bit-identical, with and without bias, vector and scalar gate, in place (residual aliases the output), ragged shapes
    (whose ragged tiles keep the LDS form) and e4m3 operands incl. the fused quantisation of the GELU output."""
    from physicedit_amd._lib import lib
    try:
        assert lib().pe_debug_set(b"gemm_variant", var) == 0
        for (M, N, K) in ((2100, 3072, 3072), (600, 12288, 3072), (8464, 3072, 3072)):
            x, w, b = rnd((M, K), 31).cuda(), rnd((N, K), 32, K ** -0.5).cuda(), rnd((N,), 33, 0.1).cuda()
            gate, res = rnd((N,), 37, 0.5).cuda(), rnd((M, N), 38).cuda()
            xq, sc = ops.quantize_rows_e4m3(x)
            w8 = w.to(torch.float8_e4m3fn)
            outs = []
            for d in (0, 1):
                assert lib().pe_debug_set(b"gemm_direct_epilogue", d) == 0
                r1 = res.clone()
                ops.gemm(x, w, b, "gate_res", gate=gate, res=r1, out=r1)
                o = [ops.gemm(x, w, b, "gelu_sigmoid"), ops.gemm(x, w, None, "gelu_sigmoid"), ops.gemm(x, w, b, "gate_res", gate=gate, res=res),
                     ops.gemm(x, w, b, "gate_res", gate=None, res=res), r1, ops.gemm_e4m3(xq, sc, w8, b, "gelu_sigmoid"),
                     ops.gemm_e4m3(xq, sc, w8, b, "gate_res", gate=gate, res=res)]
                o += list(ops.gemm_e4m3_gelu_q8(xq, sc, w8, b))
                outs.append(o)
            for i, (p_, q_) in enumerate(zip(*outs)):
                assert torch.equal(p_, q_), (var, M, N, K, i)
    finally:
        lib().pe_debug_set(b"gemm_direct_epilogue", 1)
        lib().pe_debug_set(b"gemm_variant", 0)


@pytest.mark.parametrize("var", [15, 17, 21])
def test_gemm_direct_qk_epilogue_bit_identical(ops, var):
    """Bit 1 of gemm_direct_epilogue: the q / k sections of the QKV epilogue (per-head RMSNorm + RoPE + the attention's pre-scale) without the
    LDS round trip in the 16 x 16 accumulator layout (gemm_epilogue_direct16_qk: the butterfly of the row's chunk sums runs over two lane
    swaps and two in-lane levels).  Bit-identical with the LDS form: complete and ragged tiles, a joint offset, bf16 and e4m3 operands."""
    from physicedit_amd._lib import lib
    try:
        assert lib().pe_debug_set(b"gemm_variant", var) == 0
        for (M, seq_off) in ((2300, 0), (8704, 0), (300, 0), (37, 135)):
            x, w, b = rnd((M, 3072), 11).cuda(), rnd((9216, 3072), 12, 3072 ** -0.5).cuda(), rnd((9216,), 13, 0.1).cuda()
            nq, nk = synth.make_tensor(5, "norm_q.weight", (128,)).cuda(), synth.make_tensor(5, "norm_k.weight", (128,)).cuda()
            _, txt = O.rope_tables([(1, 8, 8)], M)
            cos, sin = txt.real.contiguous().cuda(), txt.imag.contiguous().cuda()
            outs = {}
            for d in (1, 3):
                assert lib().pe_debug_set(b"gemm_direct_epilogue", d) == 0
                q, k, vt = ops.alloc_qkv(24, seq_off + M, "cuda")
                ops.qkv_rmsnorm_rope(x, w, b, nq, nk, cos, sin, q, k, vt, seq_off, q_scale=0.1275)
                outs[d] = (q, k, vt)
            assert all(torch.equal(u, v) for u, v in zip(outs[1], outs[3])), (var, M, seq_off)
    finally:
        lib().pe_debug_set(b"gemm_direct_epilogue", 1)
        lib().pe_debug_set(b"gemm_variant", 0)


@pytest.fixture
def gemm_workspace():
    """a zeroed stream-K workspace installed for the granular pe_gemm_* calls of one test (pe_debug_set_ptr), removed afterwards"""
    from physicedit_amd._lib import lib
    n = lib().pe_gemm_workspace_bytes()
    buf = torch.zeros((n + 256,), dtype=torch.uint8, device="cuda")
    off = (-buf.data_ptr()) % 256
    ws = buf[off:off + n]
    assert lib().pe_debug_set_ptr(b"gemm_workspace", ws.data_ptr()) == 0
    yield ws
    lib().pe_debug_set_ptr(b"gemm_workspace", None)
    lib().pe_debug_set(b"gemm_variant", 0)


@pytest.mark.parametrize("M,N,K", [(8704, 3072, 3072), (8464, 3072, 12288), (8704, 9216, 3072), (8704, 12288, 3072), (8200, 2048, 128),
                                   (16500, 1288, 192)])
def test_gemm_stream_k_bit_identical(ops, gemm_workspace, M, N, K):
    """Schedule 19 (stream-K: every CU gets the same number of (tile, K tile) units; a tile cut in two hands its fp32 accumulators
    over through the workspace and the second part continues the SAME accumulation) against schedule 15, one tile per work-group:
    bit-identical for every epilogue of the block's Linears, on the four block shapes of the headline geometry (1.6 / 1.6 / 4.8 /
    6.4 rounds of tiles), a 2-K-tile shape (every segment is one or two K tiles long: the no-prefetch paths) and a ragged one;
    repeated (race screen: uneven arrival of the two parts of a tile), and the workspace's ticket and flags are zero afterwards."""
    from physicedit_amd._lib import lib
    x, w, b = rnd((M, K), 1).cuda(), rnd((N, K), 2, K ** -0.5).cuda(), rnd((N,), 3, 0.1).cuda()
    gate, res = rnd((N,), 7, 0.5).cuda(), rnd((M, N), 8).cuda()
    for epi in ("bias", "gelu_sigmoid", "gate_res"):
        kw = dict(gate=gate, res=res) if epi == "gate_res" else {}
        assert lib().pe_debug_set(b"gemm_variant", 15) == 0
        a = ops.gemm(x, w, b, epi, **kw)
        assert lib().pe_debug_set(b"gemm_variant", 19) == 0
        for _ in range(6):
            c = ops.gemm(x, w, b, epi, **kw)
            assert torch.equal(a, c), epi
        torch.cuda.synchronize()
        assert torch.count_nonzero(gemm_workspace[:4096]).item() == 0


def test_qkv_stream_k_bit_identical(ops, gemm_workspace):
    """the QKV epilogue (RMSNorm + RoPE + transposed V, scaled Q) under schedule 19 at the headline shape: bit-identical to schedule 15"""
    from physicedit_amd._lib import lib
    M, K = 8704, 3072
    x, w, b = rnd((M, K), 11).cuda(), rnd((9216, K), 12, K ** -0.5).cuda(), rnd((9216,), 13, 0.1).cuda()
    nq, nk = synth.make_tensor(5, "norm_q.weight", (128,)).cuda(), synth.make_tensor(5, "norm_k.weight", (128,)).cuda()
    _, txt = O.rope_tables([(1, 8, 8)], M)
    cos, sin = txt.real.contiguous().cuda(), txt.imag.contiguous().cuda()
    outs = {}
    for var in (15, 19, 19):
        assert lib().pe_debug_set(b"gemm_variant", var) == 0
        q, k, vt = ops.alloc_qkv(24, M, "cuda")
        ops.qkv_rmsnorm_rope(x, w, b, nq, nk, cos, sin, q, k, vt, 0, q_scale=0.1275)
        if var in outs:
            assert all(torch.equal(u, v) for u, v in zip(outs[var], (q, k, vt)))
        outs[var] = (q, k, vt)
    assert all(torch.equal(u, v) for u, v in zip(outs[15], outs[19]))
    torch.cuda.synchronize()
    assert torch.count_nonzero(gemm_workspace[:4096]).item() == 0


def test_gemm_rejects_bad_shapes(ops):
    from physicedit_amd._lib import PeError
    x, w = rnd((8, 100), 1).cuda(), rnd((16, 100), 2).cuda()
    with pytest.raises(PeError):
        ops.gemm(x, w)                      # K % 64 != 0
    with pytest.raises(ValueError):
        ops.gemm(rnd((8, 64), 1), rnd((16, 64), 2).cuda())   # CPU tensor: no fallback


# ------------------------------------------------------------------------------------------------
# QKV epilogue + attention
# ------------------------------------------------------------------------------------------------
def _qkv_ref(x, w, b, nq, nk, freqs):
    y = F.linear(x, w, b)
    q, k, v = y.chunk(3, dim=-1)
    hd = lambda t: t.reshape(1, t.shape[0], 24, 128).permute(0, 2, 1, 3)
    q, k, v = hd(q), hd(k), hd(v)
    q = O.apply_rope(O.rmsnorm(q, nq), freqs)
    k = O.apply_rope(O.rmsnorm(k, nk), freqs)
    return q[0], k[0], v[0]


@pytest.mark.parametrize("M,seq_off", [(128, 0), (300, 0), (40, 128), (37, 135)])
def test_qkv_rmsnorm_rope(ops, M, seq_off):
    K = 3072
    x, w, b = rnd((M, K), 11), rnd((9216, K), 12, K ** -0.5), rnd((9216,), 13, 0.1)
    nq, nk = synth.make_tensor(5, "norm_q.weight", (128,)), synth.make_tensor(5, "norm_k.weight", (128,))
    _, txt = O.rope_tables([(1, 8, 8)], M)
    qr, kr, vr = _qkv_ref(x, w, b, nq, nk, txt)
    S = seq_off + M
    q, k, vt = ops.alloc_qkv(24, S, "cuda")
    ops.qkv_rmsnorm_rope(x.cuda(), w.cuda(), b.cuda(), nq.cuda(), nk.cuda(), txt.real.contiguous().cuda(),
                         txt.imag.contiguous().cuda(), q, k, vt, seq_off)
    report(f"qkv.q M={M} off={seq_off}", q[:, seq_off:S], qr, 2.01, 0.06)
    report(f"qkv.k M={M} off={seq_off}", k[:, seq_off:S], kr, 2.01, 0.06)
    report(f"qkv.v M={M} off={seq_off}", ops.unpack_vt(vt, S)[:, seq_off:S], vr, 1.01, 0.03)
    # nothing outside the rows of this problem may be touched
    assert torch.count_nonzero(q[:, :seq_off]).item() == 0 and torch.count_nonzero(q[:, S:]).item() == 0
    pos = ops.vt_positions(vt.shape[2], "cuda")
    mask = torch.ones(vt.shape[2], dtype=torch.bool, device="cuda")
    mask[pos[seq_off:S]] = False
    assert torch.count_nonzero(vt[:, :, mask]).item() == 0


def test_qkv_rmsnorm_rope_scaled(ops):
    """pe_qkv_rmsnorm_rope_scaled: Q = bf16(rope(q) . q_scale) with the factor applied in fp32 before the one rounding -- within one
    ulp of bf16(q_plain . q_scale), which rounds twice; K and Vt are untouched by the factor; q_scale = 1 is the plain operator."""
    M, K = 300, 3072
    x, w, b = rnd((M, K), 11), rnd((9216, K), 12, K ** -0.5), rnd((9216,), 13, 0.1)
    nq, nk = synth.make_tensor(5, "norm_q.weight", (128,)), synth.make_tensor(5, "norm_k.weight", (128,))
    _, txt = O.rope_tables([(1, 8, 8)], M)
    args = (x.cuda(), w.cuda(), b.cuda(), nq.cuda(), nk.cuda(), txt.real.contiguous().cuda(), txt.imag.contiguous().cuda())
    q0, k0, vt0 = ops.alloc_qkv(24, M, "cuda")
    ops.qkv_rmsnorm_rope(*args, q0, k0, vt0, 0)
    q1, k1, vt1 = ops.alloc_qkv(24, M, "cuda")
    ops.qkv_rmsnorm_rope(*args, q1, k1, vt1, 0, q_scale=1.0)
    assert torch.equal(q0, q1) and torch.equal(k0, k1) and torch.equal(vt0, vt1)
    c = (1.0 / math.sqrt(128.0)) * 1.4426950408889634
    q2, k2, vt2 = ops.alloc_qkv(24, M, "cuda")
    ops.qkv_rmsnorm_rope(*args, q2, k2, vt2, 0, q_scale=c)
    assert torch.equal(k0, k2) and torch.equal(vt0, vt2)
    report("qkv.q scaled vs bf16(q . c)", q2[:, :M], (q0[:, :M].float() * c).to(BF), 1.01, 0.60)
    assert not torch.equal(q2, q0)


ATTN_DEFAULT = 5      # attention.hip g_attn_variant


@pytest.fixture
def attn_variant():
    """selects a flash-attention kernel variant for one test (pe_debug_set) and restores the default"""
    from physicedit_amd._lib import lib

    def select(v):
        assert lib().pe_debug_set(b"attn_variant", v) == 0
    yield select
    lib().pe_debug_set(b"attn_variant", ATTN_DEFAULT)
    lib().pe_debug_set(b"attn_force_split", 0)


@pytest.mark.parametrize("S,variant", [(64, 4), (100, 4), (700, 4), (1093, 4), (2208, 4), (256, 0), (1093, 0)])
def test_flash_attn(ops, attn_variant, S, variant):
    """The criterion for EVERY variant is the one the repo enforces end to end: the rms distance to the fp32 result must be the
    bf16 reference's own (torch-CPU bf16 SDPA).  Variant 0 (textbook max update) rounds P at the reference's scale and is
    also held to <= 3 ulp element-wise; variant 4 (default, lazy max: P <= 2^8 instead of <= 1) rounds P at another scale,
    so fewer outputs are bit-identical to the reference's (bound 4 ulp) at the SAME distance to fp32."""
    H = 24
    q, k, v = rnd((H, S, 128), 21), rnd((H, S, 128), 22), rnd((H, S, 128), 23)
    ref = F.scaled_dot_product_attention(q[None], k[None], v[None])[0]           # [H,S,128]
    ref = ref.permute(1, 0, 2).reshape(S, H * 128)
    sp = ops.s_pad_of(S)
    qd = torch.zeros((H, sp, 128), dtype=BF, device="cuda"); qd[:, :S] = q.cuda()
    kd = torch.full((H, sp, 128), float("nan"), dtype=BF, device="cuda"); kd[:, :S] = k.cuda()   # pad rows are masked
    attn_variant(variant)
    out = ops.flash_attn(qd, kd, ops.pack_vt(v.cuda(), sp), S)
    if variant == 0:
        report(f"flash_attn v0 S={S}", out, ref, max_ulp=3.01, max_frac=0.50)
    else:
        report(f"flash_attn v{variant} S={S}", out, ref, max_ulp=4.01, max_frac=0.55)
    ref32 = F.scaled_dot_product_attention(q[None].float(), k[None].float(), v[None].float())[0]
    ref32 = ref32.permute(1, 0, 2).reshape(S, H * 128)
    e_gpu = (out.float().cpu() - ref32).pow(2).mean().sqrt().item()
    e_cpu = (ref.float() - ref32).pow(2).mean().sqrt().item()
    print(f"[parity] flash_attn v{variant} S={S}: rms err vs fp32 truth  hip {e_gpu:.3e}  cpu-bf16-sdpa {e_cpu:.3e}")
    assert e_gpu <= 1.1 * e_cpu + 1e-6


@pytest.mark.parametrize("S,force,variant", [(700, 3, 4), (1093, 5, 4), (300, 8, 4), (700, 3, 0)])
def test_flash_attn_split_kv(ops, attn_variant, S, force, variant):
    """Load-balancing path: leftover (head, q-block) items split along KV + combine kernel.  With the textbook max update
    (variant 0) a split item differs from the whole one only by the fp32 summation order (<= 2 ulp); with the lazy update
    (variant 4, default) each part also keeps its own running max, so P is rounded at part-dependent scales (<= 3.5 ulp)."""
    from physicedit_amd._lib import lib
    H = 24
    q, k, v = rnd((H, S, 128), 41), rnd((H, S, 128), 42), rnd((H, S, 128), 43)
    ref = F.scaled_dot_product_attention(q[None], k[None], v[None])[0].permute(1, 0, 2).reshape(S, H * 128)
    sp = ops.s_pad_of(S)
    qd = torch.zeros((H, sp, 128), dtype=BF, device="cuda"); qd[:, :S] = q.cuda()
    kd = torch.zeros((H, sp, 128), dtype=BF, device="cuda"); kd[:, :S] = k.cuda()
    vt = ops.pack_vt(v.cuda(), sp)
    attn_variant(variant)
    base = ops.flash_attn(qd, kd, vt, S, workspace=False)
    try:
        lib().pe_debug_set(b"attn_force_split", force)
        out = ops.flash_attn(qd, kd, vt, S, workspace=True)
    finally:
        lib().pe_debug_set(b"attn_force_split", 0)
    lim_ref, lim_self = (3.01, 2.01) if variant == 0 else (4.01, 3.51)
    report(f"flash_attn v{variant} split S={S} x{force} vs reference", out, ref, max_ulp=lim_ref, max_frac=0.55)
    report(f"flash_attn v{variant} split S={S} x{force} vs unsplit kernel", out, base, max_ulp=lim_self, max_frac=0.06)


def test_flash_attn_full_size(ops):
    """BASELINE cfg 2 geometry: 24 heads, S = 8192 image + 512 text tokens (816 items -> 768 whole + 48 x 5 split)."""
    H, S = 24, 8704
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn((H, S, 128), generator=g).to(BF) for _ in range(3))
    torch.set_num_threads(max(torch.get_num_threads(), 16))
    ref = F.scaled_dot_product_attention(q[None], k[None], v[None])[0].permute(1, 0, 2).reshape(S, H * 128)
    out = ops.flash_attn(q.cuda(), k.cuda(), ops.pack_vt(v.cuda(), S), S)
    out1 = ops.flash_attn(q.cuda(), k.cuda(), ops.pack_vt(v.cuda(), S), S, workspace=False)
    u = ulps(out, ref)
    print(f"[parity] flash_attn full size: mismatching {(u > 0).float().mean().item()*100:.2f}% max {u.max().item():.2f} ulp")
    assert u.max().item() <= 4.0
    report("flash_attn full size: balanced vs single-kernel", out, out1, max_ulp=2.01, max_frac=0.05)


def test_flash_attn_peaked(ops):
    """A spiked key forces large running-max jumps mid-sequence (online-softmax rescale path)."""
    H, S = 24, 512
    q, k, v = rnd((H, S, 128), 31), rnd((H, S, 128), 32), rnd((H, S, 128), 33)
    k[:, 300] = q[:, 7] * 4.0
    k[:, 470] = q[:, 200] * 6.0
    ref32 = F.scaled_dot_product_attention(q[None].float(), k[None].float(), v[None].float())[0]
    ref32 = ref32.permute(1, 0, 2).reshape(S, H * 128)
    ref = F.scaled_dot_product_attention(q[None], k[None], v[None])[0].permute(1, 0, 2).reshape(S, H * 128)
    out = ops.flash_attn(q.cuda(), k.cuda(), ops.pack_vt(v.cuda(), S), S)
    e_gpu = (out.float().cpu() - ref32).abs().max().item()
    e_cpu = (ref.float() - ref32).abs().max().item()
    print(f"[parity] flash_attn peaked: max err vs fp32 truth  hip {e_gpu:.3e}  cpu-bf16-sdpa {e_cpu:.3e}")
    assert e_gpu <= 2.0 * e_cpu + 1e-3


@pytest.mark.parametrize("S,force", [(64, 0), (100, 0), (257, 0), (700, 3), (1093, 5), (2048, 0), (4160, 3)])
def test_flash_attn_w4_bit_identical(ops, attn_variant, S, force):
    """Variant 3 (4 waves x 64 rows, one wave per SIMD, two 32-MFMA phases per KV tile pipelined across tiles) performs the
    arithmetic of the default kernel in the same order: outputs are BIT-IDENTICAL, ragged tails, split-KV partials and NaN
    bit patterns in the pad rows of K included."""
    from physicedit_amd._lib import lib
    H = 24
    g = torch.Generator().manual_seed(100 + S)
    q, k, v = (torch.randn((H, S, 128), generator=g).to(BF) for _ in range(3))
    sp = ops.s_pad_of(S)
    qd = torch.zeros((H, sp, 128), dtype=BF, device="cuda"); qd[:, :S] = q.cuda()
    kd = torch.full((H, sp, 128), float("nan"), dtype=BF, device="cuda"); kd[:, :S] = k.cuda()
    vt = ops.pack_vt(v.cuda(), sp)
    lib().pe_debug_set(b"attn_force_split", force)
    attn_variant(0)
    base = ops.flash_attn(qd, kd, vt, S).clone()
    attn_variant(3)
    out = ops.flash_attn(qd, kd, vt, S).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    assert torch.equal(out, base)
    for _ in range(20):                                       # race screen
        assert torch.equal(ops.flash_attn(qd, kd, vt, S), out)


def test_flash_attn_lazy_max_forced_rescale(ops, attn_variant):
    """Variant 4 raises a block's running max only when a row outgrows it by 2^8, so on bounded random data its rescale branch
    almost never runs after the first tile.  Force it (cdna guide T13 / rule 26): spiked keys make chosen rows jump by far more
    than 2^8 in the middle of the sequence, in tiles where other rows of the same 32-row block do not move; the result must
    agree with the textbook kernel (variant 0) to rounding and with the fp32 truth as well as the bf16 reference does."""
    H, S = 24, 1093
    q, k, v = rnd((H, S, 128), 31), rnd((H, S, 128), 32), rnd((H, S, 128), 33)
    for (key, row, gain) in ((300, 7, 8.0), (470, 200, 12.0), (700, 7, 16.0), (1000, 1090, 10.0)):
        k[:, key] = q[:, row] * gain          # score ~ gain * |q|^2 / sqrt(128) ~ 11 * gain  >> 8 / log2(e)
    ref = F.scaled_dot_product_attention(q[None], k[None], v[None])[0].permute(1, 0, 2).reshape(S, H * 128)
    ref32 = F.scaled_dot_product_attention(q[None].float(), k[None].float(), v[None].float())[0].permute(1, 0, 2).reshape(S, H * 128)
    sp = ops.s_pad_of(S)
    qd = torch.zeros((H, sp, 128), dtype=BF, device="cuda"); qd[:, :S] = q.cuda()
    kd = torch.full((H, sp, 128), float("nan"), dtype=BF, device="cuda"); kd[:, :S] = k.cuda()
    vt = ops.pack_vt(v.cuda(), sp)
    attn_variant(0)
    base = ops.flash_attn(qd, kd, vt, S).clone()
    attn_variant(4)
    out = ops.flash_attn(qd, kd, vt, S)
    assert torch.isfinite(out.float()).all()
    report("flash_attn v4 forced rescale vs v0", out, base, max_ulp=4.01, max_frac=0.55)
    e4 = (out.float().cpu() - ref32).abs().max().item()
    e0 = (base.float().cpu() - ref32).abs().max().item()
    ec = (ref.float() - ref32).abs().max().item()
    print(f"[parity] flash_attn forced rescale: max err vs fp32 truth  v4 {e4:.3e}  v0 {e0:.3e}  cpu-bf16-sdpa {ec:.3e}")
    assert e4 <= 2.0 * ec + 1e-3
    r4 = (out.float().cpu() - ref32).pow(2).mean().sqrt().item()
    rc = (ref.float() - ref32).pow(2).mean().sqrt().item()
    assert r4 <= 1.25 * rc + 1e-6      # peaked rows: the textbook update keeps the dominant P = 1.0 exact, the lazy one does not


# --- the folded form (variants 5 / 6): Q arrives multiplied by scale . log2(e), the running max goes through the MFMA C operand ---
@functools.lru_cache(maxsize=16)     # variants 5 / 7 of a case share its CPU references (the fp32 SDPA at S = 8704 is seconds of host time)
def _fold_case(S, seed, H=24, spikes=()):
    """fp32 q0 (what the QKV epilogue holds before its one rounding), its two roundings -- bf16(q0) for the reference form and
    bf16(q0 . c) for the folded kernel -- and the references: torch-CPU bf16 SDPA on bf16(q0), fp32 SDPA on q0."""
    g = torch.Generator().manual_seed(seed)
    q0 = torch.randn((H, S, 128), generator=g)
    k = torch.randn((H, S, 128), generator=g).to(BF)
    v = torch.randn((H, S, 128), generator=g).to(BF)
    for (key, row, gain) in spikes:
        k[:, key] = (q0[:, row] * gain).to(BF)
    c = (1.0 / math.sqrt(128.0)) * 1.4426950408889634
    torch.set_num_threads(max(torch.get_num_threads(), 16))
    ref = F.scaled_dot_product_attention(q0.to(BF)[None], k[None], v[None])[0].permute(1, 0, 2).reshape(S, H * 128)
    ref32 = F.scaled_dot_product_attention(q0[None], k[None].float(), v[None].float())[0].permute(1, 0, 2).reshape(S, H * 128)
    return q0.to(BF), (q0 * c).to(BF), k, v, ref, ref32


def _dev_qkv(ops, q, k, v, S, nan_pad=True):
    H = q.shape[0]
    sp = ops.s_pad_of(S)
    qd = torch.zeros((H, sp, 128), dtype=BF, device="cuda"); qd[:, :S] = q.cuda()
    kd = torch.full((H, sp, 128), float("nan") if nan_pad else 0.0, dtype=BF, device="cuda"); kd[:, :S] = k.cuda()   # pad rows are masked
    return qd, kd, ops.pack_vt(v.cuda(), sp)


def _rms(a, b):
    return (a.float().cpu() - b.float().cpu()).pow(2).mean().sqrt().item()


@pytest.mark.parametrize("S,variant", [(64, 5), (100, 5), (257, 5), (700, 5), (1093, 5), (2208, 5), (100, 6), (1093, 6),
                                       (64, 7), (100, 7), (257, 7), (700, 7), (1093, 7), (2208, 7)])
def test_flash_attn_fold(ops, attn_variant, S, variant):
    """Same criterion as test_flash_attn: the rms distance to the fp32 result must be the bf16 reference's own.  The folded kernel
    sees bf16(q . c) where the reference sees bf16(q) -- one rounding each of the same fp32 q, at different bits -- so element-wise
    it is compared with the fp32 truth at the reference's own worst error, not bit-wise with the reference.  Variant 6 (raise on every
    new maximum) runs the raise path -- score fix-up, negm rewrite, O rescale -- on nearly every tile."""
    qb, qc, k, v, ref, ref32 = _fold_case(S, 500 + S)
    qd, kd, vt = _dev_qkv(ops, qc, k, v, S)
    attn_variant(variant)
    assert ops.attn_q_prescale() > 0.12
    out = ops.flash_attn(qd, kd, vt, S, q_prescaled=True)
    assert torch.isfinite(out.float()).all()
    e_gpu, e_cpu = _rms(out, ref32), _rms(ref, ref32)
    m_gpu = (out.float().cpu() - ref32).abs().max().item()
    m_cpu = (ref.float() - ref32).abs().max().item()
    print(f"[parity] flash_attn v{variant} S={S}: rms err vs fp32 truth  hip {e_gpu:.3e}  cpu-bf16-sdpa {e_cpu:.3e};  max {m_gpu:.3e} / {m_cpu:.3e}")
    assert e_gpu <= 1.1 * e_cpu + 1e-6
    assert m_gpu <= 2.0 * m_cpu + 1e-3          # lazy max: P <= 2^8 is rounded to bf16 at another scale (cdna guide T13: ~3x max-abs)
    for _ in range(10):                                       # race screen
        assert torch.equal(ops.flash_attn(qd, kd, vt, S, q_prescaled=True), out)


@pytest.mark.parametrize("S,force,variant", [(700, 3, 5), (1093, 5, 5), (300, 8, 5), (1093, 5, 6), (700, 3, 7), (1093, 5, 7), (300, 8, 7)])
def test_flash_attn_fold_split_kv(ops, attn_variant, S, force, variant):
    """split-KV partials of the folded kernel: every part starts from its own first tile (m = that tile's row max) and ends on up to
    three fully masked dummy tiles; the merged result must agree with the unsplit kernel to rounding."""
    from physicedit_amd._lib import lib
    qb, qc, k, v, ref, ref32 = _fold_case(S, 900 + S)
    qd, kd, vt = _dev_qkv(ops, qc, k, v, S)
    attn_variant(variant)
    base = ops.flash_attn(qd, kd, vt, S, workspace=False, q_prescaled=True)
    try:
        lib().pe_debug_set(b"attn_force_split", force)
        out = ops.flash_attn(qd, kd, vt, S, workspace=True, q_prescaled=True)
    finally:
        lib().pe_debug_set(b"attn_force_split", 0)
    assert torch.isfinite(out.float()).all()
    report(f"flash_attn v{variant} split S={S} x{force} vs unsplit kernel", out, base, max_ulp=3.51, max_frac=0.06)
    assert _rms(out, ref32) <= 1.1 * _rms(ref, ref32) + 1e-6


def test_flash_attn_fold_forced_raise(ops, attn_variant):
    """The lazy raise of variant 5 on data that forces it mid-sequence (cdna guide T13 / rule 26): spiked keys make chosen rows jump
    by far more than 2^8 in tiles where the other rows of the 32-row block do not move, and one spike is NEGATIVE for its row's
    first tile (m must follow a first-tile maximum of either sign).  Against the fp32 truth as well as the bf16 reference does, and
    variant 5 == variant 6 == the exact form (variant 4 on a plain Q) to rounding."""
    S = 1093
    spikes = ((300, 7, 8.0), (470, 200, 12.0), (700, 7, 16.0), (1000, 1090, 10.0), (5, 900, -6.0))
    qb, qc, k, v, ref, ref32 = _fold_case(S, 77, spikes=spikes)
    qd, kd, vt = _dev_qkv(ops, qc, k, v, S)
    qp, _, _ = _dev_qkv(ops, qb, k, v, S)
    outs = {}
    for variant in (5, 6, 7):
        attn_variant(variant)
        outs[variant] = ops.flash_attn(qd, kd, vt, S, q_prescaled=True).clone()
    attn_variant(4)
    outs[4] = ops.flash_attn(qp, kd, vt, S).clone()
    ec_max = (ref.float() - ref32).abs().max().item()
    ec_rms = _rms(ref, ref32)
    for variant, out in outs.items():
        assert torch.isfinite(out.float()).all()
        e_max = (out.float().cpu() - ref32).abs().max().item()
        print(f"[parity] flash_attn forced raise v{variant}: max err vs fp32 truth {e_max:.3e} (cpu-bf16-sdpa {ec_max:.3e}), rms {_rms(out, ref32):.3e} / {ec_rms:.3e}")
        assert e_max <= 2.0 * ec_max + 1e-3
        assert _rms(out, ref32) <= 1.25 * ec_rms + 1e-6
    assert _rms(outs[5], outs[6]) <= 1.5 * ec_rms
    assert _rms(outs[5], outs[7]) <= 1.5 * ec_rms
    assert _rms(outs[5], outs[4]) <= 1.5 * ec_rms


@pytest.mark.parametrize("variant", [5, 7])
def test_flash_attn_fold_full_size(ops, attn_variant, variant):
    """BASELINE cfg 2 geometry with the folded kernel (816 items -> 768 whole + 48 x 5 split)."""
    H, S = 24, 8704
    qb, qc, k, v, ref, ref32 = _fold_case(S, 6)
    attn_variant(variant)
    vt = ops.pack_vt(v.cuda(), S)
    out = ops.flash_attn(qc.cuda(), k.cuda(), vt, S, q_prescaled=True)
    out1 = ops.flash_attn(qc.cuda(), k.cuda(), vt, S, workspace=False, q_prescaled=True)
    e_gpu, e_cpu = _rms(out, ref32), _rms(ref, ref32)
    print(f"[parity] flash_attn v{variant} full size: rms err vs fp32 truth  hip {e_gpu:.3e}  cpu-bf16-sdpa {e_cpu:.3e}")
    assert e_gpu <= 1.1 * e_cpu + 1e-6
    report(f"flash_attn v{variant} full size: balanced vs single-kernel", out, out1, max_ulp=3.01, max_frac=0.05)


def test_flash_attn_plain_q_takes_exact_form(ops, attn_variant):
    """pe_flash_attn on a plain Q with the folded variants selected runs the same schedule's exact form (5 -> 4, 6 -> 3): bit-identical
    to selecting those directly; pe_flash_attn_prescaled refuses when the selected variant wants a plain Q."""
    from physicedit_amd._lib import lib
    S = 700
    qb, qc, k, v, ref, ref32 = _fold_case(S, 3)
    qp, kd, vt = _dev_qkv(ops, qb, k, v, S)
    for fold, exact in ((5, 4), (6, 3)):
        attn_variant(fold)
        a = ops.flash_attn(qp, kd, vt, S).clone()
        attn_variant(exact)
        assert torch.equal(a, ops.flash_attn(qp, kd, vt, S))
    assert ops.attn_q_prescale() == 1.0
    with pytest.raises(Exception):
        ops.flash_attn(qp, kd, vt, S, q_prescaled=True)


# --- e4m3 attention (qwen_image_flash_attention(enable_fp8_attention=True), qwen_image_dit.py:24-35) ---
def _fp8_attn(ops, q, k, v, S, workspace=True, want_stats=False, stale_pad=None):
    from physicedit_amd._lib import check, lib, stream_ptr
    H = q.shape[0]
    qd, kd, vt = _dev_qkv(ops, q, k, v, S, nan_pad=False)
    sp = qd.shape[1]
    if stale_pad is not None:       # what another call of a different length left in the positions of tokens >= S (finite values)
        pos = ops.vt_positions(sp, vt.device)[S:]
        vt[:, :, pos] = stale_pad[:, :, :pos.numel()].to(vt.device)
        qd[:, S:] = 3.0
        kd[:, S:] = -2.0
    n = lib().pe_flash_attn_fp8_scratch_bytes(H, sp)
    scratch = torch.empty((n + 256,), dtype=torch.uint8, device="cuda")
    base = (scratch.data_ptr() + 255) // 256 * 256
    out = torch.empty((S, H * 128), dtype=BF, device="cuda")
    ws, nb = None, 0
    if workspace:
        nb = lib().pe_flash_attn_workspace_bytes(H, S)
        ws = torch.empty((nb,), dtype=torch.uint8, device="cuda")
    check(lib().pe_flash_attn_fp8(qd.data_ptr(), kd.data_ptr(), vt.data_ptr(), out.data_ptr(), H, S, sp, H * 128, base, n,
                                  ws.data_ptr() if ws is not None else None, nb, stream_ptr()), "pe_flash_attn_fp8")
    if want_stats:      # the scratch layout: Q8 | K8 | Vt8 planes of H * S_pad * 128 bytes, then {q_std, k_std, v_std, scale_log2} as fp32
        off = base - scratch.data_ptr() + 3 * H * sp * 128
        return out, scratch[off:off + 16].view(torch.float32).cpu()
    return out


@pytest.fixture
def fp8_variant(request):
    """attn_fp8_variant knob for one test: 4 = variant 3 with the row sums taken by the matrix pipe from the e4m3 P (ones . P),
    3 = variant 1's arithmetic with one wave per SIMD (4 waves x 64 query rows), 2 = software-pipelined
    with the max-free fast path (reference kept while a row's tile sum <= 448), 1 = software-pipelined, reference raised by tiles 2^8 above
    it, 0 = the plain kernel (running max)"""
    from physicedit_amd._lib import lib, check
    check(lib().pe_debug_set(b"attn_fp8_variant", request.param), "attn_fp8_variant")
    yield request.param
    check(lib().pe_debug_set(b"attn_fp8_variant", FP8_ATTN_DEFAULT), "attn_fp8_variant")


FP8_ATTN_DEFAULT = 1


# the default kernel at every size; the plain kernel and the opt-in fast path where tails, splits and unequal scales meet (suite time)
@pytest.mark.parametrize("S,scales,fp8_variant", [(64, (1.0, 1.0, 1.0), 1), (100, (1.0, 1.0, 1.0), 1), (700, (0.7, 1.9, 3.1), 1),
                                                  (1093, (2.5, 0.4, 0.05), 1), (2208, (1.0, 1.3, 0.8), 1),
                                                  (100, (1.0, 1.0, 1.0), 0), (1093, (2.5, 0.4, 0.05), 0), (2208, (1.0, 1.3, 0.8), 0),
                                                  (100, (1.0, 1.0, 1.0), 2), (700, (0.7, 1.9, 3.1), 2), (1093, (2.5, 0.4, 0.05), 2),
                                                  (64, (1.0, 1.0, 1.0), 3), (100, (1.0, 1.0, 1.0), 3), (700, (0.7, 1.9, 3.1), 3),
                                                  (1093, (2.5, 0.4, 0.05), 3), (2208, (1.0, 1.3, 0.8), 3),
                                                  (64, (1.0, 1.0, 1.0), 4), (100, (1.0, 1.0, 1.0), 4), (700, (0.7, 1.9, 3.1), 4),
                                                  (1093, (2.5, 0.4, 0.05), 4), (2208, (1.0, 1.3, 0.8), 4)],
                         indirect=["fp8_variant"])
def test_flash_attn_fp8(ops, S, scales, fp8_variant):
    """The e4m3 attention operator against the oracle's restatement of the reference branch (global std of q, k, v in bf16, e4m3 casts,
    softmax_scale = q_std k_std / sqrt(128), P cast to e4m3, output x v_std), on tensors whose three scales differ: what the kernel
    may differ in is the fp32 summation order and the rounding of P near e4m3 ties -- a small fraction of what e4m3 operands cost
    against the fp32 truth.  Also: the three standard deviations reproduce torch.std bit for bit (as bf16)."""
    from physicedit_amd._lib import lib
    H = 24
    g = torch.Generator().manual_seed(800 + S)
    q, k, v = ((torch.randn((H, S, 128), generator=g) * sc).to(BF) for sc in scales)
    torch.set_num_threads(max(torch.get_num_threads(), 16))
    ref = O.flash_attention_fp8(q[None], k[None], v[None])[0].permute(1, 0, 2).reshape(S, H * 128)
    tau = 8.0 if fp8_variant in (1, 3, 4) else 0.0      # the pipelined kernels raise their reference lazily (oracle: lazy_tau_log2 / lazy_sum_limit)
    ref_t = O.flash_attention_fp8(q[None], k[None], v[None], kv_tile=64, lazy_tau_log2=tau,
                                  lazy_sum_limit=448.0 if fp8_variant == 2 else None,
                                  row_sum_quantised=fp8_variant == 4)[0].permute(1, 0, 2).reshape(S, H * 128)
    ref32 = F.scaled_dot_product_attention(q[None].float(), k[None].float(), v[None].float())[0].permute(1, 0, 2).reshape(S, H * 128)
    out, stats = _fp8_attn(ops, q, k, v, S, workspace=False, want_stats=True)
    assert torch.isfinite(out.float()).all()
    # the statistics: torch.std of the whole bf16 tensor (unbiased, a bf16 scalar), and the scale FA3 would receive, in the log2 domain
    q_std, k_std, v_std = q.std(), k.std(), v.std()
    want = [float(q_std), float(k_std), float(v_std), float(q_std * k_std / math.sqrt(128.0)) * 1.4426950408889634]
    print(f"[parity] flash_attn_fp8 S={S}: std / scale  hip {[round(x, 6) for x in stats.tolist()]}  torch {[round(x, 6) for x in want]}")
    assert stats[:3].tolist() == want[:3] and abs(stats[3].item() - want[3]) <= 1e-6 * abs(want[3])
    e_tile, e_glob, e_fp8 = _rms(out, ref_t), _rms(out, ref), _rms(ref, ref32)
    print(f"[parity] flash_attn_fp8 S={S} variant {fp8_variant}: rms hip vs oracle-fp8 (online, 64-key tiles, tau 2^{tau:.0f}) {e_tile:.3e}, (P against the final max) {e_glob:.3e}; "
          f"oracle-fp8 vs fp32 truth {e_fp8:.3e}; hip vs truth {_rms(out, ref32):.3e}")
    # against the restatement that quantises P where a flash kernel must (running max of 64-key tiles): summation order and e4m3 ties
    assert e_tile <= 0.05 * e_fp8 + 1e-6
    # against the form that knows the final max: the same size of P rounding noise at other rounding points
    # (the lazy form moves its rounding points further from the final-max form's than the running max does: 0.66 measured, 0.44 for variant 0)
    noise = 0.8 if fp8_variant >= 1 else 0.6
    assert e_glob <= noise * e_fp8 + 1e-6
    assert _rms(out, ref32) <= 1.1 * e_fp8 + 1e-6
    out1 = out
    out = _fp8_attn(ops, q, k, v, S)
    # the load-balanced form (leftover items split along KV): every part has its own running max, so its P is rounded to e4m3 at other
    # points than the unsplit kernel's -- the same noise again, not a bf16-ulp matter
    print(f"[parity] flash_attn_fp8 S={S}: balanced vs single-kernel rms {_rms(out, out1):.3e}; balanced vs truth {_rms(out, ref32):.3e}")
    assert _rms(out, out1) <= noise * e_fp8 + 1e-6 and _rms(out, ref32) <= 1.1 * e_fp8 + 1e-6
    for _ in range(5):
        assert torch.equal(_fp8_attn(ops, q, k, v, S), out)


@pytest.mark.parametrize("S", [65, 100, 1093, 2192])
def test_flash_attn_fp8_ignores_stale_pad_columns(ops, S):
    """One pe_dit handle serves both prompt lengths of a CFG pair, so the planes' positions [S, S_pad) hold whatever the other branch
    wrote there: the three standard deviations (v_std sums Vt, whose token order is permuted inside 16-groups) and the output must
    not see them.  Bit for bit against zeroed pads."""
    H = 3
    g = torch.Generator().manual_seed(4100 + S)
    q, k, v = ((torch.randn((H, S, 128), generator=g) * sc).to(BF) for sc in (1.1, 0.8, 1.7))
    stale = (torch.randn((H, 128, 64), generator=g) * 9.0).to(BF)
    out0, st0 = _fp8_attn(ops, q, k, v, S, want_stats=True)
    out1, st1 = _fp8_attn(ops, q, k, v, S, want_stats=True, stale_pad=stale)
    assert torch.equal(st0, st1), (st0, st1)
    assert torch.equal(out0, out1)
    want = torch.stack([t.float().std() for t in (q, k, v)]).to(BF).float()
    assert torch.equal(st1[:3], want)


@pytest.mark.parametrize("H,S", [(256, 64), (256, 128), (256, 192), (256, 256), (256, 320), (256, 384), (4, 700), (3, 1093)])
def test_flash_attn_fp8_every_peeled_path(ops, H, S):
    """The pipelined e4m3 kernel peels its first and last iteration and alternates two S / P buffers: 1 ... 6 tiles per work-group
    without a KV split (256 items = one round) and split parts of odd and even lengths take every instantiation of the iteration body
    (an even tile count once returned stale accumulator registers).  Against the plain kernel: the same arithmetic up to the rounding
    points of P, i.e. far inside what e4m3 operands cost; against the lazy restatement of the oracle for the small ones."""
    from physicedit_amd._lib import lib, check
    g = torch.Generator().manual_seed(900 + S)
    q, k, v = ((torch.randn((H, S, 128), generator=g) * sc).to(BF) for sc in (1.0, 1.2, 0.9))
    outs = []
    try:
        for variant in (0, 1, 2, 3, 4):
            check(lib().pe_debug_set(b"attn_fp8_variant", variant), "attn_fp8_variant")
            outs.append(_fp8_attn(ops, q, k, v, S))
    finally:
        check(lib().pe_debug_set(b"attn_fp8_variant", FP8_ATTN_DEFAULT), "attn_fp8_variant")
    ref32 = F.scaled_dot_product_attention(q[None].float(), k[None].float(), v[None].float())[0].permute(1, 0, 2).reshape(S, H * 128)
    e0 = _rms(outs[0], ref32)
    for variant in (1, 2, 4):
        e1, d = _rms(outs[variant], ref32), _rms(outs[variant], outs[0])
        print(f"[parity] flash_attn_fp8 peeled paths H={H} S={S} variant {variant}: vs fp32 truth plain {e0:.3e} pipelined {e1:.3e}; pipelined vs plain {d:.3e}")
        assert torch.isfinite(outs[variant].float()).all()
        assert e1 <= 1.1 * e0 + 1e-6 and d <= 0.9 * e0 + 1e-6
    # variant 3 is variant 1 with one wave per SIMD: per row the same operations in the same order
    assert torch.equal(outs[3], outs[1]), f"one-wave layout differs from variant 1: max |d| {(outs[3].float() - outs[1].float()).abs().max().item():.3e}"
    if H == 256 and S <= 192:       # no KV split (every part of a split item starts its own reference sequence), and seconds of CPU
        for variant, kw in ((1, dict(lazy_tau_log2=8.0)), (2, dict(lazy_sum_limit=448.0)), (4, dict(lazy_tau_log2=8.0, row_sum_quantised=True))):
            ref_t = O.flash_attention_fp8(q[None], k[None], v[None], kv_tile=64, **kw)[0].permute(1, 0, 2).reshape(S, H * 128)
            assert _rms(outs[variant], ref_t) <= 0.05 * e0 + 1e-6


@pytest.mark.parametrize("S,scales", [(300, (3.0, 3.0, 1.0)), (1093, (3.0, 3.0, 1.0)), (2208, (2.0, 4.0, 0.5)), (2208, (1.0, 1.0, 1.0))])
def test_flash_attn_fp8_one_wave_layout_is_variant_1(ops, S, scales):
    """attn_fp8_variant 3 (4 waves x 64 query rows, O and Q in fixed accumulator registers) performs variant 1's operations per row in
    variant 1's order: bit-identical outputs, with and without the KV-split plan, also on logits whose tile maxima keep moving (q_std k_std
    = 8 ... 9: the pass over O, which this kernel does in asm on the accumulator registers, runs on most tiles)."""
    from physicedit_amd._lib import lib, check
    H = 24
    g = torch.Generator().manual_seed(5100 + S)
    q, k, v = ((torch.randn((H, S, 128), generator=g) * sc).to(BF) for sc in scales)
    outs = {}
    try:
        for variant in (1, 3):
            check(lib().pe_debug_set(b"attn_fp8_variant", variant), "attn_fp8_variant")
            outs[variant] = (_fp8_attn(ops, q, k, v, S, workspace=False), _fp8_attn(ops, q, k, v, S))
    finally:
        check(lib().pe_debug_set(b"attn_fp8_variant", FP8_ATTN_DEFAULT), "attn_fp8_variant")
    ref32 = F.scaled_dot_product_attention(q[None].float(), k[None].float(), v[None].float())[0].permute(1, 0, 2).reshape(S, H * 128)
    print(f"[parity] flash_attn_fp8 one-wave layout S={S} scales {scales}: rms vs fp32 truth {_rms(outs[3][0], ref32):.3e} (variant 1: {_rms(outs[1][0], ref32):.3e})")
    for a, b in zip(outs[1], outs[3]):
        assert torch.isfinite(b.float()).all()
        assert torch.equal(a, b), f"max |d| {(a.float() - b.float()).abs().max().item():.3e}"


# ------------------------------------------------------------------------------------------------
# row kernels
# ------------------------------------------------------------------------------------------------
def test_ln_modulate(ops):
    rows, D = 333, 3072
    x = rnd((rows, D), 41, 2.0) + 0.3
    sa, ca, sb, cb = rnd((D,), 42, 0.3), rnd((D,), 43, 0.3), rnd((D,), 44, 0.3), rnd((D,), 45, 0.3)
    ln = F.layer_norm(x, (D,), eps=1e-6)
    ref = torch.cat([ln[:200] * (1 + ca) + sa, ln[200:] * (1 + cb) + sb])
    out = ops.ln_modulate(x.cuda(), sa.cuda(), ca.cuda(), 200, sb.cuda(), cb.cuda())
    report("ln_modulate", out, ref, 2.01, 0.02)


def test_rmsnorm(ops):
    x, w = rnd((77, 3584), 51, 1.5), synth.make_tensor(5, "txt_norm.weight", (3584,))
    report("rmsnorm3584", ops.rmsnorm(x.cuda(), w.cuda()), O.rmsnorm(x, w), 1.01, 0.01)


def test_patchify_roundtrip(ops):
    lat = rnd((1, 16, 24, 40), 61)
    tok = ops.patchify(lat.cuda())
    assert torch.equal(tok.cpu(), O.patchify(lat)[0])
    back = ops.unpatchify(tok, 16, 24, 40)
    assert torch.equal(back.cpu(), lat)


@pytest.mark.parametrize("cfg", [1.0, 4.0])
def test_cfg_euler(ops, cfg):
    p, n, x = rnd((1, 16, 32, 32), 71), rnd((1, 16, 32, 32), 72), rnd((1, 16, 32, 32), 73)
    tab = O.FlowMatchTables(4, dynamic_shift_len=64)
    from physicedit_amd.scheduler import qwen_image_scheduler
    sch = qwen_image_scheduler()
    sch.set_timesteps(4, dynamic_shift_len=64)
    for i in range(4):
        pred = n + cfg * (p - n) if cfg != 1.0 else p
        ref = tab.step(pred, i, x)
        out = ops.cfg_euler_step(p.cuda(), n.cuda(), x.cuda(), cfg, sch.dsigma(i))
        assert torch.equal(out.cpu(), ref), (cfg, i)      # pure element-wise: bit exact


@pytest.mark.parametrize("S_img,segs,force", [(128, (12, 20, 8, 40), 0), (704, (70, 130, 9, 200), 3), (1024, (64, 333), 0)])
def test_flash_attn_token_words(ops, S_img, segs, force):
    """pe_flash_attn_masked: the EliGen mask as one word per token vs torch SDPA with the explicit [S, S] additive mask
    (scaled_dot_product_attention(attn_mask=), qwen_image_dit.py:37).  Regions are random; one prompt has an EMPTY region (its
    rows see no key in any image tile: the -inf guard), and the split-KV path runs with partials that contain no allowed key."""
    from physicedit_amd._lib import lib
    H = 24
    T = sum(segs)
    S = S_img + T
    g = torch.Generator().manual_seed(S)
    q, k, v = (torch.randn((H, S, 128), generator=g).to(BF) for _ in range(3))
    n = len(segs)
    member = torch.rand((n, S_img), generator=g) < 0.3
    member[1] = False                                   # empty region
    member[n - 1] = True                                # the global prompt
    words = torch.zeros((ops.s_pad_of(S),), dtype=torch.int64)
    for i in range(n):
        words[:S_img] |= member[i].to(torch.int64) << i
    words[:S_img] |= 1 << 31
    words[S_img:S] = torch.cat([torch.full((m,), 1 << i, dtype=torch.int64) for i, m in enumerate(segs)])
    allowed = (words[:S, None] & words[None, :S]) != 0
    bias = torch.zeros((S, S)).masked_fill(~allowed, float("-inf")).to(BF)
    ref = F.scaled_dot_product_attention(q[None], k[None], v[None], attn_mask=bias[None, None])[0]
    ref = ref.permute(1, 0, 2).reshape(S, H * 128)
    ref32 = F.scaled_dot_product_attention(q[None].float(), k[None].float(), v[None].float(), attn_mask=bias[None, None].float())[0]
    ref32 = ref32.permute(1, 0, 2).reshape(S, H * 128)
    sp = ops.s_pad_of(S)
    qd = torch.zeros((H, sp, 128), dtype=BF, device="cuda"); qd[:, :S] = q.cuda()
    kd = torch.full((H, sp, 128), float("nan"), dtype=BF, device="cuda"); kd[:, :S] = k.cuda()
    wd = (words & 0xFFFFFFFF).to(torch.uint32).view(torch.int32).cuda()
    try:
        lib().pe_debug_set(b"attn_force_split", force)
        out = ops.flash_attn(qd, kd, ops.pack_vt(v.cuda(), sp), S, token_words=wd, n_img=S_img)
    finally:
        lib().pe_debug_set(b"attn_force_split", 0)
    assert torch.isfinite(out.float()).all()
    report(f"flash_attn token words S_img={S_img} T={T}", out, ref, max_ulp=3.01, max_frac=0.50)
    e_gpu = (out.float().cpu() - ref32).pow(2).mean().sqrt().item()
    e_cpu = (ref.float() - ref32).pow(2).mean().sqrt().item()
    print(f"[parity] flash_attn token words: rms err vs fp32 truth  hip {e_gpu:.3e}  cpu-bf16-sdpa {e_cpu:.3e}")
    assert e_gpu <= 1.5 * e_cpu + 1e-6
    # the mask matters: the unmasked kernel gives something else
    plain = ops.flash_attn(qd, kd, ops.pack_vt(v.cuda(), sp), S)
    assert (plain.float().cpu() - ref32).pow(2).mean().sqrt().item() > 5 * e_cpu


def test_flash_attn_token_words_full_size(ops):
    """The EliGen mask at the BASELINE cfg 2 geometry (8192 image + 512 text tokens: three entity prompts + the global one; 146 KV
    tiles of which 8 hold text keys, 2 of the 34 query blocks hold text rows; split-KV balancing on)."""
    H, S_img, segs = 6, 8192, (40, 60, 52, 360)
    S = S_img + sum(segs)
    g = torch.Generator().manual_seed(77)
    q, k, v = (torch.randn((H, S, 128), generator=g).to(BF) for _ in range(3))
    n = len(segs)
    member = torch.rand((n, S_img), generator=g) < 0.25
    member[n - 1] = True
    words = torch.zeros((ops.s_pad_of(S),), dtype=torch.int64)
    for i in range(n):
        words[:S_img] |= member[i].to(torch.int64) << i
    words[:S_img] |= 1 << 31
    words[S_img:S] = torch.cat([torch.full((m,), 1 << i, dtype=torch.int64) for i, m in enumerate(segs)])
    allowed = (words[:S, None] & words[None, :S]) != 0
    bias = torch.zeros((S, S), dtype=BF).masked_fill(~allowed, float("-inf"))
    torch.set_num_threads(max(torch.get_num_threads(), 16))
    ref = F.scaled_dot_product_attention(q[None], k[None], v[None], attn_mask=bias[None, None])[0].permute(1, 0, 2).reshape(S, H * 128)
    wd = (words & 0xFFFFFFFF).to(torch.uint32).view(torch.int32).cuda()
    out = ops.flash_attn(q.cuda(), k.cuda(), ops.pack_vt(v.cuda(), S), S, token_words=wd, n_img=S_img)
    u = ulps(out, ref)
    print(f"[parity] flash_attn token words, full size: mismatching {(u > 0).float().mean().item()*100:.2f}% max {u.max().item():.2f} ulp")
    assert torch.isfinite(out.float()).all() and u.max().item() <= 4.0
    # text rows (the ones the mask constrains most) separately
    ut = u[S_img:]
    assert ut.max().item() <= 4.0 and (ut > 0).float().mean().item() < 0.6


@pytest.mark.parametrize("N,K,with_bias", [(3584, 3584, True), (512, 3584, True), (18944, 3584, False), (3584, 18944, False), (1001, 64, True)])
def test_gemv(ops, N, K, with_bias):
    """pe_gemv_bf16 = nn.Linear on one row (the decode step of the prompt prologue's text encoder): vs an fp32-accumulated product."""
    x, w = rnd((K,), 71), rnd((N, K), 72, K ** -0.5)
    b = rnd((N,), 73) if with_bias else None
    ref = (w.float() @ x.float() + (b.float() if with_bias else 0)).to(BF)
    out = ops.gemv(x.cuda(), w.cuda(), b.cuda() if with_bias else None)
    report(f"gemv {N}x{K}", out, ref, 1.01, 0.02)


def test_gemv_residual(ops):
    """pe_gemv_res_bf16 = residual + nn.Linear(x) with the Linear's output rounded to bf16 before the add, as eager PyTorch does."""
    N, K = 3584, 18944
    x, w, r = rnd((K,), 74), rnd((N, K), 75, K ** -0.5), rnd((N,), 76)
    ref = r + (w.float() @ x.float()).to(BF)
    out = ops.gemv(x.cuda(), w.cuda(), None, res=r.cuda())
    report("gemv + residual", out, ref, 1.01, 0.02)


def test_gemv_swiglu(ops):
    """pe_gemv_swiglu_bf16 vs act_fn(gate_proj(x)) * up_proj(x) with torch's bf16 roundings (fp32-accumulated products)."""
    N, K = 18944, 3584
    x, wg, wu = rnd((K,), 81), rnd((N, K), 82, K ** -0.5), rnd((N, K), 83, K ** -0.5)
    g = (wg.float() @ x.float()).to(BF)
    u = (wu.float() @ x.float()).to(BF)
    ref = F.silu(g) * u
    out = ops.gemv_swiglu(x.cuda(), wg.cuda(), wu.cuda())
    report("gemv_swiglu", out, ref, 2.01, 0.03)


def test_decode_qkv_rope(ops):
    """pe_decode_qkv_rope vs the same steps in torch (transformers Qwen2_5_VLAttention.forward at q_len = 1: three Linears, then
    q * cos + rotate_half(q) * sin in bf16)."""
    K, hq, hkv = 3584, 28, 4
    x = rnd((K,), 91)
    wq, wk, wv = rnd((hq * 128, K), 92, K ** -0.5), rnd((hkv * 128, K), 93, K ** -0.5), rnd((hkv * 128, K), 94, K ** -0.5)
    bq, bk, bv = rnd((hq * 128,), 95), rnd((hkv * 128,), 96), rnd((hkv * 128,), 97)
    ang = torch.rand((128,), generator=torch.Generator().manual_seed(98)) * 6.28
    cs, sn = ang.cos().to(BF), ang.sin().to(BF)

    def lin(w, b):
        return (w.float() @ x.float() + b.float()).to(BF)

    def rope(t):
        t = t.view(-1, 128)
        rot = torch.cat([-t[:, 64:], t[:, :64]], dim=-1)
        return (t * cs) + (rot * sn)
    q, k, v = ops.decode_qkv_rope(x.cuda(), wq.cuda(), bq.cuda(), wk.cuda(), bk.cuda(), wv.cuda(), bv.cuda(), cs.cuda(), sn.cuda())
    report("decode v", v.reshape(-1), lin(wv, bv), 1.01, 0.02)
    report("decode q + rope", q, rope(lin(wq, bq)), 2.01, 0.03)
    report("decode k + rope", k, rope(lin(wk, bk)), 2.01, 0.03)


@pytest.mark.parametrize("L", [1, 77, 1348, 2400])
def test_decode_attention(ops, L):
    """pe_decode_attention: one query per head against a GQA cache vs torch SDPA on the repeated cache."""
    hq, hkv = 28, 4
    q, kc, vc = rnd((hq, 128), 101), rnd((hkv, L, 128), 102), rnd((hkv, L, 128), 103)
    kr, vr = kc.repeat_interleave(hq // hkv, dim=0), vc.repeat_interleave(hq // hkv, dim=0)
    ref32 = F.scaled_dot_product_attention(q[:, None].float(), kr.float(), vr.float()).reshape(-1)
    ref = F.scaled_dot_product_attention(q[:, None], kr, vr).reshape(-1)
    out = ops.decode_attention(q.cuda(), kc.cuda(), vc.cuda(), 128 ** -0.5)
    e_gpu = (out.float().cpu() - ref32).pow(2).mean().sqrt().item()
    e_cpu = (ref.float() - ref32).pow(2).mean().sqrt().item()
    print(f"[parity] decode attention L={L}: rms err vs fp32 truth  hip {e_gpu:.3e}  cpu-bf16-sdpa {e_cpu:.3e}")
    assert torch.isfinite(out.float()).all() and e_gpu <= 1.5 * e_cpu + 1e-4


def test_attn_mix_probe_runs(ops):
    """pe_attn_mix_probe (the default attention schedule without its softmax) launches, reports the launch's nominal FLOPs and leaves
    the library usable; its output is not a result."""
    import ctypes
    from physicedit_amd._lib import check, lib, stream_ptr
    H, S = 24, 2048
    q, k = rnd((H, S, 128), 1).cuda(), rnd((H, S, 128), 2).cuda()
    vt = rnd((H, 128, S), 3).cuda()
    out = torch.empty((S, H * 128), dtype=BF, device="cuda")
    fl = ctypes.c_double(0.0)
    check(lib().pe_attn_mix_probe(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), H, S, S, H * 128, ctypes.byref(fl), stream_ptr()),
          "pe_attn_mix_probe")
    torch.cuda.synchronize()
    assert fl.value == 4.0 * S * S * 128 * H


def test_mfma_probe(ops):
    """pe_mfma_probe (measurement aid of bench.py's roofline block): launches, reports its FLOP count, zero fragments give zeros."""
    import ctypes
    from physicedit_amd._lib import check, lib, stream_ptr
    frags = torch.zeros(8 << 20, dtype=BF, device="cuda")
    out = torch.full((4 * 512,), 7.0, dtype=torch.float32, device="cuda")
    fl = ctypes.c_double(0.0)
    check(lib().pe_mfma_probe(frags.data_ptr(), out.data_ptr(), 4, 10, ctypes.byref(fl), stream_ptr()), "pe_mfma_probe")
    torch.cuda.synchronize()
    assert fl.value == 4 * 8 * 10 * 32 * 2.0 * 32 * 32 * 16 and (out == 0).all()


def test_decode_step_kernels(ops):
    """the graph-capturable decode-step launches (device-side step counter) against their host-parameter forms: q/k/v + rotary with
    step-indexed tables and in-cache k / v rows, attention over base + step + 1 rows of a larger cache, embedding row, arg-max."""
    K, hq, hkv, cap, base = 3584, 28, 4, 96, 40
    x = rnd((K,), 111).cuda()
    wq, wk, wv = (rnd((n, K), s, K ** -0.5).cuda() for n, s in ((hq * 128, 112), (hkv * 128, 113), (hkv * 128, 114)))
    bq, bk, bv = (rnd((n,), s).cuda() for n, s in ((hq * 128, 115), (hkv * 128, 116), (hkv * 128, 117)))
    ang = torch.rand((8, 128), generator=torch.Generator().manual_seed(118)) * 6.28
    cs, sn = ang.cos().to(BF).cuda(), ang.sin().to(BF).cuda()
    kc, vc = rnd((hkv, cap, 128), 119).cuda(), rnd((hkv, cap, 128), 120).cuda()
    kc0, vc0 = kc.clone(), vc.clone()
    step = torch.tensor([5], dtype=torch.int32, device="cuda")
    q = ops.decode_step_qkv(x, wq, bq, wk, bk, wv, bv, cs, sn, kc, vc, step, base)
    q_ref, k_ref, v_ref = ops.decode_qkv_rope(x, wq, bq, wk, bk, wv, bv, cs[5].contiguous(), sn[5].contiguous())
    assert torch.equal(q, q_ref.reshape(-1))
    assert torch.equal(kc[:, base + 5], k_ref.view(hkv, 128)) and torch.equal(vc[:, base + 5], v_ref.view(hkv, 128))
    keep = torch.ones(cap, dtype=torch.bool); keep[base + 5] = False
    assert torch.equal(kc[:, keep], kc0[:, keep]) and torch.equal(vc[:, keep], vc0[:, keep])
    out = ops.decode_step_attention(q, kc, vc, step, base, 128 ** -0.5)
    L = base + 6
    ref = ops.decode_attention(q.view(hq, 128), kc[:, :L].contiguous(), vc[:, :L].contiguous(), 128 ** -0.5)
    assert torch.equal(out, ref.reshape(-1))
    # a step past the capacity writes nothing
    kc1, vc1 = kc.clone(), vc.clone()
    ops.decode_step_qkv(x, wq, bq, wk, bk, wv, bv, cs, sn, kc, vc, torch.tensor([7], dtype=torch.int32, device="cuda"), cap - 7)
    assert torch.equal(kc, kc1) and torch.equal(vc, vc1)
    table = rnd((50, 256), 121).cuda()
    tok = torch.tensor([17], dtype=torch.int32, device="cuda")
    assert torch.equal(ops.decode_embed(table, tok), table[17])
    logits = rnd((5000,), 122).cuda()
    logits[4321] = logits.max() + 1
    logits[77] = logits[4321]                                      # a tie: the first index wins, as torch.argmax
    out_ids = torch.zeros(8, dtype=torch.int32, device="cuda")
    st = torch.tensor([3], dtype=torch.int32, device="cuda")
    ops.decode_argmax(logits, tok, out_ids, st)
    torch.cuda.synchronize()
    assert tok.item() == 77 == int(logits.float().argmax()) and out_ids[3].item() == 77 and st.item() == 4


@pytest.mark.parametrize("cap,base,stepv", [(96, 40, 5), (2048, 1390, 9), (3072, 2047, 1000), (1024, 0, 0), (15360, 9000, 3)])
def test_decode_attention_split_bit_identical(ops, cap, base, stepv):
    """pe_decode_step_attention_split (scores / softmax sums + P.V per wave / combine over 16 x as many work-groups) keeps every sum's
    grouping: bit-identical to the one-launch form on caches of 1 ... 9004 valid rows, stale values in its workspace included (the
    captured decode step reuses one workspace for 28 layers and every token)."""
    hq, hkv = 28, 4
    q = rnd((hq * 128,), 201 + cap).cuda()
    kc, vc = rnd((hkv, cap, 128), 202 + cap).cuda(), rnd((hkv, cap, 128), 203 + cap, 2.0).cuda()
    step = torch.tensor([stepv], dtype=torch.int32, device="cuda")
    ws = ops.decode_attention_workspace(hq, cap, "cuda")
    ws.fill_(float("nan"))
    ref = ops.decode_step_attention(q, kc, vc, step, base, 128 ** -0.5)
    for _ in range(3):
        out = ops.decode_step_attention(q, kc, vc, step, base, 128 ** -0.5, workspace=ws)
        assert torch.equal(out, ref)
    assert torch.isfinite(out.float()).all()


def test_fused_rmsnorm_single_row_forms(ops):
    """pe_gemv_norm_bf16 / pe_gemv_swiglu_norm_bf16 / pe_decode_step_qkv(norm_w=): the RMSNorm fused into the staging of x must be
    BIT-identical to pe_rmsnorm followed by the plain launch (the captured decode step relies on it to reproduce generate())."""
    K = 3584
    x = rnd((K,), 131).cuda()
    nw = (1.0 + 0.1 * rnd((K,), 132).float()).to(BF).cuda()
    xn = ops.rmsnorm(x.view(1, K), nw, 1e-6).view(-1)
    w = rnd((1024, K), 133, K ** -0.5).cuda()
    b = rnd((1024,), 134).cuda()
    assert torch.equal(ops.gemv_norm(x, nw, 1e-6, w, b), ops.gemv(xn, w, b))
    wg, wu = rnd((2048, K), 135, K ** -0.5).cuda(), rnd((2048, K), 136, K ** -0.5).cuda()
    assert torch.equal(ops.gemv_swiglu_norm(x, nw, 1e-6, wg, wu), ops.gemv_swiglu(xn, wg, wu))
    hq, hkv, cap = 28, 4, 16
    wq, wk, wv = (rnd((n, K), s, K ** -0.5).cuda() for n, s in ((hq * 128, 137), (hkv * 128, 138), (hkv * 128, 139)))
    ang = torch.rand((4, 128), generator=torch.Generator().manual_seed(140)) * 6.28
    cs, sn = ang.cos().to(BF).cuda(), ang.sin().to(BF).cuda()
    step = torch.tensor([2], dtype=torch.int32, device="cuda")
    kc1, vc1 = torch.zeros((hkv, cap, 128), dtype=BF, device="cuda"), torch.zeros((hkv, cap, 128), dtype=BF, device="cuda")
    kc2, vc2 = torch.zeros_like(kc1), torch.zeros_like(vc1)
    q1 = ops.decode_step_qkv(x, wq, None, wk, None, wv, None, cs, sn, kc1, vc1, step, 3, norm_w=nw, eps=1e-6)
    q2 = ops.decode_step_qkv(xn, wq, None, wk, None, wv, None, cs, sn, kc2, vc2, step, 3)
    assert torch.equal(q1, q2) and torch.equal(kc1, kc2) and torch.equal(vc1, vc2) and kc1[:, 5].abs().sum() > 0


@pytest.mark.parametrize("cap,base,stepv,ff", [(1024, 37, 0, 18944), (2048, 1390, 9, 18944), (96, 40, 5, 2048)])
def test_decode_layer_bit_identical(ops, cap, base, stepv, ff):
    """pe_decode_layer (round 6: ONE launch per decoder layer of the decode step -- a persistent grid walks the work items of q / k / v,
    the split attention, o_proj, gate / up and down_proj with grid-wide barriers in between) against the eight launches it replaces: the
    layer output, the appended cache rows and nothing else in the caches, bit for bit; three layers chained on one scratch and a
    second token on the same scratch (the barrier counters must be back at zero; stale values in the scratch must not matter)."""
    K, hq, hkv = 3584, 28, 4
    gen = lambda shape, seed, sc=1.0: rnd(shape, seed, sc).cuda()
    layers = []
    for l in range(3):
        s0 = 300 + 20 * l
        layers.append(dict(wq=gen((hq * 128, K), s0, K ** -0.5), bq=gen((hq * 128,), s0 + 1, 0.1), wk=gen((hkv * 128, K), s0 + 2, K ** -0.5),
                           bk=gen((hkv * 128,), s0 + 3, 0.1), wv=gen((hkv * 128, K), s0 + 4, K ** -0.5), bv=gen((hkv * 128,), s0 + 5, 0.1),
                           wo=gen((K, K), s0 + 6, K ** -0.5), wg=gen((ff, K), s0 + 7, K ** -0.5), wu=gen((ff, K), s0 + 8, K ** -0.5),
                           wd=gen((K, ff), s0 + 9, ff ** -0.5), n1=(1.0 + 0.1 * rnd((K,), s0 + 10).float()).to(BF).cuda(),
                           n2=(1.0 + 0.1 * rnd((K,), s0 + 11).float()).to(BF).cuda()))
    ang = torch.rand((cap, 128), generator=torch.Generator().manual_seed(77)) * 6.28
    cs, sn = ang.cos().to(BF).cuda(), ang.sin().to(BF).cuda()
    kc0 = [gen((hkv, cap, 128), 400 + l) for l in range(3)]
    vc0 = [gen((hkv, cap, 128), 410 + l, 2.0) for l in range(3)]
    x0 = gen((K,), 420)
    scale = 128 ** -0.5
    scratch = ops.decode_layer_scratch(hq, cap, ff, "cuda")
    scratch[8192:].fill_(0x7f)      # stale garbage everywhere but the error flag / barrier counters
    ws = ops.decode_attention_workspace(hq, cap, "cuda")
    for token in range(2):
        step = torch.tensor([stepv + token], dtype=torch.int32, device="cuda")
        kr, vr = [t.clone() for t in kc0], [t.clone() for t in vc0]
        kl, vl = [t.clone() for t in kc0], [t.clone() for t in vc0]
        x_ref, x_new = x0, x0
        bufs = torch.zeros((2, K), dtype=BF, device="cuda")
        for l, w in enumerate(layers):
            q = ops.decode_step_qkv(x_ref, w["wq"], w["bq"], w["wk"], w["bk"], w["wv"], w["bv"], cs, sn, kr[l], vr[l], step, base, norm_w=w["n1"], eps=1e-6)
            a = ops.decode_step_attention(q, kr[l], vr[l], step, base, scale, workspace=ws)
            h1 = ops.gemv(a, w["wo"], None, res=x_ref)
            hid = ops.gemv_swiglu_norm(h1, w["n2"], 1e-6, w["wg"], w["wu"])
            x_ref = ops.gemv(hid, w["wd"], None, res=h1)
            cw = ops.decode_layer_weights(w["wq"], w["bq"], w["wk"], w["bk"], w["wv"], w["bv"], w["wo"], w["wg"], w["wu"], w["wd"], w["n1"], 1e-6,
                                          w["n2"], 1e-6)
            x_new = ops.decode_layer(cw, x_new, bufs[l & 1], cs, sn, kl[l], vl[l], step, base, scale, scratch)
            torch.cuda.synchronize()
            assert ops.decode_layer_error(scratch) == 0
            assert torch.equal(x_new, x_ref), (token, l)
            assert torch.equal(kl[l], kr[l]) and torch.equal(vl[l], vr[l]), (token, l)
            assert not torch.equal(kl[l], kc0[l])          # the row was appended
        assert torch.isfinite(x_new.float()).all()
        assert int(scratch[256:256 + 5 * 9 * 128].view(torch.int32).abs().sum()) == 0      # barriers 0 .. 4: counters back at zero (the last one's are zeroed by the next launch)


@pytest.mark.parametrize("rows,dim", [(64, 768), (832, 768), (304, 64), (7, 3072)])
def test_layernorm_affine(ops, rows, dim):
    """pe_layernorm_affine vs torch.nn.functional.layer_norm on bf16 tensors (fp32 statistics, one rounding)."""
    x, w, b = rnd((rows, dim), 141) * 1.7 + 0.3, (1.0 + 0.1 * rnd((dim,), 142).float()).to(BF), (0.1 * rnd((dim,), 143).float()).to(BF)
    ref = F.layer_norm(x, (dim,), w, b)
    out = ops.layernorm_affine(x.cuda(), w.cuda(), b.cuda())
    report(f"layernorm_affine {rows}x{dim}", out, ref, 1.01, 0.02)


@pytest.mark.parametrize("nk", [64 + 64, 768 + 64, 5000])
def test_perceiver_attention(ops, nk):
    """pe_perceiver_attention vs PerceiverAttention's core with the reference's bf16-materialised intermediates (helpers.py:52-62)."""
    H, nq = 8, 64
    q, kv = rnd((nq, H * 64), 151), rnd((nk, 2 * H * 64), 152)
    k, v = kv[:, :H * 64], kv[:, H * 64:]
    split = lambda t: t.view(1, t.shape[0], H, 64).permute(0, 2, 1, 3)
    def run(dt):
        qq, kk, vv = split(q.to(dt)), split(k.to(dt)), split(v.to(dt))
        dots = torch.einsum("b h i d, b h j d -> b h i j", qq, kk) * (64 ** -0.5)
        dots = dots - dots.amax(dim=-1, keepdim=True)
        o = torch.einsum("b h i j, b h j d -> b h i d", dots.softmax(dim=-1), vv)
        return o.permute(0, 2, 1, 3).reshape(nq, H * 64)
    ref, ref32 = run(BF), run(torch.float32)
    out = ops.perceiver_attention(q.cuda(), kv.cuda(), H)
    e_hip = (out.float().cpu() - ref32).pow(2).mean().sqrt().item()
    e_ref = (ref.float() - ref32).pow(2).mean().sqrt().item()
    print(f"[parity] perceiver attention nk={nk}: rms distance to fp32: hip {e_hip:.3e}  torch-bf16 {e_ref:.3e}")
    assert torch.isfinite(out.float()).all() and e_hip <= 1.25 * e_ref + 1e-5


def test_add_signed(ops):
    x, y = rnd((3, 40), 161), rnd((3, 40), 162)
    assert torch.equal(ops.add_(x.cuda().clone(), y.cuda(), -1.0).cpu(), x - y) and torch.equal(ops.add_(x.cuda().clone(), y.cuda()).cpu(), x + y)
