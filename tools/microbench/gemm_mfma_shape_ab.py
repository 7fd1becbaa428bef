"""Interleaved A/B of the bf16 GEMM's two MFMA shapes (knob "gemm_mfma16": 1 = v_mfma_f32_16x16x32_bf16, 0 = 32x32x16) on the four Linear shapes
of a DiT block and the text stream's ragged shape, schedule 17, hot and cold weights (run on the GPU box from the repo root):

    python tools/microbench/gemm_mfma_shape_ab.py [--fp8]      (--fp8: e4m3 operands, v_mfma_scale_f32_16x16x128_f8f6f4 against 32x32x64)
"""
import sys

sys.path.insert(0, '.')
import torch

from physicedit_amd import ops
from physicedit_amd._lib import lib

BF = torch.bfloat16
FP8 = "--fp8" in sys.argv
g = torch.Generator(device='cuda').manual_seed(0)


def knob(k, v):
    if k == "gemm_mfma16" and FP8:
        v = 3 if v else 1        # bit 1 = the e4m3 kernels' shape
    assert lib().pe_debug_set(k.encode(), v) == 0


def rnd(shape, scale=1.0):
    return (torch.randn(shape, generator=g, device='cuda') * scale).to(BF)


shapes = [(8704, 12288, 3072, "gelu_sigmoid"), (8704, 3072, 12288, "gate_res"), (8704, 9216, 3072, "bias"), (8704, 3072, 3072, "gate_res"),
          (784, 12288, 3072, "gelu_sigmoid")]
NW = 12
knob("gemm_variant", 17)
# accuracy of the two shapes against fp64
for (M, N, K) in ((2048, 3072, 3072), (1024, 3072, 12288)):
    x, w, b = rnd((M, K)), rnd((N, K), K ** -0.5), rnd((N,), 0.1)
    ref = x.double() @ w.double().T + b.double()
    if FP8:
        xq, sc = ops.quantize_rows_e4m3(x)
        w8 = w.to(torch.float8_e4m3fn)
        ref = (xq.float() * sc[:, None]).double() @ w8.double().T + b.double()
    for s in (1, 0):
        knob("gemm_mfma16", s)
        o = ops.gemm_e4m3(xq, sc, w8, b, "bias") if FP8 else ops.gemm(x, w, b, "bias")
        e = (o.double() - ref).abs()
        print(f"accuracy {M}x{N}x{K} mfma16={s}: mean |err| {e.mean().item():.3e} max {e.max().item():.3e}", flush=True)


def time_shape(M, N, K, epi, rounds=5, reps=12):
    xs = [rnd((M, K)) for _ in range(3)]
    ws = [rnd((N, K), K ** -0.5) for _ in range(NW)]
    if FP8:
        xqs = [ops.quantize_rows_e4m3(x) for x in xs]
        ws = [w.to(torch.float8_e4m3fn) for w in ws]
    b = rnd((N,)); gate = rnd((N,), 0.5)
    outs = [rnd((M, N)) for _ in range(3)]
    fl = 2.0 * M * N * K
    for cold in (0, 1):
        res = {0: [], 1: []}
        for _ in range(rounds):
            for s in (0, 1):
                knob("gemm_mfma16", s)

                def run(i):
                    j = i % NW if cold else 0
                    k = i % 3 if cold else 0
                    kw = dict(gate=gate, res=outs[k]) if epi == "gate_res" else {}
                    if FP8:
                        ops.gemm_e4m3(xqs[k][0], xqs[k][1], ws[j], b, epi, out=outs[k], **kw)
                    else:
                        ops.gemm(xs[k], ws[j], b, epi, out=outs[k], **kw)
                run(0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(reps):
                    run(i)
                e1.record(); torch.cuda.synchronize()
                res[s].append(e0.elapsed_time(e1) / reps)
        med = {s: sorted(t)[len(t) // 2] for s, t in res.items()}
        print(f"{M}x{N}x{K} {epi} cold={cold}: " + "  ".join(f"mfma16={s}: {med[s]*1e3:.0f}us {fl/med[s]/1e9:.0f} TF" for s in (0, 1)) +
              f"   16/32 time ratio {med[1]/med[0]:.3f}", flush=True)


for (M, N, K, epi) in shapes:
    time_shape(M, N, K, epi)
knob("gemm_mfma16", 1)
lib().pe_debug_set(b"gemm_mfma16", 3)
