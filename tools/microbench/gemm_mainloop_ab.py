"""Interleaved timing of GEMM schedules on shapes that isolate the main loop (one round of 256 tiles, K = 32768), the per-tile fixed cost
(one round, K = 3072) and the four Linear shapes of a DiT block (GPU box, repo root):  python tools/microbench/gemm_mainloop_ab.py 17,22"""
import sys

sys.path.insert(0, '.')
import torch

from physicedit_amd import ops
from physicedit_amd._lib import lib

BF = torch.bfloat16
# "22:1" = schedule 22 with the timing-experiment knob gemm4_x = 1 (bias epilogue only; such results are WRONG on purpose)
variants = [tuple(int(t) for t in (v + ":0").split(":")[:2]) for v in (sys.argv[1] if len(sys.argv) > 1 else "15,17,22").split(",")]
only = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 99
fp8 = "--fp8" in sys.argv          # e4m3 operands (pe_gemm_e4m3): BASELINE configs[2]'s Linears
g = torch.Generator(device='cuda').manual_seed(0)


def rnd(shape, scale=1.0):
    return (torch.randn(shape, generator=g, device='cuda') * scale).to(BF)


shapes = [(4096, 4096, 32768, "bias"), (4096, 4096, 3072, "bias"), (8192, 4096, 3072, "bias"), (8704, 12288, 3072, "gelu_sigmoid"),
          (8704, 3072, 12288, "gate_res"), (8704, 9216, 3072, "bias"), (8704, 3072, 3072, "gate_res")]
for (M, N, K, epi) in shapes[:only]:
    if fp8 and K > 12288:
        K = 12288                   # the row quantiser's limit
    x, w, b, gate = rnd((M, K)), rnd((N, K), K ** -0.5), rnd((N,)), rnd((N,), 0.5)
    out = rnd((M, N))
    if fp8:
        xq, sc = ops.quantize_rows_e4m3(x)
        w8 = w.to(torch.float8_e4m3fn)
    fl = 2.0 * M * N * K
    reps = max(4, int(2e15 / fl / 100))
    res = {v: [] for v in variants}
    for rnd_i in range(5):
        for v in variants:
            assert lib().pe_debug_set(b"gemm_variant", v[0]) == 0 and lib().pe_debug_set(b"gemm4_x", v[1]) == 0
            def run():
                if fp8:
                    if epi == "gate_res":
                        ops.gemm_e4m3(xq, sc, w8, b, epi, gate=gate, res=out, out=out)
                    else:
                        ops.gemm_e4m3(xq, sc, w8, b, epi, out=out)
                elif epi == "gate_res":
                    ops.gemm(x, w, b, epi, gate=gate, res=out, out=out)
                else:
                    ops.gemm(x, w, b, epi, out=out)
            run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record(); torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / reps)
    print(f"{'e4m3 ' if fp8 else ''}{M}x{N}x{K} {epi}: " + "  ".join(f"v{v[0]}{':x%d' % v[1] if v[1] else ''}: {sorted(t)[2]*1e3:.0f}us {fl/sorted(t)[2]/1e9:.0f} TF" for v, t in res.items()), flush=True)
lib().pe_debug_set(b"gemm_variant", 17); lib().pe_debug_set(b"gemm4_x", 0)
