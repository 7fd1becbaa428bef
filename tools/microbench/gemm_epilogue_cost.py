"""What the fused epilogue costs a tile TODAY (round 6): interleaved timing of the DiT block's four Linears (+ the QKV epilogue) with the
default schedule against the timing-only build that stops a tile behind its main loop (knob gemm_no_epilogue: nothing is stored).  The
difference is the ceiling of anything that hides the epilogue under the next tile's MFMAs.  GPU box, repo root:
    python tools/microbench/gemm_epilogue_cost.py [--fp8]"""
import sys

sys.path.insert(0, '.')
import torch

from physicedit_amd import ops
from physicedit_amd._lib import lib
from physicedit_amd.rope import RopeCache

BF = torch.bfloat16
fp8 = "--fp8" in sys.argv
g = torch.Generator(device='cuda').manual_seed(0)


def rnd(shape, scale=1.0):
    return (torch.randn(shape, generator=g, device='cuda') * scale).to(BF)


shapes = [(8704, 12288, 3072, "gelu_sigmoid"), (8704, 3072, 12288, "gate_res"), (8704, 9216, 3072, "qkv"), (8704, 9216, 3072, "bias"),
          (8704, 3072, 3072, "gate_res")]
for (M, N, K, epi) in shapes:
    x, w, b, gate = rnd((M, K)), rnd((N, K), K ** -0.5), rnd((N,)), rnd((N,), 0.5)
    out = rnd((M, N)) if epi != "qkv" else None
    if epi == "qkv":
        if fp8:
            continue
        q, k, vt = ops.alloc_qkv(24, M, "cuda")
        ci, si, ct, stt = RopeCache("cuda").get([(1, 64, 64), (1, 64, 64)], 512)
        nq, nk_ = rnd((128,)), rnd((128,))
    if fp8:
        xq, sc = ops.quantize_rows_e4m3(x)
        w8 = w.to(torch.float8_e4m3fn)
    fl = 2.0 * M * N * K
    reps = max(4, int(2e15 / fl / 100))
    res = {0: [], 1: []}

    def run():
        if epi == "qkv":
            ops.qkv_rmsnorm_rope(x[:8192], w, b, nq, nk_, ci, si, q, k, vt, 0, q_scale=0.1275)
        elif fp8:
            if epi == "gate_res":
                ops.gemm_e4m3(xq, sc, w8, b, epi, gate=gate, res=out, out=out)
            else:
                ops.gemm_e4m3(xq, sc, w8, b, epi, out=out)
        elif epi == "gate_res":
            ops.gemm(x, w, b, epi, gate=gate, res=out, out=out)
        else:
            ops.gemm(x, w, b, epi, out=out)
    for rnd_i in range(5):
        for ne in (0, 1):
            assert lib().pe_debug_set(b"gemm_no_epilogue", ne) == 0
            run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record(); torch.cuda.synchronize()
            res[ne].append(e0.elapsed_time(e1) / reps)
    lib().pe_debug_set(b"gemm_no_epilogue", 0)
    t0, t1 = sorted(res[0])[2], sorted(res[1])[2]
    Mx = 8192 if epi == "qkv" else M
    print(f"{'e4m3 ' if fp8 else ''}{Mx}x{N}x{K} {epi}: with epilogue {t0*1e3:.1f} us ({2.0*Mx*N*K/t0/1e9:.0f} TF/s)  main loop only {t1*1e3:.1f} us "
          f"({2.0*Mx*N*K/t1/1e9:.0f} TF/s)  epilogue = {(t0-t1)/t0*100:.1f} % of the launch", flush=True)
