"""MFMA skipping on ragged M tiles (round 5): the negative prompt of the headline geometry has 272 tokens = one full M tile + 16 rows, so
its text-stream GEMMs compute 512 rows; with "gemm_skip_ragged" (default 1) the 32-row blocks beyond M skip their MFMAs.  Interleaved
timing of the knob on the negative forward's four Linear shapes as ONE problem of 8192 + 272 rows (34 M tiles, the last with 16 rows),
plus bit-identity of the two settings.  GPU box, repo root:  python tools/microbench/gemm_ragged_ab.py"""
import sys

sys.path.insert(0, '.')
import torch

from physicedit_amd import ops
from physicedit_amd._lib import lib

BF = torch.bfloat16
g = torch.Generator(device='cuda').manual_seed(0)


def rnd(shape, scale=1.0):
    return (torch.randn(shape, generator=g, device='cuda') * scale).to(BF)


for (M, N, K, epi) in [(8464, 12288, 3072, "gelu_sigmoid"), (8464, 3072, 12288, "gate_res"), (8464, 9216, 3072, "bias"), (8464, 3072, 3072, "gate_res"),
                       (272, 12288, 3072, "gelu_sigmoid"), (272, 3072, 12288, "bias")]:
    x, w, b, gate = rnd((M, K)), rnd((N, K), K ** -0.5), rnd((N,)), rnd((N,), 0.5)
    res0 = rnd((M, N))
    outs = {}
    for skip in (0, 1):
        lib().pe_debug_set(b"gemm_skip_ragged", skip)
        for v in (15, 17):
            lib().pe_debug_set(b"gemm_variant", v)
            outs[(skip, v)] = ops.gemm(x, w, b, epi, gate=gate, res=res0) if epi == "gate_res" else ops.gemm(x, w, b, epi)
    same = all(torch.equal(outs[(0, 15)], o) for o in outs.values())
    lib().pe_debug_set(b"gemm_variant", 17)
    out = rnd((M, N))
    fl = 2.0 * M * N * K
    reps = max(8, int(2e15 / fl / 100))
    t = {0: [], 1: []}
    for _ in range(7):
        for skip in (0, 1):
            lib().pe_debug_set(b"gemm_skip_ragged", skip)
            def run():
                if epi == "gate_res":
                    ops.gemm(x, w, b, epi, gate=gate, res=out, out=out)
                else:
                    ops.gemm(x, w, b, epi, out=out)
            run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record(); torch.cuda.synchronize()
            t[skip].append(e0.elapsed_time(e1) / reps)
    m0, m1 = sorted(t[0])[3], sorted(t[1])[3]
    print(f"{M}x{N}x{K} {epi}: bit-identical {same};  all MFMAs {m0*1e3:.1f} us  skip {m1*1e3:.1f} us  ({(m1/m0-1)*100:+.2f} %)", flush=True)
lib().pe_debug_set(b"gemm_skip_ragged", 1)
