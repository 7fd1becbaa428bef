"""Attention variants 5 / 7 timed per launch (HIP events) alone and with two block-sized GEMM launches in front of every attention launch, as in
the pipeline (run on the GPU box from the repo root):  python tools/microbench/attn_after_gemm.py [variants, default 5,7]"""
import sys
sys.path.insert(0, '.')
import torch
from physicedit_amd import ops
from physicedit_amd._lib import lib
BF = torch.bfloat16
g = torch.Generator(device='cuda').manual_seed(0)
H, S = 24, 8704
sp = ops.s_pad_of(S)
q = torch.zeros((H, sp, 128), dtype=BF, device='cuda'); q[:, :S] = torch.randn((H, S, 128), generator=g, device='cuda').to(BF)
k = torch.zeros((H, sp, 128), dtype=BF, device='cuda'); k[:, :S] = torch.randn((H, S, 128), generator=g, device='cuda').to(BF)
qc = (q.float() * (0.08838834764831845 * 1.4426950408889634)).to(BF)
vt = ops.pack_vt(torch.randn((H, S, 128), generator=g, device='cuda').to(BF), sp)
out = torch.empty((S, H * 128), dtype=BF, device='cuda')
x = (torch.randn((8704, 3072), generator=g, device='cuda')).to(BF)
w1 = (torch.randn((12288, 3072), generator=g, device='cuda') * 3072 ** -0.5).to(BF)
w2 = (torch.randn((3072, 12288), generator=g, device='cuda') * 12288 ** -0.5).to(BF)
hbuf = torch.empty((8704, 12288), dtype=BF, device='cuda'); ybuf = torch.empty((8704, 3072), dtype=BF, device='cuda')
for mode in ("alone", "after 2 GEMMs"):
    for rnd in range(2):
        for v in ([int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else (5, 7)):
            assert lib().pe_debug_set(b"attn_variant", v) == 0
            evs = []
            for i in range(60):
                if mode != "alone":
                    ops.gemm(x, w1, None, "gelu_sigmoid", out=hbuf); ops.gemm(hbuf, w2, None, "bias", out=ybuf)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ops.flash_attn(qc, k, vt, S, out=out, q_prescaled=True); e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            t = sorted(a.elapsed_time(b) for a, b in evs[10:])
            print(f"{mode} round {rnd} v{v}: median {t[len(t)//2]*1e3:.0f} us  mean {sum(t)/len(t)*1e3:.0f} us", flush=True)
