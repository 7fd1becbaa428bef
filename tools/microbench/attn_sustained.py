"""Sustained A/B of two attention variants (run on the GPU box from the repo root):  python tools/microbench/attn_sustained.py [5,7] [launches]

attn_ab.py times bursts of 10 launches; here every measurement is `launches` (default 1500, about a second) back-to-back launches of ONE
variant, so the chip's power management is in its steady state -- the regime the pipeline runs the operator in."""
import sys

sys.path.insert(0, '.')
import torch

from physicedit_amd import ops
from physicedit_amd._lib import lib

BF = torch.bfloat16
variants = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else [5, 7]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
g = torch.Generator(device='cuda').manual_seed(0)
H, S = 24, 8704
sp = ops.s_pad_of(S)
q = torch.zeros((H, sp, 128), dtype=BF, device='cuda'); q[:, :S] = torch.randn((H, S, 128), generator=g, device='cuda').to(BF)
k = torch.zeros((H, sp, 128), dtype=BF, device='cuda'); k[:, :S] = torch.randn((H, S, 128), generator=g, device='cuda').to(BF)
qc = (q.float() * (0.08838834764831845 * 1.4426950408889634)).to(BF)
vt = ops.pack_vt(torch.randn((H, S, 128), generator=g, device='cuda').to(BF), sp)
out = torch.empty((S, H * 128), dtype=BF, device='cuda')
for rnd in range(3):
    for v in variants:
        assert lib().pe_debug_set(b"attn_variant", v) == 0
        qq, pre = (qc, True) if v >= 5 else (q, False)
        ops.flash_attn(qq, k, vt, S, out=out, q_prescaled=pre)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ops.flash_attn(qq, k, vt, S, out=out, q_prescaled=pre)
        e1.record(); torch.cuda.synchronize()
        print(f"round {rnd} v{v}: {e0.elapsed_time(e1) / n * 1e3:.1f} us per launch over {n} launches", flush=True)
lib().pe_debug_set(b"attn_variant", 5)
