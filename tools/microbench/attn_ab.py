"""In-process interleaved A/B of the flash-attention variants on MI355X (run on the GPU box from the repo root):

    python tools/microbench/attn_ab.py [variants, default 5,4,0]

Variants 5 / 6 (folded scale and max) are timed on the pre-scaled Q they take (pe_flash_attn_prescaled), the others on the plain Q.

H = 24, D = 128, N(0,1) operands, S = 8704 / 8464 (the two CFG branches of the headline geometry) and 11497 (configs[4]);
median of 7 interleaved rounds x 10 launches per variant.  Also checks variant 3 == variant 0 bit for bit."""
import os
import sys

sys.path.insert(0, '.')
import torch

from physicedit_amd import ops
from physicedit_amd._lib import lib

BF = torch.bfloat16
variants = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else [5, 4, 0]
g = torch.Generator(device='cuda').manual_seed(0)
H = 24
for S in (8704, 8464, 11497):
    sp = ops.s_pad_of(S)
    q = torch.zeros((H, sp, 128), dtype=BF, device='cuda'); q[:, :S] = torch.randn((H, S, 128), generator=g, device='cuda').to(BF)
    k = torch.zeros((H, sp, 128), dtype=BF, device='cuda'); k[:, :S] = (torch.randn((H, S, 128), generator=g, device='cuda') * float(os.environ.get('AB_KSCALE', '1'))).to(BF)      # AB_KSCALE > 1: wider scores, the lazy-max raise path runs on many tiles
    c = 0.08838834764831845 * 1.4426950408889634
    qc = (q.float() * c).to(BF)
    vt = ops.pack_vt(torch.randn((H, S, 128), generator=g, device='cuda').to(BF), sp)
    out = torch.empty((S, H * 128), dtype=BF, device='cuda')
    nbytes = lib().pe_flash_attn_workspace_bytes(H, S)
    res = {v: [] for v in variants}
    outs = {}
    for rnd in range(7):
        for v in variants:
            assert lib().pe_debug_set(b"attn_variant", v) == 0
            qq, pre = (qc, True) if v >= 5 else (q, False)
            ops.flash_attn(qq, k, vt, S, out=out, q_prescaled=pre)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.flash_attn(qq, k, vt, S, out=out, q_prescaled=pre)
            e1.record(); torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / 10)
            if rnd == 0:
                outs[v] = out.clone()
    fl = 4.0 * S * S * 128 * H
    print(f"S={S}: " + "  ".join(f"v{v}: {sorted(t)[len(t)//2]*1e3:.0f} us {fl/sorted(t)[len(t)//2]/1e9:.0f} TF (best {min(t)*1e3:.0f} us)" for v, t in res.items()), flush=True)
    if 0 in outs and 3 in outs:
        print(f"   v3 == v0 bit for bit: {torch.equal(outs[0], outs[3])}", flush=True)
    if 0 in outs and 4 in outs:
        d = (outs[0].float() - outs[4].float()).abs()
        print(f"   v4 vs v0: identical {(d == 0).float().mean().item()*100:.1f} %, max |d| {d.max().item():.3e}", flush=True)
    if 5 in outs and 7 in outs:
        d = (outs[5].float() - outs[7].float()).abs()
        print(f"   v7 vs v5: identical {(d == 0).float().mean().item()*100:.1f} %, max |d| {d.max().item():.3e}", flush=True)
    if 5 in outs and 9 in outs:
        d = (outs[5].float() - outs[9].float()).abs()
        print(f"   v9 vs v5: identical {(d == 0).float().mean().item()*100:.1f} %, max |d| {d.max().item():.3e}", flush=True)
    if 5 in outs and 4 in outs:
        d = (outs[5].float() - outs[4].float()).abs()
        print(f"   v5 vs v4: identical {(d == 0).float().mean().item()*100:.1f} %, max |d| {d.max().item():.3e}", flush=True)
lib().pe_debug_set(b"attn_variant", 5)
