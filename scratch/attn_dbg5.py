import sys, math
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, torch.nn.functional as F
from physicedit_amd import ops
from physicedit_amd._lib import lib
from test_gpu_kernels import _fold_case, _dev_qkv
BF = torch.bfloat16
for S, ws in ((128, False), (192, False), (320, False), (700, False), (700, True)):
    qb, qc, k, v, ref, ref32 = _fold_case(S, 500 + S)
    qd, kd, vt = _dev_qkv(ops, qc, k, v, S)
    for var in (5, 6):
        lib().pe_debug_set(b"attn_variant", var)
        out = ops.flash_attn(qd, kd, vt, S, q_prescaled=True, workspace=ws).float().cpu()
        d = (out - ref32).abs()
        print(f"S={S} ws={ws} v{var}: rms {d.pow(2).mean().sqrt():.3e}  max {d.max():.3e}")
        rows = d.max(dim=1).values
        bad = (rows > 0.02).nonzero().flatten()
        print("   bad rows:", bad.numel(), bad[:20].tolist(), " ... per 32-row block:", [(rows[i:i+32] > 0.02).sum().item() for i in range(0, min(S, 512), 32)])
        heads = d.reshape(S, 24, 128).amax(dim=(0, 2))
        print("   per head max:", [f"{x:.2f}" for x in heads.tolist()[:8]])
