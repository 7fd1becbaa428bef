"""pe_flash_attn_fp8 variant 1 against variant 0 at a few sizes (max abs / rms rel of the difference): python tools/microbench/attn_fp8_check.py"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from physicedit_amd import ops
from physicedit_amd._lib import lib, check, stream_ptr
BF = torch.bfloat16
g = torch.Generator(device='cuda').manual_seed(0)
for H, S in ((4, 64), (256, 128), (256, 192), (256, 256), (256, 320), (4, 700), (4, 4096)):
    sp = ops.s_pad_of(S)
    q = torch.zeros((H, sp, 128), dtype=BF, device='cuda'); q[:, :S] = torch.randn((H, S, 128), generator=g, device='cuda').to(BF)
    k = torch.zeros((H, sp, 128), dtype=BF, device='cuda'); k[:, :S] = torch.randn((H, S, 128), generator=g, device='cuda').to(BF)
    vt = ops.pack_vt(torch.randn((H, S, 128), generator=g, device='cuda').to(BF), sp)
    n = lib().pe_flash_attn_fp8_scratch_bytes(H, sp)
    scratch = torch.empty((n + 256,), dtype=torch.uint8, device="cuda"); base = (scratch.data_ptr() + 255) // 256 * 256
    nb = lib().pe_flash_attn_workspace_bytes(H, S); ws = torch.empty((nb,), dtype=torch.uint8, device="cuda")
    outs = []
    for variant in (0, 1, 2):
        check(lib().pe_debug_set(b"attn_fp8_variant", variant), "knob")
        out = torch.empty((S, H * 128), dtype=BF, device='cuda')
        check(lib().pe_flash_attn_fp8(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), H, S, sp, H * 128, base, n, ws.data_ptr(), nb, stream_ptr()), "fp8")
        torch.cuda.synchronize(); outs.append(out.float())
    for variant in (1, 2):
        d = outs[variant] - outs[0]
        print(f"H={H} S={S} variant {variant} vs 0: max abs {float(d.abs().max()):.3e} rms rel {float((d.pow(2).mean() / outs[0].pow(2).mean()).sqrt()):.3e}", flush=True)
