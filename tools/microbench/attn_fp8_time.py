"""pe_flash_attn_fp8 (statistics + quantisation + e4m3 flash kernel) at S = 8704 / 8464, H = 24: median of 5 x 10 launches.
ATTN_FP8_ONLY=<variant>: that variant at S = 8704 only, 20 launches (the form the PMC passes of profiles/r06_attn_fp8_pmc.md were taken on).
python tools/microbench/attn_fp8_time.py  (GPU box, from the repo root; under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from physicedit_amd import ops
from physicedit_amd._lib import lib, check, stream_ptr
BF = torch.bfloat16
g = torch.Generator(device='cuda').manual_seed(0)
H = 24
ONLY = os.environ.get("ATTN_FP8_ONLY")
for S in ((8704,) if ONLY else (8704, 8464)):
    sp = ops.s_pad_of(S)
    q = torch.zeros((H, sp, 128), dtype=BF, device='cuda'); q[:, :S] = torch.randn((H, S, 128), generator=g, device='cuda').to(BF)
    k = torch.zeros((H, sp, 128), dtype=BF, device='cuda'); k[:, :S] = torch.randn((H, S, 128), generator=g, device='cuda').to(BF)
    v = torch.randn((H, S, 128), generator=g, device='cuda').to(BF)
    vt = ops.pack_vt(v, sp)
    truth = torch.nn.functional.scaled_dot_product_attention(q[None, :, :S].float(), k[None, :, :S].float(), v[None].float())[0].permute(1, 0, 2).reshape(S, H * 128)
    out = torch.empty((S, H * 128), dtype=BF, device='cuda')
    n = lib().pe_flash_attn_fp8_scratch_bytes(H, sp)
    scratch = torch.empty((n + 256,), dtype=torch.uint8, device="cuda"); base = (scratch.data_ptr() + 255) // 256 * 256
    nb = lib().pe_flash_attn_workspace_bytes(H, S); ws = torch.empty((nb,), dtype=torch.uint8, device="cuda")
    def run():
        check(lib().pe_flash_attn_fp8(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), H, S, sp, H * 128, base, n, ws.data_ptr(), nb, stream_ptr()), "fp8")
    if ONLY:
        check(lib().pe_debug_set(b"attn_fp8_variant", int(ONLY)), "knob")
        for _ in range(20): run()
        torch.cuda.synchronize()
        continue
    ref = None
    for variant in (0, 1, 2, 3, 4):
        check(lib().pe_debug_set(b"attn_fp8_variant", variant), "knob")
        run(); torch.cuda.synchronize()
        if variant == 0:
            ref = out.float().clone()
        elif variant == 3:
            print("  variant 3 == variant 1:", bool(torch.equal(out, o1)), flush=True)
        else:
            if variant == 1:
                o1 = out.clone()
            d = out.float() - ref
            print(f"  variant {variant} vs 0: max abs", float(d.abs().max()), "rms rel", float((d.pow(2).mean() / ref.pow(2).mean()).sqrt()), flush=True)
        print(f"  variant {variant}: rms distance to the fp32 attention {float((out.float() - truth).pow(2).mean().sqrt()):.4e}", flush=True)
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10)
        print(f"S={S} variant {variant}: e4m3 attention (stats + quantise + kernel) {sorted(ts)[2]*1e3:.0f} us", flush=True)
