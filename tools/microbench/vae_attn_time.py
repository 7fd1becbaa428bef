"""VAE mid-block attention (one head, D = 384) at the latent sizes of 1024 x 1024 (N = 16384) and 1328 x 1328 (N = 27556): time per launch
and FLOP/s (4 N^2 D).  GPU box, repo root:  python tools/microbench/vae_attn_time.py"""
import sys

sys.path.insert(0, '.')
import torch

from physicedit_amd._lib import check, lib, stream_ptr

BF = torch.bfloat16
g = torch.Generator(device='cuda').manual_seed(0)
for N in (16384, 27556, 4096):
    qkv = (torch.randn((N, 1152), generator=g, device='cuda') * 0.7).to(BF)
    sc = torch.empty((int(lib().pe_vae_attention_scratch_bytes(N)) + 256,), dtype=torch.uint8, device='cuda')
    sp = (sc.data_ptr() + 255) // 256 * 256
    out = torch.empty((N, 384), dtype=BF, device='cuda')
    run = lambda: check(lib().pe_vae_attention(qkv.data_ptr(), sp, out.data_ptr(), N, stream_ptr()))
    run()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    t = sorted(ts)[2]
    q, k, v = qkv.float()[None, None].chunk(3, dim=-1)
    ref = torch.nn.functional.scaled_dot_product_attention(q[:, :, :2048], k, v)[0, 0]
    err = (out[:2048].float() - ref).pow(2).mean().sqrt().item()
    print(f"vae_attention N={N}: {t*1e3:.0f} us  {4.0*N*N*384/t/1e9:.0f} TF/s (Vt transpose + attention + combine)   rms vs fp32 (first 2048 rows) {err:.3e}", flush=True)
