"""Where does a GEMM work-group's time go?  s_memtime stamps of wave 0 of every work-group of schedule 15 (one tile per
work-group; `pe_debug_set_ptr("gemm_stamps", buf)`), on the four Linear shapes of a DiT block incl. the QKV epilogue.

    python tools/microbench/gemm_stamps.py

stamps: 0 kernel entry, 1 main loop starts (prologue done), 2 main loop done, 3 LDS drained + barrier, 4 accumulators staged,
5 epilogue maths + stores issued."""
import sys

sys.path.insert(0, '.')
import numpy as np
import torch

from physicedit_amd import ops
from physicedit_amd._lib import lib

BF = torch.bfloat16
g = torch.Generator(device='cuda').manual_seed(0)


def rnd(shape, scale=1.0):
    return (torch.randn(shape, generator=g, device='cuda') * scale).to(BF)


def report(tag, st, ms, flops, K):
    s = st.cpu().numpy().astype(np.float64)
    d = np.diff(s[:, :6], axis=1)
    tot = s[:, 5] - s[:, 0]
    names = ["prologue", "mainloop", "drain+bar", "lds_stage", "epi+stores"]
    order = np.argsort(s[:, 0])
    span = s[:, 5].max() - s[:, 0].min()
    print(f"\n{tag}: {len(s)} tiles, kernel {ms*1e3:.0f} us ({flops/ms/1e9:.0f} TF/s); span {span:.0f} ticks -> {span/ms/1e3:.0f} MHz", flush=True)
    for nm, idx in (("first-round WGs", order[:256]), ("later WGs", order[256:])):
        if len(idx):
            print(f"  {nm} ({len(idx)}): total {tot[idx].mean():.0f}; " + "  ".join(f"{n} {d[idx, i].mean():.0f}" for i, n in enumerate(names)))
    nk = K // 64
    print(f"  main loop per K tile: {d[:, 1].mean()/nk:.0f} ticks (MFMA issue floor 2048); fixed per tile: {(tot - d[:, 1]).mean():.0f} = "
          f"{(tot - d[:, 1]).mean()/tot.mean()*100:.1f} % of the work-group's time")


def run(M, N, K, epi):
    x, w, b = rnd((M, K)), rnd((N, K), K ** -0.5), rnd((N,))
    ntiles = ((M + 255) // 256) * ((N + 255) // 256)
    st = torch.zeros((ntiles, 8), dtype=torch.int64, device='cuda')
    lib().pe_debug_set(b"gemm_variant", 15)
    lib().pe_debug_set_ptr(b"gemm_stamps", st.data_ptr())
    if epi == "qkv":
        H = N // 384
        nq, nk_ = rnd((128,)), rnd((128,))
        ang = torch.rand((M, 64), generator=g, device='cuda') * 6.28
        cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
        q, k, vt = ops.alloc_qkv(H, M, 'cuda')
        call = lambda: ops.qkv_rmsnorm_rope(x, w, b, nq, nk_, cos, sin, q, k, vt, 0)
    else:
        out = torch.empty((M, N), dtype=BF, device='cuda')
        res = rnd((M, N)) if epi == "gate_res" else None
        gate = rnd((N,)) if epi == "gate_res" else None
        call = lambda: ops.gemm(x, w, b, epi, gate=gate, res=res, out=out)
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); torch.cuda.synchronize()
    lib().pe_debug_set_ptr(b"gemm_stamps", None)
    report(f"{M}x{N}x{K} {epi}", st, e0.elapsed_time(e1), 2.0 * M * N * K, K)


run(8704, 12288, 3072, "gelu_sigmoid")
run(8704, 9216, 3072, "qkv")
run(8704, 9216, 3072, "bias")
run(8704, 3072, 3072, "gate_res")
run(8704, 3072, 12288, "gate_res")
lib().pe_debug_set(b"gemm_variant", 17)
