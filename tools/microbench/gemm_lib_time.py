"""The block's four Linear shapes timed with whatever library PE_LIB_PATH names (A/B of builds, one process per build; run the builds
alternately on one box).  GPU box, repo root:  PE_LIB_PATH=build_ab/lib_x.so python tools/microbench/gemm_lib_time.py"""
import os
import sys

sys.path.insert(0, '.')
import torch

from physicedit_amd import ops
from physicedit_amd._lib import lib

BF = torch.bfloat16
g = torch.Generator(device='cuda').manual_seed(0)
rnd = lambda shape, scale=1.0: (torch.randn(shape, generator=g, device='cuda') * scale).to(BF)
stash = torch.empty(int(lib().pe_gemm_stash_bytes()) + 256, dtype=torch.uint8, device="cuda") if hasattr(lib(), "pe_gemm_stash_bytes") else None
if stash is not None:
    lib().pe_debug_set_ptr(b"gemm_stash", (stash.data_ptr() + 255) // 256 * 256)
out_s = []
for (M, N, K, epi) in ((8704, 3072, 3072, "gate_res"), (8704, 3072, 12288, "gate_res"), (8704, 12288, 3072, "gelu_sigmoid"), (8704, 9216, 3072, "bias")):
    x, w, b, gate = rnd((M, K)), rnd((N, K), K ** -0.5), rnd((N,)), rnd((N,), 0.5)
    out = rnd((M, N))
    run = (lambda: ops.gemm(x, w, b, epi, gate=gate, res=out, out=out)) if epi == "gate_res" else (lambda: ops.gemm(x, w, b, epi, out=out))
    fl = 2.0 * M * N * K
    reps = max(4, int(2e15 / fl / 100))
    ts = []
    for r in range(5):
        run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    out_s.append(f"{M}x{N}x{K} {epi} {sorted(ts)[2]*1e3:.1f} us")
print(os.environ.get("PE_LIB_PATH", "default library") + ": " + " | ".join(out_s), flush=True)
