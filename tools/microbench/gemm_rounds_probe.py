"""What one round / 1.59 rounds / 2 / 3 rounds of 256 x 256 tiles cost under GEMM schedules 15 / 17 / 19 (bias epilogue, hot operands): the
numbers of profiles/r04_gemm_sk_probe.log.  python tools/microbench/gemm_rounds_probe.py  (GPU box, from the repo root)"""
import sys
sys.path.insert(0, '.')
import torch
from physicedit_amd import ops
from physicedit_amd._lib import lib
BF = torch.bfloat16
g = torch.Generator(device='cuda').manual_seed(0)
n = lib().pe_gemm_workspace_bytes()
buf = torch.zeros((n + 256,), dtype=torch.uint8, device="cuda")
ws = buf[(-buf.data_ptr()) % 256:][:n]
assert lib().pe_debug_set_ptr(b"gemm_workspace", ws.data_ptr()) == 0
def rnd(shape, scale=1.0):
    return (torch.randn(shape, generator=g, device='cuda') * scale).to(BF)
for (M, N, K) in ((8192, 4096, 3072), (8192, 2048, 3072), (8704, 3072, 3072), (8192, 6144, 3072), (8192, 4096, 12288), (8704, 3072, 12288)):
    x, w, bb = rnd((M, K)), rnd((N, K), K ** -0.5), rnd((N,))
    out = torch.empty((M, N), dtype=BF, device="cuda")
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    r = {}
    for rnd_i in range(5):
        for v in (15, 17, 19):
            lib().pe_debug_set(b"gemm_variant", v)
            lib().pe_debug_set(b"gemm_persist_wgs", 0)
            ops.gemm(x, w, bb, "bias", out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(12):
                ops.gemm(x, w, bb, "bias", out=out)
            e1.record(); torch.cuda.synchronize()
            r.setdefault(v, []).append(e0.elapsed_time(e1) / 12)
    fl = 2.0 * M * N * K
    print(f"{M}x{N}x{K} tiles {tiles} ({tiles/256:.2f} rounds): " + "  ".join(f"v{v} {sorted(t)[2]*1e3:.0f} us {fl/sorted(t)[2]/1e9:.0f} TF" for v, t in r.items()), flush=True)
lib().pe_debug_set(b"gemm_variant", 17)
