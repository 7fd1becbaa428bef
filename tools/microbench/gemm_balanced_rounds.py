"""408-tile Linears (N = 3072: out-proj, MLP-down): one tile per work-group on 256 + 152 CUs (schedule 15, the shipped choice) against
schedule 17 with a persistent grid of G work-groups that own ~2 tiles each (pe_debug_set("gemm_persist_wgs", G)): balanced rounds on
fewer CUs.  Interleaved, hot and cold weights.

    python tools/microbench/gemm_balanced_rounds.py [G list, default 200,208,216,256]"""
import sys

sys.path.insert(0, '.')
import torch

from physicedit_amd import ops
from physicedit_amd._lib import lib

BF = torch.bfloat16
Gs = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else [200, 208, 216, 256]
g = torch.Generator(device='cuda').manual_seed(0)


def rnd(shape, scale=1.0):
    return (torch.randn(shape, generator=g, device='cuda') * scale).to(BF)


def knob(k, v):
    assert lib().pe_debug_set(k.encode(), v) == 0


configs = [("v15", 15, 0)] + [(f"v17/G{G}", 17, G) for G in Gs]
for (M, N, K) in ((8704, 3072, 3072), (8704, 3072, 12288), (8464, 3072, 3072), (8464, 3072, 12288)):
    xs = [rnd((M, K)) for _ in range(3)]
    ws = [rnd((N, K), K ** -0.5) for _ in range(12)]
    gate = rnd((N,), 0.5)
    outs = [rnd((M, N)) for _ in range(3)]
    ref = None
    for cold in (0, 1):
        res = {c[0]: [] for c in configs}
        for rnd_i in range(5):
            for name, v, G in configs:
                knob("gemm_variant", v); knob("gemm_persist_wgs", G)
                for i in range(3):
                    ops.gemm(xs[i % 3], ws[i % 12 if cold else 0], None, "gate_res", gate=gate, res=outs[i % 3], out=outs[i % 3])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(12):
                    ops.gemm(xs[i % 3], ws[i % 12 if cold else 0], None, "gate_res", gate=gate, res=outs[i % 3], out=outs[i % 3])
                e1.record(); torch.cuda.synchronize()
                res[name].append(e0.elapsed_time(e1) / 12)
        fl = 2.0 * M * N * K
        print(f"{M}x{N}x{K} cold={cold}: " + "  ".join(f"{n}: {sorted(t)[len(t)//2]*1e3:.0f}us {fl/sorted(t)[len(t)//2]/1e9:.0f}TF" for n, t in res.items()), flush=True)
knob("gemm_variant", 17); knob("gemm_persist_wgs", 0)
