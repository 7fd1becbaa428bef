"""Stream-K (schedule 19) against the round-3 rule (schedule 17 from three rounds of tiles on, 15 below) on the four Linear shapes of
a DiT block, MI355X, N(0,1) operands: bit-identity, then interleaved timing (hot operands; cold weights: 12 matrices round robin).

    python tools/microbench/gemm_sk.py
"""
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


def setv(v):
    assert lib().pe_debug_set(b"gemm_variant", v) == 0


for (M, N, K, epi, name) in ((8704, 3072, 3072, "gate_res", "out-proj"), (8704, 3072, 12288, "gate_res", "MLP-down"),
                             (8704, 9216, 3072, "bias", "QKV (bias)"), (8704, 12288, 3072, "gelu_sigmoid", "MLP-up")):
    x, bb = rnd((M, K)), rnd((N,))
    wts = [rnd((N, K), K ** -0.5) for _ in range(12)]
    gate, res = rnd((N,), 0.5), rnd((M, N))
    kw = dict(gate=gate, res=res) if epi == "gate_res" else {}
    out = torch.empty((M, N), dtype=BF, device="cuda")
    setv(15); a = ops.gemm(x, wts[0], bb, epi, **kw).clone()
    setv(19); c = ops.gemm(x, wts[0], bb, epi, **kw).clone()
    torch.cuda.synchronize()
    same = torch.equal(a, c)
    res_t = {}
    for mode in ("hot", "cold"):
        for v in (17, 19):
            res_t[(mode, v)] = []
        for rnd_i in range(5):
            for v in (17, 19):
                setv(v)
                lib().pe_debug_set(b"gemm_sk", 0 if v == 17 else 1)
                ops.gemm(x, wts[0], bb, epi, out=out, **kw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(12):
                    ops.gemm(x, wts[i if mode == "cold" else 0], bb, epi, out=out, **kw)
                e1.record(); torch.cuda.synchronize()
                res_t[(mode, v)].append(e0.elapsed_time(e1) / 12)
    fl = 2.0 * M * N * K
    med = {k: sorted(v)[len(v) // 2] for k, v in res_t.items()}
    print(f"{name} {M}x{N}x{K}: bit-identical {same};  " + "  ".join(
        f"{mode} r3-rule {med[(mode, 17)]*1e3:.0f} us {fl/med[(mode, 17)]/1e9:.0f} TF -> stream-K {med[(mode, 19)]*1e3:.0f} us {fl/med[(mode, 19)]/1e9:.0f} TF "
        f"({(med[(mode, 17)]/med[(mode, 19)]-1)*100:+.1f} %)" for mode in ("hot", "cold")), flush=True)
assert torch.count_nonzero(ws[:4096]).item() == 0, "workspace not at rest"
lib().pe_debug_set_ptr(b"gemm_workspace", None)
setv(17); lib().pe_debug_set(b"gemm_sk", 1)
