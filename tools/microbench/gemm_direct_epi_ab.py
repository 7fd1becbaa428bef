"""The LDS-free ("direct") epilogue of complete tiles against the LDS epilogue (knob gemm_direct_epilogue), schedules 15 / 17 / 22, bf16 and
e4m3: bit-identity on the block's shapes and epilogues, then interleaved timing.  GPU box, repo root:  python tools/microbench/gemm_direct_epi_ab.py"""
import sys

sys.path.insert(0, '.')
import torch

from physicedit_amd import ops
from physicedit_amd._lib import lib

BF = torch.bfloat16
g = torch.Generator(device='cuda').manual_seed(0)


def rnd(shape, scale=1.0):
    return (torch.randn(shape, generator=g, device='cuda') * scale).to(BF)


def knob(k, v):
    assert lib().pe_debug_set(k.encode(), v) == 0


ok = True
for (M, N, K) in ((8704, 3072, 3072), (2100, 12288, 3072), (600, 3072, 12288), (8464, 3072, 3072)):
    x, w, b, gate, res = rnd((M, K)), rnd((N, K), K ** -0.5), rnd((N,)), rnd((N,), 0.5), rnd((M, N))
    xq, sc = ops.quantize_rows_e4m3(x)
    w8 = w.to(torch.float8_e4m3fn)
    for epi in ("bias", "gelu_sigmoid", "gate_res"):
        for v in (15, 17, 22):
            knob("gemm_variant", v)
            outs = []
            for d in (0, 1):
                knob("gemm_direct_epilogue", d)
                kw = dict(gate=gate, res=res) if epi == "gate_res" else {}
                a = ops.gemm(x, w, b, epi, **kw)
                a2 = ops.gemm(x, w, None, epi, **kw)
                r1 = res.clone()
                if epi == "gate_res":
                    ops.gemm(x, w, b, epi, gate=gate, res=r1, out=r1)          # in place, as the block uses it
                    a3 = ops.gemm(x, w, b, epi, gate=None, res=res)            # scalar gate
                else:
                    a3 = a
                f8 = ops.gemm_e4m3(xq, sc, w8, b, epi, **kw)
                outs.append((a, a2, r1, a3, f8))
            torch.cuda.synchronize()
            eq = all(torch.equal(p, q) for p, q in zip(*outs))
            ok &= eq
            if not eq:
                print(f"DIFF {(M, N, K)} {epi} v{v}: " + " ".join(str(int((p != q).sum())) for p, q in zip(*outs)), flush=True)
# QKV epilogue (keeps the LDS form: a direct q / k form was bit-identical and 5.6 % slower, and its registers pushed spills into the K loop): aligned and unaligned joint offsets, scaled Q, e4m3 too
for (M, seq_off) in ((8704, 0), (2300, 0), (520, 4096), (300, 135)):
    H, K = 24, 3072
    x, w, bb = rnd((M, K)), rnd((3 * H * 128, K), K ** -0.5), rnd((3 * H * 128,), 0.1)
    nq, nk_ = rnd((128,)), rnd((128,))
    ang = torch.rand((M, 64), generator=g, device='cuda') * 6.28
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    for v in (15, 17, 22):
        knob("gemm_variant", v)
        outs = []
        for d in (0, 1):
            knob("gemm_direct_epilogue", d)
            q, k, vt = ops.alloc_qkv(H, seq_off + M, 'cuda')
            ops.qkv_rmsnorm_rope(x, w, bb, nq, nk_, cos, sin, q, k, vt, seq_off, q_scale=0.1275)
            outs.append((q, k, vt))
        torch.cuda.synchronize()
        eq = all(torch.equal(p_, q_) for p_, q_ in zip(*outs))
        ok &= eq
        if not eq:
            print(f"DIFF qkv M={M} off={seq_off} v{v}: " + " ".join(str(int((p_ != q_).sum())) for p_, q_ in zip(*outs)), flush=True)
print("direct epilogue bit-identical to the LDS epilogue:", ok, flush=True)
knob("gemm_variant", 17)
M, H, K = 8704, 24, 3072
x, w, bb = rnd((M, K)), rnd((3 * H * 128, K), K ** -0.5), rnd((3 * H * 128,), 0.1)
nq, nk_ = rnd((128,)), rnd((128,))
ang = torch.rand((M, 64), generator=g, device='cuda') * 6.28
cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
q, k, vt = ops.alloc_qkv(H, M, 'cuda')
t = {0: [], 1: []}
fl = 2.0 * M * 9216 * K
for _ in range(7):
    for d in (0, 1):
        knob("gemm_direct_epilogue", d)
        ops.qkv_rmsnorm_rope(x, w, bb, nq, nk_, cos, sin, q, k, vt, 0, q_scale=0.1275)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(12):
            ops.qkv_rmsnorm_rope(x, w, bb, nq, nk_, cos, sin, q, k, vt, 0, q_scale=0.1275)
        e1.record(); torch.cuda.synchronize()
        t[d].append(e0.elapsed_time(e1) / 12)
m0, m1 = sorted(t[0])[3], sorted(t[1])[3]
print(f"8704x9216x3072 QKV (RMSNorm + RoPE + V^T): LDS epilogue {m0*1e3:.1f} us {fl/m0/1e9:.0f} TF   knob on {m1*1e3:.1f} us {fl/m1/1e9:.0f} TF  ({(m1/m0-1)*100:+.2f} %)", flush=True)
knob("gemm_variant", 17)
for fp8 in (False, True):
    for (M, N, K, epi) in [(8704, 12288, 3072, "gelu_sigmoid"), (8704, 3072, 12288, "gate_res"), (8704, 9216, 3072, "bias"), (8704, 3072, 3072, "gate_res")]:
        x, w, b, gate = rnd((M, K)), rnd((N, K), K ** -0.5), rnd((N,)), rnd((N,), 0.5)
        out = rnd((M, N))
        xq, sc = ops.quantize_rows_e4m3(x)
        w8 = w.to(torch.float8_e4m3fn)
        fl = 2.0 * M * N * K
        reps = max(8, int(2e15 / fl / 100))
        t = {0: [], 1: []}
        for _ in range(7):
            for d in (0, 1):
                knob("gemm_direct_epilogue", d)
                def run():
                    kw = dict(gate=gate, res=out) if epi == "gate_res" else {}
                    if fp8:
                        ops.gemm_e4m3(xq, sc, w8, b, epi, out=out, **kw)
                    else:
                        ops.gemm(x, w, b, epi, out=out, **kw)
                run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    run()
                e1.record(); torch.cuda.synchronize()
                t[d].append(e0.elapsed_time(e1) / reps)
        m0, m1 = sorted(t[0])[3], sorted(t[1])[3]
        print(f"{'e4m3 ' if fp8 else ''}{M}x{N}x{K} {epi}: LDS epilogue {m0*1e3:.1f} us {fl/m0/1e9:.0f} TF   direct {m1*1e3:.1f} us {fl/m1/1e9:.0f} TF  ({(m1/m0-1)*100:+.2f} %)", flush=True)
knob("gemm_direct_epilogue", 1)
