"""In-process interleaved A/B of the GEMM schedules on MI355X (run on the GPU box from the repo root):

    python tools/microbench/gemm_ab.py [variants, default 15,17] [--bands 4,6,8,16] [--quick]

1. bit-identity of every variant with schedule 15 on ragged / tiny / multi-round shapes and every epilogue, once with the
   production grid and once with an 8-work-group persistent grid (pe_debug_set("gemm_persist_wgs", 8): every work-group of
   schedule 17 then walks many tiles, crosses ragged tiles and problem boundaries);
2. race screen: 200 repeats of one launch per variant must be identical;
3. timing on the four Linear shapes of a DiT block (hot operands, and cold weights: 12 matrices round robin), median of 5
   interleaved rounds; optional sweep of the band height of the tile order.
"""
import sys

sys.path.insert(0, '.')
import torch

from physicedit_amd import ops
from physicedit_amd._lib import lib

BF = torch.bfloat16
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
variants = [int(v) for v in argv[0].split(',')] if argv else [15, 17]
bands = []
for a in sys.argv[1:]:
    if a.startswith("--bands"):
        bands = [int(b) for b in (a.split("=")[1] if "=" in a else sys.argv[sys.argv.index(a) + 1]).split(",")]
quick = "--quick" in sys.argv
g = torch.Generator(device='cuda').manual_seed(0)


def setv(v):
    assert lib().pe_debug_set(b"gemm_variant", v) == 0


def knob(k, v):
    assert lib().pe_debug_set(k.encode(), v) == 0


def rnd(shape, scale=1.0):
    return (torch.randn(shape, generator=g, device='cuda') * scale).to(BF)


ok = True
ref_v = 15
shapes_small = ((300, 3072, 3072), (257, 264, 64), (40, 18432, 3072), (272, 3072, 12288), (1, 3072, 256), (64, 64, 128),
                (4096, 3072, 64), (512, 3072, 3584), (2048, 3072, 3072), (1000, 1288, 192), (8704, 3072, 3072))
for wgs in (0, 8):
    knob("gemm_persist_wgs", wgs)
    for (M, N, K) in shapes_small:
        x, w, bb = rnd((M, K)), rnd((N, K), K ** -0.5), rnd((N,))
        gate, res = rnd((N,), 0.5), rnd((M, N))
        for epi in ("bias", "gelu_sigmoid", "gelu_erf", "silu", "gate_res"):
            kw = dict(gate=gate, res=res) if epi == "gate_res" else {}
            setv(ref_v)
            a = ops.gemm(x, w, bb, epi, **kw)
            for v in variants:
                if v == ref_v:
                    continue
                setv(v)
                c = ops.gemm(x, w, bb, epi, **kw)
                torch.cuda.synchronize()
                eq = torch.equal(a, c)
                ok &= eq
                if not eq:
                    d = (a.float() - c.float()).abs()
                    print(f"wgs={wgs} {(M, N, K)} {epi} v{v}: DIFF max {d.max().item():.4g} count {(d > 0).sum().item()}", flush=True)
            if epi == "gate_res":       # in place (res aliases out), as the block uses it
                setv(ref_v)
                r1 = res.clone(); ops.gemm(x, w, bb, epi, gate=gate, res=r1, out=r1)
                for v in variants:
                    if v == ref_v:
                        continue
                    setv(v)
                    r2 = res.clone(); ops.gemm(x, w, bb, epi, gate=gate, res=r2, out=r2)
                    torch.cuda.synchronize()
                    eq = torch.equal(r1, r2); ok &= eq
                    if not eq:
                        print(f"wgs={wgs} {(M, N, K)} gate_res in place v{v}: DIFF", flush=True)
    # QKV epilogue (per-head RMSNorm + RoPE, transposed V), aligned and unaligned joint offsets
    for (M, seq_off) in ((300, 0), (2048, 0), (37, 135), (520, 4096)):
        H, K = 24, 3072
        x, w, bb = rnd((M, K)), rnd((3 * H * 128, K), K ** -0.5), rnd((3 * H * 128,), 0.1)
        nq, nk_ = rnd((128,)), rnd((128,))
        ang = torch.rand((M, 64), generator=g, device='cuda') * 6.28
        cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
        outs = {}
        for v in [ref_v] + [v for v in variants if v != ref_v]:
            setv(v)
            q, k, vt = ops.alloc_qkv(H, seq_off + M, 'cuda')
            ops.qkv_rmsnorm_rope(x, w, bb, nq, nk_, cos, sin, q, k, vt, seq_off)
            torch.cuda.synchronize()
            outs[v] = (q, k, vt)
        for v in variants:
            if v == ref_v:
                continue
            eq = all(torch.equal(a_, b_) for a_, b_ in zip(outs[ref_v], outs[v])); ok &= eq
            if not eq:
                print(f"wgs={wgs} qkv M={M} off={seq_off} v{v}: DIFF", flush=True)
    # e4m3 operands
    for (M, N, K) in ((300, 3072, 3072), (2048, 3072, 12288), (272, 256, 128)):
        x, w, bb = rnd((M, K)), rnd((N, K), K ** -0.5).to(torch.float8_e4m3fn), rnd((N,))
        xq, sc = ops.quantize_rows_e4m3(x)
        setv(ref_v); a = ops.gemm_e4m3(xq, sc, w, bb, "gelu_sigmoid")
        for v in variants:
            if v == ref_v:
                continue
            setv(v); c = ops.gemm_e4m3(xq, sc, w, bb, "gelu_sigmoid")
            torch.cuda.synchronize()
            eq = torch.equal(a, c); ok &= eq
            if not eq:
                print(f"wgs={wgs} fp8 {(M, N, K)} v{v}: DIFF", flush=True)
    print(f"persist_wgs={wgs}: {'ALL EQUAL' if ok else 'MISMATCH'}", flush=True)
# ---- race screen
for wgs in (0, 8):
    knob("gemm_persist_wgs", wgs)
    for v in variants:
        setv(v)
        x, w = rnd((2048 if wgs else 8704, 3072)), rnd((3072, 3072), 3072 ** -0.5)
        res = rnd((x.shape[0], 3072)); gate = rnd((3072,))
        r0 = ops.gemm(x, w, None, "gate_res", gate=gate, res=res); bad = 0
        for i in range(100 if quick else 200):
            r = ops.gemm(x, w, None, "gate_res", gate=gate, res=res)
            bad += int(not torch.equal(r, r0))
        print(f"race screen v{v} persist_wgs={wgs}: {bad} differ", flush=True)
knob("gemm_persist_wgs", 0)
print("CORRECTNESS " + ("OK" if ok else "FAILED"), flush=True)

# ---- timing
shapes = [(8704, 12288, 3072, "gelu_sigmoid"), (8704, 3072, 12288, "gate_res"), (8704, 9216, 3072, "bias"), (8704, 3072, 3072, "gate_res")]
NW = 12


def time_shape(M, N, K, epi, configs, rounds=5, reps=12):
    xs = [rnd((M, K)) for _ in range(3)]
    ws = [rnd((N, K), K ** -0.5) for _ in range(NW)]
    b = rnd((N,)); gate = rnd((N,), 0.5)
    outs = [rnd((M, N)) for _ in range(3)]
    fl = 2.0 * M * N * K
    for cold in (0, 1):
        res = {c: [] for c in configs}
        for rnd_i in range(rounds):
            for c in configs:
                v, band = c
                setv(v); knob("gemm_band", band)
                kw = dict(gate=gate) if epi == "gate_res" else {}
                def run(i):
                    j = i % NW if cold else 0
                    k = i % 3 if cold else 0
                    if epi == "gate_res":
                        ops.gemm(xs[k], ws[j], b, epi, gate=gate, res=outs[k], out=outs[k])
                    else:
                        ops.gemm(xs[k], ws[j], b, epi, out=outs[k])
                run(0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(reps):
                    run(i)
                e1.record(); torch.cuda.synchronize()
                res[c].append(e0.elapsed_time(e1) / reps)
        print(f"{M}x{N}x{K} {epi} cold={cold}: " + "  ".join(
            f"v{c[0]}/b{c[1]}: {sorted(t)[len(t)//2]*1e3:.0f}us {fl/sorted(t)[len(t)//2]/1e9:.0f} TF (best {fl/min(t)/1e9:.0f})" for c, t in res.items()), flush=True)
    knob("gemm_band", 4)


for (M, N, K, epi) in shapes:
    time_shape(M, N, K, epi, [(v, 4) for v in variants])
if bands:
    for (M, N, K, epi) in shapes:
        time_shape(M, N, K, epi, [(variants[-1], b) for b in bands], rounds=3)
setv(0)
