"""s_memtime anatomy of the one-wave-per-SIMD attention kernels (variants 4 / 5) at S = 8704: per work item prologue / loop / epilogue
ticks, per KV iteration, and phase 1 / wait + barrier / phase 2 of iteration 8.  Needs a library built with -DPE_W4_STAMPS=1 for the
per-phase stamps (tools/microbench/attn_knobs.sh builds one when W4_STAMPS=1 is in its knob list):

    PE_LIB_PATH=$PWD/build_ab/libpe_stamps.so python tools/microbench/attn_stamps.py [variants, default 5,4]
"""
import sys

sys.path.insert(0, '.')
import numpy as np
import torch

from physicedit_amd import ops
from physicedit_amd._lib import lib

BF = torch.bfloat16
g = torch.Generator(device='cuda').manual_seed(0)
S, H = 8704, 24
q, k, vt = ops.alloc_qkv(H, S, "cuda")
for t in (q, k, vt):
    t.copy_(torch.randn(t.shape, generator=g, device='cuda').to(BF))
vt[:, :, S:] = 0
qc = (q.float() * (0.08838834764831845 * 1.4426950408889634)).to(BF)
for var in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "5,4").split(",")]:
    lib().pe_debug_set(b"attn_variant", var)
    qq, pre = (qc, True) if var >= 5 else (q, False)
    st = torch.zeros((2048, 10), dtype=torch.int64, device='cuda')
    lib().pe_debug_set_ptr(b"attn_stamps", st.data_ptr())
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.flash_attn(qq, k, vt, S, q_prescaled=pre); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    lib().pe_debug_set_ptr(b"attn_stamps", None)
    s = st.cpu().numpy().astype(np.float64)
    s = s[s[:, 0] > 0]
    full = s[s[:, 8] >= 100]
    span = s[:, 3].max() - s[:, 0].min()
    print(f"variant {var}: {ms*1e3:.0f} us, {len(s)} WGs ({len(full)} whole items); span {span:.0f} ticks -> {span/ms/1e3:.0f} MHz")
    pro = (full[:, 1] - full[:, 0]).mean(); loop = (full[:, 2] - full[:, 1]); epi = (full[:, 3] - full[:, 2]).mean()
    print(f"  whole items: prologue {pro:.0f}  loop {loop.mean():.0f} = {(loop / (np.ceil(full[:, 8] / 4) * 4)).mean():.0f} per iteration (matrix pipe: 2048)  epilogue {epi:.0f}")
    it = full[full[:, 4] > 0]
    if len(it):
        print(f"  iteration 8: phase 1 {(it[:, 5]-it[:, 4]).mean():.0f}  wait + barrier {(it[:, 6]-it[:, 5]).mean():.0f}  phase 2 {(it[:, 7]-it[:, 6]).mean():.0f}")
lib().pe_debug_set(b"attn_variant", 5)
