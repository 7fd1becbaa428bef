"""Parity bookkeeping shared by the `-m gpu` end-to-end tests: every comparison of the HIP path with the oracle is
reduced to the numbers the north-star sentence asks for (fraction of elements within 1e-3, max |d|, distance to an fp32
evaluation of the same graph next to the reference-bf16's own distance) and appended to gpurun_out/parity_r06.json
(scratch, never a tracked file), keyed by BASELINE.json config.  A copy is committed by hand as profiles/r06_parity.json."""
import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "parity_r06.json")


def parity_stats(got: torch.Tensor, ref: torch.Tensor, ref32: torch.Tensor = None) -> dict:
    g, r = got.detach().float().cpu(), ref.detach().float().cpu()
    d = (g - r).abs()
    st = {
        "n": int(d.numel()),
        "frac_within_1e-3": float((d <= 1e-3).float().mean()),
        "frac_bit_identical": float((d == 0).float().mean()),
        "max_abs_diff": float(d.max()),
        "mean_abs_diff": float(d.mean()),
        "ref_abs_mean": float(r.abs().mean()),
    }
    # bf16 ulp of the reference value, floored at the tensor's own mean magnitude: an end-to-end output near zero is the
    # difference of O(mean) terms, its absolute error lives at that scale
    ulp = torch.clamp(r.abs(), min=max(st["ref_abs_mean"], 2.0 ** -9)).log2().floor().exp2() * 2.0 ** -7
    st["max_ulp"] = float((d / ulp).max())
    if ref32 is not None:
        r32 = ref32.detach().float().cpu()
        e_hip = (g - r32).pow(2).mean().sqrt().item()
        e_ref = (r - r32).pow(2).mean().sqrt().item()
        st["rms_to_fp32_hip"] = e_hip
        st["rms_to_fp32_reference_bf16"] = e_ref
        st["fp32_distance_ratio"] = e_hip / max(e_ref, 1e-30)
        st["max_to_fp32_hip"] = float((g - r32).abs().max())
        st["max_to_fp32_reference_bf16"] = float((r - r32).abs().max())
    return st


def record(config: str, case: str, got, ref, ref32=None, **extra) -> dict:
    st = parity_stats(got, ref, ref32)
    st.update(extra)
    print(f"[parity:{config}] {case}: " + ", ".join(f"{k}={v:.4g}" if isinstance(v, float) else f"{k}={v}" for k, v in st.items()))
    try:
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        data = json.load(open(OUT)) if os.path.exists(OUT) else {}
        data.setdefault(config, {})[case] = st
        with open(OUT, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
    except OSError:
        pass
    return st
