"""CPU (`-m "not gpu"`) checks of the boundary and the host logic: the C-ABI library builds for
gfx950, loads, and exports exactly the symbols include/physicedit_amd.h declares; host scalar code
(scheduler, sinusoid, alpha, RoPE tables) equals the oracle bit for bit.  No kernel is launched."""
import os
import re

import pytest
import torch

import oracle.physicedit_oracle as O
from physicedit_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BF = torch.bfloat16


@pytest.fixture(scope="module")
def built_lib():
    from physicedit_amd import build, _lib
    try:
        build.build(verbose=False)
    except RuntimeError as e:
        if not os.path.exists(_lib.LIB_PATH):
            pytest.fail(f"HIP library could not be built and no prebuilt one exists: {e}")
    return _lib.lib()


def test_header_symbols_exported(built_lib):
    from physicedit_amd import _lib
    header = open(os.path.join(ROOT, "include", "physicedit_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(pe_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    for name in declared:
        assert hasattr(built_lib, name), name
    from physicedit_amd import _lib as _l
    assert built_lib.pe_abi_version() == _l.ABI_VERSION == 8
    from physicedit_amd import build as _b
    assert built_lib.pe_build_id().decode() == _b.source_hash()
    assert built_lib.pe_last_error() == b""


def test_invalid_args_fail_loudly_without_gpu(built_lib):
    """Argument validation happens before any launch: errors are codes + message, not crashes."""
    from physicedit_amd import _lib
    rc = built_lib.pe_gemm_bf16(0, None, 64, None, None, None, 64, 8, 8, 64, None, None, 0, None)
    assert rc == -1 and b"null operand" in built_lib.pe_last_error()
    rc = built_lib.pe_gemm_bf16(0, 1, 100, 1, None, 1, 64, 8, 8, 100, None, None, 0, None)
    assert rc == -1 and b"multiple of 64" in built_lib.pe_last_error()
    rc = built_lib.pe_flash_attn(1, 1, 1, 1, 24, 100, 100, 3072, 0.1, None, 0, None)
    assert rc == -1 and b"S_pad" in built_lib.pe_last_error()
    assert built_lib.pe_dit_workspace_bytes(None, 10, 10, 1) == 0
    # e4m3 entry points
    rc = built_lib.pe_quantize_rows_e4m3(1, 64, 4, 64, 1, 64, 1, None)
    assert rc == -1 and b"multiple of 128" in built_lib.pe_last_error()
    rc = built_lib.pe_quantize_rows_e4m3(1, 64, 4, 60, 1, 128, 1, None)
    assert rc == -1 and b"multiple of 8" in built_lib.pe_last_error()
    rc = built_lib.pe_gemm_e4m3(0, 1, 64, 1, 1, None, None, 0, 1, 64, 8, 8, 64, None, None, 0, None)
    assert rc == -1 and b"multiple of 128" in built_lib.pe_last_error()
    rc = built_lib.pe_gemm_e4m3(0, 1, 128, None, 1, None, None, 0, 1, 64, 8, 8, 128, None, None, 0, None)
    assert rc == -1 and b"scale_a" in built_lib.pe_last_error()
    rc = built_lib.pe_ln_modulate_e4m3(1, None, None, None, 4, 3072, 4, 1, 1, None, None, 1e-6, None)
    assert rc == -1 and b"null e4m3 output" in built_lib.pe_last_error()
    with pytest.raises(_lib.PeError):
        _lib.check(rc, "x")
    # knobs are validated where they are set: a schedule that does not exist fails at the call site, 0 = the compiled default
    assert built_lib.pe_debug_set(b"gemm_variant", 10) == -1 and b"does not exist" in built_lib.pe_last_error()
    assert built_lib.pe_debug_set(b"gemm_variant", 0) == 0 and built_lib.pe_debug_set(b"gemm_variant", 21) == 0
    assert built_lib.pe_debug_set(b"decode_layer_wgs_per_cu", 0) == -1 and built_lib.pe_debug_set(b"no_such_knob", 1) == -1
    # round 6 entry points refuse what they cannot run, before any launch
    assert built_lib.pe_decode_layer_scratch_bytes(28, 1024, 18944) > 8192 and built_lib.pe_decode_layer_scratch_bytes(0, 1024, 18944) == 0
    assert built_lib.pe_vae_attention_scratch_bytes(16384) > 384 * 16384 * 2 and built_lib.pe_vae_attention_scratch_bytes(0) == 0
    rc = built_lib.pe_vae_attention(256, 257, 256, 64, None)
    assert rc == -1 and b"256-byte aligned" in built_lib.pe_last_error()


def test_no_cpu_fallback():
    from physicedit_amd import ops
    with pytest.raises(ValueError):
        ops.gemm(torch.zeros((8, 64), dtype=BF), torch.zeros((8, 64), dtype=BF))
    from physicedit_amd._lib import PeError
    from physicedit_amd.dit import QwenImageDiTEngine
    with pytest.raises(PeError):
        QwenImageDiTEngine({}, None, device="cpu")


def test_product_code_never_imports_oracle():
    pkg = os.path.join(ROOT, "physicedit_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_scheduler_matches_oracle_and_golden(golden):
    from physicedit_amd.scheduler import qwen_image_scheduler
    g = golden("G1_scheduler")
    sch = qwen_image_scheduler()
    assert torch.equal(sch.timesteps, g["default_timesteps"])
    for steps, S0 in ((4, 1024), (4, 64), (40, 4096), (50, 6889)):
        sch.set_timesteps(steps, dynamic_shift_len=S0)
        assert torch.equal(sch.sigmas, g[f"sigmas_{steps}_{S0}"])
        assert torch.equal(sch.timesteps, g[f"timesteps_{steps}_{S0}"])
        gen = torch.Generator().manual_seed(11)
        x = torch.randn((1, 16, 8, 8), generator=gen).to(BF)
        v = torch.randn((1, 16, 8, 8), generator=gen).to(BF)
        outs = torch.stack([sch.step(v, sch.timesteps[i], x) for i in range(steps)])
        assert torch.equal(outs, g[f"step_{steps}_{S0}"])
        tab = O.FlowMatchTables(steps, dynamic_shift_len=S0)
        for i in range(steps):   # the scalar the GPU Euler kernel receives == the oracle's fp32 (sigma' - sigma)
            sigma = tab.sigmas[i]
            nxt = tab.sigmas[i + 1] if i + 1 < steps else 0
            assert sch.dsigma(i) == float((nxt - sigma).item())


def test_host_scalars_match_oracle(golden):
    from physicedit_amd.scheduler import adapter_alpha, timestep_sinusoid, qwen_image_scheduler
    g = golden("G2_time_embed")
    sch = qwen_image_scheduler()
    sch.set_timesteps(40, dynamic_shift_len=4096)
    ts = sch.timesteps.to(BF)
    assert torch.equal(timestep_sinusoid(ts / 1000), g["sinusoid_f32"])
    t_min, t_max = O.adapter_t_range()
    g9 = golden("G9_adapter")
    for tv in (1000.0, 748.0, 300.0, 20.0):
        t = torch.tensor([tv]).to(BF)
        a, oma = adapter_alpha(t, t_min, t_max)
        assert a == g9[f"alpha_{int(tv)}"].item()
        av = O.adapter_alpha(t, t_min, t_max).to(BF)
        assert oma == (1 - av).float().item()


def test_rope_tables_match_oracle():
    from physicedit_amd.rope import rope_cos_sin
    for shapes, T in (([(1, 8, 8), (1, 6, 10)], 37), ([(1, 64, 64), (1, 64, 64)], 512), ([(1, 83, 83), (1, 64, 64)], 272)):
        ci, si, ct, st = rope_cos_sin(shapes, T)
        vid, txt = O.rope_tables(shapes, T)
        assert torch.equal(ci, vid.real) and torch.equal(si, vid.imag)
        assert torch.equal(ct, txt.real) and torch.equal(st, txt.imag)


def test_lora_name_mapping_matches_oracle():
    """Key -> module-name mapping of GeneralLoRALoader.get_name_dict (lora/__init__.py:11-25)."""
    lora = synth.make_lora(1, 1, 4)
    sd = {k: torch.zeros(s, dtype=BF) for k, s in synth.dit_layout(1)}
    assert O.lora_merge(sd, lora) == len(synth.LORA_TARGETS) == 12


def test_struct_layouts_match_the_header(tmp_path):
    """The ctypes mirrors of the header's structs must have the C compiler's layout (a field missing from the ctypes side is
    silently accepted by Python and read as garbage by the library)."""
    import ctypes as C
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    from physicedit_amd import _lib as L
    fields = {"pe_dit_call": (L.DitCall, ["latents", "n_edit", "prompt_emb", "special_idx", "alpha", "rope_cos_img", "step",
                                          "noise_pred", "n_control", "control", "attn_words"]),
              "pe_control_input": (L.ControlInput, ["blocks", "conditioning", "scale"]),
              "pe_controlnet_block": (L.ControlNetBlock, ["x_rms_w", "out_b"]),
              "pe_dit_weights": (L.DitWeights, ["num_layers", "blocks", "weights_e4m3"]),
              "pe_adapter_weights": (L.AdapterWeights, ["dino_w0", "vae_b2"]),
              "pe_decode_layer_weights": (L.DecodeLayerWeights, ["q_w", "post_norm_w", "input_norm_eps", "post_norm_eps", "n_q_heads", "ff"])}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "physicedit_amd.h"', "int main(void) {"]
    for cname, (_, names) in fields.items():
        lines.append(f'  printf("{cname} %zu", sizeof({cname}));')
        for n in names:
            lines.append(f'  printf(" %zu", offsetof({cname}, {n}));')
        lines.append('  printf("\\n");')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    for line in out:
        cname, size, *offs = line.split()
        cls, names = fields[cname]
        assert C.sizeof(cls) == int(size), (cname, C.sizeof(cls), size)
        assert [getattr(cls, n).offset for n in names] == [int(o) for o in offs], cname


def test_asm_mfma_hazards_of_the_e4m3_attention_kernel():
    """flash_attn_fp8p_kernel and flash_attn_fp8w_kernel place their MFMAs as asm statements, so the compiler neither keeps their source
    registers alive while the matrix pipe reads them nor waits before it touches their results (both happened:
    profiles/r04_attention_notes.md section 5.2); the one-wave kernel also keeps O, Q and L in fixed accumulator registers that only its
    asm statements may name.  tools/mfma_asm_hazards.py compiles the file to gfx950 assembly and checks the rules statically on all four
    instantiations: a compiler upgrade or an edit that re-opens one of them fails here, without a GPU."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("mfma_asm_hazards", os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools", "mfma_asm_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main() == 0


def test_hashed_weights_checksum():
    """`synth.make_state_dict_hashed` is what ties the 60-layer fixtures G21-G24 (written by the reference in the build container) to the
    model the GPU box regenerates: pure integer arithmetic, so its bits are a constant of the repository.  One block + the embeds."""
    import hashlib
    import torch
    from physicedit_amd import synth
    sd = synth.make_state_dict_hashed(synth.dit_layout(1), 1234)
    hh = hashlib.sha256()
    for k in sorted(sd):
        hh.update(sd[k].view(torch.int16).numpy().tobytes())
    assert hh.hexdigest()[:16] == "ea475a4c871df204"
    w = sd["transformer_blocks.0.img_mlp.net.0.proj.weight"].float()
    assert abs(w.std().item() * (3 * 3072) ** 0.5 - 1.0) < 1e-2 and abs(w.mean().item()) < 1e-5      # U(+-1/sqrt(fan_in))
    u = synth.hash_uniform(7, 1 << 16, "cpu", chunk=1000)           # chunking does not change the stream
    assert torch.equal(u, synth.hash_uniform(7, 1 << 16, "cpu"))
