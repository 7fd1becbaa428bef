"""Device-resident Qwen-Image DiT + visual-thinking adapter, driven through the C-ABI composite.

Host-side mirror of the reference's operator for this path:
    model_fn_qwen_image(dit=, visual_thinking_adapter=, latents=, timestep=, prompt_emb=, ...)
        DiffSynth-Studio/diffsynth/pipelines/qwen_image_physical.py:1302-1403
Weights keep the reference's state-dict names; `params[name]` are torch
tensors on the GPU.  to_q/to_k/to_v (and add_*_proj) live as row-slices (views) of one fused
[9216,3072] buffer so the QKV projection is one GEMM, while LoRA merging and `load_state_dict`
still address them by their original names.

No arithmetic happens in this file except host scalar set-up; all tensor math is HIP kernels.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib, ops
from ._lib import AdapterWeights, DitBlockLora, DitBlockWeights, DitCall, DitWeights, check, lib, stream_ptr
from .rope import RopeCache
from .scheduler import adapter_alpha, qwen_image_scheduler, timestep_sinusoid

BF = torch.bfloat16
F8 = torch.float8_e4m3fn
D = 3072


def count_layers(sd: Dict[str, torch.Tensor]) -> int:
    n = 0
    while f"transformer_blocks.{n}.img_mod.1.weight" in sd:
        n += 1
    return n


class QwenImageDiTEngine:
    """Owns the device weights, the workspace and the prepared per-timestep tables."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], adapter_state_dict: Optional[Dict[str, torch.Tensor]] = None,
                 device="cuda"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.PeError("QwenImageDiTEngine needs a HIP device: physicedit_amd has no CPU path")
        lib()  # fail loudly now if the .so is missing
        self.num_layers = count_layers(state_dict)
        self.fp8 = False
        self.params: Dict[str, torch.Tensor] = {}
        self._fused: Dict[str, torch.Tensor] = {}
        self._ingest_dit(state_dict)
        self.adapter: Optional[Dict[str, torch.Tensor]] = None
        if adapter_state_dict is not None:
            self.adapter = {k: v.to(device=self.device, dtype=BF).contiguous() for k, v in adapter_state_dict.items()}
        sch = qwen_image_scheduler()
        # VisualThinkingDualAdapter(t_min=..., t_max=...) -- qwen_image_physical.py:225
        self.t_min, self.t_max = sch.timesteps.min().item(), sch.timesteps.max().item()
        self._handle = C.c_void_p()
        self._keep = None
        self._create()
        self._ws: Optional[torch.Tensor] = None
        self._bound = (0, 0, 0)
        self._step_of: Dict[float, int] = {}
        self.version = 0          # bumped whenever weights / LoRA state change: forks made from an older version are stale
        self.rope = RopeCache(self.device)

    # ------------------------------------------------------------------------------------------
    def _ingest_dit(self, sd):
        dev = self.device
        for i in range(self.num_layers):
            p = f"transformer_blocks.{i}.attn."
            for fused_name, parts in ((p + "img_qkv", ("to_q", "to_k", "to_v")),
                                      (p + "txt_qkv", ("add_q_proj", "add_k_proj", "add_v_proj"))):
                w = torch.empty((3 * D, D), dtype=BF, device=dev)
                b = torch.empty((3 * D,), dtype=BF, device=dev)
                for j, part in enumerate(parts):
                    w[j * D:(j + 1) * D].copy_(sd[p + part + ".weight"])
                    b[j * D:(j + 1) * D].copy_(sd[p + part + ".bias"])
                    self.params[p + part + ".weight"] = w[j * D:(j + 1) * D]
                    self.params[p + part + ".bias"] = b[j * D:(j + 1) * D]
                self._fused[fused_name + ".weight"] = w
                self._fused[fused_name + ".bias"] = b
        for k, v in sd.items():
            if k not in self.params:
                self.params[k] = v.to(device=dev, dtype=BF).contiguous()

    _QKV_PARTS = (("img_qkv", ("to_q", "to_k", "to_v")), ("txt_qkv", ("add_q_proj", "add_k_proj", "add_v_proj")))

    def enable_fp8_computation(self):
        """The state the reference reaches with ModelConfig(offload_dtype=torch.float8_e4m3fn) +
        enable_vram_management(enable_dit_fp8_computation=True) (qwen_image_physical.py:440-496): every DiT parameter
        is stored as float8_e4m3fn, every torch.nn.Linear runs AutoWrappedLinear.fp8_linear
        (vram_management/layers.py:115-151) and the RMSNorm weights are cast back to bf16 for their computation.
        Conversion of the stored weights is a one-off dtype cast at load time (torch device op); the per-call
        activation quantisation and the e4m3 GEMMs are HIP kernels.  A merged LoRA must be loaded BEFORE this call
        (the merge is a bf16 operation); hot LoRA may come before or after and stays bf16."""
        if self.fp8:
            return
        for name in list(self._fused):
            t = self._fused[name]
            self._fused[name] = t.to(F8) if name.endswith(".weight") else t.to(F8).to(BF)
        fused_views = set()
        for i in range(self.num_layers):
            p = f"transformer_blocks.{i}.attn."
            for fused, parts in self._QKV_PARTS:
                for j, part in enumerate(parts):
                    for kind in ("weight", "bias"):
                        self.params[p + part + "." + kind] = self._fused[p + fused + "." + kind][j * D:(j + 1) * D]
                        fused_views.add(p + part + "." + kind)
        linear = {k[:-len(".weight")] for k, v in self.params.items() if k.endswith(".weight") and v.dim() == 2}
        for k in list(self.params):
            if k in fused_views:
                continue
            v = self.params[k]
            if k.endswith(".weight") and k[:-len(".weight")] in linear:
                q = v.to(F8)
                if q.shape[1] % 128:       # img_in [3072,64]: the e4m3 GEMM's K granule is 128; zero columns add exact zeros
                    qp = torch.zeros((q.shape[0], (q.shape[1] + 127) // 128 * 128), dtype=F8, device=self.device)
                    qp[:, :q.shape[1]] = q
                    q = qp
                self.params[k] = q.contiguous()
            else:
                self.params[k] = v.to(F8).to(BF)
        self.fp8 = True
        self._create()
        self._apply_hot()
        self._ws, self._bound, self._step_of = None, (0, 0, 0), {}

    def _create(self):
        P, F = self.params, self._fused
        blocks = (DitBlockWeights * max(self.num_layers, 1))()
        for i in range(self.num_layers):
            p = f"transformer_blocks.{i}."
            b = blocks[i]
            b.img_mod_w, b.img_mod_b = P[p + "img_mod.1.weight"].data_ptr(), P[p + "img_mod.1.bias"].data_ptr()
            b.img_qkv_w, b.img_qkv_b = F[p + "attn.img_qkv.weight"].data_ptr(), F[p + "attn.img_qkv.bias"].data_ptr()
            b.norm_q_w, b.norm_k_w = P[p + "attn.norm_q.weight"].data_ptr(), P[p + "attn.norm_k.weight"].data_ptr()
            b.img_out_w, b.img_out_b = P[p + "attn.to_out.0.weight"].data_ptr(), P[p + "attn.to_out.0.bias"].data_ptr()
            b.img_mlp_up_w, b.img_mlp_up_b = P[p + "img_mlp.net.0.proj.weight"].data_ptr(), P[p + "img_mlp.net.0.proj.bias"].data_ptr()
            b.img_mlp_down_w, b.img_mlp_down_b = P[p + "img_mlp.net.2.weight"].data_ptr(), P[p + "img_mlp.net.2.bias"].data_ptr()
            b.txt_mod_w, b.txt_mod_b = P[p + "txt_mod.1.weight"].data_ptr(), P[p + "txt_mod.1.bias"].data_ptr()
            b.txt_qkv_w, b.txt_qkv_b = F[p + "attn.txt_qkv.weight"].data_ptr(), F[p + "attn.txt_qkv.bias"].data_ptr()
            b.norm_added_q_w, b.norm_added_k_w = P[p + "attn.norm_added_q.weight"].data_ptr(), P[p + "attn.norm_added_k.weight"].data_ptr()
            b.txt_out_w, b.txt_out_b = P[p + "attn.to_add_out.weight"].data_ptr(), P[p + "attn.to_add_out.bias"].data_ptr()
            b.txt_mlp_up_w, b.txt_mlp_up_b = P[p + "txt_mlp.net.0.proj.weight"].data_ptr(), P[p + "txt_mlp.net.0.proj.bias"].data_ptr()
            b.txt_mlp_down_w, b.txt_mlp_down_b = P[p + "txt_mlp.net.2.weight"].data_ptr(), P[p + "txt_mlp.net.2.bias"].data_ptr()
        w = DitWeights()
        w.num_layers = self.num_layers
        t = "time_text_embed.timestep_embedder."
        w.time_w1, w.time_b1 = P[t + "linear_1.weight"].data_ptr(), P[t + "linear_1.bias"].data_ptr()
        w.time_w2, w.time_b2 = P[t + "linear_2.weight"].data_ptr(), P[t + "linear_2.bias"].data_ptr()
        w.txt_norm_w = P["txt_norm.weight"].data_ptr()
        w.img_in_w, w.img_in_b = P["img_in.weight"].data_ptr(), P["img_in.bias"].data_ptr()
        w.txt_in_w, w.txt_in_b = P["txt_in.weight"].data_ptr(), P["txt_in.bias"].data_ptr()
        w.norm_out_w, w.norm_out_b = P["norm_out.linear.weight"].data_ptr(), P["norm_out.linear.bias"].data_ptr()
        w.proj_out_w, w.proj_out_b = P["proj_out.weight"].data_ptr(), P["proj_out.bias"].data_ptr()
        w.blocks = blocks
        w.weights_e4m3 = 1 if self.fp8 else 0
        adp = None
        if self.adapter is not None:
            a = AdapterWeights()
            A = self.adapter
            a.dino_w0, a.dino_b0 = A["head_dino.0.weight"].data_ptr(), A["head_dino.0.bias"].data_ptr()
            a.dino_w2, a.dino_b2 = A["head_dino.2.weight"].data_ptr(), A["head_dino.2.bias"].data_ptr()
            a.vae_w0, a.vae_b0 = A["head_vae.0.weight"].data_ptr(), A["head_vae.0.bias"].data_ptr()
            a.vae_w2, a.vae_b2 = A["head_vae.2.weight"].data_ptr(), A["head_vae.2.bias"].data_ptr()
            adp = C.byref(a)
        if self._handle:
            lib().pe_dit_destroy(self._handle)
            self._handle = C.c_void_p()
        check(lib().pe_dit_create(C.byref(w), adp, C.byref(self._handle)), "pe_dit_create")

    def set_adapter(self, adapter_state_dict: Optional[Dict[str, torch.Tensor]]):
        """(Re)bind the visual-thinking adapter weights (pipe.load_state_dict of `visual_thinking_adapter.*`,
        validate.py:55-65).  Only the C handle's pointer table is rebuilt: DiT weights, merged LoRAs and the e4m3 state are
        untouched."""
        self.adapter = None if adapter_state_dict is None else {
            k: v.to(device=self.device, dtype=BF).contiguous() for k, v in adapter_state_dict.items()}
        self._create()
        self._apply_hot()
        self._ws, self._bound, self._step_of = None, (0, 0, 0), {}

    def fork(self) -> "QwenImageDiTEngine":
        """A second execution context on the SAME device weights: own C handle, workspace and prepared tables.
        Lets the positive and the negative forward of a step run concurrently on two streams."""
        other = object.__new__(QwenImageDiTEngine)
        other.device = self.device
        other.num_layers = self.num_layers
        other.fp8 = self.fp8
        other.params, other._fused, other.adapter = self.params, self._fused, self.adapter
        other.t_min, other.t_max = self.t_min, self.t_max
        other._handle = C.c_void_p()
        other._keep = None
        other._create()
        other._hot_sets = list(getattr(self, "_hot_sets", None) or [])
        other._apply_hot()
        other._ws = None
        other._bound = (0, 0, 0)
        other._step_of = {}
        other.version = self.version
        other.rope = self.rope
        other._eligen_words = None
        return other

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                lib().pe_dit_destroy(self._handle)
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------
    # LoRA merge (GeneralLoRALoader.load, lora/__init__.py:28-45): W <- bf16(W + bf16(alpha * B @ A))
    # ------------------------------------------------------------------------------------------
    _HOT_GROUPS = {   # LoRA target -> (fused group, slot inside the group)
        "attn.to_q": ("img_qkv", 0), "attn.to_k": ("img_qkv", 1), "attn.to_v": ("img_qkv", 2),
        "attn.add_q_proj": ("txt_qkv", 0), "attn.add_k_proj": ("txt_qkv", 1), "attn.add_v_proj": ("txt_qkv", 2),
        "attn.to_out.0": ("img_out", 0), "attn.to_add_out": ("txt_out", 0),
        "img_mlp.net.2": ("img_down", 0), "txt_mlp.net.2": ("txt_down", 0),
        "img_mod.1": ("img_mod", 0), "txt_mod.1": ("txt_mod", 0),
    }
    _HOT_SHAPES = {"qkv": (3 * D, D, 3), "out": (D, D, 1), "down": (D, 4 * D, 1), "mod": (6 * D, D, 1)}

    def load_lora_hot(self, lora_state_dict: Dict[str, torch.Tensor], alpha: float = 1.0) -> int:
        """load_lora(hotload=True) (qwen_image_physical.py:264-272): the LoRA stays separate and every targeted
        Linear computes out + x @ (alpha*A).T @ B.T at run time (vram_management/layers.py:173-181).  Calling it
        again APPENDS another set, applied after the earlier ones exactly like the reference's lists of pairs;
        rank <= 128 per set, at most 8 sets."""
        found = {}
        for key, A in lora_state_dict.items():
            if ".lora_A." not in key:
                continue
            name = key.split(".lora_A.")[0]
            kb = key.replace(".lora_A.", ".lora_B.")
            parts = name.split(".")
            if parts[0] == "diffusion_model":
                parts = parts[1:]
            if len(parts) < 3 or parts[0] != "transformer_blocks" or kb not in lora_state_dict:
                raise _lib.PeError(f"load_lora(hotload=True): unsupported LoRA key {key}")
            blk, target = int(parts[1]), ".".join(parts[2:])
            if target not in self._HOT_GROUPS:
                raise _lib.PeError(f"load_lora(hotload=True): {target} is not a hot-loadable Linear of the block")
            found[(blk, target)] = ((A.to(BF) * alpha).to(self.device), lora_state_dict[kb].to(device=self.device, dtype=BF))
        if not found:
            return 0
        rank = max(a.shape[0] for a, _ in found.values())
        rp = (rank + 63) // 64 * 64
        if rp > 128:
            raise _lib.PeError(f"load_lora(hotload=True): rank {rank} > 128")
        hot = []
        for i in range(self.num_layers):
            blk = {}
            for g in ("img_qkv", "txt_qkv", "img_out", "txt_out", "img_down", "txt_down", "img_mod", "txt_mod"):
                n_out, n_in, parts_n = self._HOT_SHAPES[g.split("_")[1]]
                a = torch.zeros((parts_n * rp, n_in), dtype=BF, device=self.device)
                b = torch.zeros((n_out, parts_n * rp), dtype=BF, device=self.device)
                blk[g + "_a"], blk[g + "_b"] = a, b
            for (bi, target), (A, B) in found.items():
                if bi != i:
                    continue
                g, slot = self._HOT_GROUPS[target]
                r = A.shape[0]
                rows = B.shape[0]
                blk[g + "_a"][slot * rp: slot * rp + r].copy_(A)
                blk[g + "_b"][slot * rows:(slot + 1) * rows, slot * rp: slot * rp + r].copy_(B)
            hot.append(blk)
        sets = list(getattr(self, "_hot_sets", None) or [])
        if len(sets) >= 8:
            raise _lib.PeError("load_lora(hotload=True): at most 8 hot LoRA sets; clear_lora() first")
        sets.append((hot, rp))
        self._hot_sets = sets
        self._apply_hot()
        return len(found)

    def clear_lora(self):
        self._hot_sets = []
        check(lib().pe_dit_set_hot_lora(self._handle, None, 0), "pe_dit_set_hot_lora")
        self._step_of = {}
        self.version += 1

    def _apply_hot(self):
        self.version = getattr(self, "version", 0) + 1
        sets = getattr(self, "_hot_sets", None) or []
        check(lib().pe_dit_set_hot_lora(self._handle, None, 0), "pe_dit_set_hot_lora")
        for hot, rp in sets:
            arr = (DitBlockLora * max(self.num_layers, 1))()
            for i, blk in enumerate(hot):
                for k, t in blk.items():
                    setattr(arr[i], k, t.data_ptr())
            check(lib().pe_dit_add_hot_lora(self._handle, arr, rp), "pe_dit_add_hot_lora")
        self._step_of = {}

    def load_lora(self, lora_state_dict: Dict[str, torch.Tensor], alpha: float = 1.0, hotload: bool = False) -> int:
        if hotload:
            return self.load_lora_hot(lora_state_dict, alpha)
        if self.fp8:
            raise _lib.PeError("load_lora: merge the LoRA before enable_fp8_computation() (the merge is a bf16 operation), "
                               "or use hotload=True")
        n = 0
        for key, up in lora_state_dict.items():
            if ".lora_B." not in key:
                continue
            parts = key.split(".")
            i = parts.index("lora_B")
            if len(parts) > i + 2:
                parts.pop(i + 1)
            parts.pop(parts.index("lora_B"))
            if parts[0] == "diffusion_model":
                parts.pop(0)
            parts.pop(-1)
            name = ".".join(parts) + ".weight"
            if name not in self.params:
                continue
            down = lora_state_dict[key.replace(".lora_B.", ".lora_A.")]
            up = up.to(device=self.device, dtype=BF)
            down = down.to(device=self.device, dtype=BF)
            r = up.shape[1]
            rp = (r + 63) // 64 * 64                    # GEMM K granule; zero padding adds exact zeros
            a = torch.zeros((up.shape[0], rp), dtype=BF, device=self.device)
            a[:, :r] = up
            wt = torch.zeros((down.shape[1], rp), dtype=BF, device=self.device)
            wt[:, :r] = down.t()
            W = self.params[name]
            # W <- bf16(W + bf16(alpha * bf16(up @ down)))  (lora/__init__.py:40-44)
            check(lib().pe_lora_merge(W.data_ptr(), W.shape[0], W.shape[1], a.data_ptr(), wt.data_ptr(), rp, float(alpha),
                                      stream_ptr()), "pe_lora_merge")
            n += 1
        if n:
            # img_mod.1 / txt_mod.1 are LoRA targets: the prepared modulation rows (host map and the C tables) are stale
            self._step_of = {}
            self.version += 1
        return n

    # ------------------------------------------------------------------------------------------
    def bind(self, S_img_max: int, T_max: int, n_steps: int):
        if self._bound[0] >= S_img_max and self._bound[1] >= T_max and self._bound[2] >= n_steps and self._ws is not None:
            return
        S_img_max = max(S_img_max, self._bound[0]); T_max = max(T_max, self._bound[1]); n_steps = max(n_steps, self._bound[2])
        nbytes = lib().pe_dit_workspace_bytes(self._handle, S_img_max, T_max, n_steps)
        self._ws = None
        self._ws = torch.empty((nbytes + 256,), dtype=torch.uint8, device=self.device)
        base = (self._ws.data_ptr() + 255) // 256 * 256
        check(lib().pe_dit_bind_workspace(self._handle, base, nbytes, S_img_max, T_max, n_steps, stream_ptr()),
              "pe_dit_bind_workspace")
        self._bound = (S_img_max, T_max, n_steps)
        self._step_of = {}

    def prepare(self, timesteps_bf16: torch.Tensor):
        """timesteps_bf16: [n] CPU tensor, scheduler timesteps ALREADY rounded to the pipeline dtype
        (qwen_image_physical.py:649).  Builds temb / modulation rows for all of them."""
        assert timesteps_bf16.dim() == 1
        n = timesteps_bf16.shape[0]
        if self._ws is None or self._bound[2] < n:
            raise _lib.PeError("prepare: call bind() with n_steps >= len(timesteps) first")
        t_scaled = timesteps_bf16.cpu() / 1000          # :1342, in the pipeline dtype
        sin = timestep_sinusoid(t_scaled).to(BF).to(self.device).contiguous()   # models/utils.py:291
        check(lib().pe_dit_prepare(self._handle, sin.data_ptr(), n, stream_ptr()), "pe_dit_prepare")
        self._sin_keep = sin
        self._step_of = {float(t): i for i, t in enumerate(timesteps_bf16.float().tolist())}

    # ------------------------------------------------------------------------------------------
    def forward(self, latents: torch.Tensor, timestep: torch.Tensor, prompt_emb: torch.Tensor,
                special_idx: Optional[torch.Tensor] = None, edit_latents=None, step: Optional[int] = None,
                out: Optional[torch.Tensor] = None, controls=None, entity_prompt_emb=None, entity_masks=None,
                edit_rope_interpolation: bool = False, enable_fp8_attention: bool = False) -> torch.Tensor:
        """One model_fn call.  `timestep`: [1] tensor in the pipeline dtype.  `prompt_emb` [1,T,3584] is
        MUTATED IN PLACE on `special_idx` rows.  Returns noise_pred [1,16,h8,w8].
        `controls`: active block-wise ControlNet inputs, [(QwenImageBlockWiseControlNet, processed conditioning [S0,3072], scale)]
        (physicedit_amd.controlnet).
        `entity_prompt_emb` (list of [1,T_i,3584]) + `entity_masks` ([1,N,1,h8,w8] in {0,1}): EliGen entity control
        (QwenImageDiT.process_entity_masks, qwen_image_dit.py:433-498).
        `edit_rope_interpolation`: RoPE tables of QwenEmbedRope.forward_sampling (:1367-1368); ignored with EliGen, as in the reference.
        `enable_fp8_attention`: the e4m3 attention branch of qwen_image_flash_attention (qwen_image_dit.py:24-35; pe_flash_attn_fp8);
        like the reference's, it is not taken under an attention mask (EliGen)."""
        ops._chk(latents, "latents"), ops._chk(prompt_emb, "prompt_emb")
        h8, w8 = latents.shape[-2:]
        edits: List[torch.Tensor] = []
        if edit_latents is not None:
            edits = list(edit_latents) if isinstance(edit_latents, (list, tuple)) else [edit_latents]
        img_shapes = [(1, h8 // 2, w8 // 2)] + [(1, e.shape[-2] // 2, e.shape[-1] // 2) for e in edits]
        S_img = sum(f * h * w for f, h, w in img_shapes)
        T = prompt_emb.shape[-2]
        eligen = None
        if entity_prompt_emb is not None:
            eligen = self._eligen_inputs(prompt_emb, special_idx, entity_prompt_emb, entity_masks, img_shapes, h8, w8)
            prompt_emb_all, special_idx, seg_lens, words = eligen
            T = prompt_emb_all.shape[-2]
        if step is None:
            key = float(timestep.float().item())
            if key not in self._step_of or S_img > self._bound[0] or T > self._bound[1]:
                # ad-hoc call outside a prepared loop: (re)bind and build a 1-row table for this timestep
                self.bind(S_img, T, 1)
                self.prepare(timestep.detach().reshape(1).cpu())
            step = self._step_of[key]
        elif S_img > self._bound[0] or T > self._bound[1]:
            raise _lib.PeError(f"forward: sequence ({S_img},{T}) exceeds the bound workspace {self._bound[:2]}; "
                               "call bind() with the maximum sizes before prepare()")
        if eligen is not None:
            cos_i, sin_i, cos_t, sin_t = self.rope.get_segments(img_shapes, seg_lens)
        else:
            cos_i, sin_i, cos_t, sin_t = self.rope.get(img_shapes, T, sampling=bool(edit_rope_interpolation))
        if out is None:
            out = torch.empty((1, 16, h8, w8), dtype=BF, device=self.device)
        c = DitCall()
        c.latents, c.h8, c.w8 = latents.data_ptr(), h8, w8
        c.n_edit = len(edits)
        for i, e in enumerate(edits):
            ops._chk(e, "edit_latents")
            c.edit_latents[i] = e.data_ptr()
            c.edit_h8[i], c.edit_w8[i] = e.shape[-2], e.shape[-1]
        c.prompt_emb, c.T = (prompt_emb_all if eligen is not None else prompt_emb).data_ptr(), T
        c.attn_words = words.data_ptr() if eligen is not None else None
        c.fp8_attention = 1 if enable_fp8_attention else 0
        if special_idx is not None and special_idx.numel() > 0:
            if self.adapter is None:
                raise _lib.PeError("special tokens given but no visual_thinking_adapter weights loaded")
            c.special_idx, c.n_special = special_idx.data_ptr(), special_idx.numel()
            c.alpha, c.one_minus_alpha = adapter_alpha(timestep.detach().cpu(), self.t_min, self.t_max)
        else:
            c.special_idx, c.n_special = None, 0
        c.rope_cos_img, c.rope_sin_img = cos_i.data_ptr(), sin_i.data_ptr()
        c.rope_cos_txt, c.rope_sin_txt = cos_t.data_ptr(), sin_t.data_ptr()
        c.step = step
        c.noise_pred = out.data_ptr()
        c.n_control = 0
        if controls:
            # with the e4m3 DiT the ControlNet stays bf16, as in the reference: enable_vram_management wraps its Linears with
            # computation_dtype = the pipeline dtype (qwen_image_physical.py:478-493), only the DiT gets the fp8 dtype
            if len(controls) > 4:
                raise _lib.PeError("at most 4 active ControlNet inputs per call")
            S0 = (h8 // 2) * (w8 // 2)
            for i, (net, cond, scale) in enumerate(controls):
                ops._chk(cond, "controlnet conditioning")
                if net.num_layers < self.num_layers or tuple(cond.shape) != (S0, 3072):
                    raise _lib.PeError(f"controlnet input {i}: {net.num_layers} blocks / conditioning {tuple(cond.shape)} for a "
                                       f"{self.num_layers}-layer DiT with {S0} noise tokens")
                c.control[i].blocks = net.block_table
                c.control[i].conditioning = cond.data_ptr()
                c.control[i].scale = float(scale)
            c.n_control = len(controls)
        check(lib().pe_dit_forward(self._handle, C.byref(c), stream_ptr()), "pe_dit_forward")
        if eligen is not None:       # the adapter updated the global prompt's special rows inside the concatenated copy (:1336)
            prompt_emb.reshape(-1, prompt_emb.shape[-1]).copy_(prompt_emb_all.reshape(-1, prompt_emb.shape[-1])[T - prompt_emb.shape[-2]:])
        return out

    def _eligen_inputs(self, prompt_emb, special_idx, entity_prompt_emb, entity_masks, img_shapes, h8, w8):
        """Host side of process_entity_masks: the text stream becomes [entity prompts ..., global prompt] (txt_norm / txt_in are
        row-wise, so concatenating before them is the reference's concatenation after them), and the region mask becomes one
        word per token of the library's joint order [image | text] (include/physicedit_amd.h, pe_dit_call.attn_words)."""
        ents = [e.to(device=self.device, dtype=BF).reshape(-1, e.shape[-1]) for e in entity_prompt_emb]
        n_ent = len(ents)
        seg_lens = [e.shape[0] for e in ents] + [prompt_emb.shape[-2]]
        # the token words depend on the region masks and the lengths only: built once per image (every step of both CFG branches
        # comes through here), on the host, 4 bytes per token
        # The cache holds a REFERENCE to the masks tensor and is valid only for that very object at that version: a new tensor
        # for the next image can then never alias it (a freed tensor's address is reused by the caching allocator, and a fresh
        # tensor's _version is 0 again), whoever the caller is (DenoiseLoop, model_fn_qwen_image, a forked engine).
        key = (entity_masks._version, tuple(entity_masks.shape), tuple(seg_lens), tuple(map(tuple, img_shapes)), h8, w8)
        cache = getattr(self, "_eligen_words", None)
        if cache is None or cache[0] is not entity_masks or cache[1] != key:
            self._eligen_words = (entity_masks, key, self._eligen_token_words(entity_masks, n_ent, seg_lens, img_shapes, h8, w8))
        words = self._eligen_words[2]
        prompt_emb_all = torch.cat(ents + [prompt_emb.reshape(-1, prompt_emb.shape[-1])]).unsqueeze(0).contiguous()
        if special_idx is not None and special_idx.numel() > 0:
            # An entity entry that IS the prompt tensor (eligen_enable_on_negative hands the negative prompt_emb out N times,
            # :1177) is updated by the adapter together with it in the reference (same storage); the adapter is row-wise, so
            # running it on the aliased copies' special rows too gives exactly those values.
            offs, o = [], 0
            for e, n in zip(entity_prompt_emb, seg_lens[:-1]):
                if e.data_ptr() == prompt_emb.data_ptr() and e.shape[-2] == prompt_emb.shape[-2]:
                    offs.append(o)
                o += n
            offs.append(o)
            special_idx = torch.cat([special_idx + a for a in offs]).to(torch.int32)
        return prompt_emb_all, special_idx, seg_lens, words

    def _eligen_token_words(self, entity_masks, n_ent, seg_lens, img_shapes, h8, w8) -> torch.Tensor:
        if n_ent + 1 > 31:
            raise _lib.PeError("EliGen: at most 30 entity prompts")
        em = entity_masks.to("cpu", torch.float32)
        if em.dim() != 5 or em.shape[0] != 1 or em.shape[1] != n_ent or tuple(em.shape[-2:]) != (h8, w8):
            raise _lib.PeError(f"EliGen: entity_masks {tuple(entity_masks.shape)} for {n_ent} prompts and {h8}x{w8} latents")
        S0 = (h8 // 2) * (w8 // 2)
        S_img = sum(f * h * w for f, h, w in img_shapes)
        if S_img % S0 != 0:
            raise _lib.PeError("EliGen: every image of the sequence must have the size of the noise latents (the reference "
                               "repeats the region mask over them, qwen_image_dit.py:477-478)")
        # token (y, x) belongs to region i iff any mask pixel of its 2 x 2 latent patch (any channel) is set (:462-463, :474)
        region = em[0].amax(dim=1).reshape(n_ent, h8 // 2, 2, w8 // 2, 2).amax(dim=(2, 4)).reshape(n_ent, S0) > 0
        bits = torch.zeros((S0,), dtype=torch.int64)
        for i in range(n_ent):
            bits |= region[i].to(torch.int64) << i
        bits |= (1 << n_ent) | (1 << 31)                      # the global prompt sees every image token; image tokens see each other
        text_bits = torch.cat([torch.full((n,), 1 << i, dtype=torch.int64) for i, n in enumerate(seg_lens)])
        S = S_img + sum(seg_lens)
        words = torch.zeros(((S + 63) // 64 * 64,), dtype=torch.int64)
        words[:S_img] = bits.repeat(S_img // S0)
        words[S_img:S] = text_bits
        return (words & 0xFFFFFFFF).to(torch.uint32).view(torch.int32).to(self.device)

    def special_token_loss(self, timestep: torch.Tensor, n_special: int, gt_dino: torch.Tensor, gt_vae: torch.Tensor,
                           epsilon: float = 0.1) -> torch.Tensor:
        """VisualThinkingDualAdapter.get_loss (pipelines/helpers.py:166-183) for the adapter predictions of the LAST forward on this
        engine: the two mean squared errors come from the device (pe_dit_special_token_mse, the reference's roundings), the
        time-dependent weighting is the reference's scalar arithmetic on one-element bf16 tensors, done here on the host.
        Synchronises (training only).  Returns a 0-dim bf16 tensor like the reference."""
        gd = gt_dino.to(device=self.device, dtype=BF).contiguous()
        gv = gt_vae.to(device=self.device, dtype=BF).contiguous()
        if gd.numel() != n_special * 3584 or gv.numel() != n_special * 3584:
            raise _lib.PeError(f"special_token_loss: targets {tuple(gd.shape)} / {tuple(gv.shape)} for {n_special} special tokens")
        out2 = torch.empty(2, dtype=torch.float32, device=self.device)
        check(lib().pe_dit_special_token_mse(self._handle, gd.data_ptr(), gv.data_ptr(), int(n_special), out2.data_ptr(), stream_ptr()),
              "pe_dit_special_token_mse")
        loss_dino, loss_vae = out2.cpu().to(BF).unbind(0)                       # .mean(dim=[1, 2]) of a bf16 tensor -> bf16
        t = timestep.detach().cpu()
        alpha = ((t - self.t_min) / (self.t_max - self.t_min + 1e-6)).clamp(0.0, 1.0).view(-1, 1, 1).to(BF)     # _get_alpha + type_as
        w = alpha.squeeze()
        weight_dino = w + epsilon
        weight_vae = (1 - w) + epsilon
        total = weight_dino + weight_vae
        weight_dino, weight_vae = weight_dino / total, weight_vae / total
        return (weight_dino * loss_dino + weight_vae * loss_vae).mean()

    def debug_tensor(self, name: str, shape, dtype=BF) -> torch.Tensor:
        """Copy of an internal workspace region (tests only)."""
        ptr = lib().pe_dit_debug_ptr(self._handle, name.encode())
        if not ptr:
            raise KeyError(name)
        off = ptr - self._ws.data_ptr()
        n = 1
        for d in shape:
            n *= d
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        return self._ws[off:off + nbytes].view(dtype).reshape(shape).clone()


def special_indices(special_token_mask: Optional[torch.Tensor], device) -> Optional[torch.Tensor]:
    """bool mask [1,T] -> int32 device indices (row-major order == boolean-mask gather order)."""
    if special_token_mask is None:
        return None
    idx = torch.nonzero(special_token_mask.reshape(-1).cpu(), as_tuple=False).reshape(-1).to(torch.int32)
    return idx.to(device)


def model_fn_qwen_image(dit: QwenImageDiTEngine = None, blockwise_controlnet=None, visual_thinking_adapter=None,
                        latents=None, timestep=None, prompt_emb=None, prompt_emb_mask=None, special_token_mask=None,
                        height=None, width=None, blockwise_controlnet_conditioning=None,
                        blockwise_controlnet_inputs=None, progress_id=0, num_inference_steps=1,
                        entity_prompt_emb=None, entity_prompt_emb_mask=None, entity_masks=None, edit_latents=None,
                        context_latents=None, enable_fp8_attention=False, use_gradient_checkpointing=False,
                        use_gradient_checkpointing_offload=False, edit_rope_interpolation=False, is_train=True,
                        pseudo_special_emb_dino=None, pseudo_special_emb_vae=None, **kwargs):
    """Drop-in for the reference operator of the same name (qwen_image_physical.py:1302-1403),
    inference subset: returns (noise_pred, 0).  Unsupported reference features raise instead of
    silently differing.  `enable_fp8_attention=True` takes the e4m3 attention branch the reference takes where FlashAttention-3 exists
    (qwen_image_dit.py:24-35): here that branch is the library's own kernel (pe_flash_attn_fp8), so the flag always acts."""
    if entity_prompt_emb is not None and entity_masks is None:
        raise _lib.PeError("model_fn_qwen_image: entity_prompt_emb without entity_masks")
    want_loss = bool(is_train) and special_token_mask is not None
    if want_loss and (pseudo_special_emb_dino is None or pseudo_special_emb_vae is None):
        raise _lib.PeError("model_fn_qwen_image: is_train=True needs pseudo_special_emb_dino / pseudo_special_emb_vae (the targets of "
                           "get_loss, produced by the training-time PhysicalVisualEmbedder); inference passes is_train=False")
    edits = []
    if context_latents is not None:
        edits.append(context_latents)      # context tokens come right after the noise tokens (:1348-1351)
    if edit_latents is not None:
        edits += list(edit_latents) if isinstance(edit_latents, (list, tuple)) else [edit_latents]
    idx = None
    if special_token_mask is not None:
        idx = getattr(special_token_mask, "_pe_idx", None)
        if idx is None:
            idx = special_indices(special_token_mask, prompt_emb.device)
    controls = None
    if blockwise_controlnet_conditioning is not None:
        # :1373-1375 (img_in of every conditioning, every call) and the per-block hook :1389-1396 with its progress gate :175-177.
        # A conditioning that is already [S0,3072] was pre-processed by the caller (the result is the same every step).
        if blockwise_controlnet is None or blockwise_controlnet_inputs is None:
            raise _lib.PeError("model_fn_qwen_image: blockwise_controlnet_conditioning without blockwise_controlnet / inputs")
        processed = [c if (c.dim() == 2 and c.shape[-1] == 3072) else
                     blockwise_controlnet.models[ci.controlnet_id].process_controlnet_conditioning(c)
                     for ci, c in zip(blockwise_controlnet_inputs, blockwise_controlnet_conditioning)]
        controls = blockwise_controlnet.active_controls(blockwise_controlnet_inputs, processed, progress_id, num_inference_steps)
    pred = dit.forward(latents, timestep, prompt_emb, idx, edits or None, controls=controls,
                       entity_prompt_emb=entity_prompt_emb, entity_masks=entity_masks, edit_rope_interpolation=edit_rope_interpolation,
                       enable_fp8_attention=bool(enable_fp8_attention))
    if want_loss:        # :1337-1338: the loss value of the training path (forward only: this library has no backward)
        return pred, dit.special_token_loss(timestep, idx.numel(), pseudo_special_emb_dino, pseudo_special_emb_vae)
    return pred, 0
