"""Denoising loop of QwenImagePhysicPipeline on the HIP kernels.

Mirrors DiffSynth-Studio/diffsynth/pipelines/qwen_image_physical.py:
    :600      scheduler.set_timesteps(steps, dynamic_shift_len=(H//16)*(W//16))
    :644-661  for each timestep: posi forward, nega forward (cfg != 1), CFG combine, Euler step
    :664-667  VAE decode (physicedit_amd.vae)
The prologue (text encoder, tokenizer, physical-reasoning text) is host Python on HF transformers in
the reference and is outside this path: its outputs (prompt_emb, special_token_mask, edit image) are
the inputs here.  The loop issues kernels only: no host<->device synchronisation inside it
(the reference has three per forward: `.tolist()`, boolean-mask gather/scatter, scheduler argmin).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import _lib, ops
from .dit import QwenImageDiTEngine, special_indices
from .scheduler import qwen_image_scheduler

BF = torch.bfloat16


class DenoiseLoop:
    def __init__(self, dit: QwenImageDiTEngine, dual_stream: bool = False, cfg_pair=None):
        """dual_stream: run the positive and the negative forward of each step concurrently on two HIP streams
        (second workspace on the same weights).  They are independent until the CFG combine (:653-656), and each
        fills the CUs the other leaves idle in its partial rounds of work-groups.
        cfg_pair: a `parallel.CfgPairExchange`: THIS rank runs only the forward of its role (0 positive, 1 negative) and the
        two `noise_pred` tensors are all-gathered inside the rank pair every step (512 KiB over xGMI); both ranks then apply
        the same CFG combine + Euler update and hold identical latents.  Ignored when cfg_scale == 1."""
        self.dit = dit
        self.device = dit.device
        self.scheduler = qwen_image_scheduler()
        self.torch_dtype = BF
        self.dual_stream = dual_stream
        self.cfg_pair = cfg_pair
        self._dit_n: Optional[QwenImageDiTEngine] = None
        self._streams = None

    @torch.no_grad()
    def __call__(self, noise: torch.Tensor, prompt_emb_posi: torch.Tensor, prompt_emb_nega: Optional[torch.Tensor],
                 special_mask_posi: Optional[torch.Tensor], special_mask_nega: Optional[torch.Tensor],
                 height: int, width: int, num_inference_steps: int = 30, cfg_scale: float = 4.0,
                 edit_latents=None, exponential_shift_mu: Optional[float] = None,
                 denoising_strength: float = 1.0, blockwise_controlnet=None, blockwise_controlnet_inputs=None,
                 blockwise_controlnet_conditioning=None, eligen_posi=None, eligen_nega=None,
                 input_latents: Optional[torch.Tensor] = None, inpaint_mask: Optional[torch.Tensor] = None,
                 edit_rope_interpolation: bool = False, enable_fp8_attention: bool = False, on_step=None) -> torch.Tensor:
        """noise [1,16,H/8,W/8] (for an image-to-image run: already `scheduler.add_noise(input_latents, noise, timesteps[0])`);
        prompt_emb_* [1,T,3584] DEVICE tensors, mutated in place on their special rows across the steps exactly like
        `inputs_posi["prompt_emb"]` in the reference.  `inpaint_mask` [1,1,H/8,W/8] + `input_latents`: the blend of
        BasePipeline.step (utils/__init__.py:150-156) rides in the CFG / Euler kernel.
        `on_step(i, latents)`: called on the loop's stream behind step i's update (the reference's progress hook, :648; the latents
        tensor is re-used two steps later: clone what is kept)."""
        dev = self.device
        sch = self.scheduler
        sch.set_timesteps(num_inference_steps, denoising_strength=denoising_strength,
                          dynamic_shift_len=(height // 16) * (width // 16), exponential_shift_mu=exponential_shift_mu)
        ts = sch.timesteps.to(self.torch_dtype)        # per step: timestep.unsqueeze(0).to(dtype)  (:649)
        edits: List[torch.Tensor] = []
        if edit_latents is not None:
            edits = list(edit_latents) if isinstance(edit_latents, (list, tuple)) else [edit_latents]
        use_cfg = cfg_scale != 1.0                      # (:654)
        S_img = (height // 16) * (width // 16) + sum((e.shape[-2] // 2) * (e.shape[-1] // 2) for e in edits)
        # EliGen (:1186-1198): dict(entity_prompt_emb=[...], entity_masks=[1,N,1,h8,w8]) per CFG branch; the entity prompts join the
        # text stream, so they count for the workspace
        ent_len = lambda e: sum(x.shape[-2] for x in e["entity_prompt_emb"]) if e else 0
        T_max = max(prompt_emb_posi.shape[-2] + ent_len(eligen_posi),
                    (prompt_emb_nega.shape[-2] + ent_len(eligen_nega)) if use_cfg else 0)
        kw_p = dict(eligen_posi) if eligen_posi else {}
        kw_n = dict(eligen_nega) if eligen_nega else {}
        if edit_rope_interpolation:                    # (:1367-1368) same tables for both CFG branches
            kw_p["edit_rope_interpolation"] = kw_n["edit_rope_interpolation"] = True
        if enable_fp8_attention:                       # (:614, :1321) both CFG branches
            kw_p["enable_fp8_attention"] = kw_n["enable_fp8_attention"] = True
        self.dit._eligen_words = None                  # token words are cached per image (QwenImageDiTEngine._eligen_inputs)
        pair = self.cfg_pair if use_cfg else None
        dual = self.dual_stream and use_cfg and pair is None
        dit_n = self.dit
        if dual:
            if self._dit_n is None or self._dit_n.version != self.dit.version or self._dit_n.fp8 != self.dit.fp8:
                # (re)fork: the fork snapshots the primary's hot-LoRA / e4m3 state at fork time (own C handle)
                self._dit_n = self.dit.fork()
                self._streams = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
            dit_n = self._dit_n
            dit_n._eligen_words = None
            dit_n.bind(S_img, T_max, num_inference_steps)
            dit_n.prepare(ts)
        self.dit.bind(S_img, T_max, num_inference_steps)
        self.dit.prepare(ts)
        idx_p = special_indices(special_mask_posi, dev)
        idx_n = special_indices(special_mask_nega, dev) if use_cfg else None
        latents = noise.to(device=dev, dtype=self.torch_dtype).contiguous().clone()
        nxt = torch.empty_like(latents)
        pred_p = torch.empty_like(latents)
        pred_n = torch.empty_like(latents) if use_cfg else None
        main = torch.cuda.current_stream(dev) if dual else None
        x0 = mask = None
        if inpaint_mask is not None:
            if input_latents is None:
                raise _lib.PeError("inpaint_mask needs input_latents (the reference's step() would fail on None - None)")
            x0 = input_latents.to(device=dev, dtype=self.torch_dtype).contiguous()
            mask = inpaint_mask.to(device=dev, dtype=self.torch_dtype).contiguous()
        # block-wise ControlNet (:1373-1396): img_in of each conditioning once (the reference redoes it every call with the same
        # result), then per step the inputs whose progress window contains the step
        processed = None
        if blockwise_controlnet_conditioning is not None:
            processed = blockwise_controlnet.preprocess(blockwise_controlnet_inputs, blockwise_controlnet_conditioning)
        for i in range(num_inference_steps):
            t = ts[i:i + 1]
            ctl = (blockwise_controlnet.active_controls(blockwise_controlnet_inputs, processed, i, num_inference_steps)
                   if processed is not None else None)
            if dual:
                sp, sn = self._streams
                sp.wait_stream(main)
                sn.wait_stream(main)
                with torch.cuda.stream(sp):
                    self.dit.forward(latents, t, prompt_emb_posi, idx_p, edits or None, step=i, out=pred_p, controls=ctl, **kw_p)
                with torch.cuda.stream(sn):
                    dit_n.forward(latents, t, prompt_emb_nega, idx_n, edits or None, step=i, out=pred_n, controls=ctl, **kw_n)
                main.wait_stream(sp)
                main.wait_stream(sn)
            elif pair is not None:
                # split CFG pair: one forward here, the sibling rank runs the other one on the same latents
                if pair.role == 0:
                    self.dit.forward(latents, t, prompt_emb_posi, idx_p, edits or None, step=i, out=pred_p, controls=ctl, **kw_p)
                    pred_p, pred_n = pair.exchange(pred_p)
                else:
                    self.dit.forward(latents, t, prompt_emb_nega, idx_n, edits or None, step=i, out=pred_n, controls=ctl, **kw_n)
                    pred_p, pred_n = pair.exchange(pred_n)
            else:
                self.dit.forward(latents, t, prompt_emb_posi, idx_p, edits or None, step=i, out=pred_p, controls=ctl, **kw_p)
                if use_cfg:
                    self.dit.forward(latents, t, prompt_emb_nega, idx_n, edits or None, step=i, out=pred_n, controls=ctl, **kw_n)
            if inpaint_mask is not None:
                ops.cfg_euler_step(pred_p, pred_n, latents, cfg_scale, sch.dsigma(i), out=nxt, input_latents=x0, inpaint_mask=mask,
                                   sigma=float(sch.sigmas[i]))
            else:
                ops.cfg_euler_step(pred_p, pred_n, latents, cfg_scale, sch.dsigma(i), out=nxt)
            latents, nxt = nxt, latents
            if on_step is not None:
                on_step(i, latents)
        return latents
