"""Host side of EliGen entity control (no GPU): the one-word-per-token encoding the library's attention takes
(include/physicedit_amd.h, pe_dit_call.attn_words; built by QwenImageDiTEngine._eligen_inputs) must reproduce the reference's
[S, S] attention mask (QwenImageDiT.process_entity_masks, qwen_image_dit.py:449-498) exactly, permuted from the reference's
token order [prompts ..., image] to the library's [image | prompts ...]."""
import torch

from physicedit_amd import synth
from physicedit_amd.dit import QwenImageDiTEngine

BF = torch.bfloat16


def _engine_shell():
    eng = QwenImageDiTEngine.__new__(QwenImageDiTEngine)      # only the host helper is exercised: no library, no weights
    eng.device = torch.device("cpu")
    return eng


def test_token_words_reproduce_the_reference_mask(golden):
    g = golden("G14_eligen")
    ents = [synth.make_prompt_emb(25 + i, T) for i, T in enumerate((12, 20, 8))]
    pe = synth.make_prompt_emb(12, 40)
    idx = torch.tensor([3, 4, 5], dtype=torch.int32)
    pe_all, idx2, seg, words = _engine_shell()._eligen_inputs(pe, idx, ents, g["entity_masks"], [(1, 8, 8), (1, 8, 8)], 16, 16)
    assert seg == [12, 20, 8, 40] and pe_all.shape == (1, 80, 3584)
    assert torch.equal(pe_all[0, 40:], pe[0]) and torch.equal(pe_all[0, :12], ents[0][0])
    assert idx2.tolist() == [43, 44, 45]                       # special rows live in the global prompt, after the entity prompts
    S_img, S = 128, 128 + 80
    assert words.numel() == 256 and words.dtype == torch.int32 and int(words[S:].abs().sum()) == 0
    w = words.view(torch.uint32).to(torch.int64)[:S]
    allowed = (w[:, None] & w[None, :]) != 0
    perm = torch.cat([torch.arange(S_img, S), torch.arange(0, S_img)])
    assert torch.equal(allowed[perm][:, perm], g["attention_allowed"].bool())
    # every token sees itself: no row of the mask is empty (the kernel's -inf guard is for per-tile emptiness only)
    assert bool(allowed.diagonal().all())


def test_rope_segments_restart_per_prompt(golden):
    from physicedit_amd.rope import RopeCache
    g = golden("G14_eligen")
    cos_i, sin_i, cos_t, sin_t = RopeCache("cpu").get_segments([(1, 8, 8), (1, 8, 8)], [12, 20, 8, 40])
    assert torch.equal(cos_t, g["txt_rotary_real"]) and torch.equal(sin_t, g["txt_rotary_imag"])


def test_entity_mask_preprocessing_matches_the_reference_unit(golden):
    """QwenImageUnit_EntityControl.preprocess_masks + prepare_entity_inputs (qwen_image_physical.py:1157-1167) vs the facade's
    host code: PIL masks -> [1, N, 1, h/8, w/8] in {0, 1}."""
    from PIL import Image
    from diffsynth.pipelines.qwen_image_physical import QwenImagePhysicPipeline
    g = golden("G14_eligen")
    pipe = QwenImagePhysicPipeline.__new__(QwenImagePhysicPipeline)
    pipe.device, pipe.torch_dtype = torch.device("cpu"), BF
    out = pipe.preprocess_entity_masks([Image.fromarray(g[f"unit_mask{i}"].numpy()) for i in range(2)], 12, 20)
    assert out.dtype == BF and torch.equal(out, g["unit_entity_masks"])


def test_dino_input_preprocess_host():
    """QwenImagePhysicPipeline.dino_input_preprocess (the torchvision pipeline of :1043-1057 restated with PIL + torch.randint): shape,
    ImageNet normalisation, the crop is random but reproducible under the global torch seed, and lies inside the 1.5x resized frame."""
    import numpy as np
    from PIL import Image
    from diffsynth.pipelines.qwen_image_physical import QwenImagePhysicPipeline
    pipe = QwenImagePhysicPipeline.__new__(QwenImagePhysicPipeline)
    pipe.device, pipe.torch_dtype, pipe.dino_input_size = torch.device("cpu"), torch.bfloat16, 224
    rs = np.random.RandomState(0)
    frames = [Image.fromarray((rs.rand(h, w, 3) * 255).astype("uint8")) for h, w in ((96, 160), (300, 200))]
    torch.manual_seed(11)
    a = pipe.dino_input_preprocess(frames)
    torch.manual_seed(11)
    b = pipe.dino_input_preprocess(frames)
    c = pipe.dino_input_preprocess(frames)
    assert a.shape == (2, 3, 224, 224) and a.dtype == torch.float32 and torch.equal(a, b) and not torch.equal(a, c)
    lo, hi = (0 - 0.485) / 0.229, (1 - 0.406) / 0.225
    assert a.min().item() >= lo - 1e-5 and a.max().item() <= hi + 1e-5
