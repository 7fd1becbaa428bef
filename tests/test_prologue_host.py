"""CPU tests of the prompt prologue (diffsynth/pipelines/prompt_prologue.py) against outputs of the REFERENCE's units
(QwenImageUnit_PhysicalVerbalEmbedder / QwenImageUnit_PromptEmbedder, qwen_image_physical.py:732-990) on the same
synthetic tiny Qwen2.5-VL stack (tests/tiny_vl.py; fixture G12 written by tests/golden/make_golden.py)."""
import tempfile

import pytest
import torch

import tiny_vl
from diffsynth.pipelines import prompt_prologue as PP

BF = torch.bfloat16


@pytest.fixture(scope="module")
def prologue():
    proc = tiny_vl.make_processor(tempfile.mkdtemp(prefix="pe_tok_"))
    pp = PP.PromptPrologue(None, proc, device="cpu", torch_dtype=BF)          # adds the 66 special tokens (:529-536)
    pp.text_encoder = tiny_vl.make_text_encoder(proc.tokenizer, extra_vocab=8)
    return pp


def test_prologue_matches_reference_units(prologue, golden):
    g, meta = golden("G12_prologue", with_meta=True)
    img, img2 = tiny_vl.make_image(200, 120, 0), tiny_vl.make_image(64, 96, 1)
    phys = prologue.physical_text(meta["prompt"], img)
    assert phys == meta["physical_txt"]                # greedy generation + decode + JSON fallback, character for character
    cases = {"edit": dict(prompt=meta["prompt"], edit_image=img, physical_txt=phys), "edit_nega": dict(prompt="", edit_image=img),
             "t2i": dict(prompt="a red cube on a glass table"), "multi": dict(prompt="swap them", edit_image=[img, img2])}
    for name, kw in cases.items():
        r = prologue.embed(**kw)
        assert torch.equal(r["prompt_emb"], g[f"{name}.prompt_emb"]), name
        assert torch.equal(r["prompt_emb_mask"], g[f"{name}.prompt_emb_mask"]), name
        if f"{name}.special_token_mask" in g:
            assert torch.equal(r["special_token_mask"].to(torch.uint8), g[f"{name}.special_token_mask"]), name
            assert int(r["special_token_mask"].sum()) == PP.SPECIAL_TOKEN_NUM
        else:
            assert r["special_token_mask"] is None


def test_prologue_call_protocol(prologue):
    """The unit runner's separate-CFG protocol (utils/__init__.py:247-283): nega gets its own embedding when cfg is on,
    the posi outputs otherwise; the physical text only extends the positive prompt."""
    img = tiny_vl.make_image(96, 96, 3)
    posi, nega = prologue(None, prompt="tilt the glass", negative_prompt="", edit_image=img, cfg=True, have_text_reasoning=False)
    assert posi["prompt_emb"].shape[1] > nega["prompt_emb"].shape[1] > 64
    assert posi["prompt_emb"].dtype == BF and posi["special_token_mask"].shape[1] == posi["prompt_emb"].shape[1]
    p2, n2 = prologue(None, prompt="tilt the glass", negative_prompt="", edit_image=img, cfg=False, have_text_reasoning=False)
    assert torch.equal(p2["prompt_emb"], posi["prompt_emb"]) and torch.equal(n2["prompt_emb"], p2["prompt_emb"])
    with pytest.raises(ValueError):
        prologue(None, prompt="x", negative_prompt="", edit_image=None, cfg=True, have_text_reasoning=True)


def test_prologue_cache_returns_private_copies(prologue, monkeypatch):
    """A repeated (image, prompts) pair skips the text encoder, and what comes back are CLONES: the denoising loop mutates
    prompt_emb in place (qwen_image_physical.py:1336), which must not leak into the cache."""
    img = tiny_vl.make_image(96, 64, 5)
    prologue._cache.clear()
    kw = dict(prompt="melt the ice", negative_prompt="blurry", edit_image=img, cfg=True, have_text_reasoning=False)
    p1, n1 = prologue(None, **kw)
    calls = []
    monkeypatch.setattr(prologue, "embed", lambda *a, **k: calls.append(a) or (_ for _ in ()).throw(AssertionError("cache miss")))
    p1["prompt_emb"].zero_()                                       # what the loop does to its copy
    p2, n2 = prologue(None, **kw)                                  # same image CONTENT in a new PIL object
    p3, _ = prologue(None, **{**kw, "edit_image": tiny_vl.make_image(96, 64, 5)})
    assert not calls and float(p2["prompt_emb"].abs().sum()) > 0
    assert torch.equal(p2["prompt_emb"], p3["prompt_emb"]) and torch.equal(n1["prompt_emb"], n2["prompt_emb"])
    assert p2["prompt_emb"].data_ptr() != p3["prompt_emb"].data_ptr()
    with pytest.raises(AssertionError):                            # another image is a miss
        prologue(None, **{**kw, "edit_image": tiny_vl.make_image(96, 64, 6)})


def test_prologue_takes_a_supplied_physical_text(prologue, monkeypatch):
    """PhysicalVerbalEmbedder.process (:976-983): a fully annotated sample brings its reasoning text; nothing is generated, and the
    text extends the positive prompt exactly like a generated one."""
    img = tiny_vl.make_image(96, 96, 7)
    prologue._cache.clear()
    monkeypatch.setattr(prologue, "physical_text", lambda *a, **k: (_ for _ in ()).throw(AssertionError("generate() called")))
    txt = "Middle Transition Prompt: the ice softens\nFinal State Prompt: a puddle"
    posi, nega = prologue(None, prompt="melt the ice", negative_prompt="", edit_image=img, cfg=True, have_text_reasoning=True,
                          physical_txt=txt)
    assert prologue.last_physical_txt == txt
    ref = prologue.embed("melt the ice", img, txt)
    assert torch.equal(posi["prompt_emb"], ref["prompt_emb"]) and nega["prompt_emb"].shape[1] < posi["prompt_emb"].shape[1]


def test_parse_generation_response():
    ok = PP.parse_generation_response('noise {"middle_transition_prompt": " the vase tips "} trailing')
    assert ok == {"middle_transition_prompt": "the vase tips"}
    assert set(PP.parse_generation_response('{"physical_reasoning": "a", "middle_transition_prompt": "b", "final_state_prompt": "c"}')) == \
        {"physical_reasoning", "middle_transition_prompt", "final_state_prompt"}
    for bad in ("no json", '{"middle_transition_prompt": 3}', '{"Reasoning": "x", "final_state_prompt": "y"}', "{not json}"):
        with pytest.raises(ValueError):
            PP.parse_generation_response(bad)


def test_text_encoder_key_conversion():
    sd = {"visual.blocks.0.attn.qkv.weight": torch.zeros(1), "model.layers.0.mlp.up_proj.weight": torch.zeros(1),
          "model.language_model.norm.weight": torch.zeros(1), "lm_head.weight": torch.zeros(1)}
    assert set(PP.convert_text_encoder_keys(sd)) == {"model.visual.blocks.0.attn.qkv.weight", "model.language_model.layers.0.mlp.up_proj.weight",
                                                     "model.language_model.norm.weight", "lm_head.weight"}
