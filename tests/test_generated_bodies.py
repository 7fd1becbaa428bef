"""The generated instruction schedules under physicedit_amd/csrc/ are committed files; the build does not run their generators.  A body
that no longer matches its generator (a knob left set in the environment, an edit to one side only) is a kernel nobody described: regenerate
into a scratch directory and compare."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "physicedit_amd", "csrc")


@pytest.mark.parametrize("gen,bodies", [("gen_attn_w7.py", ["attention_w7_body.inc"]),
                                        ("gen_attn_w4.py", ["attention_w4_body.inc", "attention_w5_body.inc", "attention_w5_probe_body.inc"])])
def test_generated_body_matches_generator(tmp_path, monkeypatch, gen, bodies):
    for k in list(os.environ):
        if k.startswith("W4_") or k.startswith("W7_"):
            monkeypatch.delenv(k)
    # the generators write to <their directory>/../physicedit_amd/csrc: run a copy from a scratch tree
    tools = tmp_path / "tools"
    out = tmp_path / "physicedit_amd" / "csrc"
    tools.mkdir(); out.mkdir(parents=True)
    shutil.copy(os.path.join(ROOT, "tools", gen), tools / gen)
    spec = importlib.util.spec_from_file_location("gen_under_test", tools / gen)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main()
    for b in bodies:
        assert (out / b).read_text() == open(os.path.join(CSRC, b)).read(), f"{b} is not what tools/{gen} generates"


def test_build_tracks_generated_bodies():
    """every .inc a kernel includes is a dependency of the build (a regenerated body must trigger a recompile)"""
    from physicedit_amd import build
    incs = sorted(f for f in os.listdir(CSRC) if f.endswith(".inc"))
    assert incs and all(i in build.HEADERS for i in incs), (incs, build.HEADERS)
