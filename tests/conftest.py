import os
import sys

# The oracle is torch CPU code and the fixtures under tests/golden/ are bf16 outputs of the reference run on a host: both are only
# reproducible bit for bit on ONE matmul code path.  oneDNN picks its bf16 kernels by the host's ISA (AMX tiles, avx512_bf16 dot products
# or the avx512 jit GEMM sum a K dimension in different orders: 1-ulp differences that 60 layers amplify), and splits work by thread
# count.  Round 6 found the fixtures of rounds 1 - 5 (written on an AMX host) unreproducible on the build container's new host (AVX-512
# with DL Boost only): 16 of 21 oracle tests off by an ulp here and there.  They were regenerated with the code path PINNED to what every
# AVX-512 host has, and the pins below hold for every test process (set before torch / oneDNN initialise); make_golden.py sets the same.
# Only where the bit-exact CPU tests run (no GPU device node): on the GPU box the `-m gpu` tests compare within tolerances and run the oracle
# live at sizes that want every host core and the fastest ISA.
ON_GPU_BOX = os.path.exists("/dev/kfd")
if not ON_GPU_BOX:
    os.environ.setdefault("ONEDNN_MAX_CPU_ISA", "AVX512_CORE_VNNI")
CPU_THREADS = 8      # the thread count the fixtures were written with (oneDNN partitions by it, whatever the core count)

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if not ON_GPU_BOX:
        import torch
        torch.set_num_threads(CPU_THREADS)


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden():
    from safetensors.torch import load_file
    from safetensors import safe_open
    import json

    def _load(name, with_meta=False):
        path = os.path.join(GOLDEN, name + ".safetensors")
        t = load_file(path)
        if not with_meta:
            return t
        with safe_open(path, "pt") as f:
            meta = {k: json.loads(v) for k, v in (f.metadata() or {}).items()}
        return t, meta
    return _load
