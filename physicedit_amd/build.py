"""Build libphysicedit_amd.so (gfx950 only) in-tree with hipcc.

    python -m physicedit_amd.build [--force]

The .so is git-ignored (history stays source-only) but travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libphysicedit_amd.so")
SOURCES = ["api.hip", "gemm.hip", "gemm4.hip", "attention.hip", "elementwise.hip", "dit.hip", "vae.hip", "vae_graph.hip", "profile.hip"]
HEADERS = ["common.h", "kernels.h", "gemm_tile.h", "attention_w4_body.inc", "attention_w5_body.inc", "attention_w5_probe_body.inc", "attention_w7_body.inc", os.path.join("..", "..", "include", "physicedit_amd.h")]
# -ffp-contract=off is LOAD-BEARING for parity: with bf16-typed operands LLVM narrows
# float(bf16(a*b)) + float(c) to bf16 fmul/fadd and the default -ffp-contract=fast then fuses them
# into one fma, silently deleting a bf16 rounding the reference performs (measured: 29 % of
# ln_modulate outputs off by one ulp).  Fusion is written explicitly (fmaf) where it is wanted.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]
# per-source extras.  attention.hip: no SLP vectorisation -- it turns adjacent fp32 adds / muls / fmas of the softmax into v_pk_*_f32, which
# on gfx950 costs more than the two plain VALU it replaces when issued beside MFMAs
# gemm.hip (round 6): the deferred epilogue's slices run beside the other wave group's MFMAs and are written in scalar fp32 for the same reason
# (v_pk_*_f32 does not overlap with an MFMA at all); the tile-end epilogues use explicit two-element vectors and keep their packed operations
EXTRA_FLAGS = {"attention.hip": ["-fno-slp-vectorize"], "gemm.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def source_hash() -> str:
    """sha256 over the kernel sources and headers (sorted by name): baked into the library as pe_build_id() so a stale
    prebuilt .so cannot pass for the current sources (ABI_VERSION only changes with signatures)."""
    import hashlib
    hh = hashlib.sha256()
    names = sorted(SOURCES) + sorted(HEADERS)
    for n in names:
        path = os.path.join(CSRC, n)
        if os.path.exists(path):
            hh.update(n.encode())
            with open(path, "rb") as f:
                hh.update(f.read())
    return hh.hexdigest()[:16]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    src_hash = source_hash()
    stamp = os.path.join(OBJ, "src_hash.txt")
    old_hash = open(stamp).read().strip() if os.path.exists(stamp) else ""
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + hdrs) or (s == "api.hip" and old_hash != src_hash):   # api.hip carries the hash
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = ([hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), [])
               + ([f'-DPE_SRC_HASH="{src_hash}"'] if src.endswith("api.hip") else []) + ["-c", src, "-o", obj])
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    if jobs:
        if verbose:
            print(f"[physicedit_amd.build] compiling {len(jobs)} file(s) for gfx950 ...", flush=True)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(compile_one, jobs))
        with open(stamp, "w") as f:
            f.write(src_hash)
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in srcs]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[physicedit_amd.build] linked {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
