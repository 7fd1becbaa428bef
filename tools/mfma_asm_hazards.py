"""Static check of the hand-placed (asm volatile) MFMAs of flash_attn_fp8p_kernel against the two things the compiler cannot know about
them (profiles/r04_attention_notes.md section 5.2): an MFMA runs for 16 passes after its statement, so
  * nothing may WRITE its A / B source registers (the allocator considers them dead) and
  * nothing may READ or WRITE its result
before 18 wait states have passed, unless another MFMA of the wave was issued in between (in-order issue: that one waited for the pipe).
Compiles physicedit_amd/csrc/attention.hip to gfx950 assembly (no GPU needed) and walks the kernel's instruction list in layout order.
python tools/mfma_asm_hazards.py  ->  exit code 1 and a listing if a hazard is found."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_ZN2pe22flash_attn_fp8p_kernelILi8"


def regs(tok: str):
    tok = tok.strip().split()[0] if tok.strip() else ""
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def wait_states(ins: str) -> int:
    if ins.startswith("s_nop"):
        return int(ins.split()[1]) + 1
    return 1


def kernel_bodies(asm: str):
    """-> {symbol: instruction list} for every instantiation of the kernel"""
    lines = asm.split("\n")
    out = {}
    for st, l in enumerate(lines):
        if l.startswith(KERNEL) and ":" in l and not l.startswith("\t"):
            en = [i for i in range(st, len(lines)) if ".amdhsa_kernel" in lines[i] or lines[i].startswith(".Lfunc_end")][0]
            out[l.split(":")[0]] = [x.strip() for x in lines[st + 1:en] if x.strip() and not x.strip().startswith((";", ".", "_Z"))]
    return out


def scan(body):
    found = []
    for i, ins in enumerate(body):
        if not ins.startswith("v_mfma_scale"):
            continue
        ops = [t.strip() for t in ins.split(None, 1)[1].split(",")]
        dst, a, b = regs(ops[0]), regs(ops[1]), regs(ops[2])
        ws = 0
        for k in range(1, 64):
            if i + k >= len(body):
                break
            nxt = body[i + k]
            if nxt.startswith("v_mfma") or nxt.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_barrier")):
                break          # the next MFMA waited for the pipe; across a branch the layout order is not the execution order
            parts = nxt.split(None, 1)
            o = [t for t in parts[1].split(",")] if len(parts) > 1 else []
            writes = regs(o[0]) if o and nxt.startswith(("v_", "ds_read", "scratch_load", "global_load", "buffer_load")) else set()
            reads = set()
            for t in o[1:] if writes or nxt.startswith("v_") else o:
                reads |= regs(t)
            if ws < 18 and writes & (a | b):
                found.append((i, k, f"writes a source register after {ws} wait states of", ins, nxt))
            if ws < 18 and (reads | writes) & dst:
                found.append((i, k, f"touches the result after {ws} wait states of", ins, nxt))
            ws += wait_states(nxt)
    return found


def main() -> int:
    from physicedit_amd import build as B
    src = os.path.join(ROOT, "physicedit_amd", "csrc", "attention.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "attention.s")
        flags = [f for f in B.FLAGS if f != "-fPIC"] + B.EXTRA_FLAGS.get("attention.hip", [])
        r = subprocess.run([B._hipcc()] + flags + ["-S", "--cuda-device-only", "-o", out, src], capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr)
            return 2
        bodies = kernel_bodies(open(out).read())
    bad = 0
    for sym, body in bodies.items():
        n = sum(1 for l in body if l.startswith("v_mfma_scale"))
        found = scan(body)
        bad += len(found)
        print(f"{sym}: {len(body)} instructions, {n} MFMAs, {len(found)} hazard(s)")
        for i, k, what, ins, nxt in found:
            print(f"  +{k}: `{nxt[:70]}` {what} `{ins[:90]}` (instruction {i})")
    if not bodies:
        print("kernel not found in the assembly")
        return 2
    return 1 if bad else 0


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    sys.exit(main())
