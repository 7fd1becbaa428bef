"""Static check of the hand-placed (asm volatile) MFMAs of flash_attn_fp8p_kernel and flash_attn_fp8w_kernel against the things the compiler
cannot know about them (profiles/r04_attention_notes.md section 5.2): an MFMA runs for 16 passes after its statement, so
  * nothing may WRITE its A / B source registers (the allocator considers them dead) and
  * nothing may READ or WRITE its result
before 18 wait states have passed, unless another MFMA of the wave was issued in between (in-order issue: that one waited for the pipe).
flash_attn_fp8w_kernel also keeps O, Q and L in FIXED accumulator registers a[64:255] that only its asm statements name:
  * no instruction outside an asm statement (;;#ASMSTART ... ;;#ASMEND in the listing) may touch a64 ... a255.
Compiles physicedit_amd/csrc/attention.hip to gfx950 assembly (no GPU needed) and walks the kernel's instruction list in layout order.
python tools/mfma_asm_hazards.py  ->  exit code 1 and a listing if a hazard is found."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("_ZN2pe22flash_attn_fp8p_kernelILi8", "_ZN2pe22flash_attn_fp8w_kernelILb")
FIXED_AGPR = 64      # flash_attn_fp8w_kernel: a[64:255] belong to the asm statements
AGPR = 1000          # accumulator register n is register AGPR + n in the sets below


def _index(expr: str) -> int:
    """a register index as the assembler reads it: a number or an expression of numbers (the fp8w kernel writes 128+((13)&1)*64+...)"""
    if not re.fullmatch(r"[0-9+\-*&|>< ()]+", expr):
        raise ValueError(f"register index {expr!r}")
    return int(eval(expr, {"__builtins__": {}}))


def regs(tok: str):
    tok = tok.strip()
    if tok.startswith("a["):                      # accumulator registers, indices may be expressions with blanks
        inner = tok[2:tok.index("]")]
        lo, _, hi = inner.partition(":")
        lo = _index(lo)
        return set(range(AGPR + lo, AGPR + (_index(hi) if hi else lo) + 1))
    tok = tok.split()[0] if tok else ""
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"a(\d+)$", tok)
    if m:
        return {AGPR + int(m.group(1))}
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
        if l.startswith(KERNELS) and ":" in l and not l.startswith("\t"):
            en = [i for i in range(st, len(lines)) if ".amdhsa_kernel" in lines[i] or lines[i].startswith(".Lfunc_end")][0]
            out[l.split(":")[0]] = [x.strip() for x in lines[st + 1:en] if x.strip() and not x.strip().startswith((";", ".", "_Z"))]
    return out


def fixed_agprs_outside_asm(asm: str):
    """-> [(symbol, line)] : instructions of flash_attn_fp8w_kernel outside its asm statements that name a register a[FIXED_AGPR:255]"""
    found, sym, inside = [], None, False
    for l in asm.split("\n"):
        if l.startswith(KERNELS[1]) and ":" in l:
            sym = l.split(":")[0]
            continue
        if sym is None:
            continue
        if ".amdhsa_kernel" in l or l.startswith(".Lfunc_end"):
            sym = None
            continue
        s = l.strip()
        if "#ASMSTART" in s:
            inside = True
        elif "#ASMEND" in s:
            inside = False
        elif not inside and s and not s.startswith((";", ".")):
            named = [int(x) for x in re.findall(r"\ba(\d+)\b", s)] + [int(x) for x in re.findall(r"\ba\[(\d+)[:\]]", s)] + \
                    [int(x) for x in re.findall(r"\ba\[\d+:(\d+)\]", s)]
            if any(n >= FIXED_AGPR for n in named):
                found.append((sym, s))
    return found


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
        asm = open(out).read()
        bodies = kernel_bodies(asm)
    outside = fixed_agprs_outside_asm(asm)
    bad = len(outside)
    for sym, line in outside:
        print(f"{sym}: `{line[:100]}` names a fixed accumulator register outside an asm statement")
    for sym, body in bodies.items():
        n = sum(1 for l in body if l.startswith("v_mfma_scale"))
        found = scan(body)
        bad += len(found)
        print(f"{sym}: {len(body)} instructions, {n} MFMAs, {len(found)} hazard(s)")
        for i, k, what, ins, nxt in found:
            print(f"  +{k}: `{nxt[:70]}` {what} `{ins[:90]}` (instruction {i})")
    if len(bodies) < 4:          # fp8p<8, false>, fp8p<8, true>, fp8w<false>, fp8w<true>
        print("kernels not found in the assembly:", sorted(bodies))
        return 2
    return 1 if bad else 0


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    sys.exit(main())
