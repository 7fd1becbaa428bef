#!/usr/bin/env python
"""Generates physicedit_amd/csrc/attention_w4_body.inc: the instruction schedule of flash_attn_w4_kernel (attention variants 3/4).

The kernel runs ONE wave per SIMD, so nothing hides an instruction's issue slot except the MFMA that is executing: every
MFMA "gap" (32 cycles of matrix pipe) has room for ~5 other instructions of the same wave.  The schedule is therefore written
down explicitly, one asm statement per gap (MFMA + its LDS read + its slice of the softmax), with a scheduling fence after
each; statement boundaries cost padding instructions, so a gap is never split into several statements.  The tables below
say which work rides in which gap; the C++ around the statements (register arrays, rescale, state update) is in attention.hip.

    python tools/gen_attn_w4.py          # rewrites the .inc; the build does not run this (the .inc is committed)

Iteration i of the KV loop (ring slot ST = i & 3, score buffer P = ST & 1 holds tile i, P^1 tile i+1):
  phase 1, gap g: QK^T MFMA g of tile i+1  | K(i+1) fragment reads two k-steps ahead (gaps 4kk, 4kk+1), Vt(i) fragments
                  0..3 (gaps 28..31)       | the LATE score pairs of softmax(i) as one software-pipelined instruction stream
                                           | (pair_stream) cut evenly over the 32 gaps by issue slots (spread)
                                           | O rescale of blocks 0 / 1 in gaps 2 / 3 when the running max was raised
  phase 2, gap g: P.V MFMA g of tile i     | Vt(i) fragment f+4 in gap 2f, K(i+2) fragments 0..3 (gaps 28..31), LDS-DMA of
                  tile i+3 (gaps 4j+1)     | softmax(i+1): row max (gaps 0..7), m / alpha (gaps 8, 9), the EARLY pairs' stream
                                           | over gaps 10..31
LDS reads are asm and not counted by the compiler; the queue is, in issue order: phase 1  [K0..K3 issued at the end of the
phase 2 before] K4 K5 (gaps 0,1) K6 K7 (4,5) .. K14 K15 (20,21) Vt0..Vt3 (28..31); phase 2  Vt(f+4) in gap 2f (f = 0..11),
K0..K3 of the next tile (28..31).  One wait per two fragments opens the MFMA statement that first uses them: lgkmcnt(N) with
N = the reads issued after the second fragment by then (2, or 0 at the end of the queue).
"""
import os

PROBE = False     # set by main(): the folded body WITHOUT its softmax (no pairs, max, state, mask, rescale): attention_w5_probe_body.inc
FOLD = False      # set by main(): False -> attention_w4_body.inc (variants 3 / 4), True -> attention_w5_body.inc (variants 5 / 6)
DOT2 = False      # set by main() when W4_DOT2=1 (round 6's experiment, FOLD only): also writes attention_w9_body.inc -- the row sums taken from the PACKED
                  # bf16 pair with one v_dot2c_f32_bf16 (ps += p0 * 1 + p1 * 1) instead of two v_add_f32 on the fp32 exponentials: 32 issue slots of
                  # ~600 per tile less, l sums exactly the P the numerator multiplies (as variant 7's ones . P does) -- and 7 % slower, because a dot
                  # instruction does not overlap with an MFMA (profiles/r06_attention_notes.md, r06_valu_rate.log).  The library does not include it.
DMA_SLOTS = int(os.environ.get("W4_DMA_SLOTS", "3"))      # issue slots an LDS-DMA piece (address + m0 + buffer_load ... lds) is booked with
EARLY_PAIRS = int(os.environ.get("W4_EARLY_PAIRS", "5"))   # per block: pairs 0..4 of softmax(i+1) run in phase 2 of iteration i, pairs 5..15 in phase 1 of i+1
# FOLD: where in phase 1 the O rescale branches sit (a knob: any gap of phase 1 is legal -- after P.V(i), before P.V(i+1))
PK2 = bool(os.environ.get("W4_PK2"))       # experiment: FOLD with the double-buffered P of the exact form
RESCALE_GAP = int(os.environ.get("W4_RESCALE_GAP", "2"))
KT_BYTES = 64 * 256


def distribute(n_items, first_gap, last_gap):
    """gap of item k when n_items are spread evenly over first_gap..last_gap (inclusive)"""
    span = last_gap - first_gap + 1
    return [first_gap + (k * span) // n_items for k in range(n_items)]


class Stmt:
    def __init__(self):
        self.lines, self.outs, self.ins, self.after = [], [], [], []

    def emit(self, ind):
        if not self.lines:
            return [ind + a for a in self.after]
        body = [ind + 'asm volatile("' + self.lines[0] + ('\\n\\t"' if len(self.lines) > 1 else '"')]
        for k, ln in enumerate(self.lines[1:]):
            last = k == len(self.lines) - 2
            body.append(ind + '             "' + ln + ('"' if last else '\\n\\t"'))
        body.append(ind + "             : " + ", ".join(self.outs))
        body.append(ind + "             : " + ", ".join(self.ins) + ");")
        return body + [ind + a for a in self.after]


def score(P, b, f):
    return f"sc[{P}][{b}][{f >> 4}][{f & 15}]"


def pk_slot(P, b, q):
    f = 2 * q
    s2, r0 = f >> 4, f & 15
    if FOLD and not PK2:        # P is single buffered: a chunk is rewritten only after the last P.V MFMA that reads it (checked in gen_iter)
        return f"pk[{b}][{s2 * 2 + (r0 >> 3)}][{(r0 & 7) >> 1}]"
    return f"pk[{P}][{b}][{s2 * 2 + (r0 >> 3)}][{(r0 & 7) >> 1}]"


def pk_chunk(q):
    f = 2 * q
    return (f >> 4) * 2 + ((f & 15) >> 3)


def add_read(st, kind, slot, idx):
    """kind 'k': fragment idx = kk*2 + s2 of the K tile in ring slot `slot`; 'v': idx = c*4 + dt of the Vt tile"""
    if os.environ.get("W4_NO_READS"):       # timing experiment only
        return
    if kind == "k":
        off, dst, addr = slot * KT_BYTES + (idx & 1) * 32 * 256, f"kf[{idx}]", f"kaddr[{idx >> 1}]"
    else:
        off, dst, addr = slot * KT_BYTES + (idx & 3) * 32 * 128, f"vf[{idx}]", f"vaddr[{idx >> 2}]"
    st.lines.append(f"ds_read_b128 %[rd], %[ra] offset:{off}")
    st.outs.append(f'[rd] "=a"({dst})')
    st.ins.append(f'[ra] "v"({addr})')


class Ins:
    """one filler instruction of the softmax stream: asm text over %[name] operands, and what it does to the C++ variables behind them"""

    def __init__(self, text, defs=(), uses=(), rmw=(), slots=1, after=None, sreg=()):
        self.text, self.defs, self.uses, self.rmw, self.slots, self.after, self.sreg = text, dict(defs), dict(uses), dict(rmw), slots, after, set(sreg)


def pair_stream(pairs, P, tagp):
    """The softmax of `pairs` = [(b, q)] (scores 2q, 2q+1 of block b, tile parity P) as ONE instruction stream, software-pipelined one
    pair deep:   fma fma | add' exp add' exp cvt'   (' = the pair before; FOLD: exp add' exp add' cvt', the fmas are gone), so that no
    instruction reads the result of the one right before it.  Issue slots: v_exp_f32 2 (transcendental rate), everything else 1 (tools/microbench/valu_rate.hip)."""
    if os.environ.get("W4_NO_PAIRS") or PROBE:       # timing experiment / the mix probe
        return []

    def parts(k):
        b, q = pairs[k]
        T = f"{tagp}{k}"
        if FOLD:        # the accumulator already holds s . c - m (Q pre-scaled by c, -m fed through the MFMA's C operand)
            F = []
            E = [Ins(f"v_exp_f32 %[e{j}_{T}], %[sc{j}_{T}]", defs={f"e{j}_{T}": f"e{j}_{T}"}, uses={f"sc{j}_{T}": score(P, b, 2 * q + j)}, slots=2)
                 for j in (0, 1)]
        else:
            F = [Ins(f"v_fma_f32 %[e{j}_{T}], %[sc{j}_{T}], %[sl], -%[sub{b}]", defs={f"e{j}_{T}": f"e{j}_{T}"},
                     uses={f"sc{j}_{T}": score(P, b, 2 * q + j), "sl": "scale_log2", f"sub{b}": f"sm_sub[{b}]"}, sreg=("sl",)) for j in (0, 1)]
            E = [Ins(f"v_exp_f32 %[e{j}_{T}], %[e{j}_{T}]", rmw={f"e{j}_{T}": f"e{j}_{T}"}, slots=2) for j in (0, 1)]
        S = [Ins(f"v_add_f32 %[ps{b}], %[ps{b}], %[e{j}_{T}]", rmw={f"ps{b}": f"sm_psum[{b}]"}, uses={f"e{j}_{T}": f"e{j}_{T}"}) for j in (0, 1)]
        if DOT2:        # one instruction on the packed pair (0x3f803f80 = bf16 1.0, 1.0)
            S = [Ins(f"v_dot2c_f32_bf16 %[ps{b}], 0x3f803f80, %[pw_{T}]", rmw={f"ps{b}": f"sm_psum[{b}]"}, uses={f"pw_{T}": f"pw_{T}"})]
        C = Ins(f"v_cvt_pk_bf16_f32 %[pw_{T}], %[e0_{T}], %[e1_{T}]", defs={f"pw_{T}": f"pw_{T}"},
                uses={f"e0_{T}": f"e0_{T}", f"e1_{T}": f"e1_{T}"}, after=f"{pk_slot(P, b, q)} = pw_{T};")
        return F, E, S, C
    out = []
    prev = None
    for k in range(len(pairs)):
        F, E, S, C = parts(k)
        if prev is None:
            out += F + [E[0], E[1]]
        elif DOT2:      # exp cvt' exp dot2' : the convert reads exponentials three instructions old, the dot product a word two instructions old
            pS, pC = prev
            out += [E[0], pC, E[1], pS[0]]
        else:
            pS, pC = prev
            out += [E[0], pS[0], E[1], pS[1], pC] if FOLD else F + [pS[0], E[0], pS[1], E[1], pC]
        prev = (S, C)
    if prev is not None:
        # (DOT2: the stream's last convert would read an exponential only two instructions old: it waits, as variant 7's does)
        out += [Ins("s_nop 1"), prev[1], prev[0][0]] if DOT2 else [prev[0][0], prev[0][1], prev[1]]
    return out


def spread(stream, fixed, first_gap, last_gap):
    """Cut `stream` into the gaps first_gap..last_gap so that every gap carries about the same number of issue slots, counting what
    the gap holds already (`fixed[g]`: its LDS reads, max3s, ...).  Returns {gap: [Ins]}.  An MFMA gap hides ~5-7 slots; a gap
    loaded beyond that stretches the MFMA cadence, and a lighter gap next to it gives nothing back (the matrix pipe cannot run
    ahead), so the level matters more than the sum."""
    gaps = list(range(first_gap, last_gap + 1))
    total = sum(i.slots for i in stream) + sum(fixed.get(g, 0) for g in gaps)
    res = {g: [] for g in gaps}
    k = 0
    used_total = 0.0
    for n, g in enumerate(gaps):
        # cumulative target after this gap
        target = total * (n + 1) / len(gaps)
        used_total += fixed.get(g, 0)
        while k < len(stream) and (used_total + stream[k].slots / 2.0 <= target or n == len(gaps) - 1):
            res[g].append(stream[k])
            used_total += stream[k].slots
            k += 1
    assert k == len(stream)
    return res


def add_stream(st, instrs):
    """append `instrs` to statement st: every C++ variable becomes ONE operand -- "=&v" if the statement defines it before any
    read, "+v" if it reads and writes it, "v" / "s" if it only reads it"""
    state = {}      # name -> [cexpr, first_access ('def' | 'use'), written]
    order = []
    for i in instrs:
        for nm, ce in i.uses.items():
            if nm not in state:
                state[nm] = [ce, "use", False, nm in i.sreg]
                order.append(nm)
        for nm, ce in i.rmw.items():
            if nm not in state:
                state[nm] = [ce, "use", True, False]
                order.append(nm)
            else:
                state[nm][2] = True
        for nm, ce in i.defs.items():
            if nm not in state:
                state[nm] = [ce, "def", True, False]
                order.append(nm)
            else:
                state[nm][2] = True
        st.lines.append(i.text)
        if i.after:
            st.after.append(i.after)
    for nm in order:
        ce, first, written, sreg = state[nm]
        if first == "def":
            st.outs.append(f'[{nm}] "=&v"({ce})')
        elif written:
            st.outs.append(f'[{nm}] "+v"({ce})')
        else:
            st.ins.append(f'[{nm}] "{"s" if sreg else "v"}"({ce})')


def fixed_slots(st):
    return sum(0 if ln.startswith("s_waitcnt") or ln.startswith("v_mfma") else 1 for ln in st.lines)


def add_max(st, P, g):
    """running max over scores 4g..4g+3 of both blocks"""
    if PROBE:
        return
    for b in (0, 1):
        e = [score(P, b, 4 * g + k) for k in range(4)]
        if g == 0:
            st.outs.append(f'[mx{b}] "=&v"(sm_mx[{b}])')
        else:
            st.outs.append(f'[mx{b}] "+v"(sm_mx[{b}])')
        st.ins += [f'[m{b}{k}] "v"({e[k]})' for k in range(4)]
    if g == 0:
        st.lines += ["v_max3_f32 %[mx0], %[m00], %[m01], %[m02]", "v_max3_f32 %[mx1], %[m10], %[m11], %[m12]",
                     "v_max_f32 %[mx0], %[mx0], %[m03]", "v_max_f32 %[mx1], %[mx1], %[m13]"]
    else:
        st.lines += ["v_max3_f32 %[mx0], %[mx0], %[m00], %[m01]", "v_max3_f32 %[mx1], %[mx1], %[m10], %[m11]",
                     "v_max3_f32 %[mx0], %[mx0], %[m02], %[m03]", "v_max3_f32 %[mx1], %[mx1], %[m12], %[m13]"]


def qk_stmt(P, g, slot_k, ahead=True, negm=None):
    """QK^T MFMA g of the tile whose K sits in ring slot slot_k: kk = g >> 2, q block (g >> 1) & 1, key half g & 1.
    negm (FOLD, main loop): the first k-step accumulates onto negm[b] -- 16 registers holding -m of the lane's row -- instead of 0"""
    if negm is None:
        negm = FOLD
    st = Stmt()
    kk, b, s2 = g >> 2, (g >> 1) & 1, g & 1
    if g & 3 == 0:
        st.lines.append(f"s_waitcnt lgkmcnt({2 if kk <= 6 else 0})")
    acc = f"sc[{P}][{b}][{s2}]"
    if kk == 0 and negm:
        st.lines.append("v_mfma_f32_32x32x16_bf16 %[acc], %[fa], %[fb], %[nm]")
        st.outs.append(f'[acc] "=&v"({acc})')
        st.ins.append(f'[nm] "v"(negm[{b}])')
    elif kk == 0:
        st.lines.append("v_mfma_f32_32x32x16_bf16 %[acc], %[fa], %[fb], 0")
        st.outs.append(f'[acc] "=&v"({acc})')
    else:
        st.lines.append("v_mfma_f32_32x32x16_bf16 %[acc], %[fa], %[fb], %[acc]")
        st.outs.append(f'[acc] "+v"({acc})')
    st.ins += [f'[fa] "a"(kf[{kk * 2 + s2}])', f'[fb] "a"(qf[{b}][{kk}])']
    if ahead and (g & 3) < 2 and kk <= 5:
        add_read(st, "k", slot_k, (kk + 2) * 2 + (g & 1))
    return st


def pv_stmt(PC, g):
    """P.V MFMA g: Vt fragment f = g >> 1 (key chunk f >> 2, dt = f & 3), q block g & 1"""
    st = Stmt()
    f, b = g >> 1, g & 1
    if g & 3 == 0:
        st.lines.append(f"s_waitcnt lgkmcnt({2 if f <= 12 else 0})")
    st.lines.append("v_mfma_f32_32x32x16_bf16 %[acc], %[fa], %[fb], %[acc]")
    st.outs.append(f'[acc] "+a"(o[{b}][{f & 3}])')
    st.ins += [f'[fa] "a"(vf[{f}])', f'[fb] "v"(pk[{b}][{f >> 2}])' if FOLD and not PK2 else f'[fb] "v"(pk[{PC}][{b}][{f >> 2}])']
    return st


def pair_list(lo, hi):
    return [(b, q) for q in range(lo, hi) for b in (0, 1)]


def gen_iter(ST, out):
    PC, PN = ST & 1, (ST & 1) ^ 1
    slot_v, slot_k1, slot_k2, slot_d = ST, (ST + 1) & 3, (ST + 2) & 3, (ST + 3) & 3
    ind = "        "
    w = out.append
    w(f"    auto iter{ST} = [&](int i) __attribute__((always_inline)) {{")
    w(ind + f"// tile i: ring slot {slot_v}, scores sc[{PC}], P pk[{PC}];  tile i+1: K in slot {slot_k1}, scores sc[{PN}]")
    w(ind + "const int t_next = t_begin + i + 1;")
    w(ind + "const bool live_next = i + 1 < n;")
    w(ind + "const bool mask_next = (t_next + 1) * KV_TILE > S;      // ragged last tile, or past the end: no key of it may count")
    if ST == 0:
        w("#if PE_W4_STAMPS")
        w(ind + "const bool stamp = i == 8;")
        w(ind + "if (stamp) stamp_it[0] = (long long)__builtin_readcyclecounter();")
        w("#endif")
    late = pair_list(EARLY_PAIRS, 16)
    w(ind + "float " + ", ".join(f"e0_l{k}, e1_l{k}" for k in range(len(late))) + ";")
    w(ind + "uint32_t " + ", ".join(f"pw_l{k}" for k in range(len(late))) + ";")
    w(ind + "// ---- phase 1")
    stmts = []
    for g in range(32):
        st = qk_stmt(PN, g, slot_k1)
        if g >= 28:
            add_read(st, "v", slot_v, g - 28)
        stmts.append(st)
    fixed = {g: fixed_slots(stmts[g]) for g in range(32)}
    rg = RESCALE_GAP if FOLD else 2         # the (rarely taken) O rescale branches sit behind gaps rg, rg + 1
    fixed[rg] += 2; fixed[rg + 1] += 2
    placed = spread(pair_stream(late, PC, "l"), fixed, 0, 31)
    for g in range(32):
        st = stmts[g]
        add_stream(st, placed[g])
        out.extend(st.emit(ind))
        if g == rg and not PROBE:
            w(ind + "rescale(0);")
        if g == rg + 1 and not PROBE:
            w(ind + "rescale(1);")
        if g == 31 and not PROBE:
            w(ind + "l_run[0] = __builtin_fmaf(l_run[0], sm_alpha[0], sm_psum[0]);")
            w(ind + "l_run[1] = __builtin_fmaf(l_run[1], sm_alpha[1], sm_psum[1]);")
        w(ind + "W4_FENCE();")
    if ST == 0:
        w("#if PE_W4_STAMPS")
        w(ind + "if (stamp) stamp_it[1] = (long long)__builtin_readcyclecounter();")
        w("#endif")
    w(ind + "// own pieces of tile i+2 landed (issued one iteration ago), then everyone's; every wave is past its reads of")
    w(ind + "// tile i-1's Vt and K, so ring slot (i+3) & 3 can be refilled")
    w(ind + 'asm volatile("s_waitcnt vmcnt(0)" ::: "memory");')
    w(ind + "W4_FENCE();")
    w(ind + "__builtin_amdgcn_s_barrier();")
    w(ind + "W4_FENCE();")
    if ST == 0:
        w("#if PE_W4_STAMPS")
        w(ind + "if (stamp) stamp_it[2] = (long long)__builtin_readcyclecounter();")
        w("#endif")
    early = pair_list(0, EARLY_PAIRS)
    w(ind + "float " + ", ".join(f"e0_e{k}, e1_e{k}" for k in range(len(early))) + ";")
    w(ind + "uint32_t " + ", ".join(f"pw_e{k}" for k in range(len(early))) + ";")
    w(ind + "// ---- phase 2")
    stmts = []
    for g in range(32):
        st = pv_stmt(PC, g)
        f = g >> 1
        if g & 1 == 0 and f + 4 < 16:
            add_read(st, "v", slot_v, f + 4)
        if g >= 28:
            add_read(st, "k", slot_k2, g - 28)
        if g < 8:
            add_max(st, PN, g)
        stmts.append(st)
    fixed = {g: fixed_slots(stmts[g]) for g in range(32)}
    for g in range(32):
        if g & 3 == 1:
            fixed[g] += DMA_SLOTS           # an LDS-DMA piece is issued behind these gaps
    first_pair_gap = int(os.environ.get("W4_FIRST_EARLY_GAP", "10"))        # sm_state runs behind gaps 8 and 9
    placed = spread(pair_stream(early, PN, "e"), fixed, first_pair_gap, 31)
    if FOLD and not PK2:        # single-buffered P: chunk c of tile i is last read by P.V MFMA 8c + 7; tile i+1's packs into it must sit in a later gap
        for g, instrs in placed.items():
            for ins in instrs:
                if ins.after:
                    c = int(ins.after.split("]")[1][1:])
                    assert g > 8 * c + 7, f"early pack into chunk {c} in gap {g}: P.V still reads it"
    for g in range(32):
        st = stmts[g]
        if PROBE:
            pass
        elif g == 0 and FOLD:     # a tile past the end of this work item (iterations come in fours) has no sm_sub = inf to zero its P: every key is masked
            w(ind + f"if (mask_next || !live_next) mask_scores(std::integral_constant<int, {PN}>{{}}, t_next, live_next ? S : 0);")
        elif g == 0:
            w(ind + f"if (mask_next) mask_scores(std::integral_constant<int, {PN}>{{}}, t_next, S);")
        add_stream(st, placed.get(g, []))
        out.extend(st.emit(ind))
        if g & 3 == 1 and not os.environ.get("W4_NO_DMA"):       # W4_NO_DMA: timing experiment only (results are garbage)
            j = g >> 3
            w(ind + (f"stage_v({slot_d}, i + 3, {j});" if (g >> 2) & 1 else f"stage_k({slot_d}, i + 3, {j});"))
        if g in (8, 9) and PROBE:
            pass
        elif g in (8, 9) and FOLD:
            w(ind + f"sm_state_f(std::integral_constant<int, {PN}>{{}}, {g - 8}, std::false_type{{}});")
        elif g in (8, 9):
            w(ind + f"sm_state({g - 8}, live_next);")
        w(ind + "W4_FENCE();")
    if ST == 0:
        w("#if PE_W4_STAMPS")
        w(ind + "if (stamp) stamp_it[3] = (long long)__builtin_readcyclecounter();")
        w("#endif")
    w("    };")
    w("")


def gen_prologue(out):
    ind = "    "
    w = out.append
    w(ind + "// ---- prologue: tiles 0..2 in flight, QK^T(0), first K(1) fragments, first part of softmax(0)")
    w(ind + "stage(0, 0); stage(1, 1); stage(2, 2);")
    w(ind + 'asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // tile 0 landed')
    w(ind + "W4_FENCE();")
    w(ind + "__builtin_amdgcn_s_barrier();")
    w(ind + "W4_FENCE();")
    for idx in range(4):
        st = Stmt()
        add_read(st, "k", 0, idx)
        out.extend(st.emit(ind))
    w(ind + "W4_FENCE();")
    for g in range(32):
        out.extend(qk_stmt(0, g, 0, negm=False).emit(ind))
        w(ind + "W4_FENCE();")
    w(ind + 'asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // tile 1 landed')
    w(ind + "W4_FENCE();")
    w(ind + "__builtin_amdgcn_s_barrier();")
    w(ind + "W4_FENCE();")
    for idx in range(4):
        st = Stmt()
        add_read(st, "k", 1, idx)
        out.extend(st.emit(ind))
    w(ind + 'asm volatile("s_nop 15");                           // last QK^T MFMAs -> the score reads below')
    w(ind + "W4_FENCE();")
    if not PROBE:
        w(ind + "if ((t_begin + 1) * KV_TILE > S) mask_scores(std::integral_constant<int, 0>{}, t_begin, S);")
    early = pair_list(0, EARLY_PAIRS)
    w(ind + "{")
    ind2 = ind + "    "
    w(ind2 + "float " + ", ".join(f"e0_p{k}, e1_p{k}" for k in range(len(early))) + ";")
    w(ind2 + "uint32_t " + ", ".join(f"pw_p{k}" for k in range(len(early))) + ";")
    for g in range(8):
        st = Stmt()
        add_max(st, 0, g)
        out.extend(st.emit(ind2))
        w(ind2 + "W4_FENCE();")
    if PROBE:
        pass
    elif FOLD:
        w(ind2 + "sm_state_f(std::integral_constant<int, 0>{}, 0, std::true_type{});")
        w(ind2 + "sm_state_f(std::integral_constant<int, 0>{}, 1, std::true_type{});")
    else:
        w(ind2 + "sm_state(0, true);")
        w(ind2 + "sm_state(1, true);")
    w(ind2 + "W4_FENCE();")
    stream = pair_stream(early, 0, "p")
    step = 5 if FOLD else 7                     # one pipeline step per statement
    for k in range(0, len(stream), step):
        st = Stmt()
        add_stream(st, stream[k:k + step])
        out.extend(st.emit(ind2))
        w(ind2 + "W4_FENCE();")
    w(ind + "}")
    w("")


def main():
    global FOLD, PROBE, DOT2
    here = os.path.dirname(os.path.abspath(__file__))
    only = os.environ.get("W4_ONLY")        # "w4" / "w5": regenerate one of the bodies (knob sweeps)
    for fold, probe, dot2, name, kern in ((False, False, False, "attention_w4_body.inc", "flash_attn_w4_kernel<false> (variants 3 / 4)"),
                                          (True, False, False, "attention_w5_body.inc", "flash_attn_w4_kernel<true> (variants 5 / 6: folded scale and max)"),
                                          (True, False, True, "attention_w9_body.inc", "flash_attn_w4_kernel<true, 2> (variants 9 / 10: 5 / 6 with the row sums "
                                           "taken from the packed bf16 pairs by v_dot2c_f32_bf16)"),
                                          (True, True, False, "attention_w5_probe_body.inc", "flash_attn_w4_kernel<true, 1>: the folded schedule without "
                                           "its softmax (pe_attn_mix_probe: MFMAs, LDS fragment reads, LDS-DMA stream, barrier)")):
        if only and (only != name[10:12] or probe):
            continue
        if dot2 and not os.environ.get("W4_DOT2"):
            continue
        FOLD, PROBE, DOT2 = fold, probe, dot2
        out = ["// GENERATED by tools/gen_attn_w4.py -- do not edit; the schedule tables and their rationale are in that script.",
               f"// Included inside {kern} (attention.hip), which declares every name used here.", ""]
        for st in range(4):
            gen_iter(st, out)
        gen_prologue(out)
        path = os.path.join(here, "..", "physicedit_amd", "csrc", name)
        with open(path, "w") as f:
            f.write("\n".join(out) + "\n")
        print(f"wrote {os.path.normpath(path)}: {len(out)} lines")


if __name__ == "__main__":
    main()
