#!/usr/bin/env python
"""Generates physicedit_amd/csrc/attention_w7_body.inc: the instruction schedule of flash_attn_w7_kernel (attention variant 7): the
folded one-wave-per-SIMD schedule of variants 5 / 6 (tools/gen_attn_w4.py) on v_mfma_f32_16x16x32_bf16 -- the 4-pass MFMA shape that
sustains 12 - 16 % more FLOP/s under the chip's power limit than 32x32x16 (profiles/r05_gemm_notes.md section 7).

    python tools/gen_attn_w7.py          # rewrites the .inc; the build does not run this (the .inc is committed)

Tiling of a wave (64 query rows x a 64-key tile), all blocks 16 x 16:
  scores   sc[P][qb][m]  (f32x4)   qb = query block 0..3 (query row qb * 16 + (lane & 15)), m = MFMA key block 0..3; element r of lane group
                                   g = lane >> 4 is MFMA row 4g + r = key 32 (m >> 1) + 16 (g >> 1) + 8 (m & 1) + 4 (g & 1) + r of the tile:
                                   the K fragment of block m takes row (l & 7) + 16 (l >> 3) + 32 (m >> 1) + 8 (m & 1) for MFMA row l, so
                                   that the 8 k-slots lane group g feeds the P.V MFMA of chunk c = m >> 1 (4 keys of block 2c, then 4 of
                                   block 2c + 1) are 8 CONSECUTIVE positions of Vt's perm16 layout: one ds_read_b128 per Vt fragment
  P        pk[qb][c]     (u32x4)   the B operand of the P.V MFMAs of 32-key chunk c: bf16 pairs of sc[qb][2c][0..3], sc[qb][2c+1][0..3]
  O^T      o[qb][db]     (f32x4)   db = 16-wide d block 0..7
A "gap" is TWO MFMAs (2 x 16 cycles of matrix pipe = the 32 cycles of one 32x32x16 MFMA) that share an operand, with the gap's LDS
reads and its slice of the softmax behind them, as one asm statement:
  phase 1, gap g: QK^T of tile i+1, K fragment pair p = g >> 2 (k-step p >> 1 of 32 d, key blocks 2 (p & 1), 2 (p & 1) + 1), query block g & 3
                  K(i+1) fragment pair p + 2 in gaps 4p, 4p+1, Vt(i) fragments 0..V_AHEAD-1 in the last gaps; the LATE pairs of softmax(i)
  phase 2, gap g: P.V of tile i, chunk c = g >> 4, d block (g >> 1) & 7, query blocks 2 (g & 1), 2 (g & 1) + 1
                  Vt(i) fragment f + V_AHEAD in gap 2f, K(i+2) fragments 0..3 in gaps 28..31, LDS-DMA of tile i+3 behind gaps 4j+1;
                  softmax(i+1): row max in gaps 0..7, m / alpha behind gaps 8..11, the EARLY pairs in gaps 16..31 (P is single
                  buffered: chunk 0 of P(i) is read until gap 15)
The LDS queue is modelled: every read is appended in issue order, and a statement that uses a fragment opens with s_waitcnt lgkmcnt(N),
N = the reads issued after that fragment's.
"""
import os

DMA_SLOTS = int(os.environ.get("W7_DMA_SLOTS", "3"))
EARLY_PAIRS = int(os.environ.get("W7_EARLY_PAIRS", "2"))      # per query block (8 pairs each): pairs 0..E-1 of softmax(i+1) run in phase 2 of iteration i
RESCALE_GAP = int(os.environ.get("W7_RESCALE_GAP", "2"))
FIRST_EARLY_GAP = int(os.environ.get("W7_FIRST_EARLY_GAP", "16"))
V_AHEAD = int(os.environ.get("W7_V_AHEAD", "3"))              # Vt fragments in flight ahead of the P.V MFMAs (each is used in two gaps)
KT_BYTES = 64 * 256
MFMA = "v_mfma_f32_16x16x32_bf16"


class Stmt:
    """one gap: [s_waitcnt] MFMA filler.. MFMA filler..  -- the fillers (LDS reads, softmax slice) are cut evenly behind the two MFMAs: a 4-pass
    MFMA shadows 16 cycles, and the wave issues in order, so fillers queued behind BOTH MFMAs would wait for the second one's turn on the pipe"""

    def __init__(self):
        self.lines, self.outs, self.ins, self.after = [], [], [], []

    def emit(self, ind):
        if not self.lines:
            return [ind + a for a in self.after]
        assert len(self.outs) + len(self.ins) <= 30, f"asm statement with {len(self.outs) + len(self.ins)} operands"
        pre = [ln for ln in self.lines if ln.startswith("s_waitcnt")]
        mf = [ln for ln in self.lines if ln.startswith("v_mfma")]
        fill = [ln for ln in self.lines if not ln.startswith("s_waitcnt") and not ln.startswith("v_mfma")]
        if len(mf) >= 2 and not os.environ.get("W7_NO_SPLIT"):
            lines = list(pre)
            for k, m in enumerate(mf):
                lo, hi = (len(fill) * k + len(mf) - 1) // len(mf), (len(fill) * (k + 1) + len(mf) - 1) // len(mf)
                lines += [m] + fill[lo:hi]
        else:
            lines = pre + mf + fill
        body = [ind + 'asm volatile("' + lines[0] + ('\\n\\t"' if len(lines) > 1 else '"')]
        for k, ln in enumerate(lines[1:]):
            last = k == len(lines) - 2
            body.append(ind + '             "' + ln + ('"' if last else '\\n\\t"'))
        body.append(ind + "             : " + ", ".join(self.outs))
        body.append(ind + "             : " + ", ".join(self.ins) + ");")
        return body + [ind + a for a in self.after]


class LdsQueue:
    """issue-order model of the wave's LDS reads: wait(name) -> the lgkmcnt that guarantees `name` has landed, or None if a wait issued
    earlier already covers it"""

    def __init__(self, seed):
        self.q = list(seed)
        self.done = -1          # reads up to this index are known to have landed

    def issue(self, name):
        self.q.append(name)

    def wait(self, names):
        idx = max(len(self.q) - 1 - self.q[::-1].index(n) for n in names)
        if idx <= self.done:
            return None
        self.done = idx
        n = len(self.q) - 1 - idx
        assert n <= 15
        return n


def score(P, qb, f):
    return f"sc[{P}][{qb}][{f >> 2}][{f & 3}]"


def pk_slot(qb, q):
    return f"pk[{qb}][{q >> 2}][{q & 3}]"


def add_read(st, lq, kind, slot, idx):
    """kind 'k': fragment idx = kk * 4 + m of the K tile in ring slot `slot`; 'v': idx = c * 8 + db of the Vt tile"""
    if kind == "k":
        kk, m = idx >> 2, idx & 3
        off, dst, addr = slot * KT_BYTES + (32 * (m >> 1) + 8 * (m & 1)) * 256, f"kf[{idx}]", f"kaddr[{kk}]"
    else:
        c, db = idx >> 3, idx & 7
        off, dst, addr = slot * KT_BYTES + db * 16 * 128, f"vf[{idx}]", f"vaddr[{c}]"
    n = sum(1 for ln in st.lines if ln.startswith("ds_read"))
    st.lines.append(f"ds_read_b128 %[rd{n}], %[ra{n}] offset:{off}")
    st.outs.append(f'[rd{n}] "=a"({dst})')
    st.ins.append(f'[ra{n}] "v"({addr})')
    lq.issue(f"{kind}{idx}")


class Ins:
    def __init__(self, text, defs=(), uses=(), rmw=(), slots=1, after=None):
        self.text, self.defs, self.uses, self.rmw, self.slots, self.after = text, dict(defs), dict(uses), dict(rmw), slots, after


def pair_stream(pairs, P, tagp):
    """The softmax of `pairs` = [(qb, q)] (scores 2q, 2q+1 of query block qb, tile parity P) as one instruction stream, software-pipelined
    one pair deep: exp exp cvt' (' = the pair before); the accumulator already holds s . c - m.  v_exp_f32 takes 2 issue slots.  No row sums
    here: l is accumulated by the matrix pipe (pv_stmt: ones x P)."""
    if os.environ.get("W7_NO_PAIRS"):
        return []

    def parts(k):
        b, q = pairs[k]
        T = f"{tagp}{k}"
        E = [Ins(f"v_exp_f32 %[e{j}_{T}], %[sc{j}_{T}]", defs={f"e{j}_{T}": f"e{j}_{T}"}, uses={f"sc{j}_{T}": score(P, b, 2 * q + j)}, slots=2) for j in (0, 1)]
        C = Ins(f"v_cvt_pk_bf16_f32 %[pw_{T}], %[e0_{T}], %[e1_{T}]", defs={f"pw_{T}": f"pw_{T}"},
                uses={f"e0_{T}": f"e0_{T}", f"e1_{T}": f"e1_{T}"}, after=f"{pk_slot(b, q)} = pw_{T};")
        return E, C
    # order: exp exp cvt' -- a convert reads exps issued at least three instructions earlier (a VALU instruction that reads the result of a
    # transcendental right behind it gets the OLD register value on gfx950: no interlock, and nobody inserts the wait state inside an asm
    # statement; measured: the last pair of every stream came out wrong with exp cvt back to back); the stream's last convert waits explicitly
    out = []
    prev = None
    for k in range(len(pairs)):
        E, C = parts(k)
        out += [E[0], E[1]]
        if prev is not None:
            out.append(prev)
        prev = C
    if prev is not None:
        out += [Ins("s_nop 1", slots=1), prev]
    return out


def spread(stream, fixed, first_gap, last_gap):
    gaps = list(range(first_gap, last_gap + 1))
    total = sum(i.slots for i in stream) + sum(fixed.get(g, 0) for g in gaps)
    res = {g: [] for g in gaps}
    k = 0
    used_total = 0.0
    for n, g in enumerate(gaps):
        target = total * (n + 1) / len(gaps)
        used_total += fixed.get(g, 0)
        while k < len(stream) and (used_total + stream[k].slots / 2.0 <= target or n == len(gaps) - 1):
            res[g].append(stream[k])
            used_total += stream[k].slots
            k += 1
    assert k == len(stream)
    return res


def add_stream(st, instrs):
    state = {}
    order = []
    for i in instrs:
        for nm, ce in i.uses.items():
            if nm not in state:
                state[nm] = [ce, "use", False]
                order.append(nm)
        for nm, ce in i.rmw.items():
            if nm not in state:
                state[nm] = [ce, "use", True]
                order.append(nm)
            else:
                state[nm][2] = True
        for nm, ce in i.defs.items():
            if nm not in state:
                state[nm] = [ce, "def", True]
                order.append(nm)
            else:
                state[nm][2] = True
        st.lines.append(i.text)
        if i.after:
            st.after.append(i.after)
    for nm in order:
        ce, first, written = state[nm]
        if first == "def":
            st.outs.append(f'[{nm}] "=&v"({ce})')
        elif written:
            st.outs.append(f'[{nm}] "+v"({ce})')
        else:
            st.ins.append(f'[{nm}] "v"({ce})')


def fixed_slots(st):
    return sum(0 if ln.startswith("s_waitcnt") else 1 for ln in st.lines)


def add_max(st, P, gi):
    """running max, gap gi of 8: step s = gi >> 1 (the 4 scores of key block s) of query blocks 2 (gi & 1), 2 (gi & 1) + 1"""
    s = gi >> 1
    blocks = (2 * (gi & 1), 2 * (gi & 1) + 1)
    for b in blocks:
        e = [score(P, b, 4 * s + k) for k in range(4)]
        st.outs.append(f'[mx{b}] "{"=&v" if s == 0 else "+v"}"(sm_mx[{b}])')
        st.ins += [f'[m{b}{k}] "v"({e[k]})' for k in range(4)]
    a, b = blocks
    if s == 0:
        st.lines += [f"v_max3_f32 %[mx{a}], %[m{a}0], %[m{a}1], %[m{a}2]", f"v_max3_f32 %[mx{b}], %[m{b}0], %[m{b}1], %[m{b}2]",
                     f"v_max_f32 %[mx{a}], %[mx{a}], %[m{a}3]", f"v_max_f32 %[mx{b}], %[mx{b}], %[m{b}3]"]
    else:
        st.lines += [f"v_max3_f32 %[mx{a}], %[mx{a}], %[m{a}0], %[m{a}1]", f"v_max3_f32 %[mx{b}], %[mx{b}], %[m{b}0], %[m{b}1]",
                     f"v_max3_f32 %[mx{a}], %[mx{a}], %[m{a}2], %[m{a}3]", f"v_max3_f32 %[mx{b}], %[mx{b}], %[m{b}2], %[m{b}3]"]


def qk_stmt(lq, P, g, slot_k, ahead=True, negm=True):
    """QK^T gap g of the tile whose K sits in ring slot slot_k: fragment pair p = g >> 2 (k-step p >> 1, key blocks 2 (p & 1), 2 (p & 1) + 1)
    against query block g & 3 -- a pair of K fragments is consumed in four consecutive gaps and dead afterwards; pair p + 2 is read in gaps
    4p, 4p + 1 (8 gaps ahead of its first use)"""
    st = Stmt()
    p, qb = g >> 2, g & 3
    kk, mp = p >> 1, p & 1
    n = lq.wait([f"k{2 * p}", f"k{2 * p + 1}"])
    if n is not None:
        st.lines.append(f"s_waitcnt lgkmcnt({n})")
    for j in (0, 1):
        m = 2 * mp + j
        acc = f"sc[{P}][{qb}][{m}]"
        if kk == 0 and negm:
            st.lines.append(f"{MFMA} %[acc{j}], %[fa{j}], %[fb], %[nm]")
            st.outs.append(f'[acc{j}] "=&v"({acc})')
        elif kk == 0:
            st.lines.append(f"{MFMA} %[acc{j}], %[fa{j}], %[fb], 0")
            st.outs.append(f'[acc{j}] "=&v"({acc})')
        else:
            st.lines.append(f"{MFMA} %[acc{j}], %[fa{j}], %[fb], %[acc{j}]")
            st.outs.append(f'[acc{j}] "+v"({acc})')
        st.ins.append(f'[fa{j}] "a"(kf[{2 * p + j}])')
    st.ins.append(f'[fb] "a"(qf[{qb}][{kk}])')
    if kk == 0 and negm:
        st.ins.append(f'[nm] "v"(negm[{qb}])')
    if ahead and (g & 3) < 2 and p + 2 <= 7:
        add_read(st, lq, "k", slot_k, 2 * (p + 2) + (g & 3))
    return st


def pv_stmt(lq, g):
    """P.V gap g: Vt fragment f = g >> 1 (chunk f >> 3, d block f & 7), query blocks 2 (g & 1), 2 (g & 1) + 1.  Gaps 16c + 4qb + 3 also carry the
    row-sum MFMA of (query block qb, chunk c): lacc[qb] += ones . P -- every row of the 16 x 16 result is the sum over the chunk's 32 keys of
    the bf16 P of query column j: the softmax denominator without a single VALU add or cross-lane step (64 v_add_f32 per tile otherwise: the
    kernel is issue-bound, not pipe-bound)"""
    st = Stmt()
    f, qp = g >> 1, g & 1
    n = lq.wait([f"v{f}"])
    if n is not None:
        st.lines.append(f"s_waitcnt lgkmcnt({n})")
    for j in (0, 1):
        qb = 2 * qp + j
        st.lines.append(f"{MFMA} %[acc{j}], %[fa], %[fb{j}], %[acc{j}]")
        st.outs.append(f'[acc{j}] "+a"(o[{qb}][{f & 7}])')
        st.ins.append(f'[fb{j}] "v"(pk[{qb}][{f >> 3}])')
    st.ins.append(f'[fa] "a"(vf[{f}])')
    if g & 3 == 3:
        c, qb = g >> 4, (g >> 2) & 3
        st.lines.append(f"{MFMA} %[lacc], %[ones], %[fbl], %[lacc]")
        st.outs.append(f'[lacc] "+a"(lacc[{qb}])')
        st.ins += [f'[ones] "a"(ones)', f'[fbl] "v"(pk[{qb}][{c}])']
    return st


def pair_list(lo, hi):
    return [(b, q) for q in range(lo, hi) for b in range(4)]


def gen_iter(ST, out):
    PC, PN = ST & 1, (ST & 1) ^ 1
    slot_v, slot_k1, slot_k2, slot_d = ST, (ST + 1) & 3, (ST + 2) & 3, (ST + 3) & 3
    ind = "        "
    w = out.append
    lq = LdsQueue([f"k{j}" for j in range(4)])      # K(i+1) fragments 0..3: issued at the end of the phase 2 before (or by the prologue)
    w(f"    auto iter{ST} = [&](int i) __attribute__((always_inline)) {{")
    w(ind + f"// tile i: ring slot {slot_v}, scores sc[{PC}];  tile i+1: K in slot {slot_k1}, scores sc[{PN}]")
    w(ind + "const int t_next = t_begin + i + 1;")
    w(ind + "const bool live_next = i + 1 < n;")
    w(ind + "const bool mask_next = (t_next + 1) * KV_TILE > S;      // ragged last tile, or past the end: no key of it may count")
    late = pair_list(EARLY_PAIRS, 8)
    w(ind + "float " + ", ".join(f"e0_l{k}, e1_l{k}" for k in range(len(late))) + ";")
    w(ind + "uint32_t " + ", ".join(f"pw_l{k}" for k in range(len(late))) + ";")
    w(ind + "// ---- phase 1")
    stmts = []
    for g in range(32):
        st = qk_stmt(lq, PN, g, slot_k1)
        if g >= 32 - V_AHEAD:
            add_read(st, lq, "v", slot_v, g - (32 - V_AHEAD))
        stmts.append(st)
    fixed = {g: fixed_slots(stmts[g]) for g in range(32)}
    rg = RESCALE_GAP
    for k in range(4):
        fixed[rg + k] += 2
    placed = spread(pair_stream(late, PC, "l"), fixed, 0, 31)
    for g in range(32):
        st = stmts[g]
        add_stream(st, placed[g])
        out.extend(st.emit(ind))
        if rg <= g < rg + 4:
            w(ind + f"rescale({g - rg});")
        w(ind + "W4_FENCE();")
    w(ind + "// own pieces of tile i+2 landed (issued one iteration ago), then everyone's; every wave is past its reads of")
    w(ind + "// tile i-1's Vt and K, so ring slot (i+3) & 3 can be refilled")
    w(ind + 'asm volatile("s_waitcnt vmcnt(0)" ::: "memory");')
    w(ind + "W4_FENCE();")
    w(ind + "__builtin_amdgcn_s_barrier();")
    w(ind + "W4_FENCE();")
    early = pair_list(0, EARLY_PAIRS)
    if early:
        w(ind + "float " + ", ".join(f"e0_e{k}, e1_e{k}" for k in range(len(early))) + ";")
        w(ind + "uint32_t " + ", ".join(f"pw_e{k}" for k in range(len(early))) + ";")
    w(ind + "// ---- phase 2")
    stmts = []
    for g in range(32):
        st = pv_stmt(lq, g)
        f = g >> 1
        if g & 1 == 0 and f + V_AHEAD < 16:
            add_read(st, lq, "v", slot_v, f + V_AHEAD)
        if g >= 28:
            add_read(st, lq, "k", slot_k2, g - 28)
        if g < 8:
            add_max(st, PN, g)
        stmts.append(st)
    assert lq.q[-4:] == [f"k{j}" for j in range(4)]
    fixed = {g: fixed_slots(stmts[g]) for g in range(32)}
    for g in range(32):
        if g & 3 == 1:
            fixed[g] += DMA_SLOTS
    placed = spread(pair_stream(early, PN, "e"), fixed, FIRST_EARLY_GAP, 31)
    for g, instrs in placed.items():        # single-buffered P: chunk c of tile i is last read in gap 16c + 15
        for ins in instrs:
            if ins.after:
                c = int(ins.after.split("]")[1][1:])
                assert g > 16 * c + 15, f"early pack into chunk {c} in gap {g}: P.V still reads it"
    for g in range(32):
        st = stmts[g]
        if g == 0:     # a tile past the end of this work item (iterations come in fours) is fully masked
            w(ind + f"if (mask_next || !live_next) mask_scores(std::integral_constant<int, {PN}>{{}}, t_next, live_next ? S : 0);")
        add_stream(st, placed.get(g, []))
        out.extend(st.emit(ind))
        if g & 3 == 1:
            j = g >> 3
            w(ind + (f"stage_v({slot_d}, i + 3, {j});" if (g >> 2) & 1 else f"stage_k({slot_d}, i + 3, {j});"))
        if 8 <= g < 12:
            w(ind + f"sm_state_f(std::integral_constant<int, {PN}>{{}}, {g - 8}, std::false_type{{}});")
        w(ind + "W4_FENCE();")
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
    lq = LdsQueue([])
    for idx in range(4):
        st = Stmt()
        add_read(st, lq, "k", 0, idx)
        out.extend(st.emit(ind))
    w(ind + "W4_FENCE();")
    for g in range(32):
        out.extend(qk_stmt(lq, 0, g, 0, negm=False).emit(ind))
        w(ind + "W4_FENCE();")
    w(ind + 'asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // tile 1 landed')
    w(ind + "W4_FENCE();")
    w(ind + "__builtin_amdgcn_s_barrier();")
    w(ind + "W4_FENCE();")
    for idx in range(4):
        st = Stmt()
        add_read(st, lq, "k", 1, idx)
        out.extend(st.emit(ind))
    w(ind + 'asm volatile("s_nop 15");                           // last QK^T MFMAs -> the score reads below')
    w(ind + "W4_FENCE();")
    w(ind + "if ((t_begin + 1) * KV_TILE > S) mask_scores(std::integral_constant<int, 0>{}, t_begin, S);")
    early = pair_list(0, EARLY_PAIRS)
    w(ind + "{")
    ind2 = ind + "    "
    if early:
        w(ind2 + "float " + ", ".join(f"e0_p{k}, e1_p{k}" for k in range(len(early))) + ";")
        w(ind2 + "uint32_t " + ", ".join(f"pw_p{k}" for k in range(len(early))) + ";")
    for g in range(8):
        st = Stmt()
        add_max(st, 0, g)
        out.extend(st.emit(ind2))
        w(ind2 + "W4_FENCE();")
    for b in range(4):
        w(ind2 + f"sm_state_f(std::integral_constant<int, 0>{{}}, {b}, std::true_type{{}});")
    w(ind2 + "W4_FENCE();")
    stream = pair_stream(early, 0, "p")
    for k in range(0, len(stream), 5):
        st = Stmt()
        add_stream(st, stream[k:k + 5])
        out.extend(st.emit(ind2))
        w(ind2 + "W4_FENCE();")
    w(ind + "}")
    w("")


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    out = ["// GENERATED by tools/gen_attn_w7.py -- do not edit; the schedule tables and their rationale are in that script.",
           "// Included inside flash_attn_w7_kernel (attention variant 7, attention.hip), which declares every name used here.", ""]
    for st in range(4):
        gen_iter(st, out)
    gen_prologue(out)
    path = os.path.join(here, "..", "physicedit_amd", "csrc", "attention_w7_body.inc")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    print(f"wrote {os.path.normpath(path)}: {len(out)} lines")


if __name__ == "__main__":
    main()
