// bf16 MFMA GEMM for gfx950:  out = epilogue(A[M,K] . W[N,K]^T + bias)
//
// Replaces every torch.nn.functional.linear on the DiT hot path of the reference
// (DiffSynth-Studio/diffsynth/models/qwen_image_dit.py:282-283,313-314 QKV/out projections,
// :48,240 MLP, :333-336,347-350 modulation, :416-417,430 in/out embeds; pipelines/helpers.py:127-137
// adapter heads), with the element-wise op that follows each Linear in the reference fused into the
// epilogue AT THE REFERENCE'S ROUNDING BOUNDARIES: the Linear result is
// rounded to bf16 first (acc + bias -> bf16), then each following op rounds again.
//
// Structure (MI355X-first, not a CUDA tiling):
//   * 256x256x64 block tile, 512 threads = 8 waves as 4(M) x 2(N); each wave owns 64x128 of C as
//     4x8 v_mfma_f32_16x16x32_bf16 blocks (128 fp32 accumulators / lane; round 5: the 4-pass shape moves half the
//     accumulator bytes per FLOP of the 8-pass 32x32x16 and sustains 12-16 % more FLOP/s under the chip's power limit --
//     profiles/r05_gemm_notes.md section 7; the 32x32 form is kept behind the knob "gemm_mfma16" = 0; e4m3 uses the
//     block-scaled 32x32x64).
//   * A and W tiles go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip): A double
//     buffered, W in a three-deep ring (2 x 32 KiB + 3 x 32 KiB = all of the CU's 160 KiB).
//   * LDS image is [row][64 k] bf16 = 128 B rows; the 16-B chunk index is XORed with (row>>1)&7 so
//     every ds_read_b128 lane group covers all 64 banks.  LDS-DMA writes lane-linear, so the
//     permutation is applied to the per-lane SOURCE address and again on the read.
//   * operands are fed "swapped" (MFMA A-operand = W fragment, B-operand = activation fragment):
//     each lane then holds 4 consecutive N columns of one M row per accumulator quad.
//   * epilogue: y = bf16(acc + bias) is transposed through LDS so every lane owns 8 consecutive
//     columns of a row: 16-B bias/gate/residual loads and 16-B stores, 256-B segments.
//   * work-group ids are remapped XCD-aware (8 private L2s) and walk `band`-tile-tall bands.
//   * "grouped" launch: up to 2 problems (image stream + text stream) share one grid.
//   * schedule 17 (round 3): PERSISTENT work-groups (one per CU) that walk their tiles b, b + G, ...; the
//     main loop's tail prefetches -- which a one-tile work-group spends on clamped dummy loads -- fetch
//     the NEXT tile's first K tiles instead, the epilogue runs in the two LDS regions the stream does
//     not need, and its stores drain under the next tile's main loop.
#include "gemm_tile.h"

namespace pe {

long long* g_gemm_dbg = nullptr;

// Accumulators of a wave's 64 x 128 block in the two MFMA shapes (S16: 4 x 8 blocks of 16 x 16, else 2 x 4 blocks of 32 x 32): 32 quads either way
template <bool S16>
struct AccTile {
    typename std::conditional<S16, f32x4[4][8], f32x16[2][4]>::type v;
    PE_DEV f32x4 quad(int j) const {
        if constexpr (S16) return v[j >> 3][j & 7];
        else return f32x4{v[j >> 4][(j >> 2) & 3][4 * (j & 3)], v[j >> 4][(j >> 2) & 3][4 * (j & 3) + 1], v[j >> 4][(j >> 2) & 3][4 * (j & 3) + 2],
                          v[j >> 4][(j >> 2) & 3][4 * (j & 3) + 3]};
    }
    PE_DEV void set_quad(int j, f32x4 q) {
        if constexpr (S16) v[j >> 3][j & 7] = q;
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[j >> 4][(j >> 2) & 3][4 * (j & 3) + r] = q[r];
        }
    }
};

// One output tile per work-group (schedule 15): `bid` = tile id in the banded order.
template <int EPI, int VAR, bool FP8, bool S16>
__device__ __forceinline__ void gemm_tile(const KARG GemmArgs& args, char* smem, int bid) {
    static_assert(VAR == 15, "unknown GEMM schedule");
    constexpr int ES = FP8 ? 1 : 2;        // bytes per operand element
    constexpr int KT_BYTES = 128;          // one K tile of a row, in bytes (64 bf16 / 128 e4m3)
    const int lane = lane_id();
    const int w = wave_id();
    // fragment row of the lane inside an MFMA block and its k group: 32 rows x 2 groups of 8 (32 x 32 x 16), 16 rows x 4 groups of 8 (16 x 16 x 32)
    const int lrow = S16 ? (lane & 15) : (lane & 31), h = S16 ? (lane >> 4) : (lane >> 5);
    const int wm = w >> 1, wn = w & 1;

    PE_STAMP(0);
    const TileCoord tc0 = decode_tile(args, bid);
    const KARG GemmProblem& P = args.p[tc0.pi];
    const int M = P.M, N = P.N, K = P.K;
    const int m0 = tc0.m0, n0 = tc0.n0;

    // ---- staging sources: wave w moves pieces w*4..w*4+3 (1 KiB = 8 rows x 128 B) of A and of W
    const char* a_src[4];
    const char* w_src[4];
    {
        const char* A = (const char*)P.A;
        const char* W = (const char*)P.W;
        const int rin = lane >> 3, slot = lane & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // piece (1 KiB = 8 rows) i of this wave.  A pieces come from the wave group's OWN half (the only one it reads),
            // piece = grp*16 + 4(w&3) + i; W pieces 4w + i.
            const int piece_a = (w >> 2) * 16 + (w & 3) * 4 + i;
            const int piece_w = w * 4 + i;
            const int row_a = piece_a * 8 + rin, row_w = piece_w * 8 + rin;
            const int gr = min(m0 + row_a, M - 1);
            const int gn = min(n0 + row_w, N - 1);
            a_src[i] = A + (size_t)gr * P.lda * ES + (slot ^ ((row_a >> 1) & 7)) * 16;
            w_src[i] = W + (size_t)gn * K * ES + (slot ^ ((row_w >> 1) & 7)) * 16;
        }
    }

    AccTile<S16> accT;
    auto& acc = accT.v;
#pragma unroll
    for (int j = 0; j < 32; ++j) accT.set_quad(j, f32x4{0.f, 0.f, 0.f, 0.f});

    // per-lane fragment addressing: row = tile_row + lrow, chunk = the lane's 8 k of the k-step, swizzle (row>>1)&7
    const int sw = (lrow >> 1) & 7;

    const int nk = K * ES / KT_BYTES;
    {
        // "Ping-pong" schedule: the two wave groups (waves 0-3 = rows 0-127, waves 4-7 = rows 128-255; waves w and w+4
        // share a SIMD) run the SAME instruction stream one barrier apart, so at any time every SIMD has one wave in its MFMA
        // part and one in its load part: LDS latency, the DMA issue and the barrier skew of one group hide under the other
        // group's MFMAs (v10 interleaves them inside each wave).  Two phases of 16 MFMAs per K tile; phase p = row block
        // mi = p of the wave's C tile against all four column blocks:
        //     p0: reads A[mi 0] x KS + W[ni 0..3] x KS (20 ds_read_b128), stages this wave's 4 pieces of A(kt+1)
        //     p1: reads A[mi 1] x KS                                    , stages its 4 pieces of W(kt+2), vmcnt(4)
        // A half tiles are staged by the group that reads them (piece_a above), so "every wave of my group passed
        // the barrier behind its lgkmcnt(0)" is all the WAR protection A's two buffers need: A(kt+1) overwrites
        // A(kt-1), last read in p1 of tile kt-1.  W(kt+2) overwrites W(kt-1) (3-deep ring), last read by group 1 in
        // p0 of tile kt-1, i.e. >= 4 barriers earlier.  RAW: vmcnt(4) in p1 retires A(kt+1), W(kt+1); both groups
        // pass a barrier between that wait and the first read of tile kt+1.
        using FragT = typename std::conditional<FP8, i32x8, bf16x8>::type;
        constexpr int KS = (FP8 ? 2 : 4) / (S16 ? 2 : 1);     // MFMA k-steps per K tile
        constexpr int A_BYTES = BM * KT_BYTES, W_BYTES = BN * KT_BYTES;
        char* const a_base = smem;
        char* const w_base = smem + 2 * A_BYTES;
        const int grp = w >> 2;
        const int a_off = (wm * 64 + lrow) * 128;
        const int w_off = (wn * 128 + lrow) * 128;
        auto rd = [&](const char* rowp, int ks) -> FragT {
            if constexpr (FP8) {
                const int c0 = (S16 ? 8 : 4) * ks + 2 * h;
                const i32x4 lo = *(const i32x4*)(rowp + ((c0 ^ sw) << 4));
                const i32x4 hi = *(const i32x4*)(rowp + (((c0 + 1) ^ sw) << 4));
                return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            } else {
                return *(const bf16x8*)(rowp + (((ks * (S16 ? 4 : 2) + h) ^ sw) << 4));
            }
        };
#define PE_BAR()                                   \
    do {                                           \
        __builtin_amdgcn_sched_barrier(0);         \
        __builtin_amdgcn_s_barrier();              \
        __builtin_amdgcn_sched_barrier(0);         \
    } while (0)
        // fragments of one phase (32 rows of the wave's block x its 128 columns x the K tile): 4 of A and 16 of W in either bf16 shape
        // (32 x 32: fa[ks], fw[ni][ks]; 16 x 16: fa[mb * 2 + ks], fw[nb][ks])
        constexpr int NB = S16 ? 8 : 4;            // column blocks of the wave
        FragT fa[S16 ? 2 * KS : KS], fw4[NB][KS];
        auto rd_a1 = [&](const char* Sa, int mi) {
            if constexpr (S16) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) fa[mb * KS + ks] = rd(Sa + a_off + mi * 4096 + mb * 2048, ks);
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) fa[ks] = rd(Sa + a_off + mi * 4096, ks);
            }
        };
        auto rd_w4 = [&](const char* Sw) {
#pragma unroll
            for (int ni = 0; ni < NB; ++ni)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) fw4[ni][ks] = rd(Sw + w_off + ni * (S16 ? 2048 : 4096), ks);
        };
        auto mma16 = [&](int mi) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if constexpr (S16) {
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int nb = 0; nb < 8; ++nb) {
                            if constexpr (FP8)
                                acc[mi * 2 + mb][nb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                                    fw4[nb][ks], fa[mb * KS + ks], acc[mi * 2 + mb][nb], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                            else
                                acc[mi * 2 + mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw4[nb][ks], fa[mb * KS + ks], acc[mi * 2 + mb][nb], 0, 0, 0);
                        }
                } else {
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        if constexpr (FP8)
                            acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                                fw4[ni][ks], fa[ks], acc[mi][ni], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                        else
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw4[ni][ks], fa[ks], acc[mi][ni], 0, 0, 0);
                    }
                }
            }
        };
        auto st_a4 = [&](int i) {
            const int tc = min(i, nk - 1);
            char* base = a_base + (i & 1) * A_BYTES + (grp * 16 + (w & 3) * 4) * 1024;
#pragma unroll
            for (int j = 0; j < 4; ++j) glds16(a_src[j] + tc * KT_BYTES, base + j * 1024);
        };
        auto st_w4 = [&](int i, int slot) {
            const int tc = min(i, nk - 1);
            char* base = w_base + slot * W_BYTES + w * 4096;
#pragma unroll
            for (int j = 0; j < 4; ++j) glds16(w_src[j] + tc * KT_BYTES, base + j * 1024);
        };
#define PE_MMA16(mi, active)                       \
    do {                                           \
        PE_BAR();                                  \
        __builtin_amdgcn_s_setprio(1);             \
        if (active) mma16(mi);                     \
        __builtin_amdgcn_s_setprio(0);             \
        PE_BAR();                                  \
    } while (0)
        st_a4(0); st_w4(0, 0); st_w4(1, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // A(0), W(0) landed; W(1) may still fly
        PE_BAR();
        if (grp == 1) __builtin_amdgcn_s_barrier();        // stagger
        PE_STAMP(1);
        int ws = 0;
        // Ragged M tile (the second tile of a 272-token negative prompt holds 16 rows): a wave whose 32-row block lies beyond M skips that
        // block's fragment reads and MFMAs -- the chip is power-limited, MFMAs on clamped rows are joules -- and keeps every barrier, LDS-DMA
        // request and wait.  The loop exists twice so that full blocks (every other tile) carry no test; act1 implies act0.
        const bool act0 = m0 + wm * 64 < M || !args.skip_ragged, act1 = m0 + wm * 64 + 32 < M || !args.skip_ragged;
        auto kloop = [&](auto full_tag) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(full_tag)::value;
            for (int kt = 0; kt < nk; ++kt) {
                const char* Sa = a_base + (kt & 1) * A_BYTES;
                const int ws_n1 = ws == 2 ? 0 : ws + 1;
                const int ws_n2 = ws_n1 == 2 ? 0 : ws_n1 + 1;
                // p0
                if (FULL || act0) { rd_a1(Sa, 0); rd_w4(w_base + ws * W_BYTES); }
                st_a4(kt + 1);
                PE_MMA16(0, FULL || act0);
                // p1
                if (FULL || act1) rd_a1(Sa, 1);
                st_w4(kt + 2, ws_n2);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // this wave's pieces of A(kt+1), W(kt+1) landed
                PE_MMA16(1, FULL || act1);
                ws = ws_n1;
            }
        };
        if (act1) kloop(std::true_type{});
        else kloop(std::false_type{});
        if (grp == 0) __builtin_amdgcn_s_barrier();        // re-align the two groups
        PE_STAMP(2);
#undef PE_MMA16
#undef PE_BAR
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain every LDS-DMA (incl. the clamped tail tiles) before LDS is reused
    __syncthreads();
    PE_STAMP(3);
    char* E = smem + w * 16384;
    long long* stamp4 = nullptr;
    if constexpr (VAR == 15) stamp4 = args.dbg != nullptr ? args.dbg + (size_t)blockIdx.x * 8 + 4 : nullptr;
    if (args.no_epi) {      // timing build: keep the accumulators alive, store nothing
#pragma unroll
        for (int j = 0; j < 32; ++j) asm volatile("" ::"v"(accT.quad(j)));
    } else {
        gemm_epilogue<EPI, FP8, false, 2, 0, S16 ? 1 : 0>(P, M, N, acc, m0, n0, E, E + 8192, lane, w, stamp4, args.direct_epi);
    }
    PE_STAMP(5);
}

// ------------------------------------------------------------------------------------------
// Schedule 17: persistent work-groups.  Work-group b (one per CU) computes the tiles b, b + G, b + 2G, ... of the XCD-aware
// banded order (G = grid size, a multiple of 8: the tiles of a work-group stay on its XCD's chunk).  Its LDS-DMA requests form
// ONE stream over its tiles: stream index i < nk is K tile i of the current tile; i = nk and nk + 1 -- the requests that
// schedule 15 issues as clamped dummies in its last two iterations -- are K tiles 0 and 1 of the NEXT tile (same problem, both
// tiles complete: their source rows are this tile's plus a wave-uniform offset).  A(i) lives in buffer (ab + i) & 1, W(i) in
// ring slot (ws + i) % 3; ab and ws run on across tiles.  When the main loop ends, A(nk), W(nk) have landed (the last
// iteration's vmcnt(4)), W(nk + 1) is in flight, and the two regions that held K tile nk - 1 are dead: the epilogue stages
// the wave's 64 x 128 block there in two 32-row passes (8 KiB per wave: waves 0-3 in the A buffer, 4-7 in the W slot).  Its
// global stores are not waited for: they retire under the next tile's first K tiles (vmcnt is in order, so that tile's first
// vmcnt(4) also covers them).  What this removes per tile: the work-group launch, the cold prologue (HBM / L2 latency of the
// first K tiles with an idle matrix pipe) and most of the store drain.
// ------------------------------------------------------------------------------------------
//
// Schedule 19 (SK, round 4) = the same work-groups, "stream-K": a launch of T tiles x nk K tiles is T . nk units of equal work; the
// tile order's chunk of XCD x (T / 8 tiles) is cut into G / 8 equal unit ranges, one per work-group, so every CU does the same
// amount of work whatever T mod G is (the block Linears of the headline geometry are 1.6 / 4.8 / 6.4 rounds of 256 tiles: one-tile-
// per-work-group forms run 2 / 5 / 7).  A range covers the END of one tile (its "tail": K tiles [k1, nk)), whole tiles, and the
// BEGINNING of another (its "head": K tiles [0, k1)); ranges are at least one tile long, so a tile has at most two parts.  The
// head's holder computes it FIRST and publishes the fp32 accumulators (256 KiB, write-through stores, then a flag); the tail's
// holder runs its tail LAST, starting its accumulators from that image instead of from zero -- the K tiles are accumulated in the
// same order by the same instructions as in an unsplit tile, so the output is BIT-IDENTICAL to schedules 15 / 17 -- and runs
// the epilogue.  Positions (chunk, index) are handed out by the chunk's atomic counter in start order, and a
// work-group only ever waits for the holder of (chunk, index - 1), which has started by construction: no assumption about dispatch
// order or co-residency (cdna guide, "placement-independent protocols"), and since it publishes first thing, the wait is over
// long before it is reached.  The last work-group to take a position zeroes the counters and every consumer its flag: the workspace is all zero
// again when the launch ends (hipGraph-safe: no host-side generation number).
// One lane takes the work-group's position: index = the next free one of chunk `first` or, should that chunk be full (an XCD that runs
// more than its share of the work-groups), of the chunks after it; the last of the G work-groups to arrive zeroes the counters.
// (Not inlined: inside the kernel this retry loop trips "illegal VGPR to SGPR copy" in hipcc 7.2.)
__device__ __attribute__((noinline)) unsigned sk_take_position(unsigned* sync, unsigned first, unsigned per_chunk, unsigned G) {
    unsigned chunk = first, idx = 0;
    for (int k = 0; k < 8; ++k) {
        idx = __hip_atomic_fetch_add(sync + 1 + chunk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (idx < per_chunk) break;
        chunk = (chunk + 1) & 7u;
    }
    // every work-group of the launch holds a position once G have arrived
    const unsigned arrived = __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (arrived == G - 1) {
        for (int c = 0; c < 9; ++c) __hip_atomic_store(sync + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return chunk | (idx << 3);
}

template <int EPI, bool FP8, bool SK, int PH, bool S16>
__device__ __forceinline__ void gemm_persistent(const KARG GemmArgs& args, char* smem) {
    constexpr int ES = FP8 ? 1 : 2;
    constexpr int KT_BYTES = 128;
    using FragT = typename std::conditional<FP8, i32x8, bf16x8>::type;
    constexpr int KS = (FP8 ? 2 : 4) / (S16 ? 2 : 1);
    constexpr int NB = S16 ? 8 : 4;
    constexpr int A_BYTES = BM * KT_BYTES, W_BYTES = BN * KT_BYTES;
    const int w = wave_id();
    const int wm = w >> 1, wn = w & 1;
    const int grp = w >> 2;
    char* const a_base = smem;
    char* const w_base = smem + 2 * A_BYTES;
    const int G = (int)gridDim.x;
    const int ntiles = args.ntiles;

#define PE_BAR()                                   \
    do {                                           \
        __builtin_amdgcn_sched_barrier(0);         \
        __builtin_amdgcn_s_barrier();              \
        __builtin_amdgcn_sched_barrier(0);         \
    } while (0)

    // The work list of this work-group as segments (tile of the banded order, K tiles [ka, kb)), visited s = 0, 1, ...
    //   17: tiles b, b + G, ... whole;   19: the unit range [u0, u1) of its chunk, LAST tile first (see above)
    int sk_base = 0, sk_tf = 0, sk_tl = -1, sk_kf = 0, sk_kl = 0, sk_nk = 0, sk_pos = 0;
    if constexpr (SK) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);      // hwreg(HW_REG_XCC_ID, 0, 4): the XCD this work-group runs on
        if (threadIdx.x == 0) {
            // position = (chunk, index) from the chunk's own counter, in start order.  The chunk is the XCD this work-group RUNS on (XCC_ID:
            // its L2 then serves one chunk's tiles, as the XCD-aware order of schedules 15 / 17 arranges -- with positions taken from one
            // global ticket the 32 work-groups of an XCD worked all over the matrix and the launch took 1.2 - 1.5 x as long).  Placement
            // is speed only: should an XCD run more than G / 8 work-groups, the surplus takes the next chunk with a free position.
            const unsigned pos = sk_take_position(args.sk_sync, xcc & 7u, (unsigned)(G >> 3), (unsigned)G);
            *(volatile unsigned*)smem = pos;
        }
        __syncthreads();
        const int tk = __builtin_amdgcn_readfirstlane((int)*(volatile unsigned*)smem);
        // (the first segment's cold start passes a barrier before anything is staged into this LDS word)
        const int Gc = G >> 3, xc = tk & 7, idx = tk >> 3;
        const int q = ntiles >> 3, r = ntiles & 7;
        sk_base = xc < r ? xc * (q + 1) : r * (q + 1) + (xc - r) * q;
        const int cnt = q + (xc < r ? 1 : 0);
        sk_nk = args.p[0].K * ES / KT_BYTES;                 // the launcher checked: both problems have this K
        const int Uc = cnt * sk_nk;
        const int u0 = (int)((long long)idx * Uc / Gc), u1 = (int)((long long)(idx + 1) * Uc / Gc);
        sk_tf = u0 / sk_nk; sk_kf = u0 - sk_tf * sk_nk;
        sk_tl = (u1 - 1) / sk_nk; sk_kl = u1 - sk_tl * sk_nk;
        sk_pos = xc * Gc + idx;
    }
    struct Seg { int tile, ka, kb; bool valid; };
    auto seg_at = [&](int s_) -> Seg {
        Seg g;
        if constexpr (SK) {
            const int ti = sk_tl - s_;
            g.valid = ti >= sk_tf;
            g.tile = sk_base + ti;
            g.ka = ti == sk_tf ? sk_kf : 0;
            g.kb = ti == sk_tl ? sk_kl : sk_nk;
        } else {
            const int t = (int)blockIdx.x + s_ * G;
            g.valid = t < ntiles;
            g.tile = g.valid ? xcd_remap(t, ntiles) : 0;
            g.ka = 0; g.kb = -1;           // kb < 0: the tile's whole K
        }
        return g;
    };

    int ab = 0, ws = 0;     // A buffer / W ring slot of K tile 0 of the current segment
    bool have = false;      // K tiles 0 (A, W) and 1 (W) of the current segment were requested by the previous one
    // Round 6 ("gemm_continuous", schedule 21 only): the previous tile left the two wave groups ONE SLOT APART -- it skipped the re-align
    // barrier in front of its epilogue, so this tile skips its opening barrier and the stagger.  The operand stream is already one stream
    // over the work-group's tiles (stream index nk = K tile 0 of the next tile: same buffers, same waits, same barriers between a request,
    // its retirement and its first read as inside a tile), so the slot sequence simply runs on: group 0's {epilogue, L(0) of the next tile}
    // is concurrent with group 1's M(nk - 1), group 0's M(0) with group 1's {epilogue, L(0)}.  Only for tiles whose epilogue touches no
    // LDS (the direct forms: with the groups a slot apart, the regions an LDS epilogue stages in are still being read / already being
    // re-filled by the other group) and whose successor is prefetched.  What it removes per tile: the slot group 0 idles in front of the
    // re-align barrier, the slot group 1 idles behind the stagger, and each group's epilogue VALU / store issue runs beside the other
    // group's MFMAs instead of beside the other group's epilogue.
    bool skewed = false;
    // Round 6 ("gemm_defer_epilogue"): the previous tile parked y = bf16(acc + bias) in the stash; its epilogue proper runs as DEFER_SLICES
    // slices inside this tile's main loop (gemm_tile.h, "Deferred epilogue").  pm0 / pn0: that tile's origin (same problem as this one).
    // Only the gated-residual form: its slice is 2 loads, 20 VALU operations and a store.  The GELU form (110 VALU per slice, 16 of them
    // transcendental) was built the same way and measured +2.7 ... +4.5 % per Linear: an L slot has no slack against the other group's
    // 1024-cycle M slot, and spreading the stream through the shadows of the wave's own MFMAs is beyond hipcc's allocator (186 spilled
    // registers in the K loop) -- profiles/r06_gemm_notes.md.
    // COMPILED OUT by default (-DPE_GEMM_DEFER builds it in: tools/r06_build_ab.sh).  Against its own switched-off arm the deferred form is
    // 3 % faster on out-proj (6 % e4m3) -- but a kernel that merely CONTAINS this code is 8 - 9 % slower on both gated-residual Linears
    // than one that does not (out-proj 145.3 vs 132.8 us, MLP-down 541 vs 502 us, 10.58 vs 10.39 s per image on one box, three interleaved
    // rounds: profiles/r06_gemm_lib_ab.log): two more live tuples and the wave-uniform slice branches cost the K loop of a kernel that
    // sits at 256 registers more than the slices hide.  A/B between knobs of ONE binary cannot see that; A/B between binaries did.
#ifdef PE_GEMM_DEFER
    constexpr bool DEFER = PH == 1 && !SK && S16 && EPI == EPI_GATE_RES;
#else
    constexpr bool DEFER = false;
#endif
    constexpr int NL = 2;      // loads a slice requests: stash chunk + residual chunk
    bool pend = false;
    int pm0 = 0, pn0 = 0;
    for (int sidx = 0;; ++sidx) {
        const Seg cur = seg_at(sidx);
        if (!cur.valid) break;
        // the lane id is made opaque per tile: everything lane-dependent (staging sources, fragment and epilogue addresses) is
        // then recomputed per tile instead of being hoisted out of this loop and kept live across the main loop (spills)
        // (read from the hardware with mbcnt, not from threadIdx.x: that VGPR would have to stay live -- or be spilled and
        // reloaded here behind an s_waitcnt vmcnt(0), which waits for the previous tile's stores and the prefetch in flight)
        int lane_;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_));
        const int lane = lane_;
        const int lrow = S16 ? (lane & 15) : (lane & 31), h = S16 ? (lane >> 4) : (lane >> 5);
        const int sw = (lrow >> 1) & 7;
        const int a_off = (wm * 64 + lrow) * 128;
        const int w_off = (wn * 128 + lrow) * 128;
        const TileCoord tc0 = decode_tile(args, cur.tile);
        const KARG GemmProblem& P = args.p[tc0.pi];
        const int M = P.M, N = P.N, K = P.K;
        const int m0 = tc0.m0, n0 = tc0.n0;
        const int ka = cur.ka;
        const int nk = (cur.kb < 0 ? K * ES / KT_BYTES : cur.kb) - ka;      // K tiles of this segment
        const bool sk_publish = SK && cur.kb < sk_nk;      // a head: its accumulators go to the tail's holder
        const bool sk_consume = SK && ka > 0;              // a tail: its accumulators start from the head's
        // next segment of this work-group: prefetchable iff same problem, both tiles complete, and the streams have room (>= 2 K tiles)
        bool have_next = false;
        long long a_next = 0, w_next = 0;      // byte offset from this segment's source rows / first K tile to the next one's
        const Seg nxt = seg_at(sidx + 1);
        if (nxt.valid) {
            const TileCoord tn = decode_tile(args, nxt.tile);
            const int nk_next = (nxt.kb < 0 ? K * ES / KT_BYTES : nxt.kb) - nxt.ka;
            have_next = tn.pi == tc0.pi && nk >= 2 && nk_next >= 2 && m0 + BM <= M && tn.m0 + BM <= M && n0 + BN <= N && tn.n0 + BN <= N;
            a_next = (long long)(tn.m0 - m0) * P.lda * ES + (long long)(nxt.ka - ka) * KT_BYTES;
            w_next = (long long)(tn.n0 - n0) * K * ES + (long long)(nxt.ka - ka) * KT_BYTES;
        }
        // staging sources of this tile (recomputed per tile: nothing lane-dependent is carried across the epilogue)
        const char* a_src[4];
        const char* w_src[4];
        {
            const char* A = (const char*)P.A;
            const char* W = (const char*)P.W;
            const int rin = lane >> 3, slot = lane & 7;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int piece_a = grp * 16 + (w & 3) * 4 + i;     // A pieces of the wave group's OWN 128-row half
                const int piece_w = w * 4 + i;
                const int row_a = piece_a * 8 + rin, row_w = piece_w * 8 + rin;
                const int gr = min(m0 + row_a, M - 1);
                const int gn = min(n0 + row_w, N - 1);
                a_src[i] = A + (size_t)gr * P.lda * ES + (slot ^ ((row_a >> 1) & 7)) * 16 + (size_t)ka * KT_BYTES;
                w_src[i] = W + (size_t)gn * K * ES + (slot ^ ((row_w >> 1) & 7)) * 16 + (size_t)ka * KT_BYTES;
            }
        }
        // byte offset of stream index i (wave-uniform, branch-free): K tile ka + i of this tile; past the segment's end K tile i - nk
        // of the next segment, or (no prefetchable next segment) a clamped dummy re-read of this segment's last K tile
        const int i_lim = have_next ? 0x7fffffff : nk - 1;
        const long long a_adj = have_next ? a_next - (long long)nk * KT_BYTES : 0;
        const long long w_adj = have_next ? w_next - (long long)nk * KT_BYTES : 0;
        auto off_a = [&](int i) -> long long { return (long long)min(i, i_lim) * KT_BYTES + (a_adj & -(long long)(i >= nk)); };
        auto off_w = [&](int i) -> long long { return (long long)min(i, i_lim) * KT_BYTES + (w_adj & -(long long)(i >= nk)); };
        auto st_a4 = [&](int i, int buf) {
            long long off = off_a(i);
            asm volatile("" : "+s"(off));       // one scalar sum, then ONE 64-bit add per piece
            char* base = a_base + buf * A_BYTES + (grp * 16 + (w & 3) * 4) * 1024;
#pragma unroll
            for (int j = 0; j < 4; ++j) glds16(a_src[j] + off, base + j * 1024);
        };
        auto st_w4 = [&](int i, int slot) {
            long long off = off_w(i);
            asm volatile("" : "+s"(off));
            char* base = w_base + slot * W_BYTES + w * 4096;
#pragma unroll
            for (int j = 0; j < 4; ++j) glds16(w_src[j] + off, base + j * 1024);
        };
        auto rd = [&](const char* rowp, int ks) -> FragT {
            if constexpr (FP8) {
                const int c0 = (S16 ? 8 : 4) * ks + 2 * h;
                const i32x4 lo = *(const i32x4*)(rowp + ((c0 ^ sw) << 4));
                const i32x4 hi = *(const i32x4*)(rowp + (((c0 + 1) ^ sw) << 4));
                return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            } else {
                return *(const bf16x8*)(rowp + (((ks * (S16 ? 4 : 2) + h) ^ sw) << 4));
            }
        };

        AccTile<S16> accT;
        auto& acc = accT.v;
        // image of a tile's accumulators in the stream-K workspace: quad j (AccTile::quad) of thread tid at ((j * 512 + tid) * 16
        // bytes: every store / load instruction of a wave covers 1 KiB contiguous
        auto sk_image = [&](int seam) -> char* { return args.sk_part + (size_t)seam * SK_PART_BYTES + (size_t)(w * 64 + lane) * 16; };
        if (sk_consume) {
            // the head of this tile was published by the holder of the position before this one when it STARTED (its first segment): normally long ago
            unsigned* flag = args.sk_sync + SK_FLAG0 + (sk_pos - 1);
            if (threadIdx.x == 0) {
                while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(8);
                __hip_atomic_store(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // at rest again for the next launch
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                            // this CU's L1 (one lane + barrier: guide G16)
            }
            __syncthreads();
            const char* img = sk_image(sk_pos - 1);
#pragma unroll
            for (int j = 0; j < 32; ++j) accT.set_quad(j, *(const f32x4*)(img + (size_t)j * (GEMM_THREADS * 16)));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the main loop counts its own LDS-DMA requests only
        } else {
#pragma unroll
            for (int j = 0; j < 32; ++j) accT.set_quad(j, f32x4{0.f, 0.f, 0.f, 0.f});
        }

        FragT fa[S16 ? 2 * KS : KS], fw4[NB][KS];
        auto rd_a1 = [&](const char* Sa, int mi) {
            if constexpr (S16) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) fa[mb * KS + ks] = rd(Sa + a_off + mi * 4096 + mb * 2048, ks);
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) fa[ks] = rd(Sa + a_off + mi * 4096, ks);
            }
        };
        auto rd_w4 = [&](const char* Sw) {
#pragma unroll
            for (int ni = 0; ni < NB; ++ni)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) fw4[ni][ks] = rd(Sw + w_off + ni * (S16 ? 2048 : 4096), ks);
        };
        // the MFMAs of k-step ks of row half mi
        auto mma_ks = [&](int mi, int ks) __attribute__((always_inline)) {
            if constexpr (S16) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 8; ++nb) {
                        if constexpr (FP8)
                            acc[mi * 2 + mb][nb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                                fw4[nb][ks], fa[mb * KS + ks], acc[mi * 2 + mb][nb], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                        else
                            acc[mi * 2 + mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw4[nb][ks], fa[mb * KS + ks], acc[mi * 2 + mb][nb], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    if constexpr (FP8)
                        acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                            fw4[ni][ks], fa[ks], acc[mi][ni], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                    else
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw4[ni][ks], fa[ks], acc[mi][ni], 0, 0, 0);
                }
            }
        };
        auto mma16 = [&](int mi) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) mma_ks(mi, ks);
        };
#define PE_MMA16(mi, active)                       \
    do {                                           \
        PE_BAR();                                  \
        __builtin_amdgcn_s_setprio(1);             \
        if (active) mma16(mi);                     \
        __builtin_amdgcn_s_setprio(0);             \
        PE_BAR();                                  \
    } while (0)

        const int ws1 = ws == 2 ? 0 : ws + 1;
        if (!have) {
            // cold start of a tile: the regions may still receive the previous tile's clamped dummies, and its epilogue ran in
            // the other two: drain, meet, then stage as schedule 15 does
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PE_BAR();
            st_a4(0, ab); st_w4(0, ws); st_w4(1, ws1);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // A(0), W(0) landed; W(1) may still fly
        }
        // (have: this wave's pieces of A(0), W(0) were retired by the vmcnt(4) of the previous tile's last iteration)
        if (!skewed) {
            PE_BAR();
            if (grp == 1) __builtin_amdgcn_s_barrier();        // stagger
        }
        int abk = ab, wsk = ws;
        if constexpr (PH == 1) {
            // Schedule 21 (round 5): ONE hand-off per K tile and wave group instead of two.  A group alternates a LOAD slot (20 fragment
            // reads: the K tile's 16 weight fragments + row block 0's 4 activation fragments; all 8 of its LDS-DMA pieces) with a
            // 32-MFMA slot (row block 0, then row block 1, whose 4 activation fragments are read into row block 0's registers as
            // each k-step's last MFMA of block 0 has issued); the other group runs the same stream one barrier later.  Per K tile a
            // wave passes 2 barriers instead of 4, i.e. the matrix pipe changes hands twice instead of four times per 64 MFMAs
            // (each hand-off costs ~110 cycles of idle pipe: profiles/r03_gemm_notes.md section 7-8, r05_gemm_notes.md).
            //   group 0: slot 2kt = L(kt), 2kt+1 = M(kt);   group 1: slot 2kt+1 = L(kt), 2kt+2 = M(kt)
            // RAW: W(kt+2) (requested in L(kt)) is first read by group 0 in slot 2kt+4: every wave retires its pieces with vmcnt(8) at
            // the end of L(kt+1) -- group 1's is slot 2kt+3, a barrier before the read; A(kt+1) (requested in L(kt), the group's own half)
            // with vmcnt(4) at the end of M(kt), a barrier before L(kt+1).  WAR: W(kt+2) overwrites W(kt-1), last read by group 1 in
            // L(kt-1) = slot 2kt-1 (reads drained before its closing barrier), requested from slot 2kt on; A(kt+1) overwrites A(kt-1),
            // whose last reads were consumed by the group's own MFMAs in M(kt-1).  Same K order per output: bit-identical to 15 / 17.
            // deferred slices of the PREVIOUS tile (DEFER): stash chunk / residual chunk / gate chunk of slice kt, requested at the start of
            // M(kt), consumed behind the operand wait of L(kt + 1).  vmcnt retires in order: those loads are older than L(kt + 1)'s eight
            // LDS-DMA requests, so its vmcnt(8) covers them; the slice's store (issued behind that wait) and the next slice's loads are
            // NEWER than the A(kt + 1) pieces M(kt)'s closing wait is for, so that wait allows for them.
            u32x4 sl_v = {0, 0, 0, 0}, sl_r = {0, 0, 0, 0};
            for (int kt = 0; kt < nk; ++kt) {
                const char* Sa = a_base + abk * A_BYTES;
                const int ws_n1 = wsk == 2 ? 0 : wsk + 1;
                const int ws_n2 = ws_n1 == 2 ? 0 : ws_n1 + 1;
                // L slot
                rd_a1(Sa, 0); rd_w4(w_base + wsk * W_BYTES);
                st_a4(kt + 1, abk ^ 1);
                st_w4(kt + 2, ws_n2);
                // this wave's pieces of W(kt+1) landed; its fragment reads done.  (DEFER: so has the slice the previous M slot requested with asm
                // loads -- the registers are tied to this wait, which is where the compiler may first touch them)
                if constexpr (DEFER) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" : "+v"(sl_v), "+v"(sl_r) : : "memory");
                else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                int newer = 0;                 // VMEM operations of this K tile that are newer than its A(kt+1) requests, beyond the 4 of W(kt+2)
                if constexpr (DEFER) {
                    if (pend && kt >= 1 && kt <= DEFER_SLICES) {      // wave-uniform: slice kt - 1, beside the other wave group's MFMAs
                        __builtin_amdgcn_sched_barrier(0);
                        defer_slice_emit<EPI, FP8>(P, defer_pos(kt - 1, pm0, pn0, lane, w), sl_v, sl_r);
                        __builtin_amdgcn_sched_barrier(0);
                        newer = 1;
                    }
                }
                PE_BAR();
                // M slot
                if constexpr (DEFER) {
                    if (pend && kt < DEFER_SLICES) {
                        // asm loads (wave-uniform base + 32-bit lane offset): invisible to the compiler's own wait insertion, which would
                        // otherwise put an s_waitcnt vmcnt(0) in front of the consumer and drain the operand stream once per K tile
                        const char* sb = args.stash + (size_t)blockIdx.x * DEFER_STASH_BYTES + (size_t)kt * (GEMM_THREADS * 16) + (size_t)w * 1024;
                        asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(sl_v) : "v"(lane * 16), "s"(sb) : "memory");
                        const int l15 = lane & 15, g4 = lane >> 4, ncol = (g4 & 1) * 16 + 8 * (g4 >> 1);
                        const int mu = pm0 + wm * 64 + (kt >> 2) * 16, nu = pn0 + wn * 128 + (kt & 3) * 32;      // wave-uniform part of the chunk's position
                        const char* rb = (const char*)P.res + ((size_t)mu * P.ldr + nu) * 2;
                        asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(sl_r) : "v"((l15 * P.ldr + ncol) * 2), "s"(rb) : "memory");
                        newer += NL;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    mma_ks(0, ks);
                    __builtin_amdgcn_sched_barrier(0);
                    // row block 1's fragments of this k-step, into the registers just consumed
                    if constexpr (S16) {
                        fa[ks] = rd(Sa + a_off + 4096, ks);
                        fa[KS + ks] = rd(Sa + a_off + 4096 + 2048, ks);
                    } else {
                        fa[ks] = rd(Sa + a_off + 4096, ks);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                mma16(1);
                __builtin_amdgcn_s_setprio(0);
                // this wave's pieces of A(kt+1) landed: everything but the 4 requests of W(kt+2) and the slice traffic behind them
                if constexpr (DEFER) {
                    switch (newer) {
                        case 0: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                        case 1: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;      // the last slice's store
                        case 2: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;      // the first slice's two loads
                        default: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;     // a store and two loads
                    }
                } else {
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                }
                PE_BAR();
                abk ^= 1;
                wsk = ws_n1;
            }
        } else {
            // ragged M tile: blocks beyond M skip their MFMAs (see schedule 15).  ONE loop with wave-uniform tests here:
            // a second copy of the loop in this tile-walking kernel costs the register allocator 150+ spilled registers
            const bool act0 = m0 + wm * 64 < M || !args.skip_ragged, act1 = m0 + wm * 64 + 32 < M || !args.skip_ragged;
            for (int kt = 0; kt < nk; ++kt) {
                const char* Sa = a_base + abk * A_BYTES;
                const int ws_n1 = wsk == 2 ? 0 : wsk + 1;
                const int ws_n2 = ws_n1 == 2 ? 0 : ws_n1 + 1;
                // p0
                rd_a1(Sa, 0); rd_w4(w_base + wsk * W_BYTES);       // (unconditional: fragments that are only sometimes loaded become
                st_a4(kt + 1, abk ^ 1);                            //  loop-carried values -- 90 spilled registers)
                PE_MMA16(0, act0);
                // p1
                rd_a1(Sa, 1);
                st_w4(kt + 2, ws_n2);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // this wave's pieces of A(kt+1), W(kt+1) landed
                PE_MMA16(1, act1);
                abk ^= 1;
                wsk = ws_n1;
            }
        }
        // stay a slot apart across the tile boundary?  (wave-uniform; have_next implies both tiles complete and one problem)
        bool keep_skew = false;
        if constexpr (PH == 1 && !SK && kDirectEpi<EPI>) keep_skew = args.cont != 0 && have_next && (args.direct_epi & 1) != 0 && P.pre == nullptr && !args.no_epi;
        if (!keep_skew) {
            if (grp == 0) __builtin_amdgcn_s_barrier();    // re-align the two groups: every fragment read of the tile is done
        }
        skewed = keep_skew;
        __builtin_amdgcn_sched_barrier(0);
#undef PE_MMA16
        // abk / wsk = where stream index nk lives (next tile's K tile 0); the regions of K tile nk - 1 are free
        const int a_free = abk ^ 1;
        const int w_free = wsk == 0 ? 2 : wsk - 1;
        char* E = w < 4 ? a_base + a_free * A_BYTES + w * 8192 : w_base + w_free * W_BYTES + (w - 4) * 8192;
        bool defer_this = false;
        if constexpr (DEFER) {
            // park this tile's y and run its epilogue inside the next tile's loop?  The next tile must be this work-group's prefetched successor
            // (same problem, both complete: have_next) with room for the slices; a gated form needs its gate vector
            defer_this = args.defer != 0 && args.stash != nullptr && have_next && (args.direct_epi & 1) != 0 && P.pre == nullptr && !args.no_epi &&
                         nk >= DEFER_SLICES + 2 && P.gate != nullptr;
            if (defer_this) gemm_epilogue_dump16<EPI, FP8>(P, acc, m0, n0, lane, w, args.stash + (size_t)blockIdx.x * DEFER_STASH_BYTES + (size_t)(w * 64 + lane) * 16);
            pend = defer_this;
            pm0 = m0;
            pn0 = n0;
        }
        if (defer_this) {
        } else if (sk_publish) {
            // a head: write-through (sc1) 16-byte stores of the accumulators, every wave's drained, then the flag (guide: publish-large /
            // handoff-flag); the tail's holder acquires and reads them with plain loads
            char* img = sk_image(sk_pos);
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const f32x4 v = accT.quad(j);
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(img + (size_t)j * (GEMM_THREADS * 16)), "v"(v) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(args.sk_sync + SK_FLAG0 + sk_pos, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (args.no_epi) {      // timing build: keep the accumulators alive, store nothing
#pragma unroll
            for (int j = 0; j < 32; ++j) asm volatile("" ::"v"(accT.quad(j)));
        } else {
            gemm_epilogue<EPI, FP8, true, 2, 0, S16 ? 1 : 0>(P, M, N, acc, m0, n0, E, E, lane, w, nullptr, args.direct_epi);
        }
        ab = abk;
        ws = wsk;
        have = have_next;
    }
#undef PE_BAR
}

template <int EPI, int VAR, bool FP8, bool S16>
__global__ void __launch_bounds__(GEMM_THREADS, 2) gemm_bf16_kernel(const GemmArgs args_by_value) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // the only kernel argument sits at offset 0 of the kernarg segment
    const KARG GemmArgs& args = *(const KARG GemmArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    if constexpr (VAR == 17 || VAR == 19 || VAR == 21)
        gemm_persistent<EPI, FP8, VAR == 19, VAR == 21 ? 1 : 2, S16>(args, smem);
    else
        gemm_tile<EPI, VAR, FP8, S16>(args, smem, xcd_remap((int)blockIdx.x, (int)gridDim.x));
}

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
static int env_variant() {
    const int v = env_int("PE_GEMM_VARIANT", GEMM_DEFAULT_VARIANT);
    return v == 0 ? GEMM_DEFAULT_VARIANT : v;      // 0 = "the compiled default", as in pe_debug_set("gemm_variant", 0)
}
int g_gemm_variant = env_variant();
int g_gemm_band = env_int("PE_GEMM_BAND", GEMM_DEFAULT_BAND);
int g_gemm_skip_ragged = env_int("PE_GEMM_SKIP_RAGGED", 1);
int g_gemm_direct_epi = env_int("PE_GEMM_DIRECT_EPILOGUE", 1);
int g_gemm_no_epi = 0;      // timing only: knob "gemm_no_epilogue"
int g_gemm_defer = env_int("PE_GEMM_DEFER_EPILOGUE", 1);      // schedule 21: a tile's epilogue proper runs inside the next tile's main loop (knob "gemm_defer_epilogue"; needs a stash)
int g_gemm_cont = env_int("PE_GEMM_CONTINUOUS", 0);      // schedule 21: the wave groups stay a slot apart across tile boundaries (knob "gemm_continuous"; bit-identical, measured +0.2 ... +2.6 % per Linear: off)
// MFMA shape of the 8-wave schedules, a bit mask: bit 0 = bf16 on v_mfma_f32_16x16x32_bf16, bit 1 = e4m3 on v_mfma_scale_f32_16x16x128_f8f6f4 (both
// default since round 5); a clear bit = that dtype on the 32 x 32 blocks of rounds 1 - 4 (schedules 15 / 17 only: the A/B reference).  e4m3: the
// 16 x 16 x 128 form is 2 - 10 % slower per Linear in isolation and 1 % FASTER per image in the two-stream pipeline (less energy per FLOP: the
// other stream's kernels run faster) -- profiles/r05_gemm_notes.md section 7.
int g_gemm_mfma16 = env_int("PE_GEMM_MFMA16", 3);
int g_gemm_persist_wgs = 0;    // 0 = one work-group per CU of the current device
int g_gemm_sk = env_int("PE_GEMM_SK", 0);     // schedule 19 where it applies (A/B knob "gemm_sk"; measured slower: profiles/r04_gemm_notes.md)
// schedule 17 from this many rounds of tiles on (knob "gemm_persist_min_rounds"; G + 1 tiles at least).  Round 3 used 3: in isolation
// the 1.6-round Linears tie between 15 and 17.  In the two-stream pipeline 1 is 0.9 % faster per image (three interleaved A/B pairs on
// one box: 11.59 / 11.61 / 11.61 s with 3, 11.50 / 11.49 / 11.50 s with 1; profiles/r04_gemm_notes.md section 5): the default since round 4.
int g_gemm_persist_min_rounds = 1;
GemmWorkspace g_gemm_ws = {nullptr, 0, nullptr, 0};
constexpr size_t SK_SYNC_BYTES = 4096;        // ticket + flags (<= 992 work-groups), then the accumulator images

static int persistent_grid() {
    static std::atomic<int> cus{0};
    int n = cus.load(std::memory_order_acquire);
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
        n &= ~7;                         // a multiple of 8: a work-group's tiles stay on its XCD's chunk of the order
        if (n < 8) n = 8;
        cus.store(n, std::memory_order_release);
    }
    if (g_gemm_persist_wgs > 0) return (g_gemm_persist_wgs + 7) & ~7;
    return n;
}

#ifdef PE_GEMM_DEFER
size_t gemm_stash_bytes() { return (size_t)persistent_grid() * DEFER_STASH_BYTES; }
#else
size_t gemm_stash_bytes() { return 0; }      // the deferred epilogue is not compiled in: nobody needs a stash
#endif
size_t gemm_workspace_bytes() { return SK_SYNC_BYTES + (size_t)persistent_grid() * SK_PART_BYTES; }

template <int EPI, int VAR, bool FP8, bool S16>
static int launch_v(const GemmArgs& args, int grid, hipStream_t stream) {
    static std::atomic<bool> configured{false};   // racing first calls both configure: idempotent
    if (!configured.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<EPI, VAR, FP8, S16>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        if (e != hipSuccess) return set_error(PE_ERR_HIP, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
        configured.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((gemm_bf16_kernel<EPI, VAR, FP8, S16>), dim3(grid), dim3(GEMM_THREADS), GEMM_LDS, stream, args);
    return check_launch(FP8 ? "gemm_fp8_kernel" : "gemm_bf16_kernel");
}

template <int EPI>
static int launch_t(const GemmArgs& args, bool fp8, hipStream_t stream) {
    // instantiated schedules: 17 (default: persistent work-groups; needs more than one round of tiles), 15 (one tile per
    // work-group: the round-2 default, what 17 falls back to, and the one that can write s_memtime stamps).  (10, the round-1
    // schedule, left the library in round 5.)  Everything else that was tried is in profiles/r0*_gemm_*.md.
    const int ntiles = args.ntiles;
    int var = g_gemm_variant;
    const int G = persistent_grid();
#ifdef PE_GEMM_DEV_EPI      // development only (register / ISA checks of one instantiation: hipcc -DPE_GEMM_DEV_EPI=1 -S): never set by build.py
    (void)ntiles; (void)var;
    return fp8 ? launch_v<EPI, 21, true, true>(args, G, stream) : launch_v<EPI, 21, false, true>(args, G, stream);
#else
    // 19 (stream-K) wherever it applies -- a workspace was given, more than one round of tiles that do not fill whole rounds, one K --
    // because it is the only schedule whose time does not round T / G up; it only exists for the epilogues of the DiT block's Linears.
    // "gemm_variant" 19 forces it on every launch it can run (tests), 17 / 15 / 10 never take it.
    constexpr bool SK_EPI = EPI == EPI_QKV || EPI == EPI_GELU_SIG || EPI == EPI_GATE_RES || EPI == EPI_BIAS;
    const int nk = args.p[0].K / (fp8 ? 128 : BK);
    const bool sk_can = SK_EPI && args.sk_sync != nullptr && args.sk_part != nullptr && ntiles >= G && nk >= 2 && args.p[0].K == args.p[1].K &&
                        g_gemm_persist_wgs == 0;
    const bool sk_want = var == 19 || ((var == 17 || var == 21) && g_gemm_sk != 0 && ntiles % G != 0);
    // MFMA shape: bit 0 of "gemm_mfma16" = bf16, bit 1 = e4m3, both set by default (16 x 16 blocks).  The 32 x 32 shapes exist in schedules 15 / 17.
    const bool s16 = fp8 ? (g_gemm_mfma16 & 2) != 0 : (g_gemm_mfma16 & 1) != 0;
    if (sk_can && sk_want && s16) {
        if constexpr (SK_EPI) return fp8 ? launch_v<EPI, 19, true, true>(args, G, stream) : launch_v<EPI, 19, false, true>(args, G, stream);
    }
    if (var == 19) var = 17;
    if (var == 22) return launch_gemm4(EPI, fp8, args, ntiles, stream);
    // 17 needs more than one round of tiles (profiles/r03_gemm_notes.md: +1.2 ... +1.5 % at 4.8 / 6.4 rounds, a tie at 1.6 rounds in isolation;
    // in the two-stream pipeline it also pays at 1.6 rounds: g_gemm_persist_min_rounds above); "gemm_persist_wgs" > 0 forces it
    if ((var == 17 || var == 21) && ntiles < (g_gemm_persist_wgs > 0 || g_gemm_persist_min_rounds <= 1 ? G + 1 : g_gemm_persist_min_rounds * G)) var = 15;
    if (!s16) {
        if (var != 15) return fp8 ? launch_v<EPI, 17, true, false>(args, G, stream) : launch_v<EPI, 17, false, false>(args, G, stream);
        return fp8 ? launch_v<EPI, 15, true, false>(args, ntiles, stream) : launch_v<EPI, 15, false, false>(args, ntiles, stream);
    }
    if (var == 17) return fp8 ? launch_v<EPI, 17, true, true>(args, G, stream) : launch_v<EPI, 17, false, true>(args, G, stream);
    if (var == 21) return fp8 ? launch_v<EPI, 21, true, true>(args, G, stream) : launch_v<EPI, 21, false, true>(args, G, stream);
    return fp8 ? launch_v<EPI, 15, true, true>(args, ntiles, stream) : launch_v<EPI, 15, false, true>(args, ntiles, stream);
#endif
}

int launch_gemm(int epilogue, GemmProblem* problems, int nproblems, hipStream_t stream, const GemmWorkspace* workspace) {
    PE_REQUIRE(nproblems >= 1 && nproblems <= 2, "gemm: 1 or 2 problems per launch, got %d", nproblems);
    PE_REQUIRE(g_gemm_variant == 15 || g_gemm_variant == 17 || g_gemm_variant == 19 || g_gemm_variant == 21 || g_gemm_variant == 22,
               "gemm: gemm_variant %d does not exist",
               g_gemm_variant);
    PE_REQUIRE(g_gemm_band >= 1 && g_gemm_band <= 64, "gemm: gemm_band %d out of range", g_gemm_band);
    GemmArgs args;
    int tiles[2] = {0, 0};
    const bool fp8 = problems[0].fp8 != 0;
    for (int i = 0; i < nproblems; ++i) {
        GemmProblem& p = problems[i];
        PE_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem %d (M=%d N=%d K=%d)", i, p.M, p.N, p.K);
        PE_REQUIRE((p.fp8 != 0) == fp8, "gemm: bf16 and e4m3 problems cannot share a launch");
        if (fp8) {
            PE_REQUIRE(p.K % 128 == 0, "gemm(fp8): K=%d must be a multiple of 128", p.K);
            PE_REQUIRE(p.lda % 16 == 0 && p.lda >= p.K, "gemm(fp8): lda=%d must be >= K and a multiple of 16", p.lda);
            PE_REQUIRE(p.scale_a != nullptr, "gemm(fp8): null scale_a");
        } else {
            PE_REQUIRE(p.K % BK == 0, "gemm: K=%d must be a multiple of %d", p.K, BK);
            PE_REQUIRE(p.lda % 8 == 0 && p.lda >= p.K, "gemm: lda=%d must be >= K and a multiple of 8", p.lda);
        }
        PE_REQUIRE(p.N % 8 == 0, "gemm: N=%d must be a multiple of 8", p.N);
        PE_REQUIRE(p.A && p.W, "gemm: null operand");
        if (epilogue == EPI_QKV || epilogue == EPI_QKV_STATS) {
            PE_REQUIRE(epilogue == EPI_QKV || (p.qkv_stats != nullptr && p.stat_rb >= (p.M + 63) / 64 && (p.stat_slot == 0 || p.stat_slot == 1)),
                       "gemm(qkv + statistics): missing partial-sum table");
            PE_REQUIRE(p.N % 384 == 0, "gemm(qkv): N=%d must be 3*H*128", p.N);
            PE_REQUIRE(p.q_out && p.k_out && p.vt_out && p.norm_q_w && p.norm_k_w && p.rope_cos && p.rope_sin,
                       "gemm(qkv): missing qkv epilogue pointers");
            PE_REQUIRE(p.S_pad % 64 == 0 && p.seq_off >= 0 && p.seq_off + p.M <= p.S_pad,
                       "gemm(qkv): bad S_pad=%d seq_off=%d M=%d", p.S_pad, p.seq_off, p.M);
        } else {
            PE_REQUIRE(p.out != nullptr && p.ldo % 8 == 0 && p.ldo >= p.N, "gemm: bad out/ldo=%d", p.ldo);
        }
        if (epilogue == EPI_GATE_RES)
            PE_REQUIRE(p.res != nullptr && p.ldr % 8 == 0, "gemm(gate_res): bad residual");
        PE_REQUIRE(p.pre == nullptr || (p.ldp % 4 == 0 && p.ldp >= p.N), "gemm: bad pre-add operand (ldp=%d)", p.ldp);
        p.tilesM = (p.M + BM - 1) / BM;
        p.tilesN = (p.N + BN - 1) / BN;
        tiles[i] = p.tilesM * p.tilesN;
        args.p[i] = p;
    }
    if (nproblems == 1) args.p[1] = args.p[0];
    args.tiles0 = tiles[0];
    args.ntiles = tiles[0] + tiles[1];
    args.band = g_gemm_band;
    args.skip_ragged = g_gemm_skip_ragged;
    args.direct_epi = g_gemm_direct_epi;
    args.no_epi = g_gemm_no_epi;
    args.cont = g_gemm_cont;
    args.defer = g_gemm_defer;
    args.stash = nullptr;
    args.dbg = g_gemm_dbg;
    args.sk_sync = nullptr;
    args.sk_part = nullptr;
    if (workspace == nullptr && (g_gemm_ws.sync != nullptr || g_gemm_ws.stash != nullptr)) workspace = &g_gemm_ws;      // tests: pe_debug_set_ptr("gemm_workspace", p)
    if (gemm_stash_bytes() != 0 && workspace != nullptr && workspace->stash != nullptr && workspace->stash_bytes >= gemm_stash_bytes() &&
        ((uintptr_t)workspace->stash & 255) == 0)
        args.stash = (char*)workspace->stash;
    if (workspace != nullptr && workspace->sync != nullptr && workspace->bytes >= gemm_workspace_bytes() && ((uintptr_t)workspace->sync & 255) == 0) {
        args.sk_sync = (unsigned*)workspace->sync;
        args.sk_part = (char*)workspace->sync + SK_SYNC_BYTES;
    }
    double flops = 0.0;  // algorithmic 2*M*N*K of the launch (what the roofline fraction is quoted on)
    for (int i = 0; i < nproblems; ++i) flops += 2.0 * problems[i].M * (double)problems[i].N * problems[i].K;
    const int slot = prof_begin(PROF_GEMM, flops, stream);
    int rc;
    switch (epilogue) {
#ifdef PE_GEMM_DEV_EPI
        case PE_GEMM_DEV_EPI: rc = launch_t<PE_GEMM_DEV_EPI>(args, fp8, stream); break;
#else
        case EPI_BIAS: rc = launch_t<EPI_BIAS>(args, fp8, stream); break;
        case EPI_GELU_SIG: rc = launch_t<EPI_GELU_SIG>(args, fp8, stream); break;
        case EPI_GELU_ERF: rc = launch_t<EPI_GELU_ERF>(args, fp8, stream); break;
        case EPI_GATE_RES: rc = launch_t<EPI_GATE_RES>(args, fp8, stream); break;
        case EPI_QKV: rc = launch_t<EPI_QKV>(args, fp8, stream); break;
        case EPI_QKV_STATS: rc = launch_t<EPI_QKV_STATS>(args, fp8, stream); break;
        case EPI_SILU: rc = launch_t<EPI_SILU>(args, fp8, stream); break;
#endif
        default: rc = set_error(PE_ERR_INVALID_ARG, "gemm: unknown epilogue %d", epilogue);
    }
    prof_end(slot, stream);
    return rc;
}

}  // namespace pe
