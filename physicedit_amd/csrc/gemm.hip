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
//     2x4 v_mfma_f32_32x32x16_bf16 tiles (128 fp32 accumulators / lane).
//   * A and W tiles go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip),
//     double buffered (2 x 64 KiB of the CU's 160 KiB), one barrier per K tile.
//   * LDS image is [row][64 k] bf16 = 128 B rows; the 16-B chunk index is XORed with (row>>1)&7 so
//     every ds_read_b128 lane group covers all 64 banks.  LDS-DMA writes lane-linear, so the
//     permutation is applied to the per-lane SOURCE address and again on the read.
//   * operands are fed "swapped" (MFMA A-operand = W fragment, B-operand = activation fragment):
//     each lane then holds 4 consecutive N columns of one M row per accumulator quad.
//   * epilogue: y = bf16(acc + bias) is transposed through the (now free) LDS so every lane owns
//     8 consecutive columns of a row: 16-B bias/gate/residual loads and 16-B stores, 256-B segments.
//   * work-group ids are remapped XCD-aware (8 private L2s) and walk 8-tile-tall bands.
//   * "grouped" launch: up to 2 problems (image stream + text stream) share one grid.
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace pe {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int GEMM_THREADS = 512;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 64 KiB
constexpr int GEMM_LDS = 2 * STAGE_BYTES;        // 128 KiB
constexpr int GEMM_LDS_V10 = 5 * (STAGE_BYTES / 2);  // A x2 + W x3 = 160 KiB (all of a CU's LDS)
constexpr int BAND = 8;
constexpr int GEMM_DEFAULT_VARIANT = 15;  // validated schedule: two staggered wave groups, 2 x 16 MFMAs per K tile (see VAR list)

struct GemmArgs {
    GemmProblem p[2];
    int tiles0;
    long long* dbg;   // VAR 14 only: per work-group s_memtime stamps [grid][8] (pe_debug_set_ptr("gemm_stamps", p))
    // stream-K tail (VAR 15): blocks [0, n_full) compute whole tiles; the last sk_tiles tiles (tile ids n_full ..) are
    // cut along K into SK_WGS equal ranges of K tiles, one per block of the tail phase (blocks n_full .. n_full+SK_WGS-1)
    int n_full, sk_tiles;
    float* sk_ws;          // [SK_WGS][256*256] fp32 partial accumulators (one slot per tail block)
    unsigned* sk_flags;    // [SK_WGS] epoch of the last launch whose partial in that slot is complete
    unsigned* sk_status;   // [1] set to 1 if an owner gave up waiting (never in a correct run)
    unsigned sk_epoch;
};
constexpr int SK_WGS = 256;               // one tail block per CU
constexpr int SK_SLOT_FLOATS = BM * BN;   // 256 KiB per partial
long long* g_gemm_dbg = nullptr;

// VAR 14 = VAR 12 + time stamps of wave 0 (profiling build of the default schedule; never the production variant)
#define PE_STAMP(k)                                                                                     \
    do {                                                                                                \
        if constexpr (VAR == 14 || VAR == 15 || VAR == 16) {                                            \
            if (args.dbg != nullptr && threadIdx.x == 0) args.dbg[(size_t)blockIdx.x * 8 + (k)] = (long long)__builtin_readcyclecounter(); \
        }                                                                                               \
    } while (0)

PE_DEV int perm16(int j) { return (j & 3) | ((j & 4) << 1) | ((j & 8) >> 1); }

// VAR selects the main-loop schedule (A/B-tested in one process through pe_debug_set("gemm_variant"); the full study
// incl. the ablation variants that no longer live here is profiles/r01_gemm_ablation.md):
//   10  (round-1 default, kept as the A/B reference) pipelined clusters: fragments double buffered in registers, tile barrier
//       before the LAST cluster, staging spread over the clusters, MFMA / ds_read / LDS-DMA interleave pinned with
//       sched_group_barrier, three-deep W ring and counted vmcnt
//   14  "ping-pong": the two wave groups run the same stream one barrier apart, 4 phases x 8 MFMAs per K tile; with s_memtime
//       stamps (profiling only).  (The code paths guarded by VAR >= 12 / VAR == 13 belong to this family; 0, 8, 12 and 13
//       themselves are no longer instantiated.)
//   15  (default) ping-pong with 2 phases x 16 MFMAs per K tile, A half tiles staged by the group that reads them
//   16  15 + stream-K tail (the tiles of a partially filled last round are cut along K into 256 equal ranges, fp32
//       partials exchanged through a workspace).  Correct and deterministic, but SLOWER on MI355X (profiles/
//       r02_gemm_notes.md): the ranges of different tiles sit at different K offsets, so the L2 sharing of A/W panels
//       between the tiles of a band is lost, and 64 MB of partials move twice.  Kept as an experiment knob.
//
// FP8 = true: operands are OCP e4m3 bytes (activation rows quantised by quantize_rows_e4m3, weights stored in e4m3),
// the K tile is 128 elements (the SAME 128-B LDS rows, staging and swizzle), the MFMA is the CDNA4 block-scaled
// v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (2x the bf16 rate), and the epilogue starts with
// y = bf16(acc * scale_a[m] + bias[n])  (AutoWrappedLinear.fp8_linear, vram_management/layers.py:115-151).
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

// One output tile (or, in the stream-K tail, the K range [k_lo, k_hi) of one): `bid` = tile id in the banded order.
// sk_c < 0: a whole tile.  sk_c >= 0: this block is tail block sk_c; a range that does not start at K tile 0 ends
// with its fp32 accumulators dumped to slot sk_c (+ flag); the range that starts at 0 OWNS the tile: it adds the
// partials of the blocks sk_c+1.. that cover the rest of the tile's K and runs the epilogue.
template <int EPI, int VAR, bool FP8>
__device__ __forceinline__ void gemm_tile(const GemmArgs& args, char* smem, int bid, int k_lo, int k_hi, int sk_c) {
    constexpr int ES = FP8 ? 1 : 2;        // bytes per operand element
    constexpr int KT_BYTES = 128;          // one K tile of a row, in bytes (64 bf16 / 128 e4m3)
    int lane_ = lane_id();
    if constexpr (VAR == 16)
        asm volatile("" : "+v"(lane_));    // opaque per call: keeps LICM from hoisting the epilogue's per-lane address math out
    const int lane = lane_;                // of the caller's (<= 2 iteration) range loop and spilling it across the main loop
    const int w = wave_id();
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = w >> 1, wn = w & 1;

    PE_STAMP(0);
    const int pi = bid >= args.tiles0 ? 1 : 0;
    const GemmProblem& P = args.p[pi];
    bid -= pi ? args.tiles0 : 0;
    const int M = P.M, N = P.N, K = P.K;
    int m0, n0;
    {
        const int tilesM = P.tilesM, tilesN = P.tilesN;
        const int per_band = BAND * tilesN;
        const int band = bid / per_band;
        const int rem = bid - band * per_band;
        const int gm = min(BAND, tilesM - band * BAND);
        const int tn = rem / gm;
        const int tm = band * BAND + (rem - tn * gm);
        m0 = tm * BM;
        n0 = tn * BN;
    }

    // ---- staging sources: wave w moves pieces w*4..w*4+3 (1 KiB = 8 rows x 128 B) of A and of W
    const char* a_src[4];
    const char* w_src[4];
    {
        const char* A = (const char*)P.A;
        const char* W = (const char*)P.W;
        const int rin = lane >> 3, slot = lane & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // piece (1 KiB = 8 rows) i of this wave.  VAR 0/8/10: whole tiles, piece 4w + i.  VAR 12-14: 128-row half
            // tiles, piece = half*16 + 2w + j.  VAR 15: A pieces come from the wave group's OWN half (the only one it
            // reads), piece = grp*16 + 4(w&3) + i; W pieces 4w + i.
            const int piece_a = VAR >= 15 ? (w >> 2) * 16 + (w & 3) * 4 + i : VAR >= 12 ? (i >> 1) * 16 + w * 2 + (i & 1) : w * 4 + i;
            const int piece_w = VAR >= 15 ? w * 4 + i : piece_a;
            const int row_a = piece_a * 8 + rin, row_w = piece_w * 8 + rin;
            const int gr = min(m0 + row_a, M - 1);
            const int gn = min(n0 + row_w, N - 1);
            a_src[i] = A + (size_t)gr * P.lda * ES + (slot ^ ((row_a >> 1) & 7)) * 16;
            w_src[i] = W + (size_t)gn * K * ES + (slot ^ ((row_w >> 1) & 7)) * 16;
        }
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE_BYTES + w * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16(a_src[i] + kt * KT_BYTES, base + i * 1024);
            glds16(w_src[i] + kt * KT_BYTES, base + BM * BK * 2 + i * 1024);
        }
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // per-lane fragment addressing: row = tile_row + l31, chunk = kk*2 + h, swizzle (row>>1)&7
    const int sw = (l31 >> 1) & 7;

    const int nk = K * ES / KT_BYTES;
    if constexpr (!FP8 && VAR < 12) stage(0, 0);
    if constexpr (VAR >= 12) {
        // "Ping-pong" schedule: the two wave groups (waves 0-3 = rows 0-127, waves 4-7 = rows 128-255; waves w and w+4
        // share a SIMD) run the SAME instruction stream one barrier apart.  Each K tile is four phases per wave, one per
        // 32x64 quadrant of the wave's 64x128 C tile:
        //     load part : ds_read the quadrant's missing fragments, issue 2 LDS-DMA pieces (one 128-row half tile per phase
        //                 and work-group), barrier
        //     MFMA part : 8 (bf16) / 4 (e4m3) MFMAs under s_setprio 1, barrier
        // so at any time every SIMD has one wave in its MFMA part and one in its load part: LDS latency, the DMA issue
        // and the barrier skew of one group hide under the other group's MFMAs (v10 interleaves them inside each wave).
        //     phase  quadrant (mi, nj)   fragment reads               stages (for tile t+2)
        //     p0     (0, 0)              A[mi 0] x KS, W[ni 0,1] x KS  W half 0 -> W ring slot (t+2)%3
        //     p1     (1, 0)              A[mi 1] x KS                  W half 1
        //     p2     (1, 1)              W[ni 2,3] x KS                A half 0 -> A buffer t&1 (its last read: p1)
        //     p3     (0, 1)              -                             A half 1 ; then s_waitcnt vmcnt(8)
        // LDS: A 2 x 32 KiB + W 3 x 32 KiB (v10's rings).  RAW: a wave's pieces of tile t+1 were issued during tile t-1;
        // vmcnt(8) in p3 of tile t retires them (8 younger pieces = tile t+2's) and both groups pass >= 1 barrier
        // between that wait and the first ds_read of tile t+1.  WAR: every restaged region was last read >= 2 barriers
        // earlier (A half g only by group g in p0/p1; W ring slot of tile t-1).
        using FragT = typename std::conditional<FP8, i32x8, bf16x8>::type;
        constexpr int KS = FP8 ? 2 : 4;            // MFMA k-steps per K tile
        constexpr int A_BYTES = BM * KT_BYTES, W_BYTES = BN * KT_BYTES;
        char* const a_base = smem;
        char* const w_base = smem + 2 * A_BYTES;
        const int grp = w >> 2;
        const int a_off = (wm * 64 + l31) * 128;
        const int w_off = (wn * 128 + l31) * 128;
        auto rd = [&](const char* rowp, int ks) -> FragT {
            if constexpr (FP8) {
                const int c0 = 4 * ks + 2 * h;
                const i32x4 lo = *(const i32x4*)(rowp + ((c0 ^ sw) << 4));
                const i32x4 hi = *(const i32x4*)(rowp + (((c0 + 1) ^ sw) << 4));
                return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            } else {
                return *(const bf16x8*)(rowp + (((ks * 2 + h) ^ sw) << 4));
            }
        };
#define PE_BAR()                                   \
    do {                                           \
        __builtin_amdgcn_sched_barrier(0);         \
        __builtin_amdgcn_s_barrier();              \
        __builtin_amdgcn_sched_barrier(0);         \
    } while (0)
#define PE_MMA(a, mi, nj)                          \
    do {                                           \
        PE_BAR();                                  \
        __builtin_amdgcn_s_setprio(1);             \
        mma(a, mi, nj);                            \
        __builtin_amdgcn_s_setprio(0);             \
        PE_BAR();                                  \
    } while (0)
        if constexpr (VAR >= 15) {
            // Two phases of 16 MFMAs per K tile (half the barriers of VAR 12: the ~60-cycle barrier/turn-around cost per
            // phase is paid per 512 instead of per 256 MFMA cycles).  Phase p = row block mi = p of the wave's C tile
            // against all four column blocks:
            //     p0: reads A[mi 0] x KS + W[ni 0..3] x KS (20 ds_read_b128), stages this wave's 4 pieces of A(kt+1)
            //     p1: reads A[mi 1] x KS                                    , stages its 4 pieces of W(kt+2), vmcnt(4)
            // A half tiles are staged by the group that reads them (piece_a above), so "every wave of my group passed
            // the barrier behind its lgkmcnt(0)" is all the WAR protection A's two buffers need: A(kt+1) overwrites
            // A(kt-1), last read in p1 of tile kt-1.  W(kt+2) overwrites W(kt-1) (3-deep ring), last read by group 1 in
            // p0 of tile kt-1, i.e. >= 4 barriers earlier.  RAW: vmcnt(4) in p1 retires A(kt+1), W(kt+1); both groups
            // pass a barrier between that wait and the first read of tile kt+1.
            FragT fa[KS], fw4[4][KS];
            auto rd_a1 = [&](const char* Sa, int mi) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) fa[ks] = rd(Sa + a_off + mi * 4096, ks);
            };
            auto rd_w4 = [&](const char* Sw) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) fw4[ni][ks] = rd(Sw + w_off + ni * 4096, ks);
            };
            auto mma16 = [&](int mi) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        if constexpr (FP8)
                            acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                                fw4[ni][ks], fa[ks], acc[mi][ni], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                        else
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw4[ni][ks], fa[ks], acc[mi][ni], 0, 0, 0);
                    }
            };
            auto st_a4 = [&](int i) {          // i = K tile index relative to k_lo
                const int tc = min(k_lo + i, nk - 1);
                char* base = a_base + (i & 1) * A_BYTES + (grp * 16 + (w & 3) * 4) * 1024;
#pragma unroll
                for (int j = 0; j < 4; ++j) glds16(a_src[j] + tc * KT_BYTES, base + j * 1024);
            };
            auto st_w4 = [&](int i, int slot) {
                const int tc = min(k_lo + i, nk - 1);
                char* base = w_base + slot * W_BYTES + w * 4096;
#pragma unroll
                for (int j = 0; j < 4; ++j) glds16(w_src[j] + tc * KT_BYTES, base + j * 1024);
            };
#define PE_MMA16(mi)                               \
    do {                                           \
        PE_BAR();                                  \
        __builtin_amdgcn_s_setprio(1);             \
        mma16(mi);                                 \
        __builtin_amdgcn_s_setprio(0);             \
        PE_BAR();                                  \
    } while (0)
            st_a4(0); st_w4(0, 0); st_w4(1, 1);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // A(0), W(0) landed; W(1) may still fly
            PE_BAR();
            if (grp == 1) __builtin_amdgcn_s_barrier();        // stagger
            PE_STAMP(1);
            int ws = 0;
            const int nloc = k_hi - k_lo;
            for (int kt = 0; kt < nloc; ++kt) {
                const char* Sa = a_base + (kt & 1) * A_BYTES;
                const int ws_n1 = ws == 2 ? 0 : ws + 1;
                const int ws_n2 = ws_n1 == 2 ? 0 : ws_n1 + 1;
                // p0
                rd_a1(Sa, 0); rd_w4(w_base + ws * W_BYTES);
                st_a4(kt + 1);
                PE_MMA16(0);
                // p1
                rd_a1(Sa, 1);
                st_w4(kt + 2, ws_n2);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // this wave's pieces of A(kt+1), W(kt+1) landed
                PE_MMA16(1);
                ws = ws_n1;
            }
            if (grp == 0) __builtin_amdgcn_s_barrier();        // re-align the two groups
            PE_STAMP(2);
#undef PE_MMA16
        } else {
            constexpr bool BAL = VAR == 13;   // 13: A[mi 0] of the next tile is pre-read in p3 (reads 8/4/8/4 instead of 12/4/8/0)
            FragT fa0[KS], fa0b[KS], fa1[KS], fw[2][KS];
            auto rd_a = [&](const char* Sa, int mi, FragT (&f)[KS]) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) f[ks] = rd(Sa + a_off + mi * 4096, ks);
            };
            auto rd_w = [&](const char* Sw, int nj) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) fw[j][ks] = rd(Sw + w_off + (nj * 2 + j) * 4096, ks);
            };
            auto mma = [&](FragT (&a)[KS], int mi, int nj) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if constexpr (FP8)
                            acc[mi][nj * 2 + j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                                fw[j][ks], a[ks], acc[mi][nj * 2 + j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                        else
                            acc[mi][nj * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[j][ks], a[ks], acc[mi][nj * 2 + j], 0, 0, 0);
                    }
            };
            auto st_a = [&](int t, int half) {
                const int tc = min(t, nk - 1);
                char* base = a_base + (t & 1) * A_BYTES + (half * 16 + w * 2) * 1024;
#pragma unroll
                for (int j = 0; j < 2; ++j) glds16(a_src[half * 2 + j] + tc * KT_BYTES, base + j * 1024);
            };
            auto st_w = [&](int t, int slot, int half) {
                const int tc = min(t, nk - 1);
                char* base = w_base + slot * W_BYTES + (half * 16 + w * 2) * 1024;
#pragma unroll
                for (int j = 0; j < 2; ++j) glds16(w_src[half * 2 + j] + tc * KT_BYTES, base + j * 1024);
            };
            st_a(0, 0); st_a(0, 1); st_w(0, 0, 0); st_w(0, 0, 1);
            st_a(1, 0); st_a(1, 1); st_w(1, 1, 0); st_w(1, 1, 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // tile 0 landed; tile 1 may still fly
            PE_BAR();
            if constexpr (BAL) rd_a(a_base, 0, fa0);
            if (grp == 1) __builtin_amdgcn_s_barrier();        // stagger: group 1 runs one barrier behind group 0
            PE_STAMP(1);
            // one K tile; BAL: `cur` holds A[mi 0] of tile kt (read during the previous tile), `nxt` receives tile kt+1's
            auto tile = [&](int kt, int ws_cur, FragT (&cur)[KS], FragT (&nxt)[KS]) __attribute__((always_inline)) {
                const char* Sa = a_base + (kt & 1) * A_BYTES;
                const char* San = a_base + ((kt + 1) & 1) * A_BYTES;
                const int ws_n1 = ws_cur == 2 ? 0 : ws_cur + 1;
                const int ws_n2 = ws_n1 == 2 ? 0 : ws_n1 + 1;
                const char* Sw = w_base + ws_cur * W_BYTES;
                // p0
                if constexpr (!BAL) rd_a(Sa, 0, cur);
                rd_w(Sw, 0);
                st_w(kt + 2, ws_n2, 0);
                PE_MMA(cur, 0, 0);
                // p1
                rd_a(Sa, 1, fa1);
                st_w(kt + 2, ws_n2, 1);
                PE_MMA(fa1, 1, 0);
                // p2
                rd_w(Sw, 1);
                st_a(kt + 2, 0);
                if constexpr (BAL) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // this wave's pieces of tile kt+1 landed
                PE_MMA(fa1, 1, 1);
                // p3
                if constexpr (BAL) rd_a(San, 0, nxt);   // both groups passed a barrier since every wave's vmcnt(6)
                st_a(kt + 2, 1);
                if constexpr (!BAL) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // this wave's pieces of tile kt+1 landed
                PE_MMA(cur, 0, 1);
            };
            int ws = 0;
            int kt = 0;
            for (; kt + 1 < nk; kt += 2) {
                tile(kt, ws, fa0, fa0b);
                ws = ws == 2 ? 0 : ws + 1;
                if constexpr (BAL) tile(kt + 1, ws, fa0b, fa0); else tile(kt + 1, ws, fa0, fa0b);
                ws = ws == 2 ? 0 : ws + 1;
            }
            if (kt < nk) tile(kt, ws, fa0, fa0b);
            if (grp == 0) __builtin_amdgcn_s_barrier();        // re-align the two groups
            PE_STAMP(2);
        }
#undef PE_MMA
#undef PE_BAR
    } else if constexpr (FP8) {
        // v10's structure (A 2 x 32 KiB + W 3 x 32 KiB ring, tile barrier with vmcnt(4)) on 64-cycle MFMAs:
        // four clusters of 4 MFMAs per K tile.  Cluster c covers k-half c>>1 and output column pair c&1:
        //   c0: A(k0) x W(k0, ni 0,1)   c1: A(k0) x W(k0, ni 2,3)   c2: A(k1) x W(k1, ni 0,1)   c3: A(k1) x W(k1, ni 2,3)
        // Fragment of one 32-row tile for one MFMA = 32 bytes of k per lane (k = 32*h .. +31) = two 16-B chunks.
        constexpr int A_BYTES = BM * KT_BYTES, W_BYTES = BN * KT_BYTES;
        char* const a_base = smem;
        char* const w_base = smem + 2 * A_BYTES;
        auto frag = [&](const char* rowp, int k2) -> i32x8 {
            const int c0 = 4 * k2 + 2 * h;
            const i32x4 lo = *(const i32x4*)(rowp + ((c0 ^ sw) << 4));
            const i32x4 hi = *(const i32x4*)(rowp + (((c0 + 1) ^ sw) << 4));
            return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        };
        auto frag_a = [&](const char* Sa, int k2, i32x8 (&af)[2]) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) af[mi] = frag(Sa + (wm * 64 + mi * 32 + l31) * 128, k2);
        };
        auto frag_w = [&](const char* Sw, int k2, int pair, i32x8 (&wf)[2]) {
#pragma unroll
            for (int j = 0; j < 2; ++j) wf[j] = frag(Sw + (wn * 128 + (pair * 2 + j) * 32 + l31) * 128, k2);
        };
        const int unit = 0x7f7f7f7f;   // E8M0 block scales = 2^0
        auto mma = [&](i32x8 (&af)[2], i32x8 (&wf)[2], int pair) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[mi][pair * 2 + j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                        wf[j], af[mi], acc[mi][pair * 2 + j], 0, 0, 0, unit, 0, unit);
        };
        auto stage_a = [&](int t, int first, int count) {
            const int tc = min(t, nk - 1);
            char* base = a_base + (t & 1) * A_BYTES + w * 4096;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i >= first && i < first + count) glds16(a_src[i] + tc * KT_BYTES, base + i * 1024);
        };
        auto stage_w = [&](int t, int slot, int first, int count) {
            const int tc = min(t, nk - 1);
            char* base = w_base + slot * W_BYTES + w * 4096;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i >= first && i < first + count) glds16(w_src[i] + tc * KT_BYTES, base + i * 1024);
        };
#define PE_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#define PE_CLUSTER8(NDS)                                          \
    do {                                                          \
        PE_SGB(0x008, 1); PE_SGB(0x100, (NDS) / 2);               \
        PE_SGB(0x008, 1); PE_SGB(0x100, (NDS) / 2);               \
        PE_SGB(0x008, 1); PE_SGB(0x020, 1);                       \
        PE_SGB(0x008, 1); PE_SGB(0x020, 1);                       \
    } while (0)
        stage_a(0, 0, 4);
        stage_w(0, 0, 0, 4);
        stage_w(1, 1, 0, 4);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // A(0), W(0) landed; W(1) may still fly
        __syncthreads();
        i32x8 fa0[2], fa1[2], fwp[2], fwq[2];
        frag_a(a_base, 0, fa0);
        frag_w(w_base, 0, 0, fwp);
        stage_a(1, 0, 2);
        int ws_cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const char* Sa = a_base + (kt & 1) * A_BYTES;
            const char* San = a_base + ((kt + 1) & 1) * A_BYTES;
            const int ws_n1 = ws_cur == 2 ? 0 : ws_cur + 1;
            const int ws_n2 = ws_n1 == 2 ? 0 : ws_n1 + 1;
            const char* Sw = w_base + ws_cur * W_BYTES;
            const char* Swn = w_base + ws_n1 * W_BYTES;
            // cluster 0
            frag_w(Sw, 0, 1, fwq); frag_a(Sa, 1, fa1);
            stage_a(kt + 1, 2, 2);
            mma(fa0, fwp, 0);
            PE_CLUSTER8(8);
            // cluster 1
            frag_w(Sw, 1, 0, fwp);
            stage_w(kt + 2, ws_n2, 0, 2);
            mma(fa0, fwq, 1);
            PE_CLUSTER8(4);
            // cluster 2
            frag_w(Sw, 1, 1, fwq);
            stage_w(kt + 2, ws_n2, 2, 2);
            mma(fa1, fwp, 0);
            PE_CLUSTER8(4);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");   // all but the 4 newest (= W(kt+2))
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // cluster 3
            frag_a(San, 0, fa0); frag_w(Swn, 0, 0, fwp);
            stage_a(kt + 2, 0, 2);
            mma(fa1, fwq, 1);
            PE_CLUSTER8(8);
            ws_cur = ws_n1;
        }
#undef PE_CLUSTER8
#undef PE_SGB
    } else if constexpr (VAR == 10) {
        // v8 + a THREE-deep W ring: the weight matrix is the cold operand of every block GEMM (each of
        // the 40 GB of weights is touched once per forward), the activation tile is cache resident.  LDS:
        // A 2 x 32 KiB + W 3 x 32 KiB = the CU's whole 160 KiB.  W(kt+2) is issued during tile kt, AFTER
        // A(kt+1) in program order, so the tile barrier can wait with vmcnt(4): everything but the four
        // newest W pieces (HBM latency then spans ~1.5 K tiles instead of ~0.75).
        constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2;
        char* const a_base = smem;
        char* const w_base = smem + 2 * A_BYTES;
        auto frag_a = [&](const char* Sa, int kk, bf16x8 (&af)[2]) {
            const int coff = ((kk * 2 + h) ^ sw) << 4;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) af[mi] = *(const bf16x8*)(Sa + (wm * 64 + l31) * 128 + mi * 32 * 128 + coff);
        };
        auto frag_w = [&](const char* Sw, int kk, bf16x8 (&wf)[4]) {
            const int coff = ((kk * 2 + h) ^ sw) << 4;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) wf[ni] = *(const bf16x8*)(Sw + (wn * 128 + l31) * 128 + ni * 32 * 128 + coff);
        };
        auto mma = [&](bf16x8 (&af)[2], bf16x8 (&wf)[4]) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], af[mi], acc[mi][ni], 0, 0, 0);
        };
        auto stage_a = [&](int t, int first, int count) {
            const int tc = min(t, nk - 1);
            char* base = a_base + (t & 1) * A_BYTES + w * 4096;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i >= first && i < first + count) glds16(a_src[i] + tc * KT_BYTES, base + i * 1024);
        };
        auto stage_w = [&](int t, int slot, int first, int count) {
            const int tc = min(t, nk - 1);
            char* base = w_base + slot * W_BYTES + w * 4096;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i >= first && i < first + count) glds16(w_src[i] + tc * KT_BYTES, base + i * 1024);
        };
#define PE_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#define PE_CLUSTER_SCHED(NVMEM)                                    \
    do {                                                          \
        PE_SGB(0x008, 1); PE_SGB(0x100, 2);                       \
        PE_SGB(0x008, 1); PE_SGB(0x100, 2);                       \
        PE_SGB(0x008, 1); PE_SGB(0x100, 2);                       \
        for (int v_ = 0; v_ < (NVMEM); ++v_) { PE_SGB(0x008, 1); PE_SGB(0x020, 1); } \
        PE_SGB(0x008, 5 - (NVMEM));                               \
    } while (0)
        // prologue.  The common code above staged tile 0 into the 2-stage layout of the other variants;
        // wait for it to drain, then restage in this variant's layout (runs once per work-group).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stage_a(0, 0, 4);
        stage_w(0, 0, 0, 4);
        stage_w(1, 1, 0, 4);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // A(0), W(0) landed; W(1) may still fly
        __syncthreads();
        bf16x8 fa0[2], fw0[4], fa1[2], fw1[4];
        frag_a(a_base, 0, fa0);
        frag_w(w_base, 0, fw0);
        stage_a(1, 0, 2);
        int ws_cur = 0;                       // W ring slot of tile kt
        for (int kt = 0; kt < nk; ++kt) {
            const char* Sa = a_base + (kt & 1) * A_BYTES;
            const char* San = a_base + ((kt + 1) & 1) * A_BYTES;
            const int ws_n1 = ws_cur == 2 ? 0 : ws_cur + 1;   // slot of tile kt+1
            const int ws_n2 = ws_n1 == 2 ? 0 : ws_n1 + 1;     // slot of tile kt+2 (held tile kt-1: dead)
            const char* Sw = w_base + ws_cur * W_BYTES;
            const char* Swn = w_base + ws_n1 * W_BYTES;
            // cluster 0
            frag_a(Sa, 1, fa1); frag_w(Sw, 1, fw1);
            stage_a(kt + 1, 2, 2);
            mma(fa0, fw0);
            PE_CLUSTER_SCHED(2);
            // cluster 1
            frag_a(Sa, 2, fa0); frag_w(Sw, 2, fw0);
            stage_w(kt + 2, ws_n2, 0, 2);
            mma(fa1, fw1);
            PE_CLUSTER_SCHED(2);
            // cluster 2
            frag_a(Sa, 3, fa1); frag_w(Sw, 3, fw1);
            stage_w(kt + 2, ws_n2, 2, 2);
            mma(fa0, fw0);
            PE_CLUSTER_SCHED(2);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");   // all but the 4 newest (= W(kt+2))
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // cluster 3
            frag_a(San, 0, fa0); frag_w(Swn, 0, fw0);
            stage_a(kt + 2, 0, 2);
            mma(fa1, fw1);
            PE_CLUSTER_SCHED(2);
            ws_cur = ws_n1;
        }
#undef PE_CLUSTER_SCHED
#undef PE_SGB
    }
    static_assert(FP8 || VAR == 0 || VAR == 8 || VAR == 10 || VAR == 12 || VAR == 13 || VAR == 14 || VAR == 15 || VAR == 16, "unknown GEMM schedule");

    // ------------------------------------------------------------------------------------------
    // epilogue.  acc[mi][ni][4q+r] = C[m0 + wm*64 + mi*32 + l31][n0 + wn*128 + ni*32 + 8q + 4h + r]
    // ------------------------------------------------------------------------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain every LDS-DMA (incl. the clamped tail tiles) before LDS is reused
    __syncthreads();
    PE_STAMP(3);
    if constexpr (VAR == 16) {
        if (sk_c >= 0) {
            // partial layout: [wave][mi][ni][quad][lane] f32x4 -- the accumulator registers as they are, 1 KiB per store
            auto slot_ptr = [&](int c) { return args.sk_ws + (size_t)c * SK_SLOT_FLOATS + ((size_t)w * 32 * 64 + lane) * 4; };
            if (k_lo > 0) {
                // not the owner: publish the partial (cdna guide G16: plain stores, every wave drains, barrier, ONE lane
                // releases at agent scope and only then stores the flag)
                float* dst = slot_ptr(sk_c);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x4 v;
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = acc[mi][ni][4 * q + r];
                            *(f32x4*)(dst + (size_t)((mi * 4 + ni) * 4 + q) * 64 * 4) = v;
                        }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(args.sk_flags + sk_c, args.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return;
            }
            if (k_hi < nk) {
                // owner: the blocks sk_c+1, sk_c+2, .. cover [k_hi, nk) of this tile, each with its FIRST range
                const long long total = (long long)args.sk_tiles * nk;
                const long long tile_end = (long long)(bid - args.n_full + 1) * nk;
                for (int cc = sk_c + 1; cc < SK_WGS && (long long)cc * total / SK_WGS < tile_end; ++cc) {
                    if (threadIdx.x == 0) {
                        unsigned spins = 0;
                        while (__hip_atomic_load(args.sk_flags + cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != args.sk_epoch) {
                            __builtin_amdgcn_s_sleep(8);
                            if (++spins > (1u << 22)) {          // ~ 1 s: give up loudly instead of hanging the GPU
                                __hip_atomic_store(args.sk_status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                break;
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    }
                    __syncthreads();
                    const float* src = slot_ptr(cc);
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const f32x4 v = *(const f32x4*)(src + (size_t)((mi * 4 + ni) * 4 + q) * 64 * 4);
#pragma unroll
                                for (int r = 0; r < 4; ++r) acc[mi][ni][4 * q + r] += v[r];
                            }
                }
            }
        }
    }
    // this wave's [64 rows][128 cols] bf16 staging tile: 16-B chunk c of row r sits at chunk c ^ (r & 15), and its two 8-B
    // halves are swapped when r & 8 (rows r and r ^ 8 would otherwise land on the same banks in one ds_write_b64
    // lane group: measured 7.3k instead of ~3.5k cycles for the 64 writes per lane)
    char* E = smem + w * 16384;
    auto unswap = [](bf16x8 v, int row) -> bf16x8 {
        return (row & 8) ? __builtin_shufflevector(v, v, 4, 5, 6, 7, 0, 1, 2, 3) : v;
    };
    const int nw0 = n0 + wn * 128;
    const int mw0 = m0 + wm * 64;
    bf16x8 rv16[EPI == EPI_GATE_RES ? 16 : 1];   // residual rows of the gated-residual epilogue, prefetched
    {
        const bf16* bias = (const bf16*)P.bias;
        const bf16* pre = (const bf16*)P.pre;   // hot LoRA: `out + x @ A.T @ B.T` (vram_management/layers.py:179-180)
        float sa[2] = {1.f, 1.f};               // FP8: per-row activation scale (fp8_linear's scale_a)
        if constexpr (FP8) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) sa[mi] = P.scale_a[min(mw0 + mi * 32 + l31, M - 1)];
        }
        // bias first, then (EPI_GATE_RES) all 16 residual rows of this lane: 16-B loads that stay in flight under the LDS
        // staging below (they used to be issued 4 at a time inside the store loop: 15-21k cycles of exposed latency)
        bf16x4 bvs[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int n = nw0 + (i >> 2) * 32 + 8 * (i & 3) + 4 * h;
            bvs[i] = bf16x4{0, 0, 0, 0};
            if (bias != nullptr && n < N) bvs[i] = *(const bf16x4*)(bias + n);
        }
        if constexpr (EPI == EPI_GATE_RES) {
            const int n = nw0 + (lane & 15) * 8;
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int m = mw0 + it * 4 + (lane >> 4);
                rv16[it] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                if (m < M && n < N) rv16[it] = *(const bf16x8*)((const bf16*)P.res + (size_t)m * P.ldr + n);
            }
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = nw0 + ni * 32 + 8 * q + 4 * h;
                float b[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) b[r] = (float)bvs[ni * 4 + q][r];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    bf16x4 y;
                    if constexpr (FP8) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[r] = (bf16)(acc[mi][ni][4 * q + r] * sa[mi] + b[r]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[r] = (bf16)(acc[mi][ni][4 * q + r] + b[r]);
                    }
                    const int row = mi * 32 + l31;
                    if (pre != nullptr && n < N && mw0 + row < M) {
                        // y = pre + y : the linear's own (already rounded) output plus this low-rank product
                        const bf16x4 pv = *(const bf16x4*)(pre + (size_t)(mw0 + row) * P.ldp + n);
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[r] = (bf16)((float)pv[r] + (float)y[r]);
                    }
                    const int c = ni * 4 + q;
                    *(bf16x4*)(E + row * 256 + ((c ^ (row & 15)) << 4) + ((h ^ ((row >> 3) & 1)) << 3)) = y;
                }
            }
    }
    // no barrier here: a wave reads back only its own staging tile, and one wave's LDS accesses execute in order
    PE_STAMP(4);

    if constexpr (EPI == EPI_QKV) {
        const int HD = N / 3;
        const int section = nw0 / HD;  // wave-uniform: the wave's 128 columns are exactly one head
        const int head = (nw0 - section * HD) >> 7;
        const int S_pad = P.S_pad;
        if (section < 2) {
            const bf16* nw = (const bf16*)(section == 0 ? P.norm_q_w : P.norm_k_w);
            bf16* dst = (bf16*)(section == 0 ? P.q_out : P.k_out);
            const int c = lane & 15;
            const bf16x8 wv = *(const bf16x8*)(nw + c * 8);
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int row = it * 4 + (lane >> 4);
                const int m = mw0 + row;
                const bf16x8 v = unswap(*(const bf16x8*)(E + row * 256 + ((c ^ (row & 15)) << 4)), it << 2);   // row & 8 == (it << 2) & 8
                float y[8];
                float ss = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    y[j] = (float)v[j];
                    ss += y[j] * y[j];
                }
                ss += __shfl_xor(ss, 1, 64);
                ss += __shfl_xor(ss, 2, 64);
                ss += __shfl_xor(ss, 4, 64);
                ss += __shfl_xor(ss, 8, 64);
                // RMSNorm(128, eps 1e-6): models/utils.py:250-257
                const float rs = rsqrtf(ss * (1.0f / 128.0f) + 1e-6f);
                if (m < M) {
                    const f32x4 cs = *(const f32x4*)(P.rope_cos + (size_t)m * 64 + c * 4);
                    const f32x4 sn = *(const f32x4*)(P.rope_sin + (size_t)m * 64 + c * 4);
                    bf16x8 o;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const float x0 = bf16r(bf16r(y[2 * jj] * rs) * (float)wv[2 * jj]);
                        const float x1 = bf16r(bf16r(y[2 * jj + 1] * rs) * (float)wv[2 * jj + 1]);
                        // apply_rotary_emb_qwen: fp32 complex multiply (qwen_image_dit.py:51-57)
                        o[2 * jj] = (bf16)(x0 * cs[jj] - x1 * sn[jj]);
                        o[2 * jj + 1] = (bf16)(x0 * sn[jj] + x1 * cs[jj]);
                    }
                    *(bf16x8*)(dst + ((size_t)head * S_pad + P.seq_off + m) * 128 + c * 8) = o;
                }
            }
        } else {
            // V: written TRANSPOSED, Vt[head][d][pos(token)], tokens permuted inside aligned 16-groups
            // (pos = perm16) so the attention kernel's P.V MFMA needs no cross-lane shuffle.
            bf16* vt = (bf16*)P.vt_out + (size_t)head * 128 * S_pad;
            const int seq0 = P.seq_off + mw0;
            const int valid = min(64, M - mw0);
            if ((seq0 & 15) == 0) {
#pragma unroll 2
                for (int it = 0; it < 16; ++it) {
                    const int id = it * 64 + lane;
                    const int d = id >> 3, tg = id & 7;
                    const int gi = tg >> 1, hh = tg & 1;
                    unsigned short e[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int row = gi * 16 + (j < 4 ? 4 * hh + j : 8 + 4 * hh + (j - 4));
                        e[j] = *(const unsigned short*)(E + row * 256 + (((d >> 3) ^ (row & 15)) << 4) + ((((d & 7) * 2) ^ (row & 8))));
                    }
                    bf16* dstp = vt + (size_t)d * S_pad + seq0 + gi * 16 + hh * 8;
                    if (gi * 16 + 16 <= valid) {
                        u32x4 pk;
#pragma unroll
                        for (int j = 0; j < 4; ++j) pk[j] = (uint32_t)e[2 * j] | ((uint32_t)e[2 * j + 1] << 16);
                        *(u32x4*)dstp = pk;
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int tok = gi * 16 + (j < 4 ? 4 * hh + j : 8 + 4 * hh + (j - 4));
                            if (tok < valid) ((unsigned short*)dstp)[j] = e[j];
                        }
                    }
                }
            } else {
                // unaligned joint offset (text stream behind an odd-sized image stream): element-wise
                const int s = seq0 + lane;
                const int pos = (s & ~15) | perm16(s & 15);
                if (lane < valid) {
                    for (int d = 0; d < 128; ++d) {
                        const unsigned short e =
                            *(const unsigned short*)(E + lane * 256 + (((d >> 3) ^ (lane & 15)) << 4) + (((d & 7) * 2) ^ (lane & 8)));
                        ((unsigned short*)vt)[(size_t)d * S_pad + pos] = e;
                    }
                }
            }
        }
    } else {
        const int c = lane & 15;
        const int n = nw0 + c * 8;
        bf16* out = (bf16*)P.out;
        float g[8];
        if constexpr (EPI == EPI_GATE_RES) {
            const float gs = P.has_gate_scalar ? P.gate_scalar : 1.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = gs;
            if (P.gate != nullptr && n < N) {
                const bf16x8 gv = *(const bf16x8*)((const bf16*)P.gate + n);
#pragma unroll
                for (int j = 0; j < 8; ++j) g[j] = (float)gv[j];
            }
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = it * 4 + (lane >> 4);
            const int m = mw0 + row;
            if (m >= M || n >= N) continue;
            const bf16x8 v = unswap(*(const bf16x8*)(E + row * 256 + ((c ^ (row & 15)) << 4)), it << 2);   // row & 8 == (it << 2) & 8
            bf16x8 o;
            if constexpr (EPI == EPI_BIAS) {
                o = v;
            } else if constexpr (EPI == EPI_GELU_SIG) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float y = (float)v[j];
                    const float t = bf16r(1.702f * y);
                    const float sg = bf16r(__builtin_amdgcn_rcpf(1.0f + __expf(-t)));   // v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE divide
                    o[j] = (bf16)(y * sg);
                }
            } else if constexpr (EPI == EPI_GELU_ERF) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float y = (float)v[j];
                    o[j] = (bf16)(0.5f * y * (1.0f + erff(y * 0.70710678118654752440f)));
                }
            } else if constexpr (EPI == EPI_SILU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float y = (float)v[j];
                    o[j] = (bf16)(y * __builtin_amdgcn_rcpf(1.0f + __expf(-y)));
                }
            } else if constexpr (EPI == EPI_GATE_RES) {
                const bf16x8 rv = rv16[it];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)rv[j] + bf16r(g[j] * (float)v[j]));
            }
            *(bf16x8*)(out + (size_t)m * P.ldo + n) = o;
        }
    }
    PE_STAMP(5);
    if constexpr (VAR == 14) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PE_STAMP(6);
        if (args.dbg != nullptr && threadIdx.x == 0) args.dbg[(size_t)blockIdx.x * 8 + 7] = (long long)__builtin_amdgcn_s_memrealtime();
    }
}

template <int EPI, int VAR, bool FP8>
__global__ void __launch_bounds__(GEMM_THREADS, 2) gemm_bf16_kernel(const GemmArgs args) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nk = args.p[0].K * (FP8 ? 1 : 2) / 128;
    // whole tile: one "range" [0, nk) of tile xcd_remap(block).  Stream-K tail (VAR 15): block b of the tail phase runs on
    // XCD b % 8; every XCD gets 32 CONSECUTIVE K ranges so that a tile's partials are (mostly) exchanged through one L2.
    int c = -1, tile = 0, k0 = 0, n_seg = 1;
    long long r1 = nk;
    if (VAR == 16 && args.sk_tiles > 0 && (int)blockIdx.x >= args.n_full) {
        const int b = (int)blockIdx.x - args.n_full;
        c = (b & 7) * (SK_WGS / 8) + (b >> 3);
        const long long total = (long long)args.sk_tiles * nk;
        const long long r0 = (long long)c * total / SK_WGS;
        r1 = (long long)(c + 1) * total / SK_WGS;
        const int t0 = (int)(r0 / nk);
        k0 = (int)(r0 - (long long)t0 * nk);
        r1 -= (long long)t0 * nk;                 // relative to the first tile's K tile 0
        n_seg = r1 > nk ? 2 : 1;
        tile = args.n_full + t0;
    } else {
        tile = xcd_remap((int)blockIdx.x, (VAR == 16 && args.sk_tiles > 0) ? args.n_full : (int)gridDim.x);
    }
    for (int seg = 0; seg < n_seg; ++seg) {
        const int lo = seg == 0 ? k0 : 0;
        const int hi = (int)min((long long)nk, r1 - (long long)seg * nk);
        gemm_tile<EPI, VAR, FP8>(args, smem, tile + seg, lo, hi, c);
    }
}

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
int g_gemm_variant = env_int("PE_GEMM_VARIANT", GEMM_DEFAULT_VARIANT);

template <int EPI, int VAR, bool FP8 = false>
static int launch_v(const GemmArgs& args, int ntiles, hipStream_t stream) {
    static std::atomic<bool> configured{false};   // racing first calls both configure: idempotent
    constexpr int lds = (FP8 || VAR >= 10) ? GEMM_LDS_V10 : GEMM_LDS;
    if (!configured.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<EPI, VAR, FP8>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return set_error(PE_ERR_HIP, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
        configured.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((gemm_bf16_kernel<EPI, VAR, FP8>), dim3(ntiles), dim3(GEMM_THREADS), lds, stream, args);
    return check_launch(FP8 ? "gemm_fp8_kernel" : "gemm_bf16_kernel");
}

template <int EPI>
static int launch_t(const GemmArgs& args, int ntiles, bool fp8, hipStream_t stream) {
    // instantiated schedules: 15 (default), 10 (the round-1 schedule, kept as the A/B reference), 14 (the 4-phase ping-pong with
    // s_memtime stamps, profiling only), 16 (15 + stream-K tail, a measured negative result kept as a knob).  The other round-1 / 2
    // experiments (0, 8, 12, 13) are described in profiles/r01_gemm_ablation.md and profiles/r02_gemm_notes.md and were removed.
    if (fp8) return g_gemm_variant == 10 ? launch_v<EPI, 10, true>(args, ntiles, stream) : launch_v<EPI, 15, true>(args, ntiles, stream);
    if (g_gemm_variant == 14) return launch_v<EPI, 14>(args, ntiles, stream);
    if (g_gemm_variant == 16) return launch_v<EPI, 16>(args, ntiles, stream);
    if (g_gemm_variant == 10) return launch_v<EPI, 10>(args, ntiles, stream);
    return launch_v<EPI, GEMM_DEFAULT_VARIANT>(args, ntiles, stream);
}

int g_gemm_streamk = env_int("PE_GEMM_STREAMK", 1);
void* g_gemm_sk_ws = nullptr;     // tests / granular operators: a registered stream-K workspace (pe_debug_set_ptr)
static std::atomic<unsigned> g_sk_epoch{0};

size_t gemm_streamk_ws_bytes() { return (size_t)SK_WGS * SK_SLOT_FLOATS * sizeof(float) + 4096; }

int launch_gemm(int epilogue, GemmProblem* problems, int nproblems, hipStream_t stream, void* sk_ws) {
    PE_REQUIRE(nproblems >= 1 && nproblems <= 2, "gemm: 1 or 2 problems per launch, got %d", nproblems);
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
        if (epilogue == EPI_QKV) {
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
    args.dbg = g_gemm_dbg;
    int ntiles = tiles[0] + tiles[1];
    // stream-K tail: when the last round of work-groups would fill only part of the chip, its tiles are cut along K into
    // SK_WGS equal ranges instead (gemm_bf16_kernel).  Needs a workspace for the fp32 partials, the default schedule and
    // one common K; not worth it for a nearly full last round or for very short K.
    args.n_full = ntiles; args.sk_tiles = 0; args.sk_ws = nullptr; args.sk_flags = nullptr; args.sk_status = nullptr; args.sk_epoch = 0;
    if (sk_ws == nullptr) sk_ws = g_gemm_sk_ws;
    {
        const int nk = problems[0].K / (fp8 ? 128 : BK);
        const int tail = ntiles % SK_WGS;
        const bool same_k = nproblems == 1 || problems[0].K == problems[1].K;
        if (g_gemm_streamk && sk_ws != nullptr && g_gemm_variant == 16 && !fp8 && same_k && tail >= 32 && tail <= 224 &&
            (long long)tail * nk / SK_WGS >= 6) {
            args.n_full = ntiles - tail;
            args.sk_tiles = tail;
            args.sk_ws = (float*)sk_ws;
            args.sk_flags = (unsigned*)((char*)sk_ws + (size_t)SK_WGS * SK_SLOT_FLOATS * sizeof(float));
            args.sk_status = args.sk_flags + SK_WGS;
            unsigned e = ++g_sk_epoch;
            if (e == 0) e = ++g_sk_epoch;      // 0 is what a freshly zeroed flag holds
            args.sk_epoch = e;
            ntiles = args.n_full + SK_WGS;     // grid: whole tiles, then one tail block per CU
        }
    }
    double flops = 0.0;  // algorithmic 2*M*N*K of the launch (what the roofline fraction is quoted on)
    for (int i = 0; i < nproblems; ++i) flops += 2.0 * problems[i].M * (double)problems[i].N * problems[i].K;
    const int slot = prof_begin(PROF_GEMM, flops, stream);
    int rc;
    switch (epilogue) {
        case EPI_BIAS: rc = launch_t<EPI_BIAS>(args, ntiles, fp8, stream); break;
        case EPI_GELU_SIG: rc = launch_t<EPI_GELU_SIG>(args, ntiles, fp8, stream); break;
        case EPI_GELU_ERF: rc = launch_t<EPI_GELU_ERF>(args, ntiles, fp8, stream); break;
        case EPI_GATE_RES: rc = launch_t<EPI_GATE_RES>(args, ntiles, fp8, stream); break;
        case EPI_QKV: rc = launch_t<EPI_QKV>(args, ntiles, fp8, stream); break;
        case EPI_SILU: rc = launch_t<EPI_SILU>(args, ntiles, fp8, stream); break;
        default: rc = set_error(PE_ERR_INVALID_ARG, "gemm: unknown epilogue %d", epilogue);
    }
    prof_end(slot, stream);
    return rc;
}

}  // namespace pe
