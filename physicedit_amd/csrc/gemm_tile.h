// Shared pieces of the bf16 / e4m3 GEMM kernels (gemm.hip: the 8-wave schedules, gemm4.hip: the 4-wave schedule 22): launch arguments,
// tile order, and the fused epilogue of one 64 x 128 wave block.  Device code only; included by those two translation units.
#pragma once
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace pe {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int GEMM_THREADS = 512;
constexpr int GEMM_LDS = 5 * 32768;   // A x2 + W x3 = 160 KiB (all of a CU's LDS)
constexpr int GEMM_DEFAULT_BAND = 4;      // M tiles per band: 8 until round 4; in the two-stream pipeline 4 is 0.45 - 0.6 % faster per image (profiles/r04_gemm_notes.md section 5)

struct GemmArgs {
    GemmProblem p[2];
    int tiles0;       // tiles of problem 0 (tile ids >= tiles0 belong to problem 1)
    int ntiles;       // all tiles of the launch (schedule 17: the grid is smaller)
    int band;         // M tiles per band of the tile order
    int direct_epi;   // 1: complete tiles of the element-wise epilogues skip the LDS round trip (A/B knob "gemm_direct_epilogue", default 1)
    int skip_ragged;  // 1: row blocks beyond M skip their MFMAs (A/B knob "gemm_skip_ragged", default 1)
    int defer;        // 1: schedule 21 parks y in `stash` at a tile's end and runs the epilogue proper inside the next tile's main loop (knob "gemm_defer_epilogue")
    char* stash;      // DEFER_STASH_BYTES per work-group of the persistent grid, or null (no deferral)
    int cont;         // 1: schedule 21 keeps the two wave groups one slot apart across tile boundaries (A/B knob "gemm_continuous"; gemm.hip)
    int no_epi;       // TIMING ONLY (knob "gemm_no_epilogue"): the 8-wave schedules stop a tile behind its main loop, nothing is stored -- prices the epilogue
    long long* dbg;   // per work-group s_memtime stamps [grid][8] of schedule 15 (pe_debug_set_ptr("gemm_stamps", p)), or null
    unsigned* sk_sync;   // schedule 19: [0] arrivals, [1 + c] position counter of chunk c, [SK_FLAG0 + pos] flag of the seam behind position pos; zero at rest
    char* sk_part;       // schedule 19: fp32 accumulator images, SK_PART_BYTES per seam
};
constexpr int SK_FLAG0 = 32;                          // flags start on their own 128-byte line
constexpr size_t SK_PART_BYTES = (size_t)BM * BN * 4;   // one 256 x 256 fp32 accumulator tile
extern long long* g_gemm_dbg;
struct GemmArgs;
int launch_gemm4(int epilogue, bool fp8, const GemmArgs& args, int grid, hipStream_t stream);      // gemm4.hip
// Device code reads the launch arguments where the dispatcher put them: the kernarg segment (constant address space, scalar
// loads at a wave-uniform problem index).  Through a by-value copy `args.p[pi]` with a run-time pi is an indexed private array:
// the compiler parks fields in scratch.
#define KARG __attribute__((address_space(4)))

#define PE_STAMP(k)                                                                                     \
    do {                                                                                                \
        if constexpr (VAR == 15) {                                                                      \
            if (args.dbg != nullptr && threadIdx.x == 0) args.dbg[(size_t)blockIdx.x * 8 + (k)] = (long long)__builtin_readcyclecounter(); \
        }                                                                                               \
    } while (0)

PE_DEV int perm16(int j) { return (j & 3) | ((j & 4) << 1) | ((j & 8) >> 1); }

// VAR selects the main-loop schedule (A/B-tested in one process through pe_debug_set("gemm_variant"); the studies,
// incl. the variants that no longer live here, are profiles/r01_gemm_ablation.md, r02_gemm_notes.md, r03_gemm_notes.md):
//   15  (round-2 default) "ping-pong": the two wave groups run the same stream one barrier apart, 2 phases per K tile, A half
//       tiles staged by the group that reads them; one tile per work-group (what the persistent schedules fall back to for
//       launches of at most one round of tiles)
//   17  (round 3 / 4 default) 15's main loop in persistent work-groups with cross-tile prefetch (gemm_persistent in gemm.hip)
//   19  17 as stream-K (opt-in, measured slower)
//   21  (DEFAULT since round 5, GEMM_DEFAULT_VARIANT) 17 with ONE matrix-pipe hand-off per K tile: a LOAD slot and a 64-MFMA slot
//   22  four waves x 128 x 128, one per SIMD (gemm4.hip; opt-in)
//   (10, the round-1 schedule, left the library in round 5)
//
// FP8 = true: operands are OCP e4m3 bytes (activation rows quantised by quantize_rows_e4m3, weights stored in e4m3),
// the K tile is 128 elements (the SAME 128-B LDS rows, staging and swizzle), the MFMA is the CDNA4 block-scaled
// v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (2x the bf16 rate), and the epilogue starts with
// y = bf16(acc * scale_a[m] + bias[n])  (AutoWrappedLinear.fp8_linear, vram_management/layers.py:115-151).
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

struct TileCoord {
    int pi, m0, n0;
};
// tile id in the banded order -> problem and tile origin.  Band = `band` M tiles x all N tiles, N-major inside a band, so a
// run of consecutive ids (what one XCD's 32 CUs work on at a time) covers band x (32 / band) tiles.
PE_DEV TileCoord decode_tile(const KARG GemmArgs& args, int bid) {
    TileCoord c;
    c.pi = bid >= args.tiles0 ? 1 : 0;
    const KARG GemmProblem& P = args.p[c.pi];
    bid -= c.pi ? args.tiles0 : 0;
    const int tilesM = P.tilesM, tilesN = P.tilesN;
    const int per_band = args.band * tilesN;
    const int band = bid / per_band;
    const int rem = bid - band * per_band;
    const int gm = min(args.band, tilesM - band * args.band);
    const int tn = rem / gm;
    const int tm = band * args.band + (rem - tn * gm);
    c.m0 = tm * BM;
    c.n0 = tn * BN;
    return c;
}

// ------------------------------------------------------------------------------------------
// epilogue of one 256x256 tile.  acc[mi][ni][4q+r] = C[m0 + wm*64 + mi*32 + l31][n0 + wn*128 + ni*32 + 8q + 4h + r]
// The wave's 64 x 128 bf16 block goes through LDS as two 32-row halves (mi = 0, 1) of 8 KiB at E0 / E1: 16-B chunk c of row
// r sits at chunk c ^ (r & 15), and its two 8-B halves are swapped when r & 8 (rows r and r ^ 8 would otherwise land on the
// same banks in one ds_write_b64 lane group).  A wave reads back only what it wrote, and one wave's LDS accesses execute in
// order: no barrier.  TWO_PASS (schedule 17, E0 == E1): stage half 0, emit it, stage half 1, emit it.
//
// FAST (the tile lies inside M x N and the problem has no hot-LoRA `pre` operand: every tile of the DiT's Linears but the ragged
// text-stream one): no bounds checks, no per-chunk branches.  The general form costs ~100 exec-mask branches and their scalar
// chains per wave and tile -- the epilogue is VALU / issue bound (profiles/r03_gemm_notes.md section 6).
// ------------------------------------------------------------------------------------------
PE_DEV f32x2 up2(uint32_t p) { return f32x2{__uint_as_float(p << 16), __uint_as_float(p & 0xffff0000u)}; }   // two packed bf16 -> fp32
PE_DEV uint32_t pk2(f32x2 v) {                                                                                // v_cvt_pk_bf16_f32
    const bf16x2 b = {(bf16)v.x, (bf16)v.y};
    return __builtin_bit_cast(uint32_t, b);
}
PE_DEV f32x2 rnd2(f32x2 v) { return up2(pk2(v)); }      // bf16r of a pair: one convert for two values
// sum over the 16 lanes of a DPP row, result in every lane; the xor-butterfly order (1, 2, 4, 8) with DPP operands instead of
// ds_bpermute: after the two quad steps a quad's lanes agree, so the half-mirror (lane 7 - l) and mirror (15 - l) partners hold
// what lanes l ^ 4 and l ^ 8 hold -- same bits as the __shfl_xor form, no LDS round trips, no lane-id register
PE_DEV float row16_sum(float v) {
    const auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror
    return v;
}

// LAY: the accumulator layout.  0: 2 x 4 blocks of v_mfma 32 x 32 (e4m3, and the four-wave schedule): acc[mi][ni][4 q + r] as above.
// 1 (bf16 since round 5): 4 x 8 blocks of v_mfma_f32_16x16x32_bf16: acc[mb][nb][r] = C[m0 + wm*64 + mb*16 + (lane & 15)][n0 + wn*128 + nb*16 +
// 4 (lane >> 4) + r].  Both are 16 chunks of 4 columns per lane and 32-row half (k = 4 ni + q, resp. 8 (mb & 1) + nb); only where a chunk
// comes from and where it lands in the staged image differs.
template <int EPI, bool FP8, bool TWO_PASS, bool FAST, int NMI = 2, int MI0 = 0, int LAY = 0, typename ACC>
__device__ __forceinline__ void gemm_epilogue_body(const KARG GemmProblem& P, const int M, const int N, const ACC& acc, int m0, int n0,
                                                   char* E0, char* E1, int lane, int w, long long* stamp4) {
    // NMI / MI0: the accumulator array holds NMI 32-row blocks, this call emits blocks MI0, MI0 + 1 as the 64 x 128 block of (virtual)
    // wave `w` (the 4-wave kernel's 128 x 128 wave tile is two such calls)
    static_assert(LAY == 0 || (NMI == 2 && MI0 == 0), "layout 1 is the 8-wave kernels'");
    constexpr bool QKV = EPI == EPI_QKV || EPI == EPI_QKV_STATS;
    constexpr bool STATS = EPI == EPI_QKV_STATS;      // also: sum and sum of squares of the bf16 values this wave stores (one head of one section)
    float st1 = 0.f, st2 = 0.f;
    auto stat2 = [&](uint32_t pk) __attribute__((always_inline)) {      // two packed bf16 outputs
        const float a = __uint_as_float(pk << 16), b = __uint_as_float(pk & 0xffff0000u);
        st1 += a; st1 += b;
        st2 = __builtin_fmaf(a, a, st2); st2 = __builtin_fmaf(b, b, st2);
    };
    const int l31 = lane & 31, h = lane >> 5;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    auto unswap = [](bf16x8 v, int row) -> bf16x8 {
        return (row & 8) ? __builtin_shufflevector(v, v, 4, 5, 6, 7, 0, 1, 2, 3) : v;
    };
    const int nw0 = n0 + wn * 128;
    const int mw0 = m0 + wm * 64;
    bf16x8 rv16[EPI == EPI_GATE_RES ? 16 : 1];   // residual rows of the gated-residual epilogue, prefetched
    const bf16* bias = (const bf16*)P.bias;
    const bf16* pre = (const bf16*)P.pre;   // hot LoRA: `out + x @ A.T @ B.T` (vram_management/layers.py:179-180)
    float sa[LAY == 1 ? 4 : 2] = {1.f, 1.f};      // FP8: per-row activation scale (fp8_linear's scale_a) of the lane's rows (layout 0: 2, layout 1: 4)
    if constexpr (FP8 && LAY == 1) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) sa[mb] = P.scale_a[FAST ? mw0 + mb * 16 + l15 : min(mw0 + mb * 16 + l15, M - 1)];
    } else if constexpr (FP8) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) sa[mi] = P.scale_a[FAST ? mw0 + mi * 32 + l31 : min(mw0 + mi * 32 + l31, M - 1)];
    }
    // bias first, then (EPI_GATE_RES) all 16 residual rows of this lane: 16-B loads that stay in flight under the LDS
    // staging below (issued 4 at a time inside the store loop they cost 15-21k cycles of exposed latency)
    bf16x4 bvs[16];
    if constexpr (LAY == 1) {
        // the lane's 4 columns of each of the 8 column blocks
#pragma unroll
        for (int i = 0; i < 16; ++i) bvs[i] = bf16x4{0, 0, 0, 0};
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
            const int n = nw0 + nb * 16 + 4 * g4;
            if (bias != nullptr && (FAST || n < N)) bvs[nb] = *(const bf16x4*)(bias + n);
        }
    } else if constexpr (FAST) {
#pragma unroll
        for (int i = 0; i < 16; ++i) bvs[i] = bf16x4{0, 0, 0, 0};
        if (bias != nullptr) {      // one wave-uniform branch for the 16 loads
            const bf16* bl = bias + nw0 + 4 * h;
#pragma unroll
            for (int i = 0; i < 16; ++i) bvs[i] = *(const bf16x4*)(bl + (i >> 2) * 32 + 8 * (i & 3));
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int n = nw0 + (i >> 2) * 32 + 8 * (i & 3) + 4 * h;
            bvs[i] = bf16x4{0, 0, 0, 0};
            if (bias != nullptr && n < N) bvs[i] = *(const bf16x4*)(bias + n);
        }
    }
    if constexpr (EPI == EPI_GATE_RES) {
        const int n = nw0 + (lane & 15) * 8;
        if constexpr (FAST) {
            const bf16* rl = (const bf16*)P.res + (size_t)(mw0 + (lane >> 4)) * P.ldr + n;
#pragma unroll
            for (int it = 0; it < 16; ++it) rv16[it] = *(const bf16x8*)(rl + (size_t)(it * 4) * P.ldr);
        } else {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int m = mw0 + it * 4 + (lane >> 4);
                rv16[it] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                if (m < M && n < N) rv16[it] = *(const bf16x8*)((const bf16*)P.res + (size_t)m * P.ldr + n);
            }
        }
    }
    // gate vector / q-k norm weight of this lane's 8 columns: loaded BEFORE any store (vmcnt retires in order: a load issued
    // behind the first half's stores could only be consumed after they have drained)
    bf16x8 gate_v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (EPI == EPI_GATE_RES) {
        const int n = nw0 + (lane & 15) * 8;
        if (P.gate != nullptr && (FAST || n < N)) gate_v = *(const bf16x8*)((const bf16*)P.gate + n);
    }
    // y = bf16(acc + bias) (+ pre) of chunk k (0 .. 15) of row half mi: the lane's four columns 8q + 4h .. + 3 of block ni (k = 4 ni + q),
    // resp. 4 g4 .. + 3 of column block nb in row block 2 mi + mb (k = 8 mb + nb)
    auto y_chunk = [&](int mi, int k) __attribute__((always_inline)) -> bf16x4 {
        int n, row;
        float b[4];
        bf16x4 y;
        if constexpr (LAY == 1) {
            const int mb = k >> 3, nb = k & 7;
            n = nw0 + nb * 16 + 4 * g4;
            row = mi * 32 + mb * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) b[r] = (float)bvs[nb][r];
            if constexpr (FP8) {
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = (bf16)(acc[mi * 2 + mb][nb][r] * sa[mi * 2 + mb] + b[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = (bf16)(acc[mi * 2 + mb][nb][r] + b[r]);
            }
        } else {
            const int ni = k >> 2, q = k & 3;
            n = nw0 + ni * 32 + 8 * q + 4 * h;
            row = mi * 32 + l31;
#pragma unroll
            for (int r = 0; r < 4; ++r) b[r] = (float)bvs[k][r];
            if constexpr (FP8) {
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = (bf16)(acc[MI0 + mi][ni][4 * q + r] * sa[mi] + b[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = (bf16)(acc[MI0 + mi][ni][4 * q + r] + b[r]);
            }
        }
        if (!FAST && pre != nullptr && n < N && mw0 + row < M) {
            // y = pre + y : the linear's own (already rounded) output plus this low-rank product
            const bf16x4 pv = *(const bf16x4*)(pre + (size_t)(mw0 + row) * P.ldp + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = (bf16)((float)pv[r] + (float)y[r]);
        }
        return y;
    };
    // PACK_FIRST (schedule 17's two-pass form on full tiles): both halves are rounded and packed before anything is staged --
    // 64 registers of bf16 pairs instead of 128 accumulators + 32 of bias live while the first half is emitted, which is what
    // lets the second half's operands (RoPE tables, residual rows) be requested BEFORE the first half's stores
    constexpr bool PACK_FIRST = TWO_PASS && FAST;
    bf16x4 ypk[PACK_FIRST ? 2 : 1][PACK_FIRST ? 16 : 1];
    if constexpr (PACK_FIRST) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) ypk[mi][k] = y_chunk(mi, k);
    }
    // the row halves [mi_lo, mi_hi) -> LDS: chunk k of local row lrow = columns 8 c + 4 hs .. + 3 of the staged image
    auto stage_rows = [&](int mi_lo, int mi_hi) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                if (mi < mi_lo || mi >= mi_hi) continue;
                bf16x4 y;
                if constexpr (PACK_FIRST) y = ypk[mi][k];
                else y = y_chunk(mi, k);
                char* Eb = mi == 0 ? E0 : E1;
                const int lrow = LAY == 1 ? (k >> 3) * 16 + l15 : l31;
                const int c = LAY == 1 ? (k & 7) * 2 + (g4 >> 1) : k;
                const int hs = LAY == 1 ? (g4 & 1) : h;
                *(bf16x4*)(Eb + lrow * 256 + ((c ^ (lrow & 15)) << 4) + ((hs ^ ((lrow >> 3) & 1)) << 3)) = y;
            }
        }
    };

    // the lane's 8 rows of the staged half `mi` (one 16-B chunk = 8 consecutive columns of each)
    auto load_rows = [&](int mi, bf16x8 (&rows)[8]) __attribute__((always_inline)) {
        const char* Eb = mi == 0 ? E0 : E1;
        const int c = lane & 15;
#pragma unroll
        for (int j8 = 0; j8 < 8; ++j8) {
            const int lrow = j8 * 4 + (lane >> 4);
            rows[j8] = unswap(*(const bf16x8*)(Eb + lrow * 256 + ((c ^ (lrow & 15)) << 4)), j8 << 2);   // lrow & 8 == (j8 << 2) & 8
        }
    };
    // read the staged half `mi` back row-wise (held_tag false) or take the rows load_rows fetched earlier (true), apply the
    // epilogue proper, store
    auto emit_rows = [&](int mi, auto held_tag, const bf16x8 (&held)[8]) __attribute__((always_inline)) {
        constexpr bool HELD = decltype(held_tag)::value;
        const char* Eb = mi == 0 ? E0 : E1;
        if constexpr (QKV) {
            const int HD = N / 3;
            const int section = nw0 / HD;  // wave-uniform: the wave's 128 columns are exactly one head
            const int head = (nw0 - section * HD) >> 7;
            const int S_pad = P.S_pad;
            if (section < 2) {
                const bf16* nw = (const bf16*)(section == 0 ? P.norm_q_w : P.norm_k_w);
                bf16* dst = (bf16*)(section == 0 ? P.q_out : P.k_out);
                const int c = lane & 15;
                const bf16x8 wv = *(const bf16x8*)(nw + c * 8);
                // Q may carry the attention's scale . log2(e) (attention variants 5 / 6): multiplied in fp32 before the one rounding
                const float qs1 = (section == 0 && P.q_scale != 0.f) ? P.q_scale : 1.0f;
                const f32x2 qs = f32x2{qs1, qs1};
                // RoPE operands of the lane's 8 rows, loaded UNCONDITIONALLY (row clamped) and before the first store of this half:
                // vmcnt retires in order, so a load issued behind a store is only usable once that store has drained -- with the
                // loads inside the per-row `if (m < M)` every group of rows paid a store round trip (s_memtime stamps: 17k ticks
                // for this epilogue in steady state against 9.5k for the GELU one)
                f32x4 cs8[8], sn8[8];
#pragma unroll
                for (int j8 = 0; j8 < 8; ++j8) {
                    const int mr = mw0 + mi * 32 + j8 * 4 + (lane >> 4);
                    const int mc = FAST ? mr : min(mr, M - 1);
                    cs8[j8] = *(const f32x4*)(P.rope_cos + (size_t)mc * 64 + c * 4);
                    sn8[j8] = *(const f32x4*)(P.rope_sin + (size_t)mc * 64 + c * 4);
                }
#pragma unroll
                for (int j8 = 0; j8 < 8; ++j8) {
                    const int lrow = j8 * 4 + (lane >> 4);
                    const int m = mw0 + mi * 32 + lrow;
                    bf16x8 v;
                    if constexpr (HELD) v = held[j8];
                    else v = unswap(*(const bf16x8*)(Eb + lrow * 256 + ((c ^ (lrow & 15)) << 4)), j8 << 2);   // lrow & 8 == (j8 << 2) & 8
                    // pairs of columns as f32x2 (packed fp32 maths, one bf16 convert per pair); the operations and their order are
                    // those of the scalar form this replaces: squares summed left to right, every op rounded where the reference rounds
                    const u32x4 vp = __builtin_bit_cast(u32x4, v);
                    const u32x4 wp = __builtin_bit_cast(u32x4, wv);
                    f32x2 y2[4];
                    float ss = 0.f;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        y2[jj] = up2(vp[jj]);
                        const f32x2 sq = y2[jj] * y2[jj];
                        ss += sq.x;
                        ss += sq.y;
                    }
                    ss = row16_sum(ss);
                    // RMSNorm(128, eps 1e-6): models/utils.py:250-257
                    const float rs = __builtin_amdgcn_rsqf(ss * (1.0f / 128.0f) + 1e-6f);   // v_rsq_f32: the argument is a normal number >= 1e-6
                    if (FAST || m < M) {
                        const f32x4 cs = cs8[j8], sn = sn8[j8];
                        u32x4 o;
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const f32x2 x = rnd2(rnd2(y2[jj] * f32x2{rs, rs}) * up2(wp[jj]));
                            // apply_rotary_emb_qwen: fp32 complex multiply (qwen_image_dit.py:51-57)
                            const f32x2 a = x * f32x2{cs[jj], cs[jj]};
                            const f32x2 b = f32x2{x.y, x.x} * f32x2{sn[jj], sn[jj]};
                            o[jj] = pk2(f32x2{a.x - b.x, a.y + b.y} * qs);
                        }
                        *(u32x4*)(dst + ((size_t)head * S_pad + P.seq_off + m) * 128 + c * 8) = o;
                        if constexpr (STATS) {
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) stat2(o[jj]);
                        }
                    }
                }
            } else {
                // V: written TRANSPOSED, Vt[head][d][pos(token)], tokens permuted inside aligned 16-groups
                // (pos = perm16) so the attention kernel's P.V MFMA needs no cross-lane shuffle.
                bf16* vt = (bf16*)P.vt_out + (size_t)head * 128 * S_pad;
                const int seq0 = P.seq_off + mw0;
                const int valid = FAST ? 64 : min(64, M - mw0);
                if ((seq0 & 15) == 0) {
                    // this half holds the 16-token groups 2 mi and 2 mi + 1 of the wave's 64 tokens: 128 d x 2 groups x 2 halves
#pragma unroll 2
                    for (int it = 0; it < 8; ++it) {
                        const int id = it * 64 + lane;
                        const int d = id >> 2, tg = id & 3;
                        const int gl = tg >> 1, hh = tg & 1;       // group inside this half, 8-token half of the group
                        const int gi = mi * 2 + gl;
                        unsigned short e[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int lrow = gl * 16 + (j < 4 ? 4 * hh + j : 8 + 4 * hh + (j - 4));
                            e[j] = *(const unsigned short*)(Eb + lrow * 256 + (((d >> 3) ^ (lrow & 15)) << 4) + ((((d & 7) * 2) ^ (lrow & 8))));
                        }
                        bf16* dstp = vt + (size_t)d * S_pad + seq0 + gi * 16 + hh * 8;
                        if (FAST || gi * 16 + 16 <= valid) {
                            u32x4 pk;
#pragma unroll
                            for (int j = 0; j < 4; ++j) pk[j] = (uint32_t)e[2 * j] | ((uint32_t)e[2 * j + 1] << 16);
                            *(u32x4*)dstp = pk;
                            if constexpr (STATS) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) stat2(pk[j]);
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const int tok = gi * 16 + (j < 4 ? 4 * hh + j : 8 + 4 * hh + (j - 4));
                                if (tok < valid) {
                                    ((unsigned short*)dstp)[j] = e[j];
                                    if constexpr (STATS) stat2((uint32_t)e[j] << 16);      // (the low half is +0: adds nothing)
                                }
                            }
                        }
                    }
                } else {
                    // unaligned joint offset (text stream behind an odd-sized image stream): element-wise, lane = token
                    const int s = seq0 + lane;
                    const int pos = (s & ~15) | perm16(s & 15);
                    const int lrow = lane & 31;
                    if (lane < valid && (lane >> 5) == mi) {
                        for (int d = 0; d < 128; ++d) {
                            const unsigned short e =
                                *(const unsigned short*)(Eb + lrow * 256 + (((d >> 3) ^ (lrow & 15)) << 4) + (((d & 7) * 2) ^ (lrow & 8)));
                            ((unsigned short*)vt)[(size_t)d * S_pad + pos] = e;
                            if constexpr (STATS) stat2((uint32_t)e << 16);
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
#pragma unroll
                    for (int j = 0; j < 8; ++j) g[j] = (float)gate_v[j];
                }
            }
#pragma unroll
            for (int j8 = 0; j8 < 8; ++j8) {
                const int lrow = j8 * 4 + (lane >> 4);
                const int m = mw0 + mi * 32 + lrow;
                if (!FAST && (m >= M || n >= N)) continue;
                bf16x8 v;
                if constexpr (HELD) v = held[j8];
                else v = unswap(*(const bf16x8*)(Eb + lrow * 256 + ((c ^ (lrow & 15)) << 4)), j8 << 2);   // lrow & 8 == (j8 << 2) & 8
                bf16x8 o;
                if constexpr (EPI == EPI_BIAS) {
                    o = v;
                } else if constexpr (EPI == EPI_GELU_SIG) {
                    // x * sigmoid(1.702 x) with the reference's three bf16 roundings (qwen_image_dit.py:44-49), two columns at a time:
                    // packed fp32 multiplies / adds, one bf16 convert per pair and rounding; exp(-t) = v_exp_f32(-t log2 e) as
                    // __expf evaluates it, v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE divide
                    const u32x4 vp = __builtin_bit_cast(u32x4, v);
                    u32x4 op;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const f32x2 y = up2(vp[jj]);
                        const f32x2 t = rnd2(f32x2{1.702f, 1.702f} * y);
                        const f32x2 a = t * f32x2{-1.4426950408889634f, -1.4426950408889634f};
                        const f32x2 d = f32x2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)} + f32x2{1.0f, 1.0f};
                        const f32x2 sg = rnd2(f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)});
                        op[jj] = pk2(y * sg);
                    }
                    o = __builtin_bit_cast(bf16x8, op);
                    if constexpr (FP8) {
                        if (P.q8_out != nullptr) {
                            // the next Linear's e4m3 operand (fp8_linear's row quantisation with scale 1, GemmProblem.q8_out)
                            float f[8];
                            float amax = 0.f;
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                f[j] = (float)o[j];
                                amax = fmaxf(amax, fabsf(f[j]));
                            }
                            u32x2 pk;
                            pk[0] = pack4_e4m3(f[0], f[1], f[2], f[3]);
                            pk[1] = pack4_e4m3(f[4], f[5], f[6], f[7]);
                            *(u32x2*)((uint8_t*)P.q8_out + (size_t)m * P.ldq8 + n) = pk;
                            if (amax > 447.0f) atomicOr(P.q8_flags + m, 1u);      // rare: the row needs a scale above 1
                        }
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
                    const u32x4 rp = __builtin_bit_cast(u32x4, rv16[mi * 8 + j8]);
                    const u32x4 vp = __builtin_bit_cast(u32x4, v);
                    u32x4 op;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) op[jj] = pk2(up2(rp[jj]) + rnd2(f32x2{g[2 * jj], g[2 * jj + 1]} * up2(vp[jj])));
                    o = __builtin_bit_cast(bf16x8, op);
                }
                *(bf16x8*)(out + (size_t)m * P.ldo + n) = o;
            }
        }
    };

    bf16x8 rows[8];
    if constexpr (TWO_PASS) {
        // One 8 KiB region for both halves.  A wave's LDS operations execute in order, so half 1 may be staged as soon as the
        // READS of half 0 have been issued: its 32 ds_write_b64 then run under the epilogue maths and stores of half 0 instead of
        // behind them.
        // Not for the gated-residual and QKV epilogues: with their 16 prefetched residual rows / RoPE operands the held rows
        // push the kernel over the 256-register budget (19-31 spilled registers measured).
        constexpr bool EARLY_STAGE = EPI != EPI_GATE_RES && !QKV;
        if constexpr (EARLY_STAGE) {
            stage_rows(0, 1);
            load_rows(0, rows);
            stage_rows(1, 2);
            emit_rows(0, std::true_type{}, rows);
            load_rows(1, rows);
            emit_rows(1, std::true_type{}, rows);
        } else {
            stage_rows(0, 1);
            emit_rows(0, std::false_type{}, rows);
            stage_rows(1, 2);
            emit_rows(1, std::false_type{}, rows);
        }
    } else {
        stage_rows(0, 2);
        if (stamp4 != nullptr && threadIdx.x == 0) *stamp4 = (long long)__builtin_readcyclecounter();
        emit_rows(0, std::false_type{}, rows);
        emit_rows(1, std::false_type{}, rows);
    }
    if constexpr (STATS) {
        // the wave's 64 rows x 128 columns are one 64-row block of one head of one section: one entry of the partial-sum table, whoever
        // computes the tile (fixed summation order per entry, fixed order of the entries in the finish kernel: deterministic)
        double d1 = (double)st1, d2 = (double)st2;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            d1 += __shfl_xor(d1, o, 64);
            d2 += __shfl_xor(d2, o, 64);
        }
        if (lane == 0 && P.qkv_stats != nullptr) {
            const int HD = N / 3, section = nw0 / HD, head = (nw0 - section * HD) >> 7, rb = (m0 >> 6) + wm;
            double* dst = P.qkv_stats + ((((size_t)section * 2 + P.stat_slot) * P.stat_rb + rb) * (HD >> 7) + head) * 2;
            dst[0] = d1;
            dst[1] = d2;
        }
    }
}

// ------------------------------------------------------------------------------------------
// The same epilogue WITHOUT the LDS round trip, for complete tiles of the element-wise epilogues (round 5; bias, GELU, gate x y + residual).
// The accumulator layout already has 4 consecutive columns of a row per lane and quad, and its partner lane (l ^ 32) holds the next 4:
// one v_permlane32_swap per dword on the quad pairs (q, q + 1) leaves 16 contiguous bytes of the row in every lane (lower lanes: quad
// q, upper lanes: quad q + 1), i.e. 16-byte residual loads and 16-byte stores straight from registers -- no staging writes, no read-back,
// no swizzle arithmetic (3.3k of the 4.8k - 11k cycles an LDS epilogue takes per tile: profiles/r03_gemm_notes.md section 5).  A store
// instruction covers 32 rows x 32 bytes instead of 4 rows x 256 bytes.  The arithmetic per output element is the LDS form's, rounding
// for rounding: bit-identical.  Knob "gemm_direct_epilogue" (default 1) for the A/B.
// ------------------------------------------------------------------------------------------
template <int EPI>
constexpr bool kDirectEpi = EPI == EPI_GELU_SIG || EPI == EPI_GATE_RES;      // (bias only: measured 1 % slower bf16, 4.6 % slower e4m3 than the LDS form)

// the epilogue proper of 8 consecutive columns of one row (bf16 pairs vp -> op), shared by both layouts of the direct form
template <int EPI, bool FP8>
__device__ __forceinline__ u32x4 direct_epi_math(const KARG GemmProblem& P, u32x4 vp, u32x4 rp, const f32x2 (&g2)[4], int m, int n) {
    u32x4 op;
    if constexpr (EPI == EPI_BIAS) {
        op = vp;
    } else if constexpr (EPI == EPI_GELU_SIG) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const f32x2 y = up2(vp[jj]);
            const f32x2 t = rnd2(f32x2{1.702f, 1.702f} * y);
            const f32x2 a = t * f32x2{-1.4426950408889634f, -1.4426950408889634f};
            const f32x2 d = f32x2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)} + f32x2{1.0f, 1.0f};
            const f32x2 sg = rnd2(f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)});
            op[jj] = pk2(y * sg);
        }
        if constexpr (FP8) {
            if (P.q8_out != nullptr) {
                const bf16x8 o = __builtin_bit_cast(bf16x8, op);
                float f[8];
                float amax = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    f[e] = (float)o[e];
                    amax = fmaxf(amax, fabsf(f[e]));
                }
                u32x2 pk;
                pk[0] = pack4_e4m3(f[0], f[1], f[2], f[3]);
                pk[1] = pack4_e4m3(f[4], f[5], f[6], f[7]);
                *(u32x2*)((uint8_t*)P.q8_out + (size_t)m * P.ldq8 + n) = pk;
                if (amax > 447.0f) atomicOr(P.q8_flags + m, 1u);
            }
        }
    } else {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) op[jj] = pk2(up2(rp[jj]) + rnd2(g2[jj] * up2(vp[jj])));
    }
    return op;
}

// Layout 1 (16 x 16 blocks): a row's 16 columns of a block sit in FOUR lanes (l, l + 16, l + 32, l + 48), 4 columns each.  One
// v_permlane16_swap per dword on the column-block pairs (nb, nb + 1) -- odd 16-lane rows of block nb <-> even rows of block nb + 1 -- leaves
// 8 consecutive columns in every lane: lane group g holds columns 8 (g >> 1) .. + 7 of block nb + (g & 1); a store instruction covers
// 16 rows x 64 contiguous bytes.
template <int EPI, bool FP8>
__device__ __forceinline__ void gemm_epilogue_direct16(const KARG GemmProblem& P, const f32x4 (&acc)[4][8], int m0, int n0, int lane, int w) {
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    const int nw0 = n0 + wn * 128, mw0 = m0 + wm * 64;
    const bf16* bias = (const bf16*)P.bias;
    float sa[4] = {1.f, 1.f, 1.f, 1.f};
    if constexpr (FP8) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) sa[mb] = P.scale_a[mw0 + mb * 16 + l15];
    }
    bf16x4 bvs[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) bvs[nb] = bf16x4{0, 0, 0, 0};
    if (bias != nullptr) {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) bvs[nb] = *(const bf16x4*)(bias + nw0 + nb * 16 + 4 * g4);
    }
    // the lane's store position inside a (row block mb, column-block pair p): row mb * 16 + l15, columns (2 p + (g4 & 1)) * 16 + 8 (g4 >> 1)
    const int ncol = (g4 & 1) * 16 + 8 * (g4 >> 1);
    // residual rows in the STORE layout; row blocks 0, 1 are requested here, 2, 3 once the accumulators are dead (register budget)
    u32x4 rv[EPI == EPI_GATE_RES ? 16 : 1];
    const bf16* rl = nullptr;
    if constexpr (EPI == EPI_GATE_RES) {
        rl = (const bf16*)P.res + (size_t)(mw0 + l15) * P.ldr + nw0 + ncol;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int p_ = 0; p_ < 4; ++p_) rv[mb * 4 + p_] = *(const u32x4*)(rl + (size_t)(mb * 16) * P.ldr + p_ * 32);
    }
    uint32_t yp[4][8][2];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
            const bf16x4 b4 = bvs[nb];
            if constexpr (FP8) {
                yp[mb][nb][0] = pk2(f32x2{acc[mb][nb][0] * sa[mb] + (float)b4[0], acc[mb][nb][1] * sa[mb] + (float)b4[1]});
                yp[mb][nb][1] = pk2(f32x2{acc[mb][nb][2] * sa[mb] + (float)b4[2], acc[mb][nb][3] * sa[mb] + (float)b4[3]});
            } else {
                yp[mb][nb][0] = pk2(f32x2{acc[mb][nb][0] + (float)b4[0], acc[mb][nb][1] + (float)b4[1]});
                yp[mb][nb][1] = pk2(f32x2{acc[mb][nb][2] + (float)b4[2], acc[mb][nb][3] + (float)b4[3]});
            }
        }
    if constexpr (EPI == EPI_GATE_RES) {
#pragma unroll
        for (int mb = 2; mb < 4; ++mb)
#pragma unroll
            for (int p_ = 0; p_ < 4; ++p_) rv[mb * 4 + p_] = *(const u32x4*)(rl + (size_t)(mb * 16) * P.ldr + p_ * 32);
    }
    bf16* out = (bf16*)P.out;
    const float gs = (EPI == EPI_GATE_RES && P.has_gate_scalar) ? P.gate_scalar : 1.0f;
    // the gate vector of the lane's 4 x 8 columns (same for every row block)
    u32x4 gv[EPI == EPI_GATE_RES ? 4 : 1];
    bool has_gate = false;
    if constexpr (EPI == EPI_GATE_RES) {
        has_gate = P.gate != nullptr;
        if (has_gate) {
#pragma unroll
            for (int p_ = 0; p_ < 4; ++p_) gv[p_] = *(const u32x4*)((const bf16*)P.gate + nw0 + p_ * 32 + ncol);
        }
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int p_ = 0; p_ < 4; ++p_) {
            const int n = nw0 + p_ * 32 + ncol;
            const int m = mw0 + mb * 16 + l15;
            const auto s0 = __builtin_amdgcn_permlane16_swap(yp[mb][2 * p_][0], yp[mb][2 * p_ + 1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(yp[mb][2 * p_][1], yp[mb][2 * p_ + 1][1], false, false);
            const u32x4 vp = {s0[0], s1[0], s0[1], s1[1]};
            f32x2 g2[4] = {f32x2{gs, gs}, f32x2{gs, gs}, f32x2{gs, gs}, f32x2{gs, gs}};
            u32x4 rp = {0, 0, 0, 0};
            if constexpr (EPI == EPI_GATE_RES) {
                if (has_gate) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) g2[jj] = up2(gv[p_][jj]);
                }
                rp = rv[mb * 4 + p_];
            }
            *(u32x4*)(out + (size_t)m * P.ldo + n) = direct_epi_math<EPI, FP8>(P, vp, rp, g2, m, n);
        }
}

// The QKV epilogue's q and k sections without the LDS round trip, 16 x 16 accumulator layout (the v section is written TRANSPOSED and keeps
// the LDS form).  The wave's 128 columns are one head.  After the pair swap of gemm_epilogue_direct16 a lane holds, per row, the four
// 8-column chunks c = 4 p + 2 (g & 1) + (g >> 1), p = 0 .. 3.  Per-head RMSNorm: the LDS form sums a row's squares as 16 chunk sums (8
// columns each, left to right) joined by an xor butterfly over the chunk index (1, 2, 4, 8).  Here level 1 (c ^ 1) is the lane 32 away,
// level 2 (c ^ 2) the lane 16 away (one swap each: the two results of swapping a value with itself are the two partners' values, and their
// sum is the level's sum on both sides), levels 4 and 8 run inside the lane over p: the same additions, so the same bits.
template <bool FP8>
__device__ __forceinline__ void gemm_epilogue_direct16_qk(const KARG GemmProblem& P, const f32x4 (&acc)[4][8], int m0, int n0, int lane, int w,
                                                          int section, int head) {
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    const int nw0 = n0 + wn * 128, mw0 = m0 + wm * 64;
    const bf16* bias = (const bf16*)P.bias;
    float sa[4] = {1.f, 1.f, 1.f, 1.f};
    if constexpr (FP8) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) sa[mb] = P.scale_a[mw0 + mb * 16 + l15];
    }
    bf16x4 bvs[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) bvs[nb] = bf16x4{0, 0, 0, 0};
    if (bias != nullptr) {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) bvs[nb] = *(const bf16x4*)(bias + nw0 + nb * 16 + 4 * g4);
    }
    uint32_t yp[4][8][2];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
            const bf16x4 b4 = bvs[nb];
            if constexpr (FP8) {
                yp[mb][nb][0] = pk2(f32x2{acc[mb][nb][0] * sa[mb] + (float)b4[0], acc[mb][nb][1] * sa[mb] + (float)b4[1]});
                yp[mb][nb][1] = pk2(f32x2{acc[mb][nb][2] * sa[mb] + (float)b4[2], acc[mb][nb][3] * sa[mb] + (float)b4[3]});
            } else {
                yp[mb][nb][0] = pk2(f32x2{acc[mb][nb][0] + (float)b4[0], acc[mb][nb][1] + (float)b4[1]});
                yp[mb][nb][1] = pk2(f32x2{acc[mb][nb][2] + (float)b4[2], acc[mb][nb][3] + (float)b4[3]});
            }
        }
    const bf16* nw = (const bf16*)(section == 0 ? P.norm_q_w : P.norm_k_w);
    bf16* dst = (bf16*)(section == 0 ? P.q_out : P.k_out);
    const float qs1 = (section == 0 && P.q_scale != 0.f) ? P.q_scale : 1.0f;
    const f32x2 qs = f32x2{qs1, qs1};
    const int S_pad = P.S_pad;
    const int cg = 2 * (g4 & 1) + (g4 >> 1);
    // the norm weights of the lane's four chunks (the same for every row)
    u32x4 wpv[4];
#pragma unroll
    for (int p_ = 0; p_ < 4; ++p_) wpv[p_] = *(const u32x4*)(nw + (4 * p_ + cg) * 8);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const int m = mw0 + mb * 16 + l15;
        // RoPE operands of the row's four chunks: requested before the swaps and sums
        f32x4 cs[4], sn[4];
#pragma unroll
        for (int p_ = 0; p_ < 4; ++p_) {
            cs[p_] = *(const f32x4*)(P.rope_cos + (size_t)m * 64 + (4 * p_ + cg) * 4);
            sn[p_] = *(const f32x4*)(P.rope_sin + (size_t)m * 64 + (4 * p_ + cg) * 4);
        }
        u32x4 vp[4];
        float ss[4];
#pragma unroll
        for (int p_ = 0; p_ < 4; ++p_) {
            const auto s0 = __builtin_amdgcn_permlane16_swap(yp[mb][2 * p_][0], yp[mb][2 * p_ + 1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(yp[mb][2 * p_][1], yp[mb][2 * p_ + 1][1], false, false);
            vp[p_] = u32x4{s0[0], s1[0], s0[1], s1[1]};
            float s_ = 0.f;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const f32x2 y2 = up2(vp[p_][jj]);
                const f32x2 sq = y2 * y2;
                s_ += sq.x;
                s_ += sq.y;
            }
            ss[p_] = s_;
        }
#pragma unroll
        for (int p_ = 0; p_ < 4; ++p_) {      // levels 1 and 2 of the butterfly: chunk c ^ 1 (lane ^ 32), then c ^ 2 (lane ^ 16)
            const auto e1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(ss[p_]), __float_as_uint(ss[p_]), false, false);
            ss[p_] = __uint_as_float(e1[0]) + __uint_as_float(e1[1]);
            const auto e2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(ss[p_]), __float_as_uint(ss[p_]), false, false);
            ss[p_] = __uint_as_float(e2[0]) + __uint_as_float(e2[1]);
        }
        const float t0 = ss[0] + ss[1], t2 = ss[2] + ss[3];      // level 4 (p ^ 1), level 8 (p ^ 2)
        const float tot = t0 + t2;
        // RMSNorm(128, eps 1e-6): models/utils.py:250-257
        const float rs = __builtin_amdgcn_rsqf(tot * (1.0f / 128.0f) + 1e-6f);
#pragma unroll
        for (int p_ = 0; p_ < 4; ++p_) {
            const int c = 4 * p_ + cg;
            u32x4 o;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const f32x2 x = rnd2(rnd2(up2(vp[p_][jj]) * f32x2{rs, rs}) * up2(wpv[p_][jj]));
                // apply_rotary_emb_qwen: fp32 complex multiply (qwen_image_dit.py:51-57)
                const f32x2 a = x * f32x2{cs[p_][jj], cs[p_][jj]};
                const f32x2 b = f32x2{x.y, x.x} * f32x2{sn[p_][jj], sn[p_][jj]};
                o[jj] = pk2(f32x2{a.x - b.x, a.y + b.y} * qs);
            }
            *(u32x4*)(dst + ((size_t)head * S_pad + P.seq_off + m) * 128 + c * 8) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Deferred epilogue (round 6, schedule 21, 16 x 16 accumulator layout, the direct epilogues' tiles).  At the end of a tile a wave only
// rounds y = bf16(acc + bias) and parks it in the work-group's STASH (device memory, written and read back by the same lane: it lives in
// the L2 / Infinity Cache) -- a quarter of the direct epilogue's instructions and nothing that waits on a load.  The epilogue proper (GELU;
// gate x y + residual) runs in DEFER_SLICES slices inside the NEXT tile's main loop: slice s = the lane's 16-byte chunk of (row block s >> 2,
// column-block pair s & 3) is requested at the start of M slot s (stash chunk; residual chunk and gate chunk for the gated form) and
// computed + stored behind the operand wait of L slot s + 1, i.e. beside the other wave group's MFMAs.  Same operations and roundings per
// element as gemm_epilogue_direct16: bit-identical.  Stash layout: slice-major, then wave, then lane: every store / load instruction of a
// wave covers 1 KiB contiguous.
// ------------------------------------------------------------------------------------------
constexpr int DEFER_SLICES = 16;
constexpr size_t DEFER_STASH_BYTES = (size_t)BM * BN * 2;      // per work-group of the persistent grid

// EPI_GATE_RES parks z = bf16(gate * y) -- the gated form's next rounding, taken here where the fragment registers are free for the gate
// vector -- so that its slices need the stash chunk and the residual chunk only (out = bf16(res + z)).
template <int EPI, bool FP8>
__device__ __forceinline__ void gemm_epilogue_dump16(const KARG GemmProblem& P, const f32x4 (&acc)[4][8], int m0, int n0, int lane, int w, char* stash_lane) {
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    const int nw0 = n0 + wn * 128, mw0 = m0 + wm * 64;
    const bf16* bias = (const bf16*)P.bias;
    float sa[4] = {1.f, 1.f, 1.f, 1.f};
    if constexpr (FP8) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) sa[mb] = P.scale_a[mw0 + mb * 16 + l15];
    }
    bf16x4 bvs[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) bvs[nb] = bf16x4{0, 0, 0, 0};
    if (bias != nullptr) {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) bvs[nb] = *(const bf16x4*)(bias + nw0 + nb * 16 + 4 * g4);
    }
    u32x4 gv[EPI == EPI_GATE_RES ? 4 : 1];
    if constexpr (EPI == EPI_GATE_RES) {
        const int ncol = (g4 & 1) * 16 + 8 * (g4 >> 1);
#pragma unroll
        for (int p_ = 0; p_ < 4; ++p_) gv[p_] = *(const u32x4*)((const bf16*)P.gate + nw0 + p_ * 32 + ncol);
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int p_ = 0; p_ < 4; ++p_) {
            uint32_t yp[2][2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int nb = 2 * p_ + e;
                const bf16x4 b4 = bvs[nb];
                if constexpr (FP8) {
                    yp[e][0] = pk2(f32x2{acc[mb][nb][0] * sa[mb] + (float)b4[0], acc[mb][nb][1] * sa[mb] + (float)b4[1]});
                    yp[e][1] = pk2(f32x2{acc[mb][nb][2] * sa[mb] + (float)b4[2], acc[mb][nb][3] * sa[mb] + (float)b4[3]});
                } else {
                    yp[e][0] = pk2(f32x2{acc[mb][nb][0] + (float)b4[0], acc[mb][nb][1] + (float)b4[1]});
                    yp[e][1] = pk2(f32x2{acc[mb][nb][2] + (float)b4[2], acc[mb][nb][3] + (float)b4[3]});
                }
            }
            const auto s0 = __builtin_amdgcn_permlane16_swap(yp[0][0], yp[1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(yp[0][1], yp[1][1], false, false);
            u32x4 vp = {s0[0], s1[0], s0[1], s1[1]};
            if constexpr (EPI == EPI_GATE_RES) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) vp[jj] = pk2(up2(gv[p_][jj]) * up2(vp[jj]));
            }
            *(u32x4*)(stash_lane + (size_t)(mb * 4 + p_) * (GEMM_THREADS * 16)) = vp;
        }
}

// the chunk a lane holds in slice s: row m0 + wm * 64 + (s >> 2) * 16 + (lane & 15), columns n0 + wn * 128 + (s & 3) * 32 + ncol .. + 7
struct DeferPos { int m, n; };
PE_DEV DeferPos defer_pos(int s, int pm0, int pn0, int lane, int w) {
    const int l15 = lane & 15, g4 = lane >> 4;
    DeferPos d;
    d.m = pm0 + (w >> 1) * 64 + (s >> 2) * 16 + l15;
    d.n = pn0 + (w & 1) * 128 + (s & 3) * 32 + (g4 & 1) * 16 + 8 * (g4 >> 1);
    return d;
}
template <int EPI, bool FP8>
__device__ __forceinline__ void defer_slice_emit(const KARG GemmProblem& P, DeferPos d, u32x4 vp, u32x4 rp) {
    u32x4 op;
    if constexpr (EPI == EPI_GATE_RES) {
#pragma unroll
        // vp = bf16(gate * y): rounded when it was parked.  SCALAR adds (build.py compiles gemm.hip without SLP vectorisation): a slice runs beside
        // the other wave group's MFMAs, and v_pk_*_f32 does not overlap with an MFMA at all on gfx950 (profiles/r03_valu_rate.log: 8 packed
        // operations + 1 MFMA take the SUM of their times, 8 plain ones + 1 MFMA little more than the longer of the two)
        for (int jj = 0; jj < 4; ++jj)
            op[jj] = pk2(f32x2{__uint_as_float(rp[jj] << 16) + __uint_as_float(vp[jj] << 16),
                               __uint_as_float(rp[jj] & 0xffff0000u) + __uint_as_float(vp[jj] & 0xffff0000u)});
    } else {
        const f32x2 g2[4] = {f32x2{1.f, 1.f}, f32x2{1.f, 1.f}, f32x2{1.f, 1.f}, f32x2{1.f, 1.f}};
        op = direct_epi_math<EPI, FP8>(P, vp, rp, g2, d.m, d.n);
    }
    *(u32x4*)((bf16*)P.out + (size_t)d.m * P.ldo + d.n) = op;
}

template <int EPI, bool FP8, int NMI = 2, int MI0 = 0, int LAY = 0, typename ACC>
__device__ __forceinline__ void gemm_epilogue_direct(const KARG GemmProblem& P, const ACC& acc, int m0, int n0, int lane, int w) {
    if constexpr (LAY == 1) {
        gemm_epilogue_direct16<EPI, FP8>(P, acc, m0, n0, lane, w);
        return;
    } else {
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = w >> 1, wn = w & 1;
    const int nw0 = n0 + wn * 128, mw0 = m0 + wm * 64;
    const bf16* bias = (const bf16*)P.bias;
    float sa[2] = {1.f, 1.f};
    if constexpr (FP8) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) sa[mi] = P.scale_a[mw0 + mi * 32 + l31];
    }
    bf16x4 bvs[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) bvs[i] = bf16x4{0, 0, 0, 0};
    if (bias != nullptr) {
        const bf16* bl = bias + nw0 + 4 * h;
#pragma unroll
        for (int i = 0; i < 16; ++i) bvs[i] = *(const bf16x4*)(bl + (i >> 2) * 32 + 8 * (i & 3));
    }
    // residual rows of the gated epilogue in the STORE layout: row mi * 32 + l31, columns ni * 32 + 16 j + 8 h .. + 7.  Row block 0 is
    // requested here (its latency rides under the packing of the accumulators), row block 1 once the accumulators are dead: with all 16
    // rows in flight next to 128 accumulators the allocator spills 125 registers
    u32x4 rv[EPI == EPI_GATE_RES ? 16 : 1];
    const bf16* rl = nullptr;
    if constexpr (EPI == EPI_GATE_RES) {
        rl = (const bf16*)P.res + (size_t)(mw0 + l31) * P.ldr + nw0 + 8 * h;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int j = 0; j < 2; ++j) rv[ni * 2 + j] = *(const u32x4*)(rl + ni * 32 + 16 * j);
    }
    // y = bf16(acc + bias) as pairs, every quad of the wave's block (64 registers; the accumulators are dead behind this)
    uint32_t yp[2][4][8];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bf16x4 b4 = bvs[ni * 4 + q];
                f32x2 y0, y1;
                if constexpr (FP8) {
                    y0 = f32x2{acc[MI0 + mi][ni][4 * q] * sa[mi] + (float)b4[0], acc[MI0 + mi][ni][4 * q + 1] * sa[mi] + (float)b4[1]};
                    y1 = f32x2{acc[MI0 + mi][ni][4 * q + 2] * sa[mi] + (float)b4[2], acc[MI0 + mi][ni][4 * q + 3] * sa[mi] + (float)b4[3]};
                } else {
                    y0 = f32x2{acc[MI0 + mi][ni][4 * q] + (float)b4[0], acc[MI0 + mi][ni][4 * q + 1] + (float)b4[1]};
                    y1 = f32x2{acc[MI0 + mi][ni][4 * q + 2] + (float)b4[2], acc[MI0 + mi][ni][4 * q + 3] + (float)b4[3]};
                }
                yp[mi][ni][2 * q] = pk2(y0);
                yp[mi][ni][2 * q + 1] = pk2(y1);
            }
    if constexpr (EPI == EPI_GATE_RES) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int j = 0; j < 2; ++j) rv[8 + ni * 2 + j] = *(const u32x4*)(rl + (size_t)32 * P.ldr + ni * 32 + 16 * j);
    }
    bf16* out = (bf16*)P.out;
    const float gs = (EPI == EPI_GATE_RES && P.has_gate_scalar) ? P.gate_scalar : 1.0f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = nw0 + ni * 32 + 16 * j + 8 * h;
                const int m = mw0 + mi * 32 + l31;
                const auto sx = __builtin_amdgcn_permlane32_swap(yp[mi][ni][4 * j], yp[mi][ni][4 * j + 2], false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(yp[mi][ni][4 * j + 1], yp[mi][ni][4 * j + 3], false, false);
                const u32x4 vp = {sx[0], sy[0], sx[1], sy[1]};
                u32x4 op;
                if constexpr (EPI == EPI_BIAS) {
                    op = vp;
                } else if constexpr (EPI == EPI_GELU_SIG) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const f32x2 y = up2(vp[jj]);
                        const f32x2 t = rnd2(f32x2{1.702f, 1.702f} * y);
                        const f32x2 a = t * f32x2{-1.4426950408889634f, -1.4426950408889634f};
                        const f32x2 d = f32x2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)} + f32x2{1.0f, 1.0f};
                        const f32x2 sg = rnd2(f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)});
                        op[jj] = pk2(y * sg);
                    }
                    if constexpr (FP8) {
                        if (P.q8_out != nullptr) {
                            const bf16x8 o = __builtin_bit_cast(bf16x8, op);
                            float f[8];
                            float amax = 0.f;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                f[e] = (float)o[e];
                                amax = fmaxf(amax, fabsf(f[e]));
                            }
                            u32x2 pk;
                            pk[0] = pack4_e4m3(f[0], f[1], f[2], f[3]);
                            pk[1] = pack4_e4m3(f[4], f[5], f[6], f[7]);
                            *(u32x2*)((uint8_t*)P.q8_out + (size_t)m * P.ldq8 + n) = pk;
                            if (amax > 447.0f) atomicOr(P.q8_flags + m, 1u);
                        }
                    }
                } else {
                    f32x2 g2[4] = {f32x2{gs, gs}, f32x2{gs, gs}, f32x2{gs, gs}, f32x2{gs, gs}};
                    if (P.gate != nullptr) {
                        const u32x4 gv = *(const u32x4*)((const bf16*)P.gate + n);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) g2[jj] = up2(gv[jj]);
                    }
                    const u32x4 rp = rv[(mi * 4 + ni) * 2 + j];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) op[jj] = pk2(up2(rp[jj]) + rnd2(g2[jj] * up2(vp[jj])));
                }
                *(u32x4*)(out + (size_t)m * P.ldo + n) = op;
            }
    }
}

template <int EPI, bool FP8, bool TWO_PASS, int NMI = 2, int MI0 = 0, int LAY = 0, typename ACC>
__device__ __forceinline__ void gemm_epilogue(const KARG GemmProblem& P, const int M, const int N, const ACC& acc, int m0, int n0,
                                              char* E0, char* E1, int lane, int w, long long* stamp4, int direct = 0) {
    if constexpr (kDirectEpi<EPI>) {
        if ((direct & 1) && m0 + BM <= M && n0 + BN <= N && P.pre == nullptr) {      // wave-uniform
            gemm_epilogue_direct<EPI, FP8, NMI, MI0, LAY>(P, acc, m0, n0, lane, w);
            return;
        }
    }
    if constexpr (EPI == EPI_QKV && LAY == 1) {      // (not EPI_QKV_STATS: its sums are taken in the LDS form)
        // bit 1 of "gemm_direct_epilogue": the q / k sections of complete tiles (wave-uniform: a wave's 128 columns are one head of one section)
        if ((direct & 2) && m0 + BM <= M && n0 + BN <= N && P.pre == nullptr) {
            const int HD = N / 3;
            const int nw0 = n0 + (w & 1) * 128;
            const int section = nw0 / HD;
            if (section < 2) {
                gemm_epilogue_direct16_qk<FP8>(P, acc, m0, n0, lane, w, section, (nw0 - section * HD) >> 7);
                return;
            }
        }
    }
    if (m0 + BM <= M && n0 + BN <= N && P.pre == nullptr)      // wave-uniform
        gemm_epilogue_body<EPI, FP8, TWO_PASS, true, NMI, MI0, LAY>(P, M, N, acc, m0, n0, E0, E1, lane, w, stamp4);
    else
        gemm_epilogue_body<EPI, FP8, TWO_PASS, false, NMI, MI0, LAY>(P, M, N, acc, m0, n0, E0, E1, lane, w, stamp4);
}

}  // namespace pe
