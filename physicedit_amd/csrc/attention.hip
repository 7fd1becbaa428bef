// Joint (image + text) flash attention for gfx950, head dim 128, no mask, bf16 in / fp32 softmax.
//
// Replaces qwen_image_flash_attention (DiffSynth-Studio/diffsynth/models/qwen_image_dit.py:14-39;
// CPU oracle = the SDPA branch :37) for q,k,v [1,24,S,128] -> [1,S,3072].  Numerics follow SDPA:
// scores, softmax statistics and the P.V accumulation in fp32, P rounded to bf16 for the second
// matmul, one rounding of the normalised output to bf16.
//
// Data layout (private contract with the QKV GEMM epilogue in gemm.hip):
//   Q, K : [H][S_pad][128] bf16 (RMSNorm + RoPE already applied)
//   Vt   : [H][128][S_pad] bf16, V transposed, and inside every aligned group of 16 tokens the
//          token with in-group index j sits at position perm16(j) = swap(bit2, bit3): stored order
//          [0-3, 8-11, 4-7, 12-15].  Pad columns (>= S) are zero.
//   out  : [S][ldo] bf16, head h in columns h*128..h*128+127  ("b s (n d)")
//
// Structure: one work-group = 8 waves x 32 query rows = 256 rows of one head; K and Vt tiles of 64
// keys go HBM -> LDS by LDS-DMA, double buffered, one barrier per tile.  Both matmuls are computed
// TRANSPOSED so that all softmax state is lane-local:
//   S^T[key][q] = K . Q^T   : MFMA A = K fragment (from LDS), B = Q fragment (registers, loaded once)
//       -> lane (q = lane&31) holds 16 of the 32 keys of a 32-key sub-tile, its partner lane^32 the rest
//   O^T[d][q]  += Vt . P^T  : MFMA A = Vt fragment (from LDS), B = P fragment
//       -> the B fragment of k-step (t,kk') wants, in lane half h, keys {16kk'+4h+r} U {16kk'+8+4h+r};
//          those are exactly accumulator quads 2kk', 2kk'+1 of S^T that the lane already holds, and
//          thanks to the perm16 storage order they are ONE contiguous 16-B chunk of a Vt row.
//          No permlane / LDS round trip for P.
// LDS tiles are XOR-swizzled at 16-B granularity (applied on the LDS-DMA source address and on the
// read) so every ds_read_b128 lane group is bank-conflict free.
#include <atomic>
#include <type_traits>
#include <utility>

#include "common.h"
#include "kernels.h"

namespace pe {

typedef __attribute__((ext_vector_type(8))) int i32x8f;
typedef __attribute__((ext_vector_type(4))) int i32x4f;
constexpr int KV_TILE = 64;
constexpr int KDEPTH = 4;   // K-fragment ds_reads kept in flight ahead of the QK^T MFMAs
constexpr int ATT_STAGE = 2 * KV_TILE * 128 * 2;  // K tile 16 KiB + Vt tile 16 KiB
constexpr int ATT_LDS = 2 * ATT_STAGE;            // 64 KiB
long long* g_attn_dbg = nullptr;   // device buffer for the s_memtime stamps of variant 3 / 4 (10 per work-group), or null
// 4 (default since round 3): flash_attn_w4_kernel with the lazy max update (the running max of a 32-row block is raised only when a
// row outgrows it by 2^8): decided by the repo's parity criterion, the rms distance to an fp32 evaluation relative to the
// reference-bf16's own, at 60 layers x S = 2208 (tests/test_gpu_parity_configs.py; profiles/r03_attention_notes.md).
// 0: flash_attn_kernel, 8 waves x 32 query rows, textbook update (P rounded at the scale of the reference's SDPA; the A/B knob,
// and always the kernel of the masked EliGen form);  3: flash_attn_w4_kernel with the textbook update, bit-identical to 0.
// 5 (default since round 4) / 6: the same kernel with the scale and the running max folded out of the softmax stream (FOLD
// below): 5 = lazy max like 4, 6 = textbook raise on every new maximum (exercises the raise path on every tile: the test form).
int g_attn_variant = 5;
// what the QKV epilogue multiplies Q by (GemmProblem.q_scale) for the current variant: scale . log2(e), or 1
float attn_q_prescale(float scale) { return g_attn_variant >= 5 ? scale * 1.44269504088896340736f : 1.0f; }

// Work decomposition.  total = H * nqb equal (head, q-block) items never divide evenly over the 256 CUs
// (cfg 2: 816 items = 3.19 rounds -> 4 rounds, 80 % efficiency).  So the first n_full = floor(total/slots)
// * slots items run whole, and each of the R leftover items is split `split`-ways along the KV sequence
// (flash-decoding style): the R*split short items write un-normalised partial (O, m, l) to a scratch
// buffer and attn_combine_kernel merges them.  cfg 2: 768 whole + 48 x 5 short = 3.2 rounds.
struct AttnPlan {
    int nqb, n_full, split;   // split == 1: no short items
};

// WORDS (EliGen entity control, QwenImageDiT.process_entity_masks, qwen_image_dit.py:433-498): the reference's additive 0 / -inf
// mask [S,S] is block-structured -- image tokens see each other, a prompt sees the image tokens of its region and itself, prompts do
// not see each other -- so it is carried as ONE 32-bit word per token: image token = bit 31 | the set of prompts whose region
// contains it; token of prompt i = bit i; (q, key) is allowed iff words[q] & words[key] != 0.  Rows [0, n_img) are image tokens;
// only tiles with a non-image key, and query blocks with a non-image row, pay for the test.  Entries [S, S_pad) must be 0.
template <int NW, bool WORDS = false>
__global__ void __launch_bounds__(NW * 64, 2)
flash_attn_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K, const bf16* __restrict__ Vt,
                  bf16* __restrict__ out, int S, int S_pad, int ldo, float scale_log2, AttnPlan plan,
                  float* __restrict__ part_o, float* __restrict__ part_ml, const uint32_t* __restrict__ words = nullptr,
                  int n_img = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int Q_BLOCK = NW * 32;
    constexpr int PIECES = 16 / NW;      // 1-KiB LDS-DMA pieces of the K tile (and of the Vt tile) moved by one wave
    const int lane = lane_id();
    const int w = wave_id();
    const int l31 = lane & 31, h = lane >> 5;

    // block -> (head, q-block): consecutive remapped ids walk the q-blocks of one head, so the
    // work-groups resident on one XCD share that head's K/V in its L2.
    const int nqb = plan.nqb;
    const int nt_all = (S + KV_TILE - 1) / KV_TILE;
    int item, t_begin = 0, t_end = nt_all, part_slot = -1;
    if ((int)blockIdx.x < plan.n_full) {
        item = xcd_remap((int)blockIdx.x, plan.n_full);
    } else {
        const int j = (int)blockIdx.x - plan.n_full;
        item = plan.n_full + j / plan.split;
        const int part = j - (j / plan.split) * plan.split;
        if (plan.split > 1) {
            t_begin = (int)((long long)nt_all * part / plan.split);
            t_end = (int)((long long)nt_all * (part + 1) / plan.split);
            part_slot = j;
        }
    }
    const int head = item / nqb;
    const int qb = item - head * nqb;
    const int q0 = qb * Q_BLOCK + w * 32;

    const bf16* Qh = Q + (size_t)head * S_pad * 128;
    const bf16* Kh = K + (size_t)head * S_pad * 128;
    const bf16* Vh = Vt + (size_t)head * 128 * S_pad;

    // Q fragments (MFMA B operand): lane supplies Q[q0 + l31][kk*16 + h*8 .. +8)
    bf16x8 qf[8];
    {
        const int qrow = min(q0 + l31, S - 1);
        const bf16* qp = Qh + (size_t)qrow * 128 + h * 8;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) qf[kk] = *(const bf16x8*)(qp + kk * 16);
    }

    uint32_t wq = 0xffffffffu;
    bool q_has_text = false;
    if constexpr (WORDS) {
        wq = words[min(q0 + l31, S - 1)];
        q_has_text = q0 + 32 > n_img;              // wave-uniform: some row of this wave's 32 is not an image token
    }

    // staging: wave w moves K pieces {2w, 2w+1} (4 rows x 256 B each) and Vt pieces {2w, 2w+1}
    // (8 rows x 128 B each)
    const bf16* k_src[PIECES];
    const bf16* v_src[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int piece = w * PIECES + i;
        const int krow = piece * 4 + (lane >> 4);
        const int kchunk = (lane & 15) ^ (krow & 15);
        k_src[i] = Kh + (size_t)krow * 128 + kchunk * 8;
        const int vrow = piece * 8 + (lane >> 3);
        const int vchunk = (lane & 7) ^ ((vrow >> 1) & 7);
        v_src[i] = Vh + (size_t)vrow * S_pad + vchunk * 8;
    }
    auto stage = [&](int buf, int t) {
        char* base = smem + buf * ATT_STAGE + w * PIECES * 1024;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            glds16(k_src[i] + (size_t)t * KV_TILE * 128, base + i * 1024);
            glds16(v_src[i] + t * KV_TILE, base + KV_TILE * 256 + i * 1024);
        }
    };

    f32x16 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY;  // running max, in the scaled log2 domain
    float l_run = 0.f;        // this lane's partial row sum (its 32 of every 64 keys)

    const int k_off = l31 * 256;                    // K row l31 (+32 rows for the second sub-tile)
    const int ksw = l31 & 15;
    const int v_off = KV_TILE * 256 + l31 * 128;    // Vt row d = l31 (+32 rows per dt)
    const int vsw = (l31 >> 1) & 7;

    // Building blocks of one KV tile.  VALU work is kept minimal because it is as large as the MFMA work here:
    // accumulators start from a constant-zero C operand (no v_mov zeroing), P is packed pairwise with v_cvt_pk (no
    // v_perm), the key mask is a compile-time flag (as a run-time condition the compiler if-converted it into 32
    // v_cndmask on EVERY tile), and the O rescale is skipped when no row's running max moved (alpha == 1 in every
    // lane: same numerics).
    // Q: sc[s2][4a+r] = score(q = q0+l31, key = t*64 + s2*32 + 8a + 4h + r)
    auto qk = [&](auto buf_tag, f32x16 (&sc)[2]) {
        const char* Sb = smem + decltype(buf_tag)::value * ATT_STAGE;   // compile-time stage: LDS addresses become immediates
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const bf16x8 kf = *(const bf16x8*)(Sb + k_off + s2 * 32 * 256 + (((kk * 2 + h) ^ ksw) << 4));
                sc[s2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], kk == 0 ? zero : sc[s2], 0, 0, 0);
            }
        }
    };
    // S: online-softmax statistics; sc scores -> probabilities; returns the factor the O accumulated so far must be
    // multiplied with BEFORE this tile's P.V is added.  No LDS traffic, no MFMA.
    auto softmax = [&](int t, f32x16 (&sc)[2], auto mask_tag, float& alpha, bool& moved) {
        constexpr bool MASK = decltype(mask_tag)::value;
        if constexpr (MASK) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * KV_TILE + s2 * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                    if (key >= S) sc[s2][r] = -INFINITY;
                }
        }
        if constexpr (WORDS) {
            if (q_has_text || (t + 1) * KV_TILE > n_img) {      // wave-uniform
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const u32x4 wk = *(const u32x4*)(words + t * KV_TILE + s2 * 32 + 8 * a + 4 * h);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if ((wq & wk[r]) == 0) sc[s2][4 * a + r] = -INFINITY;
                    }
            }
        }
        float mx = sc[0][0];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[s2][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float m_new = fmaxf(m_run, mx * scale_log2);
        moved = m_new != m_run;
        if constexpr (WORDS) {
            // a row may have seen no allowed key yet (a prompt whose region starts further down the sequence): m = -inf, and
            // -inf - (-inf) must not reach exp2.  The row's P is 0 and its O, l stay 0 until an allowed key arrives.
            const bool none = m_new == -INFINITY;
            alpha = none ? 1.0f : __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            if (none) m_new = 0.0f;
        } else {
            alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
        }
        float psum = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[s2][r], scale_log2, -m_new));
                sc[s2][r] = p;
                psum += p;
            }
        l_run = __builtin_fmaf(l_run, alpha, psum);
    };
    auto rescale = [&](float alpha, bool moved) {
        if (__any(moved)) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
    };
    // P as the four MFMA B-operand fragments of the tile: pk[s2*2+k2] = 8 bf16 = accumulator quads 2*k2, 2*k2+1 of sc[s2]
    auto pack = [&](const f32x16 (&sc)[2], u32x4 (&pk)[4]) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bf16x2 two;
                    two[0] = (bf16)sc[s2][(2 * k2) * 4 + 2 * (e & 1) + 4 * (e >> 1)];
                    two[1] = (bf16)sc[s2][(2 * k2) * 4 + 2 * (e & 1) + 4 * (e >> 1) + 1];
                    pk[s2 * 2 + k2][e] = __builtin_bit_cast(uint32_t, two);
                }
    };
    // P.V: O^T += Vt . P^T
    auto pv = [&](auto buf_tag, const u32x4 (&pk)[4]) {
        const char* Sb = smem + decltype(buf_tag)::value * ATT_STAGE;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pk[s2 * 2 + k2]);
                const int vchunk = s2 * 4 + k2 * 2 + h;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8 vf = *(const bf16x8*)(Sb + v_off + dt * 32 * 128 + ((vchunk ^ vsw) << 4));
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dt], 0, 0, 0);
                }
            }
    };

    const int nt = t_end;
    const bool tail = (S & (KV_TILE - 1)) != 0 && nt == nt_all;
    {
        f32x16 sc[2];
        u32x4 pk[4];
        auto tile = [&](int t, auto buf_tag, auto mask_tag) {
            float alpha;
            bool moved;
            qk(buf_tag, sc);
            // Left alone, hipcc emits  ds_read, s_waitcnt lgkmcnt(0), MFMA  sixteen times: every MFMA then waits a full
            // LDS round trip.  Pin a KDEPTH-deep fragment prefetch (reads run KDEPTH MFMAs ahead of their consumer).
            __builtin_amdgcn_sched_group_barrier(0x100, KDEPTH, 0);
#pragma unroll
            for (int i = 0; i < 16 - KDEPTH; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, KDEPTH, 0);
            softmax(t, sc, mask_tag, alpha, moved);
            rescale(alpha, moved);
            pack(sc, pk);
            pv(buf_tag, pk);
        };
        using B0 = std::integral_constant<int, 0>;
        using B1 = std::integral_constant<int, 1>;
        auto sync = [&]() {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        };
        // Tile t lives in stage (t - t_begin) & 1; the loop is unrolled by two so the stage is a compile-time constant
        // (no per-tile address arithmetic: ~15 VALU instructions per tile, and on this chip ordinary VALU instructions
        // take MFMA issue slots).  The (only) partially filled tile is peeled out (the key mask is a template flag).
        const int nt_loop = tail ? nt - 1 : nt;
        stage(0, t_begin);
        int t = t_begin;
        for (; t + 2 <= nt_loop; t += 2) {
            sync();
            stage(1, t + 1);                     // t + 1 < nt_loop <= nt
            tile(t, B0{}, std::false_type{});
            sync();
            if (t + 2 < nt) stage(0, t + 2);
            tile(t + 1, B1{}, std::false_type{});
        }
        if (t < nt_loop) {
            sync();
            if (t + 1 < nt) stage(1, t + 1);
            tile(t, B0{}, std::false_type{});
            ++t;
        }
        if (tail) {
            sync();
            if ((t - t_begin) & 1) tile(t, B1{}, std::true_type{});
            else tile(t, B0{}, std::true_type{});
        }
    }

    // ---- normalise and store: o[dt][4a+r] = O[q0+l31][dt*32 + 8a + 4h + r]
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (part_slot >= 0) {
        // short item: un-normalised fp32 partial + (running max, row sum) for attn_combine_kernel
        float* po = part_o + ((size_t)part_slot * Q_BLOCK + w * 32 + l31) * 128 + 4 * h;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = o[dt][4 * a + r];
                *(f32x4*)(po + dt * 32 + 8 * a) = v;
            }
        if (h == 0) {
            float* pm = part_ml + ((size_t)part_slot * Q_BLOCK + w * 32 + l31) * 2;
            pm[0] = m_run;
            pm[1] = l_tot;
        }
        return;
    }
    const float inv = 1.0f / l_tot;
    const int q = q0 + l31;
    if (q < S) {
        bf16* op = out + (size_t)q * ldo + head * 128 + 4 * h;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                bf16x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (bf16)(o[dt][4 * a + r] * inv);
                *(bf16x4*)(op + dt * 32 + 8 * a) = v;
            }
    }
}


// (Variant 2, a barrier-phased "ping-pong" of the two wave groups of this kernel, was bit-identical and 4.7 % slower:
// profiles/r02_attention_notes.md.  Removed.)
constexpr int PP_STAGES = 4;                    // variants 3 / 4: four-deep K / Vt ring
constexpr int PP_LDS = PP_STAGES * ATT_STAGE;   // 128 KiB

template <int OFF> PE_DEV void lds_read_to_a(u32x4& d, int addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(d) : "v"(addr), "i"(OFF));
}
template <int OFF> PE_DEV void lds_read_to_v(u32x4& d, int addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF));
}
// WAIT >= 0: the statement opens with s_waitcnt lgkmcnt(WAIT) -- the fragment operands come from asm LDS reads the compiler does not
// count; `also` is the second fragment that wait covers (named so that no use of it can be placed above this statement)
template <int WAIT>
PE_DEV void mfma_qk(f32x16& d, const u32x4& k_frag, const u32x4& q_frag, bool first, u32x4& also) {   // d (arch) = K(acc) . Q(acc) (+ d)
    if constexpr (WAIT >= 0) {
        if (first) asm volatile("s_waitcnt lgkmcnt(%4)\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %3, 0" : "=&v"(d), "+a"(also) : "a"(k_frag), "a"(q_frag), "n"(WAIT));
        else asm volatile("s_waitcnt lgkmcnt(%4)\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %3, %0" : "+v"(d), "+a"(also) : "a"(k_frag), "a"(q_frag), "n"(WAIT));
    } else {
        if (first) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "a"(k_frag), "a"(q_frag));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "a"(k_frag), "a"(q_frag));
    }
}
template <int WAIT>
PE_DEV void mfma_pv(f32x16& d, const u32x4& v_frag, const u32x4& p_frag, u32x4& also) {               // d (acc) += Vt(acc) . P(arch)
    if constexpr (WAIT >= 0)
        asm volatile("s_waitcnt lgkmcnt(%4)\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %3, %0" : "+a"(d), "+a"(also) : "a"(v_frag), "v"(p_frag), "n"(WAIT));
    else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "a"(v_frag), "v"(p_frag));
}
// x (op) x-of-lane^32 on every lane, by v_permlane32_swap (no LDS traffic: the LDS queue is counted by hand below).  After the swap
// of two copies, one register holds the low half's value on all lanes and the other the high half's.  A VALU write needs two
// wait states before v_permlane32_swap reads it.
PE_DEV float max_with_lane_xor32(float x) {
    float y;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_max_f32 %0, %0, %1" : "+v"(x), "=&v"(y));
    return x;
}
PE_DEV float sum_with_lane_xor32(float x) {
    float y;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_add_f32 %0, %0, %1" : "+v"(x), "=&v"(y));
    return x;
}

// Variant 3: 4 waves x 64 query rows, ONE wave per SIMD with the whole 512-register file, software-pipelined across KV tiles.
//   accumulator half: O (128) + Q fragments (64) + rolling windows of K and Vt fragments (~24 + ~16); arch half: the scores
//   of two tiles (128), P (32), addresses.  Every MFMA is an asm statement whose constraints name the
//   half each operand lives in, so nothing is copied across; all LDS reads are asm with hand-counted lgkmcnt.
//   iteration i = two phases of 32 MFMAs, one barrier:
//     phase 1  QK^T(tile i+1)  ||  K(i+1) fragment reads 2 k-steps ahead, second part of softmax(i): remaining exp2, row
//                                  sums, bf16 packs; O rescale if the max moved
//     phase 2  P.V(tile i)     ||  Vt(i) fragment reads 3 ahead, LDS-DMA of tile i+3,
//                                  first part of softmax(i+1): mask, row max, m / alpha, the first SM_EARLY exp2 per block
//   every (MFMA, fillers) pair is fenced with sched_barrier(0): source order is the schedule (~5 fillers per MFMA gap).
//   The arithmetic (and its order) is that of flash_attn_kernel: results are bit-identical.
// lane id recomputed where it is needed (asm volatile: not hoisted, not merged with the kernel-entry copy): a lane constant that
// lives across the KV loop for the sake of a rarely taken branch or of the epilogue is spilled to scratch and reloaded IN the loop
// behind a vmcnt(0) (cdna guide, "4-wave structure" pitfalls)
PE_DEV int lane_id_fresh() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
namespace w4 {
constexpr int SM_EARLY = 12;      // exp2 of a block's first SM_EARLY scores run in phase 2 (of the iteration before)
// softmax(i) is cut in PAIRS of scores (one bf16x2 of P each; 16 per block): the first SM_EARLY / 2 pairs of a block run in
// phase 2 of the iteration before (slices 10..31), the rest in phase 1 (slices 0..29).  pair index e: block e & 1
constexpr int early_lo(int g) { return g < 10 ? 0 : ((g - 10) * SM_EARLY) / 22; }
constexpr int late_lo(int g) { return g >= 30 ? 32 - SM_EARLY : (g * (32 - SM_EARLY)) / 30; }
static_assert(early_lo(32) == SM_EARLY && late_lo(0) == 0 && late_lo(32) == 32 - SM_EARLY, "softmax slices must cover every pair");
// LDS queue (all reads are asm, in issue order): phase 1  [K0..K3 of the tile, issued at the end of the phase 2 before]  K4 K5 (gaps 0, 1)
// K6 K7 (gaps 4, 5) ... K14 K15 (gaps 20, 21), Vt0..Vt3 (gaps 28..31); phase 2  Vt(f+4) in gap 2f, K0..K3 of the next tile in gaps
// 28..31.  One wait per two fragments, folded into the MFMA statement that first uses them: lgkmcnt(N), N = the reads issued after
// the second fragment by then.
constexpr int k_wait2(int kk) { return kk <= 6 ? 2 : 0; }      // before MFMA 4kk: fragments 2kk, 2kk+1
constexpr int v_wait2(int f) { return f <= 12 ? 2 : 0; }       // before MFMA 2f (f even): fragments f, f+1
}  // namespace w4

// FOLD (variants 5 / 6, round 4): the two v_fma_f32 per score pair (s . scale_log2 - m) leave the softmax stream.  Q arrives already
// multiplied by scale . log2(e) (the QKV epilogue applies it in fp32 BEFORE its one bf16 rounding: GemmProblem.q_scale), and the
// running max enters through the matrix pipe: the first k-step MFMA of every 32 x 32 score tile accumulates onto negm[b] (16
// registers holding -m of the lane's query row) instead of 0, so the accumulator IS the exp2 argument.  The row max of a tile is then
// measured relative to m: block b is "raised" when some row's max exceeds tau, by delta = max(row max, 0) per row -- m += delta,
// negm = -m, the tile's 32 scores -= delta, alpha = exp2(-delta) for O and l -- a wave-uniform branch that bounded data takes on the
// first tiles only.  P is single buffered (the registers negm takes): a 16-key chunk of P(i+1) is packed only after the last P.V MFMA
// of tile i that reads the chunk (asserted by the generator).
#ifndef PE_W4_PK2
#define PE_W4_PK2 0      // experiment (with W4_PK2=1 at generation): FOLD with the double-buffered P of the exact form
#endif
// PROBE (pe_attn_mix_probe): the folded schedule without its softmax -- the MFMAs, their LDS fragment reads, the LDS-DMA stream and the
// barrier of every iteration; P is whatever bf16 data K's first rows hold.  What it sustains on N(0,1) operands is the ceiling of this
// tiling and staging, as pe_gemm_mix_probe's is for the GEMM.  Its output is meaningless.
// MODE (FOLD only): 0 = the kernel, 1 = the mix probe.  (2 was round 6's experiment -- the row sums taken from the packed bf16 pairs with
// v_dot2c_f32_bf16, generated by tools/gen_attn_w4.py with W4_DOT2=1: correct, 32 issue slots per tile fewer, and 7 % SLOWER alone / 5.6 % per
// image, because dot instructions, like v_pk_*_f32, do not overlap with an MFMA on gfx950: profiles/r06_attention_notes.md.  Not in the library.)
template <bool FOLD, int MODE = 0>
__global__ void __launch_bounds__(256, 1)
flash_attn_w4_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K, const bf16* __restrict__ Vt,
                     bf16* __restrict__ out, int S, int S_pad, int ldo, float scale_log2, AttnPlan plan,
                     float* __restrict__ part_o, float* __restrict__ part_ml, float tau, long long* __restrict__ dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool PROBE = MODE == 1;
    constexpr int Q_BLOCK = 256;
    constexpr int KT_BYTES = KV_TILE * 256;        // 16 KiB
    constexpr int V_BASE = PP_STAGES * KT_BYTES;   // K ring [0, 64 KiB), Vt ring [64 KiB, 128 KiB); tile t sits in slot t & 3
    const long long stamp_begin = (long long)__builtin_readcyclecounter();
    long long stamp_it[4] = {0, 0, 0, 0};
    const int lane = lane_id();
    const int w = wave_id();
    const int l31 = lane & 31, h = lane >> 5;

    const int nqb = plan.nqb;
    const int nt_all = (S + KV_TILE - 1) / KV_TILE;
    int item, t_begin = 0, t_end = nt_all, part_slot = -1;
    if ((int)blockIdx.x < plan.n_full) {
        item = xcd_remap((int)blockIdx.x, plan.n_full);
    } else {
        const int j = (int)blockIdx.x - plan.n_full;
        item = plan.n_full + j / plan.split;
        const int part = j - (j / plan.split) * plan.split;
        if (plan.split > 1) {
            t_begin = (int)((long long)nt_all * part / plan.split);
            t_end = (int)((long long)nt_all * (part + 1) / plan.split);
            part_slot = j;
        }
    }
    const int head = item / nqb;
    const int qb = item - head * nqb;
    const int q0 = qb * Q_BLOCK + w * 64;
    const bf16* Qh = Q + (size_t)head * S_pad * 128;
    const bf16* Kh = K + (size_t)head * S_pad * 128;
    const bf16* Vh = Vt + (size_t)head * 128 * S_pad;

    u32x4 qf[2][8];          // Q fragments: defined in the accumulator half by asm, never moved again
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int qrow = min(q0 + b * 32 + l31, S - 1);
        const bf16* qp = Qh + (size_t)qrow * 128 + h * 8;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const u32x4 tq = *(const u32x4*)(qp + kk * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(qf[b][kk][e]) : "v"(tq[e]));
        }
    }
    // staging: wave w moves K pieces 4w..4w+3 (4 rows x 256 B each) and Vt pieces 4w..4w+3 (8 rows x 128 B each); the source is a
    // wave-uniform tile base plus a 32-bit per-lane byte offset (64-bit per-lane pointers would not fit the register budget)
    uint32_t k_off[4], v_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = w * 4 + i;
        const int krow = piece * 4 + (lane >> 4);
        k_off[i] = (uint32_t)(krow * 128 + ((lane & 15) ^ (krow & 15)) * 8) * 2u;
        const int vrow = piece * 8 + (lane >> 3);
        v_off[i] = (uint32_t)(vrow * S_pad + ((lane & 7) ^ ((vrow >> 1) & 7)) * 8) * 2u;
    }
    const int n = t_end - t_begin;
    // buffer-addressed LDS-DMA: SGPR descriptor of the head + SGPR tile offset + the lane's 32-bit offset (no 64-bit VALU math)
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Kh, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Vh, 0, 0x7fffffff, 0x00020000);
    auto stage_k = [&](int st, int i, int j) {      // i: tile index relative to t_begin, clamped (a tile past the end re-reads the last)
        const int t = __builtin_amdgcn_readfirstlane(t_begin + min(i, n - 1));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (__attribute__((address_space(3))) void*)(smem + st * KT_BYTES + w * 4096 + j * 1024),
                                                 16, (int)k_off[j], t * (KV_TILE * 256), 0, 0);
    };
    auto stage_v = [&](int st, int i, int j) {
        const int t = __builtin_amdgcn_readfirstlane(t_begin + min(i, n - 1));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (__attribute__((address_space(3))) void*)(smem + V_BASE + st * KT_BYTES + w * 4096 + j * 1024),
                                                 16, (int)v_off[j], t * (KV_TILE * 2), 0, 0);
    };
    auto stage = [&](int st, int i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { stage_k(st, i, j); stage_v(st, i, j); }
    };

    f32x16 o[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(o[b][dt][r]));
    float m_run[2] = {FOLD ? 0.f : -INFINITY, FOLD ? 0.f : -INFINITY}, l_run[2] = {0.f, 0.f};
    int kaddr[8], vaddr[4];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) kaddr[kk] = l31 * 256 + (((kk * 2 + h) ^ (l31 & 15)) << 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) vaddr[c] = V_BASE + l31 * 128 + (((c * 2 + h) ^ ((l31 >> 1) & 7)) << 4);

    f32x16 sc[2][2][2];      // [tile parity][q block][key half]
    std::conditional_t<FOLD && !PE_W4_PK2, u32x4[2][4], u32x4[2][2][4]> pk;       // P, bf16 pairs: [tile parity (not FOLD)][q block][16-key chunk]
    f32x16 negm[2];          // FOLD: -m of the lane's row of block b, 16 copies = the C operand of the tile's first MFMAs
    u32x4 kf[16], vf[16];    // K / Vt fragments, accumulator half: ~6 / ~4 live at a time
    float sm_mx[2], sm_sub[2], sm_alpha[2], sm_psum[2];
    bool sm_moved[2];

    // raises block b's running max (only when some row exceeds it by more than tau, see below) and derives what the exp2 slices
    // and the O rescale need.  `live` = the tile exists: iterations are issued in fours, a tile past the end contributes P = 0
    // (exp2(s - inf)) and leaves m, l alone.
    auto sm_state = [&](int b, bool live) __attribute__((always_inline)) {
        const float mx = max_with_lane_xor32(sm_mx[b]);
        // the running max is only raised (and O rescaled) when some row of the block exceeds it by more than tau (log2 units):
        // P = exp2(s - m) then stays <= 2^tau, and O / l does not depend on which m was used.  tau = 0 is the textbook
        // update, bit-identical to flash_attn_kernel.
        const float m_cand = fmaxf(m_run[b], mx * scale_log2);
        sm_moved[b] = live && __any(m_cand - m_run[b] > tau);
        const float m_new = sm_moved[b] ? m_cand : m_run[b];
        sm_alpha[b] = __builtin_amdgcn_exp2f(m_run[b] - m_new);
        m_run[b] = m_new;
        sm_sub[b] = live ? m_new : INFINITY;
        sm_psum[b] = 0.f;
        asm volatile("" : "+v"(sm_alpha[b]), "+v"(sm_sub[b]), "+v"(m_run[b]), "+v"(sm_psum[b]));
    };
    // FOLD form of sm_state for the scores sc[P][b] = s . c - m of one tile.  FIRST (the work item's first tile, m = 0 so far): m
    // becomes the row max whatever its sign and nothing is rescaled (O = l = 0).  A tile past the end arrives fully masked: row max
    // -inf, no move, P = exp2(-inf) = 0.
    auto sm_state_f = [&](auto p_tag, int b, auto first_tag) __attribute__((always_inline)) {
        constexpr int P = decltype(p_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        const float mx = max_with_lane_xor32(sm_mx[b]);
        const bool fix = FIRST || __any(mx > tau);
        sm_moved[b] = !FIRST && fix;
        float alpha = 1.0f;
        if (fix) {
            const float delta = FIRST ? mx : fmaxf(mx, 0.f);
            if constexpr (!FIRST) alpha = __builtin_amdgcn_exp2f(-delta);
            m_run[b] += delta;
            const float nm = -m_run[b];
            // the 16 copies are written by ONE instruction that defines the whole tuple: the exact-fp32 MFMA D = A . B with
            // A[i][k] = (k == 0), B[0][j] = -m of query row j (lane halves h = 0 / 1 carry k = 0 / 1 and hold the same m): 1 x nm + 0 x nm.
            // (16 C++ assignments make the register allocator keep two copies of the tuple and move it on the path that does NOT
            // raise; 16 tied asm operands spill).  Early clobber: a multi-pass MFMA may not write over its A / B operands.
            // s_nop: a VALU result needs wait states before an MFMA reads it as A / B, and nobody inserts them for an asm statement
            // (measured: with the select right in front of the MFMA the tuple came out wrong, with two instructions between, right)
            asm volatile("s_nop 3\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=&v"(negm[b]) : "v"(lane_id_fresh() < 32 ? 1.0f : 0.0f), "v"(nm));
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[P][b][s2][r] -= delta;
        }
        sm_alpha[b] = alpha;
        sm_psum[b] = 0.f;
        asm volatile("" : "+v"(sm_alpha[b]), "+v"(m_run[b]), "+v"(sm_psum[b]));
    };
    // keys at or past s_lim -> -inf: the ragged last tile, and every tile past the end of the sequence (iterations are issued in
    // fours; such a tile re-reads the last one, whose pad rows may hold anything, NaN included).  s_lim = S, or 0 (FOLD: a tile past
    // the end of a split-KV part holds real keys that are not this item's)
    auto mask_scores = [&](auto p_tag, int t, int s_lim) __attribute__((always_inline)) {
        constexpr int P = decltype(p_tag)::value;
        const int hf = FOLD ? lane_id_fresh() >> 5 : h;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int f = 0; f < 32; ++f) {
                const int s2 = f >> 4, r = f & 15;
                const int key = t * KV_TILE + s2 * 32 + 8 * (r >> 2) + 4 * hf + (r & 3);
                if (key >= s_lim) sc[P][b][s2][r] = -INFINITY;
            }
    };
    auto rescale = [&](int b) __attribute__((always_inline)) {     // O(block b) *= alpha, after the P.V of the tile before is done
        if (sm_moved[b]) {
            const float alpha = sm_alpha[b];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 16; r += 8) {
                    float t0, t1, t2, t3, t4, t5, t6, t7;
                    f32x16& acc = o[b][dt];
                    asm volatile("v_accvgpr_read_b32 %8, %0\n\tv_accvgpr_read_b32 %9, %1\n\tv_accvgpr_read_b32 %10, %2\n\tv_accvgpr_read_b32 %11, %3\n\t"
                                 "v_accvgpr_read_b32 %12, %4\n\tv_accvgpr_read_b32 %13, %5\n\tv_accvgpr_read_b32 %14, %6\n\tv_accvgpr_read_b32 %15, %7\n\t"
                                 "v_mul_f32 %8, %8, %16\n\tv_mul_f32 %9, %9, %16\n\tv_mul_f32 %10, %10, %16\n\tv_mul_f32 %11, %11, %16\n\t"
                                 "v_mul_f32 %12, %12, %16\n\tv_mul_f32 %13, %13, %16\n\tv_mul_f32 %14, %14, %16\n\tv_mul_f32 %15, %15, %16\n\t"
                                 "v_accvgpr_write_b32 %0, %8\n\tv_accvgpr_write_b32 %1, %9\n\tv_accvgpr_write_b32 %2, %10\n\tv_accvgpr_write_b32 %3, %11\n\t"
                                 "v_accvgpr_write_b32 %4, %12\n\tv_accvgpr_write_b32 %5, %13\n\tv_accvgpr_write_b32 %6, %14\n\tv_accvgpr_write_b32 %7, %15"
                                 : "+a"(acc[r]), "+a"(acc[r + 1]), "+a"(acc[r + 2]), "+a"(acc[r + 3]), "+a"(acc[r + 4]), "+a"(acc[r + 5]),
                                   "+a"(acc[r + 6]), "+a"(acc[r + 7]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5),
                                   "=&v"(t6), "=&v"(t7)
                                 : "v"(alpha));
                }
        }
    };

#define W4_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifndef PE_W4_STAMPS
#define PE_W4_STAMPS 0
#endif
    // the instruction schedule: lambdas iter0..iter3 (one per ring slot) and the prologue, generated by tools/gen_attn_w4.py
    long long stamp_loop;
    if constexpr (PROBE) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int r = 0; r < 16; ++r) negm[b][r] = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) pk[b][c] = *(const u32x4*)(Kh + (size_t)((b * 4 + c) * 64 + lane) * 8);       // stand-in for P
            l_run[b] = 1.0f;
        }
#include "attention_w5_probe_body.inc"
        stamp_loop = (long long)__builtin_readcyclecounter();
        for (int i = 0; i < n; i += 4) {
            iter0(i); iter1(i + 1); iter2(i + 2); iter3(i + 3);
        }
    } else if constexpr (FOLD) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) negm[b][r] = 0.f;
#include "attention_w5_body.inc"
        stamp_loop = (long long)__builtin_readcyclecounter();
        for (int i = 0; i < n; i += 4) {
            iter0(i); iter1(i + 1); iter2(i + 2); iter3(i + 3);
        }
    } else {
#include "attention_w4_body.inc"
        stamp_loop = (long long)__builtin_readcyclecounter();
        for (int i = 0; i < n; i += 4) {
            iter0(i); iter1(i + 1); iter2(i + 2); iter3(i + 3);
        }
    }
    const long long stamp_loop_end = (long long)__builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 7" ::: "memory");   // stray DMA / K reads; last P.V -> reads of O

    const int lane_e = FOLD ? lane_id_fresh() : lane;      // FOLD: nothing lane-derived lives across the loop for the epilogue's sake
    const int l31e = lane_e & 31, he = lane_e >> 5;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const float l_tot = sum_with_lane_xor32(l_run[b]);
        if (part_slot >= 0) {
            float* po = part_o + ((size_t)part_slot * Q_BLOCK + w * 64 + b * 32 + l31e) * 128 + 4 * he;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = o[b][dt][4 * a + r];
                    *(f32x4*)(po + dt * 32 + 8 * a) = v;
                }
            if (he == 0) {
                float* pm = part_ml + ((size_t)part_slot * Q_BLOCK + w * 64 + b * 32 + l31e) * 2;
                pm[0] = m_run[b];
                pm[1] = l_tot;
            }
            continue;
        }
        const float inv = 1.0f / l_tot;
        const int q = q0 + b * 32 + l31e;
        // A row's 8 consecutive columns 8a .. 8a+7 sit in two lanes (l31e and l31e + 32: 4 columns = 8 bytes each).  One
        // v_permlane32_swap per dword on the column groups (a, a + 1) -- upper half of group a <-> lower half of group a + 1 --
        // leaves 16 contiguous bytes of the row in every lane (lower lanes: group a, upper lanes: group a + 1): 8 dwordx4 stores
        // per 32-row block and lane instead of 16 dwordx2 (the store tail is issue-bound: cdna guide T21).  Same bytes, same values.
        bf16* op = out + (size_t)min(q, S - 1) * ldo + head * 128 + 8 * he;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int a = 0; a < 4; a += 2) {
                bf16x4 v0, v1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v0[r] = (bf16)(o[b][dt][4 * a + r] * inv);
                    v1[r] = (bf16)(o[b][dt][4 * (a + 1) + r] * inv);
                }
                const u32x2 w0 = __builtin_bit_cast(u32x2, v0), w1 = __builtin_bit_cast(u32x2, v1);
                const auto sx = __builtin_amdgcn_permlane32_swap(w0[0], w1[0], false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(w0[1], w1[1], false, false);
                const u32x4 pk = {sx[0], sy[0], sx[1], sy[1]};
                if (q < S) *(u32x4*)(op + dt * 32 + 8 * a) = pk;
            }
    }
    if (dbg != nullptr && threadIdx.x == 0) {      // profiling only (pe_debug_set_ptr("attn_stamps")): s_memtime of wave 0
        long long* d = dbg + (size_t)blockIdx.x * 10;
        d[0] = stamp_begin; d[1] = stamp_loop; d[2] = stamp_loop_end; d[3] = (long long)__builtin_readcyclecounter();
        d[4] = stamp_it[0]; d[5] = stamp_it[1]; d[6] = stamp_it[2]; d[7] = stamp_it[3]; d[8] = n; d[9] = 0;
    }
}

// ================================================================================================================
// Variant 7 (round 5): the folded one-wave-per-SIMD schedule of variants 5 / 6 on v_mfma_f32_16x16x32_bf16 -- the 4-pass MFMA shape spends less
// energy per FLOP (half the accumulator traffic), and on this power-limited chip energy is time (profiles/r05_gemm_notes.md section 7).
// Same work item, ring, staging, split-KV plan and arithmetic (Q pre-scaled, -m through the C operand, lazy max with tau); what changes is
// the tiling of a wave's 64 x 64 score tile into 16 x 16 blocks and everything that follows from the accumulator layout:
//   * lane l: l15 = l & 15 is the query row of a 16-row block qb (0..3), g = l >> 4 its k group; an MFMA block's 4 registers are rows 4g..4g+3
//   * MFMA key block m (0..3) does NOT take 16 consecutive keys: its row i is key 32 (m >> 1) + 16 (i >> 3) + 8 (m & 1) + (i & 7) of the tile,
//     i.e. lane group g holds keys 32c + 16 (g >> 1) + 8e + 4 (g & 1) + r (m = 2c + e).  The 8 k-slots a lane group feeds the P.V MFMA of the
//     32-key chunk c -- its 4 registers of block 2c, then of block 2c + 1 -- are then positions 8u .. 8u + 7 of a 16-group of Vt's perm16 layout
//     (u = g & 1, group 2c + (g >> 1)): ONE ds_read_b128 per Vt fragment, Vt's layout in memory unchanged.
//   * K tile in LDS: chunk ^ key(row), key = (row & 7) | ((row >> 1) & 8) (not row & 15): the 16 rows an MFMA block's fragment read touches
//     ((l & 7) + 16 (l >> 3) + const) get 16 distinct keys = the lane's l15: conflict-free ds_read_b128.
//   * -m of the lane's row as the C operand is 4 registers (written by one exact v_mfma_f32_16x16x4_f32), row max / row sum reductions cross
//     the 4 lane groups (v_permlane32_swap + v_permlane16_swap), the epilogue pairs d blocks with v_permlane16_swap for 16-byte stores.
// Not bit-identical with variants 3 - 6 (another summation order inside the MFMAs); same distance to an fp32 result.
// The instruction schedule (two MFMAs per 32-cycle gap) is generated: tools/gen_attn_w7.py -> attention_w7_body.inc.
// ================================================================================================================
PE_DEV float max_with_lane_xor16(float x) {      // v_permlane16_swap: odd 16-lane rows of the first <-> even rows of the second operand
    float y;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1\n\tv_max_f32 %0, %0, %1" : "+v"(x), "=&v"(y));
    return x;
}
PE_DEV float sum_with_lane_xor16(float x) {
    float y;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1\n\tv_add_f32 %0, %0, %1" : "+v"(x), "=&v"(y));
    return x;
}

__global__ void __launch_bounds__(256, 1)
flash_attn_w7_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K, const bf16* __restrict__ Vt,
                     bf16* __restrict__ out, int S, int S_pad, int ldo, AttnPlan plan,
                     float* __restrict__ part_o, float* __restrict__ part_ml, float tau) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int Q_BLOCK = 256;
    constexpr int KT_BYTES = KV_TILE * 256;        // 16 KiB
    constexpr int V_BASE = PP_STAGES * KT_BYTES;   // K ring [0, 64 KiB), Vt ring [64 KiB, 128 KiB); tile t sits in slot t & 3
    const int lane = lane_id();
    const int w = wave_id();
    const int l15 = lane & 15, g4 = lane >> 4;

    const int nqb = plan.nqb;
    const int nt_all = (S + KV_TILE - 1) / KV_TILE;
    int item, t_begin = 0, t_end = nt_all, part_slot = -1;
    if ((int)blockIdx.x < plan.n_full) {
        item = xcd_remap((int)blockIdx.x, plan.n_full);
    } else {
        const int j = (int)blockIdx.x - plan.n_full;
        item = plan.n_full + j / plan.split;
        const int part = j - (j / plan.split) * plan.split;
        if (plan.split > 1) {
            t_begin = (int)((long long)nt_all * part / plan.split);
            t_end = (int)((long long)nt_all * (part + 1) / plan.split);
            part_slot = j;
        }
    }
    const int head = item / nqb;
    const int qblk = item - head * nqb;
    const int q0 = qblk * Q_BLOCK + w * 64;
    const bf16* Qh = Q + (size_t)head * S_pad * 128;
    const bf16* Kh = K + (size_t)head * S_pad * 128;
    const bf16* Vh = Vt + (size_t)head * 128 * S_pad;

    // Q fragments [query block][k-step of 32], accumulator half.  ONE asm statement defines all 16 tuples (global loads straight into the
    // accumulator half, one wait): tuples assembled from per-register v_accvgpr_write statements are 64 separate live ranges to the allocator,
    // which then parks most of them in VGPRs and copies them in front of every QK^T statement (with no wait states before the MFMA)
    u32x4 qf[4][4];
    {
        const bf16* qp[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) qp[b] = Qh + (size_t)min(q0 + b * 16 + l15, S - 1) * 128 + g4 * 8;
        asm volatile(
            "global_load_dwordx4 %0, %16, off\n\tglobal_load_dwordx4 %1, %16, off offset:64\n\t"
            "global_load_dwordx4 %2, %16, off offset:128\n\tglobal_load_dwordx4 %3, %16, off offset:192\n\t"
            "global_load_dwordx4 %4, %17, off\n\tglobal_load_dwordx4 %5, %17, off offset:64\n\t"
            "global_load_dwordx4 %6, %17, off offset:128\n\tglobal_load_dwordx4 %7, %17, off offset:192\n\t"
            "global_load_dwordx4 %8, %18, off\n\tglobal_load_dwordx4 %9, %18, off offset:64\n\t"
            "global_load_dwordx4 %10, %18, off offset:128\n\tglobal_load_dwordx4 %11, %18, off offset:192\n\t"
            "global_load_dwordx4 %12, %19, off\n\tglobal_load_dwordx4 %13, %19, off offset:64\n\t"
            "global_load_dwordx4 %14, %19, off offset:128\n\tglobal_load_dwordx4 %15, %19, off offset:192\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&a"(qf[0][0]), "=&a"(qf[0][1]), "=&a"(qf[0][2]), "=&a"(qf[0][3]), "=&a"(qf[1][0]), "=&a"(qf[1][1]), "=&a"(qf[1][2]), "=&a"(qf[1][3]),
              "=&a"(qf[2][0]), "=&a"(qf[2][1]), "=&a"(qf[2][2]), "=&a"(qf[2][3]), "=&a"(qf[3][0]), "=&a"(qf[3][1]), "=&a"(qf[3][2]), "=&a"(qf[3][3])
            : "v"(qp[0]), "v"(qp[1]), "v"(qp[2]), "v"(qp[3])
            : "memory");
    }
    // staging as in flash_attn_w4_kernel, with this kernel's K swizzle key
    uint32_t k_off[4], v_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = w * 4 + i;
        const int krow = piece * 4 + (lane >> 4);
        const int kkey = (krow & 7) | ((krow >> 1) & 8);
        k_off[i] = (uint32_t)(krow * 128 + ((lane & 15) ^ kkey) * 8) * 2u;
        const int vrow = piece * 8 + (lane >> 3);
        v_off[i] = (uint32_t)(vrow * S_pad + ((lane & 7) ^ ((vrow >> 1) & 7)) * 8) * 2u;
    }
    const int n = t_end - t_begin;
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Kh, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Vh, 0, 0x7fffffff, 0x00020000);
    auto stage_k = [&](int st, int i, int j) {      // i: tile index relative to t_begin, clamped (a tile past the end re-reads the last)
        const int t = __builtin_amdgcn_readfirstlane(t_begin + min(i, n - 1));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (__attribute__((address_space(3))) void*)(smem + st * KT_BYTES + w * 4096 + j * 1024),
                                                 16, (int)k_off[j], t * (KV_TILE * 256), 0, 0);
    };
    auto stage_v = [&](int st, int i, int j) {
        const int t = __builtin_amdgcn_readfirstlane(t_begin + min(i, n - 1));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (__attribute__((address_space(3))) void*)(smem + V_BASE + st * KT_BYTES + w * 4096 + j * 1024),
                                                 16, (int)v_off[j], t * (KV_TILE * 2), 0, 0);
    };
    auto stage = [&](int st, int i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { stage_k(st, i, j); stage_v(st, i, j); }
    };

    f32x4 o[4][8];           // O^T [query block][d block], accumulator half (whole tuples: see rescale)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int db = 0; db < 8; ++db) {
            o[b][db] = f32x4{0.f, 0.f, 0.f, 0.f};
            asm volatile("" : "+a"(o[b][db]));
        }
    float m_run[4] = {0.f, 0.f, 0.f, 0.f};
    // softmax denominators, accumulated by the matrix pipe: lacc[b] += ones . P per 32-key chunk (every register of the tuple holds the row sum of
    // the lane's query over all keys so far -- of the bf16 P the numerator uses); `ones` = a fragment of bf16 1.0
    f32x4 lacc[4];
    u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    asm volatile("" : "+a"(ones));
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        lacc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("" : "+a"(lacc[b]));
    }
    int kaddr[4], vaddr[2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) kaddr[kk] = ((l15 & 7) + 16 * (l15 >> 3)) * 256 + (((kk * 4 + g4) ^ l15) << 4);
#pragma unroll
    for (int c = 0; c < 2; ++c) vaddr[c] = V_BASE + l15 * 128 + (((c * 4 + g4) ^ ((l15 >> 1) & 7)) << 4);

    f32x4 sc[2][4][4];       // [tile parity][query block][MFMA key block]
    u32x4 pk[4][2];          // P, bf16 pairs: [query block][32-key chunk]
    f32x4 negm[4];           // -m of the lane's row of block b, 4 copies = the C operand of the tile's first MFMAs
    u32x4 kf[16], vf[16];    // K / Vt fragments, accumulator half: ~8 / ~5 live at a time
    float sm_mx[4], sm_alpha[4];
    bool sm_moved[4];

    // sm_state_f of flash_attn_w4_kernel for the scores sc[P][b][*] = s . c - m of one tile
    auto sm_state_f = [&](auto p_tag, int b, auto first_tag) __attribute__((always_inline)) {
        constexpr int P = decltype(p_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        const float mx = max_with_lane_xor16(max_with_lane_xor32(sm_mx[b]));
        const bool fix = FIRST || __any(mx > tau);
        sm_moved[b] = !FIRST && fix;
        float alpha = 1.0f;
        if (fix) {
            const float delta = FIRST ? mx : fmaxf(mx, 0.f);
            if constexpr (!FIRST) alpha = __builtin_amdgcn_exp2f(-delta);
            m_run[b] += delta;
            const float nm = -m_run[b];
            // the 4 copies by ONE instruction that defines the tuple: the exact-fp32 MFMA D = A . B, A[i][k] = (k == 0), B[k][j] = -m of query
            // row j (the four k groups hold the same m): 1 x nm + 3 x (0 x nm)
            asm volatile("s_nop 3\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(negm[b]) : "v"(lane_id_fresh() < 16 ? 1.0f : 0.0f), "v"(nm));
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[P][b][m][r] -= delta;
        }
        sm_alpha[b] = alpha;
        asm volatile("" : "+v"(sm_alpha[b]), "+v"(m_run[b]));
    };
    // keys at or past s_lim -> -inf (the ragged last tile; every tile past the end of the sequence or of a split-KV part: s_lim = 0)
    auto mask_scores = [&](auto p_tag, int t, int s_lim) __attribute__((always_inline)) {
        constexpr int P = decltype(p_tag)::value;
        const int gf = lane_id_fresh() >> 4;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = t * KV_TILE + 32 * (m >> 1) + 16 * (gf >> 1) + 8 * (m & 1) + 4 * (gf & 1) + r;
                    if (key >= s_lim) sc[P][b][m][r] = -INFINITY;
                }
    };
    // O(block b) *= alpha, after the P.V of the tile before is done.  Plain vector arithmetic on the whole 4-register tuples (the compiler reads /
    // writes the accumulator half around the multiply): with the per-element asm operands of flash_attn_w4_kernel's rescale the merge behind this
    // rarely taken branch splits 32 tuples into 128 single-register live ranges, and the allocator then gathers and scatters the O tuples around
    // every P.V statement (300 v_accvgpr_mov per iteration)
    // The empty asm re-defines every tuple HERE: without it the register allocator may place the accumulator -> VGPR copies of the multiply
    // anywhere behind the tuple's last P.V statement -- right behind an MFMA it cannot see inside the asm, with no wait states (measured: the
    // raise path of one block read stale accumulators).  Behind gap 2 the P.V MFMAs of the tile before are > 100 cycles old.
    auto rescale = [&](int b) __attribute__((always_inline)) {
#pragma unroll
        for (int db = 0; db < 8; ++db) asm volatile("" : "+a"(o[b][db]));
        asm volatile("" : "+a"(lacc[b]));
        if (sm_moved[b]) {
            const float alpha = sm_alpha[b];
#pragma unroll
            for (int db = 0; db < 8; ++db) o[b][db] *= alpha;
            lacc[b] *= alpha;
        }
    };

#ifndef W4_FENCE
#define W4_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) negm[b][r] = 0.f;
#include "attention_w7_body.inc"
    for (int i = 0; i < n; i += 4) {
        iter0(i); iter1(i + 1); iter2(i + 2); iter3(i + 3);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 7" ::: "memory");   // stray DMA / K reads; last P.V -> reads of O

    const int lane_e = lane_id_fresh();      // nothing lane-derived lives across the loop for the epilogue's sake
    const int l15e = lane_e & 15, ge = lane_e >> 4;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const float l_tot = lacc[b][0];      // every register of every lane group holds the whole row's sum
        if (part_slot >= 0) {
            float* po = part_o + ((size_t)part_slot * Q_BLOCK + w * 64 + b * 16 + l15e) * 128 + 4 * ge;
#pragma unroll
            for (int db = 0; db < 8; ++db) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = o[b][db][r];
                *(f32x4*)(po + db * 16) = v;
            }
            if (ge == 0) {
                float* pm = part_ml + ((size_t)part_slot * Q_BLOCK + w * 64 + b * 16 + l15e) * 2;
                pm[0] = m_run[b];
                pm[1] = l_tot;
            }
            continue;
        }
        const float inv = 1.0f / l_tot;
        const int q = q0 + b * 16 + l15e;
        // a row's 16 columns of a d block sit in four lanes (4 columns each); one v_permlane16_swap per dword on the block pairs (db, db + 1)
        // leaves 8 consecutive columns in every lane: lane group g holds columns 8 (g >> 1) .. + 7 of block db + (g & 1)
        bf16* op = out + (size_t)min(q, S - 1) * ldo + head * 128 + (ge & 1) * 16 + 8 * (ge >> 1);
#pragma unroll
        for (int db = 0; db < 8; db += 2) {
            bf16x4 v0, v1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v0[r] = (bf16)(o[b][db][r] * inv);
                v1[r] = (bf16)(o[b][db + 1][r] * inv);
            }
            const u32x2 w0 = __builtin_bit_cast(u32x2, v0), w1 = __builtin_bit_cast(u32x2, v1);
            const auto sx = __builtin_amdgcn_permlane16_swap(w0[0], w1[0], false, false);
            const auto sy = __builtin_amdgcn_permlane16_swap(w0[1], w1[1], false, false);
            const u32x4 pkd = {sx[0], sy[0], sx[1], sy[1]};
            if (q < S) *(u32x4*)(op + db * 16) = pkd;
        }
    }
}

// merge the `split` partials of each leftover (head, q-block): O = sum_i O_i 2^(m_i - M) / sum_i l_i 2^(m_i - M)
__global__ void __launch_bounds__(256) attn_combine_kernel(const float* __restrict__ part_o,
                                                           const float* __restrict__ part_ml, bf16* __restrict__ out,
                                                           int S, int ldo, AttnPlan plan, int Q_BLOCK,
                                                           const float* __restrict__ out_scale = nullptr) {
    // one block = 8 query rows x 32 four-column chunks of one leftover item (grid.x = items * Q_BLOCK/8: the merge is a
    // short dependent chain per element, so it wants many small blocks, not a loop)
    const int per_item = Q_BLOCK / 8;
    const int r_item = (int)blockIdx.x / per_item;
    const int it = (int)blockIdx.x - r_item * per_item;
    const int item = plan.n_full + r_item;
    const int head = item / plan.nqb;
    const int qb = item - head * plan.nqb;
    const int e = it * 256 + (int)threadIdx.x;
    const int row = e >> 5, c = e & 31;
    const int q = qb * Q_BLOCK + row;
    if (q >= S) return;
    float M = -INFINITY;
    for (int i = 0; i < plan.split; ++i)
        M = fmaxf(M, part_ml[((size_t)(r_item * plan.split + i) * Q_BLOCK + row) * 2]);
    float L = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < plan.split; ++i) {
        const size_t base = (size_t)(r_item * plan.split + i) * Q_BLOCK + row;
        const float wgt = __builtin_amdgcn_exp2f(part_ml[base * 2] - M);
        L = __builtin_fmaf(part_ml[base * 2 + 1], wgt, L);
        const f32x4 v = *(const f32x4*)(part_o + base * 128 + c * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(v[j], wgt, acc[j]);
    }
    const float inv = 1.0f / L;
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (bf16)(acc[j] * inv);
    if (out_scale != nullptr) {                 // the e4m3 form: x.to(bf16) * v_std, a second bf16 rounding
        const float vs = *out_scale;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16)((float)o[j] * vs);
    }
    *(bf16x4*)(out + (size_t)q * ldo + head * 128 + c * 4) = o;
}

int g_attn_slots = 256;      // CUs: concurrently resident work-groups = slots x (8 / waves per work-group)
int g_attn_force_split = 0;  // tests: force an R / split decomposition on small problems

// S_q: the query rows [0, S_q) are wanted (whole q-blocks are computed: rows up to the block's end come for free); keys are always [0, S)
static AttnPlan make_plan(int H, int S, bool have_ws, int Q_BLOCK, int slots, int S_q) {
    AttnPlan p;
    p.nqb = (S_q + Q_BLOCK - 1) / Q_BLOCK;
    const int total = H * p.nqb;
    p.n_full = total;
    p.split = 1;
    if (!have_ws) return p;
    const int nt = (S + KV_TILE - 1) / KV_TILE;
    int R = total % slots;
    int split = R > 0 ? slots / R : 1;
    if (g_attn_force_split > 1) { R = total < 4 ? total : 4; split = g_attn_force_split; }
    if (split > 8) split = 8;
    if (split > nt) split = nt;
    if (R > 0 && split >= 2) {
        p.n_full = total - R;
        p.split = split;
    }
    return p;
}

int launch_attn_mix_probe(const void* q, const void* k, const void* vt, void* out, int H, int S, int S_pad, int ldo, hipStream_t stream) {
    PE_REQUIRE(q && k && vt && out && H > 0 && S >= 1024 && S_pad % KV_TILE == 0 && S_pad >= S && ldo % 8 == 0 && ldo >= H * 128 &&
                   ((uintptr_t)out & 15) == 0, "attn_mix_probe: bad arguments");
    static std::atomic<bool> configured{false};
    if (!configured.load(std::memory_order_acquire)) {
        const hipError_t e = hipFuncSetAttribute((const void*)flash_attn_w4_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS);
        if (e != hipSuccess) return set_error(PE_ERR_HIP, "attn_mix_probe: hipFuncSetAttribute: %s", hipGetErrorString(e));
        configured.store(true, std::memory_order_release);
    }
    AttnPlan plan;
    plan.nqb = (S + 255) / 256;
    plan.n_full = H * plan.nqb;
    plan.split = 1;
    hipLaunchKernelGGL((flash_attn_w4_kernel<true, 1>), dim3(plan.n_full), dim3(256), PP_LDS, stream, (const bf16*)q, (const bf16*)k,
                       (const bf16*)vt, (bf16*)out, S, S_pad, ldo, 1.0f, plan, (float*)nullptr, (float*)nullptr, 8.0f, (long long*)nullptr);
    return check_launch("flash_attn_w4_kernel<probe>");
}

size_t flash_attn_workspace_bytes(int H, int S) {
    (void)H; (void)S;
    // at most (slots - 1) leftover items x 8 partials ... bounded by slots short items in practice; size for the cap
    return (size_t)g_attn_slots * 256 * (128 + 2) * sizeof(float);   // slots x Q_BLOCK is the same for both variants
}

int launch_flash_attn(const void* q, const void* k, const void* vt, void* out, int H, int S, int S_pad,
                      int ldo, float scale, void* workspace, size_t workspace_bytes, hipStream_t stream, const void* words,
                      int n_img, bool q_prescaled, int S_q) {
    PE_REQUIRE(q && k && vt && out, "flash_attn: null pointer");
    PE_REQUIRE(S_q >= 0 && S_q <= S, "flash_attn: S_q=%d query rows of S=%d", S_q, S);
    if (S_q == 0) S_q = S;
    PE_REQUIRE(words == nullptr || (n_img >= 0 && n_img <= S && ((uintptr_t)words & 15) == 0),
               "flash_attn: token words need 0 <= n_img <= S and a 16-byte aligned buffer");      // always the 8-wave masked kernel
    PE_REQUIRE(H > 0 && S > 0, "flash_attn: empty problem (H=%d S=%d)", H, S);
    PE_REQUIRE(S_pad % KV_TILE == 0 && S_pad >= S, "flash_attn: S_pad=%d must be a multiple of %d and >= S=%d",
               S_pad, KV_TILE, S);
    PE_REQUIRE(ldo % 4 == 0 && ldo >= H * 128, "flash_attn: bad ldo=%d", ldo);
    PE_REQUIRE(g_attn_variant == 0 || (g_attn_variant >= 3 && g_attn_variant <= 8), "flash_attn: attn_variant %d does not exist", g_attn_variant);
    // variants 5 / 6 need Q = q . scale . log2(e) (attn_q_prescale()); a caller with a plain Q gets the same schedule's exact form
    int variant = g_attn_variant;
    if (variant >= 5 && !q_prescaled) variant = (variant == 6 || variant == 8) ? 3 : 4;      // (7 -> 4, 8 -> 3: the 32 x 32 schedule's exact forms)
    // the one-wave-per-SIMD kernels store 16-byte vectors: rows must be 16-byte aligned (the 8-wave kernel needs 8)
    if (variant >= 3 && (ldo % 8 != 0 || ((uintptr_t)out & 15) != 0)) variant = 0;
    static std::atomic<bool> configured{false};   // racing first calls both configure: idempotent
    if (!configured.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void*)flash_attn_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)flash_attn_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)flash_attn_w4_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)flash_attn_w4_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)flash_attn_w7_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS);
        if (e != hipSuccess) return set_error(PE_ERR_HIP, "flash_attn: hipFuncSetAttribute: %s", hipGetErrorString(e));
        configured.store(true, std::memory_order_release);
    }
    const bool have_ws = workspace != nullptr && workspace_bytes >= flash_attn_workspace_bytes(H, S) &&
                         ((uintptr_t)workspace & 15) == 0;
    constexpr int Q_BLOCK = 256;                 // every variant: one work-group = 256 query rows of one head
    const int slots = g_attn_slots;
    const AttnPlan plan = make_plan(H, S, have_ws, Q_BLOCK, slots, S_q);
    const int total = H * plan.nqb;
    const int n_short = (total - plan.n_full) * plan.split;
    PE_REQUIRE(plan.split == 1 || n_short <= slots, "flash_attn: internal plan error");
    float* part_o = (float*)workspace;
    float* part_ml = part_o ? part_o + (size_t)g_attn_slots * 256 * 128 : nullptr;
    const float scale_log2 = q_prescaled ? 1.0f : scale * 1.44269504088896340736f;     // s . 1 - m is exact: Q carries the factor
    const int slot = prof_begin(PROF_ATTN, 4.0 * (double)S_q * S * 128.0 * H, stream);  // QK^T + PV of the wanted query rows
    const dim3 grid(plan.n_full + (plan.split > 1 ? n_short : 0));
    if (words != nullptr)
        hipLaunchKernelGGL((flash_attn_kernel<8, true>), grid, dim3(512), ATT_LDS, stream, (const bf16*)q, (const bf16*)k,
                           (const bf16*)vt, (bf16*)out, S, S_pad, ldo, scale_log2, plan, part_o, part_ml, (const uint32_t*)words, n_img);
    else if (variant >= 7)      // 8 = 7 with the textbook max update (tests: the raise path on nearly every tile)
        hipLaunchKernelGGL(flash_attn_w7_kernel, grid, dim3(256), PP_LDS, stream, (const bf16*)q, (const bf16*)k, (const bf16*)vt,
                           (bf16*)out, S, S_pad, ldo, plan, part_o, part_ml, variant == 7 ? 8.0f : 0.0f);
    else if (variant >= 5)
        hipLaunchKernelGGL(flash_attn_w4_kernel<true>, grid, dim3(256), PP_LDS, stream, (const bf16*)q, (const bf16*)k, (const bf16*)vt,
                           (bf16*)out, S, S_pad, ldo, 1.0f, plan, part_o, part_ml, variant == 5 ? 8.0f : 0.0f, g_attn_dbg);
    else if (variant >= 3)
        hipLaunchKernelGGL(flash_attn_w4_kernel<false>, grid, dim3(256), PP_LDS, stream, (const bf16*)q, (const bf16*)k, (const bf16*)vt,
                           (bf16*)out, S, S_pad, ldo, scale_log2, plan, part_o, part_ml, variant == 4 ? 8.0f : 0.0f, g_attn_dbg);
    else
        hipLaunchKernelGGL((flash_attn_kernel<8, false>), grid, dim3(512), ATT_LDS, stream, (const bf16*)q, (const bf16*)k,
                           (const bf16*)vt, (bf16*)out, S, S_pad, ldo, scale_log2, plan, part_o, part_ml, (const uint32_t*)nullptr, 0);
    int rc = check_launch("flash_attn_kernel");
    if (rc == PE_OK && plan.split > 1) {
        hipLaunchKernelGGL(attn_combine_kernel, dim3((total - plan.n_full) * (Q_BLOCK / 8)), dim3(256), 0, stream, part_o, part_ml,
                           (bf16*)out, S, ldo, plan, Q_BLOCK);
        rc = check_launch("attn_combine_kernel");
    }
    prof_end(slot, stream);
    return rc;
}

// ================================================================================================================
// e4m3 attention: qwen_image_flash_attention(enable_fp8_attention=True), qwen_image_dit.py:24-35 -- the branch the reference only takes
// with FlashAttention-3 on Hopper.  q, k, v are divided by their GLOBAL standard deviations (torch.std over the whole [1,H,S,128]
// tensor: unbiased, a bf16 scalar), cast to float8_e4m3fn, both matmuls run on e4m3 operands (CDNA4: v_mfma_scale_f32_32x32x64_f8f6f4
// with unit block scales, 2x the bf16 rate), softmax_scale = q_std * k_std / sqrt(128), and the bf16 output is multiplied by v_std.
// What FA3 does inside is restated from its published design (oracle/physicedit_oracle.py flash_attention_fp8: P = exp(s - max) cast
// to e4m3, fp32 row sums and accumulation): parity of the P quantisation itself is UNPINNED -- FA3 cannot run here.
//
//   attn_fp8_stats_kernel / _finish : sum and sum of squares of Q, K (rows < S) and Vt, in double from fp32 per-thread partials, in a
//                                     fixed order -> {q_std, k_std, v_std} rounded to bf16, scale_log2 = bf16r(bf16r(q_std k_std) /
//                                     sqrt(128)) * log2 e
//   attn_fp8_quant_kernel           : Q8, K8 [H][S_pad][128] = e4m3(bf16r(x / std)); Vt8 [H][128][S_pad] likewise, keys re-ordered
//                                     inside every 64-key tile so that the P.V MFMA needs no shuffle (below)
//   flash_attn_fp8_kernel           : 8 waves x 32 query rows like flash_attn_kernel; a KV tile is 8 KiB of K8 + 8 KiB of Vt8.
// Operand layout of the e4m3 MFMA (tools/microbench/mfma_scale_f8_layout_probe.hip): lane l supplies row l & 31 and the 32
// contiguous bytes k = 32 (l >> 5) .. + 31; D as the bf16 MFMA.  S^T = K . Q^T leaves lane (q = l & 31, h = l >> 5) with the scores of
// keys 32 s2 + 8 a + 4 h + b (r = 4 a + b of accumulator s2): 32 of the tile's 64 keys, i.e. exactly the 32 bytes of ITS half of
// the P.V MFMA's 64 k-slots once packed to e4m3 -- k-slot 32 h + j (j = 16 s2 + 4 a + b) is key 32 s2 + 8 a + 4 h + b, which is the
// order Vt8 stores a tile's keys in.
// ================================================================================================================
constexpr int F8_STAGE = 2 * KV_TILE * 128;       // K8 tile 8 KiB + Vt8 tile 8 KiB
constexpr int F8_LDS = 2 * F8_STAGE;
constexpr int F8_STAT_WGS = 512;

// grid (F8_STAT_WGS, 3): tensor z = Q (rows < S of every head), K (same), Vt (the positions of tokens < S; what the pad columns hold is ignored)
__global__ void __launch_bounds__(256) attn_fp8_stats_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                             const bf16* __restrict__ Vt, int H, int S, int S_pad,
                                                             double* __restrict__ part) {
    __shared__ double red[2][4];
    const int z = (int)blockIdx.y;
    const bf16* base = z == 0 ? Q : z == 1 ? K : Vt;
    // rows of 128 elements; Q / K: row r of head h is valid iff r < S.  Vt: every row d of a head has S_pad columns = S_pad / 128 ... walk
    // it as [H * 128][S_pad] in chunks of 8 elements
    const size_t n8 = z < 2 ? (size_t)H * S_pad * 16 : (size_t)H * 128 * (S_pad / 8);
    float s1 = 0.f, s2 = 0.f;
    double d1 = 0.0, d2 = 0.0;
    int cnt = 0;
    const int per_row8 = S_pad >> 3;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)F8_STAT_WGS * 256) {
        int g0 = 0, hh = 0;       // Vt: first token of the chunk's 16-group, which 8-position half of the group the chunk is
        if (z < 2) {
            const size_t row = i >> 4;                      // 16 chunks of 8 per 128-wide row
            if ((int)(row % (size_t)S_pad) >= S) continue;
        } else {
            // Vt's columns [S, S_pad) belong to no token: a caller that re-uses one plane for sequences of different lengths (the CFG
            // pair's two prompt lengths on one pe_dit handle) leaves another call's values there.  Position p of an aligned 16-group
            // holds token perm16(p): positions 8 hh + j (j = 0..7) are tokens 4 hh + (j & 3) + 8 (j >> 2).
            const int c0 = (int)(i % (size_t)per_row8) << 3;
            g0 = c0 & ~15;
            hh = (c0 >> 3) & 1;
        }
        const bf16x8 v = *(const bf16x8*)(base + i * 8);
        const bool ragged = z == 2 && g0 + 16 > S;          // the group that straddles S and the all-pad groups behind it (counted as
                                                            // zeros, not skipped: the fp32 partials keep the grouping they have on zeroed pads)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x = (float)v[j];
            if (ragged && g0 + 4 * hh + (j & 3) + 8 * (j >> 2) >= S) x = 0.f;
            s1 += x;
            s2 = __builtin_fmaf(x, x, s2);
        }
        if (++cnt == 64) { d1 += (double)s1; d2 += (double)s2; s1 = s2 = 0.f; cnt = 0; }      // fp32 partials of <= 512 elements
    }
    d1 += (double)s1;
    d2 += (double)s2;
    for (int o = 32; o > 0; o >>= 1) {
        d1 += __shfl_xor(d1, o, 64);
        d2 += __shfl_xor(d2, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = d1; red[1][threadIdx.x >> 6] = d2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[((size_t)z * F8_STAT_WGS + blockIdx.x) * 2] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        part[((size_t)z * F8_STAT_WGS + blockIdx.x) * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}
// stats[0..2] = q_std, k_std, v_std (bf16 values as float), stats[3] = softmax scale in the log2 domain
__global__ void __launch_bounds__(192) attn_fp8_stats_finish_kernel(const double* __restrict__ part, double n, float* __restrict__ stats) {
    // wave z reduces tensor z: lane l adds the partials l, l + 64, ... in that order, then the lanes meet in a fixed butterfly
    __shared__ float sd[3];
    const int z = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    double a = 0.0, b = 0.0;
    for (int i = lane; i < F8_STAT_WGS; i += 64) {
        a += part[((size_t)z * F8_STAT_WGS + i) * 2];
        b += part[((size_t)z * F8_STAT_WGS + i) * 2 + 1];
    }
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
    }
    if (lane == 0) {
        const double mean = a / n;
        const double var = (b - n * mean * mean) / (n - 1.0);           // torch.std: unbiased
        sd[z] = bf16r((float)sqrt(var > 0.0 ? var : 0.0));
        stats[z] = sd[z];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float qk = bf16r(sd[0] * sd[1]);                           // q_std * k_std: a bf16 tensor product ...
        stats[3] = bf16r(qk / 11.3137084989847603904f) * 1.44269504088896340736f;      // ... / sqrt(128): bf16 again; then log2 e for exp2
    }
}

// The same three standard deviations from the QKV epilogue's partial sums (EPI_QKV_STATS, gemm_tile.h): wave z adds the n_part entries of
// section z in index order (lane l: entries l, l + 64, ...), then the lanes meet in a fixed butterfly -- no pass over q / k / vt.
__global__ void __launch_bounds__(1024) attn_fp8_stats_finish_parts_kernel(const double* __restrict__ part, int n_part, double n, float* __restrict__ stats) {
    // one work-group of 16 waves: per section, thread t adds entries t, t + 1024, ... in that order, the waves meet in a fixed butterfly and
    // thread z adds the 16 wave sums in wave order (a single wave walking all ~6800 entries took 40 us: as long as the pass it replaces)
    __shared__ double red[3][16][2];
    __shared__ float sd[3];
    const int t = (int)threadIdx.x, lane = t & 63, w = t >> 6;
#pragma unroll
    for (int z = 0; z < 3; ++z) {
        double a = 0.0, b = 0.0;
        for (int i = t; i < n_part; i += 1024) {
            a += part[((size_t)z * n_part + i) * 2];
            b += part[((size_t)z * n_part + i) * 2 + 1];
        }
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_xor(a, o, 64);
            b += __shfl_xor(b, o, 64);
        }
        if (lane == 0) { red[z][w][0] = a; red[z][w][1] = b; }
    }
    __syncthreads();
    if (t < 3) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < 16; ++i) { a += red[t][i][0]; b += red[t][i][1]; }
        const double mean = a / n;
        const double var = (b - n * mean * mean) / (n - 1.0);           // torch.std: unbiased
        sd[t] = bf16r((float)sqrt(var > 0.0 ? var : 0.0));
        stats[t] = sd[t];
    }
    __syncthreads();
    if (t == 0) {
        const float qk = bf16r(sd[0] * sd[1]);
        stats[3] = bf16r(qk / 11.3137084989847603904f) * 1.44269504088896340736f;
    }
}

// Q8 / K8: thread = 16 consecutive elements of a row; Vt8: thread = 16 consecutive BYTE positions of a row's tile (two 8-element groups of
// the perm16 source order: see the layout note above)
__global__ void __launch_bounds__(256) attn_fp8_quant_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                             const bf16* __restrict__ Vt, uint8_t* __restrict__ Q8,
                                                             uint8_t* __restrict__ K8, uint8_t* __restrict__ Vt8, int H, int S_pad,
                                                             const float* __restrict__ stats) {
    const int z = (int)blockIdx.y;
    const size_t n16 = (size_t)H * S_pad * 8;          // 16-element groups per tensor
    const float sd = stats[z];
    auto q16 = [&](const bf16x8 a, const bf16x8 b) -> u32x4 {
        float f[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) { f[j] = bf16r((float)a[j] / sd); f[8 + j] = bf16r((float)b[j] / sd); }      // x / std: a bf16 tensor
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = pack4_e4m3(f[4 * j], f[4 * j + 1], f[4 * j + 2], f[4 * j + 3]);
        return o;
    };
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        if (z < 2) {
            const bf16* src = (z == 0 ? Q : K) + i * 16;
            *(u32x4*)((z == 0 ? Q8 : K8) + i * 16) = q16(*(const bf16x8*)src, *(const bf16x8*)(src + 8));
        } else {
            // row = i / (S_pad / 16) of [H * 128], p16 = 16-byte slot inside the row: tile = p16 >> 2, hh = (p16 >> 1) & 1, s2 = p16 & 1
            const int per_row = S_pad / 16;
            const size_t row = i / per_row;
            const int p16 = (int)(i - row * per_row);
            const int tile = p16 >> 2, hh = (p16 >> 1) & 1, s2 = p16 & 1;
            const bf16* src = Vt + row * S_pad + tile * 64 + (2 * s2) * 16 + 8 * hh;
            *(u32x4*)(Vt8 + row * S_pad + tile * 64 + 32 * hh + 16 * s2) = q16(*(const bf16x8*)src, *(const bf16x8*)(src + 16));
        }
    }
}

template <int NW>
__global__ void __launch_bounds__(NW * 64, 2)
flash_attn_fp8_kernel(const uint8_t* __restrict__ Q8, const uint8_t* __restrict__ K8, const uint8_t* __restrict__ Vt8,
                      bf16* __restrict__ out, int S, int S_pad, int ldo, const float* __restrict__ stats, AttnPlan plan,
                      float* __restrict__ part_o, float* __restrict__ part_ml) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int Q_BLOCK = NW * 32;
    static_assert(NW == 8, "one K8 piece and one Vt8 piece per wave");
    const int lane = lane_id();
    const int w = wave_id();
    const int l31 = lane & 31, h = lane >> 5;
    const float scale_log2 = stats[3], v_std = stats[2];
    const int nqb = plan.nqb;
    const int nt_all = (S + KV_TILE - 1) / KV_TILE;
    int item, t_begin = 0, t_end = nt_all, part_slot = -1;
    if ((int)blockIdx.x < plan.n_full) {
        item = xcd_remap((int)blockIdx.x, plan.n_full);
    } else {
        const int j = (int)blockIdx.x - plan.n_full;
        item = plan.n_full + j / plan.split;
        const int part = j - (j / plan.split) * plan.split;
        if (plan.split > 1) {
            t_begin = (int)((long long)nt_all * part / plan.split);
            t_end = (int)((long long)nt_all * (part + 1) / plan.split);
            part_slot = j;
        }
    }
    const int head = item / nqb;
    const int qb = item - head * nqb;
    const int q0 = qb * Q_BLOCK + w * 32;
    const uint8_t* Qh = Q8 + (size_t)head * S_pad * 128;
    const uint8_t* Kh = K8 + (size_t)head * S_pad * 128;
    const uint8_t* Vh = Vt8 + (size_t)head * 128 * S_pad;

    i32x8f qf[2];            // Q fragments (B operand): query row l31, bytes 64 kk + 32 h .. + 31
    {
        const int qrow = min(q0 + l31, S - 1);
        const uint8_t* qp = Qh + (size_t)qrow * 128 + h * 32;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const i32x4f lo = *(const i32x4f*)(qp + kk * 64), hi = *(const i32x4f*)(qp + kk * 64 + 16);
            qf[kk] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }
    // staging: wave w moves K8 piece w (8 key rows x 128 B) and Vt8 piece w (16 d rows x 64 B): 16-byte chunk c of row r sits at
    // chunk c ^ (r & 7) (K8) / c ^ ((r >> 1) & 3) (Vt8) of the LDS row
    const uint8_t* k_src;
    const uint8_t* v_src;
    {
        const int krow = w * 8 + (lane >> 3);
        k_src = Kh + (size_t)krow * 128 + (((lane & 7) ^ (krow & 7)) << 4);
        const int vrow = w * 16 + (lane >> 2);
        v_src = Vh + (size_t)vrow * S_pad + (((lane & 3) ^ ((vrow >> 1) & 3)) << 4);
    }
    auto stage = [&](int buf, int t) {
        char* base = smem + buf * F8_STAGE + w * 1024;
        glds16(k_src + (size_t)t * KV_TILE * 128, base);
        glds16(v_src + t * KV_TILE, base + KV_TILE * 128);
    };
    f32x16 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int ksw = l31 & 7, vsw = (l31 >> 1) & 3;

    auto rd32 = [&](const char* rowp, int c0, int sw) -> i32x8f {      // the 32 bytes at chunks c0, c0 + 1 of a swizzled LDS row
        const i32x4f lo = *(const i32x4f*)(rowp + ((c0 ^ sw) << 4)), hi = *(const i32x4f*)(rowp + (((c0 + 1) ^ sw) << 4));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto tile = [&](int t, auto buf_tag, auto mask_tag) {
        constexpr bool MASK = decltype(mask_tag)::value;
        const char* Sb = smem + decltype(buf_tag)::value * F8_STAGE;
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 sc[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const i32x8f kf = rd32(Sb + (s2 * 32 + l31) * 128, kk * 4 + 2 * h, ksw);
                sc[s2] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qf[kk], kk == 0 ? zero : sc[s2], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            }
        if constexpr (MASK) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * KV_TILE + s2 * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                    if (key >= S) sc[s2][r] = -INFINITY;
                }
        }
        float mx = sc[0][0];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[s2][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * scale_log2);
        const bool moved = m_new != m_run;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
        i32x8f pk;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                float p[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    p[b] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[s2][4 * a + b], scale_log2, -m_new));
                    psum += p[b];
                }
                pk[s2 * 4 + a] = (int)pack4_e4m3(p[0], p[1], p[2], p[3]);      // k-slot bytes 16 s2 + 4 a + b of this lane's half
            }
        l_run = __builtin_fmaf(l_run, alpha, psum);
        if (__any(moved)) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const i32x8f vf = rd32(Sb + KV_TILE * 128 + (dt * 32 + l31) * 64, 2 * h, vsw);
            o[dt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf, pk, o[dt], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    auto sync = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    const int nt = t_end;
    const bool tail = (S & (KV_TILE - 1)) != 0 && nt == nt_all;
    const int nt_loop = tail ? nt - 1 : nt;
    stage(0, t_begin);
    int t = t_begin;
    for (; t + 2 <= nt_loop; t += 2) {
        sync();
        stage(1, t + 1);
        tile(t, B0{}, std::false_type{});
        sync();
        if (t + 2 < nt) stage(0, t + 2);
        tile(t + 1, B1{}, std::false_type{});
    }
    if (t < nt_loop) {
        sync();
        if (t + 1 < nt) stage(1, t + 1);
        tile(t, B0{}, std::false_type{});
        ++t;
    }
    if (tail) {
        sync();
        if ((t - t_begin) & 1) tile(t, B1{}, std::true_type{});
        else tile(t, B0{}, std::true_type{});
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (part_slot >= 0) {
        float* po = part_o + ((size_t)part_slot * Q_BLOCK + w * 32 + l31) * 128 + 4 * h;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = o[dt][4 * a + r];
                *(f32x4*)(po + dt * 32 + 8 * a) = v;
            }
        if (h == 0) {
            float* pm = part_ml + ((size_t)part_slot * Q_BLOCK + w * 32 + l31) * 2;
            pm[0] = m_run;
            pm[1] = l_tot;
        }
        return;
    }
    const float inv = 1.0f / l_tot;
    const int q = q0 + l31;
    if (q < S) {
        bf16* op = out + (size_t)q * ldo + head * 128 + 4 * h;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                bf16x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (bf16)((float)(bf16)(o[dt][4 * a + r] * inv) * v_std);     // x.to(bf16) * v_std
                *(bf16x4*)(op + dt * 32 + 8 * a) = v;
            }
    }
}


// Variant 1 (default): the same arithmetic in a SOFTWARE-PIPELINED, hand-interleaved order.  In the kernel above the two waves of a SIMD
// walk a tile in phase (barrier, 4 MFMAs of S, softmax, 4 MFMAs of P.V): while one issues MFMAs the other wants the matrix pipe too, so
// matrix time and softmax time ADD (measured: 2600 cycles per tile and SIMD = 1024 of MFMA + ~1600 of VALU).  Here iteration t issues
// S(t+1) = K(t+1) Q^T and O += P(t-1) V(t-1) -- eight MFMAs that do not depend on this iteration's softmax -- one per slice of the
// softmax of S(t), which the previous iteration left in registers; every (LDS reads, MFMA) and every VALU slice is fenced with
// sched_barrier(0), so source order is the schedule: each wave's own stream keeps the matrix pipe fed while it issues VALU.
// K is staged two tiles ahead and V one tile behind it (two 8 KiB rings each: the same 32 KiB), one barrier per tile.
// The lane halves meet in v_permlane32_swap (no LDS round trip); the row sum runs in two interleaved partial sums per lane, so l differs
// from variant 0's in the last bits.  No packed fp32 (v_pk_*: slower than two plain VALU beside MFMAs on gfx950; the file is compiled
// with -fno-slp-vectorize).
int g_attn_fp8_variant = 1;
constexpr float F8_TAU = 8.0f;
constexpr float F8_SUM_LIMIT = 448.0f;     // variant 2: a row's 64 P of one tile may sum to the largest e4m3 at most
#define F8_FENCE() __builtin_amdgcn_sched_barrier(0)

template <int NW, bool SUM>
__global__ void __launch_bounds__(NW * 64, 2)
flash_attn_fp8p_kernel(const uint8_t* __restrict__ Q8, const uint8_t* __restrict__ K8, const uint8_t* __restrict__ Vt8,
                       bf16* __restrict__ out, int S, int S_pad, int ldo, const float* __restrict__ stats, AttnPlan plan,
                       float* __restrict__ part_o, float* __restrict__ part_ml) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int Q_BLOCK = NW * 32;
    constexpr int RING = KV_TILE * 128;           // one K8 or Vt8 tile: 8 KiB
    static_assert(NW == 8, "one K8 piece and one Vt8 piece per wave");
    const int lane = lane_id();
    const int w = wave_id();
    const int l31 = lane & 31, h = lane >> 5;
    const float scale_log2 = stats[3], v_std = stats[2];
    const int nqb = plan.nqb;
    const int nt_all = (S + KV_TILE - 1) / KV_TILE;
    int item, t_begin = 0, t_end = nt_all, part_slot = -1;
    if ((int)blockIdx.x < plan.n_full) {
        item = xcd_remap((int)blockIdx.x, plan.n_full);
    } else {
        const int j = (int)blockIdx.x - plan.n_full;
        item = plan.n_full + j / plan.split;
        const int part = j - (j / plan.split) * plan.split;
        if (plan.split > 1) {
            t_begin = (int)((long long)nt_all * part / plan.split);
            t_end = (int)((long long)nt_all * (part + 1) / plan.split);
            part_slot = j;
        }
    }
    const int head = item / nqb;
    const int qb = item - head * nqb;
    const int q0 = qb * Q_BLOCK + w * 32;
    const uint8_t* Qh = Q8 + (size_t)head * S_pad * 128;
    const uint8_t* Kh = K8 + (size_t)head * S_pad * 128;
    const uint8_t* Vh = Vt8 + (size_t)head * 128 * S_pad;

    i32x8f qf[2];
    {
        const int qrow = min(q0 + l31, S - 1);
        const uint8_t* qp = Qh + (size_t)qrow * 128 + h * 32;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const i32x4f lo = *(const i32x4f*)(qp + kk * 64), hi = *(const i32x4f*)(qp + kk * 64 + 16);
            qf[kk] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }
    const uint8_t* k_src;
    const uint8_t* v_src;
    {
        const int krow = w * 8 + (lane >> 3);
        k_src = Kh + (size_t)krow * 128 + (((lane & 7) ^ (krow & 7)) << 4);
        const int vrow = w * 16 + (lane >> 2);
        v_src = Vh + (size_t)vrow * S_pad + (((lane & 3) ^ ((vrow >> 1) & 3)) << 4);
    }
    // LDS: K ring (2 x 8 KiB) then V ring (2 x 8 KiB)
    auto stage_k = [&](int buf, int t) { glds16(k_src + (size_t)t * KV_TILE * 128, smem + buf * RING + w * 1024); };
    auto stage_v = [&](int buf, int t) { glds16(v_src + t * KV_TILE, smem + (2 + buf) * RING + w * 1024); };
    f32x16 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int ksw = l31 & 7, vsw = (l31 >> 1) & 3;
    const int krow_off = l31 * 128, vrow_off = l31 * 64;

    auto rd32 = [&](const char* rowp, int c0, int sw) -> i32x8f {
        const i32x4f lo = *(const i32x4f*)(rowp + ((c0 ^ sw) << 4)), hi = *(const i32x4f*)(rowp + (((c0 + 1) ^ sw) << 4));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // matrix slot j of an iteration: 0..3 = S(t+1) pieces (s2 = j & 1, kk = j >> 1: the two accumulators alternate), 4..7 = P(t-1) V(t-1), dt = j - 4
    auto frag = [&](const char* Kb, const char* Vb, int j) -> i32x8f {
        return j < 4 ? rd32(Kb + (j & 1) * 32 * 128 + krow_off, (j >> 1) * 4 + 2 * h, ksw) : rd32(Vb + (j - 4) * 32 * 64 + vrow_off, 2 * h, vsw);
    };
    // every MFMA is a volatile asm statement: the compiler may neither sink it into the next iteration (its result is only read there) nor
    // re-order it against the pinned softmax slices.  Operands come from LDS reads (s_waitcnt is inserted by the compiler) or were written
    // hundreds of cycles earlier; readers of the results are >= 64 cycles downstream (hand-checked: the compiler does not see these hazards)
    int unit = 0x7f7f7f7f;
    asm volatile("" : "+v"(unit));
    // drain: the wait states a 16-pass MFMA needs before anything reads its result, INSIDE the asm statement -- the compiler is free to put
    // register copies of the result right behind the statement (it did: the two arms of a branch merged their accumulators that way)
#define F8_MFMA "v_mfma_scale_f32_32x32x64_f8f6f4 "
#define F8_DRAIN "\n\ts_nop 15\n\ts_nop 7"
    auto mm = [&](int j, const i32x8f& f, f32x16 (&sn)[2], const i32x8f& p, bool drain = false) {
        if (j < 2)
            asm volatile(F8_MFMA "%0, %1, %2, 0, %3, %3 op_sel_hi:[0,0,0]" : "=&v"(sn[j & 1]) : "v"(f), "v"(qf[0]), "v"(unit));
        else if (j < 4 && !drain)
            asm volatile(F8_MFMA "%0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(sn[j & 1]) : "v"(f), "v"(qf[1]), "v"(unit));
        else if (j < 4)
            asm volatile(F8_MFMA "%0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" F8_DRAIN : "+v"(sn[j & 1]) : "v"(f), "v"(qf[1]), "v"(unit));
        else if (!drain)
            asm volatile(F8_MFMA "%0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(o[j - 4]) : "v"(f), "v"(p), "v"(unit));
        else
            asm volatile(F8_MFMA "%0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" F8_DRAIN : "+v"(o[j - 4]) : "v"(f), "v"(p), "v"(unit));
    };
    auto sync = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    const int n = t_end - t_begin;
    const bool tail = (S & (KV_TILE - 1)) != 0 && t_end == nt_all;
    // iteration i (tile t = t_begin + i), parity P = i & 1: K(t+1) sits in K ring 1 - P, V(t-1) in V ring 1 - P; K(t+2) -> K ring P, V(t) -> V ring P.
    // sc: S(t) (in), sn: S(t+1) (out); pq: P(t-1) (in), pc: P(t) (out)
    auto iter = [&](int t, auto par_tag, auto prev_tag, auto last_tag, f32x16 (&sc)[2], f32x16 (&sn)[2], const i32x8f& pq, i32x8f& pc) {
        constexpr int P = decltype(par_tag)::value;
        constexpr bool PREV = decltype(prev_tag)::value, LAST = decltype(last_tag)::value;
        const char* Kb = smem + (1 - P) * RING;
        const char* Vb = smem + (2 + 1 - P) * RING;
        auto on = [](int j) { return j < 4 ? !LAST : PREV; };
        sync();
        if constexpr (!LAST) {
            if (t + 2 < t_end) stage_k(P, t + 2);
        }
        stage_v(P, t);
        i32x8f f[4];
        if (on(0)) {
            f[0] = frag(Kb, Vb, 0);
            f[1] = frag(Kb, Vb, 1);
        }
        F8_FENCE();
        // slot(j): issue MFMA j, then the LDS reads of slot j + 2's fragment (two VALU slices ahead of its use).  The MFMA is
        // asynchronous and the compiler does not know: its A / B registers are "dead" after the asm statement and would be handed to the
        // very next LDS read or VALU temporary while the matrix pipe still reads them (measured: 4 % rms error).  F8_KEEP holds every
        // fragment (and P) live until the NEXT MFMA of this wave has issued, i.e. until this one has left the pipe.
#define F8_KEEP(x) asm volatile("" ::"v"(x))
#define F8_SLOT(j)                                                   \
        F8_FENCE();                                                  \
        if (on(j)) mm(j, f[(j) & 3], sn, pq);                        \
        if ((j) >= 1 && on((j) - 1)) F8_KEEP(f[((j) - 1) & 3]);      \
        if ((j) + 2 < 8 && on((j) + 2)) f[((j) + 2) & 3] = frag(Kb, Vb, (j) + 2); \
        F8_FENCE();
        if constexpr (LAST) {
            if (tail) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = t * KV_TILE + s2 * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                        if (key >= S) sc[s2][r] = -INFINITY;
                    }
            }
            F8_FENCE();
        }
        if constexpr (SUM) {
            // Variant 2: no maximum on the fast path.  P = exp2(s c - m) against the reference m the row already has; a lane adds up its 32 P,
            // the two lanes of a row exchange their sums, and as long as the row's 64 P sum to <= 448 (the largest e4m3: then no single P
            // saturates) the tile is done -- 4 VALU per score, none of them waiting for a reduction.  Otherwise (always on a part's first
            // tile: m = -inf gives inf or NaN sums) the WAVE redoes the tile the long way: rows over the limit move m to the tile's maximum,
            // O and l are rescaled, every P is recomputed.  Per row the rule does not depend on which rows share a wave.
            const float nm0 = -m_run;
            float ps0 = 0.f, ps1 = 0.f;
            auto group = [&](int g, float nm) {            // scores 4 a .. 4 a + 3 of accumulator s2 -> dword g = 4 s2 + a of this lane's k-slots
                const int s2 = g >> 2, a = g & 3;
                const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[s2][4 * a], scale_log2, nm));
                const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[s2][4 * a + 1], scale_log2, nm));
                const float p2 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[s2][4 * a + 2], scale_log2, nm));
                const float p3 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[s2][4 * a + 3], scale_log2, nm));
                ps0 += p0;
                ps1 += p1;
                ps0 += p2;
                ps1 += p3;
                int v = pc[g];                           // both halves are overwritten: no zero fill
                v = __builtin_amdgcn_cvt_pk_fp8_f32(p0, p1, v, false);
                v = __builtin_amdgcn_cvt_pk_fp8_f32(p2, p3, v, true);
                pc[g] = v;
                asm volatile("" : "+v"(pc[g]));          // computed HERE (not sunk to its reader in the next iteration)
            };
            F8_SLOT(0)
            group(0, nm0);
            F8_SLOT(1)
            group(1, nm0);
            F8_SLOT(2)
            group(2, nm0);
            F8_SLOT(3)
            group(3, nm0);
            F8_SLOT(4)
            group(4, nm0);
            F8_SLOT(5)
            group(5, nm0);
            F8_SLOT(6)
            group(6, nm0);
            F8_SLOT(7)
            group(7, nm0);
            float ps = ps0 + ps1;
            const float row = sum_with_lane_xor32(ps);
            const bool over = !(row <= F8_SUM_LIMIT);          // NaN (a masked key against m = -inf) counts as over
            if (on(7)) {
                F8_KEEP(f[3]);
                F8_KEEP(pq);
            } else if (on(3)) {
                F8_KEEP(f[3]);                           // no MFMA behind slot 3 in the first iteration: fragment 3 lives to the tail
            }
            F8_FENCE();
            if (__any(over)) {
                float mx = sc[0][0];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[s2][r]);
                mx = max_with_lane_xor32(mx);
                const float m_new = over ? mx * scale_log2 : m_run;
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);      // rows that stay: exp2(0) = 1
                m_run = m_new;
                if (on(7)) asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");      // the last P.V MFMA of this iteration -> the pass over O
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
                l_run *= alpha;
                ps0 = ps1 = 0.f;
                const float nm1 = -m_new;
#pragma unroll
                for (int g = 0; g < 8; ++g) group(g, nm1);
                ps = ps0 + ps1;
            }
            l_run += ps;
            asm volatile("" : "+v"(l_run));
        } else {
            float mx0 = sc[0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx0 = fmaxf(mx0, sc[0][r]);
            asm volatile("" : "+v"(mx0));            // pins: pure arithmetic is otherwise placed wherever instruction selection likes
            F8_SLOT(0)
            float mx = sc[1][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[1][r]);
            mx = fmaxf(mx, mx0);
            asm volatile("" : "+v"(mx));
            F8_SLOT(1)
            mx = max_with_lane_xor32(mx);
            // lazily raised reference: m moves to the tile's maximum only when that exceeds it by more than 2^F8_TAU (the first tile always:
            // m = -inf); otherwise P = exp2(s c - m) <= 2^8 = 256 < 448 = the largest e4m3 and O, l keep their scale: no pass over O
            const float m_tile = mx * scale_log2;
            const bool moved = m_tile - m_run > F8_TAU;
            const float m_new = moved ? m_tile : m_run;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            const float nm = -m_new;
            float ps0 = 0.f, ps1 = 0.f;
            auto group = [&](int g) {            // scores 4 a .. 4 a + 3 of accumulator s2 -> dword g = 4 s2 + a of this lane's k-slots
                const int s2 = g >> 2, a = g & 3;
                const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[s2][4 * a], scale_log2, nm));
                const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[s2][4 * a + 1], scale_log2, nm));
                const float p2 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[s2][4 * a + 2], scale_log2, nm));
                const float p3 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[s2][4 * a + 3], scale_log2, nm));
                ps0 += p0;
                ps1 += p1;
                ps0 += p2;
                ps1 += p3;
                int v = pc[g];                           // both halves are overwritten: no zero fill
                v = __builtin_amdgcn_cvt_pk_fp8_f32(p0, p1, v, false);
                v = __builtin_amdgcn_cvt_pk_fp8_f32(p2, p3, v, true);
                pc[g] = v;
                asm volatile("" : "+v"(pc[g]));          // computed HERE (not sunk to its reader in the next iteration)
            };
            group(0);
            F8_SLOT(2)
            group(1);
            F8_SLOT(3)
            group(2);
            F8_SLOT(4)
            group(3);
            F8_SLOT(5)
            group(4);
            F8_SLOT(6)
            group(5);
            F8_SLOT(7)
            group(6);
            group(7);
            l_run = __builtin_fmaf(l_run, alpha, ps0 + ps1);
            asm volatile("" : "+v"(l_run));
            if (on(7)) {
                F8_KEEP(f[3]);
                F8_KEEP(pq);
            } else if (on(3)) {
                F8_KEEP(f[3]);                           // no MFMA behind slot 3 in the first iteration: fragment 3 lives to the tail
            }
            F8_FENCE();
            if (__any(moved)) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            }
        }
#undef F8_SLOT
#undef F8_KEEP
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using No = std::false_type;
    using Yes = std::true_type;
    f32x16 sa[2], sb[2];
    i32x8f pa, pb;
#pragma unroll
    for (int j = 0; j < 8; ++j) pa[j] = pb[j] = 0;
    if (n > 0) {                                     // a part of a split item may own no tile at all (more parts than tiles)
        stage_k(0, t_begin);
        sync();
        if (n > 1) stage_k(1, t_begin + 1);
        {                                                                         // S(t_begin)
            i32x8f f4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) f4[j] = frag(smem, smem, j);
#pragma unroll
            for (int j = 0; j < 4; ++j) mm(j, f4[j], sa, pa, j == 3);             // drained: the first iteration reads S
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(f4[j]));            // live while the matrix pipe reads them (see F8_KEEP)
        }
        // even iterations: S in sa -> sb, P(t) -> pa (P(t-1) in pb); odd iterations the other way round
        if (n == 1) {
            iter(t_begin, P0{}, No{}, Yes{}, sa, sb, pb, pa);
        } else {
            iter(t_begin, P0{}, No{}, No{}, sa, sb, pb, pa);
            int i = 1;
            for (; i + 2 <= n - 1; i += 2) {
                iter(t_begin + i, P1{}, Yes{}, No{}, sb, sa, pa, pb);
                iter(t_begin + i + 1, P0{}, Yes{}, No{}, sa, sb, pb, pa);
            }
            if (i < n - 1) {
                iter(t_begin + i, P1{}, Yes{}, No{}, sb, sa, pa, pb);
                ++i;
            }
            if (i & 1) iter(t_begin + i, P1{}, Yes{}, Yes{}, sb, sa, pa, pb);
            else iter(t_begin + i, P0{}, Yes{}, Yes{}, sa, sb, pb, pa);
        }
        i32x8f pl;                                       // P(t_end - 1), selected before the barrier (a VALU write needs distance to the MFMA)
#pragma unroll
        for (int j = 0; j < 8; ++j) pl[j] = ((n - 1) & 1) ? pb[j] : pa[j];
        asm volatile("" : "+v"(pl));
        sync();                                          // V(t_end - 1) landed
        {
            const char* Vb = smem + (2 + ((n - 1) & 1)) * RING;
            i32x8f f4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) f4[j] = frag(Vb, Vb, 4 + j);
#pragma unroll
            for (int j = 0; j < 4; ++j) mm(4 + j, f4[j], sa, pl, j == 3);         // drained: the epilogue reads O
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(f4[j]));
            asm volatile("" ::"v"(pl));
            asm volatile("" ::"v"(pa), "v"(pb), "v"(qf[0]), "v"(qf[1]), "v"(unit));      // nothing an MFMA reads is ever "dead" before the end
        }
    }
    const float l_tot = sum_with_lane_xor32(l_run);
    if (part_slot >= 0) {
        float* po = part_o + ((size_t)part_slot * Q_BLOCK + w * 32 + l31) * 128 + 4 * h;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = o[dt][4 * a + r];
                *(f32x4*)(po + dt * 32 + 8 * a) = v;
            }
        if (h == 0) {
            float* pm = part_ml + ((size_t)part_slot * Q_BLOCK + w * 32 + l31) * 2;
            pm[0] = m_run;
            pm[1] = l_tot;
        }
        return;
    }
    const float inv = 1.0f / l_tot;
    const int q = q0 + l31;
    if (q < S) {
        bf16* op = out + (size_t)q * ldo + head * 128 + 4 * h;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                bf16x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (bf16)((float)(bf16)(o[dt][4 * a + r] * inv) * v_std);     // x.to(bf16) * v_std
                *(bf16x4*)(op + dt * 32 + 8 * a) = v;
            }
    }
}

#define F8W_CLOB "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
// registers of matrix slot m (the assembler evaluates the index expressions): Q of row block m & 1, k half m >> 2; O of row block m & 1, channels 32 ((m >> 1) - 4) ..
#define F8W_QREG(m) "a[96+((" #m ")&1)*16+((" #m ")>>2)*8:96+((" #m ")&1)*16+((" #m ")>>2)*8+7]"
#define F8W_OREG(m) "a[128+((" #m ")&1)*64+(((" #m ")>>1)-4)*16:128+((" #m ")&1)*64+(((" #m ")>>1)-4)*16+15]"
#define F8W_OBASE(rb, dt) "128+" #rb "*64+" #dt "*16"
#define F8W_MM_(m, sn, f, p, TAIL)                                                                                                         \
    do {                                                                                                                                   \
        if constexpr ((m) < 4)                                                                                                             \
            asm volatile(F8_MFMA "%0, %1, " F8W_QREG((m) & 7) ", 0, %2, %2 op_sel_hi:[0,0,0]" TAIL : "=&v"(sn[(m) & 1][((m) >> 1) & 1]) : "v"(f), "v"(unit) : F8W_CLOB); \
        else if constexpr ((m) < 8)                                                                                                        \
            asm volatile(F8_MFMA "%0, %1, " F8W_QREG((m) & 7) ", %0, %2, %2 op_sel_hi:[0,0,0]" TAIL : "+v"(sn[(m) & 1][((m) >> 1) & 1]) : "v"(f), "v"(unit) : F8W_CLOB); \
        else                                                                                                                               \
            asm volatile(F8_MFMA F8W_OREG((m) | 8) ", %0, %1, " F8W_OREG((m) | 8) ", %2, %2 op_sel_hi:[0,0,0]" TAIL ::"v"(f), "v"(p[(m) & 1]), "v"(unit) : F8W_CLOB); \
    } while (0)
// row sums by the matrix pipe (variant 4): L of row block rb at a[64 + 16 rb ..] += ones . P
#define F8W_LREG(rb) "a[64+" #rb "*16:64+" #rb "*16+15]"
#define F8W_ML_(rb, p, TAIL) asm volatile(F8_MFMA F8W_LREG(rb) ", %0, %1, " F8W_LREG(rb) ", %2, %2 op_sel_hi:[0,0,0]" TAIL ::"v"(ones8), "v"(p[rb]), "v"(unit) : F8W_CLOB)
#define F8W_RESCALE1(B, al)                                                                                                                \
    do {                                                                                                                                   \
        float t0_;                                                                                                                         \
        asm volatile("v_accvgpr_read_b32 %0, a[" B "]\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a[" B "], %0" : "=&v"(t0_) : "v"(al) : F8W_CLOB); \
    } while (0)
#define F8W_MM(m, sn, f, p) F8W_MM_(m, sn, f, p, "")
#define F8W_MM_DRAIN(m, sn, f, p) F8W_MM_(m, sn, f, p, F8_DRAIN)
// O tuple -> 16 floats (after a drained MFMA)
#define F8W_RD(B, i) "v_accvgpr_read_b32 %" #i ", a[" B "+" #i "]\n\t"
#define F8W_READ16(B, d)                                                                                                                   \
    asm volatile(F8W_RD(B, 0) F8W_RD(B, 1) F8W_RD(B, 2) F8W_RD(B, 3) F8W_RD(B, 4) F8W_RD(B, 5) F8W_RD(B, 6) F8W_RD(B, 7) F8W_RD(B, 8)       \
                 F8W_RD(B, 9) F8W_RD(B, 10) F8W_RD(B, 11) F8W_RD(B, 12) F8W_RD(B, 13) F8W_RD(B, 14) F8W_RD(B, 15)                          \
                 : "=v"(d[0]), "=v"(d[1]), "=v"(d[2]), "=v"(d[3]), "=v"(d[4]), "=v"(d[5]), "=v"(d[6]), "=v"(d[7]), "=v"(d[8]), "=v"(d[9]),  \
                   "=v"(d[10]), "=v"(d[11]), "=v"(d[12]), "=v"(d[13]), "=v"(d[14]), "=v"(d[15])::F8W_CLOB)
// O tuple *= alpha (the rare pass over O): read, multiply, write back, four temporaries in turn
#define F8W_RS(B, i, t) "v_accvgpr_read_b32 %" #t ", a[" B "+" #i "]\n\tv_mul_f32 %" #t ", %" #t ", %4\n\tv_accvgpr_write_b32 a[" B "+" #i "], %" #t "\n\t"
#define F8W_RESCALE16(B, al)                                                                                                               \
    do {                                                                                                                                   \
        float t0_, t1_, t2_, t3_;                                                                                                          \
        asm volatile(F8W_RS(B, 0, 0) F8W_RS(B, 1, 1) F8W_RS(B, 2, 2) F8W_RS(B, 3, 3) F8W_RS(B, 4, 0) F8W_RS(B, 5, 1) F8W_RS(B, 6, 2)          \
                     F8W_RS(B, 7, 3) F8W_RS(B, 8, 0) F8W_RS(B, 9, 1) F8W_RS(B, 10, 2) F8W_RS(B, 11, 3) F8W_RS(B, 12, 0) F8W_RS(B, 13, 1)      \
                     F8W_RS(B, 14, 2) F8W_RS(B, 15, 3)                                                                                      \
                     : "=&v"(t0_), "=&v"(t1_), "=&v"(t2_), "=&v"(t3_) : "v"(al) : F8W_CLOB);                                                 \
    } while (0)
// 32 accumulator registers from a[B] on = 0
#define F8W_ZR(B, i) "v_accvgpr_write_b32 a[" B "+" #i "], 0\n\t"
#define F8W_ZR8(B, i) F8W_ZR(B, i) F8W_ZR(B, i + 1) F8W_ZR(B, i + 2) F8W_ZR(B, i + 3) F8W_ZR(B, i + 4) F8W_ZR(B, i + 5) F8W_ZR(B, i + 6) F8W_ZR(B, i + 7)
#define F8W_ZR32(B) F8W_ZR8(B, 0) F8W_ZR8(B, 8) F8W_ZR8(B, 16) F8W_ZR8(B, 24)
#define F8W_QW(B, i) "v_accvgpr_write_b32 a[" B "+" #i "], %" #i "\n\t"
#define F8W_QWRITE(B, q)                                                                                                                   \
    asm volatile(F8W_QW(B, 0) F8W_QW(B, 1) F8W_QW(B, 2) F8W_QW(B, 3) F8W_QW(B, 4) F8W_QW(B, 5) F8W_QW(B, 6) F8W_QW(B, 7)                     \
                 ::"v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7]) : F8W_CLOB)

// Variant 3: variant 1's arithmetic, value for value, with ONE wave per SIMD: 4 waves x 64 query rows (two 32-row blocks rb) instead of
// 8 x 32.  In variant 1 the two waves of a SIMD share its VALU port by age (the older wave finishes its tile in ~1550 cycles and waits
// ~1100 in the barrier, profiles/r04_attention_notes.md); one wave with twice the rows has the port to itself, every K8 / Vt8 fragment
// it reads from LDS feeds two MFMAs (half the ds_read_b128 per score), and a tile's barrier is met by 4 waves.  The registers of two
// waves in one: O (64 rows x 128: 128 registers) and Q (32) live in the accumulator half, in FIXED registers a[128:255] / a[96:127] that
// only the asm statements below name (every one of them lists a96 ... a255 as clobbered, so the compiler keeps nothing of its own there;
// a C++ value with an "a" constraint is copied to architectural registers at every basic-block boundary: 128 v_accvgpr_read per tile
// in the ISA of the first build); S(t), S(t+1), P(t-1), P(t) and four fragments in the 256 architectural registers.
// tools/mfma_asm_hazards.py checks that nothing outside the asm statements touches a[96:255].  16 matrix slots per iteration
// (8 of S(t+1), 8 of P(t-1) V(t-1)), one softmax chunk of S(t) behind each.  Per row the operations and their order are variant 1's
// (the two partial row sums per lane included): outputs are bit-identical with it.
// LSUM (variant 4): the row sums come from the matrix pipe -- two more MFMAs per tile, L += ones . P on the e4m3 P the numerator multiplies,
// in a third pair of accumulator tuples -- instead of 64 v_add_f32 per lane and tile: l is then the sum of the QUANTISED P (the oracle
// restates it: row_sum_quantised=True).  Not the published FlashAttention-3 form (fp32 sums of the unquantised P): opt-in.
template <bool LSUM>
__global__ void __launch_bounds__(256, 1)
flash_attn_fp8w_kernel(const uint8_t* __restrict__ Q8, const uint8_t* __restrict__ K8, const uint8_t* __restrict__ Vt8,
                       bf16* __restrict__ out, int S, int S_pad, int ldo, const float* __restrict__ stats, AttnPlan plan,
                       float* __restrict__ part_o, float* __restrict__ part_ml) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int Q_BLOCK = 256;
    constexpr int RING = KV_TILE * 128;           // one K8 or Vt8 tile: 8 KiB
    const int lane = lane_id();
    const int w = wave_id();
    const int l31 = lane & 31, h = lane >> 5;
    const float scale_log2 = stats[3], v_std = stats[2];
    const int nqb = plan.nqb;
    const int nt_all = (S + KV_TILE - 1) / KV_TILE;
    int item, t_begin = 0, t_end = nt_all, part_slot = -1;
    if ((int)blockIdx.x < plan.n_full) {
        item = xcd_remap((int)blockIdx.x, plan.n_full);
    } else {
        const int j = (int)blockIdx.x - plan.n_full;
        item = plan.n_full + j / plan.split;
        const int part = j - (j / plan.split) * plan.split;
        if (plan.split > 1) {
            t_begin = (int)((long long)nt_all * part / plan.split);
            t_end = (int)((long long)nt_all * (part + 1) / plan.split);
            part_slot = j;
        }
    }
    const int head = item / nqb;
    const int qb = item - head * nqb;
    const int q0 = qb * Q_BLOCK + w * 64;
    const uint8_t* Qh = Q8 + (size_t)head * S_pad * 128;
    const uint8_t* Kh = K8 + (size_t)head * S_pad * 128;
    const uint8_t* Vh = Vt8 + (size_t)head * 128 * S_pad;

    // Q: [rb][kk] at a[96 + 16 rb + 8 kk ..]
    {
        i32x8f qv[2][2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int qrow = min(q0 + rb * 32 + l31, S - 1);
            const uint8_t* qp = Qh + (size_t)qrow * 128 + h * 32;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const i32x4f lo = *(const i32x4f*)(qp + kk * 64), hi = *(const i32x4f*)(qp + kk * 64 + 16);
                qv[rb][kk] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
        F8W_QWRITE("96", qv[0][0]);
        F8W_QWRITE("96+8", qv[0][1]);
        F8W_QWRITE("96+16", qv[1][0]);
        F8W_QWRITE("96+24", qv[1][1]);
    }
    // a wave stages two 1 KiB pieces of K8 (8 keys each, 32 keys apart) and two of Vt8 (16 channels each, 64 apart): the swizzles only
    // look at the low row bits, which the two pieces share
    const uint8_t* k_src;
    const uint8_t* v_src;
    {
        const int krow = w * 8 + (lane >> 3);
        k_src = Kh + (size_t)krow * 128 + (((lane & 7) ^ (krow & 7)) << 4);
        const int vrow = w * 16 + (lane >> 2);
        v_src = Vh + (size_t)vrow * S_pad + (((lane & 3) ^ ((vrow >> 1) & 3)) << 4);
    }
    const size_t v_piece = (size_t)64 * S_pad;
    auto stage_k = [&](int buf, int t) {
        glds16(k_src + (size_t)t * KV_TILE * 128, smem + buf * RING + w * 1024);
        glds16(k_src + (size_t)t * KV_TILE * 128 + 32 * 128, smem + buf * RING + (w + 4) * 1024);
    };
    auto stage_v = [&](int buf, int t) {
        glds16(v_src + t * KV_TILE, smem + (2 + buf) * RING + w * 1024);
        glds16(v_src + v_piece + t * KV_TILE, smem + (2 + buf) * RING + (w + 4) * 1024);
    };
    asm volatile(F8W_ZR32("128") F8W_ZR32("160") F8W_ZR32("192") F8W_ZR32("224")::: F8W_CLOB);      // O = 0
    i32x8f ones8;
    if constexpr (LSUM) {
        asm volatile(F8W_ZR32("64")::: F8W_CLOB);                                      // L = 0 (a[64:95])
#pragma unroll
        for (int j = 0; j < 8; ++j) ones8[j] = 0x38383838;                             // 1.0 as e4m3, 32 k-slots per lane
        asm volatile("" : "+v"(ones8));
    }
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    const int ksw = l31 & 7, vsw = (l31 >> 1) & 3;
    const int krow_off = l31 * 128, vrow_off = l31 * 64;

    auto rd32 = [&](const char* rowp, int c0, int sw) -> i32x8f {
        const i32x4f lo = *(const i32x4f*)(rowp + ((c0 ^ sw) << 4)), hi = *(const i32x4f*)(rowp + (((c0 + 1) ^ sw) << 4));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    // fragment fr of an iteration: 0..3 = K8 (s2 = fr & 1, kk = fr >> 1), 4..7 = Vt8 (dt = fr - 4); matrix slot m uses fragment m >> 1 for row block m & 1
    auto frag = [&](const char* Kb, const char* Vb, int fr) -> i32x8f {
        return fr < 4 ? rd32(Kb + (fr & 1) * 32 * 128 + krow_off, (fr >> 1) * 4 + 2 * h, ksw) : rd32(Vb + (fr - 4) * 32 * 64 + vrow_off, 2 * h, vsw);
    };
    int unit = 0x7f7f7f7f;
    asm volatile("" : "+v"(unit));
    // the asm-MFMA rules of variant 1 (above): volatile statements, operands kept live until the next MFMA has issued, the wait states of
    // a result that is read right away inside the statement (F8W_MM / F8W_MM_DRAIN)
    auto sync = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    const int n = t_end - t_begin;
    const bool tail = (S & (KV_TILE - 1)) != 0 && t_end == nt_all;
    // iteration i (tile t = t_begin + i), parity P = i & 1: K(t+1) sits in K ring 1 - P, V(t-1) in V ring 1 - P; K(t+2) -> K ring P, V(t) -> V ring P.
    // sc: S(t) (in), sn: S(t+1) (out); pq: P(t-1) (in), pc: P(t) (out)
    auto iter = [&](int t, auto par_tag, auto prev_tag, auto last_tag, f32x16 (&sc)[2][2], f32x16 (&sn)[2][2], const i32x8f (&pq)[2],
                    i32x8f (&pc)[2]) {
        constexpr int P = decltype(par_tag)::value;
        constexpr bool PREV = decltype(prev_tag)::value, LAST = decltype(last_tag)::value;
        const char* Kb = smem + (1 - P) * RING;
        const char* Vb = smem + (2 + 1 - P) * RING;
        auto on = [](int m) { return m < 8 ? !LAST : PREV; };
        sync();
        if constexpr (!LAST) {
            if (t + 2 < t_end) stage_k(P, t + 2);
        }
        stage_v(P, t);
        i32x8f f[4];
        if (on(0)) {
            f[0] = frag(Kb, Vb, 0);
            f[1] = frag(Kb, Vb, 1);
        } else if (on(8)) {                          // the last iteration has no S MFMAs: its first fragments are Vt8's
            f[0] = frag(Kb, Vb, 4);
            f[1] = frag(Kb, Vb, 5);
        }
        F8_FENCE();
#define F8_KEEP(x) asm volatile("" ::"v"(x))
        // slot m: MFMA m (fragment m >> 1, row block m & 1); behind the second MFMA of a fragment the LDS reads of the fragment two ahead
        // (the first two of an iteration are read in front of slot 0)
#define F8W_SLOT(m)                                                                   \
        F8_FENCE();                                                                   \
        if (on(m)) F8W_MM(m, sn, f[((m) >> 1) & 3], pq);                              \
        if ((m) >= 1 && on((m) - 1)) F8_KEEP(f[(((m) - 1) >> 1) & 3]);                \
        if (((m) & 1) && (m) + 3 < 16 && on(m) && on((m) + 3)) f[(((m) + 3) >> 1) & 3] = frag(Kb, Vb, ((m) + 3) >> 1); \
        F8_FENCE();
        if constexpr (LAST) {
            if (tail) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = t * KV_TILE + s2 * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                            if (key >= S) sc[rb][s2][r] = -INFINITY;
                        }
            }
            F8_FENCE();
        }
        float alpha[2], nm[2], ps0[2] = {0.f, 0.f}, ps1[2] = {0.f, 0.f};
        bool moved[2];
        auto max_a = [&](int rb) -> float {
            float mx0 = sc[rb][0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx0 = fmaxf(mx0, sc[rb][0][r]);
            asm volatile("" : "+v"(mx0));            // pins: pure arithmetic is otherwise placed wherever instruction selection likes
            return mx0;
        };
        auto max_b = [&](int rb, float mx0) -> float {
            float mx = sc[rb][1][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[rb][1][r]);
            mx = fmaxf(mx, mx0);
            asm volatile("" : "+v"(mx));
            return mx;
        };
        auto reference = [&](int rb, float mx) {     // lazily raised reference (variant 1): m moves only when the tile's maximum is 2^F8_TAU above it
            mx = max_with_lane_xor32(mx);
            const float m_tile = mx * scale_log2;
            moved[rb] = m_tile - m_run[rb] > F8_TAU;
            const float m_new = moved[rb] ? m_tile : m_run[rb];
            alpha[rb] = __builtin_amdgcn_exp2f(m_run[rb] - m_new);
            m_run[rb] = m_new;
            nm[rb] = -m_new;
        };
        auto group = [&](int rb, int g) {            // scores 4 a .. 4 a + 3 of accumulator s2 -> dword g = 4 s2 + a of this lane's k-slots
            const int s2 = g >> 2, a = g & 3;
            const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[rb][s2][4 * a], scale_log2, nm[rb]));
            const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[rb][s2][4 * a + 1], scale_log2, nm[rb]));
            const float p2 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[rb][s2][4 * a + 2], scale_log2, nm[rb]));
            const float p3 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[rb][s2][4 * a + 3], scale_log2, nm[rb]));
            if constexpr (!LSUM) {
                ps0[rb] += p0;
                ps1[rb] += p1;
                ps0[rb] += p2;
                ps1[rb] += p3;
            }
            int v = pc[rb][g];                       // both halves are overwritten: no zero fill
            v = __builtin_amdgcn_cvt_pk_fp8_f32(p0, p1, v, false);
            v = __builtin_amdgcn_cvt_pk_fp8_f32(p2, p3, v, true);
            pc[rb][g] = v;
            asm volatile("" : "+v"(pc[rb][g]));      // computed HERE (not sunk to its reader in the next iteration)
        };
        auto row_sum = [&](int rb) {
            if constexpr (!LSUM) {
                l_run[rb] = __builtin_fmaf(l_run[rb], alpha[rb], ps0[rb] + ps1[rb]);
                asm volatile("" : "+v"(l_run[rb]));
            }
        };
        // the softmax of S(t) needs nothing of this iteration's MFMAs: its first chunks run while the first fragment reads are in flight
        float mx = max_a(0);
        mx = max_b(0, mx);
        F8W_SLOT(0)
        reference(0, mx);
        group(0, 0);
        F8W_SLOT(1)
        mx = max_a(1);
        F8W_SLOT(2)
        mx = max_b(1, mx);
        F8W_SLOT(3)
        reference(1, mx);
        group(0, 1);
        F8W_SLOT(4)
        group(0, 2);
        F8W_SLOT(5)
        group(0, 3);
        F8W_SLOT(6)
        group(0, 4);
        F8W_SLOT(7)
        group(0, 5);
        F8W_SLOT(8)
        group(0, 6);
        F8W_SLOT(9)
        group(0, 7);
        row_sum(0);
        F8W_SLOT(10)
        group(1, 0);
        F8W_SLOT(11)
        group(1, 1);
        F8W_SLOT(12)
        group(1, 2);
        F8W_SLOT(13)
        group(1, 3);
        F8W_SLOT(14)
        group(1, 4);
        F8W_SLOT(15)
        group(1, 5);
        if constexpr (LSUM) {
            F8_FENCE();
            if (on(15)) {
                F8W_ML_(0, pq, "");
                F8_KEEP(f[3]);
            }
            F8_FENCE();
        }
        group(1, 6);
        if constexpr (LSUM) {
            F8_FENCE();
            if (on(15)) F8W_ML_(1, pq, "");
            F8_FENCE();
        }
        group(1, 7);
        row_sum(1);
        if (on(15)) {
            F8_KEEP(f[3]);
            F8_KEEP(pq[0]);
            F8_KEEP(pq[1]);
            if constexpr (LSUM) F8_KEEP(ones8);
        } else if (on(7)) {
            F8_KEEP(f[3]);                           // no MFMA behind slot 7 in the first iteration: fragment 3 lives to the tail
        }
        F8_FENCE();
        // the pass over O (rare: the reference moved).  The last P.V MFMA is four softmax groups (> 300 cycles) upstream
        if (__any(moved[0])) {
            F8W_RESCALE16(F8W_OBASE(0, 0), alpha[0]);
            F8W_RESCALE16(F8W_OBASE(0, 1), alpha[0]);
            F8W_RESCALE16(F8W_OBASE(0, 2), alpha[0]);
            F8W_RESCALE16(F8W_OBASE(0, 3), alpha[0]);
            if constexpr (LSUM) F8W_RESCALE1("64", alpha[0]);
        }
        if (__any(moved[1])) {
            F8W_RESCALE16(F8W_OBASE(1, 0), alpha[1]);
            F8W_RESCALE16(F8W_OBASE(1, 1), alpha[1]);
            F8W_RESCALE16(F8W_OBASE(1, 2), alpha[1]);
            F8W_RESCALE16(F8W_OBASE(1, 3), alpha[1]);
            if constexpr (LSUM) F8W_RESCALE1("64+16", alpha[1]);
        }
#undef F8W_SLOT
#undef F8_KEEP
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using No = std::false_type;
    using Yes = std::true_type;
    f32x16 sa[2][2], sb[2][2];
    i32x8f pa[2], pb[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int j = 0; j < 8; ++j) pa[rb][j] = pb[rb][j] = 0;
    if (n > 0) {                                     // a part of a split item may own no tile at all (more parts than tiles)
        stage_k(0, t_begin);
        sync();
        if (n > 1) stage_k(1, t_begin + 1);
        {                                                                         // S(t_begin)
            i32x8f f4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) f4[j] = frag(smem, smem, j);
            F8W_MM(0, sa, f4[0], pa);
            F8W_MM(1, sa, f4[0], pa);
            F8W_MM(2, sa, f4[1], pa);
            F8W_MM(3, sa, f4[1], pa);
            F8W_MM_DRAIN(4, sa, f4[2], pa);                                       // drained: the first iteration reads S
            F8W_MM_DRAIN(5, sa, f4[2], pa);
            F8W_MM_DRAIN(6, sa, f4[3], pa);
            F8W_MM_DRAIN(7, sa, f4[3], pa);
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(f4[j]));            // live while the matrix pipe reads them
        }
        // even iterations: S in sa -> sb, P(t) -> pa (P(t-1) in pb); odd iterations the other way round
        if (n == 1) {
            iter(t_begin, P0{}, No{}, Yes{}, sa, sb, pb, pa);
        } else {
            iter(t_begin, P0{}, No{}, No{}, sa, sb, pb, pa);
            int i = 1;
            for (; i + 2 <= n - 1; i += 2) {
                iter(t_begin + i, P1{}, Yes{}, No{}, sb, sa, pa, pb);
                iter(t_begin + i + 1, P0{}, Yes{}, No{}, sa, sb, pb, pa);
            }
            if (i < n - 1) {
                iter(t_begin + i, P1{}, Yes{}, No{}, sb, sa, pa, pb);
                ++i;
            }
            if (i & 1) iter(t_begin + i, P1{}, Yes{}, Yes{}, sb, sa, pa, pb);
            else iter(t_begin + i, P0{}, Yes{}, Yes{}, sa, sb, pb, pa);
        }
        i32x8f pl[2];                                    // P(t_end - 1), selected before the barrier (a VALU write needs distance to the MFMA)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
            for (int j = 0; j < 8; ++j) pl[rb][j] = ((n - 1) & 1) ? pb[rb][j] : pa[rb][j];
            asm volatile("" : "+v"(pl[rb]));
        }
        sync();                                          // V(t_end - 1) landed
        {
            const char* Vb = smem + (2 + ((n - 1) & 1)) * RING;
            i32x8f f4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) f4[j] = frag(Vb, Vb, 4 + j);
            if constexpr (LSUM) {
                F8W_ML_(0, pl, "");
                F8W_ML_(1, pl, "");
            }
            F8W_MM(8, sa, f4[0], pl);
            F8W_MM(9, sa, f4[0], pl);
            F8W_MM(10, sa, f4[1], pl);
            F8W_MM(11, sa, f4[1], pl);
            F8W_MM(12, sa, f4[2], pl);
            F8W_MM(13, sa, f4[2], pl);
            F8W_MM(14, sa, f4[3], pl);
            F8W_MM_DRAIN(15, sa, f4[3], pl);                                      // drained (the pipe is in order): the epilogue reads O
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(f4[j]));
            asm volatile("" ::"v"(pl[0]), "v"(pl[1]));
            if constexpr (LSUM) asm volatile("" ::"v"(ones8));
            asm volatile("" ::"v"(pa[0]), "v"(pa[1]), "v"(pb[0]), "v"(pb[1]), "v"(unit));      // nothing an MFMA reads is ever "dead" before the end
        }
    }
    float ov[2][4][16];
    F8W_READ16(F8W_OBASE(0, 0), ov[0][0]);
    F8W_READ16(F8W_OBASE(0, 1), ov[0][1]);
    F8W_READ16(F8W_OBASE(0, 2), ov[0][2]);
    F8W_READ16(F8W_OBASE(0, 3), ov[0][3]);
    F8W_READ16(F8W_OBASE(1, 0), ov[1][0]);
    F8W_READ16(F8W_OBASE(1, 1), ov[1][1]);
    F8W_READ16(F8W_OBASE(1, 2), ov[1][2]);
    F8W_READ16(F8W_OBASE(1, 3), ov[1][3]);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        float l_tot;
        if constexpr (LSUM) {            // every register of L's tuple holds the whole row's sum (all 64 k-slots): the first one
            if (rb == 0) asm volatile("v_accvgpr_read_b32 %0, a[64]" : "=v"(l_tot)::F8W_CLOB);
            else asm volatile("v_accvgpr_read_b32 %0, a[64+16]" : "=v"(l_tot)::F8W_CLOB);
        } else {
            l_tot = sum_with_lane_xor32(l_run[rb]);
        }
        const int row = w * 64 + rb * 32 + l31;
        if (part_slot >= 0) {
            float* po = part_o + ((size_t)part_slot * Q_BLOCK + row) * 128 + 4 * h;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = ov[rb][dt][4 * a + r];
                    *(f32x4*)(po + dt * 32 + 8 * a) = v;
                }
            if (h == 0) {
                float* pm = part_ml + ((size_t)part_slot * Q_BLOCK + row) * 2;
                pm[0] = m_run[rb];
                pm[1] = l_tot;
            }
            continue;
        }
        const float inv = 1.0f / l_tot;
        const int q = qb * Q_BLOCK + row;
        if (q < S) {
            bf16* op = out + (size_t)q * ldo + head * 128 + 4 * h;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    bf16x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (bf16)((float)(bf16)(ov[rb][dt][4 * a + r] * inv) * v_std);     // x.to(bf16) * v_std
                    *(bf16x4*)(op + dt * 32 + 8 * a) = v;
                }
        }
    }
}

size_t flash_attn_fp8_scratch_bytes(int H, int S_pad) {
    return 3 * (size_t)H * S_pad * 128 + 256 + (size_t)3 * F8_STAT_WGS * 2 * sizeof(double) + 256;
}

int launch_flash_attn_fp8(const void* q, const void* k, const void* vt, void* out, int H, int S, int S_pad, int ldo, void* scratch,
                          size_t scratch_bytes, void* workspace, size_t workspace_bytes, hipStream_t stream, int S_q, const double* qkv_stats) {
    PE_REQUIRE(q && k && vt && out && scratch, "flash_attn_fp8: null pointer");
    PE_REQUIRE(S_q >= 0 && S_q <= S, "flash_attn_fp8: S_q=%d query rows of S=%d", S_q, S);
    if (S_q == 0) S_q = S;       // (the three standard deviations are always those of the WHOLE tensors: torch.std over [1,H,S,128])
    PE_REQUIRE(H > 0 && S > 1 && S_pad % KV_TILE == 0 && S_pad >= S, "flash_attn_fp8: bad shape (H=%d S=%d S_pad=%d)", H, S, S_pad);
    PE_REQUIRE(ldo % 4 == 0 && ldo >= H * 128, "flash_attn_fp8: bad ldo=%d", ldo);
    PE_REQUIRE(scratch_bytes >= flash_attn_fp8_scratch_bytes(H, S_pad) && ((uintptr_t)scratch & 255) == 0,
               "flash_attn_fp8: scratch of %zu bytes, 256-byte aligned, needed", flash_attn_fp8_scratch_bytes(H, S_pad));
    static std::atomic<bool> configured{false};
    if (!configured.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void*)flash_attn_fp8_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, F8_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)flash_attn_fp8p_kernel<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, F8_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)flash_attn_fp8p_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, F8_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)flash_attn_fp8w_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, F8_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)flash_attn_fp8w_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, F8_LDS);
        if (e != hipSuccess) return set_error(PE_ERR_HIP, "flash_attn_fp8: hipFuncSetAttribute: %s", hipGetErrorString(e));
        configured.store(true, std::memory_order_release);
    }
    const size_t plane = (size_t)H * S_pad * 128;
    uint8_t* q8 = (uint8_t*)scratch;
    uint8_t* k8 = q8 + plane;
    uint8_t* vt8 = k8 + plane;
    float* stats = (float*)(vt8 + plane);
    double* part = (double*)((char*)stats + 256);
    const int slot = prof_begin(PROF_ATTN, 4.0 * (double)S_q * S * 128.0 * H, stream);
    if (qkv_stats != nullptr) {      // the QKV epilogue left the sums (EPI_QKV_STATS): 2 slots x row blocks x heads entries per section
        hipLaunchKernelGGL(attn_fp8_stats_finish_parts_kernel, dim3(1), dim3(1024), 0, stream, qkv_stats, 2 * qkv_stats_row_blocks(S_pad) * H,
                           (double)H * S * 128.0, stats);
    } else {
        hipLaunchKernelGGL(attn_fp8_stats_kernel, dim3(F8_STAT_WGS, 3), dim3(256), 0, stream, (const bf16*)q, (const bf16*)k, (const bf16*)vt,
                           H, S, S_pad, part);
        hipLaunchKernelGGL(attn_fp8_stats_finish_kernel, dim3(1), dim3(192), 0, stream, (const double*)part, (double)H * S * 128.0, stats);
    }
    hipLaunchKernelGGL(attn_fp8_quant_kernel, dim3(1024, 3), dim3(256), 0, stream, (const bf16*)q, (const bf16*)k, (const bf16*)vt, q8, k8,
                       vt8, H, S_pad, (const float*)stats);
    int rc = check_launch("attn_fp8_quant_kernel");
    if (rc != PE_OK) { prof_end(slot, stream); return rc; }
    const bool have_ws = workspace != nullptr && workspace_bytes >= flash_attn_workspace_bytes(H, S) && ((uintptr_t)workspace & 15) == 0;
    constexpr int Q_BLOCK = 256;
    const AttnPlan plan = make_plan(H, S, have_ws, Q_BLOCK, g_attn_slots, S_q);
    const int total = H * plan.nqb;
    const int n_short = (total - plan.n_full) * plan.split;
    float* part_o = (float*)workspace;
    float* part_ml = part_o ? part_o + (size_t)g_attn_slots * 256 * 128 : nullptr;
    if (g_attn_fp8_variant == 0)
        hipLaunchKernelGGL((flash_attn_fp8_kernel<8>), dim3(plan.n_full + (plan.split > 1 ? n_short : 0)), dim3(512), F8_LDS, stream,
                           (const uint8_t*)q8, (const uint8_t*)k8, (const uint8_t*)vt8, (bf16*)out, S, S_pad, ldo, (const float*)stats, plan,
                           part_o, part_ml);
    else if (g_attn_fp8_variant == 1)
        hipLaunchKernelGGL((flash_attn_fp8p_kernel<8, false>), dim3(plan.n_full + (plan.split > 1 ? n_short : 0)), dim3(512), F8_LDS, stream,
                           (const uint8_t*)q8, (const uint8_t*)k8, (const uint8_t*)vt8, (bf16*)out, S, S_pad, ldo, (const float*)stats, plan,
                           part_o, part_ml);
    else if (g_attn_fp8_variant == 2)
        hipLaunchKernelGGL((flash_attn_fp8p_kernel<8, true>), dim3(plan.n_full + (plan.split > 1 ? n_short : 0)), dim3(512), F8_LDS, stream,
                           (const uint8_t*)q8, (const uint8_t*)k8, (const uint8_t*)vt8, (bf16*)out, S, S_pad, ldo, (const float*)stats, plan,
                           part_o, part_ml);
    else if (g_attn_fp8_variant == 3)
        hipLaunchKernelGGL(flash_attn_fp8w_kernel<false>, dim3(plan.n_full + (plan.split > 1 ? n_short : 0)), dim3(256), F8_LDS, stream,
                           (const uint8_t*)q8, (const uint8_t*)k8, (const uint8_t*)vt8, (bf16*)out, S, S_pad, ldo, (const float*)stats, plan,
                           part_o, part_ml);
    else
        hipLaunchKernelGGL(flash_attn_fp8w_kernel<true>, dim3(plan.n_full + (plan.split > 1 ? n_short : 0)), dim3(256), F8_LDS, stream,
                           (const uint8_t*)q8, (const uint8_t*)k8, (const uint8_t*)vt8, (bf16*)out, S, S_pad, ldo, (const float*)stats, plan,
                           part_o, part_ml);
    rc = check_launch("flash_attn_fp8_kernel");
    if (rc == PE_OK && plan.split > 1) {
        hipLaunchKernelGGL(attn_combine_kernel, dim3((total - plan.n_full) * (Q_BLOCK / 8)), dim3(256), 0, stream, part_o, part_ml, (bf16*)out,
                           S, ldo, plan, Q_BLOCK, (const float*)(stats + 2));
        rc = check_launch("attn_combine_kernel");
    }
    prof_end(slot, stream);
    return rc;
}

}  // namespace pe
