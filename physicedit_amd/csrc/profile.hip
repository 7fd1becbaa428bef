// In-library kernel timing with HIP events recorded on the launch stream (what bench.py's
// `roofline` block is computed from).  When enabled, every `sample_every`-th launch of a kernel class
// is bracketed by an event pair from a fixed pool; pe_profile_read() synchronises those events and
// returns, per class, the number of sampled launches, their summed duration and the summed
// ALGORITHMIC work (FLOPs for MFMA kernels, bytes for HBM-bound ones) the launcher declared.
#include <vector>

#include "../../include/physicedit_amd.h"
#include "common.h"
#include "kernels.h"

namespace pe {

struct ProfSlot {
    hipEvent_t a, b;
    int kind;
    double work;
};

static struct {
    bool on = false;
    int sample_every = 1;
    std::vector<ProfSlot> pool;
    size_t used = 0;
    unsigned long long seen[PROF_KINDS] = {0};
} g_prof;

int prof_begin(int kind, double work, hipStream_t stream) {
    if (!g_prof.on) return -1;
    const unsigned long long n = g_prof.seen[kind]++;
    if (n % (unsigned long long)g_prof.sample_every != 0) return -1;
    if (g_prof.used >= g_prof.pool.size()) return -1;
    ProfSlot& s = g_prof.pool[g_prof.used];
    s.kind = kind;
    s.work = work;
    if (hipEventRecord(s.a, stream) != hipSuccess) return -1;
    return (int)g_prof.used++;
}

void prof_end(int slot, hipStream_t stream) {
    if (slot >= 0) (void)hipEventRecord(g_prof.pool[slot].b, stream);
}

// ------------------------------------------------------------------------------------------------
// Matrix-pipe probe: nothing but v_mfma_f32_16x16x32_bf16 -- the shape the GEMM runs on since round 5 (16 independent accumulators per wave,
// 8 waves per work-group, operands read once from `frags`, no LDS / memory access inside the loop).  On all-zero fragments it runs at the
// nominal dense peak (2.4 GHz); on random fragments the chip's power limit lowers the clock, and the rate it reports is the ceiling any bf16
// MFMA kernel has on such data before a byte is moved (profiles/r02_gemm_notes.md section 5; the 32x32x16 shape of rounds 1 - 4 sustains
// 14 - 16 % less: profiles/r05_gemm_notes.md section 7, tools/microbench/mfma_shape_power.hip measures both).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) mfma_probe_kernel(const uint4* __restrict__ frags, float* __restrict__ out, int iters) {
    typedef __attribute__((ext_vector_type(8))) __bf16 probe_bf16x8;
    typedef __attribute__((ext_vector_type(4))) float probe_f32x4;
    const int tid = threadIdx.x + blockIdx.x * 512;
    uint4 av[4], bv[4];
    for (int i = 0; i < 4; ++i) {
        av[i] = frags[(tid * 8 + i) & 0xfffff];
        bv[i] = frags[(tid * 8 + 4 + i) & 0xfffff];
    }
    probe_f32x4 acc[16];
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {      // 64 MFMAs x 16 KiFLOP = the 32 x 32 KiFLOP an iteration had in the 32x32x16 form
#pragma unroll
            for (int i = 0; i < 16; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(probe_bf16x8, av[(i + k) & 3]),
                                                                 __builtin_bit_cast(probe_bf16x8, bv[(i >> 2) & 3]), acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j];
    out[tid] = s;
}

// ------------------------------------------------------------------------------------------------
// "Attainable" probe: the GEMM main loop's per-MFMA resource mix without its barriers, its dependencies on arriving data and
// its epilogue.  8 waves per work-group (two per SIMD), all 160 KiB of LDS, per "K tile" and wave 64 MFMAs (v_mfma_f32_16x16x32_bf16) fed by
//   MODE 1: + 24 ds_read_b128 fragment reads (0.375 per 4-pass MFMA: the 64 x 128 wave tile's ratio) from the swizzled LDS image
//   MODE 2: + 8 LDS-DMA pieces of 1 KiB (global_load_lds_dwordx4) streamed from an L2-resident window into the ring
// -- the instructions gemm_bf16_kernel<.., 15 / 17> issues per K tile, free-running (the waves drift, nothing waits for a
// barrier, counted vmcnt only bounds the DMA queue).  On N(0,1) data the rates of MODE 0 (pe_mfma_probe) / 1 / 2 price the
// matrix pipe alone, the LDS fragment traffic, and the operand stream under the chip's power limit: what is left between MODE 2
// and the measured GEMM is schedule (barriers, prologue, epilogue, tile quantisation), what is above MODE 2 is not reachable by
// a 256 x 256 x 64 tiling with this wave tile at all.
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(512, 2) gemm_mix_probe_kernel(const char* __restrict__ src, unsigned window_bytes /* power of two */,
                                                               float* __restrict__ out, int iters) {
    // (round 5: on v_mfma_f32_16x16x32_bf16, 4 x 8 blocks of 16 x 16 per wave, as the GEMM: per K tile and wave 64 MFMAs of 4 passes, the same
    // 24 fragment reads and 8 LDS-DMA pieces)
    extern __shared__ __attribute__((aligned(16))) char probe_smem[];
    typedef __attribute__((ext_vector_type(8))) __bf16 probe_bf16x8;
    typedef __attribute__((ext_vector_type(4))) float probe_f32x4;
    const int lane = (int)(threadIdx.x & 63);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int l15 = lane & 15, g = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    const int sw = (l15 >> 1) & 7;
    // every work-group of an XCD (block b runs on XCD b % 8) streams the same window: the stream is L2 resident
    const char* win = src + (size_t)(blockIdx.x & 7) * window_bytes;
    const unsigned wmask = window_bytes - 1u;
    const unsigned lane_off = (unsigned)lane * 16u;
    // fill the whole LDS image once (160 pieces of 1 KiB, 20 per wave)
    for (int i = 0; i < 20; ++i) {
        const unsigned piece = (unsigned)(w * 20 + i);
        glds16(win + ((piece * 1024u + lane_off) & wmask), probe_smem + piece * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    probe_f32x4 acc[4][8];
    for (int mi = 0; mi < 4; ++mi)
        for (int ni = 0; ni < 8; ++ni)
            for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.f;
    const int a_off = (wm * 64 + l15) * 128;
    const int w_off = (wn * 128 + l15) * 128;
    char* const a_base = probe_smem;
    char* const w_base = probe_smem + 2 * 32768;
    probe_bf16x8 fa[2][2], fw[8][2];       // one 32-row phase: 2 activation blocks x 2 k-steps, 8 weight blocks x 2 k-steps
    // MODE 0 never touches LDS again: fragments are read once
    for (int ks = 0; ks < 2; ++ks) {
        for (int mi = 0; mi < 2; ++mi) fa[mi][ks] = *(const probe_bf16x8*)(a_base + a_off + mi * 2048 + (((ks * 4 + g) ^ sw) << 4));
        for (int ni = 0; ni < 8; ++ni) fw[ni][ks] = *(const probe_bf16x8*)(w_base + w_off + ni * 2048 + (((ks * 4 + g) ^ sw) << 4));
    }
    unsigned stream_off = (unsigned)blockIdx.x * 65536u + (unsigned)w * 8192u;
    int ab = 0, ws = 0;
    for (int it = 0; it < iters; ++it) {
        const char* Sa = a_base + ab * 32768;
        const char* Sw = w_base + ws * 32768;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            if constexpr (MODE >= 1) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) fa[mi][ks] = *(const probe_bf16x8*)(Sa + a_off + (ph * 2 + mi) * 2048 + (((ks * 4 + g) ^ sw) << 4));
                if (ph == 0) {
#pragma unroll
                    for (int ni = 0; ni < 8; ++ni)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) fw[ni][ks] = *(const probe_bf16x8*)(Sw + w_off + ni * 2048 + (((ks * 4 + g) ^ sw) << 4));
                }
            }
            if constexpr (MODE >= 2) {
                // 4 pieces per half K tile: A pieces into the other A buffer, W pieces into the ring slot two ahead
                char* dst = ph == 0 ? a_base + (ab ^ 1) * 32768 + w * 4096 : w_base + ((ws + 2) % 3) * 32768 + w * 4096;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    glds16(win + ((stream_off + lane_off) & wmask), dst + j * 1024);
                    stream_off += 1024u;
                }
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 8; ++ni)
                        acc[ph * 2 + mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni][ks], fa[mi][ks], acc[ph * 2 + mi][ni], 0, 0, 0);
        }
        stream_off += 65536u - 8192u;
        ab ^= 1;
        ws = ws == 2 ? 0 : ws + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int mi = 0; mi < 4; ++mi)
        for (int ni = 0; ni < 8; ++ni)
            for (int r = 0; r < 4; ++r) s += acc[mi][ni][r];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

}  // namespace pe

using namespace pe;

extern "C" {

int pe_gemm_mix_probe(int mode, const void* src, size_t src_bytes, void* out, int blocks, int iters, double* flops, void* stream) {
    PE_REQUIRE(src && out && blocks > 0 && blocks <= 4096 && iters > 0, "pe_gemm_mix_probe: bad arguments");
    PE_REQUIRE(mode >= 0 && mode <= 2, "pe_gemm_mix_probe: mode %d (0 MFMA only, 1 + LDS fragment reads, 2 + LDS-DMA stream)", mode);
    PE_REQUIRE(src_bytes >= (size_t)8 * (1u << 20) && src_bytes % 8 == 0 && ((src_bytes / 8) & (src_bytes / 8 - 1)) == 0 &&
                   src_bytes / 8 <= (1u << 30),
               "pe_gemm_mix_probe: src_bytes must be 8 x a power of two >= 1 MiB (one window per XCD)");
    const unsigned window = (unsigned)(src_bytes / 8);
    constexpr int lds = 160 * 1024;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_mix_probe_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_mix_probe_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_mix_probe_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return set_error(PE_ERR_HIP, "pe_gemm_mix_probe: hipFuncSetAttribute: %s", hipGetErrorString(e));
        configured = true;
    }
    if (mode == 0)
        hipLaunchKernelGGL(gemm_mix_probe_kernel<0>, dim3(blocks), dim3(512), lds, (hipStream_t)stream, (const char*)src, window, (float*)out, iters);
    else if (mode == 1)
        hipLaunchKernelGGL(gemm_mix_probe_kernel<1>, dim3(blocks), dim3(512), lds, (hipStream_t)stream, (const char*)src, window, (float*)out, iters);
    else
        hipLaunchKernelGGL(gemm_mix_probe_kernel<2>, dim3(blocks), dim3(512), lds, (hipStream_t)stream, (const char*)src, window, (float*)out, iters);
    if (flops) *flops = (double)blocks * 8.0 * (double)iters * 32.0 * (2.0 * 32 * 32 * 16);
    return check_launch("gemm_mix_probe_kernel");
}

int pe_attn_mix_probe(const void* q, const void* k, const void* vt, void* out, int H, int S, int S_pad, int ldo, double* flops, void* stream) {
    if (flops) *flops = 4.0 * (double)S * S * 128.0 * H;
    return launch_attn_mix_probe(q, k, vt, out, H, S, S_pad, ldo, (hipStream_t)stream);
}

int pe_mfma_probe(const void* frags, void* out, int blocks, int iters, double* flops, void* stream) {
    PE_REQUIRE(frags && out && blocks > 0 && blocks <= 4096 && iters > 0, "pe_mfma_probe: bad arguments");
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream, (const uint4*)frags, (float*)out, iters);
    if (flops) *flops = (double)blocks * 8.0 * (double)iters * 32.0 * (2.0 * 32 * 32 * 16);
    return check_launch("mfma_probe_kernel");
}

int pe_profile_enable(int max_events, int sample_every) {
    PE_REQUIRE(max_events > 0 && max_events <= (1 << 20) && sample_every > 0, "pe_profile_enable: bad arguments");
    pe_profile_disable();
    g_prof.pool.resize((size_t)max_events);
    for (auto& s : g_prof.pool) {
        if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess)
            return set_error(PE_ERR_HIP, "pe_profile_enable: hipEventCreate failed");
    }
    g_prof.used = 0;
    for (auto& c : g_prof.seen) c = 0;
    g_prof.sample_every = sample_every;
    g_prof.on = true;
    return PE_OK;
}

void pe_profile_disable(void) {
    g_prof.on = false;
    for (auto& s : g_prof.pool) {
        (void)hipEventDestroy(s.a);
        (void)hipEventDestroy(s.b);
    }
    g_prof.pool.clear();
    g_prof.used = 0;
}

int pe_profile_read(int kind, long long* launches_seen, long long* sampled, double* total_ms, double* total_work) {
    PE_REQUIRE(kind >= 0 && kind < PROF_KINDS, "pe_profile_read: kind=%d", kind);
    PE_REQUIRE(launches_seen && sampled && total_ms && total_work, "pe_profile_read: null output");
    long long n = 0;
    double ms = 0.0, work = 0.0;
    for (size_t i = 0; i < g_prof.used; ++i) {
        const ProfSlot& s = g_prof.pool[i];
        if (s.kind != kind) continue;
        if (hipEventSynchronize(s.b) != hipSuccess) return set_error(PE_ERR_HIP, "pe_profile_read: event sync failed");
        float t = 0.f;
        if (hipEventElapsedTime(&t, s.a, s.b) != hipSuccess) return set_error(PE_ERR_HIP, "pe_profile_read: elapsed failed");
        ms += t;
        work += s.work;
        ++n;
    }
    *launches_seen = (long long)g_prof.seen[kind];
    *sampled = n;
    *total_ms = ms;
    *total_work = work;
    return PE_OK;
}

}  // extern "C"
