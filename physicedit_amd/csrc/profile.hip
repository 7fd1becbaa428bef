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
// Matrix-pipe probe: nothing but v_mfma_f32_32x32x16_bf16 (8 independent accumulators per wave, 8 waves per work-group, operands
// read once from `frags`, no LDS / memory access inside the loop).  On all-zero fragments it runs at the nominal dense peak
// (2.4 GHz); on random fragments the chip's power limit lowers the clock, and the rate it reports is the ceiling any bf16 MFMA
// kernel has on such data before a byte is moved (profiles/r02_gemm_notes.md section 5).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) mfma_probe_kernel(const uint4* __restrict__ frags, float* __restrict__ out, int iters) {
    typedef __attribute__((ext_vector_type(8))) __bf16 probe_bf16x8;
    typedef __attribute__((ext_vector_type(16))) float probe_f32x16;
    const int tid = threadIdx.x + blockIdx.x * 512;
    uint4 av[4], bv[4];
    for (int i = 0; i < 4; ++i) {
        av[i] = frags[(tid * 8 + i) & 0xfffff];
        bv[i] = frags[(tid * 8 + 4 + i) & 0xfffff];
    }
    probe_f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(probe_bf16x8, av[(i + k) & 3]),
                                                                 __builtin_bit_cast(probe_bf16x8, bv[(i >> 1) & 3]), acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[tid] = s;
}

}  // namespace pe

using namespace pe;

extern "C" {

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
