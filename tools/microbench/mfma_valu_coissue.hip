// Can VALU work hide under MFMAs on gfx950?  One block per CU, W waves per SIMD; each iteration issues 8 independent
// 32x32x16 bf16 MFMAs and NV filler VALU instructions (v_fma_f32 or v_exp_f32), same wave, interleaved by the compiler's
// source order (sched_barrier fences keep the groups).  Reports cycles per iteration (s_memtime) for the slowest wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NV, int KIND, bool MF>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, int iters) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (MF) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[g], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV / 8; ++j) {
                const int idx = (g * (NV / 8) + j) & 15;
                if (KIND == 0) v[idx] = __builtin_fmaf(v[idx], 1.0001f, 0.5f);
                else v[idx] = __builtin_amdgcn_exp2f(v[idx]) ;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NV, int KIND, bool MF>
void run(const char* name, int threads) {
    float* out; long long* cyc; const int iters = 2000, blocks = 256;
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&cyc, blocks * 8 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, KIND, MF>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, KIND, MF>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // ns per iteration per SIMD-resident wave set; MFMA-bound ideal: waves/SIMD * 8 * 32 cycles
    printf("%-34s threads %3d: %.1f ns/iter  (%.0f cycles @2.4GHz)\n", name, threads, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int threads : {256, 512}) {
        run<0, 0, true>("8 MFMA", threads);
        run<32, 0, false>("32 fma", threads);
        run<32, 0, true>("8 MFMA + 32 fma", threads);
        run<64, 0, false>("64 fma", threads);
        run<64, 0, true>("8 MFMA + 64 fma", threads);
        run<16, 1, false>("16 exp", threads);
        run<16, 1, true>("8 MFMA + 16 exp", threads);
        run<32, 1, false>("32 exp", threads);
        run<32, 1, true>("8 MFMA + 32 exp", threads);
    }
    return 0;
}
