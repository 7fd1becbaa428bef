// Which bf16 MFMA shape sustains more FLOP/s under the chip's power limit (N(0,1) operands in registers, no memory traffic)?
// v_mfma_f32_32x32x16_bf16 (8 passes, 32 KiFLOP) against v_mfma_f32_16x16x32_bf16 (4 passes, 16 KiFLOP: twice the operand registers
// read per flop).  8 waves per CU, 16 independent accumulator sets per wave.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/mfma_shape_power.hip -o /tmp/mfma_shape_power && /tmp/mfma_shape_power
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE>
__global__ void __launch_bounds__(512) probe(const uint4* __restrict__ frags, float* __restrict__ out, int iters) {
    const int tid = threadIdx.x + blockIdx.x * 512;
    bf16x8 av[4], bv[4];
    for (int i = 0; i < 4; ++i) {
        av[i] = __builtin_bit_cast(bf16x8, frags[(tid * 8 + i) & 0xfffff]);
        bv[i] = __builtin_bit_cast(bf16x8, frags[(tid * 8 + 4 + i) & 0xfffff]);
    }
    float s = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[(i + k) & 3], bv[(i >> 1) & 3], acc[i], 0, 0, 0);
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 16; ++j) s += acc[i][j];
    } else {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[(i + k) & 3], bv[(i >> 2) & 3], acc[i], 0, 0, 0);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 4; ++j) s += acc[i][j];
    }
    out[tid] = s;
}
// Does the ORDER in which a wave's MFMAs walk its operand fragments change the power (= the sustained rate)?  16 x 16 x 32 bf16, 16 accumulators,
// 4 A and 4 B fragments: MODE 0 = A changes with every MFMA, B every 4th (the GEMM's loop order); 1 = B every MFMA, A every 4th;
// 2 = both change with every MFMA; 3 = both change every 4th MFMA only (not a GEMM: the same product four times)
template <int MODE>
__global__ void __launch_bounds__(512) probe_order(const uint4* __restrict__ frags, float* __restrict__ out, int iters) {
    const int tid = threadIdx.x + blockIdx.x * 512;
    bf16x8 av[4], bv[4];
    for (int i = 0; i < 4; ++i) {
        av[i] = __builtin_bit_cast(bf16x8, frags[(tid * 8 + i) & 0xfffff]);
        bv[i] = __builtin_bit_cast(bf16x8, frags[(tid * 8 + 4 + i) & 0xfffff]);
    }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ia = MODE == 0 ? (i & 3) : MODE == 1 ? (i >> 2) : MODE == 2 ? (i & 3) : (i >> 2);
                const int ib = MODE == 0 ? (i >> 2) : MODE == 1 ? (i & 3) : MODE == 2 ? ((i + (i >> 2)) & 3) : (i >> 2);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[(ia + k) & 3], bv[ib], acc[i], 0, 0, 0);
            }
    float s = 0.f;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j];
    out[tid] = s;
}
typedef __attribute__((ext_vector_type(8))) int i32x8;
// the e4m3 pair: v_mfma_scale_f32_32x32x64_f8f6f4 (16 passes, 128 KiFLOP) against v_mfma_scale_f32_16x16x128_f8f6f4 (8 passes, 64 KiFLOP), unit scales
template <int SHAPE>
__global__ void __launch_bounds__(512) probe8(const uint4* __restrict__ frags, float* __restrict__ out, int iters) {
    const int tid = threadIdx.x + blockIdx.x * 512;
    i32x8 av[4], bv[4];
    for (int i = 0; i < 4; ++i) {
        const uint4 a0 = frags[(tid * 16 + 2 * i) & 0xfffff], a1 = frags[(tid * 16 + 2 * i + 1) & 0xfffff];
        const uint4 b0 = frags[(tid * 16 + 8 + 2 * i) & 0xfffff], b1 = frags[(tid * 16 + 9 + 2 * i) & 0xfffff];
        av[i] = i32x8{(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
        bv[i] = i32x8{(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
    }
    float s = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av[(i + k) & 3], bv[(i >> 1) & 3], acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 16; ++j) s += acc[i][j];
    } else {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av[(i + k) & 3], bv[(i >> 2) & 3], acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 4; ++j) s += acc[i][j];
    }
    out[tid] = s;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
template <typename K> static double run(K kernel, const uint4* frags, float* out, int iters, double flop_per_iter_wave) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    double best = 0;
    for (int rep = 0; rep < 3; ++rep) {
        const int n = rep == 0 ? 2000 : iters;
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(kernel, dim3(512), dim3(512), 0, 0, frags, out, n);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep) { const double tf = 512.0 * 8 * n * flop_per_iter_wave / (ms * 1e-3) / 1e12; if (tf > best) best = tf; }
    }
    return best;
}
int main() {
    const size_t n = (size_t)8 << 20;
    std::vector<uint16_t> h(n);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n; i += 2) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; const double u1 = ((s >> 11) + 1.0) / 9007199254740993.0;
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; const double u2 = (s >> 11) / 9007199254740992.0;
        const double r = sqrt(-2.0 * log(u1));
        h[i] = f2bf((float)(r * cos(6.283185307179586 * u2))); h[i + 1] = f2bf((float)(r * sin(6.283185307179586 * u2)));
    }
    uint4 *dr, *dz; float* out;
    CK(hipMalloc(&dr, n * 2)); CK(hipMalloc(&dz, n * 2)); CK(hipMalloc(&out, 512 * 512 * 4));
    CK(hipMemcpy(dr, h.data(), n * 2, hipMemcpyHostToDevice)); CK(hipMemset(dz, 0, n * 2));
    const double f32 = 32.0 * 2 * 32 * 32 * 16, f16 = 64.0 * 2 * 16 * 16 * 32;
    printf("shape                       N(0,1) TF/s   zeros TF/s\n");
    printf("v_mfma_f32_32x32x16_bf16  %10.0f %12.0f\n", run(probe<32>, dr, out, 100000, f32), run(probe<32>, dz, out, 100000, f32));
    printf("v_mfma_f32_16x16x32_bf16  %10.0f %12.0f\n", run(probe<16>, dr, out, 100000, f16), run(probe<16>, dz, out, 100000, f16));
    printf("16x16x32 bf16, A changes every MFMA, B every 4th   %10.0f\n", run(probe_order<0>, dr, out, 100000, f16));
    printf("16x16x32 bf16, B changes every MFMA, A every 4th   %10.0f\n", run(probe_order<1>, dr, out, 100000, f16));
    printf("16x16x32 bf16, both change every MFMA              %10.0f\n", run(probe_order<2>, dr, out, 100000, f16));
    printf("16x16x32 bf16, both change every 4th MFMA          %10.0f\n", run(probe_order<3>, dr, out, 100000, f16));
    // e4m3 operands: N(0,1) values rounded to e4m3 (sign, 4-bit exponent bias 7, 3-bit mantissa)
    std::vector<uint8_t> h8(n * 2);
    for (size_t i = 0; i < n * 2; ++i) {
        uint32_t u; float f; const uint16_t b = h[i % n]; u = (uint32_t)b << 16; memcpy(&f, &u, 4);
        const float a = fabsf(f); uint8_t e = 0;
        if (a >= 0.001953125f) { int ex; const float m = frexpf(a, &ex); int E = ex - 1 + 7; if (E < 1) { e = (uint8_t)lrintf(a * 512.f); } else { int mm = (int)lrintf((m * 2 - 1) * 8); if (mm == 8) { mm = 0; ++E; } e = (uint8_t)((E << 3) | mm); } }
        h8[i] = e | (f < 0 ? 0x80 : 0);
    }
    CK(hipMemcpy(dr, h8.data(), n * 2, hipMemcpyHostToDevice));
    const double g32 = 32.0 * 2 * 32 * 32 * 64, g16 = 64.0 * 2 * 16 * 16 * 128;
    printf("v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3)   %10.0f %12.0f\n", run(probe8<32>, dr, out, 50000, g32), run(probe8<32>, dz, out, 50000, g32));
    printf("v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3)  %10.0f %12.0f\n", run(probe8<16>, dr, out, 50000, g16), run(probe8<16>, dz, out, 50000, g16));
    return 0;
}
