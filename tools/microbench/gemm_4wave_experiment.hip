// Experiment: 4-wave bf16 GEMM main loop, 256x256x64 block tile, wave tile 128x128 (acc 256 regs), one wave per SIMD.
// out[M,N] = bf16(A[M,K] . W[N,K]^T).  Standalone: correctness vs a naive kernel on sampled entries, timing vs flops.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include <string.h>
#include <math.h>
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define DEV __device__ __forceinline__
DEV void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
DEV int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
constexpr int BM = 256, BN = 256, BK = 64, BAND = 8;
constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2;   // 32 KiB each
constexpr int LDS_BYTES = 2 * A_BYTES + 3 * W_BYTES;          // 160 KiB

template <int KD>   // unused knob
__global__ void __launch_bounds__(256, 1) gemm4w(const bf16* __restrict__ A, const bf16* __restrict__ W, bf16* __restrict__ out,
                                                int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = w >> 1, wn = w & 1;
    const int tilesM = (M + BM - 1) / BM, tilesN = (N + BN - 1) / BN;
    int bid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    int m0, n0;
    {
        const int per_band = BAND * tilesN, band = bid / per_band, rem = bid - band * per_band;
        const int gm = min(BAND, tilesM - band * BAND), tn = rem / gm, tm = band * BAND + (rem - tn * gm);
        m0 = tm * BM; n0 = tn * BN;
    }
    // staging: wave w moves A pieces w*8..w*8+7 and W pieces likewise (1 KiB = 8 rows x 128 B each)
    const char* a_src[8];
    const char* w_src[8];
    {
        const int rin = lane >> 3, slot = lane & 7;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = (w * 8 + i) * 8 + rin, chunk = slot ^ ((row >> 1) & 7);
            a_src[i] = (const char*)A + (size_t)min(m0 + row, M - 1) * K * 2 + chunk * 16;
            w_src[i] = (const char*)W + (size_t)min(n0 + row, N - 1) * K * 2 + chunk * 16;
        }
    }
    const int nk = K / BK;
    char* const a_base = smem;
    char* const w_base = smem + 2 * A_BYTES;
    auto stage_a = [&](int t, int first, int count) {
        const int tc = min(t, nk - 1);
        char* base = a_base + (t & 1) * A_BYTES + w * 8192;
#pragma unroll
        for (int i = 0; i < 8; ++i) if (i >= first && i < first + count) glds16(a_src[i] + tc * 128, base + i * 1024);
    };
    auto stage_w = [&](int t, int slot, int first, int count) {
        const int tc = min(t, nk - 1);
        char* base = w_base + slot * W_BYTES + w * 8192;
#pragma unroll
        for (int i = 0; i < 8; ++i) if (i >= first && i < first + count) glds16(w_src[i] + tc * 128, base + i * 1024);
    };
    f32x16 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    const int sw = (l31 >> 1) & 7;
    auto frag_a = [&](const char* Sa, int kk, bf16x8 (&af)[4]) {
        const int coff = ((kk * 2 + h) ^ sw) << 4;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[mi] = *(const bf16x8*)(Sa + (wm * 128 + mi * 32 + l31) * 128 + coff);
    };
    auto frag_w = [&](const char* Sw, int kk, bf16x8 (&wf)[4]) {
        const int coff = ((kk * 2 + h) ^ sw) << 4;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) wf[ni] = *(const bf16x8*)(Sw + (wn * 128 + ni * 32 + l31) * 128 + coff);
    };
    auto mma = [&](bf16x8 (&af)[4], bf16x8 (&wf)[4]) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], af[mi], acc[mi][ni], 0, 0, 0);
    };
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#define CLUSTER(NV)                                                         \
    do {                                                                    \
        for (int i_ = 0; i_ < 8; ++i_) { SGB(0x008, 1); SGB(0x100, 1); }    \
        for (int i_ = 0; i_ < (NV); ++i_) { SGB(0x008, 1); SGB(0x020, 1); } \
        SGB(0x008, 8 - (NV));                                               \
    } while (0)
    stage_a(0, 0, 8);
    stage_w(0, 0, 0, 8);
    stage_w(1, 1, 0, 8);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __syncthreads();
    bf16x8 fa0[4], fw0[4], fa1[4], fw1[4];
    frag_a(a_base, 0, fa0);
    frag_w(w_base, 0, fw0);
    stage_a(1, 0, 4);
    int ws_cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const char* Sa = a_base + (kt & 1) * A_BYTES;
        const char* San = a_base + ((kt + 1) & 1) * A_BYTES;
        const int ws_n1 = ws_cur == 2 ? 0 : ws_cur + 1, ws_n2 = ws_n1 == 2 ? 0 : ws_n1 + 1;
        const char* Sw = w_base + ws_cur * W_BYTES;
        const char* Swn = w_base + ws_n1 * W_BYTES;
        frag_a(Sa, 1, fa1); frag_w(Sw, 1, fw1);
        stage_a(kt + 1, 4, 4);
        mma(fa0, fw0);
        CLUSTER(4);
        frag_a(Sa, 2, fa0); frag_w(Sw, 2, fw0);
        stage_w(kt + 2, ws_n2, 0, 4);
        mma(fa1, fw1);
        CLUSTER(4);
        frag_a(Sa, 3, fa1); frag_w(Sw, 3, fw1);
        stage_w(kt + 2, ws_n2, 4, 4);
        mma(fa0, fw0);
        CLUSTER(4);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");   // all but the 8 newest (= W(kt+2))
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        frag_a(San, 0, fa0); frag_w(Swn, 0, fw0);
        stage_a(kt + 2, 0, 4);
        mma(fa1, fw1);
        CLUSTER(4);
        ws_cur = ws_n1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // plain epilogue (experiment): acc[mi][ni][4q+r] = C[m0+wm*128+mi*32+l31][n0+wn*128+ni*32+8q+4h+r]
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm * 128 + mi * 32 + l31;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * 128 + ni * 32 + 8 * q + 4 * h;
                bf16x4 y;
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = (bf16)acc[mi][ni][4 * q + r];
                if (m < M && n < N) *(bf16x4*)(out + (size_t)m * N + n) = y;
            }
    }
}

__global__ void naive(const bf16* A, const bf16* W, float* ref, const int* ms, const int* ns, int cnt, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += (float)A[(size_t)ms[i] * K + k] * (float)W[(size_t)ns[i] * K + k];
    ref[i] = s;
}
__global__ void fill(bf16* p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (bf16)(((int)(x & 0xffff) - 32768) / 32768.0f * scale);
    }
}
int main(int argc, char** argv) {
    const int shapes[][3] = {{8192, 12288, 3072}, {8192, 3072, 12288}, {8192, 8192, 8192}, {8704, 9216, 3072}, {300, 520, 256}};
    hipFuncSetAttribute((const void*)gemm4w<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        bf16 *A, *W, *O;
        hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&O, (size_t)M * N * 2);
        hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, A, (size_t)M * K, 1u, 1.0f);
        hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, W, (size_t)N * K, 7u, 0.05f);
        const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
        auto run = [&]() { hipLaunchKernelGGL(gemm4w<4>, dim3(tiles), dim3(256), LDS_BYTES, 0, A, W, O, M, N, K); };
        run(); hipDeviceSynchronize();
        // check 4096 sampled entries
        const int cnt = 4096; std::vector<int> ms(cnt), ns(cnt);
        for (int i = 0; i < cnt; ++i) { ms[i] = rand() % M; ns[i] = rand() % N; }
        ms[0] = M - 1; ns[0] = N - 1; ms[1] = 0; ns[1] = 0;
        int *dms, *dns; float* dref; hipMalloc(&dms, cnt * 4); hipMalloc(&dns, cnt * 4); hipMalloc(&dref, cnt * 4);
        hipMemcpy(dms, ms.data(), cnt * 4, hipMemcpyHostToDevice); hipMemcpy(dns, ns.data(), cnt * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(naive, dim3((cnt + 255) / 256), dim3(256), 0, 0, A, W, dref, dms, dns, cnt, K);
        std::vector<float> ref(cnt); hipMemcpy(ref.data(), dref, cnt * 4, hipMemcpyDeviceToHost);
        std::vector<uint16_t> ho((size_t)M * N); hipMemcpy(ho.data(), O, (size_t)M * N * 2, hipMemcpyDeviceToHost);
        double maxrel = 0; int bad = 0;
        for (int i = 0; i < cnt; ++i) {
            uint32_t bits = (uint32_t)ho[(size_t)ms[i] * N + ns[i]] << 16; float got; memcpy(&got, &bits, 4);
            const double rel = fabs(got - ref[i]) / fmax(fabs(ref[i]), 0.05);
            if (rel > maxrel) maxrel = rel;
            if (rel > 0.02) ++bad;
        }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) run();
        hipEventRecord(e0);
        const int iters = 20;
        for (int i = 0; i < iters; ++i) run();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms_; hipEventElapsedTime(&ms_, e0, e1);
        printf("4-wave %dx%dx%d: %.0f us  %.0f TF/s   check: max rel err %.4f, bad %d/%d\n", M, N, K, ms_ * 1e3 / iters,
               2.0 * M * N * K / (ms_ * 1e-3 / iters) / 1e12, maxrel, bad, cnt);
        hipFree(A); hipFree(W); hipFree(O); hipFree(dms); hipFree(dns); hipFree(dref);
    }
    return 0;
}
