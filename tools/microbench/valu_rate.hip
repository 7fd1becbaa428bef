// Issue cost of single VALU instructions on gfx950, alone and next to a running MFMA stream (one wave per SIMD, as in
// flash_attn_w4_kernel): s_memtime around 64 x 32 independent instructions.
//   hipcc --offload-arch=gfx950 -O2 -o valu_rate tools/microbench/valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

template <int MODE, bool MFMA>
__global__ void __launch_bounds__(256, 1) k(long long* out, float seed) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    f32x2 p0 = {seed, seed}, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f, p4 = p0 + 4.f, p5 = p0 + 5.f, p6 = p0 + 6.f, p7 = p0 + 7.f;
    f32x2 c = {1.0001f, 0.9999f};
    float cs = 1.0001f;
    f32x16 acc0 = {}, acc1 = {};
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(seed + i); fb[i] = (__bf16)(seed - i); }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 64; ++it) {
        // 4 groups of (optional MFMA + 8 VALU) per iteration
        if constexpr (MODE == 0) {          // v_fma_f32
            for (int g = 0; g < 4; ++g) {
                if constexpr (MFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(fa), "v"(fb));
                asm volatile("v_fma_f32 %0, %0, %8, %8\n\tv_fma_f32 %1, %1, %8, %8\n\tv_fma_f32 %2, %2, %8, %8\n\tv_fma_f32 %3, %3, %8, %8\n\t"
                             "v_fma_f32 %4, %4, %8, %8\n\tv_fma_f32 %5, %5, %8, %8\n\tv_fma_f32 %6, %6, %8, %8\n\tv_fma_f32 %7, %7, %8, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cs));
            }
        } else if constexpr (MODE == 1) {   // v_pk_fma_f32
            for (int g = 0; g < 4; ++g) {
                if constexpr (MFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(fa), "v"(fb));
                asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n\tv_pk_fma_f32 %1, %1, %8, %8\n\tv_pk_fma_f32 %2, %2, %8, %8\n\tv_pk_fma_f32 %3, %3, %8, %8\n\t"
                             "v_pk_fma_f32 %4, %4, %8, %8\n\tv_pk_fma_f32 %5, %5, %8, %8\n\tv_pk_fma_f32 %6, %6, %8, %8\n\tv_pk_fma_f32 %7, %7, %8, %8"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c));
            }
        } else if constexpr (MODE == 2) {   // v_exp_f32
            for (int g = 0; g < 4; ++g) {
                if constexpr (MFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(fa), "v"(fb));
                asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
                             "v_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        } else if constexpr (MODE == 3) {   // v_pk_add_f32
            for (int g = 0; g < 4; ++g) {
                if constexpr (MFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(fa), "v"(fb));
                asm volatile("v_pk_add_f32 %0, %0, %8\n\tv_pk_add_f32 %1, %1, %8\n\tv_pk_add_f32 %2, %2, %8\n\tv_pk_add_f32 %3, %3, %8\n\t"
                             "v_pk_add_f32 %4, %4, %8\n\tv_pk_add_f32 %5, %5, %8\n\tv_pk_add_f32 %6, %6, %8\n\tv_pk_add_f32 %7, %7, %8"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c));
            }
        } else if constexpr (MODE == 4) {   // v_add_f32
            for (int g = 0; g < 4; ++g) {
                if constexpr (MFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(fa), "v"(fb));
                asm volatile("v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %8\n\tv_add_f32 %2, %2, %8\n\tv_add_f32 %3, %3, %8\n\t"
                             "v_add_f32 %4, %4, %8\n\tv_add_f32 %5, %5, %8\n\tv_add_f32 %6, %6, %8\n\tv_add_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cs));
            }
        } else if constexpr (MODE == 5) {   // v_cvt_pk_bf16_f32
            for (int g = 0; g < 4; ++g) {
                if constexpr (MFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(fa), "v"(fb));
                asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n\tv_cvt_pk_bf16_f32 %1, %1, %2\n\tv_cvt_pk_bf16_f32 %2, %2, %3\n\tv_cvt_pk_bf16_f32 %3, %3, %4\n\t"
                             "v_cvt_pk_bf16_f32 %4, %4, %5\n\tv_cvt_pk_bf16_f32 %5, %5, %6\n\tv_cvt_pk_bf16_f32 %6, %6, %7\n\tv_cvt_pk_bf16_f32 %7, %7, %0"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        } else if constexpr (MODE == 6) {   // v_pk_mul_f32
            for (int g = 0; g < 4; ++g) {
                if constexpr (MFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(fa), "v"(fb));
                asm volatile("v_pk_mul_f32 %0, %0, %8\n\tv_pk_mul_f32 %1, %1, %8\n\tv_pk_mul_f32 %2, %2, %8\n\tv_pk_mul_f32 %3, %3, %8\n\t"
                             "v_pk_mul_f32 %4, %4, %8\n\tv_pk_mul_f32 %5, %5, %8\n\tv_pk_mul_f32 %6, %6, %8\n\tv_pk_mul_f32 %7, %7, %8"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c));
            }
        } else if constexpr (MODE == 7) {   // v_rcp_f32
            for (int g = 0; g < 4; ++g) {
                if constexpr (MFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(fa), "v"(fb));
                asm volatile("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3\n\t"
                             "v_rcp_f32 %4, %4\n\tv_rcp_f32 %5, %5\n\tv_rcp_f32 %6, %6\n\tv_rcp_f32 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        } else if constexpr (MODE == 10) {  // v_dot2c_f32_bf16 (VOP2: d += a.lo * b.lo + a.hi * b.hi; round 6: one of these on the packed P replaces two row-sum adds)
            for (int g = 0; g < 4; ++g) {
                if constexpr (MFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(fa), "v"(fb));
                asm volatile("v_dot2c_f32_bf16 %0, 0x3f803f80, %8\n\tv_dot2c_f32_bf16 %1, 0x3f803f80, %8\n\tv_dot2c_f32_bf16 %2, 0x3f803f80, %8\n\tv_dot2c_f32_bf16 %3, 0x3f803f80, %8\n\t"
                             "v_dot2c_f32_bf16 %4, 0x3f803f80, %8\n\tv_dot2c_f32_bf16 %5, 0x3f803f80, %8\n\tv_dot2c_f32_bf16 %6, 0x3f803f80, %8\n\tv_dot2c_f32_bf16 %7, 0x3f803f80, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cs));
            }
        } else if constexpr (MODE == 11) {  // v_dot2_f32_bf16 (VOP3P)
            for (int g = 0; g < 4; ++g) {
                if constexpr (MFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(fa), "v"(fb));
                asm volatile("v_dot2_f32_bf16 %0, %8, %8, %0\n\tv_dot2_f32_bf16 %1, %8, %8, %1\n\tv_dot2_f32_bf16 %2, %8, %8, %2\n\tv_dot2_f32_bf16 %3, %8, %8, %3\n\t"
                             "v_dot2_f32_bf16 %4, %8, %8, %4\n\tv_dot2_f32_bf16 %5, %8, %8, %5\n\tv_dot2_f32_bf16 %6, %8, %8, %6\n\tv_dot2_f32_bf16 %7, %8, %8, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cs));
            }
        } else if constexpr (MODE == 12) {  // ONE dependent chain of v_dot2c_f32_bf16 (the row sum is one accumulator per block)
            for (int g = 0; g < 4; ++g) {
                if constexpr (MFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(fa), "v"(fb));
                asm volatile("v_dot2c_f32_bf16 %0, 0x3f803f80, %1\n\tv_dot2c_f32_bf16 %0, 0x3f803f80, %1\n\tv_dot2c_f32_bf16 %0, 0x3f803f80, %1\n\tv_dot2c_f32_bf16 %0, 0x3f803f80, %1\n\t"
                             "v_dot2c_f32_bf16 %0, 0x3f803f80, %1\n\tv_dot2c_f32_bf16 %0, 0x3f803f80, %1\n\tv_dot2c_f32_bf16 %0, 0x3f803f80, %1\n\tv_dot2c_f32_bf16 %0, 0x3f803f80, %1"
                             : "+v"(a0) : "v"(cs));
            }
        } else if constexpr (MODE == 13) {  // ONE dependent chain of v_add_f32 (what the row sum is today)
            for (int g = 0; g < 4; ++g) {
                if constexpr (MFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(fa), "v"(fb));
                asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\t"
                             "v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1"
                             : "+v"(a0) : "v"(cs));
            }
        } else if constexpr (MODE == 8) {   // nothing but the MFMAs
            for (int g = 0; g < 4; ++g) {
                if constexpr (MFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(fa), "v"(fb));
            }
        } else if constexpr (MODE == 9) {   // v_lshlrev_b32 (integer ALU)
            for (int g = 0; g < 4; ++g) {
                if constexpr (MFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(fa), "v"(fb));
                asm volatile("v_lshlrev_b32 %0, 1, %0\n\tv_lshlrev_b32 %1, 1, %1\n\tv_lshlrev_b32 %2, 1, %2\n\tv_lshlrev_b32 %3, 1, %3\n\t"
                             "v_lshlrev_b32 %4, 1, %4\n\tv_lshlrev_b32 %5, 1, %5\n\tv_lshlrev_b32 %6, 1, %6\n\tv_lshlrev_b32 %7, 1, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + acc0[0] + acc1[1];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (s == 12345.678f) out[1] = 1;
}

template <int MODE, bool MFMA>
static void run(const char* name, long long* d) {
    long long h[2];
    for (int r = 0; r < 2; ++r) {
        hipLaunchKernelGGL((k<MODE, MFMA>), dim3(256), dim3(256), 0, 0, d, 0.5f);
        if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return;
    }
    // 64 iterations x 4 groups: per group 8 VALU (+ 1 MFMA)
    printf("%-22s %s: %7.1f ticks per group of 8 (%5.2f per instruction)\n", name, MFMA ? "with 1 MFMA per 8" : "alone            ", h[0] / 256.0, h[0] / 256.0 / 8.0);
}

int main() {
    long long* d;
    if (hipMalloc(&d, 16) != hipSuccess) return 1;
    run<8, true>("mfma only", d);
    run<0, false>("v_fma_f32", d);       run<0, true>("v_fma_f32", d);
    run<1, false>("v_pk_fma_f32", d);    run<1, true>("v_pk_fma_f32", d);
    run<4, false>("v_add_f32", d);       run<4, true>("v_add_f32", d);
    run<3, false>("v_pk_add_f32", d);    run<3, true>("v_pk_add_f32", d);
    run<6, false>("v_pk_mul_f32", d);    run<6, true>("v_pk_mul_f32", d);
    run<2, false>("v_exp_f32", d);       run<2, true>("v_exp_f32", d);
    run<7, false>("v_rcp_f32", d);       run<7, true>("v_rcp_f32", d);
    run<5, false>("v_cvt_pk_bf16_f32", d); run<5, true>("v_cvt_pk_bf16_f32", d);
    run<9, false>("v_lshlrev_b32", d);   run<9, true>("v_lshlrev_b32", d);
    run<10, false>("v_dot2c_f32_bf16", d); run<10, true>("v_dot2c_f32_bf16", d);
    run<11, false>("v_dot2_f32_bf16", d);  run<11, true>("v_dot2_f32_bf16", d);
    run<12, false>("v_dot2c chain", d);    run<12, true>("v_dot2c chain", d);
    run<13, false>("v_add_f32 chain", d);  run<13, true>("v_add_f32 chain", d);
    return 0;
}
