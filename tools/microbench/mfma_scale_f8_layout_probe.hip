// Probe: operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x e4m3, unit scales) on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void k(const uint8_t* A /*[32][64]*/, const uint8_t* B /*[32 n][64 k]*/, float* C /*[32][32]*/) {
    const int l = threadIdx.x, l31 = l & 31, h = l >> 5;
    i32x8 a, b;
    const int* ap = (const int*)(A + l31 * 64 + h * 32);
    const int* bp = (const int*)(B + l31 * 64 + h * 32);
    for (int i = 0; i < 8; ++i) { a[i] = ap[i]; b[i] = bp[i]; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;   // i index (A rows)
        C[row * 32 + l31] = c[r];                          // j index (B "rows" = n)
    }
}

// v_mfma_scale_f32_16x16x128_f8f6f4: guess lane l: row l & 15, k = 32 (l >> 4) .. + 31 contiguous bytes; D: lane l holds i = 4 (l >> 4) + r, j = l & 15
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k16(const uint8_t* A /*[16][128]*/, const uint8_t* B /*[16 n][128 k]*/, float* C /*[16][16]*/) {
    const int l = threadIdx.x, l15 = l & 15, g = l >> 4;
    i32x8 a, b;
    const int* ap = (const int*)(A + l15 * 128 + g * 32);
    const int* bp = (const int*)(B + l15 * 128 + g * 32);
    for (int i = 0; i < 8; ++i) { a[i] = ap[i]; b[i] = bp[i]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int r = 0; r < 4; ++r) C[(4 * g + r) * 16 + l15] = c[r];
}

static float e4m3_to_float(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f;
    if (e == 0) f = ldexpf((float)m / 8.0f, -6);
    else if (e == 15 && m == 7) f = NAN;
    else f = ldexpf(1.0f + (float)m / 8.0f, e - 7);
    return s ? -f : f;
}

int main() {
    uint8_t hA[32 * 64], hB[32 * 64];
    srand(1);
    for (int i = 0; i < 32 * 64; ++i) {
        do { hA[i] = rand() & 0xff; } while ((hA[i] & 0x7f) == 0x7f || (hA[i] & 0x78) > 0x48);   // no NaN, |x| <= ~4
        do { hB[i] = rand() & 0xff; } while ((hB[i] & 0x7f) == 0x7f || (hB[i] & 0x78) > 0x48);
    }
    uint8_t *dA, *dB; float* dC; float hC[32 * 32];
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double ref = 0;
        for (int kk = 0; kk < 64; ++kk) ref += (double)e4m3_to_float(hA[i * 64 + kk]) * e4m3_to_float(hB[j * 64 + kk]);
        maxerr = fmax(maxerr, fabs(ref - hC[i * 32 + j])); maxref = fmax(maxref, fabs(ref));
    }
    printf("layout guess (lane l: row l&31, k = 32*(l>>5)..+31 contiguous bytes): max err %.6g (max |ref| %.4g) -> %s\n", maxerr, maxref,
           maxerr < 1e-3 * maxref ? "MATCH" : "MISMATCH");
    printf("C[0][0..3] = %g %g %g %g\n", hC[0], hC[1], hC[2], hC[3]);
    // 16 x 16 x 128: the same 2048 bytes read as [16][128]
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(hC, dC, 16 * 16 * 4, hipMemcpyDeviceToHost);
    maxerr = 0; maxref = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double ref = 0;
        for (int kk = 0; kk < 128; ++kk) ref += (double)e4m3_to_float(hA[i * 128 + kk]) * e4m3_to_float(hB[j * 128 + kk]);
        maxerr = fmax(maxerr, fabs(ref - hC[i * 16 + j])); maxref = fmax(maxref, fabs(ref));
    }
    printf("16x16x128 layout guess (lane l: row l&15, k = 32*(l>>4)..+31; D[4(l>>4)+r][l&15]): max err %.6g (max |ref| %.4g) -> %s\n", maxerr, maxref,
           maxerr < 1e-3 * maxref ? "MATCH" : "MISMATCH");
    return 0;
}
