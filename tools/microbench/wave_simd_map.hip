#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(512, 2) k(unsigned* out) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
}
int main() {
    unsigned* d; unsigned h[8 * 4];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(4), dim3(512), 65536, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 4; ++b) {
        printf("block %d:", b);
        for (int w = 0; w < 8; ++w) printf("  w%d: simd %u cu %u wave_slot %u", w, (h[b * 8 + w] >> 4) & 3, (h[b * 8 + w] >> 8) & 15, h[b * 8 + w] & 15);
        printf("\n");
    }
    return 0;
}
