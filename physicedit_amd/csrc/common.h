// Shared device/host helpers for the gfx950 (CDNA4) kernels of the PhysicEdit hot path.
// Wave size is 64 everywhere (hard-coded: gfx950 only, no dual paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pe {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define PE_DEV __device__ __forceinline__

// round-to-nearest-even f32 -> bf16 -> f32: the reference's "every op rounds to bf16" boundary
PE_DEV float bf16r(float x) { return (float)(bf16)x; }
PE_DEV float b2f(bf16 x) { return (float)x; }

PE_DEV int lane_id() { return (int)(threadIdx.x & 63); }
PE_DEV int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// async global -> LDS copy, 16 B per lane; LDS destination = wave-uniform base + lane*16
PE_DEV void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

PE_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// bijective XCD-aware remap of a linear workgroup id (8 XCDs, block b runs on XCD b % 8):
// gives every XCD a contiguous chunk of the tile order so neighbouring tiles share an L2.
PE_DEV int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// four fp32 -> OCP e4m3fn bytes (v_cvt_pk_fp8_f32 on gfx950, round to nearest even, saturating)
PE_DEV uint32_t pack4_e4m3(float a, float b, float c, float d) {
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (uint32_t)v;
}

// exact-erf GELU / sigmoid helpers in fp32
PE_DEV float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

}  // namespace pe

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
#define PE_OK 0
#define PE_ERR_INVALID_ARG (-1)
#define PE_ERR_UNSUPPORTED (-2)
#define PE_ERR_HIP (-3)

namespace pe {
int set_error(int code, const char* fmt, ...);
int check_launch(const char* what);
}  // namespace pe

#define PE_REQUIRE(cond, ...)                                             \
    do {                                                                  \
        if (!(cond)) return pe::set_error(PE_ERR_INVALID_ARG, __VA_ARGS__); \
    } while (0)
