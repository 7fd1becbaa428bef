// VAE encode/decode kernels for gfx950 (QwenImageVAE on single frames).
//
// Reference: DiffSynth-Studio/diffsynth/models/qwen_image_vae.py.  With feat_cache=None and T=1 every
// QwenImageCausalConv3d (:8-50) sees two zero frames in front, so it IS a 2-D 3x3 conv with
// weight[:, :, 2] -- these kernels compute exactly that, 1/3 of the nominal MACs.
//
// Activations are NHWC bf16 with the channel count padded to a multiple of 32 ("Cp", pad lanes are
// zero); a conv is an implicit GEMM  out[pixel][cout] = sum_{tap,cin} in[pixel+tap][cin] * W[cout][tap][cin]:
//   * block = 256 output pixels x 96 output channels, 4 waves (64 px x 96 co each = 2x3
//     v_mfma_f32_32x32x16_bf16 tiles), K tiles of 32 input channels of one tap;
//   * both operands go HBM -> LDS by LDS-DMA; the per-lane SOURCE address does the im2col gather
//     (halo taps read a zero page, nearest-exact 2x upsampling reads pixel (y>>1, x>>1), stride-2
//     reads (2y+dy, 2x+dx)), so no im2col buffer and no upsampled tensor ever exist in HBM;
//   * epilogue: bias -> bf16 (the conv's own rounding) -> optional residual add -> bf16, transposed
//     through LDS so stores are 16 B per lane along channels.
#include <atomic>
#include "common.h"
#include "kernels.h"

namespace pe {

// ================================================================================================
// implicit-GEMM conv
// ================================================================================================
constexpr int CV_BM = 256, CV_BN = 96, CV_BK = 32, CV_THREADS = 256;
constexpr int CV_STAGE = (CV_BM + CV_BN) * CV_BK * 2;  // 22 KiB
constexpr int CV_LDS = 4 * 64 * CV_BN * 2;             // 48 KiB epilogue tile >= 2 stages (44 KiB)

struct ConvArgs {
    const bf16* in;    // [Hin*Win][Cin_p]
    const bf16* w;     // [Cout_p][taps][Cin_p]
    const bf16* bias;  // [Cout_p]
    const bf16* res;   // [Hout*Wout][Cout_p] or null
    bf16* out;         // [Hout*Wout][Cout_p]
    const bf16* zero;  // >= 64 B of zeros
    int Hin, Win, Hout, Wout, Cin_p, Cout_p;
    int ksize, stride, pad, upsample;
};

__global__ void __launch_bounds__(CV_THREADS, 2) conv_nhwc_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id();
    const int w = wave_id();
    const int l31 = lane & 31, h = lane >> 5;
    const int npix = a.Hout * a.Wout;
    const int tilesN = (a.Cout_p + CV_BN - 1) / CV_BN;
    const int bid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int tn = bid % tilesN;
    const int p0 = (bid / tilesN) * CV_BM;
    const int n0 = tn * CV_BN;
    const int taps = a.ksize * a.ksize;
    const int cpt = a.Cin_p / CV_BK;   // k tiles per tap
    const int nk = taps * cpt;
    const int Hc = a.upsample ? a.Hin * 2 : a.Hin;  // conv-input grid
    const int Wc = a.upsample ? a.Win * 2 : a.Win;

    // A staging: 16 pieces (16 rows x 64 B); wave w moves pieces 4w..4w+3
    int ay[4], ax[4];
    const int a_slot = lane & 3;
    int a_chunk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (w * 4 + i) * 16 + (lane >> 2);
        const int p = min(p0 + row, npix - 1);
        ay[i] = (p / a.Wout) * a.stride - a.pad;
        ax[i] = (p % a.Wout) * a.stride - a.pad;
        a_chunk[i] = a_slot ^ ((row >> 2) & 3);
    }
    // W staging: 6 pieces (16 rows x 64 B): waves 0,1 move 2, waves 2,3 move 1
    const int nwp = w < 2 ? 2 : 1;
    const int wp0 = w < 2 ? w * 2 : 2 + w;
    const bf16* w_src[2];
    const int Ktot = taps * a.Cin_p;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wp0 + i) * 16 + (lane >> 2);
        const int chunk = a_slot ^ ((row >> 2) & 3);
        const int gn = min(n0 + row, a.Cout_p - 1);
        w_src[i] = a.w + (size_t)gn * Ktot + chunk * 8;
    }
    auto stage = [&](int buf, int kt) {
        const int tap = kt / cpt;
        const int c0 = (kt - tap * cpt) * CV_BK;
        const int dy = tap / a.ksize, dx = tap - dy * a.ksize;
        char* base = smem + buf * CV_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int iy = ay[i] + dy, ix = ax[i] + dx;
            const bool ok = (unsigned)iy < (unsigned)Hc && (unsigned)ix < (unsigned)Wc;
            const int sy = a.upsample ? iy >> 1 : iy, sx = a.upsample ? ix >> 1 : ix;
            const bf16* src = ok ? a.in + ((size_t)sy * a.Win + sx) * a.Cin_p + c0 + a_chunk[i] * 8 : a.zero;
            glds16(src, base + (w * 4 + i) * 1024);
        }
        for (int i = 0; i < nwp; ++i) glds16(w_src[i] + kt * CV_BK, base + CV_BM * CV_BK * 2 + (wp0 + i) * 1024);
    };

    f32x16 acc[2][3];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 3; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int sw = (l31 >> 2) & 3;
    const int a_off = (w * 64 + l31) * 64;
    const int w_off = CV_BM * CV_BK * 2 + l31 * 64;

    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
        const char* S = smem + (kt & 1) * CV_STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int coff = ((kk * 2 + h) ^ sw) << 4;
            bf16x8 af[2], wf[3];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) af[mi] = *(const bf16x8*)(S + a_off + mi * 32 * 64 + coff);
#pragma unroll
            for (int ni = 0; ni < 3; ++ni) wf[ni] = *(const bf16x8*)(S + w_off + ni * 32 * 64 + coff);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 3; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], af[mi], acc[mi][ni], 0, 0, 0);
        }
    }

    // epilogue: acc[mi][ni][4q+r] = out[p0 + w*64 + mi*32 + l31][n0 + ni*32 + 8q + 4h + r]
    __syncthreads();
    char* E = smem + w * (64 * CV_BN * 2);  // [64 rows][96 cols] bf16
#pragma unroll
    for (int ni = 0; ni < 3; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = n0 + ni * 32 + 8 * q + 4 * h;
            float b[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.bias != nullptr && n < a.Cout_p) {
                const bf16x4 bv = *(const bf16x4*)(a.bias + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) b[r] = (float)bv[r];
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                bf16x4 y;
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = (bf16)(acc[mi][ni][4 * q + r] + b[r]);
                *(bf16x4*)(E + (mi * 32 + l31) * (CV_BN * 2) + (ni * 4 + q) * 16 + h * 8) = y;
            }
        }
    __syncthreads();
#pragma unroll 4
    for (int it = 0; it < 12; ++it) {
        const int item = it * 64 + lane;
        const int row = item / 12, c = item - row * 12;
        const int p = p0 + w * 64 + row;
        const int n = n0 + c * 8;
        if (p >= npix || n >= a.Cout_p) continue;
        bf16x8 v = *(const bf16x8*)(E + row * (CV_BN * 2) + c * 16);
        if (a.res != nullptr) {
            const bf16x8 rv = *(const bf16x8*)(a.res + (size_t)p * a.Cout_p + n);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (bf16)((float)v[j] + (float)rv[j]);  // x + h (:152)
        }
        *(bf16x8*)(a.out + (size_t)p * a.Cout_p + n) = v;
    }
}

int launch_conv_nhwc(const void* in, const void* w, const void* bias, const void* res, void* out, const void* zero,
                     int Hin, int Win, int Cin_p, int Cout_p, int ksize, int stride, int upsample,
                     hipStream_t stream) {
    PE_REQUIRE(in && w && out && zero, "conv: null pointer");
    PE_REQUIRE(Hin > 0 && Win > 0, "conv: empty input");
    PE_REQUIRE(Cin_p % 32 == 0 && Cout_p % 32 == 0, "conv: channel counts must be padded to 32 (Cin_p=%d Cout_p=%d)", Cin_p, Cout_p);
    PE_REQUIRE(ksize == 1 || ksize == 3, "conv: ksize=%d", ksize);
    PE_REQUIRE(stride == 1 || (stride == 2 && ksize == 3 && !upsample), "conv: stride=%d", stride);
    PE_REQUIRE(!upsample || (ksize == 3 && stride == 1), "conv: upsample only with 3x3 stride 1");
    ConvArgs a;
    a.in = (const bf16*)in; a.w = (const bf16*)w; a.bias = (const bf16*)bias; a.res = (const bf16*)res;
    a.out = (bf16*)out; a.zero = (const bf16*)zero;
    a.Hin = Hin; a.Win = Win; a.Cin_p = Cin_p; a.Cout_p = Cout_p; a.ksize = ksize; a.stride = stride;
    a.upsample = upsample;
    if (stride == 2) {
        // ZeroPad2d((0,1,0,1)) + Conv2d(3, stride 2)  (:249,251)
        PE_REQUIRE(Hin % 2 == 0 && Win % 2 == 0, "conv: stride-2 input must be even");
        a.pad = 0; a.Hout = Hin / 2; a.Wout = Win / 2;
    } else {
        a.pad = ksize == 3 ? 1 : 0;
        a.Hout = upsample ? Hin * 2 : Hin; a.Wout = upsample ? Win * 2 : Win;
    }
    static std::atomic<bool> configured{false};   // racing first calls both configure: idempotent
    if (!configured.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_nhwc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CV_LDS);
        if (e != hipSuccess) return set_error(PE_ERR_HIP, "conv: hipFuncSetAttribute: %s", hipGetErrorString(e));
        configured.store(true, std::memory_order_release);
    }
    const int npix = a.Hout * a.Wout;
    const int grid = ((npix + CV_BM - 1) / CV_BM) * ((Cout_p + CV_BN - 1) / CV_BN);
    const int slot = prof_begin(PROF_CONV, 2.0 * npix * (double)Cout_p * ksize * ksize * Cin_p, stream);
    hipLaunchKernelGGL(conv_nhwc_kernel, dim3(grid), dim3(CV_THREADS), CV_LDS, stream, a);
    prof_end(slot, stream);
    return check_launch("conv_nhwc_kernel");
}

// ================================================================================================
// QwenImageRMS_norm (:76-77) [+ SiLU]:  F.normalize(x, dim=C) * sqrt(C) * gamma (+ 0.0), bf16 graph:
//   nrm = bf16(sqrt(sum x^2)); y = bf16(x / max(nrm, 1e-12)); y = bf16(y * f32(sqrt(C)));
//   y = bf16(y * gamma); [y = bf16(silu(y))]          (rounding model pinned by tests/golden G7)
// 16 lanes per pixel (<= 3 x 16-B chunks per lane), 4 pixels per wave.
// ================================================================================================
__global__ void __launch_bounds__(256) vae_rmsnorm_kernel(const bf16* __restrict__ x, const bf16* __restrict__ gamma,
                                                          bf16* __restrict__ out, int npix, int C, int Cp,
                                                          float sqrtC, int silu) {
    const int lane = lane_id();
    const int j = lane & 15;
    const int p = ((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6)) * 4 + (lane >> 4);
    const bool live = p < npix;
    const int nch = C >> 3;  // valid 16-B chunks per pixel (C % 8 == 0)
    const bf16* xp = x + (size_t)(live ? p : 0) * Cp;
    float v[3][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = j + i * 16;
        if (c < nch) {
            const bf16x8 t = *(const bf16x8*)(xp + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[i][e] = (float)t[e];
                ss += v[i][e] * v[i][e];
            }
        }
    }
    ss += __shfl_xor(ss, 1, 64);
    ss += __shfl_xor(ss, 2, 64);
    ss += __shfl_xor(ss, 4, 64);
    ss += __shfl_xor(ss, 8, 64);
    const float nrm = fmaxf(bf16r(sqrtf(ss)), 1e-12f);
    if (!live) return;
    bf16* op = out + (size_t)p * Cp;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = j + i * 16;
        if (c < nch) {
            const bf16x8 g = *(const bf16x8*)(gamma + c * 8);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float y = bf16r(v[i][e] / nrm);
                y = bf16r(y * sqrtC);
                y = bf16r(y * (float)g[e]);
                if (silu) y = y / (1.0f + __expf(-y));
                o[e] = (bf16)y;
            }
            *(bf16x8*)(op + c * 8) = o;
        }
    }
}

int launch_vae_rmsnorm(const void* x, const void* gamma, void* out, int npix, int C, int Cp, int silu,
                       hipStream_t stream) {
    PE_REQUIRE(x && gamma && out, "vae_rmsnorm: null pointer");
    PE_REQUIRE(npix > 0 && C % 8 == 0 && C <= 384 && Cp >= C && Cp % 8 == 0, "vae_rmsnorm: bad shape C=%d Cp=%d", C, Cp);
    const float sqrtC = (float)sqrt((double)C);  // python float dim**0.5 -> fp32 opmath scalar
    const int grid = (npix + 15) / 16;
    const int slot = prof_begin(PROF_ROW, 4.0 * (double)npix * C, stream);
    hipLaunchKernelGGL(vae_rmsnorm_kernel, dim3(grid), dim3(256), 0, stream, (const bf16*)x, (const bf16*)gamma,
                       (bf16*)out, npix, C, Cp, sqrtC, silu);
    prof_end(slot, stream);
    return check_launch("vae_rmsnorm_kernel");
}

// ================================================================================================
// layout converters
// ================================================================================================
// NCHW [C][HW] -> NHWC [HW][Cp] (pad channels zero), optional latent de-normalisation
//   decode: x / std_inv + mean  (:723-724), both tables already rounded to bf16
// mode 0: copy; mode 1: (x / b) + a   [a = mean, b = std_inv]
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const bf16* __restrict__ in, bf16* __restrict__ out,
                                                           int C, int HW, int Cp, int mode,
                                                           const bf16* __restrict__ ta, const bf16* __restrict__ tb) {
    const size_t total = (size_t)HW * Cp;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % Cp);
        const size_t p = i / Cp;
        float v = 0.f;
        if (c < C) {
            v = (float)in[(size_t)c * HW + p];
            if (mode == 1) v = bf16r(bf16r(v / (float)tb[c]) + (float)ta[c]);
        }
        out[i] = (bf16)v;
    }
}

// NHWC [HW][Cp] -> NCHW [C][HW], optional latent normalisation
//   encode: (x - mean) * std_inv  (:714)
// mode 0: copy; mode 2: (x - a) * b
__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const bf16* __restrict__ in, bf16* __restrict__ out,
                                                           int C, int HW, int Cp, int mode,
                                                           const bf16* __restrict__ ta, const bf16* __restrict__ tb) {
    const size_t total = (size_t)HW * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t p = i % HW;
        const int c = (int)(i / HW);
        float v = (float)in[p * Cp + c];
        if (mode == 2) v = bf16r(bf16r(v - (float)ta[c]) * (float)tb[c]);
        out[i] = (bf16)v;
    }
}

int launch_nchw_to_nhwc(const void* in, void* out, int C, int HW, int Cp, int mode, const void* ta, const void* tb,
                        hipStream_t stream) {
    PE_REQUIRE(in && out && C > 0 && HW > 0 && Cp >= C, "nchw_to_nhwc: bad arguments");
    PE_REQUIRE(mode == 0 || (mode == 1 && ta && tb), "nchw_to_nhwc: mode %d", mode);
    const size_t total = (size_t)HW * Cp;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid), dim3(256), 0, stream, (const bf16*)in, (bf16*)out, C, HW, Cp,
                       mode, (const bf16*)ta, (const bf16*)tb);
    return check_launch("nchw_to_nhwc_kernel");
}

int launch_nhwc_to_nchw(const void* in, void* out, int C, int HW, int Cp, int mode, const void* ta, const void* tb,
                        hipStream_t stream) {
    PE_REQUIRE(in && out && C > 0 && HW > 0 && Cp >= C, "nhwc_to_nchw: bad arguments");
    PE_REQUIRE(mode == 0 || (mode == 2 && ta && tb), "nhwc_to_nchw: mode %d", mode);
    const size_t total = (size_t)HW * C;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid), dim3(256), 0, stream, (const bf16*)in, (bf16*)out, C, HW, Cp,
                       mode, (const bf16*)ta, (const bf16*)tb);
    return check_launch("nhwc_to_nchw_kernel");
}

// ================================================================================================
// mid-block attention: one head, D = 384  (QwenImageAttentionBlock.forward :173-198, SDPA :187)
//   qkv [N][1152] bf16 token-major (q | k | v) from the 1x1 to_qkv conv; out [N][384].
// Same transposed-product formulation as attention.hip, sized for D=384: one wave per SIMD (the
// 192 fp32 O accumulators + 96 Q registers need the whole 512-entry file), 4 waves x 32 queries per
// work-group, K / Vt tiles of 32 keys double-buffered in LDS.
// ================================================================================================
constexpr int VA_D = 384, VA_KV = 32, VA_THREADS = 256;
constexpr int VA_STAGE = 2 * VA_KV * VA_D * 2;  // K 24 KiB + Vt 24 KiB
constexpr int VA_LDS = 2 * VA_STAGE;            // 96 KiB

// Vt[d][pos(s)] from the v columns of qkv (pos = perm16 inside aligned 16-groups; pad columns zero)
__global__ void __launch_bounds__(256) vae_vt_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ vt, int N, int Np) {
    const size_t total = (size_t)VA_D * Np;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int pos = (int)(i % Np);
        const int d = (int)(i / Np);
        const int j = pos & 15;
        const int s = (pos & ~15) | ((j & 3) | ((j & 4) << 1) | ((j & 8) >> 1));  // perm16 is an involution
        vt[i] = s < N ? qkv[(size_t)s * (3 * VA_D) + 2 * VA_D + d] : (bf16)0.f;
    }
}

// Round 6: (a) the keys are SPLIT over gridDim.y work-groups per query block (flash-decoding style: 128 query blocks at 1024 x 1024 filled
// half of the 256 CUs) -- a part writes its un-normalised (O, m, l) in fp32 and vae_attn_combine_kernel merges them; (b) the 192 O
// accumulators are rescaled only in tiles where some row of the wave raised its maximum (alpha = 1 for every row otherwise: the
// multiply it skips is exact), which after the first tiles is almost never.
__global__ void __launch_bounds__(VA_THREADS, 1)
vae_attn_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ Vt, bf16* __restrict__ out, int N, int Np,
                float scale_log2, float* __restrict__ part_o, float* __restrict__ part_ml) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id();
    const int w = wave_id();
    const int l31 = lane & 31, h = lane >> 5;
    const int q0 = (int)blockIdx.x * 128 + w * 32;

    bf16x8 qf[24];
    {
        const int qrow = min(q0 + l31, N - 1);
        const bf16* qp = qkv + (size_t)qrow * (3 * VA_D) + h * 8;
#pragma unroll
        for (int kk = 0; kk < 24; ++kk) qf[kk] = *(const bf16x8*)(qp + kk * 16);
    }
    // staging: K tile 32 rows x 768 B = 24 pieces, Vt tile 384 rows x 64 B = 24 pieces; 6 + 6 per wave
    const bf16* k_src[6];
    const bf16* v_src[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int piece = w * 6 + i;
        const int o = piece * 1024 + lane * 16;      // byte offset inside the K tile image
        const int krow = o / 768, slot = (o - krow * 768) >> 4;
        const int kchunk = (slot & ~15) | ((slot ^ krow) & 15);
        k_src[i] = qkv + (size_t)krow * (3 * VA_D) + VA_D + kchunk * 8;   // + t*32 rows at stage time (clamped)
        const int vrow = piece * 16 + (lane >> 2);
        const int vchunk = (lane & 3) ^ ((vrow >> 2) & 3);
        v_src[i] = Vt + (size_t)vrow * Np + vchunk * 8;
    }
    const int nt_all = (N + VA_KV - 1) / VA_KV;
    const int nsplit = (int)gridDim.y, part = (int)blockIdx.y;
    const int t_begin = (int)((long long)nt_all * part / nsplit), t_end = (int)((long long)nt_all * (part + 1) / nsplit);
    auto stage = [&](int buf, int t) {
        char* base = smem + buf * VA_STAGE + w * 6144;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            // K rows beyond N (last tile) are clamped to a valid row; their scores are masked
            const int o = (w * 6 + i) * 1024 + lane * 16;
            const int krow = o / 768;
            const int grow = min(t * VA_KV + krow, N - 1) - krow;
            glds16(k_src[i] + (size_t)grow * (3 * VA_D), base + i * 1024);
            glds16(v_src[i] + t * VA_KV, base + VA_KV * VA_D * 2 + i * 1024);
        }
    };

    f32x16 o[12];
#pragma unroll
    for (int dt = 0; dt < 12; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int k_off = l31 * 768;
    const int ksw = l31 & 15;
    const int v_off = VA_KV * VA_D * 2 + l31 * 64;
    const int vsw = (l31 >> 2) & 3;

    stage(0, t_begin);
    for (int t = t_begin; t < t_end; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < t_end) stage((t - t_begin + 1) & 1, t + 1);
        const char* Sb = smem + ((t - t_begin) & 1) * VA_STAGE;
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 24; ++kk) {
            const int c = kk * 2 + h;
            const bf16x8 kf = *(const bf16x8*)(Sb + k_off + (((c & ~15) | ((c ^ ksw) & 15)) << 4));
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], st, 0, 0, 0);
            if ((kk & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // keep <= 4 K fragments in flight (register budget)
        }
        if (t == nt_all - 1 && (N & (VA_KV - 1)) != 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * VA_KV + 8 * (r >> 2) + 4 * h + (r & 3);
                if (key >= N) st[r] = -INFINITY;
            }
        }
        float mx = st[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * scale_log2);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], scale_log2, -m_new));
            st[r] = p;
            psum += p;
        }
        l_run = __builtin_fmaf(l_run, alpha, psum);
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {      // wave-uniform: some row's maximum moved (x 1.0f elsewhere: exact)
            // O lives in the accumulator half (the P.V MFMAs below are asm with "+a" operands: with builtins hipcc moved the 192 values through
            // the arch half every tile -- 480 v_accvgpr moves and 22 spill reloads per 48 MFMAs, each reload behind an s_waitcnt vmcnt(0) that
            // also waits for the NEXT tile's LDS-DMA: 4.7 us per tile).  Here, rarely, they are read out, scaled and written back; the empty
            // statement re-defines every tuple at this point, so no copy can be hoisted to right behind an MFMA the compiler does not know
            // about (the last P.V MFMA is 24 QK^T MFMAs old by now).
#pragma unroll
            for (int dt = 0; dt < 12; ++dt) {
                asm volatile("" : "+a"(o[dt]));
                o[dt] *= alpha;
                asm volatile("" : "+a"(o[dt]));
            }
        }
        // P of both 16-key chunks is packed BEFORE the first P.V MFMA.  The MFMAs are asm statements ("+a": O stays in the accumulator half),
        // so three things are nobody's job but this code's (profiles/r04_attention_notes.md section 5): a VALU result needs wait states
        // before an MFMA reads it as an operand (s_nop in front of the first one); an MFMA reads its A / B registers while it RUNS, so they
        // must not be handed to another instruction before the next MFMA of the wave has issued (the empty "keep" statements); and nothing
        // may read O before the last MFMA has drained (the s_nop pair + tuple re-definitions behind the loop).
        bf16x8 pf2[2];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pf2[k2][e] = (bf16)st[(2 * k2) * 4 + e];
                pf2[k2][4 + e] = (bf16)st[(2 * k2 + 1) * 4 + e];
            }
        asm volatile("s_nop 4" : "+v"(pf2[0]), "+v"(pf2[1]));
        bf16x8 vprev = pf2[0];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int vchunk = k2 * 2 + h;
#pragma unroll
            for (int dt = 0; dt < 12; ++dt) {
                const bf16x8 vf = *(const bf16x8*)(Sb + v_off + dt * 32 * 64 + ((vchunk ^ vsw) << 4));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o[dt]) : "v"(vf), "v"(pf2[k2]));
                asm volatile("" : : "v"(vprev));      // keep: the fragment of the MFMA before this one
                vprev = vf;
                if ((dt & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_nop 7" : : "v"(vprev), "v"(pf2[0]), "v"(pf2[1]));      // keep the last operands until the last MFMA has read them
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");      // the last P.V MFMAs (asm: invisible to the hazard recognizer) -> reads of O
#pragma unroll
    for (int dt = 0; dt < 12; ++dt) asm volatile("" : "+a"(o[dt]));      // (volatile statements keep their order: every read of O is behind the drain)
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const int q = q0 + l31;
    if (nsplit > 1) {
        // un-normalised partial: O [part][q][384] fp32 in the output's column order, (m, l) [part][q][2]
        if (q < N) {
            float* po = part_o + ((size_t)part * N + q) * VA_D + 4 * h;
#pragma unroll
            for (int dt = 0; dt < 12; ++dt)
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    *(f32x4*)(po + dt * 32 + 8 * a) = f32x4{o[dt][4 * a], o[dt][4 * a + 1], o[dt][4 * a + 2], o[dt][4 * a + 3]};
            if (h == 0) {
                part_ml[((size_t)part * N + q) * 2] = m_run;
                part_ml[((size_t)part * N + q) * 2 + 1] = l_tot;
            }
        }
        return;
    }
    const float inv = 1.0f / l_tot;
    if (q < N) {
        bf16* op = out + (size_t)q * VA_D + 4 * h;
#pragma unroll
        for (int dt = 0; dt < 12; ++dt)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                bf16x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (bf16)(o[dt][4 * a + r] * inv);
                *(bf16x4*)(op + dt * 32 + 8 * a) = v;
            }
    }
}

// O = sum_i O_i 2^(m_i - M) / sum_i l_i 2^(m_i - M): one thread = four columns of one query row
__global__ void __launch_bounds__(256) vae_attn_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                               bf16* __restrict__ out, int N, int nsplit) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t q = e / (VA_D / 4);
    const int c = (int)(e - q * (VA_D / 4));
    if (q >= (size_t)N) return;
    float M = -INFINITY;
    for (int i = 0; i < nsplit; ++i) M = fmaxf(M, part_ml[((size_t)i * N + q) * 2]);
    float L = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < nsplit; ++i) {
        const float wgt = __builtin_amdgcn_exp2f(part_ml[((size_t)i * N + q) * 2] - M);
        L = __builtin_fmaf(part_ml[((size_t)i * N + q) * 2 + 1], wgt, L);
        const f32x4 v = *(const f32x4*)(part_o + ((size_t)i * N + q) * VA_D + c * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(v[j], wgt, acc[j]);
    }
    const float inv = 1.0f / L;
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (bf16)(acc[j] * inv);
    *(bf16x4*)(out + q * VA_D + c * 4) = o;
}

// how many ways the keys are split: as many as fill the CUs once (the combine's traffic grows with every part)
static int vae_attention_split(int N) {
    const int nqb = (N + 127) / 128, nt = (N + VA_KV - 1) / VA_KV;
    int s = 256 / nqb;
    if (s > 4) s = 4;
    if (s > nt) s = nt;
    return s < 1 ? 1 : s;
}

size_t vae_attention_scratch_bytes(int N) {
    if (N <= 0) return 0;
    const size_t Np = (size_t)(N + 31) / 32 * 32;
    const size_t vt = ((size_t)VA_D * Np * 2 + 255) / 256 * 256;
    const int s = vae_attention_split(N);
    return vt + (s > 1 ? (size_t)s * N * (VA_D + 2) * sizeof(float) : 0);
}

int launch_vae_attention(const void* qkv, void* vt_scratch, void* out, int N, hipStream_t stream) {
    PE_REQUIRE(qkv && vt_scratch && out && N > 0, "vae_attention: bad arguments");
    PE_REQUIRE(((uintptr_t)vt_scratch & 255) == 0, "vae_attention: the scratch must be 256-byte aligned");
    const int Np = (N + 31) / 32 * 32;
    const int nsplit = vae_attention_split(N);
    float* part_o = (float*)((char*)vt_scratch + ((size_t)VA_D * Np * 2 + 255) / 256 * 256);
    float* part_ml = part_o + (size_t)nsplit * N * VA_D;
    {
        const size_t total = (size_t)VA_D * Np;
        const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        hipLaunchKernelGGL(vae_vt_kernel, dim3(grid), dim3(256), 0, stream, (const bf16*)qkv, (bf16*)vt_scratch, N, Np);
        int rc = check_launch("vae_vt_kernel");
        if (rc) return rc;
    }
    static std::atomic<bool> configured{false};   // racing first calls both configure: idempotent
    if (!configured.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void*)vae_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, VA_LDS);
        if (e != hipSuccess) return set_error(PE_ERR_HIP, "vae_attention: hipFuncSetAttribute: %s", hipGetErrorString(e));
        configured.store(true, std::memory_order_release);
    }
    const float scale_log2 = (float)(1.0 / sqrt((double)VA_D)) * 1.44269504088896340736f;
    const int slot = prof_begin(PROF_ATTN, 4.0 * (double)N * N * VA_D, stream);
    hipLaunchKernelGGL(vae_attn_kernel, dim3((N + 127) / 128, nsplit), dim3(VA_THREADS), VA_LDS, stream, (const bf16*)qkv,
                       (const bf16*)vt_scratch, (bf16*)out, N, Np, scale_log2, part_o, part_ml);
    int rc = check_launch("vae_attn_kernel");
    if (rc == PE_OK && nsplit > 1) {
        hipLaunchKernelGGL(vae_attn_combine_kernel, dim3((unsigned)(((size_t)N * (VA_D / 4) + 255) / 256)), dim3(256), 0, stream, (const float*)part_o,
                           (const float*)part_ml, (bf16*)out, N, nsplit);
        rc = check_launch("vae_attn_combine_kernel");
    }
    prof_end(slot, stream);
    return rc;
}

}  // namespace pe
