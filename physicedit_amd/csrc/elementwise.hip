// HBM-bound row / element-wise kernels of the DiT forward and the sampler (gfx950).
// Every arithmetic step rounds to bf16 exactly where the reference's bf16 torch graph does
// (SURVEY.md Appendix A); loads and stores are 16 B per lane.
#include <atomic>
#include "common.h"
#include "kernels.h"

namespace pe {


// ------------------------------------------------------------------------------------------------
// LayerNorm(elementwise_affine=False, eps) -> x*(1+scale) + shift
//   QwenImageTransformerBlock._modulate   qwen_image_dit.py:355-357, norms :337,344,351,352
//   AdaLayerNorm(single)                  models/utils.py:304-309
// One wave per row (dim = 3072: 6 x 16 B per lane, row kept in registers, two-pass moments in fp32).
// ------------------------------------------------------------------------------------------------
template <int VPL, bool QUANT>  // 16-B vectors per lane; dim = VPL * 512.  QUANT: also emit the row as e4m3 + scale
__global__ void __launch_bounds__(256) ln_modulate_kernel(const bf16* __restrict__ x, bf16* __restrict__ out,
                                                          int rows, int dim, int rows_a,
                                                          const bf16* __restrict__ shift_a,
                                                          const bf16* __restrict__ scale_a,
                                                          const bf16* __restrict__ shift_b,
                                                          const bf16* __restrict__ scale_b, float eps,
                                                          uint8_t* __restrict__ q_out, float* __restrict__ q_scale) {
    const int lane = lane_id();
    const int row = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16* xr = x + (size_t)row * dim;
    float v[VPL][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const bf16x8 t = *(const bf16x8*)(xr + (i * 64 + lane) * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[i][j] = (float)t[j];
            sum += v[i][j];
        }
    }
    const float mean = wave_sum(sum) / (float)dim;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = v[i][j] - mean;
            sq += d * d;
        }
    const float rstd = rsqrtf(wave_sum(sq) / (float)dim + eps);
    const bf16* shift = row < rows_a ? shift_a : shift_b;
    const bf16* scale = row < rows_a ? scale_a : scale_b;
    bf16* orow = out + (size_t)row * dim;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 8;
        const bf16x8 sc = *(const bf16x8*)(scale + c);
        const bf16x8 sh = *(const bf16x8*)(shift + c);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float n = bf16r((v[i][j] - mean) * rstd);   // LayerNorm output (bf16)
            const float s1 = bf16r(1.0f + (float)sc[j]);       // 1 + scale
            const float p = bf16r(n * s1);                     // x * (1 + scale)
            o[j] = (bf16)(p + (float)sh[j]);                   // + shift
            if constexpr (QUANT) v[i][j] = (float)o[j];        // keep the rounded row for the quantiser below
        }
        if (!QUANT || out != nullptr) *(bf16x8*)(orow + c) = o;
    }
    if constexpr (QUANT) {
        // fp8_linear's activation quantisation of this row (see quantize_rows_e4m3_kernel), fused: the row is still in
        // registers, so the consumer GEMM's e4m3 operand costs no extra pass over HBM
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(v[i][j]));
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o2, 64));
        const float scq = fmaxf(bf16r(mx * (1.0f / 448.0f)), 1.0f);
        const float dv = scq + 1e-8f;
        if (lane == 0) q_scale[row] = scq;
        uint8_t* qrow = q_out + (size_t)row * dim;
        if (dv == 1.0f) {
            // scale 1 (every row whose largest magnitude is <= 447, i.e. all of them in practice): 1 + 1e-8f IS 1.0f, x / 1.0f IS x --
            // the same bytes without 48 IEEE divisions per lane (they made this kernel compute-bound: 30 us against 21.7 us for the
            // bf16-only form that moves more bytes)
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                u32x2 pk;
                pk[0] = pack4_e4m3(v[i][0], v[i][1], v[i][2], v[i][3]);
                pk[1] = pack4_e4m3(v[i][4], v[i][5], v[i][6], v[i][7]);
                *(u32x2*)(qrow + (i * 64 + lane) * 8) = pk;
            }
        } else {
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                u32x2 pk;
                pk[0] = pack4_e4m3(v[i][0] / dv, v[i][1] / dv, v[i][2] / dv, v[i][3] / dv);
                pk[1] = pack4_e4m3(v[i][4] / dv, v[i][5] / dv, v[i][6] / dv, v[i][7] / dv);
                *(u32x2*)(qrow + (i * 64 + lane) * 8) = pk;
            }
        }
    }
}

int launch_ln_modulate(const void* x, void* out, int rows, int dim, int rows_a, const void* shift_a,
                       const void* scale_a, const void* shift_b, const void* scale_b, float eps,
                       hipStream_t stream) {
    return launch_ln_modulate_quant(x, out, rows, dim, rows_a, shift_a, scale_a, shift_b, scale_b, eps, nullptr, nullptr,
                                    stream);
}

// q_out != null: additionally (out == null: only) emit the row as e4m3 bytes [rows, dim] + per-row scale (fp8_linear)
int launch_ln_modulate_quant(const void* x, void* out, int rows, int dim, int rows_a, const void* shift_a,
                             const void* scale_a, const void* shift_b, const void* scale_b, float eps, void* q_out,
                             float* q_scale, hipStream_t stream) {
    PE_REQUIRE(x && (out || q_out) && shift_a && scale_a, "ln_modulate: null pointer");
    PE_REQUIRE(q_out == nullptr || q_scale != nullptr, "ln_modulate: q_out without q_scale");
    PE_REQUIRE(rows > 0, "ln_modulate: rows=%d", rows);
    PE_REQUIRE(dim == 3072, "ln_modulate: dim=%d unsupported (DiT width 3072 only)", dim);
    if (!shift_b) shift_b = shift_a;
    if (!scale_b) scale_b = scale_a;
    const double bytes = (2.0 + (out ? 2.0 : 0.0) + (q_out ? 1.0 : 0.0)) * (double)rows * dim;
    const int slot = prof_begin(PROF_ROW, bytes, stream);
    if (q_out)
        hipLaunchKernelGGL((ln_modulate_kernel<6, true>), dim3((rows + 3) / 4), dim3(256), 0, stream, (const bf16*)x,
                           (bf16*)out, rows, dim, rows_a, (const bf16*)shift_a, (const bf16*)scale_a,
                           (const bf16*)shift_b, (const bf16*)scale_b, eps, (uint8_t*)q_out, q_scale);
    else
        hipLaunchKernelGGL((ln_modulate_kernel<6, false>), dim3((rows + 3) / 4), dim3(256), 0, stream, (const bf16*)x,
                           (bf16*)out, rows, dim, rows_a, (const bf16*)shift_a, (const bf16*)scale_a,
                           (const bf16*)shift_b, (const bf16*)scale_b, eps, (uint8_t*)nullptr, (float*)nullptr);
    prof_end(slot, stream);
    return check_launch("ln_modulate_kernel");
}

// ------------------------------------------------------------------------------------------------
// RMSNorm with weight (models/utils.py:250-257): txt_norm over 3584.  One wave per row.
// ------------------------------------------------------------------------------------------------
template <int VPL>
__global__ void __launch_bounds__(256) rmsnorm_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                      bf16* __restrict__ out, int rows, int dim, float eps) {
    const int lane = lane_id();
    const int row = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16* xr = x + (size_t)row * dim;
    float v[VPL][8];
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const bf16x8 t = *(const bf16x8*)(xr + (i * 64 + lane) * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[i][j] = (float)t[j];
            sq += v[i][j] * v[i][j];
        }
    }
    const float rs = rsqrtf(wave_sum(sq) / (float)dim + eps);
    bf16* orow = out + (size_t)row * dim;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 8;
        const bf16x8 wv = *(const bf16x8*)(w + c);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)(bf16r(v[i][j] * rs) * (float)wv[j]);
        *(bf16x8*)(orow + c) = o;
    }
}

int launch_rmsnorm(const void* x, const void* w, void* out, int rows, int dim, float eps, hipStream_t stream) {
    PE_REQUIRE(x && w && out, "rmsnorm: null pointer");
    PE_REQUIRE(rows > 0, "rmsnorm: rows=%d", rows);
    PE_REQUIRE(dim == 3584, "rmsnorm: dim=%d unsupported (text width 3584 only)", dim);
    hipLaunchKernelGGL((rmsnorm_kernel<7>), dim3((rows + 3) / 4), dim3(256), 0, stream, (const bf16*)x,
                       (const bf16*)w, (bf16*)out, rows, dim, eps);
    return check_launch("rmsnorm_kernel");
}

// ------------------------------------------------------------------------------------------------
// BlockWiseControlBlock input (models/qwen_image_controlnet.py:16-18): x_rms(x) + y_rms(y) over 3072, each RMSNorm rounded as
// models/utils.py:250-257 (x * rsqrt -> bf16, * weight -> bf16), then the bf16 add.  One wave per row.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) dual_rmsnorm_add_kernel(const bf16* __restrict__ x, const bf16* __restrict__ wx,
                                                               const bf16* __restrict__ y, const bf16* __restrict__ wy,
                                                               bf16* __restrict__ out, int rows, float eps) {
    constexpr int VPL = 6, DIM = 3072;
    const int lane = lane_id();
    const int row = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16* xr = x + (size_t)row * DIM;
    const bf16* yr = y + (size_t)row * DIM;
    float vx[VPL][8], vy[VPL][8];
    float sx = 0.f, sy = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const bf16x8 tx = *(const bf16x8*)(xr + (i * 64 + lane) * 8);
        const bf16x8 ty = *(const bf16x8*)(yr + (i * 64 + lane) * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            vx[i][j] = (float)tx[j];
            vy[i][j] = (float)ty[j];
            sx += vx[i][j] * vx[i][j];
            sy += vy[i][j] * vy[i][j];
        }
    }
    const float rx = rsqrtf(wave_sum(sx) / (float)DIM + eps);
    const float ry = rsqrtf(wave_sum(sy) / (float)DIM + eps);
    bf16* orow = out + (size_t)row * DIM;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 8;
        const bf16x8 wxv = *(const bf16x8*)(wx + c);
        const bf16x8 wyv = *(const bf16x8*)(wy + c);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            o[j] = (bf16)(bf16r(bf16r(vx[i][j] * rx) * (float)wxv[j]) + bf16r(bf16r(vy[i][j] * ry) * (float)wyv[j]));
        *(bf16x8*)(orow + c) = o;
    }
}

int launch_dual_rmsnorm_add(const void* x, const void* wx, const void* y, const void* wy, void* out, int rows, int dim, float eps,
                            hipStream_t stream) {
    PE_REQUIRE(x && wx && y && wy && out, "dual_rmsnorm_add: null pointer");
    PE_REQUIRE(rows > 0, "dual_rmsnorm_add: rows=%d", rows);
    PE_REQUIRE(dim == 3072, "dual_rmsnorm_add: dim=%d unsupported (DiT width 3072 only)", dim);
    hipLaunchKernelGGL(dual_rmsnorm_add_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, (const bf16*)x, (const bf16*)wx,
                       (const bf16*)y, (const bf16*)wy, (bf16*)out, rows, eps);
    return check_launch("dual_rmsnorm_add_kernel");
}

// x[i] = bf16(x[i] + sign * y[i]) over n elements (n % 8 == 0)
__global__ void __launch_bounds__(256) add_inplace_kernel(bf16* __restrict__ x, const bf16* __restrict__ y, size_t n8, float sign) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        bf16x8 a = *(const bf16x8*)(x + i * 8);
        const bf16x8 b = *(const bf16x8*)(y + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = (bf16)((float)a[j] + sign * (float)b[j]);
        *(bf16x8*)(x + i * 8) = a;
    }
}

int launch_add_inplace(void* x, const void* y, size_t n, hipStream_t stream, float sign) {
    PE_REQUIRE(x && y && n % 8 == 0, "add_inplace: null pointer or n %% 8 != 0");
    const size_t n8 = n / 8;
    const int blocks = (int)((n8 + 255) / 256 < 4096 ? (n8 + 255) / 256 : 4096);
    hipLaunchKernelGGL(add_inplace_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, stream, (bf16*)x, (const bf16*)y, n8, sign);
    return check_launch("add_inplace_kernel");
}

// ------------------------------------------------------------------------------------------------
// The two small kernels of the training-time prior's Perceiver resamplers (pipelines/helpers.py:8-109; row f2 of SURVEY.md section 8).
// nn.LayerNorm with affine parameters: fp32 statistics (two passes over the row), (x - mean) * rstd * w + b, ONE rounding.  One wave
// per row, any dim % 8 == 0.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) layernorm_affine_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                               const bf16* __restrict__ b, bf16* __restrict__ out, int rows, int dim,
                                                               float eps) {
    const int lane = lane_id();
    const int row = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16* xr = x + (size_t)row * dim;
    float s = 0.f;
    for (int c = lane * 8; c < dim; c += 512) {
        const bf16x8 t = *(const bf16x8*)(xr + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (float)t[j];
    }
    const float mean = wave_sum(s) / (float)dim;
    float v = 0.f;
    for (int c = lane * 8; c < dim; c += 512) {
        const bf16x8 t = *(const bf16x8*)(xr + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = (float)t[j] - mean;
            v += d * d;
        }
    }
    const float rstd = rsqrtf(wave_sum(v) / (float)dim + eps);
    for (int c = lane * 8; c < dim; c += 512) {
        const bf16x8 t = *(const bf16x8*)(xr + c);
        const bf16x8 wv = *(const bf16x8*)(w + c);
        const bf16x8 bv = *(const bf16x8*)(b + c);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)(((float)t[j] - mean) * rstd * (float)wv[j] + (float)bv[j]);
        *(bf16x8*)(out + (size_t)row * dim + c) = o;
    }
}

int launch_layernorm_affine(const void* x, const void* w, const void* b, void* out, int rows, int dim, float eps, hipStream_t stream) {
    PE_REQUIRE(x && w && b && out && rows > 0 && dim > 0 && dim % 8 == 0, "layernorm_affine: bad arguments (rows=%d dim=%d)", rows, dim);
    hipLaunchKernelGGL(layernorm_affine_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, (const bf16*)x, (const bf16*)w,
                       (const bf16*)b, (bf16*)out, rows, dim, eps);
    return check_launch("layernorm_affine_kernel");
}

// PerceiverAttention core (helpers.py:52-62) for heads of 64: q [nq, H*64], kv [nk, 2*H*64] (keys in the first H*64 columns, values
// in the last), out [nq, H*64].  The reference materialises every intermediate in bf16, and so does this: dots = bf16(q . k) (fp32
// accumulation), * scale -> bf16, - row max -> bf16, softmax in fp32 -> bf16, attn . v -> bf16.  One work-group per (head, query): the
// whole problem is 64 queries x <= 10 k keys.
// SDPA = true: the numerics of torch's scaled_dot_product_attention instead (DINOv2's self-attention, transformers
// Dinov2WithRegistersSelfAttention -> F.scaled_dot_product_attention): scores and softmax statistics in fp32, the un-normalised
// P = exp(s - max) rounded to bf16 for the second product, fp32 accumulation, ONE rounding of O / sum.
template <bool SDPA>
__global__ void __launch_bounds__(256) perceiver_attn_kernel(const bf16* __restrict__ q, const bf16* __restrict__ kv,
                                                             bf16* __restrict__ out, int nk, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) char pa_smem[];
    float* sc = (float*)pa_smem;                   // [nk]
    __shared__ float qs[64];
    __shared__ float red[8];
    __shared__ float part[4][64];
    const int h = (int)blockIdx.x, qi = (int)blockIdx.y, t = (int)threadIdx.x;
    const int ldq = H * 64, ldkv = 2 * H * 64;
    if (t < 64) qs[t] = (float)q[(size_t)qi * ldq + h * 64 + t];
    __syncthreads();
    float mx = -INFINITY;
    for (int j = t; j < nk; j += 256) {
        const bf16* kr = kv + (size_t)j * ldkv + h * 64;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const bf16x8 k8 = *(const bf16x8*)(kr + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = __builtin_fmaf(qs[c * 8 + e], (float)k8[e], acc);
        }
        const float s = SDPA ? acc * scale : bf16r(bf16r(acc) * scale);
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = t; j < nk; j += 256) {
        const float p = SDPA ? __expf(sc[j] - mx) : __expf(bf16r(sc[j] - mx));
        sc[j] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    if ((t & 63) == 0) red[4 + (t >> 6)] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    const int d = t & 63, grp = t >> 6;
    float o = 0.f;
    for (int j = grp; j < nk; j += 4)
        o = __builtin_fmaf(SDPA ? bf16r(sc[j]) : bf16r(sc[j] * inv), (float)kv[(size_t)j * ldkv + H * 64 + h * 64 + d], o);
    part[grp][d] = o;
    __syncthreads();
    if (t < 64) {
        const float tot = part[0][t] + part[1][t] + part[2][t] + part[3][t];
        out[(size_t)qi * ldq + h * 64 + t] = (bf16)(SDPA ? tot * inv : tot);
    }
}

int launch_perceiver_attn(const void* q, const void* kv, void* out, int nq, int nk, int heads, float scale, hipStream_t stream,
                          int sdpa) {
    PE_REQUIRE(q && kv && out && nq > 0 && nk > 0 && nk <= 15360 && heads > 0, "perceiver_attn: bad arguments (nq=%d nk=%d)", nq, nk);
    if (sdpa)
        hipLaunchKernelGGL((perceiver_attn_kernel<true>), dim3(heads, nq), dim3(256), (size_t)nk * 4, stream, (const bf16*)q,
                           (const bf16*)kv, (bf16*)out, nk, heads, scale);
    else
        hipLaunchKernelGGL((perceiver_attn_kernel<false>), dim3(heads, nq), dim3(256), (size_t)nk * 4, stream, (const bf16*)q,
                           (const bf16*)kv, (bf16*)out, nk, heads, scale);
    return check_launch("perceiver_attn_kernel");
}

// ------------------------------------------------------------------------------------------------
// GEMV: y[n] = bf16(sum_k W[n,k] x[k] + bias[n]) -- nn.Linear on ONE row (the autoregressive decode of the prompt prologue's text
// encoder: 15 GB of weights read once per generated token, i.e. HBM bound).  One wave per output row at a time, 2 rows in flight
// per wave, 16-B loads, x staged once per work-group in LDS, fp32 accumulation in k order per lane + a wave reduction.
// ------------------------------------------------------------------------------------------------
constexpr int GEMV_ROWS = 16;     // rows per work-group (4 waves x 2 rows x 2 rounds)

// x -> LDS for the single-row kernels of a 256-thread work-group.  With norm_w (K == 3584 only): the input is RMSNorm(x) * norm_w,
// every wave recomputes the row statistic with rmsnorm_kernel<7>'s lane -> element mapping and summation order, so the staged
// values are bit-identical to a separate pe_rmsnorm launch (which the captured decode step saves: 2 of 8 launches per layer).
PE_DEV void stage_row(bf16* __restrict__ xs, const bf16* __restrict__ x, int K, const bf16* __restrict__ norm_w, float eps) {
    if (!norm_w) {
        for (int i = threadIdx.x * 8; i < K; i += 256 * 8) *(bf16x8*)(xs + i) = *(const bf16x8*)(x + i);
    } else {
        const int lane = lane_id(), w = (int)(threadIdx.x >> 6);
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const bf16x8 t = *(const bf16x8*)(x + (i * 64 + lane) * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) sq += (float)t[j] * (float)t[j];
        }
        const float rs = rsqrtf(wave_sum(sq) / 3584.0f + eps);
        for (int i = w; i < 7; i += 4) {
            const int c = (i * 64 + lane) * 8;
            const bf16x8 t = *(const bf16x8*)(x + c);
            const bf16x8 wv = *(const bf16x8*)(norm_w + c);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)(bf16r((float)t[j] * rs) * (float)wv[j]);
            *(bf16x8*)(xs + c) = o;
        }
    }
    __syncthreads();
}

// Two rows of W against the staged x: per lane the 16-byte chunks k = lane * 8 + 512 i in ascending i, eight fmas per chunk in element
// order -- the summation order of the round-2 kernels, so tokens decoded with them are reproduced bit for bit -- but with the weight
// chunks of four steps requested before the first is used (8 x 16 B in flight per lane instead of 2) and with the non-temporal hint:
// a decode step reads every weight byte once (cdna guide: nt on weight streams that one CU reads once, -5 ... -10 % per layer).
PE_DEV void gemv_dot2(const bf16* __restrict__ xs, const bf16* __restrict__ wa, const bf16* __restrict__ wb, int K, int lane, float& sa,
                      float& sb) {
    sa = 0.f;
    sb = 0.f;
    int k = lane * 8;
    for (; k + 3 * 512 < K; k += 4 * 512) {
        bf16x8 va[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            va[u] = __builtin_nontemporal_load((const bf16x8*)(wa + k + u * 512));
            vb[u] = __builtin_nontemporal_load((const bf16x8*)(wb + k + u * 512));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bf16x8 xv = *(const bf16x8*)(xs + k + u * 512);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sa = __builtin_fmaf((float)va[u][j], (float)xv[j], sa);
                sb = __builtin_fmaf((float)vb[u][j], (float)xv[j], sb);
            }
        }
    }
    for (; k < K; k += 512) {
        const bf16x8 xv = *(const bf16x8*)(xs + k);
        const bf16x8 va = __builtin_nontemporal_load((const bf16x8*)(wa + k));
        const bf16x8 vb = __builtin_nontemporal_load((const bf16x8*)(wb + k));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sa = __builtin_fmaf((float)va[j], (float)xv[j], sa);
            sb = __builtin_fmaf((float)vb[j], (float)xv[j], sb);
        }
    }
}
// ROUNDS = 2: 16 rows per work-group (the lm_head: 9504 work-groups); ROUNDS = 1: 8 rows, for the launches of a few thousand rows
// (o_proj / down_proj, N = 3584: 224 work-groups of 16 rows left 32 CUs idle and four waves per CU to stream up to 600 KiB)
template <int ROUNDS>
__global__ void __launch_bounds__(256) gemv_bf16_kernel(const bf16* __restrict__ x, const bf16* __restrict__ W,
                                                        const bf16* __restrict__ bias, const bf16* __restrict__ res,
                                                        bf16* __restrict__ y, int N, int K, const bf16* __restrict__ norm_w, float eps) {
    extern __shared__ __attribute__((aligned(16))) char gemv_smem[];
    bf16* xs = (bf16*)gemv_smem;
    stage_row(xs, x, K, norm_w, eps);
    const int lane = lane_id();
    const int w = (int)(threadIdx.x >> 6);
    const int row0 = (int)blockIdx.x * (8 * ROUNDS) + w * 2;
#pragma unroll
    for (int rnd = 0; rnd < ROUNDS; ++rnd) {
        const int ra = row0 + rnd * 8, rb = ra + 1;
        if (ra >= N) break;
        const bf16* wa = W + (size_t)ra * K;
        const bf16* wb = W + (size_t)min(rb, N - 1) * K;
        float sa, sb;
        gemv_dot2(xs, wa, wb, K, lane, sa, sb);
        sa = wave_sum(sa);
        sb = wave_sum(sb);
        if (lane == 0) {                           // res: the decoder layer's `residual + linear(x)`, the sum rounded separately
            const float ya = bf16r(sa + (bias ? (float)bias[ra] : 0.f));
            y[ra] = res ? (bf16)((float)res[ra] + ya) : (bf16)ya;
            if (rb < N) {
                const float yb = bf16r(sb + (bias ? (float)bias[rb] : 0.f));
                y[rb] = res ? (bf16)((float)res[rb] + yb) : (bf16)yb;
            }
        }
    }
}

// y[n] = bf16(silu(g) * u) with g = bf16(Wg[n,:] . x), u = bf16(Wu[n,:] . x): the gated MLP's first half on one row
// (Qwen2MLP.forward: act_fn(gate_proj(x)) * up_proj(x); SiLU in fp32 with one rounding, then the bf16 product)
__global__ void __launch_bounds__(256) gemv_swiglu_kernel(const bf16* __restrict__ x, const bf16* __restrict__ Wg,
                                                          const bf16* __restrict__ Wu, bf16* __restrict__ y, int N, int K,
                                                          const bf16* __restrict__ norm_w, float eps) {
    extern __shared__ __attribute__((aligned(16))) char gemv_smem[];
    bf16* xs = (bf16*)gemv_smem;
    stage_row(xs, x, K, norm_w, eps);
    const int lane = lane_id();
    const int w = (int)(threadIdx.x >> 6);
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
        const int n = (int)blockIdx.x * 8 + rnd * 4 + w;
        if (n >= N) break;
        const bf16* wa = Wg + (size_t)n * K;
        const bf16* wb = Wu + (size_t)n * K;
        float sa, sb;
        gemv_dot2(xs, wa, wb, K, lane, sa, sb);
        sa = wave_sum(sa);
        sb = wave_sum(sb);
        if (lane == 0) {
            const float g = bf16r(sa), u = bf16r(sb);
            y[n] = (bf16)(bf16r(g / (1.0f + __expf(-g))) * u);
        }
    }
}

int launch_gemv_swiglu(const void* x, const void* Wg, const void* Wu, void* y, int N, int K, hipStream_t stream, const void* norm_w,
                       float eps) {
    PE_REQUIRE(x && Wg && Wu && y, "gemv_swiglu: null pointer");
    PE_REQUIRE(!norm_w || K == 3584, "gemv_swiglu: the fused RMSNorm is for the text width 3584 (K=%d)", K);
    PE_REQUIRE(N > 0 && K > 0 && K % 8 == 0 && K <= 32768, "gemv_swiglu: N=%d K=%d (K must be a multiple of 8, at most 32768)", N, K);
    hipLaunchKernelGGL(gemv_swiglu_kernel, dim3((N + 7) / 8), dim3(256), (size_t)K * 2, stream, (const bf16*)x, (const bf16*)Wg,
                       (const bf16*)Wu, (bf16*)y, N, K, (const bf16*)norm_w, eps);
    return check_launch("gemv_swiglu_kernel");
}

// Decode-step attention input of a GQA transformer layer (transformers Qwen2_5_VLAttention.forward, q_len = 1): q / k / v
// projections of ONE row in one launch over the concatenated row space (three weight / bias sets), then rotary embedding of the
// q and k heads:  y = bf16(bf16(x * cos) + bf16(rotate_half(x) * sin))  with rotate_half(x) = cat(-x[64:], x[:64]) per 128-wide
// head (apply_multimodal_rotary_pos_emb after the mrope section selection, which the caller does once per step).
__global__ void __launch_bounds__(256) gemv3_kernel(const bf16* __restrict__ x, const bf16* __restrict__ W0, const bf16* __restrict__ b0,
                                                    int N0, const bf16* __restrict__ W1, const bf16* __restrict__ b1, int N1,
                                                    const bf16* __restrict__ W2, const bf16* __restrict__ b2, int N2,
                                                    bf16* __restrict__ y0, bf16* __restrict__ y1, bf16* __restrict__ y2, int K,
                                                    const bf16* __restrict__ cs, const bf16* __restrict__ sn,
                                                    const int* __restrict__ step, int base, int ld,
                                                    const bf16* __restrict__ norm_w, float eps) {
    extern __shared__ __attribute__((aligned(16))) char gemv_smem[];
    bf16* xs = (bf16*)gemv_smem;
    // captured-graph form (step != nullptr): the rotary tables are [n_steps][128] and indexed by the device-side step counter, and
    // the k / v rows go straight into the caches [n_kv][ld][128] at row base + *step
    int pos = -1;
    if (step) {
        const int st = *step;
        cs += (size_t)st * 128;
        sn += (size_t)st * 128;
        pos = base + st;
        if (pos >= ld) return;                     // cache full: the host never replays that far; never write past it
    }
    stage_row(xs, x, K, norm_w, eps);
    const int lane = lane_id();
    const int w = (int)(threadIdx.x >> 6);
    // work item p = a PAIR of rows: for q / k the rows (i, i + 64) of one 128-wide head, which the rotary embedding mixes; for v two
    // neighbouring rows.  4 pairs per work-group (576 work-groups for the 7B text model's q / k / v; 8 pairs left 288).
    const int npairs = (N0 + N1 + N2) / 2;
    {
        const int p = (int)blockIdx.x * 4 + w;
        if (p >= npairs) return;
        const bf16* wr;
        const bf16* br;
        bf16* yr;
        int ra, rb;
        bool rope = true;
        int q2 = p;
        bool cached = false;
        if (q2 < N0 / 2) { wr = W0; br = b0; yr = y0; }
        else if ((q2 -= N0 / 2) < N1 / 2) { wr = W1; br = b1; yr = y1; cached = pos >= 0; }
        else { q2 -= N1 / 2; wr = W2; br = b2; yr = y2; rope = false; cached = pos >= 0; }
        if (rope) { ra = (q2 >> 6) * 128 + (q2 & 63); rb = ra + 64; }
        else { ra = q2 * 2; rb = ra + 1; }
        const bf16* rowa = wr + (size_t)ra * K;
        const bf16* rowb = wr + (size_t)rb * K;
        float sa, sb;
        gemv_dot2(xs, rowa, rowb, K, lane, sa, sb);
        sa = wave_sum(sa);
        sb = wave_sum(sb);
        if (lane == 0) {
            const float a = bf16r(sa + (br ? (float)br[ra] : 0.f));
            const float b = bf16r(sb + (br ? (float)br[rb] : 0.f));
            // output row r of head r >> 7: plain vector, or row `pos` of that head's cache plane
            const size_t oa = cached ? ((size_t)(ra >> 7) * ld + pos) * 128 + (ra & 127) : (size_t)ra;
            const size_t ob = cached ? ((size_t)(rb >> 7) * ld + pos) * 128 + (rb & 127) : (size_t)rb;
            if (rope) {     // q * cos + rotate_half(q) * sin, rotate_half(q) = cat(-q[64:], q[:64]); every product and the sum rounded
                const int i = ra & 127;
                yr[oa] = (bf16)(bf16r(a * (float)cs[i]) + bf16r(-b * (float)sn[i]));
                yr[ob] = (bf16)(bf16r(b * (float)cs[i + 64]) + bf16r(a * (float)sn[i + 64]));
            } else {
                yr[oa] = (bf16)a;
                yr[ob] = (bf16)b;
            }
        }
    }
}

int launch_decode_qkv(const void* x, const void* Wq, const void* bq, const void* Wk, const void* bk, const void* Wv,
                      const void* bv, const void* cos_sel, const void* sin_sel, void* q, void* k, void* v, int n_q_heads,
                      int n_kv_heads, int K, hipStream_t stream, const int* step, int base, int ld, const void* norm_w, float eps) {
    PE_REQUIRE(x && Wq && Wk && Wv && cos_sel && sin_sel && q && k && v, "decode_qkv: null pointer");
    PE_REQUIRE(!norm_w || K == 3584, "decode_qkv: the fused RMSNorm is for the text width 3584 (K=%d)", K);
    PE_REQUIRE(n_q_heads > 0 && n_kv_heads > 0 && K > 0 && K % 8 == 0 && K <= 32768, "decode_qkv: bad shape");
    PE_REQUIRE(!step || (base >= 0 && ld > base), "decode_qkv: cache of %d rows cannot take row %d", ld, base);
    const int N0 = n_q_heads * 128, N1 = n_kv_heads * 128;
    hipLaunchKernelGGL(gemv3_kernel, dim3(((N0 + 2 * N1) / 2 + 3) / 4), dim3(256), (size_t)K * 2, stream, (const bf16*)x,
                       (const bf16*)Wq, (const bf16*)bq, N0, (const bf16*)Wk, (const bf16*)bk, N1, (const bf16*)Wv, (const bf16*)bv, N1,
                       (bf16*)q, (bf16*)k, (bf16*)v, K, (const bf16*)cos_sel, (const bf16*)sin_sel, step, base, ld,
                       (const bf16*)norm_w, eps);
    return check_launch("gemv3_kernel");
}

// One query token against a KV cache [n_kv][ld][128] (GQA: query head h reads kv head h / (n_q / n_kv)), no mask:
// softmax(q K^T * scale) V with fp32 scores and sums, P rounded to bf16 before P.V (as the fused SDPA kernels do).
// One work-group of 16 waves per query head: the launch is latency bound (28 heads, ~0.7 MB of cache each), so what counts is the
// number of loads in flight per CU.
constexpr int DEC_NT = 1024;
__global__ void __launch_bounds__(DEC_NT) attn_decode_kernel(const bf16* __restrict__ q, const bf16* __restrict__ Kc,
                                                             const bf16* __restrict__ Vc, bf16* __restrict__ out, int n_q, int n_kv,
                                                             int L, float scale, const int* __restrict__ step, int base, int ld) {
    extern __shared__ __attribute__((aligned(16))) char dec_smem[];
    float* sc = (float*)dec_smem;                  // [L] scores, then probabilities
    constexpr int NW = DEC_NT / 64;
    __shared__ float red[2 * NW];
    __shared__ float part[NW][128];
    const int h = (int)blockIdx.x, t = (int)threadIdx.x;
    const int kvh = h / (n_q / n_kv);
    if (step) L = min(base + *step + 1, ld);       // captured-graph form: the cache planes hold ld rows, base + *step + 1 are valid
    const bf16* kb = Kc + (size_t)kvh * ld * 128;
    const bf16* vb = Vc + (size_t)kvh * ld * 128;
    // scores: 4 lanes share one key row (64 contiguous bytes each), DEC_NT / 4 keys per work-group iteration, two shuffles per key
    float mx = -INFINITY;
    {
        const int s4 = t & 3, g4 = t >> 2;
        float qf[32];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bf16x8 t8 = *(const bf16x8*)(q + (size_t)h * 128 + s4 * 32 + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[c * 8 + j] = (float)t8[j];
        }
#pragma unroll 4
        for (int j = g4; j < L; j += DEC_NT / 4) {
            const bf16* kr = kb + (size_t)j * 128 + s4 * 32;
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bf16x8 k8 = *(const bf16x8*)(kr + c * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc = __builtin_fmaf(qf[c * 8 + e], (float)k8[e], acc);
            }
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            acc *= scale;
            if (s4 == 0) sc[j] = acc;
            mx = fmaxf(mx, acc);
        }
    }
    const int sub = t & 15, grp = t >> 4;          // P.V below: DEC_NT / 16 groups, 4 per wave
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[w]);
    float sum = 0.f;
    for (int j = t; j < L; j += DEC_NT) {
        const float p = __expf(sc[j] - mx);
        sum += p;
        sc[j] = bf16r(p);
    }
    sum = wave_sum(sum);
    if ((t & 63) == 0) red[NW + (t >> 6)] = sum;
    __syncthreads();
    sum = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) sum += red[NW + w];
    // P.V: lane `sub` owns output channels 8 sub .. 8 sub + 7, group `grp` every (DEC_NT / 16)-th key; the four groups of a wave meet
    // by shuffles, the waves in LDS
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int j = grp; j < L; j += DEC_NT / 16) {
        const bf16x8 v8 = *(const bf16x8*)(vb + (size_t)j * 128 + sub * 8);
        const float p = sc[j];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = __builtin_fmaf(p, (float)v8[e], acc[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        acc[e] += __shfl_xor(acc[e], 16, 64);
        acc[e] += __shfl_xor(acc[e], 32, 64);
    }
    if ((t & 63) < 16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) part[t >> 6][sub * 8 + e] = acc[e];
    }
    __syncthreads();
    if (t < 128) {
        float o = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) o += part[w][t];
        out[(size_t)h * 128 + t] = (bf16)(o / sum);
    }
}

int launch_attn_decode(const void* q, const void* Kc, const void* Vc, void* out, int n_q_heads, int n_kv_heads, int L, float scale,
                       hipStream_t stream, const int* step, int base) {
    PE_REQUIRE(q && Kc && Vc && out, "attn_decode: null pointer");
    PE_REQUIRE(n_q_heads > 0 && n_kv_heads > 0 && n_q_heads % n_kv_heads == 0 && L > 0 && L <= 15360, "attn_decode: bad shape (L=%d)", L);
    PE_REQUIRE(!step || (base >= 0 && base < L), "attn_decode: base=%d outside the cache of %d rows", base, L);
    static std::atomic<bool> configured{false};    // the score buffer of a 15360-row cache + the static partials exceed the 64 KiB default
    if (!configured.load(std::memory_order_acquire)) {
        const hipError_t e = hipFuncSetAttribute((const void*)attn_decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 15360 * 4);
        if (e != hipSuccess) return set_error(PE_ERR_HIP, "attn_decode: hipFuncSetAttribute: %s", hipGetErrorString(e));
        configured.store(true, std::memory_order_release);
    }
    // with a step counter, L is the CAPACITY of the cache planes (and of the score buffer in LDS); the kernel reads the valid length
    hipLaunchKernelGGL(attn_decode_kernel, dim3(n_q_heads), dim3(DEC_NT), (size_t)L * 4, stream, (const bf16*)q, (const bf16*)Kc,
                       (const bf16*)Vc, (bf16*)out, n_q_heads, n_kv_heads, L, scale, step, base, L);
    return check_launch("attn_decode_kernel");
}

// ---- the same operator as THREE small launches over 448 + 448 + 28 work-groups instead of one over 28 (round 5; captured decode step).
// The one-launch form is latency bound: 28 work-groups on 256 CUs, 17 us of a 110 us layer.  The split keeps every sum's grouping --
// the output is bit-identical, so the greedy tokens are too:
//   scores   (head, block b of 16): keys j = 64 b + g .. step 1024, the same 4-lane dot product per key; scores and the block's maximum
//            go to the workspace (a maximum does not depend on its grouping)
//   softmax + P.V (head, w of 16): ONE wave = wave w of the 16-wave work-group: its 64 lanes sum exp(s - m) over keys lane + 64 w + 1024 i
//            exactly as slots t = 64 w + lane did, then the same butterfly; its four 16-lane groups accumulate P.V over keys
//            4 w + (lane >> 4) + 64 i in the same order (P = bf16(exp(s - m)) recomputed per lane: same bits), the same two shuffles
//   combine  (head): the 16 wave sums and the 16 x 128 partial outputs added in wave order, o / sum
constexpr int DEC_SB = 16;      // score blocks per head = waves of the one-launch form
__global__ void __launch_bounds__(256) attn_decode_scores_kernel(const bf16* __restrict__ q, const bf16* __restrict__ Kc, float* __restrict__ sc_g,
                                                                 float* __restrict__ mx_g, int n_q, int n_kv, float scale,
                                                                 const int* __restrict__ step, int base, int ld) {
    __shared__ float red[4];
    const int h = (int)blockIdx.x, b = (int)blockIdx.y, t = (int)threadIdx.x;
    const int kvh = h / (n_q / n_kv);
    const int L = min(base + *step + 1, ld);
    const bf16* kb = Kc + (size_t)kvh * ld * 128;
    float* sc = sc_g + (size_t)h * ld;
    const int s4 = t & 3, g4 = t >> 2;
    float qf[32];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const bf16x8 t8 = *(const bf16x8*)(q + (size_t)h * 128 + s4 * 32 + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[c * 8 + j] = (float)t8[j];
    }
    float mx = -INFINITY;
#pragma unroll 2
    for (int j = b * 64 + g4; j < L; j += 64 * DEC_SB) {
        const bf16* kr = kb + (size_t)j * 128 + s4 * 32;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bf16x8 k8 = *(const bf16x8*)(kr + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = __builtin_fmaf(qf[c * 8 + e], (float)k8[e], acc);
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc *= scale;
        if (s4 == 0) sc[j] = acc;
        mx = fmaxf(mx, acc);
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    if (t == 0) mx_g[h * DEC_SB + b] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ void __launch_bounds__(64) attn_decode_pv_kernel(const float* __restrict__ sc_g, const float* __restrict__ mx_g,
                                                            const bf16* __restrict__ Vc, float* __restrict__ red_g, float* __restrict__ part_g,
                                                            int n_q, int n_kv, const int* __restrict__ step, int base, int ld) {
    const int h = (int)blockIdx.x, w = (int)blockIdx.y, lane = (int)threadIdx.x;
    const int kvh = h / (n_q / n_kv);
    const int L = min(base + *step + 1, ld);
    const bf16* vb = Vc + (size_t)kvh * ld * 128;
    const float* sc = sc_g + (size_t)h * ld;
    float mx = mx_g[h * DEC_SB];
#pragma unroll
    for (int b = 1; b < DEC_SB; ++b) mx = fmaxf(mx, mx_g[h * DEC_SB + b]);
    float sum = 0.f;
    for (int j = w * 64 + lane; j < L; j += 64 * DEC_SB) sum += __expf(sc[j] - mx);
    sum = wave_sum(sum);
    if (lane == 0) red_g[h * DEC_SB + w] = sum;
    const int sub = lane & 15, grp = 4 * w + (lane >> 4);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int j = grp; j < L; j += 64) {
        const bf16x8 v8 = *(const bf16x8*)(vb + (size_t)j * 128 + sub * 8);
        const float p = bf16r(__expf(sc[j] - mx));
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = __builtin_fmaf(p, (float)v8[e], acc[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        acc[e] += __shfl_xor(acc[e], 16, 64);
        acc[e] += __shfl_xor(acc[e], 32, 64);
    }
    if (lane < 16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) part_g[((size_t)h * DEC_SB + w) * 128 + sub * 8 + e] = acc[e];
    }
}

__global__ void __launch_bounds__(128) attn_decode_combine_kernel(const float* __restrict__ red_g, const float* __restrict__ part_g,
                                                                 bf16* __restrict__ out) {
    const int h = (int)blockIdx.x, t = (int)threadIdx.x;
    float sum = 0.f, o = 0.f;
#pragma unroll
    for (int w = 0; w < DEC_SB; ++w) sum += red_g[h * DEC_SB + w];
#pragma unroll
    for (int w = 0; w < DEC_SB; ++w) o += part_g[((size_t)h * DEC_SB + w) * 128 + t];
    out[(size_t)h * 128 + t] = (bf16)(o / sum);
}

size_t attn_decode_workspace_bytes(int n_q_heads, int cache_len) {
    return ((size_t)n_q_heads * cache_len + (size_t)n_q_heads * DEC_SB * (2 + 128)) * sizeof(float);
}

int launch_attn_decode_split(const void* q, const void* Kc, const void* Vc, void* out, int n_q_heads, int n_kv_heads, int cache_len,
                             float scale, hipStream_t stream, const int* step, int base, void* workspace, size_t workspace_bytes) {
    PE_REQUIRE(q && Kc && Vc && out && step && workspace, "attn_decode_split: null pointer");
    PE_REQUIRE(n_q_heads > 0 && n_kv_heads > 0 && n_q_heads % n_kv_heads == 0 && cache_len > 0 && cache_len <= 15360,
               "attn_decode_split: bad shape (cache_len=%d)", cache_len);
    PE_REQUIRE(base >= 0 && base < cache_len, "attn_decode_split: base=%d outside the cache of %d rows", base, cache_len);
    PE_REQUIRE(workspace_bytes >= attn_decode_workspace_bytes(n_q_heads, cache_len) && ((uintptr_t)workspace & 15) == 0,
               "attn_decode_split: workspace of %zu bytes, 16-byte aligned, needed", attn_decode_workspace_bytes(n_q_heads, cache_len));
    float* sc_g = (float*)workspace;
    float* mx_g = sc_g + (size_t)n_q_heads * cache_len;
    float* red_g = mx_g + (size_t)n_q_heads * DEC_SB;
    float* part_g = red_g + (size_t)n_q_heads * DEC_SB;
    hipLaunchKernelGGL(attn_decode_scores_kernel, dim3(n_q_heads, DEC_SB), dim3(256), 0, stream, (const bf16*)q, (const bf16*)Kc, sc_g, mx_g,
                       n_q_heads, n_kv_heads, scale, step, base, cache_len);
    hipLaunchKernelGGL(attn_decode_pv_kernel, dim3(n_q_heads, DEC_SB), dim3(64), 0, stream, (const float*)sc_g, (const float*)mx_g,
                       (const bf16*)Vc, red_g, part_g, n_q_heads, n_kv_heads, step, base, cache_len);
    hipLaunchKernelGGL(attn_decode_combine_kernel, dim3(n_q_heads), dim3(128), 0, stream, (const float*)red_g, (const float*)part_g, (bf16*)out);
    return check_launch("attn_decode_split");
}

// ------------------------------------------------------------------------------------------------
// Round 6: ONE launch per decoder layer of the captured decode step (instead of eight: q/k/v, three attention launches, o_proj, gate/up,
// down_proj).  A persistent grid (two work-groups per CU, all resident) walks the SAME work items the separate kernels' blocks are --
// every output element is computed by the same instructions in the same order (gemv_dot2 / wave_sum / the attention split's groupings), so
// the greedy tokens are identical -- with a grid-wide barrier between the phases.  What it saves: per launch the ramp (first loads with an
// empty memory pipeline), the tail and the launch gap, ~4 us each for kernels that stream 3 - 270 MB (profiles/r04_prologue.md: 466 MB per
// layer in 105 us against 74 us at the achievable HBM rate).  A work-group stages its input row ONCE per phase, not once per 8 rows.
//   barrier j: arrive on counter bar[j] (release: every wave has drained its own stores), wait until all G have arrived, one lane acquires
//   (invalidates this CU's L1).  The last arriver of barrier j zeroes the counter of the barrier BEFORE it (everybody has left that one), the
//   last arriver of barrier 0 the previous launch's last counter (stream order: that launch is over): no host reset, hipGraph-safe.
//   Placement-independent (cdna guide): nothing assumes which work-group runs where; it assumes that all G are RESIDENT, which the launcher
//   guarantees by grid size (2 per CU at 256 threads, 38 KiB LDS, <= 128 registers).  A wait that exceeds ~1 s gives up and raises
//   *err (the host checks it): a wrong token beats a hung GPU.
// ------------------------------------------------------------------------------------------------
constexpr int DL_BARRIERS = 6;

// Two-level arrival: work-group b adds to the counter of group b & 7 (its own 128-byte line; with the default dispatch order those are the
// work-groups of one XCD, but nothing depends on that), the last arriver of a group adds to the barrier's global counter, everybody polls
// that one word.  (One flat counter for 512 work-groups measured ~60 us per barrier: 512 agent-scope atomics on one line, serialised at the
// memory side, under 512 pollers.)  Layout: barrier j owns 9 lines of 32 words at bar + j * 288: [0] global, [32 (1 + g)] group g.
constexpr int DL_BAR_WORDS = 9 * 32;
int g_decode_layer_wgs_per_cu = 2;      // knob "decode_layer_wgs_per_cu" (1 .. 8; capped by what is resident; 2 measured best: a barrier costs its fan-in)
int g_decode_layer_no_barrier = 0;      // TIMING ONLY (knob "decode_layer_no_barrier"): the phases run without grid barriers -- wrong results
PE_DEV void dl_barrier(unsigned* bar, int j, int G, unsigned* err) {
    if (bar == nullptr) { __syncthreads(); return; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // every wave: its own stores have left for the L2
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* mine = bar + j * DL_BAR_WORDS;
        unsigned* prev = bar + (j == 0 ? DL_BARRIERS - 1 : j - 1) * DL_BAR_WORDS;
        const int g = (int)blockIdx.x & 7;
        const unsigned in_group = (unsigned)((G - g + 7) >> 3);      // work-groups b < G with b & 7 == g
        const unsigned old = __hip_atomic_fetch_add(mine + 32 * (1 + g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == in_group - 1) {
            __hip_atomic_store(prev + 32 * (1 + g), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // everybody has left the barrier before
            const unsigned og = __hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (og == 7u) __hip_atomic_store(prev, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        int spins = 0;
        while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 8u) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 24)) { *err = 1u + (unsigned)j; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // this CU's L1 (one lane + barrier)
    }
    __syncthreads();
}

// one wave's pair of rows of a Linear on the staged row: gemv_bf16_kernel's body
PE_DEV void dl_gemv_rows(const bf16* xs, const bf16* W, const bf16* res, bf16* y, int N, int K, int ra, int lane) {
    if (ra >= N) return;
    const int rb = ra + 1;
    float sa, sb;
    gemv_dot2(xs, W + (size_t)ra * K, W + (size_t)min(rb, N - 1) * K, K, lane, sa, sb);
    sa = wave_sum(sa);
    sb = wave_sum(sb);
    if (lane == 0) {
        const float ya = bf16r(sa);
        y[ra] = res ? (bf16)((float)res[ra] + ya) : (bf16)ya;
        if (rb < N) {
            const float yb = bf16r(sb);
            y[rb] = res ? (bf16)((float)res[rb] + yb) : (bf16)yb;
        }
    }
}

__global__ void __launch_bounds__(256, 4) decode_layer_kernel(const DecodeLayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) char gemv_smem[];
    __shared__ float red[4];
    bf16* xs = (bf16*)gemv_smem;
    const int G = (int)gridDim.x, b0 = (int)blockIdx.x;
    const int t = (int)threadIdx.x, lane = lane_id(), w = t >> 6;
    const int K = a.K, FF = a.FF;
    const int st = *a.step;
    const int pos = a.base + st;
    if (pos >= a.ld) return;                       // cache full (every work-group alike): the host never replays that far
    const int L = min(pos + 1, a.ld);
    const int N0 = a.n_q * 128, N1 = a.n_kv * 128;

    // ---- phase 1: q / k / v + rotary + cache append (gemv3_kernel: work item p = a pair of rows, one per wave)
    {
        const bf16* cs = a.cs + (size_t)st * 128;
        const bf16* sn = a.sn + (size_t)st * 128;
        const int npairs = (N0 + 2 * N1) / 2;
        if (b0 * 4 < npairs) stage_row(xs, a.x, K, a.ln1, a.eps1);
        for (int vb = b0; vb * 4 < npairs; vb += G) {
            const int p = vb * 4 + w;
            if (p >= npairs) continue;
            const bf16* wr;
            const bf16* br;
            bf16* yr;
            int ra, rb, q2 = p;
            bool rope = true, cached = false;
            if (q2 < N0 / 2) { wr = a.Wq; br = a.bq; yr = a.q; }
            else if ((q2 -= N0 / 2) < N1 / 2) { wr = a.Wk; br = a.bk; yr = a.Kc; cached = true; }
            else { q2 -= N1 / 2; wr = a.Wv; br = a.bv; yr = a.Vc; rope = false; cached = true; }
            if (rope) { ra = (q2 >> 6) * 128 + (q2 & 63); rb = ra + 64; }
            else { ra = q2 * 2; rb = ra + 1; }
            float sa, sb;
            gemv_dot2(xs, wr + (size_t)ra * K, wr + (size_t)rb * K, K, lane, sa, sb);
            sa = wave_sum(sa);
            sb = wave_sum(sb);
            if (lane == 0) {
                const float va = bf16r(sa + (br ? (float)br[ra] : 0.f));
                const float vb2 = bf16r(sb + (br ? (float)br[rb] : 0.f));
                const size_t oa = cached ? ((size_t)(ra >> 7) * a.ld + pos) * 128 + (ra & 127) : (size_t)ra;
                const size_t ob = cached ? ((size_t)(rb >> 7) * a.ld + pos) * 128 + (rb & 127) : (size_t)rb;
                if (rope) {
                    const int i = ra & 127;
                    yr[oa] = (bf16)(bf16r(va * (float)cs[i]) + bf16r(-vb2 * (float)sn[i]));
                    yr[ob] = (bf16)(bf16r(vb2 * (float)cs[i + 64]) + bf16r(va * (float)sn[i + 64]));
                } else {
                    yr[oa] = (bf16)va;
                    yr[ob] = (bf16)vb2;
                }
            }
        }
    }
    dl_barrier(a.bar, 0, G, a.err);

    // ---- phase 2: scores (attn_decode_scores_kernel: item = (head, block of 16))
    for (int vb = b0; vb < a.n_q * DEC_SB; vb += G) {
        const int h = vb / DEC_SB, b = vb - h * DEC_SB;
        const int kvh = h / (a.n_q / a.n_kv);
        const bf16* kb = a.Kc + (size_t)kvh * a.ld * 128;
        float* sc = a.sc_g + (size_t)h * a.ld;
        const int s4 = t & 3, g4 = t >> 2;
        float qf[32];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bf16x8 t8 = *(const bf16x8*)(a.q + (size_t)h * 128 + s4 * 32 + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[c * 8 + j] = (float)t8[j];
        }
        float mx = -INFINITY;
#pragma unroll 2
        for (int j = b * 64 + g4; j < L; j += 64 * DEC_SB) {
            const bf16* kr = kb + (size_t)j * 128 + s4 * 32;
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bf16x8 k8 = *(const bf16x8*)(kr + c * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc = __builtin_fmaf(qf[c * 8 + e], (float)k8[e], acc);
            }
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            acc *= a.scale;
            if (s4 == 0) sc[j] = acc;
            mx = fmaxf(mx, acc);
        }
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        __syncthreads();                               // (red is free again: the previous item's reader is done)
        if ((t & 63) == 0) red[t >> 6] = mx;
        __syncthreads();
        if (t == 0) a.mx_g[h * DEC_SB + b] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    }
    dl_barrier(a.bar, 1, G, a.err);

    // ---- phase 3: softmax sums + P.V (attn_decode_pv_kernel: item = (head, wave of 16) = ONE wave)
    for (int it = b0 * 4 + w; it < a.n_q * DEC_SB; it += G * 4) {
        const int h = it / DEC_SB, ww = it - h * DEC_SB;
        const int kvh = h / (a.n_q / a.n_kv);
        const bf16* vbp = a.Vc + (size_t)kvh * a.ld * 128;
        const float* sc = a.sc_g + (size_t)h * a.ld;
        float mx = a.mx_g[h * DEC_SB];
#pragma unroll
        for (int b = 1; b < DEC_SB; ++b) mx = fmaxf(mx, a.mx_g[h * DEC_SB + b]);
        float sum = 0.f;
        for (int j = ww * 64 + lane; j < L; j += 64 * DEC_SB) sum += __expf(sc[j] - mx);
        sum = wave_sum(sum);
        if (lane == 0) a.red_g[h * DEC_SB + ww] = sum;
        const int sub = lane & 15, grp = 4 * ww + (lane >> 4);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int j = grp; j < L; j += 64) {
            const bf16x8 v8 = *(const bf16x8*)(vbp + (size_t)j * 128 + sub * 8);
            const float p = bf16r(__expf(sc[j] - mx));
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = __builtin_fmaf(p, (float)v8[e], acc[e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            acc[e] += __shfl_xor(acc[e], 16, 64);
            acc[e] += __shfl_xor(acc[e], 32, 64);
        }
        if (lane < 16) {
#pragma unroll
            for (int e = 0; e < 8; ++e) a.part_g[((size_t)h * DEC_SB + ww) * 128 + sub * 8 + e] = acc[e];
        }
    }
    dl_barrier(a.bar, 2, G, a.err);

    // ---- phase 4: combine (attn_decode_combine_kernel: 128 threads per head, two heads per work-group)
    for (int h = b0 * 2 + (t >> 7); h < a.n_q; h += G * 2) {
        const int c = t & 127;
        float sum = 0.f, o = 0.f;
#pragma unroll
        for (int ww = 0; ww < DEC_SB; ++ww) sum += a.red_g[h * DEC_SB + ww];
#pragma unroll
        for (int ww = 0; ww < DEC_SB; ++ww) o += a.part_g[((size_t)h * DEC_SB + ww) * 128 + c];
        a.a[(size_t)h * 128 + c] = (bf16)(o / sum);
    }
    dl_barrier(a.bar, 3, G, a.err);

    // ---- phase 5: h1 = x + o_proj(a)   (gemv_bf16_kernel<1>: 8 rows per item, 2 per wave)
    {
        if (b0 * 8 < K) stage_row(xs, a.a, N0, nullptr, 0.f);
        for (int vb = b0; vb * 8 < K; vb += G) dl_gemv_rows(xs, a.Wo, a.x, a.h1, K, N0, vb * 8 + w * 2, lane);
    }
    dl_barrier(a.bar, 4, G, a.err);

    // ---- phase 6: hid = silu(gate(n)) * up(n), n = RMSNorm(h1) * ln2   (gemv_swiglu_kernel: 8 rows per item, two rounds of one per wave)
    {
        if (b0 * 8 < FF) stage_row(xs, a.h1, K, a.ln2, a.eps2);
        for (int vb = b0; vb * 8 < FF; vb += G) {
#pragma unroll
            for (int rnd = 0; rnd < 2; ++rnd) {
                const int n = vb * 8 + rnd * 4 + w;
                if (n >= FF) break;
                float sa, sb;
                gemv_dot2(xs, a.Wg + (size_t)n * K, a.Wu + (size_t)n * K, K, lane, sa, sb);
                sa = wave_sum(sa);
                sb = wave_sum(sb);
                if (lane == 0) {
                    const float g = bf16r(sa), u = bf16r(sb);
                    a.hid[n] = (bf16)(bf16r(g / (1.0f + __expf(-g))) * u);
                }
            }
        }
    }
    dl_barrier(a.bar, 5, G, a.err);

    // ---- phase 7: x_out = h1 + down_proj(hid)
    {
        if (b0 * 8 < K) stage_row(xs, a.hid, FF, nullptr, 0.f);
        for (int vb = b0; vb * 8 < K; vb += G) dl_gemv_rows(xs, a.Wd, a.h1, a.x_out, K, FF, vb * 8 + w * 2, lane);
    }
}

int launch_decode_layer(const DecodeLayerArgs& a, hipStream_t stream) {
    PE_REQUIRE(a.x && a.x_out && a.Wq && a.Wk && a.Wv && a.Wo && a.Wg && a.Wu && a.Wd && a.ln1 && a.ln2 && a.cs && a.sn && a.Kc && a.Vc && a.step,
               "decode_layer: null pointer");
    PE_REQUIRE(a.q && a.a && a.h1 && a.hid && a.sc_g && a.bar && a.err, "decode_layer: null scratch");
    PE_REQUIRE(a.FF <= 32768 && a.K <= 32768, "decode_layer: rows of at most 32768 elements");
    PE_REQUIRE(a.K == 3584 && a.n_q * 128 == a.K && a.n_q % a.n_kv == 0 && a.FF > 0 && a.FF % 8 == 0 && a.FF <= 32768,
               "decode_layer: shape (K=%d n_q=%d n_kv=%d FF=%d): the fused norms are for the text width 3584", a.K, a.n_q, a.n_kv, a.FF);
    PE_REQUIRE(a.base >= 0 && a.ld > a.base && a.ld <= 15360, "decode_layer: cache of %d rows, base %d", a.ld, a.base);
    static std::atomic<int> grid{0};
    static std::atomic<int> grid_for{0};
    int G = grid.load(std::memory_order_acquire);
    const size_t lds = (size_t)(a.FF > a.K ? a.FF : a.K) * 2;      // the staged row of the widest phase
    if (G == 0 || grid_for.load(std::memory_order_acquire) != g_decode_layer_wgs_per_cu) {
        int dev = 0, per_cu = 0;
        hipDeviceProp_t prop;
        hipError_t e = hipFuncSetAttribute((const void*)decode_layer_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        if (e == hipSuccess) e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
        if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_layer_kernel, 256, lds);
        if (e != hipSuccess) return set_error(PE_ERR_HIP, "decode_layer: %s", hipGetErrorString(e));
        PE_REQUIRE(per_cu >= 1, "decode_layer: the kernel does not fit a CU");
        const int want = g_decode_layer_wgs_per_cu < 1 ? 1 : g_decode_layer_wgs_per_cu;
        G = prop.multiProcessorCount * (per_cu < want ? per_cu : want);      // every work-group resident: the grid barrier's one assumption
        grid.store(G, std::memory_order_release);
        grid_for.store(g_decode_layer_wgs_per_cu, std::memory_order_release);
    }
    DecodeLayerArgs b = a;
    if (g_decode_layer_no_barrier) b.bar = nullptr;
    hipLaunchKernelGGL(decode_layer_kernel, dim3(G), dim3(256), lds, stream, b);
    return check_launch("decode_layer_kernel");
}

// Training-loss head of the visual-thinking adapter (VisualThinkingDualAdapter.get_loss, pipelines/helpers.py:166-183), the part that
// touches tensors: F.mse_loss(pred, gt, reduction='none').mean(dim=[1, 2]) for the two heads -- (pred - gt) rounded to bf16, its
// square rounded to bf16, fp32 mean.  One work-group per head; out[head] = fp32 mean (the caller rounds it to bf16 as .mean() does).
__global__ void __launch_bounds__(1024) adapter_mse_kernel(const bf16* __restrict__ pd, const bf16* __restrict__ gd,
                                                           const bf16* __restrict__ pv, const bf16* __restrict__ gv,
                                                           size_t n, float* __restrict__ out) {
    __shared__ float red[16];
    const bf16* p = blockIdx.x == 0 ? pd : pv;
    const bf16* g = blockIdx.x == 0 ? gd : gv;
    float acc = 0.f;
    for (size_t i = (size_t)threadIdx.x * 8; i < n; i += 1024 * 8) {
        const bf16x8 a = *(const bf16x8*)(p + i);
        const bf16x8 b = *(const bf16x8*)(g + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = bf16r((float)a[j] - (float)b[j]);
            acc += bf16r(d * d);
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s2 = 0.f;
        for (int w = 0; w < 16; ++w) s2 += red[w];
        out[blockIdx.x] = s2 / (float)n;
    }
}

int launch_adapter_mse(const void* pred_dino, const void* gt_dino, const void* pred_vae, const void* gt_vae, size_t n, float* out,
                       hipStream_t stream) {
    PE_REQUIRE(pred_dino && gt_dino && pred_vae && gt_vae && out && n > 0 && n % 8 == 0, "adapter_mse: bad arguments");
    hipLaunchKernelGGL(adapter_mse_kernel, dim3(2), dim3(1024), 0, stream, (const bf16*)pred_dino, (const bf16*)gt_dino,
                       (const bf16*)pred_vae, (const bf16*)gt_vae, n, out);
    return check_launch("adapter_mse_kernel");
}

// Greedy sampling glue of a captured decode step: token embedding lookup by a device-side token id, and arg-max of the logits
// (first index among equal maxima, like torch.argmax) that also appends the token to the output list and advances the step counter.
__global__ void __launch_bounds__(256) embed_row_kernel(const bf16* __restrict__ table, const int* __restrict__ token, bf16* __restrict__ x,
                                                        int dim, int vocab) {
    const int tk = min(max(*token, 0), vocab - 1);
    for (int i = (int)(blockIdx.x * 256 + threadIdx.x) * 8; i < dim; i += (int)gridDim.x * 256 * 8)
        *(bf16x8*)(x + i) = *(const bf16x8*)(table + (size_t)tk * dim + i);
}

__global__ void __launch_bounds__(1024) argmax_step_kernel(const bf16* __restrict__ logits, int V, int* __restrict__ token,
                                                           int* __restrict__ out_ids, int* __restrict__ step, int max_steps) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    float best = -INFINITY;
    int idx = 0x7fffffff;
    // 16-byte loads: thread t scans elements 8 t .. 8 t + 7 of every 8192-element stride in ascending order (the first maximum stays),
    // then the tail; ties across threads go to the lower index below, so the result is the first maximum whatever the partition
    const int V8 = V & ~7;
    for (int i = (int)threadIdx.x * 8; i < V8; i += 1024 * 8) {
        const bf16x8 t8 = *(const bf16x8*)(logits + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = (float)t8[j];
            if (v > best) { best = v; idx = i + j; }
        }
    }
    for (int i = V8 + (int)threadIdx.x; i < V; i += 1024) {
        const float v = (float)logits[i];
        if (v > best || (v == best && i < idx)) { best = v; idx = i; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        if (idx == 0x7fffffff) idx = 0;            // all NaN / -inf
        const int st = *step;
        *token = idx;
        if (st < max_steps) out_ids[st] = idx;
        *step = st + 1;
    }
}

int launch_embed_row(const void* table, const int* token, void* x, int dim, int vocab, hipStream_t stream) {
    PE_REQUIRE(table && token && x && dim > 0 && dim % 8 == 0 && vocab > 0, "embed_row: bad arguments");
    hipLaunchKernelGGL(embed_row_kernel, dim3((dim / 8 + 255) / 256), dim3(256), 0, stream, (const bf16*)table, token, (bf16*)x, dim, vocab);
    return check_launch("embed_row_kernel");
}

int launch_argmax_step(const void* logits, int V, int* token, int* out_ids, int* step, int max_steps, hipStream_t stream) {
    PE_REQUIRE(logits && token && out_ids && step && V > 0 && max_steps > 0, "argmax_step: bad arguments");
    hipLaunchKernelGGL(argmax_step_kernel, dim3(1), dim3(1024), 0, stream, (const bf16*)logits, V, token, out_ids, step, max_steps);
    return check_launch("argmax_step_kernel");
}

int launch_gemv(const void* x, const void* W, const void* bias, void* y, int N, int K, hipStream_t stream, const void* res,
                const void* norm_w, float eps) {
    PE_REQUIRE(x && W && y, "gemv: null pointer");
    PE_REQUIRE(!norm_w || K == 3584, "gemv: the fused RMSNorm is for the text width 3584 (K=%d)", K);
    PE_REQUIRE(N > 0 && K > 0 && K % 8 == 0 && K <= 32768, "gemv: N=%d K=%d (K must be a multiple of 8, at most 32768)", N, K);
    if (N <= 8192)
        hipLaunchKernelGGL(gemv_bf16_kernel<1>, dim3((N + 7) / 8), dim3(256), (size_t)K * 2, stream, (const bf16*)x,
                           (const bf16*)W, (const bf16*)bias, (const bf16*)res, (bf16*)y, N, K, (const bf16*)norm_w, eps);
    else
        hipLaunchKernelGGL(gemv_bf16_kernel<2>, dim3((N + GEMV_ROWS - 1) / GEMV_ROWS), dim3(256), (size_t)K * 2, stream, (const bf16*)x,
                           (const bf16*)W, (const bf16*)bias, (const bf16*)res, (bf16*)y, N, K, (const bf16*)norm_w, eps);
    return check_launch("gemv_bf16_kernel");
}

// ------------------------------------------------------------------------------------------------
// SiLU (torch.nn.SiLU on bf16: fp32 inside, one rounding)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) silu_kernel(const bf16* __restrict__ x, bf16* __restrict__ out, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const bf16x8 t = *(const bf16x8*)(x + i * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float y = (float)t[j];
            o[j] = (bf16)(y / (1.0f + __expf(-y)));
        }
        *(bf16x8*)(out + i * 8) = o;
    }
}

int launch_silu(const void* x, void* out, size_t n, hipStream_t stream) {
    PE_REQUIRE(x && out, "silu: null pointer");
    PE_REQUIRE(n % 8 == 0 && n > 0, "silu: n=%zu must be a positive multiple of 8", n);
    const size_t n8 = n / 8;
    const int grid = (int)((n8 + 255) / 256 < 2048 ? (n8 + 255) / 256 : 2048);
    hipLaunchKernelGGL(silu_kernel, dim3(grid), dim3(256), 0, stream, (const bf16*)x, (bf16*)out, n8);
    return check_launch("silu_kernel");
}

// ------------------------------------------------------------------------------------------------
// Row quantiser of AutoWrappedLinear.fp8_linear (vram_management/layers.py:126-137), GPU semantics of the reference:
//   x_max   = max|x[m,:]|                     (bf16, exact)
//   scale_a = clamp(x_max / 448, min=1)       bf16 tensor / python scalar on the GPU = x * (1/448) in fp32, rounded to bf16
//   xq      = e4m3fn( float(x) / (float(scale_a) + 1e-8) )     fp32 IEEE division, RNE conversion
// One 256-thread block per row; a thread keeps up to MAXC 8-element chunks in registers (K <= 2048*MAXC).
// Columns [K, Kp) of the output are zero-filled (K tile of the e4m3 GEMM is 128).
// ------------------------------------------------------------------------------------------------
template <int MAXC>
__global__ void __launch_bounds__(256) quantize_rows_e4m3_kernel(const bf16* __restrict__ x, int ldx, int K,
                                                                 uint8_t* __restrict__ out, int Kp,
                                                                 float* __restrict__ scale) {
    __shared__ float red[4];
    const int row = (int)blockIdx.x;
    const int t = (int)threadIdx.x;
    const bf16* xr = x + (size_t)row * ldx;
    const int nch = K >> 3;
    bf16x8 v[MAXC];
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = i * 256 + t;
        if (c < nch) {
            v[i] = *(const bf16x8*)(xr + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf((float)v[i][j]));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float sc = fmaxf(bf16r(mx * (1.0f / 448.0f)), 1.0f);
    const float dv = sc + 1e-8f;
    if (t == 0) scale[row] = sc;
    uint8_t* orow = out + (size_t)row * Kp;
    if (dv == 1.0f) {          // scale 1: x / (1 + 1e-8f) is x / 1.0f is x (see ln_modulate_kernel): no divisions
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = i * 256 + t;
            if (c < nch) {
                u32x2 o;
                o[0] = pack4_e4m3((float)v[i][0], (float)v[i][1], (float)v[i][2], (float)v[i][3]);
                o[1] = pack4_e4m3((float)v[i][4], (float)v[i][5], (float)v[i][6], (float)v[i][7]);
                *(u32x2*)(orow + c * 8) = o;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = i * 256 + t;
            if (c < nch) {
                u32x2 o;
                o[0] = pack4_e4m3((float)v[i][0] / dv, (float)v[i][1] / dv, (float)v[i][2] / dv, (float)v[i][3] / dv);
                o[1] = pack4_e4m3((float)v[i][4] / dv, (float)v[i][5] / dv, (float)v[i][6] / dv, (float)v[i][7] / dv);
                *(u32x2*)(orow + c * 8) = o;
            }
        }
    }
    for (int c = nch + t; c < (Kp >> 3); c += 256) *(u32x2*)(orow + c * 8) = u32x2{0u, 0u};
}

// Fix-up pass of the quantisation fused into the producer GEMM's epilogue (GemmProblem.q8_out): the epilogue wrote e4m3(y) for
// every element, which IS fp8_linear's quantisation whenever the row's scale is 1 (max|row| <= 447 => bf16(max / 448) <= 1 =>
// clamp(.., min=1) = 1 and x / (1 + 1e-8f) == x in fp32), and flagged the rows holding anything larger.  One block per row:
// clear flag -> write scale 1 and leave; raised flag -> the full row quantisation from the bf16 row (same arithmetic as
// quantize_rows_e4m3_kernel), and the flag is lowered for the next launch.
template <int MAXC>
__global__ void __launch_bounds__(256) requant_flagged_rows_kernel(const bf16* __restrict__ x, int ldx, int K,
                                                                   uint8_t* __restrict__ out, int Kp, float* __restrict__ scale,
                                                                   unsigned* __restrict__ flags) {
    __shared__ float red[4];
    const int row = (int)blockIdx.x;
    const int t = (int)threadIdx.x;
    if (flags[row] == 0u) {
        if (t == 0) scale[row] = 1.0f;
        return;
    }
    const bf16* xr = x + (size_t)row * ldx;
    const int nch = K >> 3;
    bf16x8 v[MAXC];
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = i * 256 + t;
        if (c < nch) {
            v[i] = *(const bf16x8*)(xr + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf((float)v[i][j]));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float sc = fmaxf(bf16r(mx * (1.0f / 448.0f)), 1.0f);
    const float dv = sc + 1e-8f;
    if (t == 0) {
        scale[row] = sc;
        flags[row] = 0u;
    }
    uint8_t* orow = out + (size_t)row * Kp;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = i * 256 + t;
        if (c < nch) {
            u32x2 o;
            o[0] = pack4_e4m3((float)v[i][0] / dv, (float)v[i][1] / dv, (float)v[i][2] / dv, (float)v[i][3] / dv);
            o[1] = pack4_e4m3((float)v[i][4] / dv, (float)v[i][5] / dv, (float)v[i][6] / dv, (float)v[i][7] / dv);
            *(u32x2*)(orow + c * 8) = o;
        }
    }
}

int launch_requant_flagged_rows(const void* x, int ldx, int M, int K, void* out, int Kp, float* scale, unsigned* flags,
                                hipStream_t stream) {
    PE_REQUIRE(x && out && scale && flags, "requant_flagged_rows: null pointer");
    PE_REQUIRE(M > 0 && K > 0 && K % 8 == 0 && K <= 12288 && Kp == K && K % 128 == 0,
               "requant_flagged_rows: M=%d K=%d Kp=%d (K = Kp: multiple of 128, <= 12288)", M, K, Kp);
    PE_REQUIRE(ldx >= K && ldx % 8 == 0, "requant_flagged_rows: ldx=%d", ldx);
    const int slot = prof_begin(PROF_ROW, 8.0 * (double)M, stream);   // bytes: flag read + scale write per row
    hipLaunchKernelGGL((requant_flagged_rows_kernel<6>), dim3(M), dim3(256), 0, stream, (const bf16*)x, ldx, K, (uint8_t*)out, Kp, scale,
                       flags);
    prof_end(slot, stream);
    return check_launch("requant_flagged_rows_kernel");
}

int launch_quantize_rows_e4m3(const void* x, int ldx, int M, int K, void* out, int Kp, float* scale, hipStream_t stream) {
    PE_REQUIRE(x && out && scale, "quantize_rows_e4m3: null pointer");
    PE_REQUIRE(M > 0 && K > 0 && K % 8 == 0 && K <= 12288, "quantize_rows_e4m3: M=%d K=%d (K: multiple of 8, <= 12288)", M, K);
    PE_REQUIRE(ldx >= K && ldx % 8 == 0, "quantize_rows_e4m3: ldx=%d", ldx);
    PE_REQUIRE(Kp >= K && Kp % 128 == 0, "quantize_rows_e4m3: Kp=%d must be >= K and a multiple of 128", Kp);
    const int slot = prof_begin(PROF_ROW, 3.0 * (double)M * K, stream);   // bytes: read bf16 + write e4m3
    if (K <= 4096)
        hipLaunchKernelGGL((quantize_rows_e4m3_kernel<2>), dim3(M), dim3(256), 0, stream, (const bf16*)x, ldx, K,
                           (uint8_t*)out, Kp, scale);
    else
        hipLaunchKernelGGL((quantize_rows_e4m3_kernel<6>), dim3(M), dim3(256), 0, stream, (const bf16*)x, ldx, K,
                           (uint8_t*)out, Kp, scale);
    prof_end(slot, stream);
    return check_launch("quantize_rows_e4m3_kernel");
}

// ------------------------------------------------------------------------------------------------
// patchify "C (H P) (W Q) -> (H W) (C P Q)", P=Q=2  (qwen_image_physical.py:1344) and its inverse (:1402)
// latents [C, H2, W2]; tokens [(H2/2)*(W2/2), C*4].  One thread per (token, channel): 4 elements.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) patchify_kernel(const bf16* __restrict__ lat, bf16* __restrict__ tok,
                                                       int C, int H2, int W2, int inverse) {
    const int Wt = W2 / 2;
    const size_t total = (size_t)(H2 / 2) * Wt * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const size_t t = i / C;
        const int wi = (int)(t % Wt), hi = (int)(t / Wt);
        bf16* tp = tok + t * (size_t)(C * 4) + c * 4;
        bf16* lp = const_cast<bf16*>(lat) + ((size_t)c * H2 + 2 * hi) * W2 + 2 * wi;
        if (!inverse) {
            const bf16x2 r0 = *(const bf16x2*)lp;
            const bf16x2 r1 = *(const bf16x2*)(lp + W2);
            bf16x4 o;
            o[0] = r0[0]; o[1] = r0[1]; o[2] = r1[0]; o[3] = r1[1];
            *(bf16x4*)tp = o;
        } else {
            const bf16x4 v = *(const bf16x4*)tp;
            bf16x2 r0, r1;
            r0[0] = v[0]; r0[1] = v[1]; r1[0] = v[2]; r1[1] = v[3];
            *(bf16x2*)lp = r0;
            *(bf16x2*)(lp + W2) = r1;
        }
    }
}

static int launch_patch(const void* lat, void* tok, int C, int H2, int W2, int inverse, hipStream_t stream) {
    PE_REQUIRE(lat && tok, "patchify: null pointer");
    PE_REQUIRE(C > 0 && H2 > 0 && W2 > 0 && H2 % 2 == 0 && W2 % 2 == 0, "patchify: bad shape C=%d H=%d W=%d", C, H2, W2);
    const size_t total = (size_t)(H2 / 2) * (W2 / 2) * C;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(patchify_kernel, dim3(grid), dim3(256), 0, stream, (const bf16*)lat, (bf16*)tok, C, H2, W2, inverse);
    return check_launch("patchify_kernel");
}
int launch_patchify(const void* latents, void* tokens, int C, int H2, int W2, hipStream_t stream) {
    return launch_patch(latents, tokens, C, H2, W2, 0, stream);
}
int launch_unpatchify(const void* tokens, void* latents, int C, int H2, int W2, hipStream_t stream) {
    return launch_patch(latents, const_cast<void*>(tokens), C, H2, W2, 1, stream);
}

// ------------------------------------------------------------------------------------------------
// special-token gather / adapter mix + in-place scatter
//   prompt_emb[special_token_mask] -> adapter -> prompt_emb[special_token_mask] = ...
//   (qwen_image_physical.py:1333-1336; mix: helpers.py:160-162)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gather_rows_kernel(const bf16* __restrict__ src, const int* __restrict__ idx,
                                                          bf16* __restrict__ dst, int dim8) {
    const int r = (int)blockIdx.x;
    const bf16* s = src + (size_t)idx[r] * dim8 * 8;
    bf16* d = dst + (size_t)r * dim8 * 8;
    for (int i = (int)threadIdx.x; i < dim8; i += 256) *(bf16x8*)(d + i * 8) = *(const bf16x8*)(s + i * 8);
}

int launch_gather_rows(const void* src, const int* idx, void* dst, int nrows, int dim, hipStream_t stream) {
    PE_REQUIRE(src && idx && dst, "gather_rows: null pointer");
    PE_REQUIRE(nrows > 0 && dim % 8 == 0, "gather_rows: bad shape %d x %d", nrows, dim);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(nrows), dim3(256), 0, stream, (const bf16*)src, idx, (bf16*)dst, dim / 8);
    return check_launch("gather_rows_kernel");
}

__global__ void __launch_bounds__(256) adapter_mix_scatter_kernel(const bf16* __restrict__ dino,
                                                                  const bf16* __restrict__ vae, float alpha,
                                                                  float oma, const int* __restrict__ idx,
                                                                  bf16* __restrict__ pe, int dim8) {
    const int r = (int)blockIdx.x;
    const bf16* a = dino + (size_t)r * dim8 * 8;
    const bf16* b = vae + (size_t)r * dim8 * 8;
    bf16* d = pe + (size_t)(idx ? idx[r] : r) * dim8 * 8;     // idx == null: identity row map (pe_adapter_forward)
    for (int i = (int)threadIdx.x; i < dim8; i += 256) {
        const bf16x8 av = *(const bf16x8*)(a + i * 8);
        const bf16x8 bv = *(const bf16x8*)(b + i * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)(bf16r(alpha * (float)av[j]) + bf16r(oma * (float)bv[j]));
        *(bf16x8*)(d + i * 8) = o;
    }
}

int launch_adapter_mix_scatter(const void* dino, const void* vae, float alpha, float one_minus_alpha,
                               const int* idx, void* prompt_emb, int nrows, int dim, hipStream_t stream) {
    PE_REQUIRE(dino && vae && prompt_emb, "adapter_mix: null pointer");
    PE_REQUIRE(nrows > 0 && dim % 8 == 0, "adapter_mix: bad shape %d x %d", nrows, dim);
    hipLaunchKernelGGL(adapter_mix_scatter_kernel, dim3(nrows), dim3(256), 0, stream, (const bf16*)dino,
                       (const bf16*)vae, alpha, one_minus_alpha, idx, (bf16*)prompt_emb, dim / 8);
    return check_launch("adapter_mix_scatter_kernel");
}

// ------------------------------------------------------------------------------------------------
// CFG combine + Euler step in one pass over the latent
//   noise_pred = nega + cfg*(posi - nega)                  qwen_image_physical.py:656
//   latents    = latents + noise_pred * (sigma' - sigma)   schedulers/flow_match.py:81
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) cfg_euler_kernel(const bf16* __restrict__ posi, const bf16* __restrict__ nega,
                                                        const bf16* __restrict__ lat, bf16* __restrict__ out,
                                                        size_t n, float cfg, int use_cfg, float dsigma,
                                                        const bf16* __restrict__ x0, const bf16* __restrict__ mask, size_t plane,
                                                        float sigma) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float np = (float)posi[i];
        const float l = (float)lat[i];
        if (use_cfg) {
            const float ng = (float)nega[i];
            const float d = bf16r(np - ng);
            const float e = bf16r(cfg * d);
            np = bf16r(ng + e);
        }
        if (mask) {
            // BasePipeline.step with an inpaint mask (utils/__init__.py:146-154): outside the mask the model's prediction is
            // replaced by the one that leads back to the input latents; every torch op rounds to bf16
            const float m = (float)mask[i % plane];
            const float expect = bf16r(bf16r(l - (float)x0[i]) / sigma);             // scheduler.return_to_timestep
            np = bf16r(bf16r(expect * bf16r(1.0f - m)) + bf16r(np * m));             // blend_with_mask
        }
        out[i] = (bf16)(l + bf16r(np * dsigma));
    }
}

int launch_cfg_euler(const void* posi, const void* nega, const void* latents, void* out, size_t n,
                     float cfg_scale, int use_cfg, float dsigma, hipStream_t stream,
                     const void* input_latents, const void* mask, size_t plane, float sigma) {
    PE_REQUIRE(posi && latents && out && (!use_cfg || nega), "cfg_euler: null pointer");
    PE_REQUIRE(n > 0, "cfg_euler: empty");
    PE_REQUIRE(!mask || (input_latents && plane > 0 && n % plane == 0 && sigma > 0.f),
               "cfg_euler: inpaint needs input latents, a mask plane that divides n (%zu / %zu) and sigma > 0", n, plane);
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(cfg_euler_kernel, dim3(grid), dim3(256), 0, stream, (const bf16*)posi, (const bf16*)nega,
                       (const bf16*)latents, (bf16*)out, n, cfg_scale, use_cfg, dsigma, (const bf16*)input_latents,
                       (const bf16*)mask, mask ? plane : (size_t)1, sigma);
    return check_launch("cfg_euler_kernel");
}

}  // namespace pe
