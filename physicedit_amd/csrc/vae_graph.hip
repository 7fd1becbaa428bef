// VAE and adapter composites of the C-ABI (include/physicedit_amd.h): QwenImageVAE.encode / .decode as fixed launch
// sequences over the kernels of vae.hip, and VisualThinkingDualAdapter.forward over the GEMM.  Host code only.
//   encode  DiffSynth-Studio/diffsynth/models/qwen_image_vae.py:706-717  (Encoder3d :344-448, ResidualBlock :81-152,
//           Resample :218-300, AttentionBlock :156-198, MidBlock :304-340)
//   decode  :719-729 (Decoder3d :522-636, UpBlock :452-518)
//   image io BasePipeline.preprocess_image / vae_output_to_image, pipelines/utils/__init__.py:60-66,76-83 -- fused into the
//           entry / exit layout kernels (uint8 HWC <-> bf16 NHWC), no stand-alone element-wise pass
//   adapter pipelines/helpers.py:123-164
// Activations live in a caller-provided workspace: four slots of H*W*96 bf16 (the largest activation of either
// direction) handed out by a tiny pool; a residual block needs at most four live at once.
#include <new>
#include <string.h>

#include "../../include/physicedit_amd.h"
#include "common.h"
#include "kernels.h"

using namespace pe;

namespace pe {

// uint8 HWC -> bf16 NHWC (3 channels padded to Cp with zeros): x = bf16(u8); x = bf16(x * (2/255)); x = bf16(x + (-1))
__global__ void __launch_bounds__(256) u8hwc_to_nhwc_kernel(const uint8_t* __restrict__ in, bf16* __restrict__ out, size_t npix,
                                                            int Cp) {
    const size_t total = npix * Cp;
    const float k = (float)((1.0 - (-1.0)) / 255.0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % Cp);
        const size_t p = i / Cp;
        float v = 0.f;
        if (c < 3) v = bf16r(bf16r((float)in[p * 3 + c] * k) + (-1.0f));
        out[i] = (bf16)v;
    }
}

// bf16 NHWC (Cp channels, 3 valid) -> uint8 HWC: ((x - (-1)) * (255/2)).clip(0, 255) in bf16, truncated to uint8
__global__ void __launch_bounds__(256) nhwc_to_u8hwc_kernel(const bf16* __restrict__ in, uint8_t* __restrict__ out, size_t npix,
                                                            int Cp) {
    const size_t total = npix * 3;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t p = i / 3;
        const int c = (int)(i - p * 3);
        float v = bf16r((float)in[p * Cp + c] - (-1.0f));
        v = bf16r(v * 127.5f);
        v = fminf(fmaxf(v, 0.f), 255.f);
        out[i] = (uint8_t)v;                    // truncation, as torch's .to(uint8)
    }
}

static int launch_u8_to_nhwc(const void* in, void* out, size_t npix, int Cp, hipStream_t stream) {
    const size_t total = npix * Cp;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(u8hwc_to_nhwc_kernel, dim3(grid), dim3(256), 0, stream, (const uint8_t*)in, (bf16*)out, npix, Cp);
    return check_launch("u8hwc_to_nhwc_kernel");
}

static int launch_nhwc_to_u8(const void* in, void* out, size_t npix, int Cp, hipStream_t stream) {
    const size_t total = npix * 3;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(nhwc_to_u8hwc_kernel, dim3(grid), dim3(256), 0, stream, (const bf16*)in, (uint8_t*)out, npix, Cp);
    return check_launch("nhwc_to_u8hwc_kernel");
}

}  // namespace pe

struct pe_vae {
    pe_vae_weights w;
};

namespace {

constexpr int MAX_CH = 96;      // channels of the full-resolution activations (dim = 96)
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Pool {                   // four activation slots + the attention scratch behind them
    char* base = nullptr;
    size_t slot = 0;
    bool used[4] = {false, false, false, false};
    char* get() {
        for (int i = 0; i < 4; ++i)
            if (!used[i]) { used[i] = true; return base + i * slot; }
        return nullptr;
    }
    void put(const char* p) {
        if (!p) return;
        const size_t i = (size_t)(p - base) / slot;
        if (i < 4) used[i] = false;
    }
    char* extra() { return base + 4 * slot; }
};

struct Ctx {
    const pe_vae_weights* w;
    Pool pool;
    hipStream_t stream;
    int rc = PE_OK;
};

#define VG_TRY(expr)                      \
    do {                                  \
        if (c.rc == PE_OK) c.rc = (expr); \
    } while (0)

// out = conv(x) (+ res); returns the output slot (x is NOT released here)
char* conv(Ctx& c, const pe_vae_conv& cv, const char* x, int H, int W, const char* res, int stride, int upsample) {
    char* out = c.pool.get();
    if (!out) { if (c.rc == PE_OK) c.rc = set_error(PE_ERR_INVALID_ARG, "pe_vae: activation pool exhausted"); return nullptr; }
    VG_TRY(launch_conv_nhwc(x, cv.w, cv.b, res, out, c.w->zero_page, H, W, cv.cin_p, cv.cout_p, cv.ksize, stride, upsample, c.stream));
    return out;
}

char* norm(Ctx& c, const void* gamma, const char* x, size_t npix, int C, int silu) {
    char* out = c.pool.get();
    if (!out) { if (c.rc == PE_OK) c.rc = set_error(PE_ERR_INVALID_ARG, "pe_vae: activation pool exhausted"); return nullptr; }
    VG_TRY(launch_vae_rmsnorm(x, gamma, out, (int)npix, C, C, silu, c.stream));
    return out;
}

// QwenImageResidualBlock.forward (:112-152); consumes (releases) x
char* res_block(Ctx& c, const pe_vae_res& r, char* x, int H, int W) {
    const size_t npix = (size_t)H * W;
    char* h = x;
    if (r.shortcut.w) h = conv(c, r.shortcut, x, H, W, nullptr, 1, 0);
    char* y = norm(c, r.norm1_g, x, npix, r.conv1.cin_p, 1);
    if (h != x) c.pool.put(x);
    char* y1 = conv(c, r.conv1, y, H, W, nullptr, 1, 0);
    c.pool.put(y);
    char* y2 = norm(c, r.norm2_g, y1, npix, r.conv1.cout_p, 1);
    c.pool.put(y1);
    char* out = conv(c, r.conv2, y2, H, W, h, 1, 0);      // x + h fused into the conv epilogue
    c.pool.put(y2);
    c.pool.put(h);
    return out;
}

// QwenImageAttentionBlock.forward (:173-198); consumes x
char* attn_block(Ctx& c, const pe_vae_attn& a, char* x, int H, int W) {
    const size_t N = (size_t)H * W;
    char* y = norm(c, a.norm_g, x, N, 384, 0);
    char* qkv = conv(c, a.to_qkv, y, H, W, nullptr, 1, 0);          // [N, 1152]
    c.pool.put(y);
    char* o = c.pool.get();
    if (!o) { if (c.rc == PE_OK) c.rc = set_error(PE_ERR_INVALID_ARG, "pe_vae: activation pool exhausted"); return nullptr; }
    VG_TRY(launch_vae_attention(qkv, c.pool.extra(), o, (int)N, c.stream));
    c.pool.put(qkv);
    char* out = conv(c, a.proj, o, H, W, x, 1, 0);                  // x + proj(o) (:198)
    c.pool.put(o);
    c.pool.put(x);
    return out;
}

char* mid_block(Ctx& c, const pe_vae_mid& m, char* x, int H, int W) {
    x = res_block(c, m.res0, x, H, W);
    x = attn_block(c, m.attn, x, H, W);
    return res_block(c, m.res1, x, H, W);
}

size_t slot_bytes(int H, int W) {
    // full-resolution slots hold [H*W][96]; the mid block (H/8 x W/8 pixels) needs [N][1152] for qkv: 1152/64 < 96 always
    return align_up((size_t)H * W * MAX_CH * 2, 256);
}

int setup(Ctx& c, pe_vae_handle h, void* ws, size_t ws_bytes, int H, int W, void* stream, const char* who) {
    PE_REQUIRE(h, "%s: null handle", who);
    PE_REQUIRE(H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0, "%s: image %dx%d must be a positive multiple of 8", who, H, W);
    const size_t need = pe_vae_workspace_bytes(H, W);
    PE_REQUIRE(ws && ws_bytes >= need && ((uintptr_t)ws & 255) == 0, "%s: workspace of %zu bytes (256-B aligned) needed, got %zu", who,
               need, ws_bytes);
    c.w = &h->w;
    c.pool.base = (char*)ws;
    c.pool.slot = slot_bytes(H, W);
    c.stream = (hipStream_t)stream;
    return PE_OK;
}

}  // namespace

extern "C" {

int pe_vae_create(const pe_vae_weights* w, pe_vae_handle* out) {
    PE_REQUIRE(w && out, "pe_vae_create: null argument");
    PE_REQUIRE(w->zero_page && w->mean && w->inv_std, "pe_vae_create: zero_page / mean / inv_std missing");
    pe_vae* h = new (std::nothrow) pe_vae();
    PE_REQUIRE(h, "pe_vae_create: out of host memory");
    h->w = *w;
    *out = h;
    return PE_OK;
}

void pe_vae_destroy(pe_vae_handle h) { delete h; }

size_t pe_vae_workspace_bytes(int H, int W) {
    if (H <= 0 || W <= 0) return 0;
    const size_t N = (size_t)(H / 8) * (W / 8);
    return 4 * slot_bytes(H, W) + align_up(vae_attention_scratch_bytes((int)N), 256);      // + the mid-block attention's scratch (Vt, key-split partials)
}

int pe_vae_encode(pe_vae_handle h, const void* image, int input_format, int H, int W, void* latents, void* ws, size_t ws_bytes,
                  void* stream) {
    Ctx c;
    int rc = setup(c, h, ws, ws_bytes, H, W, stream, "pe_vae_encode");
    if (rc) return rc;
    PE_REQUIRE(image && latents, "pe_vae_encode: null tensor");
    PE_REQUIRE(input_format == PE_IMAGE_BF16_NCHW || input_format == PE_IMAGE_U8_HWC, "pe_vae_encode: input_format %d", input_format);
    const pe_vae_weights& w = h->w;
    char* a = c.pool.get();
    if (input_format == PE_IMAGE_U8_HWC) VG_TRY(launch_u8_to_nhwc(image, a, (size_t)H * W, 32, c.stream));
    else VG_TRY(launch_nchw_to_nhwc(image, a, 3, H * W, 32, 0, nullptr, nullptr, c.stream));
    char* t = conv(c, w.enc_conv_in, a, H, W, nullptr, 1, 0);
    c.pool.put(a);
    a = t;
    int idx = 0;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 2; ++j) a = res_block(c, w.enc_res[idx++], a, H, W);
        if (i != 3) {
            t = conv(c, w.enc_down[i], a, H, W, nullptr, 2, 0);      // ZeroPad2d((0,1,0,1)) + stride-2 conv (:249)
            c.pool.put(a);
            a = t;
            H /= 2; W /= 2;
        }
    }
    a = mid_block(c, w.enc_mid, a, H, W);
    t = norm(c, w.enc_norm_out_g, a, (size_t)H * W, 384, 1);
    c.pool.put(a);
    a = conv(c, w.enc_conv_out, t, H, W, nullptr, 1, 0);
    c.pool.put(t);
    t = conv(c, w.quant_conv, a, H, W, nullptr, 1, 0);
    c.pool.put(a);
    VG_TRY(launch_nhwc_to_nchw(t, latents, 16, H * W, w.quant_conv.cout_p, 2, w.mean, w.inv_std, c.stream));   // (mu - mean) * 1/std (:713-714)
    return c.rc;
}

int pe_vae_decode(pe_vae_handle h, const void* latents, int H8, int W8, void* image, int output_format, void* ws, size_t ws_bytes,
                  void* stream) {
    Ctx c;
    int rc = setup(c, h, ws, ws_bytes, H8 * 8, W8 * 8, stream, "pe_vae_decode");
    if (rc) return rc;
    PE_REQUIRE(image && latents, "pe_vae_decode: null tensor");
    PE_REQUIRE(output_format == PE_IMAGE_BF16_NCHW || output_format == PE_IMAGE_U8_HWC, "pe_vae_decode: output_format %d", output_format);
    const pe_vae_weights& w = h->w;
    int H = H8, W = W8;
    char* a = c.pool.get();
    VG_TRY(launch_nchw_to_nhwc(latents, a, 16, H * W, 32, 1, w.mean, w.inv_std, c.stream));    // z / (1/std) + mean (:723-724)
    char* t = conv(c, w.post_quant_conv, a, H, W, nullptr, 1, 0);
    c.pool.put(a);
    a = conv(c, w.dec_conv_in, t, H, W, nullptr, 1, 0);
    c.pool.put(t);
    a = mid_block(c, w.dec_mid, a, H, W);
    int idx = 0;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 3; ++j) a = res_block(c, w.dec_res[idx++], a, H, W);
        if (i != 3) {
            t = conv(c, w.dec_up[i], a, H, W, nullptr, 1, 1);        // nearest-exact 2x fused into the conv's gather (:213-214,240)
            c.pool.put(a);
            a = t;
            H *= 2; W *= 2;
        }
    }
    t = norm(c, w.dec_norm_out_g, a, (size_t)H * W, 96, 1);
    c.pool.put(a);
    a = conv(c, w.dec_conv_out, t, H, W, nullptr, 1, 0);
    c.pool.put(t);
    if (output_format == PE_IMAGE_U8_HWC) VG_TRY(launch_nhwc_to_u8(a, image, (size_t)H * W, w.dec_conv_out.cout_p, c.stream));
    else VG_TRY(launch_nhwc_to_nchw(a, image, 3, H * W, w.dec_conv_out.cout_p, 0, nullptr, nullptr, c.stream));
    return c.rc;
}

size_t pe_adapter_workspace_bytes(int n) { return n <= 0 ? 0 : align_up((size_t)n * (10752 + 2 * 3584) * 2, 256); }

int pe_adapter_forward(const pe_adapter_weights* ad, const void* x, int n, float alpha, float one_minus_alpha, void* out,
                       void* ws, size_t ws_bytes, void* stream_) {
    PE_REQUIRE(ad && x && out && n > 0, "pe_adapter_forward: bad arguments");
    PE_REQUIRE(ws && ws_bytes >= pe_adapter_workspace_bytes(n), "pe_adapter_forward: workspace of %zu bytes needed, got %zu",
               pe_adapter_workspace_bytes(n), ws_bytes);
    hipStream_t stream = (hipStream_t)stream_;
    constexpr int TXT = 3584, HID = 10752;
    char* hid = (char*)ws;
    char* o_dino = hid + (size_t)n * HID * 2;
    char* o_vae = o_dino + (size_t)n * TXT * 2;
    int rc;
    for (int head = 0; head < 2; ++head) {
        GemmProblem p;
        memset(&p, 0, sizeof(p));
        p.A = x; p.lda = TXT; p.W = head == 0 ? ad->dino_w0 : ad->vae_w0; p.bias = head == 0 ? ad->dino_b0 : ad->vae_b0;
        p.out = hid; p.ldo = HID; p.M = n; p.N = HID; p.K = TXT;
        if ((rc = launch_gemm(EPI_GELU_ERF, &p, 1, stream))) return rc;          // Linear + nn.GELU() (helpers.py:127-131)
        memset(&p, 0, sizeof(p));
        p.A = hid; p.lda = HID; p.W = head == 0 ? ad->dino_w2 : ad->vae_w2; p.bias = head == 0 ? ad->dino_b2 : ad->vae_b2;
        p.out = head == 0 ? o_dino : o_vae; p.ldo = TXT; p.M = n; p.N = TXT; p.K = HID;
        if ((rc = launch_gemm(EPI_BIAS, &p, 1, stream))) return rc;
    }
    // alpha * dino + (1 - alpha) * vae, each op rounded (helpers.py:158-160); identity row map
    return launch_adapter_mix_scatter(o_dino, o_vae, alpha, one_minus_alpha, nullptr, out, n, TXT, stream);
}

}  // extern "C"
