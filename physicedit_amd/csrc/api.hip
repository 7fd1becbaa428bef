// C-ABI entry points for the granular operators + error plumbing (include/physicedit_amd.h).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/physicedit_amd.h"
#include "common.h"
#include "kernels.h"

namespace pe {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(PE_ERR_HIP, "%s: launch failed: %s", what, hipGetErrorString(e));
    return PE_OK;
}

}  // namespace pe

using namespace pe;

extern "C" {

const char* pe_last_error(void) { return g_err; }
int pe_abi_version(void) { return 8; }
#ifndef PE_SRC_HASH
#define PE_SRC_HASH "unknown"
#endif
const char* pe_build_id(void) { return PE_SRC_HASH; }

int pe_debug_set(const char* key, int value) {
    PE_REQUIRE(key != nullptr, "pe_debug_set: null key");
    if (!strcmp(key, "gemm_variant")) {
        PE_REQUIRE(value == 0 || value == 15 || value == 17 || value == 19 || value == 21 || value == 22,
                   "gemm_variant %d does not exist (0 = default, 15, 17, 19, 21, 22)", value);
        g_gemm_variant = value == 0 ? GEMM_DEFAULT_VARIANT : value;
        return PE_OK;
    }
    if (!strcmp(key, "dit_qkv_stats")) { g_dit_qkv_stats = value != 0; return PE_OK; }
    if (!strcmp(key, "dit_trim_last_block")) { g_dit_trim_last_block = value != 0; return PE_OK; }
    if (!strcmp(key, "attn_variant")) { g_attn_variant = value; return PE_OK; }
    if (!strcmp(key, "attn_fp8_variant")) { PE_REQUIRE(value >= 0 && value <= 4, "attn_fp8_variant: 0 ... 4"); g_attn_fp8_variant = value; return PE_OK; }
    if (!strcmp(key, "gemm_sk")) { g_gemm_sk = value; return PE_OK; }
    if (!strcmp(key, "gemm4_x")) { g_gemm4_x = value; return PE_OK; }
    if (!strcmp(key, "gemm_skip_ragged")) { g_gemm_skip_ragged = value; return PE_OK; }
    if (!strcmp(key, "decode_layer_wgs_per_cu")) { PE_REQUIRE(value >= 1 && value <= 8, "decode_layer_wgs_per_cu: 1 .. 8"); g_decode_layer_wgs_per_cu = value; return PE_OK; }
    if (!strcmp(key, "decode_layer_no_barrier")) { g_decode_layer_no_barrier = value != 0; return PE_OK; }
    if (!strcmp(key, "gemm_defer_epilogue")) { g_gemm_defer = value != 0; return PE_OK; }
    if (!strcmp(key, "gemm_continuous")) { g_gemm_cont = value != 0; return PE_OK; }
    if (!strcmp(key, "gemm_no_epilogue")) { g_gemm_no_epi = value != 0; return PE_OK; }
    if (!strcmp(key, "gemm_direct_epilogue")) { g_gemm_direct_epi = value; return PE_OK; }
    if (!strcmp(key, "gemm_mfma16")) { g_gemm_mfma16 = value; return PE_OK; }
    if (!strcmp(key, "gemm_persist_min_rounds")) { PE_REQUIRE(value >= 1 && value <= 64, "gemm_persist_min_rounds out of range"); g_gemm_persist_min_rounds = value; return PE_OK; }
    if (!strcmp(key, "gemm_band")) { PE_REQUIRE(value >= 1 && value <= 64, "gemm_band out of range"); g_gemm_band = value; return PE_OK; }
    if (!strcmp(key, "gemm_persist_wgs")) { PE_REQUIRE(value >= 0 && value <= 1024, "gemm_persist_wgs out of range"); g_gemm_persist_wgs = value; return PE_OK; }
    if (!strcmp(key, "attn_slots")) { PE_REQUIRE(value > 0 && value <= 256, "attn_slots out of range"); g_attn_slots = value; return PE_OK; }
    if (!strcmp(key, "attn_force_split")) { g_attn_force_split = value; return PE_OK; }
    return set_error(PE_ERR_INVALID_ARG, "pe_debug_set: unknown key %s", key);
}

int pe_debug_set_ptr(const char* key, void* p) {
    PE_REQUIRE(key != nullptr, "pe_debug_set_ptr: null key");
    if (!strcmp(key, "gemm_stamps")) { g_gemm_dbg = (long long*)p; return PE_OK; }
    if (!strcmp(key, "attn_stamps")) { g_attn_dbg = (long long*)p; return PE_OK; }
    // tests: a zeroed device buffer of pe_gemm_workspace_bytes() bytes (256-byte aligned) for the granular pe_gemm_* calls, or null
    if (!strcmp(key, "gemm_stash")) { g_gemm_ws.stash = p; g_gemm_ws.stash_bytes = p ? gemm_stash_bytes() : 0; return PE_OK; }
    if (!strcmp(key, "gemm_workspace")) { g_gemm_ws.sync = p; g_gemm_ws.bytes = p ? gemm_workspace_bytes() : 0; return PE_OK; }
    return set_error(PE_ERR_INVALID_ARG, "pe_debug_set_ptr: unknown key %s", key);
}

size_t pe_gemm_workspace_bytes(void) { return gemm_workspace_bytes(); }
size_t pe_gemm_stash_bytes(void) { return gemm_stash_bytes(); }

int pe_gemm_bf16(int epilogue, const void* A, int lda, const void* W, const void* bias, void* out, int ldo,
                 int M, int N, int K, const void* gate, const void* res, int ldr, void* stream) {
    PE_REQUIRE(epilogue != EPI_QKV, "pe_gemm_bf16: use pe_qkv_rmsnorm_rope for the QKV epilogue");
    GemmProblem p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.W = W; p.bias = bias; p.out = out;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldo = ldo;
    p.gate = gate; p.res = res; p.ldr = ldr;
    return launch_gemm(epilogue, &p, 1, (hipStream_t)stream);
}

int pe_gemm_bf16_pre(int epilogue, const void* A, int lda, const void* W, const void* bias, const void* pre, int ldp,
                     void* out, int ldo, int M, int N, int K, const void* gate, const void* res, int ldr, void* stream) {
    PE_REQUIRE(epilogue != EPI_QKV, "pe_gemm_bf16_pre: use pe_qkv_rmsnorm_rope for the QKV epilogue");
    GemmProblem p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.W = W; p.bias = bias; p.out = out; p.pre = pre; p.ldp = ldp;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldo = ldo;
    p.gate = gate; p.res = res; p.ldr = ldr;
    return launch_gemm(epilogue, &p, 1, (hipStream_t)stream);
}

int pe_lora_merge(void* W, int N, int K, const void* up, const void* down_t, int r, float alpha, void* stream) {
    PE_REQUIRE(W && up && down_t, "pe_lora_merge: null pointer");
    PE_REQUIRE(r > 0 && r % 64 == 0, "pe_lora_merge: rank %d must be padded to a multiple of 64 (zero columns)", r);
    GemmProblem p;
    memset(&p, 0, sizeof(p));
    p.A = up; p.lda = r; p.W = down_t; p.out = W; p.ldo = K; p.M = N; p.N = K; p.K = r;
    p.res = W; p.ldr = K; p.has_gate_scalar = 1; p.gate_scalar = alpha;
    return launch_gemm(EPI_GATE_RES, &p, 1, (hipStream_t)stream);
}

int pe_ln_modulate_e4m3(const void* x, void* out_bf16, void* out_e4m3, float* out_scale, int rows, int dim, int rows_a,
                        const void* shift_a, const void* scale_a, const void* shift_b, const void* scale_b, float eps,
                        void* stream) {
    PE_REQUIRE(out_e4m3 && out_scale, "pe_ln_modulate_e4m3: null e4m3 output");
    return launch_ln_modulate_quant(x, out_bf16, rows, dim, rows_a, shift_a, scale_a, shift_b, scale_b, eps, out_e4m3,
                                    out_scale, (hipStream_t)stream);
}

int pe_quantize_rows_e4m3(const void* x, int ldx, int M, int K, void* out, int Kp, float* scale, void* stream) {
    return launch_quantize_rows_e4m3(x, ldx, M, K, out, Kp, scale, (hipStream_t)stream);
}

int pe_gemm_e4m3(int epilogue, const void* Aq, int lda, const float* scale_a, const void* Wq, const void* bias,
                 const void* pre, int ldp, void* out, int ldo, int M, int N, int K, const void* gate, const void* res,
                 int ldr, void* stream) {
    PE_REQUIRE(epilogue != EPI_QKV, "pe_gemm_e4m3: the QKV epilogue is reached through pe_dit_forward");
    GemmProblem p;
    memset(&p, 0, sizeof(p));
    p.A = Aq; p.W = Wq; p.bias = bias; p.out = out; p.pre = pre; p.ldp = ldp;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldo = ldo;
    p.gate = gate; p.res = res; p.ldr = ldr;
    p.fp8 = 1; p.scale_a = scale_a;
    return launch_gemm(epilogue, &p, 1, (hipStream_t)stream);
}

int pe_gemm_e4m3_gelu_q8(const void* Aq, int lda, const float* scale_a, const void* Wq, const void* bias, void* out, int ldo,
                         void* q8_out, float* q8_scale, unsigned* q8_flags, int M, int N, int K, void* stream) {
    PE_REQUIRE(q8_out && q8_scale && q8_flags, "pe_gemm_e4m3_gelu_q8: null e4m3 output");
    PE_REQUIRE(N % 128 == 0, "pe_gemm_e4m3_gelu_q8: N=%d must be a multiple of 128 (it is the next Linear's K)", N);
    GemmProblem p;
    memset(&p, 0, sizeof(p));
    p.A = Aq; p.W = Wq; p.bias = bias; p.out = out;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldo = ldo;
    p.fp8 = 1; p.scale_a = scale_a;
    p.q8_out = q8_out; p.ldq8 = N; p.q8_flags = q8_flags;
    int rc = launch_gemm(EPI_GELU_SIG, &p, 1, (hipStream_t)stream);
    if (rc) return rc;
    return launch_requant_flagged_rows(out, ldo, M, N, q8_out, N, q8_scale, q8_flags, (hipStream_t)stream);
}

int pe_qkv_rmsnorm_rope(const void* x, int ldx, const void* Wqkv, const void* bqkv, int M, int H, int K,
                        const void* norm_q_w, const void* norm_k_w, const float* rope_cos,
                        const float* rope_sin, void* q_out, void* k_out, void* vt_out, int seq_off, int S_pad,
                        void* stream) {
    GemmProblem p;
    memset(&p, 0, sizeof(p));
    p.A = x; p.W = Wqkv; p.bias = bqkv;
    p.M = M; p.N = 3 * H * 128; p.K = K; p.lda = ldx;
    p.norm_q_w = norm_q_w; p.norm_k_w = norm_k_w; p.rope_cos = rope_cos; p.rope_sin = rope_sin;
    p.q_out = q_out; p.k_out = k_out; p.vt_out = vt_out; p.seq_off = seq_off; p.S_pad = S_pad;
    return launch_gemm(EPI_QKV, &p, 1, (hipStream_t)stream);
}

int pe_flash_attn(const void* q, const void* k, const void* vt, void* out, int H, int S, int S_pad, int ldo,
                  float scale, void* workspace, size_t workspace_bytes, void* stream) {
    return launch_flash_attn(q, k, vt, out, H, S, S_pad, ldo, scale, workspace, workspace_bytes, (hipStream_t)stream);
}

float pe_attn_q_prescale(float scale) { return attn_q_prescale(scale); }

int pe_qkv_rmsnorm_rope_scaled(const void* x, int ldx, const void* Wqkv, const void* bqkv, int M, int H, int K,
                               const void* norm_q_w, const void* norm_k_w, const float* rope_cos,
                               const float* rope_sin, void* q_out, void* k_out, void* vt_out, int seq_off, int S_pad,
                               float q_scale, void* stream) {
    GemmProblem p;
    memset(&p, 0, sizeof(p));
    p.A = x; p.W = Wqkv; p.bias = bqkv;
    p.M = M; p.N = 3 * H * 128; p.K = K; p.lda = ldx;
    p.norm_q_w = norm_q_w; p.norm_k_w = norm_k_w; p.rope_cos = rope_cos; p.rope_sin = rope_sin;
    p.q_out = q_out; p.k_out = k_out; p.vt_out = vt_out; p.seq_off = seq_off; p.S_pad = S_pad;
    p.q_scale = q_scale;
    return launch_gemm(EPI_QKV, &p, 1, (hipStream_t)stream);
}

int pe_flash_attn_prescaled(const void* q, const void* k, const void* vt, void* out, int H, int S, int S_pad, int ldo,
                            float scale, void* workspace, size_t workspace_bytes, void* stream) {
    if (attn_q_prescale(scale) == 1.0f)
        return pe::set_error(PE_ERR_INVALID_ARG, "pe_flash_attn_prescaled: the selected attention variant takes a plain Q (pe_attn_q_prescale() == 1)");
    return launch_flash_attn(q, k, vt, out, H, S, S_pad, ldo, scale, workspace, workspace_bytes, (hipStream_t)stream, nullptr, 0, true);
}

int pe_flash_attn_masked(const void* q, const void* k, const void* vt, void* out, int H, int S, int S_pad, int ldo, float scale,
                         void* workspace, size_t workspace_bytes, const void* token_words, int n_img, void* stream) {
    if (token_words == nullptr) return pe::set_error(PE_ERR_INVALID_ARG, "pe_flash_attn_masked: null token_words");
    return launch_flash_attn(q, k, vt, out, H, S, S_pad, ldo, scale, workspace, workspace_bytes, (hipStream_t)stream, token_words, n_img);
}

size_t pe_flash_attn_workspace_bytes(int H, int S) { return flash_attn_workspace_bytes(H, S); }

size_t pe_flash_attn_fp8_scratch_bytes(int H, int S_pad) { return flash_attn_fp8_scratch_bytes(H, S_pad); }

int pe_flash_attn_fp8(const void* q, const void* k, const void* vt, void* out, int H, int S, int S_pad, int ldo, void* scratch,
                      size_t scratch_bytes, void* workspace, size_t workspace_bytes, void* stream) {
    return launch_flash_attn_fp8(q, k, vt, out, H, S, S_pad, ldo, scratch, scratch_bytes, workspace, workspace_bytes, (hipStream_t)stream);
}

int pe_ln_modulate(const void* x, void* out, int rows, int dim, int rows_a, const void* shift_a,
                   const void* scale_a, const void* shift_b, const void* scale_b, float eps, void* stream) {
    return launch_ln_modulate(x, out, rows, dim, rows_a, shift_a, scale_a, shift_b, scale_b, eps, (hipStream_t)stream);
}

int pe_rmsnorm(const void* x, const void* w, void* out, int rows, int dim, float eps, void* stream) {
    return launch_rmsnorm(x, w, out, rows, dim, eps, (hipStream_t)stream);
}

int pe_dual_rmsnorm_add(const void* x, const void* wx, const void* y, const void* wy, void* out, int rows, int dim, float eps,
                        void* stream) {
    return launch_dual_rmsnorm_add(x, wx, y, wy, out, rows, dim, eps, (hipStream_t)stream);
}

int pe_add_bf16(void* x, const void* y, size_t n, float sign, void* stream) {
    PE_REQUIRE(sign == 1.0f || sign == -1.0f, "pe_add_bf16: sign must be +1 or -1");
    return launch_add_inplace(x, y, n, (hipStream_t)stream, sign);
}

int pe_layernorm_affine(const void* x, const void* weight, const void* bias, void* out, int rows, int dim, float eps, void* stream) {
    return launch_layernorm_affine(x, weight, bias, out, rows, dim, eps, (hipStream_t)stream);
}

int pe_perceiver_attention(const void* q, const void* kv, void* out, int n_queries, int n_keys, int heads, float scale, void* stream) {
    return launch_perceiver_attn(q, kv, out, n_queries, n_keys, heads, scale, (hipStream_t)stream);
}

int pe_sdpa_heads64(const void* q, const void* kv, void* out, int n_queries, int n_keys, int heads, float scale, void* stream) {
    return launch_perceiver_attn(q, kv, out, n_queries, n_keys, heads, scale, (hipStream_t)stream, 1);
}

int pe_gemv_bf16(const void* x, const void* W, const void* bias, void* y, int N, int K, void* stream) {
    return launch_gemv(x, W, bias, y, N, K, (hipStream_t)stream);
}

int pe_gemv_res_bf16(const void* x, const void* W, const void* bias, const void* res, void* y, int N, int K, void* stream) {
    return launch_gemv(x, W, bias, y, N, K, (hipStream_t)stream, res);
}

int pe_gemv_swiglu_bf16(const void* x, const void* Wg, const void* Wu, void* y, int N, int K, void* stream) {
    return launch_gemv_swiglu(x, Wg, Wu, y, N, K, (hipStream_t)stream);
}

int pe_decode_qkv_rope(const void* x, const void* Wq, const void* bq, const void* Wk, const void* bk, const void* Wv, const void* bv,
                       const void* cos_sel, const void* sin_sel, void* q, void* k, void* v, int n_q_heads, int n_kv_heads, int K,
                       void* stream) {
    return launch_decode_qkv(x, Wq, bq, Wk, bk, Wv, bv, cos_sel, sin_sel, q, k, v, n_q_heads, n_kv_heads, K, (hipStream_t)stream);
}

int pe_decode_attention(const void* q, const void* k_cache, const void* v_cache, void* out, int n_q_heads, int n_kv_heads, int L,
                        float scale, void* stream) {
    return launch_attn_decode(q, k_cache, v_cache, out, n_q_heads, n_kv_heads, L, scale, (hipStream_t)stream);
}

int pe_decode_step_qkv(const void* x, const void* Wq, const void* bq, const void* Wk, const void* bk, const void* Wv, const void* bv,
                       const void* cos_table, const void* sin_table, void* q, void* k_cache, void* v_cache, int n_q_heads,
                       int n_kv_heads, int K, const int* step, int base_len, int cache_len, const void* norm_w, float eps,
                       void* stream) {
    PE_REQUIRE(step, "pe_decode_step_qkv: null step counter");
    return launch_decode_qkv(x, Wq, bq, Wk, bk, Wv, bv, cos_table, sin_table, q, k_cache, v_cache, n_q_heads, n_kv_heads, K,
                             (hipStream_t)stream, step, base_len, cache_len, norm_w, eps);
}

int pe_gemv_norm_bf16(const void* x, const void* norm_w, float eps, const void* W, const void* bias, void* y, int N, int K,
                      void* stream) {
    PE_REQUIRE(norm_w, "pe_gemv_norm_bf16: null norm weight");
    return launch_gemv(x, W, bias, y, N, K, (hipStream_t)stream, nullptr, norm_w, eps);
}

int pe_gemv_swiglu_norm_bf16(const void* x, const void* norm_w, float eps, const void* Wg, const void* Wu, void* y, int N, int K,
                             void* stream) {
    PE_REQUIRE(norm_w, "pe_gemv_swiglu_norm_bf16: null norm weight");
    return launch_gemv_swiglu(x, Wg, Wu, y, N, K, (hipStream_t)stream, norm_w, eps);
}

int pe_decode_step_attention(const void* q, const void* k_cache, const void* v_cache, void* out, int n_q_heads, int n_kv_heads,
                             const int* step, int base_len, int cache_len, float scale, void* stream) {
    PE_REQUIRE(step, "pe_decode_step_attention: null step counter");
    return launch_attn_decode(q, k_cache, v_cache, out, n_q_heads, n_kv_heads, cache_len, scale, (hipStream_t)stream, step, base_len);
}

size_t pe_decode_attention_workspace_bytes(int n_q_heads, int cache_len) { return attn_decode_workspace_bytes(n_q_heads, cache_len); }

int pe_decode_step_attention_split(const void* q, const void* k_cache, const void* v_cache, void* out, int n_q_heads, int n_kv_heads,
                                   const int* step, int base_len, int cache_len, float scale, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    return launch_attn_decode_split(q, k_cache, v_cache, out, n_q_heads, n_kv_heads, cache_len, scale, (hipStream_t)stream, step, base_len,
                                    workspace, workspace_bytes);
}

static size_t dl_align(size_t x) { return (x + 255) / 256 * 256; }
size_t pe_decode_layer_scratch_bytes(int n_q_heads, int cache_len, int ff) {
    if (n_q_heads <= 0 || cache_len <= 0 || ff <= 0) return 0;
    const size_t K = (size_t)n_q_heads * 128;
    return 8192 + dl_align(K * 2) * 3 + dl_align((size_t)ff * 2) + dl_align(attn_decode_workspace_bytes(n_q_heads, cache_len));
}
int pe_decode_layer(const pe_decode_layer_weights* w, const void* x, void* x_out, const void* cos_table, const void* sin_table, void* k_cache,
                    void* v_cache, const int* step, int base_len, int cache_len, float scale, void* scratch, size_t scratch_bytes, void* stream) {
    PE_REQUIRE(w && x && x_out && scratch, "pe_decode_layer: null argument");
    PE_REQUIRE(((uintptr_t)scratch & 255) == 0 && scratch_bytes >= pe_decode_layer_scratch_bytes(w->n_q_heads, cache_len, w->ff),
               "pe_decode_layer: scratch of %zu bytes, 256-byte aligned, needed", pe_decode_layer_scratch_bytes(w->n_q_heads, cache_len, w->ff));
    DecodeLayerArgs a;
    memset(&a, 0, sizeof(a));
    const size_t K = (size_t)w->n_q_heads * 128;
    char* p = (char*)scratch;
    a.err = (unsigned*)p; a.bar = (unsigned*)(p + 256); p += 8192;      // [0] error flag; 6 barriers x 9 lines of counters behind it
    a.q = (bf16*)p; p += dl_align(K * 2);
    a.a = (bf16*)p; p += dl_align(K * 2);
    a.h1 = (bf16*)p; p += dl_align(K * 2);
    a.hid = (bf16*)p; p += dl_align((size_t)w->ff * 2);
    a.sc_g = (float*)p;
    a.mx_g = a.sc_g + (size_t)w->n_q_heads * cache_len;
    a.red_g = a.mx_g + (size_t)w->n_q_heads * 16;
    a.part_g = a.red_g + (size_t)w->n_q_heads * 16;
    a.x = (const bf16*)x; a.x_out = (bf16*)x_out;
    a.Wq = (const bf16*)w->q_w; a.bq = (const bf16*)w->q_b; a.Wk = (const bf16*)w->k_w; a.bk = (const bf16*)w->k_b;
    a.Wv = (const bf16*)w->v_w; a.bv = (const bf16*)w->v_b; a.Wo = (const bf16*)w->o_w; a.Wg = (const bf16*)w->gate_w;
    a.Wu = (const bf16*)w->up_w; a.Wd = (const bf16*)w->down_w; a.ln1 = (const bf16*)w->input_norm_w; a.ln2 = (const bf16*)w->post_norm_w;
    a.eps1 = w->input_norm_eps; a.eps2 = w->post_norm_eps;
    a.cs = (const bf16*)cos_table; a.sn = (const bf16*)sin_table; a.Kc = (bf16*)k_cache; a.Vc = (bf16*)v_cache;
    a.step = step; a.base = base_len; a.ld = cache_len; a.n_q = w->n_q_heads; a.n_kv = w->n_kv_heads; a.scale = scale;
    a.K = (int)K; a.FF = w->ff;
    return launch_decode_layer(a, (hipStream_t)stream);
}

int pe_decode_embed(const void* table, const int* token, void* x, int dim, int vocab, void* stream) {
    return launch_embed_row(table, token, x, dim, vocab, (hipStream_t)stream);
}

int pe_decode_argmax(const void* logits, int vocab, int* token, int* out_ids, int* step, int max_steps, void* stream) {
    return launch_argmax_step(logits, vocab, token, out_ids, step, max_steps, (hipStream_t)stream);
}

int pe_patchify(const void* latents, void* tokens, int C, int H2, int W2, void* stream) {
    return launch_patchify(latents, tokens, C, H2, W2, (hipStream_t)stream);
}

int pe_unpatchify(const void* tokens, void* latents, int C, int H2, int W2, void* stream) {
    return launch_unpatchify(tokens, latents, C, H2, W2, (hipStream_t)stream);
}

int pe_cfg_euler_step(const void* posi, const void* nega, const void* latents, void* latents_out, size_t n,
                      float cfg_scale, int use_cfg, float dsigma, void* stream) {
    return launch_cfg_euler(posi, nega, latents, latents_out, n, cfg_scale, use_cfg, dsigma, (hipStream_t)stream);
}

int pe_cfg_inpaint_euler_step(const void* posi, const void* nega, const void* latents, const void* input_latents,
                              const void* inpaint_mask, void* latents_out, size_t n, size_t plane, float cfg_scale, int use_cfg,
                              float sigma, float dsigma, void* stream) {
    PE_REQUIRE(input_latents && inpaint_mask, "pe_cfg_inpaint_euler_step: null input latents / mask");
    return launch_cfg_euler(posi, nega, latents, latents_out, n, cfg_scale, use_cfg, dsigma, (hipStream_t)stream, input_latents,
                            inpaint_mask, plane, sigma);
}

int pe_conv2d_nhwc(const void* in, const void* w, const void* bias, const void* res, void* out, const void* zero_page,
                   int Hin, int Win, int Cin_p, int Cout_p, int ksize, int stride, int upsample2x, void* stream) {
    return launch_conv_nhwc(in, w, bias, res, out, zero_page, Hin, Win, Cin_p, Cout_p, ksize, stride, upsample2x,
                            (hipStream_t)stream);
}

int pe_vae_rmsnorm(const void* x, const void* gamma, void* out, int npix, int C, int Cp, int silu, void* stream) {
    return launch_vae_rmsnorm(x, gamma, out, npix, C, Cp, silu, (hipStream_t)stream);
}

int pe_nchw_to_nhwc(const void* in, void* out, int C, int HW, int Cp, int mode, const void* ta, const void* tb,
                    void* stream) {
    return launch_nchw_to_nhwc(in, out, C, HW, Cp, mode, ta, tb, (hipStream_t)stream);
}

int pe_nhwc_to_nchw(const void* in, void* out, int C, int HW, int Cp, int mode, const void* ta, const void* tb,
                    void* stream) {
    return launch_nhwc_to_nchw(in, out, C, HW, Cp, mode, ta, tb, (hipStream_t)stream);
}

size_t pe_vae_attention_scratch_bytes(int N) { return vae_attention_scratch_bytes(N); }
int pe_vae_attention(const void* qkv, void* vt_scratch, void* out, int N, void* stream) {
    return launch_vae_attention(qkv, vt_scratch, out, N, (hipStream_t)stream);
}

}  // extern "C"
