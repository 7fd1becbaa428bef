// Host-side launcher interface shared by the .hip translation units (internal; the public C-ABI
// is include/physicedit_amd.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace pe {

// ---------------------------------------------------------------------------------------------
// GEMM:  out[M,N] = epilogue( A[M,K] . W[N,K]^T + bias[N] )      bf16 in, fp32 accumulate
// ---------------------------------------------------------------------------------------------
enum GemmEpilogue {
    EPI_BIAS = 0,      // y = bf16(acc + bias)
    EPI_GELU_SIG = 1,  // ApproximateGELU: y * sigmoid(1.702 y), each op rounded (qwen_image_dit.py:47-49)
    EPI_GELU_ERF = 2,  // nn.GELU() exact erf (helpers.py:127-131)
    EPI_GATE_RES = 3,  // out = res + gate[n] * y, each op rounded (qwen_image_dit.py:386-387,398-399)
    EPI_QKV = 4,       // per-head RMSNorm(q,k) + RoPE(q,k), head-major Q/K and transposed V
    EPI_SILU = 5,      // y -> silu(y) (time MLP linear_1 + SiLU; AdaLN's silu(temb) is a separate tensor)
    EPI_QKV_STATS = 6, // EPI_QKV that also leaves the sum and the sum of squares of what it wrote (round 6: the e4m3 attention's global standard
                       // deviations of q, k, v without a pass over the 160 MB it has just written; only pe_dit_forward with fp8_attention takes it)
};

struct GemmProblem {
    const void* A;     // [M,K] bf16, row stride lda elements
    const void* W;     // [N,K] bf16 contiguous (nn.Linear layout)
    const void* bias;  // [N] bf16 or null
    void* out;         // [M,N] bf16, row stride ldo (unused by EPI_QKV)
    int M, N, K;
    int lda, ldo;
    // EPI_GATE_RES
    const void* gate;  // [N] bf16 (null => the scalar gate below, default 1)
    int has_gate_scalar;   // gate == null: out = res + bf16(gate_scalar * y) with an fp32 scalar (LoRA merge alpha)
    float gate_scalar;
    const void* res;   // [M,N] bf16, row stride ldr (may alias out)
    int ldr;
    // optional for every epilogue: y = bf16(pre[m][n] + bf16(acc + bias)) before the epilogue proper
    const void* pre;   // [M,N] bf16, row stride ldp, or null
    int ldp;
    // EPI_QKV: N = 3*H*128; columns [0,HD) q, [HD,2HD) k, [2HD,3HD) v
    const void* norm_q_w;   // [128] bf16
    const void* norm_k_w;   // [128] bf16
    const float* rope_cos;  // [M,64] fp32 (row = token of THIS problem)
    const float* rope_sin;  // [M,64]
    void* q_out;            // [H][S_pad][128] bf16
    void* k_out;            // [H][S_pad][128]
    void* vt_out;           // [H][128][S_pad], tokens permuted inside 16-groups (see attention.hip)
    float q_scale;          // Q is stored as bf16(rope(q) * q_scale), the factor applied in fp32 before that one rounding; 0 = 1
    int seq_off;            // joint-sequence row of this problem's row 0
    int S_pad;
    // EPI_QKV_STATS: partial sums [section 3][slot 2][stat_rb][H][2] doubles (sum, sum of squares of the bf16 outputs of one 64-row block and
    // head, fp32 per lane over its <= 512 values, then double); the caller zeroes the buffer (blocks that do not exist stay zero)
    double* qkv_stats;
    int stat_rb, stat_slot;
    int tilesM, tilesN;     // filled by the launcher
    // e4m3 operands (fp8_linear, vram_management/layers.py:115-151): A [M,K] and W [N,K] are OCP e4m3 bytes
    // (lda in elements = bytes, K % 128 == 0); y = bf16(acc * scale_a[m] + bias[n]) before the epilogue proper
    int fp8;
    const float* scale_a;   // [M] fp32
    // e4m3 operands + EPI_GELU_SIG: the NEXT Linear's activation operand, produced here (fp8_linear's row quantisation of this
    // output, fused): q8_out [M, ldq8] bytes receives e4m3(out) -- exact whenever the row's scale is 1, i.e. max|row| / 448
    // rounds to <= 1 -- and q8_flags[m] is raised for rows holding a value above 447 (launch_requant_flagged_rows redoes those
    // rows from the bf16 output and writes every row's scale).  null: off.
    void* q8_out;
    int ldq8;
    unsigned* q8_flags;     // [M] (indexed like scale_a: row of THIS problem), zero between launches
};

// scratch of the stream-K schedule (gemm.hip, schedule 19): `bytes` >= gemm_workspace_bytes(), 256-byte aligned, ZERO when first used
// (the kernels leave it zero), private to one stream at a time.  Without it the launch runs schedules 15 / 17.
struct GemmWorkspace {
    void* sync;
    size_t bytes;
    // round 6: the deferred epilogue's stash (gemm_tile.h): `stash_bytes` >= gemm_stash_bytes(), 256-byte aligned, any contents, private to one
    // stream at a time.  Without it the persistent schedule runs every epilogue at its tile's end.
    void* stash;
    size_t stash_bytes;
};
size_t gemm_workspace_bytes();
size_t gemm_stash_bytes();
extern GemmWorkspace g_gemm_ws;
// the GEMM schedule production runs (pe_debug_set("gemm_variant", 0) returns to it): 21 since round 5 (17 with one hand-off per K tile; with the
// 16 x 16 MFMA shapes it is 2 - 4 % faster per block Linear and 0.7 - 1.2 % per image than 17: profiles/r05_gemm_notes.md section 7)
constexpr int GEMM_DEFAULT_VARIANT = 21;
extern int g_gemm_sk, g_gemm_persist_min_rounds, g_gemm4_x, g_gemm_skip_ragged, g_gemm_direct_epi, g_gemm_mfma16, g_gemm_no_epi, g_gemm_cont, g_gemm_defer;
int launch_gemm(int epilogue, GemmProblem* problems, int nproblems, hipStream_t stream, const GemmWorkspace* workspace = nullptr);
extern int g_gemm_band;          // M tiles per band of the tile order (default 4)
extern int g_gemm_persist_wgs;   // schedule 17: work-groups of the persistent grid (0 = one per CU)
extern int g_gemm_variant;  // experiment knobs (pe_debug_set); production paths use the compiled defaults
extern int g_attn_variant;
extern long long* g_attn_dbg;
extern long long* g_gemm_dbg;   // device buffer for the time stamps of the profiling GEMM variant (14), or null

// ---------------------------------------------------------------------------------------------
// flash attention over the joint sequence (no mask), D = 128
// ---------------------------------------------------------------------------------------------
int launch_flash_attn(const void* q, const void* k, const void* vt, void* out, int H, int S, int S_pad,
                      int ldo, float scale, void* workspace, size_t workspace_bytes, hipStream_t stream,
                      const void* words = nullptr, int n_img = 0,    // words: EliGen token words (attention.hip), or null
                      bool q_prescaled = false,                       // Q was written with GemmProblem.q_scale = attn_q_prescale(scale)
                      int S_q = 0);                                   // > 0: only the query rows [0, S_q) are wanted (keys stay [0, S)); rows of `out` beyond
                                                                      // the last 256-row block that holds a wanted row are left untouched
float attn_q_prescale(float scale);   // scale * log2(e) when the current attention variant wants it folded into Q, else 1
size_t flash_attn_workspace_bytes(int H, int S);
// the default attention kernel's schedule without its softmax (attainable-ceiling probe; the output is meaningless)
int launch_attn_mix_probe(const void* q, const void* k, const void* vt, void* out, int H, int S, int S_pad, int ldo, hipStream_t stream);
// the reference's enable_fp8_attention branch (attention.hip, "e4m3 attention"): q, k, vt in the bf16 layouts of launch_flash_attn (plain
// Q); scratch: flash_attn_fp8_scratch_bytes(H, S_pad), 256-byte aligned (e4m3 copies, the three std and their partial sums)
size_t flash_attn_fp8_scratch_bytes(int H, int S_pad);
// EPI_QKV_STATS' partial sums for a joint sequence of up to S_pad rows in two problems (image, text): row blocks per slot, bytes
inline int qkv_stats_row_blocks(int S_pad) { return S_pad / 64 + 4; }
inline size_t qkv_stats_bytes(int H, int S_pad) { return (size_t)3 * 2 * qkv_stats_row_blocks(S_pad) * H * 2 * sizeof(double); }
int launch_flash_attn_fp8(const void* q, const void* k, const void* vt, void* out, int H, int S, int S_pad, int ldo, void* scratch,
                          size_t scratch_bytes, void* workspace, size_t workspace_bytes, hipStream_t stream, int S_q = 0,
                          const double* qkv_stats = nullptr);      // the QKV epilogue's partial sums (EPI_QKV_STATS): no statistics pass over q / k / vt
extern int g_dit_qkv_stats;           // dit.hip: 1 (default) = the e4m3 attention's statistics come from the QKV epilogue (knob "dit_qkv_stats")
extern int g_dit_trim_last_block;     // dit.hip: 1 (default) = the last block computes only what survives it (knob "dit_trim_last_block")
extern int g_attn_slots, g_attn_force_split, g_attn_fp8_variant;

// ---------------------------------------------------------------------------------------------
// row kernels / elementwise
// ---------------------------------------------------------------------------------------------
// LayerNorm(no affine, eps) -> *(1+scale) -> +shift, rows [0,rows_a) use mod_a, the rest mod_b
int launch_ln_modulate(const void* x, void* out, int rows, int dim, int rows_a, const void* shift_a,
                       const void* scale_a, const void* shift_b, const void* scale_b, float eps,
                       hipStream_t stream);
int launch_ln_modulate_quant(const void* x, void* out, int rows, int dim, int rows_a, const void* shift_a,
                             const void* scale_a, const void* shift_b, const void* scale_b, float eps, void* q_out,
                             float* q_scale, hipStream_t stream);
int launch_rmsnorm(const void* x, const void* w, void* out, int rows, int dim, float eps, hipStream_t stream);
// fp8_linear's activation quantisation: scale[m] = max(bf16(max|x[m,:]| * (1/448)), 1); out = e4m3(x / (scale + 1e-8)),
// columns [K, Kp) zero-filled
int launch_quantize_rows_e4m3(const void* x, int ldx, int M, int K, void* out, int Kp, float* scale, hipStream_t stream);
// second half of the fused quantisation (GemmProblem.q8_out): scale[m] = 1 for the rows whose flag is clear; flagged rows are
// quantised from the bf16 row like launch_quantize_rows_e4m3 does, and their flag is cleared
int launch_requant_flagged_rows(const void* x, int ldx, int M, int K, void* out, int Kp, float* scale, unsigned* flags,
                                hipStream_t stream);
int launch_silu(const void* x, void* out, size_t n, hipStream_t stream);
int launch_dual_rmsnorm_add(const void* x, const void* wx, const void* y, const void* wy, void* out, int rows, int dim, float eps,
                            hipStream_t stream);
int launch_add_inplace(void* x, const void* y, size_t n, hipStream_t stream, float sign = 1.0f);
int launch_layernorm_affine(const void* x, const void* w, const void* b, void* out, int rows, int dim, float eps, hipStream_t stream);
int launch_perceiver_attn(const void* q, const void* kv, void* out, int nq, int nk, int heads, float scale, hipStream_t stream,
                          int sdpa = 0);
int launch_gemv(const void* x, const void* W, const void* bias, void* y, int N, int K, hipStream_t stream, const void* res = nullptr,
                const void* norm_w = nullptr, float eps = 0.f);
int launch_decode_qkv(const void* x, const void* Wq, const void* bq, const void* Wk, const void* bk, const void* Wv,
                      const void* bv, const void* cos_sel, const void* sin_sel, void* q, void* k, void* v, int n_q_heads,
                      int n_kv_heads, int K, hipStream_t stream, const int* step = nullptr, int base = 0, int ld = 0,
                      const void* norm_w = nullptr, float eps = 0.f);
size_t attn_decode_workspace_bytes(int n_q_heads, int cache_len);
// elementwise.hip: one decoder layer of the decode step in one launch (round 6)
struct DecodeLayerArgs {
    const bf16* x;          // layer input [K]
    bf16* x_out;            // layer output [K]
    const bf16 *Wq, *bq, *Wk, *bk, *Wv, *bv, *Wo, *Wg, *Wu, *Wd, *ln1, *ln2;
    float eps1, eps2;
    const bf16 *cs, *sn;    // rotary tables [steps][128]
    bf16 *Kc, *Vc;          // caches [n_kv][ld][128]
    const int* step;
    int base, ld, n_q, n_kv;
    float scale;
    bf16 *q, *a, *h1, *hid; // scratch rows: [n_q*128], [n_q*128], [K], [FF]
    float *sc_g, *mx_g, *red_g, *part_g;      // the split attention's workspace (attn_decode_workspace_bytes)
    unsigned* bar;          // 8 counters, zero at rest
    unsigned* err;
    int K, FF;
};
int launch_decode_layer(const DecodeLayerArgs& a, hipStream_t stream);
extern int g_decode_layer_wgs_per_cu, g_decode_layer_no_barrier;
int launch_attn_decode_split(const void* q, const void* Kc, const void* Vc, void* out, int n_q_heads, int n_kv_heads, int cache_len,
                             float scale, hipStream_t stream, const int* step, int base, void* workspace, size_t workspace_bytes);
int launch_attn_decode(const void* q, const void* Kc, const void* Vc, void* out, int n_q_heads, int n_kv_heads, int L, float scale,
                       hipStream_t stream, const int* step = nullptr, int base = 0);
int launch_adapter_mse(const void* pred_dino, const void* gt_dino, const void* pred_vae, const void* gt_vae, size_t n, float* out,
                       hipStream_t stream);
int launch_embed_row(const void* table, const int* token, void* x, int dim, int vocab, hipStream_t stream);
int launch_argmax_step(const void* logits, int V, int* token, int* out_ids, int* step, int max_steps, hipStream_t stream);
int launch_gemv_swiglu(const void* x, const void* Wg, const void* Wu, void* y, int N, int K, hipStream_t stream,
                       const void* norm_w = nullptr, float eps = 0.f);
int launch_patchify(const void* latents, void* tokens, int C, int H2, int W2, hipStream_t stream);
int launch_unpatchify(const void* tokens, void* latents, int C, int H2, int W2, hipStream_t stream);
int launch_gather_rows(const void* src, const int* idx, void* dst, int nrows, int dim, hipStream_t stream);
int launch_adapter_mix_scatter(const void* dino, const void* vae, float alpha, float one_minus_alpha,
                               const int* idx, void* prompt_emb, int nrows, int dim, hipStream_t stream);
int launch_cfg_euler(const void* posi, const void* nega, const void* latents, void* out, size_t n,
                     float cfg_scale, int use_cfg, float dsigma, hipStream_t stream,
                     const void* input_latents = nullptr, const void* mask = nullptr, size_t plane = 0, float sigma = 0.f);

// ---------------------------------------------------------------------------------------------
// VAE (vae.hip): NHWC bf16 activations, channels padded to a multiple of 32
// ---------------------------------------------------------------------------------------------
int launch_conv_nhwc(const void* in, const void* w, const void* bias, const void* res, void* out, const void* zero,
                     int Hin, int Win, int Cin_p, int Cout_p, int ksize, int stride, int upsample,
                     hipStream_t stream);
int launch_vae_rmsnorm(const void* x, const void* gamma, void* out, int npix, int C, int Cp, int silu,
                       hipStream_t stream);
int launch_nchw_to_nhwc(const void* in, void* out, int C, int HW, int Cp, int mode, const void* ta, const void* tb,
                        hipStream_t stream);
int launch_nhwc_to_nchw(const void* in, void* out, int C, int HW, int Cp, int mode, const void* ta, const void* tb,
                        hipStream_t stream);
size_t vae_attention_scratch_bytes(int N);      // Vt + (when the keys are split over several work-groups per query block) the fp32 partials
int launch_vae_attention(const void* qkv, void* vt_scratch, void* out, int N, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// optional in-library timing (profile.hip): HIP events around sampled launches
// ---------------------------------------------------------------------------------------------
enum { PROF_GEMM = 0, PROF_ATTN = 1, PROF_ROW = 2, PROF_CONV = 3, PROF_KINDS = 4 };
int prof_begin(int kind, double work, hipStream_t stream);  // -> slot or -1
void prof_end(int slot, hipStream_t stream);

}  // namespace pe
