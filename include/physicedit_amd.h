/* physicedit_amd -- C-ABI of the MI355X-native PhysicEdit denoising hot path.
 *
 * The reference (liangbingzhao/PhysicEdit) is pure Python: it has no FFI of its own.  Its operator
 * boundary for this path is the Python call
 *     model_fn_qwen_image(dit=, visual_thinking_adapter=, latents=, timestep=, prompt_emb=, ...)
 *         DiffSynth-Studio/diffsynth/pipelines/qwen_image_physical.py:1302-1403
 * driven by QwenImagePhysicPipeline.__call__ (:644-661) plus vae.encode / vae.decode
 *         DiffSynth-Studio/diffsynth/models/qwen_image_vae.py:706,719.
 * This header is what a ctypes binding of those call sites binds instead (INTEGRATION.md shows the
 * stub).  Conventions:
 *   - every data pointer is a DEVICE pointer into a caller-owned buffer (a torch tensor's
 *     data_ptr()); the library borrows it for the call and never frees it;
 *   - bf16 tensors are contiguous unless a row stride is given; shapes are explicit ints;
 *   - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream) passed as void*;
 *   - every entry point returns 0 on success, <0 on failure (pe_last_error() has the message);
 *     nothing throws across the boundary; no entry point synchronises the device;
 *   - scratch memory is a caller-provided workspace (size from pe_dit_workspace_bytes);
 *   - threads: operators and composites may be called concurrently from several host threads as long as each pe_dit / pe_vae
 *     handle (and its workspace) is used by one thread at a time -- a handle is the unit of re-entrancy, a second pe_dit_create() on the
 *     same weight table gives another one (weights are borrowed, not copied); pe_last_error() is per thread.  The experiment knobs (pe_debug_set*) and the event profiler
 *     (pe_profile_*) are process-global measurement tools: set them while no other thread is launching.
 */
#ifndef PHYSICEDIT_AMD_H
#define PHYSICEDIT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PE_OK 0
#define PE_ERR_INVALID_ARG (-1)
#define PE_ERR_UNSUPPORTED (-2)
#define PE_ERR_HIP (-3)

/* Last error message of the calling thread ("" if none). */
const char* pe_last_error(void);
/* ABI version of this header; bumped on any signature, struct or documented-semantics change.
 * 8 (round 6): counts round 5's additions (pe_decode_attention_workspace_bytes, pe_decode_step_attention_split, pe_flash_attn_fp8
 * ignoring what the planes' pad columns hold) and round 6's (pe_dit_forward runs the LAST block only on the rows that survive it:
 * the text stream and the edit-image rows of the handle's residual buffer keep block L - 2's values -- knob "dit_trim_last_block"). */
int pe_abi_version(void);
/* Hash of the kernel sources this library was built from (physicedit_amd/build.py:source_hash). */
const char* pe_build_id(void);
/* Experiment knobs for in-process A/B benchmarking of kernel schedules ("gemm_variant", "attn_variant", ...).
 * Production callers never need it: the compiled defaults are the validated schedules.
 * "gemm_variant": 21 default since round 5 (persistent work-groups with cross-tile prefetch, ONE matrix-pipe hand-off per K tile between the two
 * wave groups; launches of at most one round of tiles run 15 -- "gemm_persist_min_rounds", default 1; value 0 = back to the default); 17 = two
 * hand-offs per K tile (the round-3/4 default); 15 = one tile per work-group (the round-2 default; writes s_memtime stamps when "gemm_stamps"
 * is attached); 19 = stream-K (bit-identical, needs a workspace: pe_gemm_workspace_bytes; "gemm_sk" = 1 lets 17 / 21 take it where tiles
 * do not fill whole rounds; measured slower, default off); 22 = four waves x 128 x 128, one wave per SIMD, one tile per work-group (gemm4.hip,
 * 32 x 32 MFMA blocks).  15 / 17 / 19 / 21 are bit-identical with each other.  "gemm_mfma16" (bit mask, default 3): bit 0 = the bf16 8-wave schedules
 * use v_mfma_f32_16x16x32_bf16 (12 - 16 % more FLOP/s under the power limit than 32x32x16, -10 .. -13 % per block Linear: profiles/r05_gemm_notes.md
 * section 7), bit 1 = the e4m3 ones v_mfma_scale_f32_16x16x128_f8f6f4 (slower per Linear in isolation, 1 % faster per image); a clear bit = that
 * dtype on the 32 x 32 blocks of rounds 1 - 4, schedules 15 / 17 (another summation order inside a K tile: same accuracy, not bit-identical with
 * the default, bit-identical with 22).  "gemm_skip_ragged" (default 1): 32-row blocks beyond M skip their MFMAs;
 * "gemm_direct_epilogue" (bit mask, default 1): bit 0 = complete tiles of the GELU / gate + residual epilogues skip the LDS round trip; bit 1 = so do
 * the q / k sections of the QKV epilogue (bit-identical, 1.2 % slower: off).  "gemm_band": M tiles per band of
 * the XCD-aware tile order (default 4).  "gemm_persist_wgs": work-groups of the persistent schedules' grid (0 = one per CU).
 * "dit_qkv_stats" (default 1): with fp8_attention the QKV epilogue leaves the sums the e4m3 attention's global q / k / v standard deviations
 * need (bit-identical to the separate pass, 0, which reads the three tensors again).
 * "dit_trim_last_block" (default 1): pe_dit_forward launches of the last block only what reaches the output -- the S0 noise rows' attention
 * queries, out-projection, norm2 and MLP (the reference slices image[:, :S0] behind it: qwen_image_physical.py:1398-1402); 0 = the whole block
 * (tests that read the text stream behind it).
 * "attn_variant": 5 default (4 waves x 64 query rows, one wave per SIMD, lazy running max, the softmax scale folded into Q and the max
 * fed through the MFMA C operand: pe_attn_q_prescale / pe_flash_attn_prescaled; same distance to an fp32 result as the reference's own
 * bf16 SDPA, profiles/r04_attention_notes.md); 6 = 5 with the textbook max update; 4 / 3 = the same schedule with scale and max applied
 * per score (round 3's default / its textbook form, bit-identical to 0); 7 = the schedule of 5 on v_mfma_f32_16x16x32_bf16 (round 5: 16 x 16
 * blocks, softmax denominators through the matrix pipe; 2 % faster alone, 5 % slower per image: opt-in, profiles/r05_attention_notes.md),
 * 8 = 7 with the textbook max update; 0 = 8 waves x 32 query rows, textbook update (the round-1/2 default, and always the kernel of the
 * masked form).  "attn_slots", "attn_force_split": load-balancing knobs of the tests.
 * "attn_fp8_variant": kernel of pe_flash_attn_fp8: 1 default (software-pipelined: the MFMAs of S(t+1) and P(t-1) V(t-1) ride between the
 * slices of the softmax of S(t); the softmax reference is raised lazily, by tiles whose maximum exceeds it by 2^8 -- oracle:
 * flash_attention_fp8(kv_tile=64, lazy_tau_log2=8)); 2 = 1 with a maximum-free fast path (a row keeps its reference while its 64 P of
 * the tile sum to 448 at most, else the wave redoes the tile with the maximum; oracle: lazy_sum_limit=448; -5 % alone, -0.4 % per
 * configs[2] image); 3 = the arithmetic of 1, value for value, with one wave per SIMD (4 waves x 64 query rows, O and Q in fixed
 * accumulator registers; round 6: bit-identical with 1, the same time alone, +0.5 % per image); 4 = 3 with the row sums taken by the
 * matrix pipe from the e4m3 P (l += ones . P: 64 v_add_f32 per lane and tile fewer; oracle: row_sum_quantised=True; -10 % alone,
 * -1.5 % per configs[2] image; not FlashAttention-3's published form, hence opt-in); 0 = the plain kernel (running maximum; oracle:
 * kv_tile=64). */
int pe_debug_set(const char* key, int value);
/* Device buffer for a profiling variant's output ("gemm_stamps": long long [work-groups][8] s_memtime stamps of
 * gemm variant 15; "attn_stamps": long long [work-groups][10] of attention variants 3 - 6 built with -DPE_W4_STAMPS=1; "gemm_workspace": stream-K scratch for the granular pe_gemm_* calls);
 * NULL detaches it. */
int pe_debug_set_ptr(const char* key, void* device_ptr);

/* ---------------------------------------------------------------------------------------------
 * Granular operators (each is one kernel launch; used by the parity tests and by the composites)
 * ------------------------------------------------------------------------------------------- */

/* epilogue selector for pe_gemm_bf16 */
enum {
    PE_EPI_BIAS = 0,      /* torch.nn.functional.linear                                              */
    PE_EPI_GELU_SIG = 1,  /* ApproximateGELU  qwen_image_dit.py:42-49   y*sigmoid(1.702 y)            */
    PE_EPI_GELU_ERF = 2,  /* nn.GELU()        pipelines/helpers.py:127-131                            */
    PE_EPI_GATE_RES = 3,  /* res + gate*y     qwen_image_dit.py:386-387,398-399 ; lora/__init__.py:41 */
    PE_EPI_SILU = 5       /* Linear + SiLU    models/utils.py:267-269                                 */
};

/* Bytes of the stream-K scratch of the GEMM (schedule 19: a launch's tiles x K tiles are cut into equal ranges per CU, a tile
 * split between two CUs hands its fp32 accumulators over through this buffer; outputs are bit-identical to the unsplit schedules).
 * pe_dit_workspace_bytes() includes one per handle.  The granular pe_gemm_* calls run the unsplit schedules unless a test installs
 * a zeroed buffer with pe_debug_set_ptr("gemm_workspace", p). */
size_t pe_gemm_workspace_bytes(void);
/* Bytes of the deferred-epilogue stash of the persistent GEMM schedule (round 6 experiment: a tile parks bf16(gate * y) there and its
 * gate + residual epilogue runs inside the next tile's main loop; outputs bit-identical).  0 unless the library was built with
 * -DPE_GEMM_DEFER (measured slower per image: profiles/r06_gemm_notes.md); then 128 KiB per CU, any contents, and
 * pe_dit_workspace_bytes() includes one per handle.  The granular pe_gemm_* calls run every epilogue at its tile's end unless a test
 * installs a stash with pe_debug_set_ptr("gemm_stash", p). */
size_t pe_gemm_stash_bytes(void);
/* out[M,N] = epilogue(A[M,K] @ W[N,K]^T + bias[N]); bf16, fp32 accumulate.
 * gate[N] (nullable => 1) and res[M,N] (row stride ldr, may alias out) only for PE_EPI_GATE_RES.
 * Requires K % 64 == 0, N % 8 == 0, lda/ldo/ldr % 8 == 0. */
int pe_gemm_bf16(int epilogue, const void* A, int lda, const void* W, const void* bias, void* out, int ldo,
                 int M, int N, int K, const void* gate, const void* res, int ldr, void* stream);

/* As pe_gemm_bf16 with a pre-add operand: y = bf16(pre[m][n] + bf16(A@W^T + bias)) and THEN the epilogue.
 * This is the runtime ("hot") LoRA step `out + x @ lora_A.T @ lora_B.T` of AutoWrappedLinear.forward
 * (vram_management/layers.py:173-181): pre = the base Linear's output, A = x @ lora_A.T, W = lora_B. */
int pe_gemm_bf16_pre(int epilogue, const void* A, int lda, const void* W, const void* bias, const void* pre, int ldp,
                     void* out, int ldo, int M, int N, int K, const void* gate, const void* res, int ldr, void* stream);

/* GeneralLoRALoader.load (lora/__init__.py:28-45) for one target Linear, in place and with the reference's roundings:
 * W[N,K] = bf16(W + bf16(alpha * bf16(up[N,r] @ down[r,K]))).  down_t is down transposed, [K,r]; r is the rank zero-padded
 * to a multiple of 64 (zero columns add exact zeros); alpha is applied as an fp32 scalar like `alpha * tensor` in torch. */
int pe_lora_merge(void* W, int N, int K, const void* up, const void* down_t, int r, float alpha, void* stream);

/* e4m3 ("FP8 computation") Linear: AutoWrappedLinear.fp8_linear, vram_management/layers.py:115-151, reached when the
 * DiT is stored in float8_e4m3fn and enable_vram_management(enable_dit_fp8_computation=True) is on
 * (pipelines/qwen_image_physical.py:440-496).  Two launches per Linear:
 *   pe_quantize_rows_e4m3:  scale[m] = max(bf16(max|x[m,:]| * (1/448)), 1);  out[m,:K] = e4m3fn(x[m,:] / (scale[m] + 1e-8)),
 *                           out[m,K:Kp] = 0.  x bf16 [M,K] (row stride ldx), out bytes [M,Kp], Kp % 128 == 0, K % 8 == 0.
 *   pe_gemm_e4m3:           y = bf16(acc * scale_a[m] + bias[n]) with acc = Aq[M,K] . Wq[N,K]^T in fp32 on the
 *                           block-scaled CDNA4 MFMA (unit block scales), then pre-add / epilogue as pe_gemm_bf16_pre.
 *                           Aq, Wq are OCP e4m3fn bytes; K % 128 == 0; lda % 16 == 0; bias/pre/gate/res/out bf16.
 * torch._scaled_mm, which the reference calls here, does not run on CPU with per-row scales, so there is no CPU fixture; the
 * pair is pinned on the GPU: the quantiser bit-exact against torch's device ops, pe_gemm_e4m3 bit-identical to
 * torch._scaled_mm on the same operands (tests/test_gpu_fp8.py, profiles/r02_parity.json). */
int pe_quantize_rows_e4m3(const void* x, int ldx, int M, int K, void* out, int Kp, float* scale, void* stream);
/* pe_ln_modulate followed by pe_quantize_rows_e4m3 in one pass over the row (the form the e4m3 composite uses in front of
 * the QKV and MLP-up Linears): out_e4m3 [rows,dim] bytes + out_scale [rows]; out_bf16 nullable (skipped when null).
 * Bit-identical to the two separate calls. */
int pe_ln_modulate_e4m3(const void* x, void* out_bf16, void* out_e4m3, float* out_scale, int rows, int dim, int rows_a,
                        const void* shift_a, const void* scale_a, const void* shift_b, const void* scale_b, float eps,
                        void* stream);
int pe_gemm_e4m3(int epilogue, const void* Aq, int lda, const float* scale_a, const void* Wq, const void* bias,
                 const void* pre, int ldp, void* out, int ldo, int M, int N, int K, const void* gate, const void* res,
                 int ldr, void* stream);
/* The MLP-up Linear of a block in e4m3 mode with the NEXT Linear's activation quantisation fused: out = ApproximateGELU(fp8_linear(x))
 * in bf16 as pe_gemm_e4m3(PE_EPI_GELU_SIG) writes it, plus that output's fp8_linear row quantisation -- q8_out [M,N] e4m3 bytes and
 * q8_scale [M] -- without a second pass over it: the epilogue stores e4m3(out), exact for every row whose scale is 1 (max|row| <= 447),
 * and raises q8_flags[m] for the others, which a second, row-wise launch re-quantises from `out` (it also writes every scale and
 * lowers the flags).  Bit-identical to pe_quantize_rows_e4m3(out).  q8_flags: [M] uint32, zero on entry and on return. */
int pe_gemm_e4m3_gelu_q8(const void* Aq, int lda, const float* scale_a, const void* Wq, const void* bias, void* out, int ldo,
                         void* q8_out, float* q8_scale, unsigned* q8_flags, int M, int N, int K, void* stream);

/* Fused QKV projection of one stream (QwenDoubleStreamAttention.forward, qwen_image_dit.py:282-302):
 * x[M,K] @ Wqkv[3*H*128,K]^T + b, per-head RMSNorm(q,k) (weights norm_q_w/norm_k_w [128]), RoPE(q,k)
 * with fp32 tables rope_cos/rope_sin [M,64]; writes head-major Q,K [H][S_pad][128] at rows
 * seq_off..seq_off+M and the transposed/permuted Vt [H][128][S_pad] consumed by pe_flash_attn. */
int pe_qkv_rmsnorm_rope(const void* x, int ldx, const void* Wqkv, const void* bqkv, int M, int H, int K,
                        const void* norm_q_w, const void* norm_k_w, const float* rope_cos,
                        const float* rope_sin, void* q_out, void* k_out, void* vt_out, int seq_off, int S_pad,
                        void* stream);

/* softmax(Q K^T * scale) V over the joint sequence, no mask (qwen_image_flash_attention, :14-39).
 * Q,K [H][S_pad][128], Vt [H][128][S_pad] as written by pe_qkv_rmsnorm_rope; out [S][ldo] "s (h d)". */
int pe_flash_attn(const void* q, const void* k, const void* vt, void* out, int H, int S, int S_pad, int ldo,
                  float scale, void* workspace, size_t workspace_bytes, void* stream);
/* The default attention kernel (round 4) takes Q PRE-MULTIPLIED by scale * log2(e): the factor leaves the softmax's instruction
 * stream and is applied where Q is produced, in fp32, before Q's one rounding to bf16 (same number of roundings as the reference's
 * q; qwen_image_dit.py:293-302 then :14-39).  pe_attn_q_prescale(scale) = that factor for the selected kernel (1 when it wants a
 * plain Q); pe_qkv_rmsnorm_rope_scaled = pe_qkv_rmsnorm_rope storing bf16(rope(q) * q_scale); pe_flash_attn_prescaled =
 * pe_flash_attn on such a Q (error when pe_attn_q_prescale(scale) == 1).  pe_dit_forward uses this trio; pe_flash_attn itself
 * keeps taking a plain Q (it runs the same schedule's exact form). */
float pe_attn_q_prescale(float scale);
int pe_qkv_rmsnorm_rope_scaled(const void* x, int ldx, const void* Wqkv, const void* bqkv, int M, int H, int K,
                               const void* norm_q_w, const void* norm_k_w, const float* rope_cos,
                               const float* rope_sin, void* q_out, void* k_out, void* vt_out, int seq_off, int S_pad,
                               float q_scale, void* stream);
int pe_flash_attn_prescaled(const void* q, const void* k, const void* vt, void* out, int H, int S, int S_pad, int ldo,
                            float scale, void* workspace, size_t workspace_bytes, void* stream);
/* The reference's enable_fp8_attention branch as one operator (it takes it only where FlashAttention-3 exists; here it is this kernel):
 * q, k, vt in pe_flash_attn's bf16 layouts with a PLAIN Q.  Computes the three global standard deviations (torch.std semantics, bf16),
 * e4m3(q / q_std), e4m3(k / k_std), e4m3(v / v_std), softmax(q8 k8^T q_std k_std / sqrt(128)) with P cast to e4m3 for the second
 * e4m3 matmul, and bf16(bf16(out) * v_std).  scratch: pe_flash_attn_fp8_scratch_bytes(H, S_pad) bytes, 256-byte aligned; workspace
 * as for pe_flash_attn (nullable).  Rows / positions of tokens >= S (the planes' padding up to S_pad) may hold anything FINITE: the
 * statistics and the softmax skip them (Vt's token order inside a 16-group is undone for that), so one set of planes can serve
 * sequences of different lengths without re-zeroing.  What FlashAttention-3 does INSIDE its kernel is restated from its published
 * design (oracle flash_attention_fp8): parity of the P quantisation is unpinned. */
size_t pe_flash_attn_fp8_scratch_bytes(int H, int S_pad);
int pe_flash_attn_fp8(const void* q, const void* k, const void* vt, void* out, int H, int S, int S_pad, int ldo, void* scratch,
                      size_t scratch_bytes, void* workspace, size_t workspace_bytes, void* stream);
/* Optional scratch for pe_flash_attn (16-B aligned).  With it, the (head, q-block) items that do not fill a
 * whole round of the 256 CUs are split along KV and merged by a second small kernel (load balance); without
 * it (NULL) the launch is a single kernel. */
size_t pe_flash_attn_workspace_bytes(int H, int S);
/* pe_flash_attn under the reference's EliGen attention mask (process_entity_masks, models/qwen_image_dit.py:433-498;
 * scaled_dot_product_attention(attn_mask=) at :37), given as one uint32 per token (see pe_dit_call.attn_words): rows
 * [0, n_img) of the sequence are image tokens, (a, b) attend iff token_words[a] & token_words[b] != 0.  token_words: device,
 * 16-byte aligned, S_pad entries, zero beyond S. */
int pe_flash_attn_masked(const void* q, const void* k, const void* vt, void* out, int H, int S, int S_pad, int ldo, float scale,
                         void* workspace, size_t workspace_bytes, const void* token_words, int n_img, void* stream);

/* LayerNorm(no affine, eps) * (1 + scale) + shift on rows of width 3072; rows [0,rows_a) use
 * (shift_a, scale_a), the remaining rows (shift_b, scale_b)  (qwen_image_dit.py:355-357,372-376). */
int pe_ln_modulate(const void* x, void* out, int rows, int dim, int rows_a, const void* shift_a,
                   const void* scale_a, const void* shift_b, const void* scale_b, float eps, void* stream);

/* RMSNorm with weight over rows of width 3584 (txt_norm; models/utils.py:250-257). */
int pe_rmsnorm(const void* x, const void* w, void* out, int rows, int dim, float eps, void* stream);
/* Small operators of the training-time prior (QwenImageUnit_PhysicalVisualEmbedder, pipelines/
 * qwen_image_physical.py:1071-1118 -> PerceiverResampler, pipelines/helpers.py:8-109); its Linears are pe_gemm_bf16 launches.
 * pe_add_bf16: x = bf16(x + sign * y), sign = +1 / -1 (positional / frame embeddings, residuals, `middle - source`), n % 8 == 0.
 * pe_layernorm_affine: nn.LayerNorm(dim) with weight and bias, fp32 statistics, one rounding; dim % 8 == 0.
 * pe_perceiver_attention: PerceiverAttention's core for heads of 64 (helpers.py:52-62): q [n_queries, heads*64], kv [n_keys,
 *   2*heads*64] (keys | values), out [n_queries, heads*64]; dots, scaled dots, dots - max, softmax and the result are each rounded
 *   to bf16 as the reference's tensor ops round them.  n_keys <= 15360. */
int pe_add_bf16(void* x, const void* y, size_t n, float sign, void* stream);
int pe_layernorm_affine(const void* x, const void* weight, const void* bias, void* out, int rows, int dim, float eps, void* stream);
int pe_perceiver_attention(const void* q, const void* kv, void* out, int n_queries, int n_keys, int heads, float scale, void* stream);
/* Small unmasked attention with heads of 64 and the numerics of torch's scaled_dot_product_attention (DINOv2's self-attention,
 * pipelines/dinov2.py:8-31 -> transformers Dinov2WithRegistersSelfAttention): fp32 scores and softmax statistics, the un-normalised
 * P = exp(s - max) rounded to bf16 for the second product, fp32 accumulation, one rounding of O / sum.  Same operand layout as
 * pe_perceiver_attention: q [n_queries, heads*64], kv [n_keys, 2*heads*64] (keys first), out [n_queries, heads*64]. */
int pe_sdpa_heads64(const void* q, const void* kv, void* out, int n_queries, int n_keys, int heads, float scale, void* stream);

/* nn.Linear applied to ONE row: y[N] = bf16(W[N,K] . x[K] + bias[N]) (fp32 accumulation, one rounding), bias nullable, K % 8 == 0.
 * The HBM-bound shape of autoregressive decoding: used by the prompt prologue (Qwen2.5-VL `generate`,
 * pipelines/qwen_image_physical.py:859-873) in place of the BLAS GEMV behind torch.nn.functional.linear. */
int pe_gemv_bf16(const void* x, const void* W, const void* bias, void* y, int N, int K, void* stream);
/* the same followed by a decoder layer's residual: y[n] = bf16(res[n] + bf16(W[n,:] . x + bias[n])); y may alias res */
int pe_gemv_res_bf16(const void* x, const void* W, const void* bias, const void* res, void* y, int N, int K, void* stream);
/* The gated MLP's first half on one row (transformers Qwen2MLP.forward): y[n] = bf16(silu(bf16(Wg[n,:].x)) * bf16(Wu[n,:].x)),
 * SiLU evaluated in fp32 and rounded once, as torch.nn.SiLU does on a bf16 tensor. */
int pe_gemv_swiglu_bf16(const void* x, const void* Wg, const void* Wu, void* y, int N, int K, void* stream);
/* Decode step of a GQA attention layer with 128-wide heads (transformers Qwen2_5_VLAttention.forward at q_len = 1):
 * pe_decode_qkv_rope: q / k / v = Linear(x) (three weight / bias sets, one launch), then rotary embedding of the q and k heads with
 *   the section-selected tables cos_sel / sin_sel [128] bf16: y = bf16(bf16(t * cos) + bf16(rotate_half(t) * sin)), fused into
 *   the projection kernel (a wave computes rows i and i + 64 of a head together).
 *   q [n_q_heads*128], k, v [n_kv_heads*128] bf16.
 * pe_decode_attention: softmax(q K^T * scale) V of that one query against the cache k_cache, v_cache [n_kv_heads][L][128]
 *   (query head h reads kv head h / (n_q_heads / n_kv_heads)); fp32 scores and sums, P rounded to bf16 before P.V.  L <= 15360. */
int pe_decode_qkv_rope(const void* x, const void* Wq, const void* bq, const void* Wk, const void* bk, const void* Wv, const void* bv,
                       const void* cos_sel, const void* sin_sel, void* q, void* k, void* v, int n_q_heads, int n_kv_heads, int K,
                       void* stream);
int pe_decode_attention(const void* q, const void* k_cache, const void* v_cache, void* out, int n_q_heads, int n_kv_heads, int L,
                        float scale, void* stream);
/* The same decode step in a form a captured graph (hipGraph) can replay: everything that changes from token to token is read from
 * DEVICE memory.  `step` counts the tokens generated so far; the q / k / v launch takes its rotary tables from row *step of
 * cos_table / sin_table [n_steps][128] and writes the new k / v rows straight into the caches [n_kv_heads][cache_len][128] at row
 * base_len + *step (base_len = prompt length); the attention reads base_len + *step + 1 valid rows; pe_decode_embed copies row
 * *token of the embedding table; pe_decode_argmax writes the arg-max of the logits (first index among equal maxima) to *token and
 * out_ids[*step], then increments *step.  A step past the cache capacity writes nothing.  cache_len <= 15360.
 * Used by diffsynth/pipelines/prompt_prologue.py::GraphDecoder for the greedy `generate` of qwen_image_physical.py:859-873. */
int pe_decode_step_qkv(const void* x, const void* Wq, const void* bq, const void* Wk, const void* bk, const void* Wv, const void* bv,
                       const void* cos_table, const void* sin_table, void* q, void* k_cache, void* v_cache, int n_q_heads,
                       int n_kv_heads, int K, const int* step, int base_len, int cache_len, const void* norm_w /* nullable */, float eps,
                       void* stream);
/* Single-row Linears whose input is RMSNorm(x) * norm_w (text width K = 3584; also the `norm_w` of pe_decode_step_qkv): the
 * normalisation happens while x is staged in LDS, bit-identical to pe_rmsnorm followed by the plain form. */
int pe_gemv_norm_bf16(const void* x, const void* norm_w, float eps, const void* W, const void* bias, void* y, int N, int K,
                      void* stream);
int pe_gemv_swiglu_norm_bf16(const void* x, const void* norm_w, float eps, const void* Wg, const void* Wu, void* y, int N, int K,
                             void* stream);
int pe_decode_step_attention(const void* q, const void* k_cache, const void* v_cache, void* out, int n_q_heads, int n_kv_heads,
                             const int* step, int base_len, int cache_len, float scale, void* stream);
/* pe_decode_step_attention as three small launches over 16 x as many work-groups (scores + block maxima, softmax sums + P.V per wave of
 * the one-launch form, combine): every sum keeps its grouping, the output is bit-identical.  workspace: fp32 scratch of
 * pe_decode_attention_workspace_bytes(n_q_heads, cache_len) bytes, 16-byte aligned, private to the stream. */
size_t pe_decode_attention_workspace_bytes(int n_q_heads, int cache_len);
int pe_decode_step_attention_split(const void* q, const void* k_cache, const void* v_cache, void* out, int n_q_heads, int n_kv_heads,
                                   const int* step, int base_len, int cache_len, float scale, void* workspace, size_t workspace_bytes,
                                   void* stream);
/* ONE decoder layer of the decode step in one launch (round 6; transformers Qwen2_5_VLDecoderLayer.forward at q_len = 1): input norm +
 * q / k / v + rotary + cache append, the single-query attention, o_proj + residual, post-attention norm + gate / up + SiLU, down_proj +
 * residual -- the work items and summation orders of pe_decode_step_qkv, pe_decode_step_attention_split, pe_gemv_res_bf16 and
 * pe_gemv_swiglu_norm_bf16, run by a persistent grid with grid-wide barriers between them: the same bits per output element, one launch
 * instead of eight.  Text width 3584 (28 heads of 128), n_q_heads a multiple of n_kv_heads, ff <= 32768.
 *   scratch: pe_decode_layer_scratch_bytes(n_q_heads, cache_len, ff) bytes, 256-byte aligned, ZERO when first used (the barrier counters
 *   at its start are reset by the launches themselves), private to one stream.  Word 0 of the scratch is an error flag: non-zero after a
 *   launch whose grid barrier timed out (the outputs are then meaningless); the caller reads it with the tokens. */
typedef struct pe_decode_layer_weights {
    const void *q_w, *q_b, *k_w, *k_b, *v_w, *v_b, *o_w, *gate_w, *up_w, *down_w, *input_norm_w, *post_norm_w;
    float input_norm_eps, post_norm_eps;
    int n_q_heads, n_kv_heads, ff;
} pe_decode_layer_weights;
size_t pe_decode_layer_scratch_bytes(int n_q_heads, int cache_len, int ff);
int pe_decode_layer(const pe_decode_layer_weights* w, const void* x, void* x_out, const void* cos_table, const void* sin_table, void* k_cache,
                    void* v_cache, const int* step, int base_len, int cache_len, float scale, void* scratch, size_t scratch_bytes, void* stream);
int pe_decode_embed(const void* table, const int* token, void* x, int dim, int vocab, void* stream);
int pe_decode_argmax(const void* logits, int vocab, int* token, int* out_ids, int* step, int max_steps, void* stream);
/* BlockWiseControlBlock input (models/qwen_image_controlnet.py:16-18): out = bf16(RMSNorm(x; wx) + RMSNorm(y; wy)), rows of
 * dim = 3072, each RMSNorm with the roundings of models/utils.py:250-257. */
int pe_dual_rmsnorm_add(const void* x, const void* wx, const void* y, const void* wy, void* out, int rows, int dim, float eps,
                        void* stream);

/* "C (H 2) (W 2) -> (H W) (C 2 2)" and back (qwen_image_physical.py:1344,1402); latents [C,H2,W2]. */
int pe_patchify(const void* latents, void* tokens, int C, int H2, int W2, void* stream);
int pe_unpatchify(const void* tokens, void* latents, int C, int H2, int W2, void* stream);

/* latents_out = latents + (nega + cfg*(posi-nega)) * dsigma   (qwen_image_physical.py:656 +
 * schedulers/flow_match.py:81); use_cfg = 0 => noise_pred = posi (cfg_scale == 1.0 branch, :654). */
int pe_cfg_euler_step(const void* posi, const void* nega, const void* latents, void* latents_out, size_t n,
                      float cfg_scale, int use_cfg, float dsigma, void* stream);
/* The same step of an inpainting run (BasePipeline.step with inpaint_mask, utils/__init__.py:146-156): between the CFG combination
 * and the Euler update, noise_pred = expected * (1 - mask) + noise_pred * mask with expected = (latents - input_latents) / sigma
 * (schedulers/flow_match.py:85-91, sigma = sigmas[progress_id] > 0), every operation rounded to bf16 as the reference's tensor
 * ops are.  inpaint_mask: [plane] bf16 (one H/8 x W/8 plane, broadcast over the n / plane channels). */
int pe_cfg_inpaint_euler_step(const void* posi, const void* nega, const void* latents, const void* input_latents,
                              const void* inpaint_mask, void* latents_out, size_t n, size_t plane, float cfg_scale, int use_cfg,
                              float sigma, float dsigma, void* stream);

/* ---------------------------------------------------------------------------------------------
 * DiT composite
 * ------------------------------------------------------------------------------------------- */
typedef struct pe_dit_block_weights {
    /* image stream */
    const void *img_mod_w, *img_mod_b;           /* img_mod.1            [18432,3072],[18432] */
    const void *img_qkv_w, *img_qkv_b;           /* to_q|to_k|to_v       [9216,3072],[9216]   */
    const void *norm_q_w, *norm_k_w;             /* attn.norm_q/norm_k   [128]                */
    const void *img_out_w, *img_out_b;           /* attn.to_out.0        [3072,3072],[3072]   */
    const void *img_mlp_up_w, *img_mlp_up_b;     /* img_mlp.net.0.proj   [12288,3072]         */
    const void *img_mlp_down_w, *img_mlp_down_b; /* img_mlp.net.2        [3072,12288]         */
    /* text stream */
    const void *txt_mod_w, *txt_mod_b;           /* txt_mod.1 */
    const void *txt_qkv_w, *txt_qkv_b;           /* add_q_proj|add_k_proj|add_v_proj */
    const void *norm_added_q_w, *norm_added_k_w;
    const void *txt_out_w, *txt_out_b;           /* attn.to_add_out */
    const void *txt_mlp_up_w, *txt_mlp_up_b;
    const void *txt_mlp_down_w, *txt_mlp_down_b;
} pe_dit_block_weights;

typedef struct pe_dit_weights {
    int num_layers;
    const void *time_w1, *time_b1;     /* time_text_embed.timestep_embedder.linear_1 [3072,256] */
    const void *time_w2, *time_b2;     /* ...linear_2 [3072,3072] */
    const void* txt_norm_w;            /* [3584] */
    const void *img_in_w, *img_in_b;   /* [3072,64] */
    const void *txt_in_w, *txt_in_b;   /* [3072,3584] */
    const void *norm_out_w, *norm_out_b; /* norm_out.linear [6144,3072] */
    const void *proj_out_w, *proj_out_b; /* [64,3072] */
    const pe_dit_block_weights* blocks;  /* host array [num_layers]; copied by pe_dit_create */
    /* 0: every weight is bf16.  1: "FP8 computation" (enable_vram_management(enable_dit_fp8_computation=True) on a DiT
     * stored in float8_e4m3fn, qwen_image_physical.py:440-496): every *_w of a Linear ([N,K], K padded to a multiple
     * of 128 with zeros -- only img_in, [3072,128]) is OCP e4m3fn bytes and runs as fp8_linear
     * (vram_management/layers.py:115-151); biases and RMSNorm weights stay bf16 pointers holding bf16(e4m3(value)).
     * The adapter (a separate module in the reference) and hot LoRA operands are always bf16. */
    int weights_e4m3;
} pe_dit_weights;

typedef struct pe_adapter_weights {    /* VisualThinkingDualAdapter, pipelines/helpers.py:123-140 */
    const void *dino_w0, *dino_b0, *dino_w2, *dino_b2; /* [10752,3584],[10752],[3584,10752],[3584] */
    const void *vae_w0, *vae_b0, *vae_w2, *vae_b2;
} pe_adapter_weights;

/* One BlockWiseControlBlock (models/qwen_image_controlnet.py:6-27): rms(x) + rms(y) -> Linear -> exact-erf GELU -> Linear. */
typedef struct pe_controlnet_block {
    const void *x_rms_w, *y_rms_w;             /* [3072] bf16 */
    const void *in_w, *in_b, *out_w, *out_b;   /* [3072,3072], [3072] bf16 */
} pe_controlnet_block;

/* One ControlNetInput that is ACTIVE for this call (the progress gate of QwenImageBlockwiseMultiControlNet.blockwise_forward,
 * pipelines/qwen_image_physical.py:172-180, is host logic).  After every transformer block l the first S0 (noise) rows of the
 * image stream become  image + sum_i bf16(controlnet_i.blocks[l](image, conditioning_i) * scale_i)  (:1389-1396). */
typedef struct pe_control_input {
    const pe_controlnet_block* blocks;  /* HOST array [num_layers of the DiT] of device pointers; read during the call only */
    const void* conditioning;           /* [S0,3072] bf16 = controlnet.img_in(patchify(conditioning latents)), :164-170 */
    float scale;
} pe_control_input;

/* geometry + per-call inputs of one model_fn_qwen_image call */
typedef struct pe_dit_call {
    const void* latents;        /* [16,h8,w8] bf16 (B = 1) */
    int h8, w8;
    int n_edit;                 /* 0..4 edit/context latents */
    const void* edit_latents[4];
    int edit_h8[4], edit_w8[4];
    void* prompt_emb;           /* [T,3584] bf16, special rows MUTATED IN PLACE (:1336) */
    int T;
    const int* special_idx;     /* device int32[n_special]: rows of prompt_emb under special_token_mask */
    int n_special;              /* 0 => no adapter (QwenImagePipeline behaviour) */
    float alpha, one_minus_alpha; /* adapter mix weights as bf16-rounded floats (helpers.py:142-150,158) */
    const float *rope_cos_img, *rope_sin_img; /* [S_img,64] fp32, QwenEmbedRope vid_freqs */
    const float *rope_cos_txt, *rope_sin_txt; /* [T,64]     fp32, QwenEmbedRope txt_freqs */
    int step;                   /* row of the tables built by pe_dit_prepare */
    void* noise_pred;           /* out: [16,h8,w8] bf16 */
    int n_control;              /* 0..4 active block-wise ControlNet inputs (bf16 weights only) */
    pe_control_input control[4];
    /* EliGen entity control (QwenImageDiT.process_entity_masks, models/qwen_image_dit.py:433-498), or NULL.  The caller passes the
     * prompts CONCATENATED in prompt_emb / rope_*_txt (entity prompts ..., global prompt; T = the total; special_idx rows offset
     * accordingly) and one uint32 per token of the library's joint order [image rows | text rows], S_pad(64) entries, zero beyond
     * S: image token = bit 31 | bit i for every prompt i whose region contains it (the global prompt: all of them);
     * token of prompt i = bit i.  Attention between tokens a, b is allowed iff words[a] & words[b] != 0 -- the reference's
     * additive 0 / -inf mask.  Default attention kernel only. */
    const unsigned int* attn_words;
    /* != 0: qwen_image_flash_attention(enable_fp8_attention=True) (models/qwen_image_dit.py:24-35): q, k, v / their global std -> e4m3,
     * both attention matmuls on e4m3 operands, softmax_scale = q_std k_std / sqrt(128), output x v_std (pe_flash_attn_fp8).  Like the
     * reference, only without a mask: ignored when attn_words is set. */
    int fp8_attention;
} pe_dit_call;

/* Runtime ("hot") LoRA operands of one block -- load_lora(hotload=True), qwen_image_physical.py:264-272 +
 * vram_management/layers.py:173-181: out = linear(x) + (x @ A.T) @ B.T, each op rounded.  r = rank padded to a
 * multiple of 64 with zeros.  The three attention projections of a stream share one fused pair:
 * qkv_a = [Aq; Ak; Av] ([3r, 3072]) and qkv_b = blockdiag(Bq, Bk, Bv) ([9216, 3r]).  NULL group = no LoRA there. */
typedef struct pe_dit_block_lora {
    const void *img_qkv_a, *img_qkv_b, *img_out_a, *img_out_b, *img_down_a, *img_down_b, *img_mod_a, *img_mod_b;
    const void *txt_qkv_a, *txt_qkv_b, *txt_out_a, *txt_out_b, *txt_down_a, *txt_down_b, *txt_mod_a, *txt_mod_b;
} pe_dit_block_lora;

typedef struct pe_dit* pe_dit_handle;

int pe_dit_create(const pe_dit_weights* w, const pe_adapter_weights* adapter /* nullable */, pe_dit_handle* out);
void pe_dit_destroy(pe_dit_handle h);

/* Hot LoRA sets.  pe_dit_set_hot_lora replaces all sets by one (blocks != NULL: host array [num_layers], copied) or clears
 * them (NULL); pe_dit_add_hot_lora appends another set -- load_lora(hotload=True) called again: AutoWrappedLinear keeps lists
 * of pairs and adds them in load order (vram_management/layers.py:173-181).  r <= 128, at most 8 sets.  Call before
 * pe_dit_prepare: the modulation rows depend on them. */
int pe_dit_set_hot_lora(pe_dit_handle h, const pe_dit_block_lora* blocks, int r);
int pe_dit_add_hot_lora(pe_dit_handle h, const pe_dit_block_lora* blocks, int r);

/* Bytes of workspace needed for sequences up to (S_img_max image tokens, T_max text tokens) and
 * n_steps prepared timesteps.  The stream-K scratch (64 MiB) is part of it only while the opt-in schedule is switched on
 * ("gemm_sk" != 0 or "gemm_variant" 19): pe_dit_workspace_bytes and pe_dit_bind_workspace must see the SAME knob state, and
 * a handle bound without it never takes schedule 19 whatever the knob says later. */
size_t pe_dit_workspace_bytes(pe_dit_handle h, int S_img_max, int T_max, int n_steps);

/* Bind a workspace (zero-fills the regions that must start finite).  Must precede prepare/forward. */
int pe_dit_bind_workspace(pe_dit_handle h, void* workspace, size_t bytes, int S_img_max, int T_max, int n_steps,
                          void* stream);

/* Hoisted timestep work, once per image: sinusoid [n_steps,256] bf16 (TemporalTimesteps output cast
 * to bf16, models/utils.py:291) -> time MLP -> temb[n_steps,3072]; then for every block and both
 * streams mod = Linear(SiLU(temb)) (qwen_image_dit.py:369-370) and norm_out.linear(SiLU(temb))
 * (models/utils.py:305).  The same rows serve the posi and the nega forward of a step. */
int pe_dit_prepare(pe_dit_handle h, const void* sinusoid_bf16, int n_steps, void* stream);

/* One model_fn_qwen_image call (is_train=False): adapter on the special tokens (in place),
 * patchify, embeds, all blocks, AdaLN head, unpatchify. */
int pe_dit_forward(pe_dit_handle h, const pe_dit_call* call, void* stream);

/* Debug/test taps: device pointers into the bound workspace (valid after pe_dit_forward). */
const void* pe_dit_debug_ptr(pe_dit_handle h, const char* name);
/* Training-loss head (model_fn's `is_train=True` branch, qwen_image_physical.py:1337-1338 -> VisualThinkingDualAdapter.get_loss,
 * pipelines/helpers.py:166-183): the two per-head mean squared errors between the adapter predictions of the LAST pe_dit_forward on
 * this handle (its n_special rows x 3584) and the targets gt_dino / gt_vae [n_special, 3584] bf16, with the reference's roundings
 * ((pred - gt) -> bf16, square -> bf16, fp32 mean).  out2: two floats on the device.  The time-dependent weighting of the two
 * (scalar arithmetic) is host code: physicedit_amd/dit.py::special_token_loss. */
int pe_dit_special_token_mse(pe_dit_handle h, const void* gt_dino, const void* gt_vae, int n_special, float* out2, void* stream);

/* ---------------------------------------------------------------------------------------------
 * VAE operators (QwenImageVAE.encode/decode, models/qwen_image_vae.py:706-729).  Activations are
 * NHWC bf16, B = 1, channel count padded to a multiple of 32 with zeros.
 * ------------------------------------------------------------------------------------------- */

/* 2-D conv = QwenImageCausalConv3d at T=1 (only temporal tap 2 contributes, :40-50) / nn.Conv2d.
 * w [Cout_p][ksize*ksize][Cin_p] (tap-major, channel-minor), bias [Cout_p], res nullable
 * [Hout*Wout][Cout_p] added after the conv's own rounding (ResidualBlock `x + h`, :152).
 * ksize 3: stride 1 pads 1 on every side; stride 2 is ZeroPad2d((0,1,0,1)) + stride-2 conv (:249);
 * upsample2x reads the input through nearest-exact 2x upsampling (:240) without materialising it.
 * zero_page: >= 64 B of device zeros (halo reads).  ksize 1: plain 1x1 conv. */
int pe_conv2d_nhwc(const void* in, const void* w, const void* bias, const void* res, void* out, const void* zero_page,
                   int Hin, int Win, int Cin_p, int Cout_p, int ksize, int stride, int upsample2x, void* stream);

/* QwenImageRMS_norm (:76-77) over the C valid channels of each pixel, optional fused SiLU. */
int pe_vae_rmsnorm(const void* x, const void* gamma, void* out, int npix, int C, int Cp, int silu, void* stream);

/* layout converters.  mode 0 copy; mode 1 (decode, :723-724) y = x / tb[c] + ta[c];
 * mode 2 (encode, :713-714) y = (x - ta[c]) * tb[c]; ta = mean, tb = 1/std, bf16 [C]. */
int pe_nchw_to_nhwc(const void* in, void* out, int C, int HW, int Cp, int mode, const void* ta, const void* tb,
                    void* stream);
int pe_nhwc_to_nchw(const void* in, void* out, int C, int HW, int Cp, int mode, const void* ta, const void* tb,
                    void* stream);

/* single-head attention, D = 384 (QwenImageAttentionBlock, :173-198): qkv [N][1152] (q|k|v), out [N][384];
 * vt_scratch: pe_vae_attention_scratch_bytes(N) bytes, 256-byte aligned (the transposed V, and -- round 6, ABI 8 -- the fp32 partials
 * of the key split that fills the CUs when there are fewer than 256 query blocks). */
size_t pe_vae_attention_scratch_bytes(int N);
int pe_vae_attention(const void* qkv, void* vt_scratch, void* out, int N, void* stream);

/* ---------------------------------------------------------------------------------------------
 * VAE composites: QwenImageVAE.encode / .decode (models/qwen_image_vae.py:706-717 / :719-729) as ONE call each, B = 1,
 * single frame.  Weights are a table of device pointers in the repacked layout of pe_conv2d_nhwc (conv weights
 * [Cout_p][k*k][Cin_p] with only the last temporal tap of the causal conv3d, channels zero-padded to 32; RMS-norm gammas
 * [C]); the library copies the table, not the weights.  Image I/O of the pipeline is fused into the first / last kernel:
 *   PE_IMAGE_BF16_NCHW  [3,H,W] bf16 in [-1,1]            (what vae.encode / vae.decode take / return in the reference)
 *   PE_IMAGE_U8_HWC     [H,W,3] uint8: encode applies BasePipeline.preprocess_image (pipelines/utils/__init__.py:60-66,
 *                       bf16(bf16(u8) * 2/255) - 1), decode applies vae_output_to_image (:76-83, ((x+1)*127.5).clip(0,255)
 *                       truncated to uint8), both with the reference's bf16 rounding points.
 * ------------------------------------------------------------------------------------------- */
enum { PE_IMAGE_BF16_NCHW = 0, PE_IMAGE_U8_HWC = 1 };

typedef struct pe_vae_conv {       /* QwenImageCausalConv3d at T = 1 / nn.Conv2d */
    const void *w, *b;             /* [cout_p][ksize*ksize][cin_p], [cout_p] bf16; w == NULL: layer absent */
    int cin_p, cout_p, ksize;      /* ksize 1 or 3 */
} pe_vae_conv;
typedef struct pe_vae_res {        /* QwenImageResidualBlock (:81-152) */
    pe_vae_conv conv1, conv2, shortcut;   /* shortcut.w == NULL when in_dim == out_dim (nn.Identity) */
    const void *norm1_g, *norm2_g;        /* QwenImageRMS_norm gamma */
} pe_vae_res;
typedef struct pe_vae_attn {       /* QwenImageAttentionBlock (:156-198), dim 384 */
    const void* norm_g;
    pe_vae_conv to_qkv, proj;
} pe_vae_attn;
typedef struct pe_vae_mid { pe_vae_res res0; pe_vae_attn attn; pe_vae_res res1; } pe_vae_mid;   /* QwenImageMidBlock (:304-340) */
typedef struct pe_vae_weights {
    /* encoder (:344-448): conv_in, 4 stages x 2 residual blocks with a stride-2 resample after stages 0..2, mid, head */
    pe_vae_conv enc_conv_in;
    pe_vae_res enc_res[8];
    pe_vae_conv enc_down[3];       /* down_blocks.{2,5,8}.resample.1 */
    pe_vae_mid enc_mid;
    const void* enc_norm_out_g;
    pe_vae_conv enc_conv_out, quant_conv;
    /* decoder (:522-636): post_quant_conv, conv_in, mid, 4 stages x 3 residual blocks with a 2x upsample+conv after 0..2 */
    pe_vae_conv post_quant_conv, dec_conv_in;
    pe_vae_mid dec_mid;
    pe_vae_res dec_res[12];
    pe_vae_conv dec_up[3];         /* up_blocks.{0,1,2}.upsamplers.0.resample.1 */
    const void* dec_norm_out_g;
    pe_vae_conv dec_conv_out;
    const void *mean, *inv_std;    /* bf16 [16]: latents mean and 1/std (:667-704) */
    const void* zero_page;         /* >= 64 B of device zeros */
} pe_vae_weights;
typedef struct pe_vae* pe_vae_handle;

int pe_vae_create(const pe_vae_weights* w, pe_vae_handle* out);
void pe_vae_destroy(pe_vae_handle h);
/* Scratch for an H x W image (encode) / an (H/8) x (W/8) latent (decode): four activation slots + attention scratch. */
size_t pe_vae_workspace_bytes(int H, int W);
/* image ([3,H,W] bf16 or [H,W,3] uint8) -> normalised latents [16,H/8,W/8] bf16. */
int pe_vae_encode(pe_vae_handle h, const void* image, int input_format, int H, int W, void* latents, void* workspace,
                  size_t workspace_bytes, void* stream);
/* normalised latents [16,H8,W8] bf16 -> image ([3,8*H8,8*W8] bf16 or [8*H8,8*W8,3] uint8). */
int pe_vae_decode(pe_vae_handle h, const void* latents, int H8, int W8, void* image, int output_format, void* workspace,
                  size_t workspace_bytes, void* stream);

/* VisualThinkingDualAdapter.forward (pipelines/helpers.py:152-164) on n rows of width 3584:
 * out = bf16(alpha * head_dino(x)) + bf16((1-alpha) * head_vae(x)), heads = Linear 3584->10752, exact-erf GELU, Linear
 * 10752->3584.  alpha / one_minus_alpha are the bf16-rounded mix weights of _get_alpha (:142-150). */
size_t pe_adapter_workspace_bytes(int n);
int pe_adapter_forward(const pe_adapter_weights* adapter, const void* x, int n, float alpha, float one_minus_alpha, void* out,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Measurement aid: a loop of bf16 MFMAs only (no LDS, no memory traffic) over the fragments at `frags` (16 MiB of bf16, caller
 * filled: zeros -> the nominal dense peak; random values -> the rate the chip's POWER LIMIT allows on such operands, the ceiling of
 * every bf16 MFMA kernel on that data).  out: blocks * 512 floats.  *flops receives the FLOPs of the launch; time it with events. */
int pe_mfma_probe(const void* frags, void* out, int blocks, int iters, double* flops, void* stream);
/* Measurement aid: the bf16 GEMM main loop's per-MFMA resource mix, free-running (no barriers, no epilogue, no wait for arriving
 * data): 512-thread work-groups with 160 KiB of LDS, per K tile and wave 32 MFMAs plus -- mode 1 -- the 24 ds_read_b128 fragment
 * reads of the 64 x 128 wave tile and -- mode 2 -- also the 8 LDS-DMA pieces of the operand stream, read from `src` (src_bytes =
 * 8 windows of a power-of-two size >= 1 MiB, one per XCD, so the stream is L2 resident).  Mode 0 = MFMAs only in the same
 * kernel.  On random data the three rates bracket what any schedule of this tiling can reach under the power limit
 * (bench.py `roofline.attainable_ceiling`).  out: blocks * 512 floats; *flops receives the launch's FLOPs. */
int pe_gemm_mix_probe(int mode, const void* src, size_t src_bytes, void* out, int blocks, int iters, double* flops, void* stream);
/* The attention twin: the default flash-attention kernel's schedule WITHOUT its softmax -- per KV tile and wave the 64 MFMAs, their 32
 * LDS fragment reads, 8 LDS-DMA pieces and the barrier -- on the caller's Q / K / Vt (pe_flash_attn layouts; N(0,1) data for a
 * meaningful rate).  *flops = 4 S^2 128 H, the launch's nominal work; `out` receives garbage.  What it sustains is the ceiling of the
 * tiling and staging; bench.py reports flash attention's rate as a fraction of it (other_kernels.flash_attn). */
int pe_attn_mix_probe(const void* q, const void* k, const void* vt, void* out, int H, int S, int S_pad, int ldo, double* flops,
                      void* stream);

/* ---------------------------------------------------------------------------------------------
 * Measurement: HIP-event timing of sampled launches, recorded on the launch stream.
 * kind: 0 = MFMA GEMM (work = algorithmic FLOPs 2MNK), 1 = flash attention (4*S*S*128*H FLOPs),
 *       2 = row kernels (work = algorithmic bytes), 3 = VAE convolutions (FLOPs).
 * ------------------------------------------------------------------------------------------- */
int pe_profile_enable(int max_events, int sample_every);
void pe_profile_disable(void);
int pe_profile_read(int kind, long long* launches_seen, long long* sampled, double* total_ms, double* total_work);

#ifdef __cplusplus
}
#endif
#endif /* PHYSICEDIT_AMD_H */
