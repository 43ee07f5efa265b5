/*
 * glm4moe_hip.h — C-ABI of libglm4moe_hip.so: the MI355X (gfx950) kernels behind the
 * GLM-4-MoE per-layer forward path of nano-vllm.
 *
 * The reference has no FFI: its operator surface is Python nn.Modules.  Each
 * entry point below replaces the device work of one reference operator; the Python modules in
 * glm4_moe_nano-vllm_amd/nanovllm/ keep the reference's names and call these through ctypes
 * (INTEGRATION.md shows the binding a maintainer would add to the reference).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *     (PyTorch storage).  The library never allocates, frees, synchronises or reads back:
 *     every call only enqueues kernels on `stream` (a hipStream_t), so a sequence of calls
 *     can be captured in a hipGraph.  (One exception, host-only and outside any captured
 *     sequence: the glm_comm_* helpers that allocate the one-shot all-reduce's peer-written
 *     communication memory.)
 *   - bf16 tensors are passed as `const void*` / `void*` (2-byte elements, row-major,
 *     innermost stride 1).  "ld*" / "*_stride" arguments are in ELEMENTS.
 *   - return value: GLM_OK (0) or a negative GLM_ERR_* code; glm_last_error() returns a
 *     thread-local message for the last failing call.  No exceptions cross the ABI.
 *   - thread safety: calls are re-entrant; one host thread per stream is the intended use
 *     (one process per GPU, reference: nanovllm/engine/llm_engine.py:23-30).
 */
#ifndef GLM4MOE_HIP_H
#define GLM4MOE_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* glm_stream_t; /* hipStream_t */

#define GLM_OK               0
#define GLM_ERR_INVALID     (-1) /* bad pointer / size / alignment */
#define GLM_ERR_UNSUPPORTED (-2) /* shape outside what the gfx950 kernels implement */
#define GLM_ERR_LAUNCH      (-3) /* hipGetLastError() != hipSuccess after the launch */
#define GLM_ERR_WORKSPACE   (-4) /* workspace too small */

/* ABI version (bumped when a signature changes) and last error text. */
int         glm_abi_version(void);
const char* glm_last_error(void);

/* Launch-path options (host-only; r04).  Every switch a launch consults — which body a one-row GEMM takes, the loop
 * variant of the tiled GEMM, the prefill-attention kernel ... — lives in one table that is filled from the environment
 * ONCE (at the first use) and changed only through glm_set_option: a launch never reads the environment, so a HIP graph and
 * the eager launches around it see one setting.  Names: "tinym" (GLM4MOE_TINYM), "fp8w_native" (GLM4MOE_FP8W_NATIVE),
 * "largem_variant" (GLM_LARGEM_VARIANT), "largem_cfg" (GLM_LARGEM_CFG: 0 by shape, 1 = A, 2 = B), "fa_mfma32"
 * (GLM_FA_MFMA32), "fa_rows64" (GLM_FA_ROWS64: prefill attention with 64 query rows per wave).  glm_option_name(i) enumerates them (NULL past the end).  The reference has no switches of its own on
 * this path (its Config dataclass, config.py:6-26, holds engine sizes only).                                          */
int         glm_get_option(const char* name, int* value);
int         glm_set_option(const char* name, int value);
const char* glm_option_name(int index);

/* ---------------------------------------------------------------------------------------
 * a2  layers/layernorm.py:17-26  RMSNorm.rms_forward
 *     y = bf16( bf16(x32 * rsqrt(mean(x32^2) + eps)) * w )         x,y: [T,H] bf16, w: [H] bf16
 * ------------------------------------------------------------------------------------- */
int glm_rmsnorm(const void* x, const void* weight, void* y,
                int64_t T, int H, float eps, glm_stream_t stream);

/* a3  layers/layernorm.py:29-40  RMSNorm.add_rms_forward
 *     s = f32(x) + f32(residual_in); residual_out = bf16(s); y = bf16(bf16(s*rstd(s)) * w)
 *     residual_out may alias residual_in or x; y may alias x.                              */
int glm_add_rmsnorm(const void* x, const void* residual_in, const void* weight,
                    void* y, void* residual_out,
                    int64_t T, int H, float eps, glm_stream_t stream);

/* a4  layers/RMSNorm.py:19-33  per-head RMSNorm (weight multiply in fp32, one rounding)
 *     x: [T, n_heads, hd] with token stride x_token_stride (head stride = hd); y likewise.  */
int glm_head_rmsnorm(const void* x, int64_t x_token_stride, const void* weight,
                     void* y, int64_t y_token_stride,
                     int64_t T, int n_heads, int hd, float eps, glm_stream_t stream);

/* a6  layers/rotary_embedding.py:6-14,39-55  NeoX rotary on the first rotary_dim dims of each
 *     head, pass-through for the rest.  cos_sin_cache: f32 [max_pos, rotary_dim] = cos|sin.
 *     q: [T,nh,hd] (token stride q_stride), k: [T,nkv,hd]; outputs contiguous-or-strided
 *     (q_out may alias q, k_out may alias k).  positions: int64 [T].                        */
int glm_rope_neox(const void* q, int64_t q_stride, const void* k, int64_t k_stride,
                  void* q_out, int64_t q_out_stride, void* k_out, int64_t k_out_stride,
                  const int64_t* positions, const float* cos_sin_cache, int64_t max_pos,
                  int64_t T, int nh, int nkv, int hd, int rotary_dim, glm_stream_t stream);

/* a7  layers/attention.py:10-42  store_kvcache: cache.view(-1,D)[slot] = k[i].view(D); slot -1
 *     skipped.  k,v: [N, D] rows with token strides; caches: [n_slots, D] contiguous.       */
int glm_kv_store(const void* k, int64_t k_stride, const void* v, int64_t v_stride,
                 void* k_cache, void* v_cache, const int32_t* slot_mapping,
                 int64_t N, int D, int64_t n_slots, glm_stream_t stream);

/* a4+a6+a7 fused (models/glm4_moe/attention_new.py:329-357): from the packed projection
 *     qkv [T, (nh+2nkv)*hd] (row stride qkv_stride) produce q_out [T,nh,hd] (contiguous) with
 *     optional per-head RMSNorm + rotary, k_out/v_out [T,nkv,hd] (contiguous, may be NULL), and
 *     scatter the rotated K and V rows into the paged cache (k_cache NULL or slot_mapping NULL
 *     = no store).  q_norm_w / k_norm_w NULL = use_qk_norm off.                             */
int glm_qkv_split_norm_rope_store(const void* qkv, int64_t qkv_stride,
                                  void* q_out, void* k_out, void* v_out,
                                  const void* q_norm_w, const void* k_norm_w, float eps,
                                  const int64_t* positions, const float* cos_sin_cache,
                                  int64_t max_pos, int rotary_dim,
                                  void* k_cache, void* v_cache, const int32_t* slot_mapping,
                                  int64_t n_slots,
                                  int64_t T, int nh, int nkv, int hd, glm_stream_t stream);

/* a5 + a6 + a7 in ONE launch at prefill row counts (r04): the QKV projection (linear.py:74-75, attention_new.py:329) whose
 *     epilogue splits the packed row, rotates q and k (rotary_embedding.py:39-55, partial NeoX, rotary_dim 64 of head_dim
 *     128) and writes q [T, nh, 128], k / v [T, nkv, 128] (optional) and the K/V cache rows (attention.py:10-42; slot -1 or
 *     >= n_slots skipped) — glm_gemm_bf16 followed by glm_qkv_split_norm_rope_store without the packed [T, (nh+2nkv)*128]
 *     row ever reaching memory, bit-identical to the two launches.  No per-head RMSNorm (use_qk_norm models take the two
 *     launches).  GLM_ERR_UNSUPPORTED (nothing enqueued): head_dim != 128, rotary_dim != 64, fewer than 96 rows, K % 64 != 0,
 *     operands beyond 2 GiB, unaligned pointers.                                                                      */
int glm_gemm_qkv_rope_store(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, int64_t T,
                            int nh, int nkv, int hd, int K, const int64_t* positions, const float* cos_sin_cache,
                            int rotary_dim, void* q_out, void* k_out, void* v_out, void* k_cache, void* v_cache,
                            const int32_t* slot_mapping, int64_t n_slots, glm_stream_t stream);

/* layers/activation.py:12-14  SiluAndMul: y[t,i] = bf16( bf16(silu(x[t,i])) * x[t,I+i] )    */
int glm_silu_and_mul(const void* x, void* y, int64_t T, int I, glm_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a8  layers/attention.py:101-107  flash_attn_with_kvcache decode: one query token per
 *     sequence over a paged KV cache.
 *     q: [B,nh,hd] contiguous; caches: [n_blocks, block_size, nkv, hd]; context_lens int32[B]
 *     (0 ⇒ zero output row); block_tables int32 [B, bt_stride] (-1 padded); out [B,nh,hd].
 *     workspace: glm_paged_attn_decode_workspace_bytes() bytes (may be NULL when that is 0), 16-byte aligned.  Its
 *     first 64 KiB are the arrival counters of the split merge (done by the last split to finish, in the same
 *     launch): they must be ZERO before the first launch on a workspace; the kernel puts the zeros back.
 *     num_splits: 0 = choose; ≥1 = forced number of KV splits per (sequence, kv-head).
 *     glm_paged_attn_decode_splits() = the split count a launch with these arguments uses (1 ⇒ no workspace is
 *     read); size the workspace with THAT count — workspace_bytes(…, 0) is the worst case (64 splits).        */
size_t glm_paged_attn_decode_workspace_bytes(int64_t B, int nh, int hd, int num_splits);
int glm_paged_attn_decode_splits(int64_t B, int nkv, int block_size, int64_t bt_stride, int num_splits,
                                 int max_context_len);
int glm_paged_attn_decode(const void* q, const void* k_cache, const void* v_cache,
                          const int32_t* context_lens, const int32_t* block_tables,
                          int64_t bt_stride, void* out,
                          int64_t B, int nh, int nkv, int hd, int block_size, int64_t n_blocks,
                          float scale, int num_splits, int max_context_len,
                          void* workspace, size_t workspace_bytes, glm_stream_t stream);

/* a9  layers/attention.py:85-100  flash_attn_varlen_func(causal=True[, block_table]):
 *     q: [Tq,nh,hd]; if block_tables == NULL k,v: [Tk,nkv,hd] contiguous, else k,v are the
 *     paged caches.  cu_seqlens_q/k int32 [B+1].  Causal mask is bottom-right aligned
 *     (query i of a sequence sees keys 0..(lk-lq)+i).                                        */
int glm_varlen_attn_prefill(const void* q, const void* k, const void* v,
                            const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                            const int32_t* block_tables, int64_t bt_stride,
                            void* out, int64_t B, int max_seqlen_q,
                            int nh, int nkv, int hd, int block_size, float scale,
                            glm_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a11 models/glm4_moe/moe.py:28-50  GlmMoeSelectTopk: logits=bf16(x·Wgᵀ); s=bf16(sigmoid);
 *     choice=f32(s)+bias; top-k (ties → lowest expert id); w=s[idx]; w/=bf16(sum w)+1e-20;
 *     w*=routed_scaling_factor.  Outputs: topk_idx int64 [T,k] (ascending expert id per row),
 *     topk_w bf16 [T,k] (same order); logits_out bf16 [T,E] optional (NULL = not written).   */
int glm_router_topk(const void* x, const void* gate_w, const float* e_score_correction_bias,
                    int64_t* topk_idx, void* topk_w, void* logits_out,
                    int64_t T, int H, int E, int k, float routed_scaling_factor,
                    glm_stream_t stream);

/* a11 at prefill sizes: R2..R7 from logits that a GEMM produced (R1 = glm_gemm_bf16(x, W_gate): [T, E] bf16, row stride
 *     ld_logits) — the same selection code as glm_router_topk, one wave per token.  idx ascending per row.          */
int glm_router_select(const void* logits, int64_t ld_logits, const float* e_score_correction_bias,
                      int64_t* topk_idx, void* topk_w, int64_t T, int E, int k,
                      float routed_scaling_factor, glm_stream_t stream);

/* Token→expert bucketing for the grouped GEMM (replaces the mask/where/any loop of
 *     moe.py:135-147).  Rows are grouped by expert id ascending; if with_shared != 0 one more
 *     group (id E) holding every token once is appended (the shared expert, moe.py:165).
 *     G = E + (with_shared?1:0), R = T*k + (with_shared?T:0).
 *     expert_offsets int32 [G+1]; sorted_token int32 [R]; sorted_scale bf16 [R] (routing
 *     weight of the row; 1.0 for shared rows); pair_pos int32 [T*k] (sorted row of pair).    */
int glm_moe_align(const int64_t* topk_idx, const void* topk_w,
                  int32_t* expert_offsets, int32_t* sorted_token, void* sorted_scale,
                  int32_t* pair_pos, int64_t T, int k, int E, int with_shared,
                  glm_stream_t stream);

/* The same bucketing over many workgroups, for prefill sizes (the kernel above is ONE workgroup: 200 us at T = 16384,
 *     k = 8): count / offsets / scatter launches over chunks of 8192 pairs.  workspace: glm_moe_align_workspace_bytes(),
 *     int32; its first E words must be ZERO on entry (the caller clears them for every call).  Same contract for the
 *     outputs; the order of the rows inside a group is unspecified in both.                                  */
size_t glm_moe_align_workspace_bytes(int64_t T, int k, int E);
int glm_moe_align_parallel(const int64_t* topk_idx, const void* topk_w, int32_t* expert_offsets,
                           int32_t* sorted_token, void* sorted_scale, int32_t* pair_pos, int64_t T, int k, int E,
                           int with_shared, int32_t* workspace, size_t workspace_bytes, glm_stream_t stream);

/* a13 (first half) models/glm4_moe/mlp.py:49-53 for every group at once:
 *     act[r, i] = bf16( bf16(silu(bf16(x[tok(r)]·Wg_gᵀ))) * bf16(x[tok(r)]·Wu_gᵀ) )
 *     w13: [G, 2I, H] (rows gate|up, mlp.py:118) with element strides (w13_group_stride,
 *     w13_row_stride, 1): a padded row pitch (e.g. H+64) keeps the 16 rows a wave streams off
 *     the same HBM channels (measured +19 %% at H=4096).  expert_offsets NULL ⇒ one group of R rows;
 *     sorted_token NULL ⇒ tok(r) = r.  x: [T,H] (row stride ldx); act: [R,I] contiguous.    */
int glm_grouped_gemm_gateup_silu(const void* x, int64_t ldx, const void* w13,
                                 int64_t w13_group_stride, int64_t w13_row_stride,
                                 const int32_t* expert_offsets, const int32_t* sorted_token,
                                 void* act, int64_t R, int H, int I, int G,
                                 glm_stream_t stream);

/* a13 (second half) + the per-row scale of moe.py:149:
 *     y[r, :] = bf16( bf16(act[r]·Wd_gᵀ) * scale[r] ) for groups g < n_scaled_groups,
 *     y[r, :] = bf16(act[r]·Wd_gᵀ) otherwise (shared expert / dense MLP).
 *     w2: [G, H, I] with element strides (w2_group_stride, w2_row_stride, 1); y: [R, H]
 *     (row stride ldy).                                            */
int glm_grouped_gemm_down(const void* act, const void* w2,
                          int64_t w2_group_stride, int64_t w2_row_stride,
                          const int32_t* expert_offsets, const void* sorted_scale,
                          int n_scaled_groups, void* y, int64_t ldy,
                          int64_t R, int H, int I, int G, glm_stream_t stream);

/* a10 + a3 + a11 + bucketing for ONE token (attention_new.py o_proj, decode_layer.py:86-93, moe.py:28-50), r03:
 *     o = bf16(x·Woᵀ)  (x [1, K], Wo [H, K], no bias);  then, inside the same launch (the output row reaches the workgroups
 *     that continue through write-through stores and a completion counter):
 *     s = f32(o) + f32(residual_in); residual_out = bf16(s); y_norm = bf16(bf16(s*rstd)*norm_weight);
 *     (topk_idx, topk_w) = router R1..R7 on y_norm;  expert_offsets / sorted_token / sorted_scale / pair_pos as
 *     glm_moe_align(T = 1).  Bit-identical to glm_gemm_bf16 followed by glm_add_rmsnorm_router_topk_align.
 *     o_row: [H] bf16 scratch (the projection's output row).  ticket: 16 + 4 E bytes of scratch, 16-byte aligned; its first
 *     two uint32 are counters that must be ZERO before the first launch (the launch leaves them zero), E floats of logits
 *     follow at byte 16.  The tail runs in ceil(E / 16) designated workgroups that requested their 16 gate rows into LDS at
 *     kernel start and wait (bounded) for the projection's completion counter.  A wait that expires (it cannot while the
 *     GPU makes progress) poisons every output (y_norm, topk_w: NaN; expert_offsets: no rows; pair_pos: -1) and stores
 *     1 + the workgroup's number into the THIRD uint32 of `ticket` (sticky status; the host must then re-zero the scratch).
 *     GLM_ERR_UNSUPPORTED (nothing enqueued): H > 4096, H < 2048, E > 256, K outside the window the tail's LDS scratch
 *     fits (at H = 4096: 9472 <= K <= 14336), fewer than 2 ceil(E/16) row tiles, unaligned buffers.                  */
int glm_gemm_bf16_router_tail(const void* x, const void* w, int64_t ldw, int K, int H, void* o_row, uint32_t* ticket,
                              const void* residual_in, const void* norm_weight, float eps, void* y_norm,
                              void* residual_out, const void* gate_w, const float* e_score_correction_bias, int E, int k,
                              float routed_scaling_factor, int64_t* topk_idx, void* topk_w, int32_t* expert_offsets,
                              int32_t* sorted_token, void* sorted_scale, int32_t* pair_pos, int with_shared,
                              glm_stream_t stream);

/* N4 on the fp8 MFMA (r05; BASELINE configs[4] "fp8 weight-quantised experts on CDNA4 fp8 MFMA"; OPT-IN, W8A8 — the
 *     weight-only entry points above stay the parity path).  No reference counterpart: the reference's quantised linears are
 *     int4 AWQ dequant-then-GEMM (layers/linear_awq_new.py:254-255); the expert arithmetic replaced is mlp.py:49-54 / moe.py:149.
 *     Both operands OCP e4m3 on v_mfma_scale_f32_16x16x128_f8f6f4, fp32 accumulation:
 *     glm_quantize_fp8_rows          x [T, H] bf16 -> x_q [T, H] e4m3 + one fp32 scale per row (amax / 448);
 *     glm_grouped_gemm_gateup_silu_fp8a8   acc[r, n] = sum_k w13_q[g, n, k] x_q[tok(r), k] (block scales 2^0);
 *         gate / up = acc * w13_scale[g, n] * x_scale[tok(r)]; act = bf16(bf16(silu(bf16 gate)) * bf16 up) as a13, then quantised
 *         in MX form by the epilogue: per (row, 32 columns) an E8M0 scale 2^(floor(log2 amax) - 8), one step up where amax would
 *         pass 448 (nothing saturates), and e4m3 = RNE(v / scale).
 *         act_q [R, I] bytes (row stride ld_act); act_bscale [R, 64] bytes, block b at (b & 3) * 16 + (b >> 2) of the row's record.
 *     glm_grouped_gemm_down_fp8a8    y[r, :] = bf16(bf16(sum_k w2_q[g, n, k] act_q[r, k] 2^(bscale[r, k / 32] - 127) * w2_scale[g, n])
 *         * scale[r]) for groups < n_scaled_groups, unscaled otherwise — the block scales go into the MFMA itself.
 *     Strides of the fp8 tensors in BYTES (= elements).  Grouped only, 256-row blocks: GLM_ERR_UNSUPPORTED (nothing enqueued)
 *     when R / G < 192, H or I not multiples of 128, I > 2048 — the caller then takes the weight-only path.            */
int glm_quantize_fp8_rows(const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int64_t T, int H,
                          glm_stream_t stream);
int glm_grouped_gemm_gateup_silu_fp8a8(const void* x_q, int64_t ldx, const float* x_scale, const void* w13_q,
                                       int64_t w13_group_stride, int64_t w13_row_stride, const float* w13_scale,
                                       int64_t w13_scale_group_stride, const int32_t* expert_offsets,
                                       const int32_t* sorted_token, void* act_q, int64_t ld_act, void* act_bscale,
                                       int64_t R, int H, int I, int G, glm_stream_t stream);
int glm_grouped_gemm_down_fp8a8(const void* act_q, int64_t ld_act, const void* act_bscale, const void* w2_q,
                                int64_t w2_group_stride, int64_t w2_row_stride, const float* w2_scale,
                                int64_t w2_scale_group_stride, const int32_t* expert_offsets, const void* sorted_scale,
                                int n_scaled_groups, void* y, int64_t ldy, int64_t R, int H, int I, int G,
                                glm_stream_t stream);

/* a10 + a3 + a11 + bucketing + a12/a13 for ONE token in ONE persistent launch (r05; BASELINE configs[1], greedy bs = 1):
 *     what glm_gemm_bf16_router_tail, glm_grouped_gemm_gateup_silu and glm_grouped_gemm_down do for one token row —
 *     attention_new.py o_proj, decode_layer.py:86-93, moe.py:28-50, the expert loop moe.py:135-151 with mlp.py:49-54 —
 *     and what the reference replays as one CUDA graph (engine/model_runner.py:223-258).  Every output is bit-identical
 *     to those three launches.  One 512-thread workgroup per CU streams o_proj -> shared + routed gate|up -> down with the
 *     next weight unit always requested before the current one's activations are waited for; the o_proj row, the E logits
 *     and the act rows travel between workgroups as 8-byte {epoch, value} granules (csrc/decode_engine.hip).
 *     x [K] attention output row; w_o [H, K]; w13 [E+1, 2I, H] / w2 [E+1, H, I] stacked experts, group E = the shared
 *     expert (required); act [k+1, I] optional (NULL = not stored); y [k+1, H]: row j = the j-th selected expert in
 *     ascending id (scaled by its routing weight), row k = the shared expert — what glm_moe_combine / the next layer's
 *     glm_gemm_bf16_combine_norm read through pair_pos with shared_base = k.
 *     workspace: glm_decode_mlp_engine_workspace_bytes(H, E, k, I) bytes, 16-byte aligned, ZERO before the first launch and
 *     never touched by anything else: word 0 = the launch counter (epoch; tags are never reset, so no memset node), word 1 =
 *     sticky status — 0, or a code | workgroup when a bounded wait expired (it cannot while the GPU makes progress): y is
 *     then NaN-filled by that workgroup and the host must raise (hip_ops.check_router_tail) and re-zero the workspace.
 *     GLM_ERR_UNSUPPORTED (nothing enqueued): H outside [2048, 4096], I > 1536, E > 256 or E > CUs, k > 64, K / H / I not
 *     multiples of 8, unaligned buffers.                                                                              */
size_t glm_decode_mlp_engine_workspace_bytes(int H, int E, int k, int I);
int glm_decode_mlp_engine(const void* x, const void* w_o, int64_t ldw_o, int K, int H, const void* residual_in,
                          const void* norm_weight, float eps, void* y_norm, void* residual_out, const void* gate_w,
                          const float* e_score_correction_bias, int E, int k, float routed_scaling_factor,
                          int64_t* topk_idx, void* topk_w, int32_t* expert_offsets, int32_t* sorted_token,
                          void* sorted_scale, int32_t* pair_pos, const void* w13, int64_t w13_group_stride,
                          int64_t w13_row_stride, const void* w2, int64_t w2_group_stride, int64_t w2_row_stride, int I,
                          void* act, void* y, int64_t ldy, void* workspace, glm_stream_t stream);

/* N4 (no reference counterpart; BASELINE config 5): the same two grouped projections with the
 *     expert weights stored as OCP fp8 e4m3 (1 byte/element) plus one fp32 scale per weight row:
 *     W'[n,k] = scale[n] * fp8(Wq[n,k]).  Weight-only quantisation: activations stay bf16, the
 *     fp8 values are converted exactly to bf16 in registers, accumulated in fp32 on the bf16 MFMA,
 *     and the row scale is applied to the fp32 accumulator before the bf16 roundings of a13.
 *     w13_q: [G, 2I, H] uint8 with element strides (group, row, 1); w13_scale: f32 [G, 2I]
 *     (row stride 1, group stride w13_scale_group_stride); likewise w2_q [G, H, I], w2_scale [G, H].
 *     Few rows per group (decode): the weights are streamed as fp8 (HBM-bound, half the bytes).  Many rows per group
 *     (prefill, MFMA-bound): the tiled MFMA kernel stages the fp8 bytes themselves (LDS-DMA, half the weight tile's LDS
 *     bytes) and converts a fragment to bf16 between its LDS read and the MFMA — no extra pass, no scratch.  Only a K that
 *     is not a multiple of 64 needs expand_scratch (glm_fp8w_expand_scratch_bytes(G, weight rows, K) bytes — 0 when the
 *     stack is covered — 16-byte aligned, contents scratch): the stack is then expanded exactly to bf16 by one streaming
 *     pass and the bf16 kernel runs on the copy; both forms are bit-identical.  With NULL where scratch would be needed the
 *     streaming kernel is used.                                                                                       */
size_t glm_fp8w_expand_scratch_bytes(int G, int rows, int cols);
int glm_grouped_gemm_gateup_silu_fp8w(const void* x, int64_t ldx, const void* w13_q,
                                      int64_t w13_group_stride, int64_t w13_row_stride,
                                      const float* w13_scale, int64_t w13_scale_group_stride,
                                      const int32_t* expert_offsets, const int32_t* sorted_token,
                                      void* act, int64_t R, int H, int I, int G,
                                      void* expand_scratch, size_t expand_scratch_bytes, glm_stream_t stream);
int glm_grouped_gemm_down_fp8w(const void* act, const void* w2_q,
                               int64_t w2_group_stride, int64_t w2_row_stride,
                               const float* w2_scale, int64_t w2_scale_group_stride,
                               const int32_t* expert_offsets, const void* sorted_scale,
                               int n_scaled_groups, void* y, int64_t ldy,
                               int64_t R, int H, int I, int G,
                               void* expand_scratch, size_t expand_scratch_bytes, glm_stream_t stream);

/* a12 models/glm4_moe/moe.py:133,151,171: out[t] = bf16-accumulate of the token's k routed
 *     rows in ascending expert order (one bf16 rounding per add, starting from 0), then
 *     + shared row (bf16 add) when shared_base >= 0 (row index shared_base + t of y).       */
int glm_moe_combine(const void* y, int64_t ldy, const int32_t* pair_pos, void* out,
                    int64_t T, int k, int H, int64_t shared_base, glm_stream_t stream);

/* a5/a10 layers/linear.py:53,75,152  y = bf16(x·Wᵀ + b): x [M,K] (ldx), W [N,K] (row pitch ldw),
 *     bias [N] bf16 or NULL, y [M,N] (ldy).  Picks the kernel from M: the GEMV body
 *     of single-sequence decode (M <= 2), the weight-streaming small-M kernel, or the LDS-tiled MFMA kernel.                                                         */
int glm_gemm_bf16(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias,
                  void* y, int64_t ldy, int64_t M, int N, int K, glm_stream_t stream);

/* Single-sequence decode (BASELINE config 2: greedy bs = 1; T = 1 token row), r03.
 * a13 at T = 1: the same operators as glm_grouped_gemm_gateup_silu / glm_grouped_gemm_down (same arguments, same
 *     rounding points) on the GEMV body — one persistent workgroup per CU, fp32 FMA, weights streamed 16-32 KiB deep per
 *     wave.  T = number of token rows of x (the GEMV body takes T = 1, more rows run the entry points above); PRECONDITION: no group holds more than T rows (true for a top-k routing: a
 *     token's experts are distinct, the shared group holds T rows).  Shapes the body does not cover fall through to the
 *     entry points above.                                                                                             */
int glm_grouped_gemm_gateup_silu_fewtok(const void* x, int64_t ldx, int64_t T, const void* w13,
                                        int64_t w13_group_stride, int64_t w13_row_stride,
                                        const int32_t* expert_offsets, const int32_t* sorted_token,
                                        void* act, int64_t R, int H, int I, int G, glm_stream_t stream);
int glm_grouped_gemm_down_fewtok(const void* act, int64_t T, const void* w2, int64_t w2_group_stride,
                                 int64_t w2_row_stride, const int32_t* expert_offsets, const void* sorted_scale,
                                 int n_scaled_groups, void* y, int64_t ldy, int64_t R, int H, int I, int G,
                                 glm_stream_t stream);
/* a3 + a5 of layer i+1 with a12 of layer i, T = 1 (decode_layer.py:47-49 + attention_new.py qkv_proj + moe.py:133-171):
 *     c = glm_moe_combine(y_prev, pair_pos, shared_base);  s = f32(c) + f32(residual_in);  residual_out = bf16(s);
 *     xn = bf16(bf16(s * rstd(s)) * norm_weight);  y = bf16(xn·Wᵀ + bias)
 *     in ONE launch: every workgroup of the GEMV recomputes the (tiny) combine + norm behind its first weight loads,
 *     workgroup 0 writes residual_out (and x_out = xn when not NULL).  Bit-identical to
 *     glm_moe_combine_add_rmsnorm followed by glm_gemm_bf16.  GLM_ERR_UNSUPPORTED when T > 1 or H > 4096.             */
int glm_gemm_bf16_combine_norm(const void* y_prev, int64_t ld_yprev, const int32_t* pair_pos, int64_t T, int k,
                               int64_t shared_base, const void* residual_in, const void* norm_weight, float eps,
                               void* residual_out, void* x_out, const void* w, int64_t ldw, const void* bias,
                               void* y, int64_t ldy, int N, int H, glm_stream_t stream);

/* a10+a3 fused pair for decode batches (M <= 32), no reference counterpart as an API: the row-parallel o_proj
 *     (layers/linear.py:152, K = nh*hd >> N) is computed as `nslices` K-slice partial sums by separate workgroups,
 *     ypart [nslices][M][N] fp32 (no bias), and the consumer — the post-attention add-RMSNorm, decode_layer.py:51 —
 *     sums them in slice order, rounds to bf16 (the linear's output rounding) and continues as glm_add_rmsnorm.
 *     glm_gemm_bf16_partial_slices(K, want) returns the slice count to use for a requested one (whole 128-wide
 *     K chunks, no empty slice).                                                              */
int glm_gemm_bf16_partial_slices(int K, int want);
int glm_gemm_bf16_partial(const void* x, int64_t ldx, const void* w, int64_t ldw, float* ypart,
                          int64_t M, int N, int K, int nslices, glm_stream_t stream);
int glm_add_rmsnorm_partial(const float* part, int nslices, int64_t slice_stride,
                            const void* residual_in, const void* weight, void* y, void* residual_out,
                            int64_t T, int H, float eps, glm_stream_t stream);

/* a3+a11 fused: the post-attention add-RMSNorm (layers/layernorm.py:29-40, decode_layer.py:51) and the router of the
 *     same token (models/glm4_moe/moe.py:28-50) in one launch.  x is a bf16 [T,H] tensor, or NULL with `part` =
 *     glm_gemm_bf16_partial output (then as glm_add_rmsnorm_partial).  y [T,H] receives the normalised rows (the
 *     experts' input), residual_out the updated residual; topk_idx/topk_w as glm_router_topk.  H <= 8192.        */
int glm_add_rmsnorm_router_topk(const void* x, const float* part, int nslices, int64_t slice_stride,
                                const void* residual_in, const void* norm_weight, float eps, void* y,
                                void* residual_out, const void* gate_w, const float* e_score_correction_bias,
                                int64_t* topk_idx, void* topk_w, int64_t T, int H, int E, int k,
                                float routed_scaling_factor, glm_stream_t stream);
/*     … and glm_moe_align in the same launch: the workgroup that finishes last (agent-scope release / ticket /
 *     acquire, no co-residency or placement assumption) buckets all T*k pairs.  expert_offsets … pair_pos as
 *     glm_moe_align.  With E >= 32 the experts of a token are also split over ceil(E / 16) workgroups (a CU pulls
 *     1/4 of the gate matrix at E = 64 instead of all of it); the last of them to finish gathers the token's logits
 *     and selects.  workspace: glm_router_workspace_bytes(T, E) bytes, 16-byte aligned; its first 262208 bytes
 *     (arrival counters) must be ZERO before the first launch on it — the kernel puts the zeros back; launches
 *     sharing a workspace must be stream-ordered.  Needs 8 (E + 1) <= 2 H bytes of LDS reuse.                    */
size_t glm_router_workspace_bytes(int64_t T, int E);
int glm_add_rmsnorm_router_topk_align(const void* x, const float* part, int nslices, int64_t slice_stride,
                                      const void* residual_in, const void* norm_weight, float eps, void* y,
                                      void* residual_out, const void* gate_w, const float* e_score_correction_bias,
                                      int64_t* topk_idx, void* topk_w, int64_t T, int H, int E, int k,
                                      float routed_scaling_factor, int32_t* expert_offsets, int32_t* sorted_token,
                                      void* sorted_scale, int32_t* pair_pos, int with_shared, uint32_t* workspace,
                                      size_t workspace_bytes, glm_stream_t stream);

/* a12+a3 fused: glm_moe_combine of layer i and the input add-RMSNorm of layer i+1 (decode_layer.py:47-49) in one
 *     launch at TP=1: out = normalised rows, residual_out = bf16(combine + residual_in).  H <= 16384.            */
int glm_moe_combine_add_rmsnorm(const void* y, int64_t ldy, const int32_t* pair_pos, int64_t T, int k, int H,
                                int64_t shared_base, const void* residual_in, const void* weight,
                                void* out, void* residual_out, float eps, glm_stream_t stream);

/* a6+a7+a8 fused for decode (no head RMSNorm, rotary_dim 64): glm_paged_attn_decode that starts from the packed QKV
 *     projection output [B, (nh+2nkv)*hd] — rotates Q in registers, and the workgroup that owns position len-1 rotates
 *     the new K row and stores K/V at slot_mapping[b] (layers/attention.py:10-42) before it reads them back.
 *     Same results as glm_qkv_split_norm_rope_store followed by glm_paged_attn_decode.                       */
int glm_paged_attn_decode_qkv(const void* qkv, int64_t qkv_stride, const int64_t* positions,
                              const float* cos_sin_cache, int64_t n_pos, int rotary_dim, const int32_t* slot_mapping,
                              void* k_cache, void* v_cache, const int32_t* context_lens,
                              const int32_t* block_tables, int64_t bt_stride, void* out, int64_t B,
                              int nh, int nkv, int hd, int block_size, int64_t n_blocks, float scale,
                              int num_splits, int max_context_len, void* workspace,
                              size_t workspace_bytes, glm_stream_t stream);

/* The same launch fed by the QKV projection's fp32 K-slice partial sums (glm_gemm_bf16_partial, [nslices][B][(nh+2nkv)*hd],
 *     slices slice_stride floats apart) and the projection's bias: a Q/K/V element is formed on load as
 *     bf16(slice 0 + slice 1 + ... + bias), the linear layer's own output rounding (linear.py:74-75).  The QKV GEMM then
 *     runs as a K-split over more workgroups (decode: 27 -> 22 us at M=32) and its reduction costs no launch.        */
int glm_paged_attn_decode_qkv_partial(const float* qkv_part, int nslices, int64_t slice_stride, const void* qkv_bias,
                                      const int64_t* positions, const float* cos_sin_cache, int64_t n_pos,
                                      int rotary_dim, const int32_t* slot_mapping, void* k_cache, void* v_cache,
                                      const int32_t* context_lens, const int32_t* block_tables, int64_t bt_stride,
                                      void* out, int64_t B, int nh, int nkv, int hd, int block_size, int64_t n_blocks,
                                      float scale, int num_splits, int max_context_len, void* workspace,
                                      size_t workspace_bytes, glm_stream_t stream);

/* N3  expert-parallel dispatch / combine (BASELINE config 4; no reference counterpart: parallel_state_ref.py:792-810
 *     only has dead dispatch/combine stubs).  Row pack / unpack around the all-to-all: gather dst[i] = src[idx[i]],
 *     scatter dst[idx[i]] = src[i] (rows of `cols` bf16; idx < 0 skips).  glm_moe_combine_scaled = glm_moe_combine on
 *     the UNSCALED expert rows that came back from their owners: out[t] = sum over the token's rows in ascending
 *     expert id of bf16(y[row] * row_scale[row]) (one bf16 rounding per add, moe.py:149-151), then + shared_rows[t]
 *     (moe.py:171) when given.                                                                                      */
int glm_gather_rows(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, const int32_t* idx, int64_t n, int cols,
                    glm_stream_t stream);
int glm_scatter_rows(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, const int32_t* idx, int64_t n, int cols,
                     glm_stream_t stream);
int glm_moe_combine_scaled(const void* y, int64_t ldy, const int32_t* pair_pos, const void* row_scale,
                           const void* shared_rows, int64_t ld_shared, void* out, int64_t T, int k, int H,
                           glm_stream_t stream);

/* N3  expert-parallel dispatch plan with FIXED-capacity slots (graph-capturable; BASELINE config 4): from the replicated
 *     routing topk_idx / topk_w [T, k] (identical on every rank) compute, on the device, everything rank `rank` of `ep`
 *     needs for the all-to-all with EQUAL splits — no count exchange, no host read.  Rank r owns experts
 *     [r E/ep, (r+1) E/ep) and the token slice [r per, (r+1) per), per = ceil(T/ep).  Send buffer = [ep][cap] rows,
 *     destination d's rows at the head of slot d (expert ascending, then pair index); receive buffer likewise by source.
 *       send_gather [ep*cap]  row of the full hidden-state matrix for each send row, -1 = unused (glm_gather_rows skips)
 *       slot_scale  [ep*cap]  bf16 routing weight of the pair in that row       pair_slot [Tl*k]  row of pair (t-lo, j)
 *       recv_offs   [E/ep+1]  expert_offsets over the received rows             recv_gidx [ep*cap] receive-buffer row of
 *                             expert-ordered row i, -1 beyond recv_offs[E/ep]
 *       overflow    [1]       0, or the largest (source, destination) row count above cap (cap = Tl * min(k, E/ep)
 *                             cannot overflow).  Reference intent: the dead dispatch/combine stubs,
 *                             distributed/parallel_state_ref.py:792-810.                                             */
int glm_ep_plan(const int64_t* topk_idx, const void* topk_w, int64_t T, int k, int E, int ep, int rank, int cap,
                int32_t* send_gather, void* slot_scale, int32_t* pair_slot, int32_t* recv_offs, int32_t* recv_gidx,
                int32_t* overflow, glm_stream_t stream);
/*     the grouped expert GEMMs over such capacity-sized row arrays: R_cap sizes the grid (expert_offsets decide which rows
 *     are real), R_expected picks the kernel configuration the real row count would get                              */
int glm_grouped_gemm_gateup_silu_cap(const void* x, int64_t ldx, const void* w13, int64_t w13_group_stride,
                                     int64_t w13_row_stride, const int32_t* expert_offsets, const int32_t* sorted_token,
                                     void* act, int64_t R_cap, int64_t R_expected, int H, int I, int G, glm_stream_t stream);
int glm_grouped_gemm_down_cap(const void* act, const void* w2, int64_t w2_group_stride, int64_t w2_row_stride,
                              const int32_t* expert_offsets, const void* sorted_scale, int n_scaled_groups, void* y,
                              int64_t ldy, int64_t R_cap, int64_t R_expected, int H, int I, int G, glm_stream_t stream);

/* N3  one-shot all-reduce(sum) of a small bf16 tensor over directly mapped peer buffers (the RowParallel all-reduce,
 *     layers/linear.py:153-154, as ONE hop over all xGMI links instead of a ring), PUSH form.  Every rank owns a receive
 *     buffer of glm_allreduce_oneshot_buffer_elems(world, n_max) = 2 * world * n_max bf16 ([2 slots][sender][n_max]) and
 *     8 uint32 flags, both mapped into every peer (hipIpc) and ZERO before the first call; peer_bufs / peer_flags are HOST
 *     arrays of `world` device pointers as seen by THIS process (entry `rank` = own).  state: 4 uint32 in this rank's
 *     device memory, zero before the first call: [0] epoch (completed calls — advanced by the kernel, so calls captured in
 *     a HIP graph stay correct under replay), [1]/[2] internal counters, [3] status (0 = ok; q+1 = a call gave up waiting
 *     for rank q: that call's output is NaN).  Every rank must issue the same sequence of calls.  The sum is formed in
 *     ascending rank order in fp32 and rounded once: bit-identical on every rank.  Spins are bounded.  out may alias in.
 *     NOT the default path: RCCL is; this entry has run with world = 1, with two ranks on two streams of one device and
 *     with two processes on one device (eager and under graph replay) only.                                          */
size_t glm_allreduce_oneshot_buffer_elems(int world, int64_t n_max);
/*     Communication memory for the receive buffer and the flags (r04) — the ONE exception to "the library never
 *     allocates": memory that peer GPUs write over xGMI and the owner polls / reads inside a running kernel must not be
 *     ordinary (coarse-grained, L2-cacheable) hipMalloc storage.  Host-only helpers, they synchronise and are not
 *     capturable: glm_comm_alloc = hipExtMallocWithFlags(uncached | fine-grained) + zero fill; the export / open pair
 *     maps it into the peers' processes (hipIpc; handle = GLM_COMM_IPC_HANDLE_BYTES opaque bytes, picklable) the way the
 *     reference's ranks share their POSIX shm segment (engine/model_runner.py:51-54 — control plane there, data here).
 *     glm_comm_free PARKS the block for reuse by a later glm_comm_alloc of the same device, kind and size; it does not
 *     hand it back to the runtime (r04: recycled uncached pages behind PyTorch's allocator gave stale reads of ordinary
 *     tensors later in the process, profiles/r04/c1_stale_readback.txt).  Freeing a foreign pointer is GLM_ERR_INVALID. */
#define GLM_COMM_UNCACHED           0
#define GLM_COMM_FINEGRAINED        1
#define GLM_COMM_IPC_HANDLE_BYTES  64
int glm_comm_alloc(void** ptr, size_t bytes, int kind);
int glm_comm_free(void* ptr);
int glm_comm_ipc_export(const void* ptr, void* handle);
int glm_comm_ipc_open(const void* handle, void** ptr);
int glm_comm_ipc_close(void* ptr);
int glm_allreduce_oneshot(const void* in, void* out, void* const* peer_bufs, void* const* peer_flags, int rank, int world,
                          int64_t n, int64_t n_max, void* state, glm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GLM4MOE_HIP_H */
