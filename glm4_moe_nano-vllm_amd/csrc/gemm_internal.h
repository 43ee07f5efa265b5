// gemm_internal.h — shared parameter block + internal launchers of the two GEMM regimes.
#pragma once
#include "common.h"

enum GemmMode : int {
    GEMM_PLAIN = 0,   // y[r,n] = bf16(acc + bias[n])                              (a5, a10)
    GEMM_GATEUP = 1,  // act[r,i] = bf16(bf16(silu(bf16(g))) * bf16(u)), W rows gate|up  (a13)
    GEMM_DOWN = 2,    // y[r,n] = bf16(bf16(acc) * scale[r]) for groups < n_scaled    (a12/a13)
};

struct GemmParams {
    const bf16_t* x;      // activations, row r_src at x + r_src*ldx, K contiguous
    int64_t ldx;
    const bf16_t* w;      // group g at w + g*w_gstride; row n at + n*ldw, K contiguous
    int64_t w_gstride;
    int64_t ldw;
    const float* wscale;  // fp8 weights only: per-row dequantisation scale, group g at + g*wscale_gstride
    int64_t wscale_gstride;
    int w_fp8;            // gemm_largem only: w points at OCP fp8 e4m3 BYTES (strides in bytes), converted on the fragment read
    // gemm_largem, r05 (glm_launch_gemm_largem_a8): BOTH operands fp8 e4m3 on the block-scaled MFMA.  x, w: bytes, ldx / ldw /
    // w_gstride in bytes; wscale: per weight row.  GATEUP: xscale = per SOURCE token row fp32 scale (block scales 1); the
    // epilogue quantises act per (row, 32 columns) — y = fp8 bytes [R, I] (ldy in bytes), ybscale = E8M0 bytes [R, 64] at
    // (block & 3) * 16 + (block >> 2).  DOWN: xbscale = that table for the act rows, y = bf16 as ever.
    int a8;
    const float* xscale;
    const unsigned char* xbscale;
    unsigned char* ybscale;
    const int32_t* offs;  // [G+1] row offsets per group, or nullptr (one group of R rows)
    const int32_t* src;   // source row of sorted row r, or nullptr (identity)
    const bf16_t* scale;  // [R] per-row scale (GEMM_DOWN) or nullptr
    const bf16_t* bias;   // [N] (GEMM_PLAIN) or nullptr
    bf16_t* y;            // [R, N] rows at y + r*ldy
    int64_t ldy;
    int64_t R;            // total rows (all groups); with offs: an UPPER BOUND is enough (it sizes the grid, offs decide)
    int64_t R_hint;       // 0, or the row count to pick the kernel configuration by when R is a loose bound (fixed-capacity
                          // expert-parallel buffers: R = capacity, R_hint = the expected rows)
    int N;                // output columns (GATEUP: I = act columns; W then has 2I rows)
    int K;
    int G;                // number of groups
    int n_scaled;         // GEMM_DOWN: groups [0, n_scaled) multiply by scale
    int ksplit;           // GEMM_PLAIN small-M only: K slices computed by separate workgroups (0/1 = off)
    float* ypart;         // ksplit > 1: fp32 partial sums [ksplit][R][N] instead of y
    // gemm_largem, GEMM_PLAIN only (r04): the QKV projection's epilogue does what qkv_post_kernel does — split into q / k /
    // v, partial NeoX rotary on q and k (attention_new.py:329-357), K/V rows into the paged cache (attention.py:10-42) —
    // instead of storing the packed row.  qkv_q != nullptr switches it on; y is then unused.  head_dim 128, rotary_dim 64.
    bf16_t* qkv_q;        // [R, nh, 128]
    bf16_t* qkv_k;        // [R, nkv, 128] or nullptr
    bf16_t* qkv_v;
    bf16_t* qkv_kc;       // paged caches viewed as [n_slots, nkv, 128], or nullptr
    bf16_t* qkv_vc;
    const int32_t* qkv_slots;    // [R], -1 = no cache row
    const int64_t* qkv_pos;      // [R]
    const float* qkv_cs;         // cos_sin_cache [max_pos, 64] = cos(32) | sin(32)
    int64_t qkv_nslots;
    int qkv_nh, qkv_nkv;
};

// weight-streaming kernels for few rows per group (decode); returns GLM_* code
int glm_launch_gemm_smallm(int mode, const GemmParams& p, hipStream_t st);
// K-split partial sums (p.ksplit > 1, p.ypart): one fp32 [R,N] slab per K slice, no bias, no rounding
int glm_launch_gemm_smallm_partial(const GemmParams& p, hipStream_t st);
int glm_smallm_partial_slices(int K, int want);  // slices actually used for a requested count (whole 128-wide chunks)
// single-sequence decode (gemm_tinym.hip): ONE token row per weight matrix / per expert group, fp32 FMA GEMV with
// a persistent workgroup per CU.  n_tokens: rows of x (grouped gate|up).  GLM_ERR_UNSUPPORTED = shape not covered (the
// caller then takes the small-M kernels).
int glm_tinym_max_rows();
int glm_launch_gemm_tinym(int mode, const GemmParams& p, int n_tokens, hipStream_t st);
// ... with the activation rows COMPUTED in the kernel: moe_combine(y_prev) + add-RMSNorm (the previous MoE block's combine
// and this layer's input norm), p.x ignored; residual_out / x_out (optional) written by one workgroup
int glm_launch_gemm_tinym_combine_norm(const GemmParams& p, const void* y_prev, int64_t ld_yprev, const int32_t* pair_pos,
                                       int k, int64_t shared_base, const void* residual_in, const void* norm_w,
                                       void* residual_out, void* x_out, float eps, hipStream_t st);
// ... and, for ONE row, continued by the post-attention add-RMSNorm + router + bucketing in the last workgroup to finish
int glm_launch_gemm_tinym_router_tail(const GemmParams& p, unsigned* ticket, float* logits_scratch, const void* residual_in, const void* norm_w,
                                      float eps, void* y_norm, void* residual_out, const void* gate_w, const float* bias, int E,
                                      int k, float rsf, int64_t* topk_idx, void* topk_w, int32_t* offs, int32_t* tok,
                                      void* scale, int32_t* pos, int with_shared, hipStream_t st);
// same, weights stored as OCP fp8 e4m3 (1 byte/element, ldw/w_gstride in elements) + per-row scale
int glm_launch_gemm_smallm_fp8w(int mode, const GemmParams& p, hipStream_t st);
// LDS-tiled MFMA kernels for many rows per group (prefill); p.wscale != nullptr: per-weight-row fp32 scale applied to
// the accumulator before the bf16 roundings (weights expanded from fp8, elementwise.hip)
int glm_launch_gemm_largem(int mode, const GemmParams& p, hipStream_t st);
// ... fp8 x fp8 (p.a8): grouped GATEUP / DOWN on 256-row blocks only; GLM_ERR_UNSUPPORTED otherwise (nothing enqueued)
int glm_launch_gemm_largem_a8(int mode, const GemmParams& p, hipStream_t st);
// ... GEMM_PLAIN with the QKV epilogue (p.qkv_*): GLM_ERR_UNSUPPORTED when the shape is outside the product loop's cover
int glm_launch_gemm_largem_qkv(const GemmParams& p, hipStream_t st);
// exact fp8 e4m3 -> bf16 expansion of a weight stack [G, rows, cols] (element strides)
int glm_launch_fp8_expand(const void* q, int64_t ld_q, int64_t gs_q, void* out, int64_t ld_out, int64_t gs_out, int G,
                          int rows, int cols, hipStream_t st);
// row pitch (elements) of the expanded copy: padded like hip_ops.padded_weight (no 4 KiB multiples)
static inline int64_t glm_expand_pitch(int cols) { return cols + (((int64_t)cols * 2) % 4096 == 0 ? 64 : 0); }

// SiLU for the GEMM epilogues: g * rcp(1 + 2^(-g*log2 e)) on the hardware exp2/rcp units
// (each within 1 ulp fp32).  The result is rounded to bf16 right away, so against the precise
// expf + IEEE-division form it can differ by one bf16 ulp only on a rounding boundary; the
// precise form costs ~6x the VALU work, which a 1-workgroup-per-CU GEMM epilogue cannot hide.
__device__ __forceinline__ float glm_silu(float g) {
    return g * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * g));
}

// Locate the (group, row-block) a workgroup owns: combo index `target` enumerates
// ceil(cnt_g / MB) blocks per group in group order.  Executed by the whole first wave;
// result broadcast through `sh` (3 ints of LDS).  g = -1 ⇒ no such combo.
__device__ __forceinline__ void glm_locate_block(const int32_t* __restrict__ offs, int G, int64_t R,
                                                 int MB, int target, int* sh) {
    const int tid = threadIdx.x;
    if (offs == nullptr) {
        if (tid == 0) {
            const int64_t m0 = (int64_t)target * MB;
            if (m0 < R) {
                sh[0] = 0;
                sh[1] = (int)m0;
                sh[2] = (int)((R - m0) < MB ? (R - m0) : MB);
            } else {
                sh[0] = -1;
            }
        }
        return;
    }
    if (tid >= 64) return;
    if (tid == 0) sh[0] = -1;
    int base = 0;
    for (int b = 0; b < G; b += 64) {
        const int gi = b + tid;
        int lo = 0, cnt = 0;
        if (gi < G) {
            lo = offs[gi];
            cnt = offs[gi + 1] - lo;
        }
        const int nt = (cnt + MB - 1) / MB;
        int incl = nt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (tid >= o) incl += v;
        }
        const int excl = incl - nt;
        if (target >= base + excl && target < base + incl) {
            const int m0 = lo + (target - base - excl) * MB;
            sh[0] = gi;
            sh[1] = m0;
            sh[2] = (lo + cnt - m0) < MB ? (lo + cnt - m0) : MB;
        }
        base += __shfl(incl, 63, 64);
        if (base > target) break;  // uniform
    }
}

// ---- launch-path options (api.hip) -------------------------------------------------------------------------------------
// Every switch a launch path consults lives in ONE table, filled from the environment ONCE (first use) and changed only
// through glm_set_option: a launch never reads the environment, so a HIP graph and the eager launches around it cannot see
// two different settings, and the effective set can be printed (bench.py puts it into its line's `config.options`).
enum GlmOption { GLM_OPT_TINYM = 0, GLM_OPT_FP8W_NATIVE, GLM_OPT_LARGEM_VARIANT, GLM_OPT_LARGEM_CFG, GLM_OPT_FA_MFMA32, GLM_OPT_FA_ROWS64, GLM_OPT_FA_PERSIST, GLM_OPT_COUNT };
int glm_option(int id);
// Kernels that were built, measured against the shipped ones and LOST (r03: prefill attention on 32x32x16 MFMAs; r04: prefill
// attention with 64 query rows per wave) are compiled into the STUDY library only (`build.py --study` ->
// lib/libglm4moe_hip_study.so, -DGLM_STUDY_KERNELS=1); in the product library their options answer GLM_ERR_UNSUPPORTED.
#ifndef GLM_STUDY_KERNELS
#define GLM_STUDY_KERNELS 0
#endif
// prefill_attn64.hip (study build): the 64-rows-per-wave prefill attention (arguments validated by glm_varlen_attn_prefill)
int glm_launch_prefill_attn64(const void* q, const void* k, const void* v, const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                              const int32_t* block_tables, int64_t bt_stride, void* out, int64_t B, int max_seqlen_q, int nh, int nkv,
                              int block_size, float scale, hipStream_t st);

// prefill_attn_persist.hip (r05): the contiguous-K/V prefill attention as persistent workgroups (option fa_persist); returns
// GLM_ERR_UNSUPPORTED without enqueuing anything when a slot's block list does not fit its table
int glm_launch_prefill_attn_persist(const void* q, const void* k, const void* v, const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                                    void* out, int64_t B, int max_seqlen_q, int nh, int nkv, float scale, hipStream_t st);
