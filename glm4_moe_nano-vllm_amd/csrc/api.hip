// api.hip — error plumbing, ABI version and the GEMM entry points (dispatch between the
// weight-streaming small-M kernels and the LDS-tiled large-M kernels).
#include "gemm_internal.h"
#include <string.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";

void glm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int glm_abi_version(void) { return 10; }
extern "C" const char* glm_last_error(void) { return g_err; }

// ---- launch-path options: name, environment variable read once, default ------------------------------------------------
namespace {
struct OptDesc {
    const char* name;
    const char* env;
    int dflt;
};
const OptDesc k_opts[GLM_OPT_COUNT] = {
    {"tinym", "GLM4MOE_TINYM", 1},                 // 0: one token row keeps the small-M MFMA kernels (A/B, bit-identity tests)
    {"fp8w_native", "GLM4MOE_FP8W_NATIVE", 1},     // 0: fp8 experts at prefill row counts take the expansion path
    {"largem_variant", "GLM_LARGEM_VARIANT", -1},  // >= 0: loop variant bits of the tiled GEMM (0 = the plain loop)
    {"largem_cfg", "GLM_LARGEM_CFG", 0},           // 0 = by shape, 1 = 128-row blocks (A), 2 = 256-row blocks (B)
    {"fa_mfma32", "GLM_FA_MFMA32", -1},            // -1 = the build's default, 0 = 16x16x32 kernel, 1 = 32x32x16 kernel
    {"fa_rows64", "GLM_FA_ROWS64", 0},             // 1 = prefill attention with 64 query rows per wave (prefill_attn64.hip)
    {"fa_persist", "GLM_FA_PERSIST", 0},           // 1 = prefill attention as persistent workgroups (prefill_attn_persist.hip; study library)
};
int g_opt[GLM_OPT_COUNT];
bool g_opt_init = false;
void opts_init() {
    if (g_opt_init) return;
    for (int i = 0; i < GLM_OPT_COUNT; ++i) {
        const char* e = getenv(k_opts[i].env);
        int v = k_opts[i].dflt;
        if (e && *e) {
            if (i == GLM_OPT_LARGEM_CFG) v = (e[0] == 'A' || e[0] == 'a') ? 1 : (e[0] == 'B' || e[0] == 'b') ? 2 : atoi(e);
            else v = atoi(e);
        }
        g_opt[i] = v;
    }
    g_opt_init = true;
}
int opt_index(const char* name) {
    for (int i = 0; name && i < GLM_OPT_COUNT; ++i)
        if (strcmp(name, k_opts[i].name) == 0) return i;
    return -1;
}
}  // namespace

int glm_option(int id) {
    opts_init();
    return g_opt[id];
}

extern "C" int glm_get_option(const char* name, int* value) {
    const int i = opt_index(name);
    GLM_CHECK_ARG(i >= 0 && value, GLM_ERR_INVALID, "glm_get_option: unknown option '%s'", name ? name : "(null)");
    *value = glm_option(i);
    return GLM_OK;
}

extern "C" int glm_set_option(const char* name, int value) {
    const int i = opt_index(name);
    GLM_CHECK_ARG(i >= 0, GLM_ERR_INVALID, "glm_set_option: unknown option '%s'", name ? name : "(null)");
    opts_init();
    g_opt[i] = value;
    return GLM_OK;
}

extern "C" const char* glm_option_name(int index) { return index >= 0 && index < GLM_OPT_COUNT ? k_opts[index].name : nullptr; }

namespace {
// rows per group above which the MFMA-bound tiled kernel is used instead of weight streaming
constexpr int64_t LARGEM_MIN_AVG_ROWS = 96;

// Many rows per group with fp8 weights (prefill): the GEMM is MFMA-bound.  First choice: the tiled kernel takes the fp8
// bytes as they are (DMA-staged as fp8, converted exactly to bf16 between the fragment read and the MFMA; per-row
// dequantisation scale on the fp32 accumulators) — no expansion pass, no scratch.  What that kernel does not cover (a K
// that is not a multiple of 64, offsets past 2^31) takes the r02 path when the caller brought scratch: the stack is
// expanded (exactly) to bf16 by one streaming pass and the bf16 kernel runs on the copy, same accumulator scaling — the two
// are bit-identical (test_fp8_weights_native_operand_equals_expansion).  GLM_ERR_UNSUPPORTED: neither applies.
bool fp8w_native_enabled() { return glm_option(GLM_OPT_FP8W_NATIVE) != 0; }  // 0: force the expansion path

int fp8w_large(int mode, GemmParams p, int rows, int cols, void* scratch, size_t scratch_bytes, hipStream_t st) {
    if (fp8w_native_enabled()) {
        GemmParams q = p;
        q.w_fp8 = 1;
        const int rc = glm_launch_gemm_largem(mode, q, st);
        if (rc != GLM_ERR_UNSUPPORTED) return rc;
    }
    if (scratch == nullptr) return GLM_ERR_UNSUPPORTED;
    const int64_t pitch = glm_expand_pitch(cols);
    const size_t need = (size_t)p.G * rows * pitch * sizeof(bf16_t);
    GLM_CHECK_ARG(scratch_bytes >= need && glm_aligned16(scratch), GLM_ERR_WORKSPACE,
                  "fp8 expert GEMM: expansion scratch %zu < %zu bytes", scratch_bytes, need);
    // fp8 strides are in elements = bytes
    const int rc = glm_launch_fp8_expand(p.w, p.ldw, p.w_gstride, scratch, pitch, (int64_t)rows * pitch, p.G, rows, cols, st);
    if (rc != GLM_OK) return rc;
    p.w = (const bf16_t*)scratch;
    p.ldw = pitch;
    p.w_gstride = (int64_t)rows * pitch;
    return glm_launch_gemm_largem(mode, p, st);  // p.wscale stays set
}

// option tinym = 0 keeps the small-M kernels at one token row (A/B runs and the bit-identity tests of the fused launches)
bool tinym_enabled() { return glm_option(GLM_OPT_TINYM) != 0; }

int dispatch(int mode, const GemmParams& p, hipStream_t st) {
    if (p.R <= 0) return GLM_OK;
    const int64_t avg = (p.R_hint > 0 ? p.R_hint : p.R) / (p.G > 0 ? p.G : 1);
    if (avg >= LARGEM_MIN_AVG_ROWS) return glm_launch_gemm_largem(mode, p, st);
    return glm_launch_gemm_smallm(mode, p, st);
}
}  // namespace

extern "C" int glm_gemm_bf16(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias,
                             void* y, int64_t ldy, int64_t M, int N, int K, glm_stream_t stream) {
    GLM_CHECK_ARG(x && w && y, GLM_ERR_INVALID, "glm_gemm_bf16: null pointer");
    GLM_CHECK_ARG(M >= 0 && N > 0 && K > 0 && ldx >= K && ldy >= N && ldw >= K, GLM_ERR_INVALID,
                  "glm_gemm_bf16: bad shape M=%ld N=%d K=%d ldx=%ld ldy=%ld", (long)M, N, K, (long)ldx, (long)ldy);
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)x; p.ldx = ldx;
    p.w = (const bf16_t*)w; p.w_gstride = 0; p.ldw = ldw;
    p.bias = (const bf16_t*)bias;
    p.y = (bf16_t*)y; p.ldy = ldy;
    p.R = M; p.N = N; p.K = K; p.G = 1;
    // single-sequence decode: the GEMV body (two rows keep the small-M MFMA kernels: gemm_tinym.hip, glm_tinym_max_rows)
    if (M == 1 && tinym_enabled()) {
        const int rc = glm_launch_gemm_tinym(GEMM_PLAIN, p, (int)M, glm_s(stream));
        if (rc != GLM_ERR_UNSUPPORTED) return rc;
    }
    return dispatch(GEMM_PLAIN, p, glm_s(stream));
}

extern "C" int glm_gemm_qkv_rope_store(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, int64_t T,
                                       int nh, int nkv, int hd, int K, const int64_t* positions, const float* cos_sin_cache,
                                       int rotary_dim, void* q_out, void* k_out, void* v_out, void* k_cache, void* v_cache,
                                       const int32_t* slot_mapping, int64_t n_slots, glm_stream_t stream) {
    GLM_CHECK_ARG(x && w && positions && cos_sin_cache && q_out, GLM_ERR_INVALID, "glm_gemm_qkv_rope_store: null pointer");
    GLM_CHECK_ARG(T >= 0 && nh > 0 && nkv > 0 && K > 0 && ldx >= K && ldw >= K, GLM_ERR_INVALID,
                  "glm_gemm_qkv_rope_store: bad shape T=%ld nh=%d nkv=%d K=%d", (long)T, nh, nkv, K);
    GLM_CHECK_ARG((k_cache == nullptr) == (v_cache == nullptr) && (k_out == nullptr) == (v_out == nullptr), GLM_ERR_INVALID,
                  "glm_gemm_qkv_rope_store: K and V outputs come in pairs");
    // the fused epilogue exists for the tiled kernel at head_dim 128 / rotary_dim 64 (GLM-4.5 family); anything else, and row
    // counts the tiled kernel is not chosen for, is the caller's two-launch path (glm_gemm_bf16 + glm_qkv_split_norm_rope_store)
    if (hd != 128 || rotary_dim != 64 || T < LARGEM_MIN_AVG_ROWS) {
        glm_set_error("glm_gemm_qkv_rope_store: not covered (hd=%d rotary_dim=%d T=%ld)", hd, rotary_dim, (long)T);
        return GLM_ERR_UNSUPPORTED;
    }
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)x; p.ldx = ldx;
    p.w = (const bf16_t*)w; p.ldw = ldw;
    p.bias = (const bf16_t*)bias;
    p.R = T; p.N = (nh + 2 * nkv) * hd; p.K = K; p.G = 1;
    p.qkv_q = (bf16_t*)q_out; p.qkv_k = (bf16_t*)k_out; p.qkv_v = (bf16_t*)v_out;
    p.qkv_kc = (bf16_t*)k_cache; p.qkv_vc = (bf16_t*)v_cache;
    p.qkv_slots = slot_mapping; p.qkv_nslots = n_slots;
    p.qkv_pos = positions; p.qkv_cs = cos_sin_cache;
    p.qkv_nh = nh; p.qkv_nkv = nkv;
    return glm_launch_gemm_largem_qkv(p, glm_s(stream));
}

extern "C" int glm_gemm_bf16_combine_norm(const void* y_prev, int64_t ld_yprev, const int32_t* pair_pos, int64_t T, int k,
                                          int64_t shared_base, const void* residual_in, const void* norm_weight,
                                          float eps, void* residual_out, void* x_out, const void* w, int64_t ldw,
                                          const void* bias, void* y, int64_t ldy, int N, int H, glm_stream_t stream) {
    GLM_CHECK_ARG(y_prev && pair_pos && residual_in && norm_weight && residual_out && w && y, GLM_ERR_INVALID,
                  "glm_gemm_bf16_combine_norm: null pointer");
    GLM_CHECK_ARG(T >= 1 && k > 0 && N > 0 && H > 0 && ld_yprev >= H && ldw >= H && ldy >= N, GLM_ERR_INVALID,
                  "glm_gemm_bf16_combine_norm: bad shape T=%ld k=%d N=%d H=%d", (long)T, k, N, H);
    GLM_CHECK_ARG(T <= glm_tinym_max_rows(), GLM_ERR_UNSUPPORTED, "glm_gemm_bf16_combine_norm: T=%ld > %d rows", (long)T,
                  glm_tinym_max_rows());
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.w = (const bf16_t*)w; p.ldw = ldw;
    p.bias = (const bf16_t*)bias;
    p.y = (bf16_t*)y; p.ldy = ldy;
    p.R = T; p.N = N; p.K = H; p.G = 1;
    return glm_launch_gemm_tinym_combine_norm(p, y_prev, ld_yprev, pair_pos, k, shared_base, residual_in, norm_weight,
                                              residual_out, x_out, eps, glm_s(stream));
}

extern "C" int glm_gemm_bf16_router_tail(const void* x, const void* w, int64_t ldw, int K, int H, void* o_row,
                                         uint32_t* ticket, const void* residual_in, const void* norm_weight, float eps,
                                         void* y_norm, void* residual_out, const void* gate_w,
                                         const float* e_score_correction_bias, int E, int k, float routed_scaling_factor,
                                         int64_t* topk_idx, void* topk_w, int32_t* expert_offsets, int32_t* sorted_token,
                                         void* sorted_scale, int32_t* pair_pos, int with_shared, glm_stream_t stream) {
    GLM_CHECK_ARG(x && w && o_row && ticket && residual_in && norm_weight && y_norm && residual_out && gate_w &&
                      e_score_correction_bias && topk_idx && topk_w && expert_offsets && sorted_token && sorted_scale && pair_pos,
                  GLM_ERR_INVALID, "glm_gemm_bf16_router_tail: null pointer");
    GLM_CHECK_ARG(K > 0 && H > 0 && ldw >= K && E > 0 && k > 0 && k <= E, GLM_ERR_INVALID,
                  "glm_gemm_bf16_router_tail: bad shape K=%d H=%d E=%d k=%d", K, H, E, k);
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)x; p.ldx = K;
    p.w = (const bf16_t*)w; p.ldw = ldw;
    p.y = (bf16_t*)o_row; p.ldy = H;
    p.R = 1; p.N = H; p.K = K; p.G = 1;
    // ticket words 0/1 = the two counters, words 4.. = E floats of logits scratch (the caller provides 16 + 4 E bytes)
    return glm_launch_gemm_tinym_router_tail(p, ticket, reinterpret_cast<float*>(ticket + 4), residual_in, norm_weight, eps, y_norm, residual_out, gate_w,
                                             e_score_correction_bias, E, k, routed_scaling_factor, topk_idx, topk_w,
                                             expert_offsets, sorted_token, sorted_scale, pair_pos, with_shared, glm_s(stream));
}

extern "C" int glm_gemm_bf16_partial_slices(int K, int want) { return glm_smallm_partial_slices(K, want); }

extern "C" int glm_gemm_bf16_partial(const void* x, int64_t ldx, const void* w, int64_t ldw, float* ypart,
                                     int64_t M, int N, int K, int nslices, glm_stream_t stream) {
    GLM_CHECK_ARG(x && w && ypart, GLM_ERR_INVALID, "glm_gemm_bf16_partial: null pointer");
    GLM_CHECK_ARG(M >= 0 && N > 0 && K > 0 && ldx >= K && ldw >= K && nslices > 1, GLM_ERR_INVALID,
                  "glm_gemm_bf16_partial: bad shape M=%ld N=%d K=%d slices=%d", (long)M, N, K, nslices);
    if (M == 0) return GLM_OK;
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)x; p.ldx = ldx;
    p.w = (const bf16_t*)w; p.w_gstride = 0; p.ldw = ldw;
    p.R = M; p.N = N; p.K = K; p.G = 1;
    p.ksplit = nslices; p.ypart = ypart;
    return glm_launch_gemm_smallm_partial(p, glm_s(stream));
}

extern "C" int glm_grouped_gemm_gateup_silu(const void* x, int64_t ldx, const void* w13,
                                            int64_t w13_group_stride, int64_t w13_row_stride,
                                            const int32_t* expert_offsets, const int32_t* sorted_token,
                                            void* act, int64_t R, int H, int I, int G,
                                            glm_stream_t stream) {
    GLM_CHECK_ARG(x && w13 && act, GLM_ERR_INVALID, "glm_grouped_gemm_gateup_silu: null pointer");
    GLM_CHECK_ARG(R >= 0 && H > 0 && I > 0 && G >= 1 && ldx >= H && w13_row_stride >= H &&
                      (G == 1 || w13_group_stride >= (int64_t)2 * I * w13_row_stride),
                  GLM_ERR_INVALID, "glm_grouped_gemm_gateup_silu: bad shape");
    GLM_CHECK_ARG(expert_offsets != nullptr || G == 1, GLM_ERR_INVALID,
                  "glm_grouped_gemm_gateup_silu: expert_offsets required when G > 1");
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)x; p.ldx = ldx;
    p.w = (const bf16_t*)w13; p.w_gstride = w13_group_stride; p.ldw = w13_row_stride;
    p.offs = expert_offsets; p.src = sorted_token;
    p.y = (bf16_t*)act; p.ldy = I;
    p.R = R; p.N = I; p.K = H; p.G = G;
    return dispatch(GEMM_GATEUP, p, glm_s(stream));
}

extern "C" int glm_grouped_gemm_gateup_silu_fewtok(const void* x, int64_t ldx, int64_t T, const void* w13,
                                                   int64_t w13_group_stride, int64_t w13_row_stride,
                                                   const int32_t* expert_offsets, const int32_t* sorted_token,
                                                   void* act, int64_t R, int H, int I, int G, glm_stream_t stream) {
    GLM_CHECK_ARG(T >= 1 && expert_offsets != nullptr, GLM_ERR_INVALID, "glm_grouped_gemm_gateup_silu_fewtok: bad arguments");
    if (T == 1 && tinym_enabled() && x && w13 && act && R > 0 && H > 0 && I > 0 && G >= 1 && ldx >= H &&
        w13_row_stride >= H) {
        GemmParams p;
        memset(&p, 0, sizeof(p));
        p.x = (const bf16_t*)x; p.ldx = ldx;
        p.w = (const bf16_t*)w13; p.w_gstride = w13_group_stride; p.ldw = w13_row_stride;
        p.offs = expert_offsets; p.src = sorted_token;
        p.y = (bf16_t*)act; p.ldy = I;
        p.R = R; p.N = I; p.K = H; p.G = G;
        const int rc = glm_launch_gemm_tinym(GEMM_GATEUP, p, (int)T, glm_s(stream));
        if (rc != GLM_ERR_UNSUPPORTED) return rc;
    }
    return glm_grouped_gemm_gateup_silu(x, ldx, w13, w13_group_stride, w13_row_stride, expert_offsets, sorted_token, act, R,
                                        H, I, G, stream);
}

extern "C" int glm_grouped_gemm_down(const void* act, const void* w2, int64_t w2_group_stride,
                                     int64_t w2_row_stride, const int32_t* expert_offsets,
                                     const void* sorted_scale, int n_scaled_groups, void* y,
                                     int64_t ldy, int64_t R, int H, int I, int G, glm_stream_t stream) {
    GLM_CHECK_ARG(act && w2 && y, GLM_ERR_INVALID, "glm_grouped_gemm_down: null pointer");
    GLM_CHECK_ARG(R >= 0 && H > 0 && I > 0 && G >= 1 && ldy >= H && w2_row_stride >= I &&
                      (G == 1 || w2_group_stride >= (int64_t)H * w2_row_stride),
                  GLM_ERR_INVALID, "glm_grouped_gemm_down: bad shape");
    GLM_CHECK_ARG(expert_offsets != nullptr || G == 1, GLM_ERR_INVALID,
                  "glm_grouped_gemm_down: expert_offsets required when G > 1");
    GLM_CHECK_ARG(n_scaled_groups == 0 || sorted_scale != nullptr, GLM_ERR_INVALID,
                  "glm_grouped_gemm_down: sorted_scale required when n_scaled_groups > 0");
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)act; p.ldx = I;
    p.w = (const bf16_t*)w2; p.w_gstride = w2_group_stride; p.ldw = w2_row_stride;
    p.offs = expert_offsets; p.src = nullptr;
    p.scale = (const bf16_t*)sorted_scale; p.n_scaled = n_scaled_groups;
    p.y = (bf16_t*)y; p.ldy = ldy;
    p.R = R; p.N = H; p.K = I; p.G = G;
    return dispatch(GEMM_DOWN, p, glm_s(stream));
}

extern "C" int glm_grouped_gemm_down_fewtok(const void* act, int64_t T, const void* w2, int64_t w2_group_stride,
                                            int64_t w2_row_stride, const int32_t* expert_offsets,
                                            const void* sorted_scale, int n_scaled_groups, void* y, int64_t ldy,
                                            int64_t R, int H, int I, int G, glm_stream_t stream) {
    GLM_CHECK_ARG(T >= 1 && expert_offsets != nullptr, GLM_ERR_INVALID, "glm_grouped_gemm_down_fewtok: bad arguments");
    if (T == 1 && tinym_enabled() && act && w2 && y && R > 0 && H > 0 && I > 0 && G >= 1 && ldy >= H &&
        w2_row_stride >= I && (n_scaled_groups == 0 || sorted_scale != nullptr)) {
        GemmParams p;
        memset(&p, 0, sizeof(p));
        p.x = (const bf16_t*)act; p.ldx = I;
        p.w = (const bf16_t*)w2; p.w_gstride = w2_group_stride; p.ldw = w2_row_stride;
        p.offs = expert_offsets;
        p.scale = (const bf16_t*)sorted_scale; p.n_scaled = n_scaled_groups;
        p.y = (bf16_t*)y; p.ldy = ldy;
        p.R = R; p.N = H; p.K = I; p.G = G;
        const int rc = glm_launch_gemm_tinym(GEMM_DOWN, p, (int)T, glm_s(stream));
        if (rc != GLM_ERR_UNSUPPORTED) return rc;
    }
    return glm_grouped_gemm_down(act, w2, w2_group_stride, w2_row_stride, expert_offsets, sorted_scale, n_scaled_groups, y,
                                 ldy, R, H, I, G, stream);
}

// Fixed-capacity expert-parallel buffers (glm_ep_plan): the row arrays are sized for the worst case, expert_offsets say
// how many rows are real.  R_cap sizes the grid (surplus workgroups find no block and leave), R_expected picks the kernel
// configuration the real row count would get.
extern "C" int glm_grouped_gemm_gateup_silu_cap(const void* x, int64_t ldx, const void* w13, int64_t w13_group_stride,
                                                int64_t w13_row_stride, const int32_t* expert_offsets,
                                                const int32_t* sorted_token, void* act, int64_t R_cap, int64_t R_expected,
                                                int H, int I, int G, glm_stream_t stream) {
    GLM_CHECK_ARG(x && w13 && act && expert_offsets, GLM_ERR_INVALID, "glm_grouped_gemm_gateup_silu_cap: null pointer");
    GLM_CHECK_ARG(R_cap >= 0 && R_expected >= 0 && H > 0 && I > 0 && G >= 1 && ldx >= H && w13_row_stride >= H &&
                      (G == 1 || w13_group_stride >= (int64_t)2 * I * w13_row_stride),
                  GLM_ERR_INVALID, "glm_grouped_gemm_gateup_silu_cap: bad shape");
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)x; p.ldx = ldx;
    p.w = (const bf16_t*)w13; p.w_gstride = w13_group_stride; p.ldw = w13_row_stride;
    p.offs = expert_offsets; p.src = sorted_token;
    p.y = (bf16_t*)act; p.ldy = I;
    p.R = R_cap; p.R_hint = R_expected > 0 ? R_expected : 1; p.N = I; p.K = H; p.G = G;
    return dispatch(GEMM_GATEUP, p, glm_s(stream));
}

extern "C" int glm_grouped_gemm_down_cap(const void* act, const void* w2, int64_t w2_group_stride, int64_t w2_row_stride,
                                         const int32_t* expert_offsets, const void* sorted_scale, int n_scaled_groups,
                                         void* y, int64_t ldy, int64_t R_cap, int64_t R_expected, int H, int I, int G,
                                         glm_stream_t stream) {
    GLM_CHECK_ARG(act && w2 && y && expert_offsets, GLM_ERR_INVALID, "glm_grouped_gemm_down_cap: null pointer");
    GLM_CHECK_ARG(R_cap >= 0 && R_expected >= 0 && H > 0 && I > 0 && G >= 1 && ldy >= H && w2_row_stride >= I &&
                      (G == 1 || w2_group_stride >= (int64_t)H * w2_row_stride) &&
                      (n_scaled_groups == 0 || sorted_scale != nullptr),
                  GLM_ERR_INVALID, "glm_grouped_gemm_down_cap: bad shape");
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)act; p.ldx = I;
    p.w = (const bf16_t*)w2; p.w_gstride = w2_group_stride; p.ldw = w2_row_stride;
    p.offs = expert_offsets;
    p.scale = (const bf16_t*)sorted_scale; p.n_scaled = n_scaled_groups;
    p.y = (bf16_t*)y; p.ldy = ldy;
    p.R = R_cap; p.R_hint = R_expected > 0 ? R_expected : 1; p.N = H; p.K = I; p.G = G;
    return dispatch(GEMM_DOWN, p, glm_s(stream));
}

extern "C" int glm_grouped_gemm_gateup_silu_fp8w(const void* x, int64_t ldx, const void* w13_q,
                                                 int64_t w13_group_stride, int64_t w13_row_stride,
                                                 const float* w13_scale, int64_t w13_scale_group_stride,
                                                 const int32_t* expert_offsets, const int32_t* sorted_token,
                                                 void* act, int64_t R, int H, int I, int G,
                                                 void* expand_scratch, size_t expand_scratch_bytes,
                                                 glm_stream_t stream) {
    GLM_CHECK_ARG(x && w13_q && w13_scale && act, GLM_ERR_INVALID, "glm_grouped_gemm_gateup_silu_fp8w: null pointer");
    GLM_CHECK_ARG(R >= 0 && H > 0 && I > 0 && G >= 1 && ldx >= H && w13_row_stride >= H &&
                      (G == 1 || (w13_group_stride >= (int64_t)2 * I * w13_row_stride &&
                                  w13_scale_group_stride >= (int64_t)2 * I)) &&
                      (expert_offsets != nullptr || G == 1),
                  GLM_ERR_INVALID, "glm_grouped_gemm_gateup_silu_fp8w: bad shape");
    if (R == 0) return GLM_OK;
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)x; p.ldx = ldx;
    p.w = (const bf16_t*)w13_q; p.w_gstride = w13_group_stride; p.ldw = w13_row_stride;
    p.wscale = w13_scale; p.wscale_gstride = w13_scale_group_stride;
    p.offs = expert_offsets; p.src = sorted_token;
    p.y = (bf16_t*)act; p.ldy = I;
    p.R = R; p.N = I; p.K = H; p.G = G;
    if (R / G >= LARGEM_MIN_AVG_ROWS) {
        const int rc = fp8w_large(GEMM_GATEUP, p, 2 * I, H, expand_scratch, expand_scratch_bytes, glm_s(stream));
        if (rc != GLM_ERR_UNSUPPORTED) return rc;
    }
    return glm_launch_gemm_smallm_fp8w(GEMM_GATEUP, p, glm_s(stream));
}

extern "C" int glm_grouped_gemm_down_fp8w(const void* act, const void* w2_q, int64_t w2_group_stride,
                                          int64_t w2_row_stride, const float* w2_scale,
                                          int64_t w2_scale_group_stride, const int32_t* expert_offsets,
                                          const void* sorted_scale, int n_scaled_groups, void* y,
                                          int64_t ldy, int64_t R, int H, int I, int G, void* expand_scratch,
                                          size_t expand_scratch_bytes, glm_stream_t stream) {
    GLM_CHECK_ARG(act && w2_q && w2_scale && y, GLM_ERR_INVALID, "glm_grouped_gemm_down_fp8w: null pointer");
    GLM_CHECK_ARG(R >= 0 && H > 0 && I > 0 && G >= 1 && ldy >= H && w2_row_stride >= I &&
                      (G == 1 || (w2_group_stride >= (int64_t)H * w2_row_stride && w2_scale_group_stride >= H)) &&
                      (expert_offsets != nullptr || G == 1) && (n_scaled_groups == 0 || sorted_scale != nullptr),
                  GLM_ERR_INVALID, "glm_grouped_gemm_down_fp8w: bad shape");
    if (R == 0) return GLM_OK;
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)act; p.ldx = I;
    p.w = (const bf16_t*)w2_q; p.w_gstride = w2_group_stride; p.ldw = w2_row_stride;
    p.wscale = w2_scale; p.wscale_gstride = w2_scale_group_stride;
    p.offs = expert_offsets; p.src = nullptr;
    p.scale = (const bf16_t*)sorted_scale; p.n_scaled = n_scaled_groups;
    p.y = (bf16_t*)y; p.ldy = ldy;
    p.R = R; p.N = H; p.K = I; p.G = G;
    if (R / G >= LARGEM_MIN_AVG_ROWS) {
        const int rc = fp8w_large(GEMM_DOWN, p, H, I, expand_scratch, expand_scratch_bytes, glm_s(stream));
        if (rc != GLM_ERR_UNSUPPORTED) return rc;
    }
    return glm_launch_gemm_smallm_fp8w(GEMM_DOWN, p, glm_s(stream));
}

// r05: fp8 x fp8 on the block-scaled MFMA (BASELINE configs[4]); prefill-sized groups only
extern "C" int glm_grouped_gemm_gateup_silu_fp8a8(const void* x_q, int64_t ldx, const float* x_scale, const void* w13_q,
                                                  int64_t w13_group_stride, int64_t w13_row_stride, const float* w13_scale,
                                                  int64_t w13_scale_group_stride, const int32_t* expert_offsets,
                                                  const int32_t* sorted_token, void* act_q, int64_t ld_act, void* act_bscale,
                                                  int64_t R, int H, int I, int G, glm_stream_t stream) {
    GLM_CHECK_ARG(x_q && x_scale && w13_q && w13_scale && act_q && act_bscale && expert_offsets, GLM_ERR_INVALID,
                  "glm_grouped_gemm_gateup_silu_fp8a8: null pointer");
    GLM_CHECK_ARG(R >= 0 && H > 0 && I > 0 && G >= 1 && ldx >= H && w13_row_stride >= H && ld_act >= I &&
                      (G == 1 || (w13_group_stride >= (int64_t)2 * I * w13_row_stride && w13_scale_group_stride >= (int64_t)2 * I)),
                  GLM_ERR_INVALID, "glm_grouped_gemm_gateup_silu_fp8a8: bad shape");
    if (R == 0) return GLM_OK;
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.a8 = 1;
    p.x = (const bf16_t*)x_q; p.ldx = ldx; p.xscale = x_scale;
    p.w = (const bf16_t*)w13_q; p.w_gstride = w13_group_stride; p.ldw = w13_row_stride;
    p.wscale = w13_scale; p.wscale_gstride = w13_scale_group_stride;
    p.offs = expert_offsets; p.src = sorted_token;
    p.y = (bf16_t*)act_q; p.ldy = ld_act; p.ybscale = (unsigned char*)act_bscale;
    p.R = R; p.N = I; p.K = H; p.G = G;
    return glm_launch_gemm_largem_a8(GEMM_GATEUP, p, glm_s(stream));
}

extern "C" int glm_grouped_gemm_down_fp8a8(const void* act_q, int64_t ld_act, const void* act_bscale, const void* w2_q,
                                           int64_t w2_group_stride, int64_t w2_row_stride, const float* w2_scale,
                                           int64_t w2_scale_group_stride, const int32_t* expert_offsets,
                                           const void* sorted_scale, int n_scaled_groups, void* y, int64_t ldy, int64_t R,
                                           int H, int I, int G, glm_stream_t stream) {
    GLM_CHECK_ARG(act_q && act_bscale && w2_q && w2_scale && y && expert_offsets, GLM_ERR_INVALID,
                  "glm_grouped_gemm_down_fp8a8: null pointer");
    GLM_CHECK_ARG(R >= 0 && H > 0 && I > 0 && G >= 1 && ldy >= H && w2_row_stride >= I && ld_act >= I &&
                      (G == 1 || (w2_group_stride >= (int64_t)H * w2_row_stride && w2_scale_group_stride >= H)) &&
                      (n_scaled_groups == 0 || sorted_scale != nullptr),
                  GLM_ERR_INVALID, "glm_grouped_gemm_down_fp8a8: bad shape");
    if (R == 0) return GLM_OK;
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.a8 = 1;
    p.x = (const bf16_t*)act_q; p.ldx = ld_act; p.xbscale = (const unsigned char*)act_bscale;
    p.w = (const bf16_t*)w2_q; p.w_gstride = w2_group_stride; p.ldw = w2_row_stride;
    p.wscale = w2_scale; p.wscale_gstride = w2_scale_group_stride;
    p.offs = expert_offsets; p.src = nullptr;
    p.scale = (const bf16_t*)sorted_scale; p.n_scaled = n_scaled_groups;
    p.y = (bf16_t*)y; p.ldy = ldy;
    p.R = R; p.N = H; p.K = I; p.G = G;
    return glm_launch_gemm_largem_a8(GEMM_DOWN, p, glm_s(stream));
}

extern "C" size_t glm_fp8w_expand_scratch_bytes(int G, int rows, int cols) {
    // 0: the tiled kernel takes this stack's fp8 bytes directly (K a multiple of its 64-wide K tile)
    if (fp8w_native_enabled() && cols % 64 == 0) return 0;
    return (size_t)G * rows * glm_expand_pitch(cols) * sizeof(bf16_t);
}
