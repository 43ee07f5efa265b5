// gemm_smallm.hip — weight-streaming bf16 GEMM for the decode regime (few rows per weight
// matrix): dense QKV/O projections (a5/a10) and the grouped expert up/gate/down projections
// (a12/a13) of nanovllm's GLM-4-MoE layer.  gfx950 only.
//
// Roofline: HBM.  At bs=32 every weight byte is used for <= 64 FLOP, so the kernel is a
// streaming reader: algorithmic bytes = all touched weight bytes, read exactly once.
// out^T = W · x^T with mfma_f32_16x16x32_bf16: the 16 weight rows of a tile sit on the MFMA row axis (A operand),
// tokens on the lane/column axis (B operand).  Two bodies share the work decomposition, the K-slice reduction and
// the epilogues:
//
//   * LDS-DMA body (smallm_body_dma, further down; what the decode step runs for bf16 weights): per-wave 2-stage
//     LDS rings filled by global_load_lds (weights with the nt hint), no workgroup barrier — see its header.
//   * VGPR body (smallm_body): weights straight from HBM into VGPRs in fragment order (64 B contiguous per row per
//     instruction); the (tiny, L2-resident) activation chunk is staged once per workgroup per 128-wide K chunk in
//     LDS, XOR-swizzled (chunk ^ (row & 15)) so the 16-row fragment reads (ds_read_b128) are bank-conflict free,
//     double-buffered, one barrier per chunk.  Each weight register is refilled for the NEXT chunk right after its
//     last MFMA, so one full chunk (8 x 1 KiB per wave) stays in flight across the barrier; the activation loads are
//     issued before it in program order, so waiting for them never drains the stream.  Used for fp8 weights, for
//     single-sequence decode, very short K and more than 32 rows per block.
//     (Measured alternatives that lost on the same box: B fragments straight from L2 without LDS and barriers,
//     298 vs 267 us for the grouped gate|up; two-chunk-deep weight registers at half the resident waves, 281 us.)
//
//   * waves of a workgroup own different weight rows (WR) and/or different K slices of the same rows (WS); K-slice
//     partials are reduced through LDS.  Config sweeps (tools/sweep_*.sh) picked small workgroups everywhere.
//   * grouped mode: blockIdx.y enumerates (expert, 32-row block) pairs, located by a wave scan over
//     expert_offsets — no host sync, graph-capturable (the reference's per-expert Python loop with
//     .any()/where host syncs is moe.py:135-151).
#include "gemm_internal.h"
#include <stdlib.h>
#include <type_traits>

namespace {

// 8 OCP e4m3 bytes (two dwords) -> 8 bf16, exact (every e4m3 value, subnormals included, is a bf16 value): gfx950's packed
// convert takes two bytes per instruction (4 per fragment; cvt_pk_f32_fp8 + cvt_pk_bf16_f32 took 8)
#ifndef GLM_SM_FP8_CVT
#define GLM_SM_FP8_CVT 1  // 0: the two-step conversion (A/B)
#endif
__device__ __forceinline__ u32x4_t sm_fp8x8_to_bf16x8(unsigned d0, unsigned d1) {
#if GLM_SM_FP8_CVT
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
    const bf2_t a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d0, 1.0f, false);
    const bf2_t b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d0, 1.0f, true);
    const bf2_t c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d1, 1.0f, false);
    const bf2_t d = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d1, 1.0f, true);
    return u32x4_t{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, c),
                   __builtin_bit_cast(unsigned, d)};
#else
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    const f32x2_t a = __builtin_amdgcn_cvt_pk_f32_fp8(d0, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(d0, true);
    const f32x2_t c = __builtin_amdgcn_cvt_pk_f32_fp8(d1, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(d1, true);
    return u32x4_t{pack2bf(a[0], a[1]), pack2bf(b[0], b[1]), pack2bf(c[0], c[1]), pack2bf(d[0], d[1])};
#endif
}

// 16 B of zeros that out-of-range K lanes load instead of real data (K tails, short K slices)
__device__ const u32x4_t g_zero_page[4] = {};

// 16-byte load through an explicit GLOBAL address-space pointer: the ?: with the zero page makes
// the pointer generic, and a flat_load counts on vmcnt AND lgkmcnt, out of order (vmcnt(0) drains)
__device__ __forceinline__ u32x4_t gload16(const void* p) {
    return *reinterpret_cast<const __attribute__((address_space(1))) u32x4_t*>(
        (const __attribute__((address_space(1))) void*)p);
}

// Timing-only ablation builds (tools/ablate_smallm.sh): bit 0 no MFMA, bit 1 no x staging,
// bit 2 no B-fragment LDS reads, bit 3 no barrier.  Results are wrong in those builds.
#ifndef GLM_ABLATE
#define GLM_ABLATE 0
#endif
#ifndef GLM_PIN_BFR
#define GLM_PIN_BFR 0
#endif

#ifndef GLM_SM_KS
#define GLM_SM_KS 4
#endif
constexpr int SM_KS = GLM_SM_KS;   // k-steps (of 32) per chunk
constexpr int SM_KC = 32 * SM_KS;  // 128 elements = 256 B per row per chunk
constexpr int SM_XROWB = SM_KC * 2;  // bytes per activation row per chunk in LDS
constexpr int SM_CPR = SM_KC / 8;    // 16-B chunks per row

// epilogue shared by both variants: lane (c,g4) holds, per (rt, mt), columns n..n+3 of token
// row 16*mt + c
template <int MODE, int RT, int MT>
__device__ __forceinline__ void smallm_epilogue(const GemmParams& p, f32x4_t (&acc)[RT][MT], int grp, int m0,
                                                int mcnt, int ncol0, int c, int g4) {
    if (MODE == GEMM_PLAIN && p.ypart != nullptr) {
        // K-split: this workgroup's fp32 partial sums; bias and the bf16 rounding belong to the consumer
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = 16 * mt + c;
            if (row >= mcnt) continue;
            float* prow = p.ypart + (int64_t)(m0 + row) * p.N;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int n = ncol0 + 16 * rt + 4 * g4;
                if (n < p.N) *reinterpret_cast<f32x4_t*>(prow + n) = acc[rt][mt];
            }
        }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = 16 * mt + c;
        if (row >= mcnt) continue;
        bf16_t* yrow = p.y + (int64_t)(m0 + row) * p.ldy;
        if (MODE == GEMM_GATEUP) {
            // row tiles [0, RT/2) are gate columns, [RT/2, RT) the matching up columns
#pragma unroll
            for (int j = 0; j < RT / 2; ++j) {
                const int n = ncol0 + 16 * j + 4 * g4;
                if (n < p.N) {  // N % 4 == 0 is checked on the host
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        o[r] = rbf(glm_silu(rbf(acc[j][mt][r]))) * rbf(acc[RT / 2 + j][mt][r]);
                    u32x2_t v = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
                    *reinterpret_cast<u32x2_t*>(yrow + n) = v;
                }
            }
        } else {
            float sc = 1.0f;
            bool scaled = false;
            if (MODE == GEMM_DOWN && grp < p.n_scaled) {
                sc = bf2f(p.scale[m0 + row]);
                scaled = true;
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int n = ncol0 + 16 * rt + 4 * g4;
                if (n >= p.N) continue;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float a = acc[rt][mt][r];
                    if (MODE == GEMM_PLAIN && p.bias) a += bf2f(p.bias[n + r]);
                    if (scaled) a = rbf(a) * sc;
                    o[r] = a;
                }
                u32x2_t v = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
                *reinterpret_cast<u32x2_t*>(yrow + n) = v;
            }
        }
    }
}

// One workgroup's work for a block of at most 16*MT rows of group `grp` (rows m0 .. m0+mcnt).
// MT is a template parameter so that the chunk loop is straight-line code: every load is
// unconditional (clamped address + select), the last chunk is peeled, and hipcc can place
// counted vmcnt waits instead of draining at control-flow joins.
template <int MODE, int NWAVES, int WS, int RT, int MT, int XSTRIDE, bool W8>
__device__ __forceinline__ void smallm_body(const GemmParams& p, char* smem, int grp, int m0, int mcnt) {
    constexpr int NT = NWAVES * 64;
    constexpr int WR = NWAVES / WS;
    constexpr int MBT = 16 * MT;                          // rows staged by this instantiation
    constexpr int XV = (WS * MBT * SM_CPR + NT - 1) / NT;  // 16-B vectors staged per thread per chunk
    constexpr bool XFULL = (WS * MBT * SM_CPR) % NT == 0;  // every (thread, i) maps to a vector

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ws = wid % WS, wr = wid / WS;
    const int c = lane & 15, g4 = lane >> 4;
    const int K = p.K;
    const int nchunks = (K + SM_KC - 1) / SM_KC;
    const int cps = (nchunks + WS - 1) / WS;  // chunks per K slice

    // ---- weight row pointers of this wave (A operand: lane (c,g4) = row c, k-offset 8*g4) ----
    // byte pointers: weights are bf16 (2 B/element) or, with W8, OCP fp8 e4m3 (1 B/element)
    constexpr int WB = W8 ? 1 : 2;
    const char* wg = reinterpret_cast<const char*>(p.w) + (int64_t)grp * p.w_gstride * WB;
    const char* wrow[RT];
    int wrowi[RT];  // weight row index (for the fp8 per-row scale)
    int ncol0;  // first output column of this wave
    if (MODE == GEMM_GATEUP) {
        ncol0 = (blockIdx.x * WR + wr) * 16;
        const int r = min(ncol0 + c, p.N - 1);
        wrowi[0] = r;
        wrowi[RT - 1] = p.N + r;
    } else {
        ncol0 = (blockIdx.x * WR + wr) * (16 * RT);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) wrowi[rt] = min(ncol0 + 16 * rt + c, p.N - 1);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) wrow[rt] = wg + (int64_t)wrowi[rt] * p.ldw * WB;

    // ---- activation staging: vector v = tid + i*NT ↔ (slice, row, 16-B chunk) ----
    const bf16_t* xrow[XV];
#pragma unroll
    for (int i = 0; i < XV; ++i) {
        const int v = tid + i * NT;
        const int row = (v / SM_CPR) % MBT;
        const int rr = row < mcnt ? row : 0;  // padding rows re-read row 0 (their columns are never stored)
        const int64_t srow = p.src ? (int64_t)p.src[m0 + rr] : (int64_t)(m0 + rr);
        xrow[i] = p.x + srow * p.ldx;
    }
    u32x4_t xr[XV];
    auto x_load = [&](int ci) {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int v = tid + i * NT;
            const int sl = (v / (MBT * SM_CPR)) % WS, ch = v % SM_CPR;
            const int k = (sl * cps + ci) * SM_KC + ch * 8;
            // out-of-range K reads a zero page: selecting the ADDRESS keeps the loaded data
            // free of consumers until the LDS store (a data select would wait right here)
            const bf16_t* src = k < K ? xrow[i] + k : reinterpret_cast<const bf16_t*>(g_zero_page);
            xr[i] = gload16(src);
        }
    };
    auto x_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int v = tid + i * NT;
            const int sl = v / (MBT * SM_CPR), row = (v / SM_CPR) % MBT, ch = v % SM_CPR;
            if (XFULL || v < WS * MBT * SM_CPR)
                *reinterpret_cast<u32x4_t*>(smem + buf * XSTRIDE + (sl * MBT + row) * SM_XROWB +
                                            ((ch ^ (row & 15)) << 4)) = xr[i];
        }
    };

    const int cbeg = ws * cps;
    // weight registers: each one is refilled for the NEXT chunk right after its last MFMA, so a
    // full chunk (8 x 1 KiB per wave) stays in flight across the barrier.  (A 2-chunk-deep
    // variant was measured slower: it costs ~45 VGPRs, i.e. half the resident waves.)
    // bf16: one 16-B load per (row tile, k-step).  fp8: one 16-B load covers TWO k-steps of this
    // lane (16 consecutive k), so k-slot (ks, g4, j) ↔ k = 64*(ks>>1) + 16*g4 + 8*(ks&1) + j and
    // the B fragments below are read with the matching chunk index.
    constexpr int WL = W8 ? SM_KS / 2 : SM_KS;  // 16-B weight loads per row tile per chunk
    u32x4_t wreg[RT][WL];
    auto w_load = [&](int ci, int rt, int l) {
        const int k = W8 ? (cbeg + ci) * SM_KC + 64 * l + 16 * g4 : (cbeg + ci) * SM_KC + 32 * l + 8 * g4;
        const char* src = k < K ? wrow[rt] + (int64_t)k * WB : reinterpret_cast<const char*>(g_zero_page);
        wreg[rt][l] = gload16(src);
    };
    // A fragment of k-step ks from the weight registers (fp8 → bf16 is exact)
    auto a_frag = [&](int rt, int ks) -> bf16x8_t {
        if (!W8) return __builtin_bit_cast(bf16x8_t, wreg[rt][ks]);
        const u32x4_t w = wreg[rt][ks >> 1];
        const unsigned d0 = w[2 * (ks & 1)], d1 = w[2 * (ks & 1) + 1];
        return __builtin_bit_cast(bf16x8_t, sm_fp8x8_to_bf16x8(d0, d1));
    };

    f32x4_t acc[RT][MT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[rt][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- prologue ----
    x_load(0);
#pragma unroll
    for (int l = 0; l < WL; ++l)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) w_load(0, rt, l);
    x_store(0);
    __syncthreads();

    auto step = [&](int ci, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value;
        const int cur = ci & 1;
        if (MORE && !(GLM_ABLATE & 2)) {
            x_load(ci + 1);  // oldest VMEM op of the iteration: its wait never drains the W stream
            __builtin_amdgcn_sched_barrier(0);
        }
        const char* xb = smem + cur * XSTRIDE + ws * MBT * SM_XROWB;
        // all B fragments of the chunk first: the MFMA → refill sequence below then runs without
        // LDS round trips in between, so the next chunk's weight loads are issued as early as possible
        bf16x8_t bfr[SM_KS][MT];
#pragma unroll
        for (int ks = 0; ks < SM_KS; ++ks)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (GLM_ABLATE & 4) {
                    const u32x4_t z = {(unsigned)ci, 0, 0, 0};
                    bfr[ks][mt] = __builtin_bit_cast(bf16x8_t, z);
                } else {
                    const int xch = W8 ? 8 * (ks >> 1) + 2 * g4 + (ks & 1) : 4 * ks + g4;
                    bfr[ks][mt] = *reinterpret_cast<const bf16x8_t*>(xb + (16 * mt + c) * SM_XROWB +
                                                                      ((xch ^ c) << 4));
                }
            }
#if GLM_PIN_BFR
        __builtin_amdgcn_sched_barrier(0);  // experiment: keep the fragment reads as one burst
#endif
#pragma unroll
        for (int ks = 0; ks < SM_KS; ++ks) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (GLM_ABLATE & 1) {
                        const u32x4_t bb = __builtin_bit_cast(u32x4_t, bfr[ks][mt]);
                        const u32x4_t ww = __builtin_bit_cast(u32x4_t, a_frag(rt, ks));
                        acc[rt][mt][0] += __uint_as_float((ww[0] ^ ww[1] ^ ww[2] ^ ww[3] ^ bb[0]) & 0x3fffffffu);
                    } else {
                        acc[rt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_frag(rt, ks), bfr[ks][mt],
                                                                             acc[rt][mt], 0, 0, 0);
                    }
                }
            if (MORE && (!W8 || (ks & 1))) {  // a register is free again after its last k-step
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) w_load(ci + 1, rt, W8 ? ks >> 1 : ks);
            }
        }
        if (MORE && !(GLM_ABLATE & 2)) x_store(cur ^ 1);
        if (!(GLM_ABLATE & 8)) __syncthreads();
    };
    for (int ci = 0; ci + 1 < cps; ++ci) step(ci, std::true_type{});
    step(cps - 1, std::false_type{});

    // ---- K-slice reduction through LDS (WS > 1): slice 0 accumulates the others ----
    if (WS > 1) {
        float* red = reinterpret_cast<float*>(smem);  // x buffers are dead after the last barrier
        if (ws != 0) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    *reinterpret_cast<f32x4_t*>(red + ((((wr * (WS - 1) + (ws - 1)) * RT + rt) * MT + mt) * 64 + lane) * 4) =
                        acc[rt][mt];
        }
        __syncthreads();
        if (ws != 0) return;
#pragma unroll
        for (int s = 1; s < WS; ++s)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[rt][mt] += *reinterpret_cast<const f32x4_t*>(
                        red + ((((wr * (WS - 1) + (s - 1)) * RT + rt) * MT + mt) * 64 + lane) * 4);
    }

    if (W8) {
        // dequantisation: accumulator register r of row tile rt belongs to weight row base+4*g4+r
        const float* ws = p.wscale + (int64_t)grp * p.wscale_gstride;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int rbase = (MODE == GEMM_GATEUP ? (rt == 0 ? ncol0 : p.N + ncol0) : ncol0 + 16 * rt) + 4 * g4;
            const int rmax = MODE == GEMM_GATEUP ? 2 * p.N - 1 : p.N - 1;
            float sc[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[r] = ws[min(rbase + r, rmax)];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[rt][mt][r] *= sc[r];
        }
    }
    smallm_epilogue<MODE, RT, MT>(p, acc, grp, m0, mcnt, ncol0, c, g4);
}

template <int MODE, int NWAVES, int WS, int RT, int MTMAX, bool W8>
__global__ __launch_bounds__(NWAVES * 64) void gemm_smallm_kernel(const GemmParams p_in) {
    GemmParams p = p_in;
    if (MODE == GEMM_PLAIN && p_in.ksplit > 1) {
        // K slice of this workgroup (whole 128-wide chunks; the host sizes ksplit so that no slice is empty)
        const int chunks = (p_in.K + SM_KC - 1) / SM_KC;
        const int per = (chunks + p_in.ksplit - 1) / p_in.ksplit;
        const int kb = blockIdx.z * per * SM_KC;
        p.x = p_in.x + kb;
        p.w = p_in.w + kb;
        p.K = min(per * SM_KC, p_in.K - kb);
        p.ypart = p_in.ypart + (int64_t)blockIdx.z * p_in.R * p_in.N;
    }
    constexpr int MB = 16 * MTMAX;
    constexpr int XBUF = WS * MB * SM_XROWB;  // bytes per x buffer (sized for MTMAX)
    static_assert(MODE != GEMM_GATEUP || RT == 2, "gate|up needs two row tiles");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* sh = reinterpret_cast<int*>(smem + 2 * XBUF);
    glm_locate_block(p.offs, p.G, p.R, MB, blockIdx.y, sh);
    __syncthreads();
    const int grp = sh[0];
    if (grp < 0) return;
    const int m0 = sh[1], mcnt = sh[2];
    const int nmt = (mcnt + 15) >> 4;
    if (nmt <= 1) smallm_body<MODE, NWAVES, WS, RT, 1, XBUF, W8>(p, smem, grp, m0, mcnt);
    else if (nmt <= 2 || MTMAX == 2) smallm_body<MODE, NWAVES, WS, RT, 2, XBUF, W8>(p, smem, grp, m0, mcnt);
    else smallm_body<MODE, NWAVES, WS, RT, MTMAX, XBUF, W8>(p, smem, grp, m0, mcnt);
}

// ==========================================================================================
// LDS-DMA variant of the same weight-streaming GEMM (bf16 weights, decode row counts).
// Structure of the paged-attention kernel: every wave owns a 2-stage LDS ring; one stage = the wave's 16*RT weight
// rows and the 16*MT activation rows of ONE 128-wide K chunk (256 B per row), both brought in by
// global_load_lds_dwordx4 (4 rows x 256 B per instruction, source-side chunk swizzle chunk ^ (row & 15) so the
// ds_read_b128 fragment reads are conflict free).  No workgroup barrier in the loop: a wave waits only for its own
// DMA with a counted vmcnt, reads the stage's fragments into registers and immediately refills the stage with the
// chunk two steps ahead.  The weight DMAs carry the nt hint (read once, by this CU only); the activation rows do not
// (every workgroup of the group re-reads them from L2).  Waves split K (WS) and rows (WR) as in the VGPR variant;
// K-slice partials are reduced through LDS after the loop.
typedef __attribute__((address_space(3))) void sm_lds_void;
typedef __attribute__((address_space(1))) const void sm_glb_void;
template <int AUX>
__device__ __forceinline__ void sm_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((sm_glb_void*)gsrc, (sm_lds_void*)lds_wave_base, 16, 0, AUX);
}
template <int OFF>
__device__ __forceinline__ void sm_lds_read_b128(u32x4_t& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
// wait for every LDS read; all fragment registers are named so that no consumer is scheduled above the wait
template <int NA, int NB>
__device__ __forceinline__ void sm_lds_wait(u32x4_t (&a)[NA][4], u32x4_t (&b)[NB][4]) {
    if constexpr (NA == 1 && NB == 1)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(b[0][0]), "+v"(b[0][1]),
                       "+v"(b[0][2]), "+v"(b[0][3])::"memory");
    else if constexpr (NA == 2 && NB == 1)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]),
                       "+v"(a[1][2]), "+v"(a[1][3]), "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[0][3])::"memory");
    else if constexpr (NA == 1 && NB == 2)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(b[0][0]), "+v"(b[0][1]),
                       "+v"(b[0][2]), "+v"(b[0][3]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]), "+v"(b[1][3])::"memory");
    else if constexpr (NA == 2 && NB == 2)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]),
                       "+v"(a[1][2]), "+v"(a[1][3]), "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[0][3]),
                       "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]), "+v"(b[1][3])::"memory");
    else if constexpr (NA == 4 && NB == 1)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]),
                       "+v"(a[1][2]), "+v"(a[1][3]), "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[2][2]), "+v"(a[2][3]),
                       "+v"(a[3][0]), "+v"(a[3][1]), "+v"(a[3][2]), "+v"(a[3][3]), "+v"(b[0][0]), "+v"(b[0][1]),
                       "+v"(b[0][2]), "+v"(b[0][3])::"memory");
    else {
        static_assert(NA == 4 && NB == 2, "fragment counts covered: 1/2/4 row tiles x 1/2 token tiles");
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]),
                       "+v"(a[1][2]), "+v"(a[1][3]), "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[2][2]), "+v"(a[2][3]),
                       "+v"(a[3][0]), "+v"(a[3][1]), "+v"(a[3][2]), "+v"(a[3][3]), "+v"(b[0][0]), "+v"(b[0][1]),
                       "+v"(b[0][2]), "+v"(b[0][3]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]), "+v"(b[1][3])::"memory");
    }
}

// counted wait with a run-time count (wave-uniform): leave the newest `n` vector-memory operations in flight
template <int LO, int HI>
__device__ __forceinline__ void sm_wait_leave(int n) {
    if constexpr (LO >= HI) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LO) : "memory");
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (n <= MID) sm_wait_leave<LO, MID>(n);
        else sm_wait_leave<MID + 1, HI>(n);
    }
}
#ifndef GLM_SM_XSKIP
#define GLM_SM_XSKIP 1  // do not fetch activation rows beyond the block's row count (their outputs are never stored)
#endif

// TPW > 1 (grouped modes, WS == 1): a wave walks TPW consecutive column tiles as ONE stream — the ring keeps
// prefetching across tile boundaries, so the start-up and drain bubbles of a short K (down: 11 chunks) are paid once
// per TPW tiles; results are stored at each tile's last chunk.
template <int MODE, int NWAVES, int WS, int RT, int MT, int TPW>
__device__ __forceinline__ void smallm_body_dma(const GemmParams& p, char* smem, int grp, int m0, int mcnt) {
    static_assert(SM_KC == 128 && (RT == 1 || RT == 2 || RT == 4) && MT <= 2, "one stage row = 256 B; 1/2/4 row tiles, at most 2 token tiles");
    static_assert(TPW == 1 || (WS == 1 && MODE != GEMM_PLAIN), "column-tile streams: grouped modes without a K split");
    constexpr int WR = NWAVES / WS;
    constexpr int WROWS = 16 * RT, XROWS = 16 * MT;
    constexpr int WT = WROWS * 256, STAGE = (WROWS + XROWS) * 256;
    constexpr int NW_DMA = WROWS / 4, NX_DMA = XROWS / 4, NDMA = NW_DMA + NX_DMA;
    constexpr int CPW = MODE == GEMM_GATEUP ? 8 * RT : 16 * RT;  // output columns per wave per tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ws = wid % WS, wr = wid / WS;
    const int c = lane & 15, g4 = lane >> 4;
    const int K = p.K;
    const int nchunks = (K + SM_KC - 1) / SM_KC;
    const int cps = (nchunks + WS - 1) / WS;  // chunks per K slice
    const int cbeg = ws * cps;
    const int nmine = max(0, min(cps, nchunks - cbeg));
    char* ring = smem + wid * 2 * STAGE;

    // ---- DMA sources: instruction i covers stage rows 4i .. 4i+3; lane -> row 4i + (lane >> 4), 16-B position lane & 15
    const int r4 = lane >> 4, pch = lane & 15;
    const int ncol_first = (blockIdx.x * TPW * WR + wr) * CPW;  // tile t adds t * WR * CPW columns
    const int64_t wstep = (int64_t)WR * CPW * p.ldw;            // weight rows advance with the columns (host: no clamping)
    const bf16_t* wg = p.w + (int64_t)grp * p.w_gstride;
    const bf16_t* wsrc[NW_DMA];
    const bf16_t* xsrc[NX_DMA];
    int wk[NW_DMA], xk[NX_DMA];  // element offset of this lane's 16-B chunk inside a K chunk
#pragma unroll
    for (int i = 0; i < NW_DMA; ++i) {
        const int row = 4 * i + r4;  // 0 .. WROWS-1
        int wi;
        if (MODE == GEMM_GATEUP)  // first half of the stage rows = gate columns, second half = the same up columns
            wi = (row < WROWS / 2 ? 0 : p.N) + min(ncol_first + row % (WROWS / 2), p.N - 1);
        else wi = min(ncol_first + row, p.N - 1);
        wsrc[i] = wg + (int64_t)wi * p.ldw;
        wk[i] = (pch ^ (row & 15)) << 3;
    }
#pragma unroll
    for (int i = 0; i < NX_DMA; ++i) {
        const int row = 4 * i + r4;  // 0 .. XROWS-1
        const int rr = row < mcnt ? row : 0;  // padding rows re-read row 0 (their columns are never stored)
        const int64_t srow = p.src ? (int64_t)p.src[m0 + rr] : (int64_t)(m0 + rr);
        xsrc[i] = p.x + srow * p.ldx;
        xk[i] = (pch ^ (row & 15)) << 3;
    }
    // per-row scale of the down projection, fetched before the stream starts (a load inside the loop would have to
    // wait behind every DMA in flight)
    float sc_pre[MT];
    bool scaled = false;
    if (TPW > 1 && MODE == GEMM_DOWN) {
        scaled = grp < p.n_scaled;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = 16 * mt + c;
            sc_pre[mt] = (scaled && row < mcnt) ? bf2f(p.scale[m0 + row]) : 1.0f;
        }
    }
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page);
    // activation DMA instructions that carry at least one real row (4 rows each); the rest of the tile keeps whatever
    // the LDS held: those token columns of the MFMAs are garbage that the epilogue never stores
    const int nx_eff = GLM_SM_XSKIP ? min(NX_DMA, (mcnt + 3) >> 2) : NX_DMA;
    const int ndma_eff = NW_DMA + nx_eff;
    // number of column tiles of this wave that exist
    int ntv = 1;
    if (TPW > 1) {
        ntv = 0;
#pragma unroll
        for (int t = 0; t < TPW; ++t) ntv += (ncol_first + t * WR * CPW) < p.N ? 1 : 0;
    }
    const int total = ntv * nmine;
    int is_t = 0, is_c = 0;  // (tile, chunk) of the next stage to issue
    auto issue = [&](int j) {
        char* st = ring + (j & 1) * STAGE;
        const int k0 = (cbeg + is_c) * SM_KC;
        const int64_t wo = TPW > 1 ? is_t * wstep : 0;
#pragma unroll
        for (int i = 0; i < NW_DMA; ++i)  // out-of-range K reads the zero page (address select)
            sm_glds16<2>(k0 + wk[i] < K ? wsrc[i] + wo + k0 + wk[i] : zero, st + i * 1024);
#pragma unroll
        for (int i = 0; i < NX_DMA; ++i)
            if (i < nx_eff) sm_glds16<0>(k0 + xk[i] < K ? xsrc[i] + k0 + xk[i] : zero, st + WT + i * 1024);
        if (++is_c == nmine) {
            is_c = 0;
            ++is_t;
        }
    };

    f32x4_t acc[RT][MT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[rt][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: lane (c, g4) reads row +c, chunk (4ks + g4) ^ c; row tiles / token tiles as immediates
    unsigned rel[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rel[ks] = c * 256 + (((4 * ks + g4) ^ c) << 4);

    if (total > 0) issue(0);
    if (total > 1) issue(1);
    const unsigned rbase = (unsigned)(uintptr_t)(const sm_lds_void*)ring;
    int cs_t = 0, cs_c = 0;  // (tile, chunk) being consumed
    for (int it = 0; it < total; ++it) {
        if (it + 1 < total) sm_wait_leave<NW_DMA + 1, NDMA>(ndma_eff);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned sb = rbase + (it & 1) * STAGE;
        u32x4_t af[RT][4], bfr[MT][4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            sm_lds_read_b128<0>(af[0][ks], sb + rel[ks]);
            if (RT > 1) sm_lds_read_b128<4096>(af[1 % RT][ks], sb + rel[ks]);
            if (RT > 2) {
                sm_lds_read_b128<8192>(af[2 % RT][ks], sb + rel[ks]);
                sm_lds_read_b128<12288>(af[3 % RT][ks], sb + rel[ks]);
            }
            sm_lds_read_b128<WT>(bfr[0][ks], sb + rel[ks]);
            if (MT > 1) sm_lds_read_b128<WT + 4096>(bfr[MT - 1][ks], sb + rel[ks]);
        }
        sm_lds_wait<RT, MT>(af, bfr);
        __builtin_amdgcn_sched_barrier(0);
        if (it + 2 < total) issue(it + 2);  // the stage is free again
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[rt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[rt][ks]),
                                                                         __builtin_bit_cast(bf16x8_t, bfr[mt][ks]),
                                                                         acc[rt][mt], 0, 0, 0);
        if (TPW > 1 && ++cs_c == nmine) {
            // last chunk of a column tile: store it (stores only; the counted vmcnt stays valid because loads return
            // in order among themselves) and start the next tile from zero
            const int ncol0 = ncol_first + cs_t * WR * CPW;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int row = 16 * mt + c;
                if (row >= mcnt) continue;
                bf16_t* yrow = p.y + (int64_t)(m0 + row) * p.ldy;
                if (MODE == GEMM_GATEUP) {
#pragma unroll
                    for (int j = 0; j < RT / 2; ++j) {
                        const int n = ncol0 + 16 * j + 4 * g4;
                        float o[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            o[r] = rbf(glm_silu(rbf(acc[j][mt][r]))) * rbf(acc[RT / 2 + j][mt][r]);
                        const u32x2_t v = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
                        *reinterpret_cast<u32x2_t*>(yrow + n) = v;
                    }
                } else {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const int n = ncol0 + 16 * rt + 4 * g4;
                        float o[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float a = acc[rt][mt][r];
                            o[r] = scaled ? rbf(a) * sc_pre[mt] : a;
                        }
                        const u32x2_t v = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
                        *reinterpret_cast<u32x2_t*>(yrow + n) = v;
                    }
                }
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[rt][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            cs_c = 0;
            ++cs_t;
        }
    }
    if (TPW > 1) return;

    const int ncol0 = ncol_first;
    // ---- K-slice reduction through LDS (WS > 1): slice 0 accumulates the others ----
    if (WS > 1) {
        __syncthreads();  // every wave is done with its ring before the scratch overwrites wave 0's
        float* red = reinterpret_cast<float*>(smem);
        if (ws != 0) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    *reinterpret_cast<f32x4_t*>(red + ((((wr * (WS - 1) + (ws - 1)) * RT + rt) * MT + mt) * 64 + lane) * 4) =
                        acc[rt][mt];
        }
        __syncthreads();
        if (ws != 0) return;
#pragma unroll
        for (int s = 1; s < WS; ++s)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[rt][mt] += *reinterpret_cast<const f32x4_t*>(
                        red + ((((wr * (WS - 1) + (s - 1)) * RT + rt) * MT + mt) * 64 + lane) * 4);
    }
    smallm_epilogue<MODE, RT, MT>(p, acc, grp, m0, mcnt, ncol0, c, g4);
}

template <int MODE, int NWAVES, int WS, int RT, int MTMAX, int TPW>
__global__ __launch_bounds__(NWAVES * 64) void gemm_smallm_dma_kernel(const GemmParams p_in) {
    GemmParams p = p_in;
    if (MODE == GEMM_PLAIN && p_in.ksplit > 1) {
        const int chunks = (p_in.K + SM_KC - 1) / SM_KC;
        const int per = (chunks + p_in.ksplit - 1) / p_in.ksplit;
        const int kb = blockIdx.z * per * SM_KC;
        p.x = p_in.x + kb;
        p.w = p_in.w + kb;
        p.K = min(per * SM_KC, p_in.K - kb);
        p.ypart = p_in.ypart + (int64_t)blockIdx.z * p_in.R * p_in.N;
    }
    constexpr int MB = 16 * MTMAX;
    constexpr int RING = NWAVES * 2 * (16 * RT + 16 * MTMAX) * 256;
    static_assert(MODE != GEMM_GATEUP || RT == 2 || RT == 4, "gate|up: gate and up tiles come in pairs");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* sh = reinterpret_cast<int*>(smem + RING);
    glm_locate_block(p.offs, p.G, p.R, MB, blockIdx.y, sh);
    __syncthreads();
    const int grp = sh[0];
    if (grp < 0) return;
    const int m0 = sh[1], mcnt = sh[2];
    if (mcnt <= 16) smallm_body_dma<MODE, NWAVES, WS, RT, 1, TPW>(p, smem, grp, m0, mcnt);
    else smallm_body_dma<MODE, NWAVES, WS, RT, (MTMAX < 2 ? MTMAX : 2), TPW>(p, smem, grp, m0, mcnt);
}

// ==========================================================================================
// Shared-activation LDS-DMA body ("sx"): the NW waves of a workgroup own different weight rows of one column tile and
// SHARE one activation tile per K chunk.  With private rings (smallm_body_dma) a wave moves as many activation bytes
// L2 -> LDS as weight bytes whenever a row block holds as many rows as the wave has weight rows — with 1-byte weights
// twice as many, which is what capped the fp8 DMA variant of r01 (239 / 111 us).  Here the activation rows of a chunk
// are fetched once per workgroup (each wave DMAs 1/NW of them), i.e. 12-25 % on top of the weight stream.
//   * 3-stage ring shared by the workgroup, stage = [wave 0 rows | ... | wave NW-1 rows | activation rows]; one chunk =
//     256 B per weight row: 128 k of bf16 or 256 k of fp8 (activation rows then 512 B).
//   * ONE workgroup barrier per chunk: a wave waits for its own DMA of chunk `it` (counted vmcnt, chunk it+1 stays in
//     flight), then the barrier makes every wave's share of the activation tile visible AND proves that everyone is done
//     reading chunk it-1, whose stage is refilled with chunk it+2 right after the fragment reads.
//   * fp8 weights are converted to bf16 fragments in registers exactly as in smallm_body (one 16-byte LDS read = two
//     k-steps of a lane), the row scale multiplies the fp32 accumulator.
//   * NST = 2 (GLM_SM_SX_STAGES): two stages and a SECOND barrier per chunk (after the fragment reads, before the
//     refill of the stage just read): 80 KiB per workgroup, i.e. two workgroups = 8 waves per CU instead of 4.
template <int MODE, int NW, int RT, int MT, bool W8, int NST>
__device__ __forceinline__ void smallm_body_sx(const GemmParams& p, char* smem, int grp, int m0, int mcnt) {
    static_assert((RT == 1 || RT == 2 || RT == 4) && MT <= 2, "1/2/4 row tiles, at most 2 token tiles");
    constexpr int KC = W8 ? 256 : 128;            // k per chunk: 256 B per weight row
    constexpr int NKS = KC / 32;                  // MFMA k-steps per chunk
    constexpr int XROWB = KC * 2;                 // activation row bytes per chunk
    constexpr int WROWS = 16 * RT, XROWS = 16 * MT;
    constexpr int WWAVE = WROWS * 256;            // one wave's weight bytes per stage
    constexpr int XT = XROWS * XROWB;             // shared activation bytes per stage
    constexpr int STAGE = NW * WWAVE + XT;
    constexpr int NW_DMA = WROWS / 4;             // weight DMA instructions per wave per chunk (4 rows x 256 B each)
    constexpr int NX_ALL = XT / 1024, NX_DMA = NX_ALL / NW;  // activation DMA instructions: all, per wave
    static_assert(NX_ALL % NW == 0, "activation tile must split evenly over the waves");
    constexpr int NDMA = NW_DMA + NX_DMA;
    constexpr int XRPI = 1024 / XROWB;            // activation rows per DMA instruction (4 or 2)
    constexpr int XCPR = XROWB / 16;              // 16-B chunks per activation row (16 or 32)
    constexpr int CPW = MODE == GEMM_GATEUP ? 8 * RT : 16 * RT;  // output columns per wave
    constexpr int WB = W8 ? 1 : 2;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g4 = lane >> 4;
    const int K = p.K;
    const int nchunks = (K + KC - 1) / KC;
    const int ncol0 = (blockIdx.x * NW + wid) * CPW;

    // ---- DMA sources ----
    const int r4 = lane >> 4, pch = lane & 15;
    const char* wg = reinterpret_cast<const char*>(p.w) + (int64_t)grp * p.w_gstride * WB;
    const char* wsrc[NW_DMA];
    int wk[NW_DMA];  // byte offset of this lane's 16-B piece inside a chunk row
#pragma unroll
    for (int i = 0; i < NW_DMA; ++i) {
        const int row = 4 * i + r4;
        int wi;
        if (MODE == GEMM_GATEUP) wi = (row < WROWS / 2 ? 0 : p.N) + min(ncol0 + row % (WROWS / 2), p.N - 1);
        else wi = min(ncol0 + row, p.N - 1);
        wsrc[i] = wg + (int64_t)wi * p.ldw * WB;
        wk[i] = (pch ^ (row & 15)) << 4;
    }
    const char* xsrc[NX_DMA];
    int xk[NX_DMA];
#pragma unroll
    for (int i = 0; i < NX_DMA; ++i) {
        const int inst = wid * NX_DMA + i;                 // this wave's share of the activation tile
        const int row = inst * XRPI + lane / XCPR;         // 0 .. XROWS-1
        const int pos = lane % XCPR;
        const int rr = row < mcnt ? row : 0;               // padding rows re-read row 0 (never stored)
        const int64_t srow = p.src ? (int64_t)p.src[m0 + rr] : (int64_t)(m0 + rr);
        xsrc[i] = reinterpret_cast<const char*>(p.x + srow * p.ldx);
        xk[i] = (pos ^ (row & 15)) << 4;                   // XOR of the low 4 bits: stays inside the row
    }
    const char* zero = reinterpret_cast<const char*>(g_zero_page);
    auto issue = [&](int ci) {
        char* st = smem + (ci % NST) * STAGE;
        const int k0 = ci * KC;
#pragma unroll
        for (int i = 0; i < NW_DMA; ++i) {
            const int kb = k0 * WB + wk[i];  // byte offset in the weight row; 16 bytes = 16 / WB elements
            sm_glds16<2>(kb / WB < K ? wsrc[i] + kb : zero, st + wid * WWAVE + i * 1024);
        }
#pragma unroll
        for (int i = 0; i < NX_DMA; ++i) {
            const int kb = k0 * 2 + xk[i];
            sm_glds16<0>(kb / 2 < K ? xsrc[i] + kb : zero, st + NW * WWAVE + (wid * NX_DMA + i) * 1024);
        }
    };

    f32x4_t acc[RT][MT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[rt][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets relative to the stage base.  weights: lane (c, g4) reads row 16 rt + c, 16-B piece
    // (4 q + g4) ^ c for q = 0..3 (bf16: k-step q; fp8: k-steps 2q, 2q+1).  activations: row 16 mt + c, piece xp ^ c.
    unsigned wrel[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) wrel[q] = wid * WWAVE + c * 256 + (((4 * q + g4) ^ c) << 4);
    unsigned xrel[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int xp = W8 ? 8 * (ks >> 1) + 2 * g4 + (ks & 1) : 4 * ks + g4;
        xrel[ks] = NW * WWAVE + c * XROWB + ((xp ^ c) << 4);
    }
    const unsigned rbase = (unsigned)(uintptr_t)(const sm_lds_void*)smem;

    if (nchunks > 0) issue(0);
    if (nchunks > 1) issue(1);
    for (int it = 0; it < nchunks; ++it) {
        if (it + 1 < nchunks) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned sb = rbase + (it % NST) * STAGE;
        u32x4_t af[RT][4], bfr[MT][NKS];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sm_lds_read_b128<0>(af[0][q], sb + wrel[q]);
            if (RT > 1) sm_lds_read_b128<4096>(af[1 % RT][q], sb + wrel[q]);
            if (RT > 2) {
                sm_lds_read_b128<8192>(af[2 % RT][q], sb + wrel[q]);
                sm_lds_read_b128<12288>(af[3 % RT][q], sb + wrel[q]);
            }
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            sm_lds_read_b128<0>(bfr[0][ks], sb + xrel[ks]);
            if (MT > 1) sm_lds_read_b128<16 * XROWB>(bfr[MT - 1][ks], sb + xrel[ks]);
        }
        // wait for every LDS read: all destinations named so that no consumer moves above the wait
        if constexpr (RT == 2 && MT == 1 && NKS == 8)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(af[0][3]), "+v"(af[1][0]),
                         "+v"(af[1][1]), "+v"(af[1][2]), "+v"(af[1][3]), "+v"(bfr[0][0]), "+v"(bfr[0][1]), "+v"(bfr[0][2]),
                         "+v"(bfr[0][3]), "+v"(bfr[0][4]), "+v"(bfr[0][5]), "+v"(bfr[0][6]), "+v"(bfr[0][7])::"memory");
        else if constexpr (RT == 2 && MT == 2 && NKS == 8)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(af[0][3]), "+v"(af[1][0]),
                         "+v"(af[1][1]), "+v"(af[1][2]), "+v"(af[1][3]), "+v"(bfr[0][0]), "+v"(bfr[0][1]), "+v"(bfr[0][2]),
                         "+v"(bfr[0][3]), "+v"(bfr[0][4]), "+v"(bfr[0][5]), "+v"(bfr[0][6]), "+v"(bfr[0][7]), "+v"(bfr[1][0]),
                         "+v"(bfr[1][1]), "+v"(bfr[1][2]), "+v"(bfr[1][3]), "+v"(bfr[1][4]), "+v"(bfr[1][5]), "+v"(bfr[1][6]),
                         "+v"(bfr[1][7])::"memory");
        else if constexpr (RT == 2 && MT == 1 && NKS == 4)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(af[0][3]), "+v"(af[1][0]),
                         "+v"(af[1][1]), "+v"(af[1][2]), "+v"(af[1][3]), "+v"(bfr[0][0]), "+v"(bfr[0][1]), "+v"(bfr[0][2]),
                         "+v"(bfr[0][3])::"memory");
        else {
            static_assert(RT == 2 && MT == 2 && NKS == 4, "fragment sets covered: 2 row tiles x 1/2 token tiles, bf16 or fp8");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(af[0][3]), "+v"(af[1][0]),
                         "+v"(af[1][1]), "+v"(af[1][2]), "+v"(af[1][3]), "+v"(bfr[0][0]), "+v"(bfr[0][1]), "+v"(bfr[0][2]),
                         "+v"(bfr[0][3]), "+v"(bfr[1][0]), "+v"(bfr[1][1]), "+v"(bfr[1][2]), "+v"(bfr[1][3])::"memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        if (NST == 2) __builtin_amdgcn_s_barrier();  // every wave holds its fragments of this stage: it may be refilled
        // NST 3: stage (it+2)%3 == (it-1)%3, everyone is past it (barrier at the top); NST 2: the stage just read
        if (it + 2 < nchunks) issue(it + 2);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                bf16x8_t a;
                if (!W8) {
                    a = __builtin_bit_cast(bf16x8_t, af[rt][ks]);
                } else {  // bytes [8 (ks & 1), +8) of piece ks >> 1: exact fp8 -> bf16
                    const u32x4_t w = af[rt][ks >> 1];
                    const unsigned d0 = w[2 * (ks & 1)], d1 = w[2 * (ks & 1) + 1];
                    a = __builtin_bit_cast(bf16x8_t, sm_fp8x8_to_bf16x8(d0, d1));
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[rt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8_t, bfr[mt][ks]),
                                                                         acc[rt][mt], 0, 0, 0);
            }
    }
    if (W8) {
        const float* ws = p.wscale + (int64_t)grp * p.wscale_gstride;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int rb = (MODE == GEMM_GATEUP ? (rt < RT / 2 ? ncol0 + 16 * rt : p.N + ncol0 + 16 * (rt - RT / 2)) : ncol0 + 16 * rt) + 4 * g4;
            const int rmax = MODE == GEMM_GATEUP ? 2 * p.N - 1 : p.N - 1;
            float sc[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[r] = ws[min(rb + r, rmax)];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[rt][mt][r] *= sc[r];
        }
    }
    smallm_epilogue<MODE, RT, MT>(p, acc, grp, m0, mcnt, ncol0, c, g4);
}

#ifndef GLM_SM_SX_STAGES
#define GLM_SM_SX_STAGES 2
#endif
template <int MODE, int NW, int RT, int MTMAX, bool W8>
__global__ __launch_bounds__(NW * 64) void gemm_smallm_sx_kernel(const GemmParams p) {
    constexpr int MB = 16 * MTMAX;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // the block search borrows the start of the ring (80 KiB x 2 workgroups leaves the CU no byte to spare)
    int* sh = reinterpret_cast<int*>(smem);
    glm_locate_block(p.offs, p.G, p.R, MB, blockIdx.y, sh);
    __syncthreads();
    const int grp = sh[0];
    const int m0 = sh[1], mcnt = sh[2];
    __syncthreads();  // everyone has read sh before the first DMA lands on it
    if (grp < 0) return;
    if (mcnt <= 16) smallm_body_sx<MODE, NW, RT, 1, W8, GLM_SM_SX_STAGES>(p, smem, grp, m0, mcnt);
    else smallm_body_sx<MODE, NW, RT, (MTMAX < 2 ? MTMAX : 2), W8, GLM_SM_SX_STAGES>(p, smem, grp, m0, mcnt);
}

template <int MODE, int NW, int RT, int MTMAX, bool W8>
int launch_cfg_sx(const GemmParams& p, hipStream_t st) {
    static_assert(MTMAX <= 2, "the shared-activation body stages at most 32 token rows");
    constexpr int MB = 16 * MTMAX;
    constexpr int RING = GLM_SM_SX_STAGES * (NW * 16 * RT * 256 + 16 * MTMAX * (W8 ? 512 : 256));
    static_assert(RING <= 160 * 1024, "the ring must fit the CU's 160 KiB of LDS");
    constexpr int LDS = RING;
    const int cols_per_wg = MODE == GEMM_GATEUP ? NW * 8 * RT : NW * RT * 16;
    const unsigned gx = (unsigned)((p.N + cols_per_wg - 1) / cols_per_wg);
    const int64_t combos = p.offs ? (p.R / MB + p.G) : (p.R + MB - 1) / MB;
    if (combos <= 0 || gx == 0) return GLM_OK;
    GLM_CHECK_ARG(combos <= 65535, GLM_ERR_UNSUPPORTED, "gemm_smallm(sx): too many row blocks (%ld)", (long)combos);
    auto kern = gemm_smallm_sx_kernel<MODE, NW, RT, MTMAX, W8>;
    GLM_SET_MAX_LDS(kern, LDS, "gemm_smallm(sx)");
    hipLaunchKernelGGL(kern, dim3(gx, (unsigned)combos, 1), dim3(NW * 64), LDS, st, p);
    GLM_CHECK_LAUNCH("gemm_smallm(sx)");
    return GLM_OK;
}

template <int MODE, int NWAVES, int WS, int RT, int MTMAX, int TPW = 1>
int launch_cfg_dma(const GemmParams& p, hipStream_t st) {
    static_assert(MTMAX <= 2, "the DMA variant stages at most 32 token rows");
    if (TPW > 1) {  // column-tile streams need whole tiles everywhere, else the plain one-tile kernel
        const int step = (MODE == GEMM_GATEUP ? (NWAVES / WS) * 8 * RT : (NWAVES / WS) * RT * 16) * TPW;
        if (p.N % step != 0) return launch_cfg_dma<MODE, NWAVES, WS, RT, MTMAX, 1>(p, st);
    }
    constexpr int WR = NWAVES / WS;
    constexpr int MB = 16 * MTMAX;
    constexpr int RING = NWAVES * 2 * (16 * RT + 16 * MTMAX) * 256;
    constexpr int REDB = WS > 1 ? WR * (WS - 1) * RT * MTMAX * 64 * 16 : 0;
    static_assert(REDB <= RING, "reduction scratch must fit in the rings");
    static_assert(RING + 16 <= 160 * 1024, "the rings of all waves must fit the CU's 160 KiB of LDS");
    constexpr int LDS = RING + 16;
    const int cols_per_wg = (MODE == GEMM_GATEUP ? WR * 8 * RT : WR * RT * 16) * TPW;
    const unsigned gx = (unsigned)((p.N + cols_per_wg - 1) / cols_per_wg);
    const int64_t combos = p.offs ? (p.R / MB + p.G) : (p.R + MB - 1) / MB;
    if (combos <= 0 || gx == 0) return GLM_OK;
    GLM_CHECK_ARG(combos <= 65535, GLM_ERR_UNSUPPORTED, "gemm_smallm: too many row blocks (%ld)", (long)combos);
    auto kern = gemm_smallm_dma_kernel<MODE, NWAVES, WS, RT, MTMAX, TPW>;
    if (LDS > 64 * 1024) GLM_SET_MAX_LDS(kern, LDS, "gemm_smallm(dma)");
    hipLaunchKernelGGL(kern, dim3(gx, (unsigned)combos, (unsigned)(p.ksplit > 1 ? p.ksplit : 1)), dim3(NWAVES * 64),
                       LDS, st, p);
    GLM_CHECK_LAUNCH("gemm_smallm(dma)");
    return GLM_OK;
}

template <int MODE, int NWAVES, int WS, int RT, int MTMAX, bool W8 = false>
int launch_cfg(const GemmParams& p, hipStream_t st) {
    constexpr int WR = NWAVES / WS;
    constexpr int MB = 16 * MTMAX;
    constexpr int XBUF = WS * MB * SM_XROWB;
    constexpr int REDB = WS > 1 ? WR * (WS - 1) * RT * MTMAX * 64 * 16 : 0;
    constexpr int LDS = (2 * XBUF > REDB ? 2 * XBUF : REDB) + 16;
    static_assert(REDB <= 2 * XBUF, "reduction scratch must fit in the x buffers");
    const int cols_per_wg = MODE == GEMM_GATEUP ? WR * 16 : WR * RT * 16;
    const unsigned gx = (unsigned)((p.N + cols_per_wg - 1) / cols_per_wg);
    const int64_t combos = p.offs ? (p.R / MB + p.G) : (p.R + MB - 1) / MB;
    if (combos <= 0 || gx == 0) return GLM_OK;
    GLM_CHECK_ARG(combos <= 65535, GLM_ERR_UNSUPPORTED, "gemm_smallm: too many row blocks (%ld)", (long)combos);
    auto kern = gemm_smallm_kernel<MODE, NWAVES, WS, RT, MTMAX, W8>;
    if (LDS > 64 * 1024) GLM_SET_MAX_LDS(kern, LDS, "gemm_smallm");
    hipLaunchKernelGGL(kern, dim3(gx, (unsigned)combos, (unsigned)(p.ksplit > 1 ? p.ksplit : 1)), dim3(NWAVES * 64),
                       LDS, st, p);
    GLM_CHECK_LAUNCH("gemm_smallm");
    return GLM_OK;
}

}  // namespace

int glm_launch_gemm_smallm(int mode, const GemmParams& p, hipStream_t st) {
    GLM_CHECK_ARG(p.K % 8 == 0 && p.N % 4 == 0 && p.ldx % 8 == 0 && p.ldy % 4 == 0 && p.ldw % 8 == 0 &&
                      glm_aligned16(p.x) && glm_aligned16(p.w) && (((uintptr_t)p.y) & 7) == 0,
                  GLM_ERR_UNSUPPORTED,
                  "gemm_smallm: needs K%%8==0, N%%4==0, ldx/ldw%%8==0, ldy%%4==0, 16B-aligned x/W (K=%d N=%d)",
                  p.K, p.N);
    const int64_t Rs = p.R_hint > 0 ? p.R_hint : p.R;  // the configuration is chosen by the expected rows, the grid by p.R
    // grouped experts: 32-row blocks (at decode batch sizes only the shared expert ever fills one;
    // a group with more rows re-reads its weights once per extra block)
    // (waves, K-split waves, row tiles per wave, max 16-row activation tiles).  Same-box sweep at the C3 decode shape
    // (tools/sweep_grouped.sh, two rounds): gate|up 4,1,2,2: 272 us; 8,4,2,2: 256; 4,2,2,2: 251; 4,4,2,2: 245-249.
    // down 4,1,2,2: 130 us; 8,1,1,2: 122; 2,1,1,2: 120; 4,2,1,2: 118-120.  Splitting K over the waves of a small
    // workgroup (16 act columns of gate|up, 32 columns of down) beats one long stream per wave.
#ifndef GLM_SM_CFG_GATEUP
#define GLM_SM_CFG_GATEUP 4, 4, 2, 2
#endif
#ifndef GLM_SM_CFG_DOWN
#define GLM_SM_CFG_DOWN 4, 2, 1, 2
#endif
    // a single sequence (k + 1 rows, as many active experts): the long unsplit streams win (42 vs 49 us); from two
    // sequences on the split config is ahead (74 vs 87 us at bs=2 ... 224 vs 231 us at bs=16; tools/sweep_batch.sh)
    // LDS-DMA variant where it measured faster (tools/sweep_all.sh, C3 decode shape): gate|up 2 waves x 2-way K split
    // 244 -> 214-218 us (6.5 TB/s); QKV 27.0 -> 25.6 us; o_proj partial sums 21.2 -> 19.7 us; down: no gain (short K).
    // -DGLM_SM_NO_DMA builds the all-VGPR dispatch for A/B runs.
#if !defined(GLM_SM_NO_DMA) && !defined(GLM_SM_DMA_GATEUP)
#define GLM_SM_DMA_GATEUP 2, 2, 2, 2
#endif
#if !defined(GLM_SM_NO_DMA) && !defined(GLM_SM_DMA_MID)
#define GLM_SM_DMA_MID 2, 1, 2, 2  // 23.7 us (4,2,2,2: 25.6; 1,1,2,2: 24.4; 4,1,2,2: 39.8)
#endif
#ifndef GLM_SM_DMA_MIN_ROWS
#define GLM_SM_DMA_MIN_ROWS 17  // total rows from which the grouped projections take the LDS-DMA body
#endif
#ifdef GLM_SM_DMA_GATEUP
    if (mode == GEMM_GATEUP && Rs >= GLM_SM_DMA_MIN_ROWS) return launch_cfg_dma<GEMM_GATEUP, GLM_SM_DMA_GATEUP>(p, st);
#endif
#if !defined(GLM_SM_NO_DMA) && !defined(GLM_SM_DMA_DOWN)
#define GLM_SM_DMA_DOWN 1, 1, 2, 2  // one-wave workgroups (32 KiB of LDS each): 122 -> 115 us; 4,1,2,2 and 2,1,2,2: 116-117
#endif
#ifdef GLM_SM_DMA_DOWN
#ifndef GLM_SM_NO_DMA
    // one group with a long reduction (the dense layer's down projection, 4096 x 10944): too few column blocks for
    // one wave each (47.7 us = 1.9 TB/s) — split K over the waves of the workgroup instead
#ifndef GLM_SM_DMA_DOWN_DENSE
#define GLM_SM_DMA_DOWN_DENSE 4, 4, 1, 2
#endif
    if (mode == GEMM_DOWN && p.G == 1 && Rs >= GLM_SM_DMA_MIN_ROWS && p.K >= 4096) return launch_cfg_dma<GEMM_DOWN, GLM_SM_DMA_DOWN_DENSE>(p, st);
#endif
    // a K of one or two chunks (TP=8 shard: 176) leaves the ring nothing to overlap: 37 vs 31 us for the VGPR body
    // few rows per group (bs=32: 4.4 on average): 16-row blocks in two-wave workgroups — 24 KiB of LDS per wave, six waves
    // per CU instead of five, and a third less activation DMA; a 32-row group (the shared expert) then reads its weights
    // twice (+1.6 % bytes).  r02 sweep at C3: 114.7 -> 108.8 us (1,1,2,1: 110.5).  Larger batches keep 32-row blocks.
#ifndef GLM_SM_DMA_DOWN_FEWROWS
#define GLM_SM_DMA_DOWN_FEWROWS 2, 1, 2, 1
#endif
    if (mode == GEMM_DOWN && Rs >= GLM_SM_DMA_MIN_ROWS && p.K >= 512 && p.G > 1 && Rs <= 6 * (int64_t)p.G)
        return launch_cfg_dma<GEMM_DOWN, GLM_SM_DMA_DOWN_FEWROWS>(p, st);
    if (mode == GEMM_DOWN && Rs >= GLM_SM_DMA_MIN_ROWS && p.K >= 512) return launch_cfg_dma<GEMM_DOWN, GLM_SM_DMA_DOWN>(p, st);
#endif
#ifndef GLM_SM_CFG_GATEUP_R16
#define GLM_SM_CFG_GATEUP_R16 4, 1, 2, 2
#endif
#ifndef GLM_SM_CFG_DOWN_R16
#define GLM_SM_CFG_DOWN_R16 2, 1, 2, 2  // single sequence (9 groups): 25.7 us (4,2,1,2: 27.8; 4,1,1,2: 25.9; 8,1,1,2: 30.3; r02 sweep)
#endif
    if (mode == GEMM_GATEUP && Rs <= 16) return launch_cfg<GEMM_GATEUP, GLM_SM_CFG_GATEUP_R16>(p, st);
    if (mode == GEMM_DOWN && Rs <= 16) return launch_cfg<GEMM_DOWN, GLM_SM_CFG_DOWN_R16>(p, st);
    if (mode == GEMM_GATEUP) return launch_cfg<GEMM_GATEUP, GLM_SM_CFG_GATEUP>(p, st);
    if (mode == GEMM_DOWN) return launch_cfg<GEMM_DOWN, GLM_SM_CFG_DOWN>(p, st);
    // dense projection: choose the K-split so that the grid covers the chip
    const int64_t rows = p.offs ? 64 : p.R;
    const int tiles32 = (p.N + 31) / 32;
#if GLM_SM_KS != 4  // experimental chunk widths: the K-split configs would exceed the LDS
    return launch_cfg<GEMM_PLAIN, 4, 1, 2, 4>(p, st);
#endif
#ifndef GLM_SM_NO_DMA
    // very wide outputs at decode batch sizes (lm_head: 151552 x 4096): the DMA body, 2-wave workgroups
    if (p.G <= 1 && rows <= 32 && tiles32 >= 1024 && p.K >= 512) return launch_cfg_dma<GEMM_PLAIN, 2, 1, 2, 2>(p, st);
#endif
    if (p.G > 1 || rows > 32 || tiles32 >= 1024) return launch_cfg<GEMM_PLAIN, 4, 1, 2, 4>(p, st);
    // (waves, K-split waves, row tiles per wave, max 16-row activation tiles); overridable for A/B builds
#ifndef GLM_SM_CFG_MID
#define GLM_SM_CFG_MID 8, 2, 1, 2  // QKV at M=32: 26.9 us (4,4,2,2: 33.2; 4,2,2,2: 28.9; 8,4,2,2: 29.5)
#endif
#ifndef GLM_SM_CFG_SMALL
#define GLM_SM_CFG_SMALL 8, 8, 1, 2
#endif
#ifdef GLM_SM_DMA_MID
    if (tiles32 >= 384) return launch_cfg_dma<GEMM_PLAIN, GLM_SM_DMA_MID>(p, st);
#endif
    if (tiles32 >= 384) return launch_cfg<GEMM_PLAIN, GLM_SM_CFG_MID>(p, st);
    return launch_cfg<GEMM_PLAIN, GLM_SM_CFG_SMALL>(p, st);
}

int glm_smallm_partial_slices(int K, int want) {
    const int chunks = (K + SM_KC - 1) / SM_KC;
    if (want < 1) want = 1;
    if (want > chunks) want = chunks;
    const int per = (chunks + want - 1) / want;
    return (chunks + per - 1) / per;  // no empty slice
}

int glm_launch_gemm_smallm_partial(const GemmParams& p, hipStream_t st) {
    GLM_CHECK_ARG(p.K % 8 == 0 && p.N % 4 == 0 && p.ldx % 8 == 0 && p.ldw % 8 == 0 && glm_aligned16(p.x) &&
                      glm_aligned16(p.w) && glm_aligned16(p.ypart) && p.R <= 32 && p.offs == nullptr && p.ksplit > 1 &&
                      p.ksplit == glm_smallm_partial_slices(p.K, p.ksplit),
                  GLM_ERR_UNSUPPORTED, "gemm_smallm(partial): needs M<=32, K%%8==0, N%%4==0, a slice count from "
                  "glm_smallm_partial_slices (M=%ld K=%d N=%d ksplit=%d)", (long)p.R, p.K, p.N, p.ksplit);
#ifndef GLM_SM_CFG_PARTIAL
#define GLM_SM_CFG_PARTIAL 4, 1, 1, 2
#endif
#if !defined(GLM_SM_NO_DMA) && !defined(GLM_SM_DMA_PARTIAL)
#define GLM_SM_DMA_PARTIAL 2, 1, 4, 2  // o_proj slices at M=32: 19.0 us (4,1,2,2: 20.1; 1,1,4,2: 21.8; 2,1,2,2: 21.2; r02 sweep)
#endif
#ifdef GLM_SM_DMA_PARTIAL
    return launch_cfg_dma<GEMM_PLAIN, GLM_SM_DMA_PARTIAL>(p, st);
#endif
    return launch_cfg<GEMM_PLAIN, GLM_SM_CFG_PARTIAL>(p, st);
}

int glm_launch_gemm_smallm_fp8w(int mode, const GemmParams& p, hipStream_t st) {
    GLM_CHECK_ARG(p.K % 16 == 0 && p.N % 4 == 0 && p.ldx % 8 == 0 && p.ldy % 4 == 0 && p.ldw % 16 == 0 &&
                      p.w_gstride % 16 == 0 && glm_aligned16(p.x) && glm_aligned16(p.w) &&
                      (((uintptr_t)p.y) & 7) == 0 && p.wscale != nullptr,
                  GLM_ERR_UNSUPPORTED,
                  "gemm_smallm(fp8 weights): needs K%%16==0, N%%4==0, ldw%%16==0, 16B-aligned x/W, scales (K=%d N=%d)",
                  p.K, p.N);
    // fp8 weights (half the bytes per row): sweep (tools/sweep_fp8.sh) gate|up 4,1,2,2: 150 us; 4,4,2,2: 198; 4,2,2,2: 141.
    // down 4,1,2,2: 74.5 us; 4,2,1,2: 85; 8,1,1,2: 68
#ifndef GLM_SM_CFG_GATEUP_W8
#define GLM_SM_CFG_GATEUP_W8 4, 2, 2, 2
#endif
#ifndef GLM_SM_CFG_DOWN_W8
#define GLM_SM_CFG_DOWN_W8 8, 1, 1, 2
#endif
    // shared-activation LDS-DMA body (4 waves, 64 act / 128 down columns per workgroup) from two sequences on;
    // a single sequence (k + 1 rows in as many groups) keeps the VGPR body.  -DGLM_SM_FP8_NO_SX: VGPR body everywhere.
#ifndef GLM_SM_FP8_NO_SX
    // 16-row blocks (a 32-row group such as the shared expert at bs=32 re-reads its weights once): 80 KiB per workgroup
    // r02, 12 layers at bs=32 (same box): gate|up 144.7 us (best 135.0) vs 149.0 (142.0) for the VGPR body; down 74.2 vs
    // 71.3 with the two-step fp8 -> bf16 conversion; with the packed convert (sm_fp8x8_to_bf16x8) the shared-activation
    // body wins for both (tools/fp8_bench.py, same box: gate|up 122.5 -> 111.5 us, down 67.5 (VGPR body) -> 60.1 us);
    // -DGLM_SM_FP8_NO_SX_DOWN keeps the VGPR body for down; 3 stages at one workgroup per CU measured 165.7 / 91.1
#ifndef GLM_SX_GATEUP_NW
#define GLM_SX_GATEUP_NW 4
#endif
#ifndef GLM_SX_DOWN_NW
#define GLM_SX_DOWN_NW 4   // waves per workgroup = 32-column slabs that share one activation tile (r03 sweep: tools/sweep_fp8_sx.sh)
#endif
    if (p.R > 16 && mode == GEMM_GATEUP) return launch_cfg_sx<GEMM_GATEUP, GLM_SX_GATEUP_NW, 2, 1, true>(p, st);
#ifndef GLM_SM_FP8_NO_SX_DOWN
    if (p.R > 16 && mode == GEMM_DOWN) return launch_cfg_sx<GEMM_DOWN, GLM_SX_DOWN_NW, 2, 1, true>(p, st);
#endif
#endif
    if (mode == GEMM_GATEUP) return launch_cfg<GEMM_GATEUP, GLM_SM_CFG_GATEUP_W8, true>(p, st);
    if (mode == GEMM_DOWN) return launch_cfg<GEMM_DOWN, GLM_SM_CFG_DOWN_W8, true>(p, st);
    glm_set_error("gemm_smallm(fp8 weights): only the grouped expert projections are quantised");
    return GLM_ERR_UNSUPPORTED;
}
