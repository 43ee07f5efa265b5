// moe_misc.hip — the small integer/latency-bound pieces of the GLM-4-MoE block for gfx950:
//   router (a11: GEMV-shaped logits + sigmoid + bias + wave-level top-k + renormalise),
//   token→expert bucketing (replaces the per-expert mask/where/any host loop, a12),
//   and the bf16-ordered combine (a12: index_add_ in ascending expert order + shared expert).
// Compiled with -ffp-contract=off: the rounding points below are the reference's
// (SURVEY.md §8a R1–R7).
#include "common.h"
#include "fused_rows.h"
#include <string.h>

namespace {

// Token→expert bucketing: one 1024-thread workgroup (counting sort with LDS atomics).
constexpr int ALIGN_MAX_G = 1024;

// inclusive prefix sum over the 64 lanes of a wave on DPP row shifts / row broadcasts (no LDS round trips)
__device__ __forceinline__ int wave_scan_incl(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);   // row_shr:1, out-of-row lanes read 0
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);   // row_shr:8 -> inclusive inside each row of 16
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
    return v;
}

// body shared by moe_align_kernel and the router's last-arriving workgroup; cnt/cur: ALIGN_MAX_G + 1 ints of LDS each.
// SC1: topk_idx / topk_w were written by other workgroups of this launch (write-through) and are read with sc1 loads
// (topk_w as aligned 4-byte words: needs k even)
template <bool SC1>
__device__ __forceinline__ void moe_align_body(const int64_t* topk_idx, const bf16_t* topk_w, int32_t* expert_offsets,
                                               int32_t* sorted_token, bf16_t* sorted_scale, int32_t* pair_pos,
                                               int64_t T, int k, int E, int with_shared, int* cnt, int* cur) {
    const int tid = threadIdx.x;
    const int64_t P = T * k;
    for (int e = tid; e <= E; e += 1024) cnt[e] = 0;
    __syncthreads();
    // a thread's first pair (all of them when P <= 1024: every decode batch) stays in registers for the second pass,
    // its routing weight fetched in the same round trip
    int64_t e_first = -1;
    bf16_t w_first = 0;
    for (int64_t p = tid; p < P; p += 1024) {
        const int64_t e = SC1 ? (int64_t)ld_sc1_u64(topk_idx + p) : topk_idx[p];
        if (p == tid) {
            e_first = e;
            if (SC1) {
                const unsigned pr = ld_sc1_u32(topk_w + (p & ~(int64_t)1));
                w_first = (bf16_t)((p & 1) ? (pr >> 16) : (pr & 0xffffu));
            } else {
                w_first = topk_w[p];
            }
        }
        if (e >= 0 && e < E) atomicAdd(&cnt[(int)e], 1);
    }
    __syncthreads();
    if (tid < 64) {  // wave 0: exclusive scan of the counts, 64 experts at a time
        int run = 0;
        for (int e0 = 0; e0 < E; e0 += 64) {
            const int e = e0 + tid;
            const int c = e < E ? cnt[e] : 0;
            const int incl = wave_scan_incl(c);
            if (e < E) {
                cur[e] = run + incl - c;
                expert_offsets[e] = run + incl - c;
            }
            run += __builtin_amdgcn_readlane(incl, 63);
        }
        if (tid == 0) {
            expert_offsets[E] = run;  // == number of valid pairs; start of the shared group
            cur[E] = run;
            if (with_shared) expert_offsets[E + 1] = run + (int)T;
        }
    }
    __syncthreads();
    for (int64_t p = tid; p < P; p += 1024) {
        const bool first = p == tid;
        const int64_t e = first ? e_first : (SC1 ? (int64_t)ld_sc1_u64(topk_idx + p) : topk_idx[p]);
        if (e >= 0 && e < E) {
            const int pos = atomicAdd(&cur[(int)e], 1);
            sorted_token[pos] = (int32_t)(p / k);
            bf16_t wv;
            if (first) {
                wv = w_first;
            } else if (SC1) {
                const unsigned pr = ld_sc1_u32(topk_w + (p & ~(int64_t)1));
                wv = (bf16_t)((p & 1) ? (pr >> 16) : (pr & 0xffffu));
            } else {
                wv = topk_w[p];
            }
            sorted_scale[pos] = wv;
            pair_pos[p] = pos;
        } else {
            pair_pos[p] = -1;
        }
    }
    if (with_shared) {
        const int sbase = cur[E];  // unchanged by the loop above (only cur[0..E-1] advance)
        for (int64_t t = tid; t < T; t += 1024) {
            sorted_token[sbase + t] = (int32_t)t;
            sorted_scale[sbase + t] = 0x3F80;  // bf16 1.0
        }
    }
}

constexpr int ROUTER_EB = 4;  // experts per dot-product pass (x chunk reused 4x)
constexpr int ROUTER_RU = 4;  // K steps whose gate-row loads are in flight together


// One 1024-thread workgroup (16 waves) per token: the op is latency-bound (0.5 MB of gate weights
// from L2), so the 64 expert dot products are spread over as many waves as a workgroup can hold.
// x row is staged in LDS (bf16); wave w owns expert blocks w, w+16, ... of ROUTER_EB experts; a
// lane covers 16-byte chunks lane, lane+64, ... of K.
constexpr int ROUTER_NT = 1024;

// Optional fused producer of the router input (PRE != 0): the post-attention add-RMSNorm of the same token
// (layers/layernorm.py:29-40), whose x is either a bf16 row (PRE 1) or fp32 K-slice partial sums of the o_proj
// (PRE 2, glm_gemm_bf16_partial).  The normalised row goes to `y` (the experts read it) and stays in LDS for the router.
struct RouterPre {
    const bf16_t* x;        // PRE 1: [T,H] bf16
    const float* part;      // PRE 2: [nslices][T][H] fp32
    int nslices;
    int64_t slice_stride;
    const bf16_t* res_in;   // [T,H]
    const bf16_t* norm_w;   // [H]
    bf16_t* y;              // [T,H] normalised output
    bf16_t* res_out;        // [T,H]
    float eps;
    // optional: the LAST workgroup to finish also builds the sorted-row structure of glm_moe_align (one launch less)
    int32_t* al_offsets;    // nullptr = off
    int32_t* al_token;
    bf16_t* al_scale;
    int32_t* al_pos;
    int al_with_shared;
    unsigned* ticket;       // zero before the first launch; the last arriver puts the zero back
    // optional: the experts of a token split over gridDim.y workgroups (each pulls E / gridDim.y gate rows); the one
    // that finishes last (per-token ticket) gathers the logits and selects
    unsigned* tok_ticket;   // [T], zero before the first launch, restored by the kernel
    float* lg_global;       // [T, E] logits (already rounded to bf16 precision)
    int gate_lds;           // split launch: the workgroup's gate rows are staged in LDS by DMA at kernel entry (host: they fit)
};

// bytes of the x-row area in front of lg[]: the bf16 row, or (fused norm) the hand-over area of the squares chain,
// 32 B per 8-element vector beyond the first 256
__host__ __device__ inline size_t router_x_bytes(int H, bool fused_norm) {
    size_t b = (size_t)H * 2;
    if (fused_norm && H > 2048 && (size_t)H * 4 - 8192 > b) b = (size_t)H * 4 - 8192;
    return (b + 15) & ~(size_t)15;
}

// Phase timeline of the fused router launch (study builds only: build.py --variant _rtrace GLM_ROUTER_TRACE=1): thread 0
// of every workgroup stamps the 100 MHz wall clock at each phase boundary; tools/router_trace.py reads it back.
#ifdef GLM_ROUTER_TRACE
constexpr int RTRACE_SLOTS = 12, RTRACE_WGS = 4096;
__device__ unsigned long long g_rtrace[RTRACE_WGS * RTRACE_SLOTS];
#define RT_MARK(i)                                                                                                \
    do {                                                                                                          \
        const unsigned wg_ = blockIdx.x * gridDim.y + blockIdx.y;                                                 \
        if (threadIdx.x == 0 && wg_ < RTRACE_WGS) g_rtrace[wg_ * RTRACE_SLOTS + (i)] = wall_clock64();            \
    } while (0)
#else
#define RT_MARK(i) ((void)0)
#endif

// workgroup barrier that orders LDS traffic only (no wait for vector-memory operations in flight)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// bytes of one wave's gate row in LDS (whole 1 KiB DMA pieces)
__host__ __device__ inline size_t router_gate_row_bytes(int H) { return (size_t)((H / 8 + 63) / 64) * 1024; }
// LDS offset of the staged gate rows: behind the x-row area, the logits and the bias
__host__ __device__ inline size_t router_gate_lds_offset(int H, int E, bool fused_norm) {
    return (router_x_bytes(H, fused_norm) + (size_t)(E > 16 ? E : 16) * 4 + ROUTER_MAX_E * 4 + 15) & ~(size_t)15;
}

// Split launch: wave w of workgroup (t, y) owns gate row e = y * ceil(E / gridDim.y) + w.  The row does not depend on the
// activations, so its 2 H bytes are requested into LDS by DMA as early as the wave can afford — no registers held, no
// load the compiler could sink — and the dot product later reads it from LDS instead of waiting 2-3 us for cold HBM.
__device__ __forceinline__ void router_stage_gate_row(const RouterPre& pre, const bf16_t* gw, char* smem, int H, int E,
                                                      bool fused_norm, int lane, int wid) {
    if (!pre.gate_lds) return;
    const int Eg = (E + (int)gridDim.y - 1) / (int)gridDim.y;
    const int e = blockIdx.y * Eg + wid;
    if (wid >= Eg || e >= E) return;
    const int nvec = H >> 3;
    const bf16_t* grow = gw + (int64_t)e * H;
    char* dst = smem + router_gate_lds_offset(H, E, fused_norm) + (size_t)wid * router_gate_row_bytes(H);
    for (int u = 0; 64 * u < nvec; ++u)  // lanes past the row re-read its last chunk into padding nobody reads
        __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)(grow + (int64_t)min(64 * u + lane, nvec - 1) * 8),
                                         (__attribute__((address_space(3))) void*)(dst + u * 1024), 16, 0, 0);
}

// SPLIT: the experts of a token are divided over gridDim.y workgroups (compile-time: the two dot-product loops in one
// kernel body pushed the 1024-thread kernel past its 128-VGPR budget)
template <int PRE, bool SPLIT>
__global__ __launch_bounds__(ROUTER_NT) void router_topk_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ gw, const float* __restrict__ bias,
    int64_t* __restrict__ topk_idx, bf16_t* __restrict__ topk_w, bf16_t* __restrict__ logits_out,
    int H, int E, int k, float rsf, const RouterPre pre) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);                        // [H]
    float* lg = reinterpret_cast<float*>(smem + router_x_bytes(H, PRE != 0));  // [E]
    const int64_t t = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nvec = H >> 3;
    RT_MARK(0);
    constexpr int EPL = ROUTER_MAX_E / 64;
    // The correction bias is needed by wave 0 only after the logits and is cold by then (~2.5 us): requested as an
    // LDS-DMA — no register to hold, nothing the compiler could sink to the use — behind the input loads (a wave's vector
    // memory returns in order) and in front of the gate-row DMAs (hipcc drains vmcnt before an LDS-DMA that follows
    // others).  No barrier up to the selection may wait for it: they are raw barriers that order LDS traffic only.
    float* bias_l = lg + (E > 16 ? E : 16);  // [ROUTER_MAX_E]
    auto stage_bias = [&]() {
        if (wid == 0) {
#pragma unroll
            for (int i = 0; i < EPL; ++i)
                if (64 * i < E)  // lanes past E re-read the last entry into slots nobody reads
                    __builtin_amdgcn_global_load_lds(
                        (__attribute__((address_space(1))) const void*)(bias + min(64 * i + lane, E - 1)),
                        (__attribute__((address_space(3))) void*)(bias_l + 64 * i), 4, 0, 0);
        }
    };
    if (PRE == 0) {
        const u32x4_t* src = reinterpret_cast<const u32x4_t*>(x + t * H);
        u32x4_t* dst = reinterpret_cast<u32x4_t*>(xs);
        for (int i = tid; i < nvec; i += ROUTER_NT) dst[i] = src[i];
        stage_bias();
    } else {
        // add-RMSNorm, one 8-element vector per thread (host checks H <= 8 * ROUTER_NT); same arithmetic and
        // rounding points as rmsnorm_vec_kernel<ADD> / add_rmsnorm_partial_kernel in elementwise.hip
        float s8[8];
        float ss = 0.f;
        const bool act = tid < nvec;
        // the norm weight is requested with the inputs: a load issued behind the gate-row DMAs below would, when waited
        // for, wait for them as well (vector memory returns in order)
        // ... and unpacked at once, so that no wait for it is left behind the DMAs
        float nwf[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (act) {
            const u32x4_t nw = reinterpret_cast<const u32x4_t*>(pre.norm_w)[tid];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                nwf[2 * j] = bflo(nw[j]);
                nwf[2 * j + 1] = bfhi(nw[j]);
            }
        }
        if (act) {
            float xv[8];
            if (PRE == 1) {
                const u32x4_t v = reinterpret_cast<const u32x4_t*>(pre.x + t * H)[tid];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xv[2 * j] = bflo(v[j]);
                    xv[2 * j + 1] = bfhi(v[j]);
                }
            } else {
                const float* pp = pre.part + t * H + (int64_t)tid * 8;
                f32x4_t a = *reinterpret_cast<const f32x4_t*>(pp), b = *reinterpret_cast<const f32x4_t*>(pp + 4);
                // slice order; all loads of a group in flight together: one memory round trip per group.  The split
                // kernel has the registers for 8 at once (o_proj hands over 8 slices: one round trip instead of two),
                // the unsplit one sits at its 128-VGPR budget with 4
                constexpr int SG = SPLIT ? 8 : 4;
                for (int s0 = 1; s0 < pre.nslices; s0 += SG) {
                    f32x4_t ta[SG], tb[SG];
#pragma unroll
                    for (int j = 0; j < SG; ++j) {
                        const int sl = min(s0 + j, pre.nslices - 1);
                        ta[j] = *reinterpret_cast<const f32x4_t*>(pp + sl * pre.slice_stride);
                        tb[j] = *reinterpret_cast<const f32x4_t*>(pp + sl * pre.slice_stride + 4);
                    }
#pragma unroll
                    for (int j = 0; j < SG; ++j)
                        if (s0 + j < pre.nslices) {
                            a += ta[j];
                            b += tb[j];
                        }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xv[j] = rbf(a[j]);  // the linear layer's own output rounding
                    xv[4 + j] = rbf(b[j]);
                }
            }
            const u32x4_t rv = reinterpret_cast<const u32x4_t*>(pre.res_in + t * H)[tid];
            u32x4_t ov;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s8[2 * j] = xv[2 * j] + bflo(rv[j]);
                s8[2 * j + 1] = xv[2 * j + 1] + bfhi(rv[j]);
                ov[j] = pack2bf(s8[2 * j], s8[2 * j + 1]);
            }
            if (blockIdx.y == 0) reinterpret_cast<u32x4_t*>(pre.res_out + t * H)[tid] = ov;
        }
        stage_bias();
        if constexpr (SPLIT) router_stage_gate_row(pre, gw, smem, H, E, true, lane, wid);
        // sum of squares in EXACTLY the stand-alone kernels' order (rmsnorm_vec_kernel / add_rmsnorm_partial_kernel run
        // 256 threads, thread j folding vectors j, j+256, ... into one sequential chain): threads >= 256 hand their 8
        // values to thread tid % 256 through LDS (the x-row area, not yet in use), which continues its chain with them.
        // A different association flips rstd by an fp32 ulp and with it single bf16 ulps of y (found by the r02
        // repetition test; the fixed seeds of the r01 equality test happened not to hit it).
        float* xch = reinterpret_cast<float*>(smem);
        RT_MARK(1);  // inputs loaded
        if (act && tid >= 256) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xch[(tid - 256) * 8 + j] = s8[j];
        }
        // the barriers of the norm order LDS traffic only and must not wait for the gate-row / bias DMAs in flight
        lds_barrier();
        if (act && tid < 256) {
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += s8[j] * s8[j];
            for (int v = tid + 256; v < nvec; v += 256) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float f = xch[(v - 256) * 8 + j];
                    ss += f * f;
                }
            }
        }
        ss = wave_sum(ss);
        if (lane == 0) lg[wid] = ss;
        lds_barrier();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < ROUTER_NT / 64; ++w) tot += lg[w];
        lds_barrier();
        const float rstd = 1.0f / sqrtf(tot / (float)H + pre.eps);
        if (act) {
            u32x4_t o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = rbf(s8[2 * i] * rstd) * nwf[2 * i];
                const float b = rbf(s8[2 * i + 1] * rstd) * nwf[2 * i + 1];
                o[i] = pack2bf(a, b);
            }
            if (blockIdx.y == 0) reinterpret_cast<u32x4_t*>(pre.y + t * H)[tid] = o;
            reinterpret_cast<u32x4_t*>(xs)[tid] = o;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    RT_MARK(2);  // normalised row in LDS

    // R1: logits = bf16( fp32 dot )
    const int G = SPLIT ? gridDim.y : 1;
    if constexpr (SPLIT) {
        // this workgroup's experts, one per wave at a time, every K step's gate-row load in flight at once.  Per expert
        // the FMA chain runs over the lane's chunks lane, lane+64, ... in ascending order and ends in the same
        // wave_sum as the blocked loop below: the logits are bit-identical however the experts are distributed.
        const int Eg = (E + G - 1) / G;
        const int e_lo = blockIdx.y * Eg, e_hi = min(E, e_lo + Eg);
        // H <= 8 * ROUTER_NT (host check) -> at most 16 K steps per lane, taken 8 at a time (all of H = 4096 in one go;
        // 16 at once cost 64 VGPRs and pushed the 1024-thread kernel over its 128-register budget)
        constexpr int STEPS = 8;
        const bool staged = pre.gate_lds != 0;  // then Eg <= 16: one expert per wave, its row is in LDS
        if (staged) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's own row has landed
        for (int e = e_lo + wid; e < e_hi; e += ROUTER_NT / 64) {
            const u32x4_t* grow = staged ? reinterpret_cast<const u32x4_t*>(smem + router_gate_lds_offset(H, E, PRE != 0) +
                                                                            (size_t)wid * router_gate_row_bytes(H))
                                         : reinterpret_cast<const u32x4_t*>(gw + (int64_t)e * H);
            float acc = 0.f;
            for (int c0 = 0; c0 < nvec; c0 += 64 * STEPS) {
                u32x4_t wv[STEPS];
#pragma unroll
                for (int u = 0; u < STEPS; ++u) {
                    wv[u] = u32x4_t{0, 0, 0, 0};
                    if (c0 + 64 * u < nvec)  // uniform guard
                        wv[u] = grow[c0 + lane + 64 * u < nvec ? c0 + lane + 64 * u : lane];
                }
#pragma unroll
                for (int u = 0; u < STEPS; ++u) {
                    const int c = c0 + lane + 64 * u;
                    if (c < nvec) {
                        const u32x4_t xv = reinterpret_cast<const u32x4_t*>(xs)[c];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc = fmaf(bflo(xv[q]), bflo(wv[u][q]), acc);
                            acc = fmaf(bfhi(xv[q]), bfhi(wv[u][q]), acc);
                        }
                    }
                }
            }
            const float sdot = wave_sum(acc);
            if (lane == 0)  // write-through: another workgroup of this launch reads it (Guideline 16 R1)
                __hip_atomic_store(reinterpret_cast<unsigned*>(pre.lg_global + t * E + e), __float_as_uint(rbf(sdot)),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // per-token hand-off: every wave drains its (write-through) stores, the workgroup meets, one lane draws the token's
        // ticket; the workgroup that draws G-1 gathers all E logits into LDS
        RT_MARK(3);  // wave 0's dot products done
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        RT_MARK(4);  // logits drained, workgroup met
        int* tflag = reinterpret_cast<int*>(lg);
        if (tid == 0) {
            unsigned* tk = pre.tok_ticket + t;
            const unsigned got = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = got == (unsigned)G - 1u;
            if (last) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tflag[0] = last;
        }
        __syncthreads();
        RT_MARK(5);  // token ticket drawn
        if (!tflag[0]) return;
        __syncthreads();
        // every logit was stored write-through and drained before its workgroup's ticket: sc1 loads, no acquire
        for (int e = tid; e < E; e += ROUTER_NT) lg[e] = __uint_as_float(ld_sc1_u32(pre.lg_global + t * E + e));
    }
    const int nblk = SPLIT ? 0 : (E + ROUTER_EB - 1) / ROUTER_EB;
    for (int b = wid; b < nblk; b += ROUTER_NT / 64) {
        const int e0 = b * ROUTER_EB;
        float acc[ROUTER_EB];
#pragma unroll
        for (int j = 0; j < ROUTER_EB; ++j) acc[j] = 0.f;
        // the gate rows come from L2 (~1 us per round trip): ROUTER_RU x ROUTER_EB row chunks are requested before the
        // first FMA (a wave has only nvec/64 = 8 steps at H=4096, waiting for each one separately cost ~8 us);
        // the FMAs keep the ascending-k order, so the sums are bit-identical to the one-step-at-a-time loop
        for (int c0 = lane; c0 < nvec; c0 += 64 * ROUTER_RU) {
            u32x4_t wv[ROUTER_RU][ROUTER_EB];
#pragma unroll
            for (int u = 0; u < ROUTER_RU; ++u) {
                const int cc = c0 + 64 * u < nvec ? c0 + 64 * u : lane;  // clamped re-read, not accumulated
#pragma unroll
                for (int j = 0; j < ROUTER_EB; ++j) {
                    const int e = e0 + j < E ? e0 + j : E - 1;
                    wv[u][j] = reinterpret_cast<const u32x4_t*>(gw + (int64_t)e * H)[cc];
                }
            }
#pragma unroll
            for (int u = 0; u < ROUTER_RU; ++u) {
                const int c = c0 + 64 * u;
                if (c < nvec) {
                    const u32x4_t xv = reinterpret_cast<const u32x4_t*>(xs)[c];
#pragma unroll
                    for (int j = 0; j < ROUTER_EB; ++j) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc[j] = fmaf(bflo(xv[q]), bflo(wv[u][j][q]), acc[j]);
                            acc[j] = fmaf(bfhi(xv[q]), bfhi(wv[u][j][q]), acc[j]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < ROUTER_EB; ++j) {
            const float s = wave_sum(acc[j]);
            if (lane == 0 && e0 + j < E) lg[e0 + j] = rbf(s);
        }
    }
    __syncthreads();
    RT_MARK(6);  // all logits of the token in LDS
    const bool fuse_align = pre.al_offsets != nullptr;  // uniform (kernel argument)
    if (wid != 0 && !fuse_align) return;
    const bool align_here = fuse_align && gridDim.x == 1;  // one token: the selecting wave buckets (see router_select_wave)
    if (wid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the bias DMA has landed
        if (align_here)
            router_select_wave<EPL>(lg, bias_l, reinterpret_cast<unsigned*>(smem), reinterpret_cast<bf16_t*>(lg), lane, E, k,
                                    rsf, t, topk_idx, topk_w, logits_out, false, pre.al_offsets, pre.al_token, pre.al_scale,
                                    pre.al_pos, pre.al_with_shared);
        else
            router_select_wave<EPL>(lg, bias_l, reinterpret_cast<unsigned*>(smem), reinterpret_cast<bf16_t*>(lg), lane, E, k,
                                    rsf, t, topk_idx, topk_w, logits_out, fuse_align);
    }  // wave 0
    RT_MARK(7);  // selection stored
    if (!fuse_align || align_here) return;

    // ---- last-arriving workgroup: token -> expert bucketing over ALL tokens' routing (glm_moe_align) ----
    // Hand-off (cdna_hip_programming.md Guideline 16, counter form): wave 0 is the only wave that stored routing data;
    // it drains its stores, lane 0 releases at agent scope and draws a ticket.  The workgroup that draws T-1 acquires,
    // waits, and only then — behind the workgroup barrier — do all its waves read the other workgroups' rows.
    // With write-through routing stores (k even) neither fence is needed: the drain before the ticket publishes, the
    // bucketing reads with sc1 loads (Guideline 16 R1; valid-forms table, row 1: one lane of each storing workgroup
    // adds to one counter, the last adder's workgroup loads behind its barrier).
    const bool wt2 = (k & 1) == 0;
    int* flag = reinterpret_cast<int*>(lg) + 32;  // beside the staged weights; lg[] holds >= 64 floats whenever E >= 64...
    if (E < 64) flag = reinterpret_cast<int*>(smem);  // ... else the x row (dead: every logit has been consumed)
    if (wid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            if (!wt2) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const unsigned got = __hip_atomic_fetch_add(pre.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = got == (unsigned)gridDim.x - 1u;  // one drawer per token (the token's selecting workgroup)
            if (last) {
                if (!wt2) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __hip_atomic_store(pre.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
            }
            flag[0] = last;
        }
    }
    __syncthreads();
    const int is_last = flag[0];
    __syncthreads();  // flag read by everyone before the LDS is reused
    RT_MARK(8);  // global ticket drawn
    if (!is_last) return;
    int* cnt = reinterpret_cast<int*>(smem);  // the x row is dead; 2 * (E + 1) ints fit (host checks)
    int* cur = cnt + (E + 1);
    if (wt2)
        moe_align_body<true>(topk_idx, topk_w, pre.al_offsets, pre.al_token, pre.al_scale, pre.al_pos, (int64_t)gridDim.x, k,
                             E, pre.al_with_shared, cnt, cur);
    else
        moe_align_body<false>(topk_idx, topk_w, pre.al_offsets, pre.al_token, pre.al_scale, pre.al_pos, (int64_t)gridDim.x, k,
                              E, pre.al_with_shared, cnt, cur);
    __syncthreads();
    RT_MARK(9);  // bucketing done
}

// ------------------------------------------------------------------------------------------
// Prefill-size router: R1 (logits = bf16(x . Wg^T), moe.py:33) is a plain [T, H] x [H, E] GEMM there (glm_gemm_bf16) and
// this kernel does R2..R7 from its bf16 output, one wave per token, with the decode kernel's own selection code.  At
// T = 16384 the token-per-workgroup kernel above costs 1.2 ms per layer (65536 workgroups of 1024 threads, each redoing
// the norm and pulling 128 KiB of gate rows from L2); norm + GEMM + this + bucketing cost 0.34 ms.
__global__ __launch_bounds__(256) void router_select_kernel(const bf16_t* __restrict__ logits, int64_t ld_logits,
                                                            const float* __restrict__ bias,
                                                            int64_t* __restrict__ topk_idx, bf16_t* __restrict__ topk_w,
                                                            int64_t T, int E, int k, float rsf) {
    constexpr int EPL = ROUTER_MAX_E / 64;
    __shared__ float lg_s[4][ROUTER_MAX_E];
    __shared__ float bias_s[ROUTER_MAX_E];
    __shared__ unsigned keys_s[4][64];
    __shared__ bf16_t wl_s[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int e = tid; e < E; e += 256) bias_s[e] = bias[e];
    const int64_t t = (int64_t)blockIdx.x * 4 + wid;
    if (t < T) {
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
            const int e = lane + 64 * i;
            if (e < E) lg_s[wid][e] = bf2f(logits[t * ld_logits + e]);
        }
    }
    __syncthreads();
    if (t >= T) return;
    router_select_wave<EPL>(lg_s[wid], bias_s, keys_s[wid], wl_s[wid], lane, E, k, rsf, t, topk_idx, topk_w, nullptr, false);
}

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void moe_align_kernel(
    const int64_t* __restrict__ topk_idx, const bf16_t* __restrict__ topk_w,
    int32_t* __restrict__ expert_offsets, int32_t* __restrict__ sorted_token,
    bf16_t* __restrict__ sorted_scale, int32_t* __restrict__ pair_pos, int64_t T, int k, int E,
    int with_shared) {
    __shared__ int cnt[ALIGN_MAX_G + 1];
    __shared__ int cur[ALIGN_MAX_G + 1];
    moe_align_body<false>(topk_idx, topk_w, expert_offsets, sorted_token, sorted_scale, pair_pos, T, k, E, with_shared, cnt, cur);
}

// ------------------------------------------------------------------------------------------
// Bucketing at prefill sizes (T k >= ALIGN_PAR_MIN_PAIRS): the single-workgroup kernel above walks 131072 pairs in 128
// serial rounds (200 us at T = 16384).  Three launches over chunks of ALIGN_CHUNK pairs instead:
//   count:   chunk c histograms its pairs in LDS and reserves, per expert, a range inside the expert's group with ONE
//            global add per (chunk, expert): base[c][e] = atomicAdd(total[e], hist[e])  (total[] zeroed by the host side)
//   offsets: exclusive scan of total[] -> expert_offsets (one wave)
//   scatter: chunk c redoes its LDS ranking from expert_offsets[e] + base[c][e] and writes the rows.
// Rows of one expert end up in arrival order of the chunks' adds — as with the LDS atomics of the small kernel the order
// inside a group is not specified (no consumer depends on it: pair_pos carries the permutation).
constexpr int ALIGN_CHUNK = 8192;  // pairs per workgroup (8 per thread)

__global__ __launch_bounds__(1024) void moe_align_count_kernel(const int64_t* __restrict__ topk_idx, int64_t P, int E,
                                                               int32_t* __restrict__ total, int32_t* __restrict__ base) {
    __shared__ int hist[ALIGN_MAX_G];
    const int tid = threadIdx.x;
    for (int e = tid; e < E; e += 1024) hist[e] = 0;
    __syncthreads();
    const int64_t p0 = (int64_t)blockIdx.x * ALIGN_CHUNK;
#pragma unroll
    for (int i = 0; i < ALIGN_CHUNK / 1024; ++i) {
        const int64_t p = p0 + i * 1024 + tid;
        if (p < P) {
            const int64_t e = topk_idx[p];
            if (e >= 0 && e < E) atomicAdd(&hist[(int)e], 1);
        }
    }
    __syncthreads();
    for (int e = tid; e < E; e += 1024) {
        const int c = hist[e];
        base[(int64_t)blockIdx.x * E + e] = c ? atomicAdd(&total[e], c) : 0;
    }
}

__global__ __launch_bounds__(64) void moe_align_offsets_kernel(const int32_t* __restrict__ total, int E, int64_t T,
                                                               int with_shared, int32_t* __restrict__ expert_offsets) {
    const int lane = threadIdx.x;
    int run = 0;
    for (int e0 = 0; e0 < E; e0 += 64) {
        const int e = e0 + lane;
        const int c = e < E ? total[e] : 0;
        const int incl = wave_scan_incl(c);
        if (e < E) expert_offsets[e] = run + incl - c;
        run += __builtin_amdgcn_readlane(incl, 63);
    }
    if (lane == 0) {
        expert_offsets[E] = run;
        if (with_shared) expert_offsets[E + 1] = run + (int)T;
    }
}

__global__ __launch_bounds__(1024) void moe_align_scatter_kernel(
    const int64_t* __restrict__ topk_idx, const bf16_t* __restrict__ topk_w, const int32_t* __restrict__ expert_offsets,
    const int32_t* __restrict__ base, int32_t* __restrict__ sorted_token, bf16_t* __restrict__ sorted_scale,
    int32_t* __restrict__ pair_pos, int64_t T, int k, int E, int with_shared) {
    __shared__ int cur[ALIGN_MAX_G];
    const int tid = threadIdx.x;
    const int64_t P = T * k;
    for (int e = tid; e < E; e += 1024) cur[e] = expert_offsets[e] + base[(int64_t)blockIdx.x * E + e];
    __syncthreads();
    const int64_t p0 = (int64_t)blockIdx.x * ALIGN_CHUNK;
#pragma unroll
    for (int i = 0; i < ALIGN_CHUNK / 1024; ++i) {
        const int64_t p = p0 + i * 1024 + tid;
        if (p < P) {
            const int64_t e = topk_idx[p];
            if (e >= 0 && e < E) {
                const int pos = atomicAdd(&cur[(int)e], 1);
                sorted_token[pos] = (int32_t)(p / k);
                sorted_scale[pos] = topk_w[p];
                pair_pos[p] = pos;
            } else {
                pair_pos[p] = -1;
            }
        }
    }
    if (with_shared) {  // the shared group's rows of this chunk's token range
        const int sbase = expert_offsets[E];
        const int64_t t0 = (int64_t)blockIdx.x * ALIGN_CHUNK;  // chunks outnumber ceil(T / ALIGN_CHUNK): k >= 1
        for (int64_t t = t0 + tid; t < min(T, t0 + ALIGN_CHUNK); t += 1024) {
            sorted_token[sbase + t] = (int32_t)t;
            sorted_scale[sbase + t] = 0x3F80;  // bf16 1.0
        }
    }
}

// ------------------------------------------------------------------------------------------
// Combine: out[t] = (((0 + y[p0]) + y[p1]) + ...) + y[shared], one bf16 rounding per add,
// p0 < p1 < ... = the token's routed rows in ascending expert id (sorted rows ascend with id).
constexpr int COMBINE_MAX_K = 64;

// row_scale != nullptr (expert-parallel combine): the rows are the experts' UNSCALED bf16 outputs as they came back
// from their owners, and the routing weight is applied here with the down projection's own rounding,
// bf16(bf16(acc) * w) (moe.py:149); shared != nullptr: the shared expert's rows live in their own [T, H] buffer.
__global__ __launch_bounds__(256) void moe_combine_kernel(const bf16_t* __restrict__ y, int64_t ldy,
                                                          const int32_t* __restrict__ pair_pos,
                                                          bf16_t* __restrict__ out, int k, int H,
                                                          int64_t shared_base, int vec_ok,
                                                          const bf16_t* __restrict__ row_scale,
                                                          const bf16_t* __restrict__ shared, int64_t ld_shared) {
    // grid = (token, column slab): a token's row is split over gridDim.y workgroups so that a
    // decode batch (32 tokens) still spreads over the chip; every workgroup re-derives the sorted
    // row list (k <= 64 integers) on its own.
    __shared__ int pos[COMBINE_MAX_K];
    __shared__ int npos;
    const int64_t t = blockIdx.x;
    if (threadIdx.x == 0) {
        int n = 0;
        for (int j = 0; j < k; ++j) {
            const int p = pair_pos[t * k + j];
            if (p < 0) continue;
            int i = n++;
            while (i > 0 && pos[i - 1] > p) {
                pos[i] = pos[i - 1];
                --i;
            }
            pos[i] = p;
        }
        npos = n;
    }
    __syncthreads();
    const int n = npos;
    if (vec_ok) {
        const int nvec = H >> 3;
        for (int c = blockIdx.y * 256 + threadIdx.x; c < nvec; c += gridDim.y * 256) {
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = 0.f;
            for (int j = 0; j < n; ++j) {
                const u32x4_t v = reinterpret_cast<const u32x4_t*>(y + (int64_t)pos[j] * ldy)[c];
                if (row_scale) {
                    const float sc = bf2f(row_scale[pos[j]]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[2 * q] = rbf(acc[2 * q] + rbf(bflo(v[q]) * sc));
                        acc[2 * q + 1] = rbf(acc[2 * q + 1] + rbf(bfhi(v[q]) * sc));
                    }
                    continue;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[2 * q] = rbf(acc[2 * q] + bflo(v[q]));
                    acc[2 * q + 1] = rbf(acc[2 * q + 1] + bfhi(v[q]));
                }
            }
            if (shared_base >= 0 || shared) {
                const u32x4_t v = shared ? reinterpret_cast<const u32x4_t*>(shared + t * ld_shared)[c]
                                         : reinterpret_cast<const u32x4_t*>(y + (shared_base + t) * ldy)[c];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[2 * q] = acc[2 * q] + bflo(v[q]);
                    acc[2 * q + 1] = acc[2 * q + 1] + bfhi(v[q]);
                }
            }
            u32x4_t o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = pack2bf(acc[2 * q], acc[2 * q + 1]);
            reinterpret_cast<u32x4_t*>(out + t * H)[c] = o;
        }
    } else {
        for (int c = blockIdx.y * 256 + threadIdx.x; c < H; c += gridDim.y * 256) {
            float acc = 0.f;
            for (int j = 0; j < n; ++j) {
                float v = bf2f(y[(int64_t)pos[j] * ldy + c]);
                if (row_scale) v = rbf(v * bf2f(row_scale[pos[j]]));
                acc = rbf(acc + v);
            }
            if (shared) acc = acc + bf2f(shared[t * ld_shared + c]);
            else if (shared_base >= 0) acc = acc + bf2f(y[(shared_base + t) * ldy + c]);
            out[t * H + c] = f2bf(acc);
        }
    }
}

// Combine fused with the NEXT layer's input add-RMSNorm (decode_layer.py:47-49 of layer i+1): one 512-thread workgroup
// per token.  x = the combine result rounded to bf16 exactly as moe_combine_kernel writes it, then
// s = x + residual, residual_out = bf16(s), y = bf16(bf16(s*rstd)*w) as rmsnorm_vec_kernel<ADD>.
constexpr int CN_NT = GLM_CN_NT;
template <int VPT>
__global__ __launch_bounds__(CN_NT) void moe_combine_add_rmsnorm_kernel(
    const bf16_t* __restrict__ y, int64_t ldy, const int32_t* __restrict__ pair_pos, int k, int64_t shared_base,
    const bf16_t* res_in, const bf16_t* __restrict__ w, bf16_t* out, bf16_t* res_out, int H, float eps) {
    __shared__ float red[CN_NT / 64];
    // the row function is shared with the one-token QKV launch that runs it as its prologue (gemm_tinym.hip)
    glm_combine_add_rmsnorm_row<VPT>(y, ldy, pair_pos, k, shared_base, res_in, w, out, res_out, nullptr, H, eps, blockIdx.x, red);
}

}  // namespace

#ifdef GLM_ROUTER_TRACE
extern "C" int glm_debug_router_trace(unsigned long long* host_dst, int clear) {
    if (clear) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_rtrace)) != hipSuccess) return -1;
        return hipMemset(p, 0, sizeof(unsigned long long) * RTRACE_WGS * RTRACE_SLOTS) == hipSuccess ? 0 : -1;
    }
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_rtrace), sizeof(unsigned long long) * RTRACE_WGS * RTRACE_SLOTS) ==
                   hipSuccess ? 0 : -1;
}
#endif

extern "C" int glm_router_topk(const void* x, const void* gate_w,
                               const float* e_score_correction_bias, int64_t* topk_idx,
                               void* topk_w, void* logits_out, int64_t T, int H, int E, int k,
                               float routed_scaling_factor, glm_stream_t stream) {
    GLM_CHECK_ARG(x && gate_w && e_score_correction_bias && topk_idx && topk_w, GLM_ERR_INVALID,
                  "glm_router_topk: null pointer");
    GLM_CHECK_ARG(T >= 0 && H > 0 && E > 0 && k > 0 && k <= E, GLM_ERR_INVALID,
                  "glm_router_topk: bad shape T=%ld H=%d E=%d k=%d", (long)T, H, E, k);
    GLM_CHECK_ARG(E <= ROUTER_MAX_E && k <= 64 && H % 8 == 0 && H <= 32768 && glm_aligned16(x) &&
                      glm_aligned16(gate_w),
                  GLM_ERR_UNSUPPORTED,
                  "glm_router_topk: needs E<=256, k<=64, H%%8==0, H<=32768, 16B-aligned x/W (E=%d k=%d H=%d)",
                  E, k, H);
    if (T == 0) return GLM_OK;
    const size_t shm = router_x_bytes(H, false) + (size_t)(E > 16 ? E : 16) * 4 + ROUTER_MAX_E * 4;  // x | logits | bias
    RouterPre pre;
    memset(&pre, 0, sizeof(pre));
    hipLaunchKernelGGL((router_topk_kernel<0, false>), dim3((unsigned)T), dim3(ROUTER_NT), shm, glm_s(stream),
                       (const bf16_t*)x, (const bf16_t*)gate_w, e_score_correction_bias, topk_idx,
                       (bf16_t*)topk_w, (bf16_t*)logits_out, H, E, k, routed_scaling_factor, pre);
    GLM_CHECK_LAUNCH("glm_router_topk");
    return GLM_OK;
}

extern "C" int glm_router_select(const void* logits, int64_t ld_logits, const float* e_score_correction_bias,
                                 int64_t* topk_idx, void* topk_w, int64_t T, int E, int k,
                                 float routed_scaling_factor, glm_stream_t stream) {
    GLM_CHECK_ARG(logits && e_score_correction_bias && topk_idx && topk_w, GLM_ERR_INVALID, "glm_router_select: null pointer");
    GLM_CHECK_ARG(T >= 0 && E > 0 && k > 0 && k <= E && ld_logits >= E, GLM_ERR_INVALID,
                  "glm_router_select: bad shape T=%ld E=%d k=%d", (long)T, E, k);
    GLM_CHECK_ARG(E <= ROUTER_MAX_E && k <= 64 && T <= 0x7fffffffll * 4, GLM_ERR_UNSUPPORTED,
                  "glm_router_select: needs E<=256, k<=64 (E=%d k=%d)", E, k);
    if (T == 0) return GLM_OK;
    hipLaunchKernelGGL(router_select_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, glm_s(stream),
                       (const bf16_t*)logits, ld_logits, e_score_correction_bias, topk_idx, (bf16_t*)topk_w, T, E, k,
                       routed_scaling_factor);
    GLM_CHECK_LAUNCH("glm_router_select");
    return GLM_OK;
}

static int add_rmsnorm_router_impl(const void* x, const float* part, int nslices, int64_t slice_stride,
                                   const void* residual_in, const void* norm_weight, float eps, void* y,
                                   void* residual_out, const void* gate_w, const float* e_score_correction_bias,
                                   int64_t* topk_idx, void* topk_w, int64_t T, int H, int E, int k,
                                   float routed_scaling_factor, int32_t* expert_offsets, int32_t* sorted_token,
                                   void* sorted_scale, int32_t* pair_pos, int with_shared, uint32_t* workspace,
                                   size_t workspace_bytes, glm_stream_t stream);

// workspace of the fused router + align launch: [global ticket | 65536 per-token tickets | T x E logits (fp32)]
constexpr size_t ROUTER_WS_TICKETS = 65536;
constexpr size_t ROUTER_WS_HEAD = 64 + ROUTER_WS_TICKETS * 4;
extern "C" size_t glm_router_workspace_bytes(int64_t T, int E) {
    return ROUTER_WS_HEAD + (size_t)(T > 0 ? T : 0) * (size_t)(E > 0 ? E : 0) * sizeof(float);
}

extern "C" int glm_add_rmsnorm_router_topk(const void* x, const float* part, int nslices, int64_t slice_stride,
                                           const void* residual_in, const void* norm_weight, float eps, void* y,
                                           void* residual_out, const void* gate_w,
                                           const float* e_score_correction_bias, int64_t* topk_idx, void* topk_w,
                                           int64_t T, int H, int E, int k, float routed_scaling_factor,
                                           glm_stream_t stream) {
    return add_rmsnorm_router_impl(x, part, nslices, slice_stride, residual_in, norm_weight, eps, y, residual_out, gate_w,
                                   e_score_correction_bias, topk_idx, topk_w, T, H, E, k, routed_scaling_factor, nullptr,
                                   nullptr, nullptr, nullptr, 0, nullptr, 0, stream);
}

extern "C" int glm_add_rmsnorm_router_topk_align(const void* x, const float* part, int nslices, int64_t slice_stride,
                                                 const void* residual_in, const void* norm_weight, float eps, void* y,
                                                 void* residual_out, const void* gate_w,
                                                 const float* e_score_correction_bias, int64_t* topk_idx, void* topk_w,
                                                 int64_t T, int H, int E, int k, float routed_scaling_factor,
                                                 int32_t* expert_offsets, int32_t* sorted_token, void* sorted_scale,
                                                 int32_t* pair_pos, int with_shared, uint32_t* workspace,
                                                 size_t workspace_bytes, glm_stream_t stream) {
    GLM_CHECK_ARG(expert_offsets && sorted_token && sorted_scale && pair_pos && workspace, GLM_ERR_INVALID,
                  "glm_add_rmsnorm_router_topk_align: null pointer");
    GLM_CHECK_ARG(workspace_bytes >= ROUTER_WS_HEAD && glm_aligned16(workspace), GLM_ERR_WORKSPACE,
                  "glm_add_rmsnorm_router_topk_align: workspace %zu < %zu bytes", workspace_bytes, ROUTER_WS_HEAD);
    GLM_CHECK_ARG(E < ALIGN_MAX_G && T * (int64_t)(k + 1) < (1ll << 31) && T <= 0x7fffffff &&
                      (size_t)2 * (E + 1) * 4 <= (size_t)H * 2,
                  GLM_ERR_UNSUPPORTED, "glm_add_rmsnorm_router_topk_align: needs E<%d, T*(k+1) < 2^31, 4(E+1) <= H",
                  ALIGN_MAX_G);
    return add_rmsnorm_router_impl(x, part, nslices, slice_stride, residual_in, norm_weight, eps, y, residual_out, gate_w,
                                   e_score_correction_bias, topk_idx, topk_w, T, H, E, k, routed_scaling_factor,
                                   expert_offsets, sorted_token, sorted_scale, pair_pos, with_shared, workspace,
                                   workspace_bytes, stream);
}

static int add_rmsnorm_router_impl(const void* x, const float* part, int nslices, int64_t slice_stride,
                                   const void* residual_in, const void* norm_weight, float eps, void* y,
                                   void* residual_out, const void* gate_w, const float* e_score_correction_bias,
                                   int64_t* topk_idx, void* topk_w, int64_t T, int H, int E, int k,
                                   float routed_scaling_factor, int32_t* expert_offsets, int32_t* sorted_token,
                                   void* sorted_scale, int32_t* pair_pos, int with_shared, uint32_t* workspace,
                                   size_t workspace_bytes, glm_stream_t stream) {
    GLM_CHECK_ARG((x != nullptr) != (part != nullptr), GLM_ERR_INVALID,
                  "glm_add_rmsnorm_router_topk: give either x (bf16) or part (fp32 partial sums)");
    GLM_CHECK_ARG(residual_in && norm_weight && y && residual_out && gate_w && e_score_correction_bias && topk_idx &&
                      topk_w, GLM_ERR_INVALID, "glm_add_rmsnorm_router_topk: null pointer");
    GLM_CHECK_ARG(T >= 0 && H > 0 && E > 0 && k > 0 && k <= E && (part == nullptr || (nslices >= 1 &&
                      slice_stride >= T * (int64_t)H)), GLM_ERR_INVALID, "glm_add_rmsnorm_router_topk: bad shape");
    GLM_CHECK_ARG(E <= ROUTER_MAX_E && k <= 64 && H % 8 == 0 && H <= 8 * ROUTER_NT && (x == nullptr || glm_aligned16(x)) &&
                      (part == nullptr || (glm_aligned16(part) && slice_stride % 4 == 0)) && glm_aligned16(residual_in) &&
                      glm_aligned16(norm_weight) && glm_aligned16(y) && glm_aligned16(residual_out) && glm_aligned16(gate_w),
                  GLM_ERR_UNSUPPORTED,
                  "glm_add_rmsnorm_router_topk: needs E<=256, k<=64, H%%8==0, H<=%d, 16B-aligned buffers (E=%d k=%d H=%d)",
                  8 * ROUTER_NT, E, k, H);
    if (T == 0) return GLM_OK;
    const size_t shm = router_x_bytes(H, true) + (size_t)(E > ROUTER_NT / 64 ? E : ROUTER_NT / 64) * 4 + ROUTER_MAX_E * 4;
    RouterPre pre;
    pre.x = (const bf16_t*)x; pre.part = part; pre.nslices = nslices; pre.slice_stride = slice_stride;
    pre.res_in = (const bf16_t*)residual_in; pre.norm_w = (const bf16_t*)norm_weight; pre.y = (bf16_t*)y;
    pre.res_out = (bf16_t*)residual_out; pre.eps = eps;
    pre.al_offsets = expert_offsets; pre.al_token = sorted_token; pre.al_scale = (bf16_t*)sorted_scale;
    pre.al_pos = pair_pos; pre.al_with_shared = with_shared; pre.ticket = workspace;
    pre.tok_ticket = nullptr; pre.lg_global = nullptr;
    // experts of a token over several workgroups (16 gate rows each) when the workspace holds the logits and the tickets
    unsigned G = 1;
    size_t lds = shm;
    pre.gate_lds = 0;
    if (workspace != nullptr && E >= 32 && T <= (int64_t)ROUTER_WS_TICKETS &&
        workspace_bytes >= ROUTER_WS_HEAD + (size_t)T * E * sizeof(float)) {
        G = (unsigned)((E + 15) / 16);
        pre.tok_ticket = workspace + 16;
        pre.lg_global = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + ROUTER_WS_HEAD);
        // the workgroup's gate rows staged in LDS when they fit beside the row and the logits (H = 4096: 16 x 8 KiB)
        const unsigned Eg = ((unsigned)E + G - 1) / G;
        const size_t with_rows = router_gate_lds_offset(H, E, true) + (size_t)Eg * router_gate_row_bytes(H);
        // ... and only while the grid is one workgroup per CU anyway: the rows cost the second workgroup of a CU
        if (Eg <= ROUTER_NT / 64 && with_rows <= 160 * 1024 && T * (int64_t)G <= 256) {
            pre.gate_lds = 1;
            lds = with_rows > shm ? with_rows : shm;
        }
    }
    const dim3 grid((unsigned)T, G);
#define GLM_ROUTER_LAUNCH(PRE_, SPLIT_)                                                                                  \
    do {                                                                                                                 \
        if (lds > 64 * 1024)                                                                                             \
            GLM_SET_MAX_LDS((router_topk_kernel<PRE_, SPLIT_>), 160 * 1024, "glm_add_rmsnorm_router_topk");               \
        hipLaunchKernelGGL((router_topk_kernel<PRE_, SPLIT_>), grid, dim3(ROUTER_NT), lds, glm_s(stream),                \
                           (const bf16_t*)y, (const bf16_t*)gate_w, e_score_correction_bias, topk_idx, (bf16_t*)topk_w,  \
                           (bf16_t*)nullptr, H, E, k, routed_scaling_factor, pre);                                       \
    } while (0)
    if (x) {
        if (G > 1) GLM_ROUTER_LAUNCH(1, true);
        else GLM_ROUTER_LAUNCH(1, false);
    } else {
        if (G > 1) GLM_ROUTER_LAUNCH(2, true);
        else GLM_ROUTER_LAUNCH(2, false);
    }
#undef GLM_ROUTER_LAUNCH
    GLM_CHECK_LAUNCH("glm_add_rmsnorm_router_topk");
    return GLM_OK;
}

extern "C" int glm_moe_align(const int64_t* topk_idx, const void* topk_w, int32_t* expert_offsets,
                             int32_t* sorted_token, void* sorted_scale, int32_t* pair_pos,
                             int64_t T, int k, int E, int with_shared, glm_stream_t stream) {
    GLM_CHECK_ARG(topk_idx && topk_w && expert_offsets && sorted_token && sorted_scale && pair_pos,
                  GLM_ERR_INVALID, "glm_moe_align: null pointer");
    GLM_CHECK_ARG(T >= 0 && k > 0 && E > 0, GLM_ERR_INVALID, "glm_moe_align: bad shape");
    GLM_CHECK_ARG(E < ALIGN_MAX_G && T * (int64_t)(k + 1) < (1ll << 31), GLM_ERR_UNSUPPORTED,
                  "glm_moe_align: needs E<%d and T*(k+1) < 2^31", ALIGN_MAX_G);
    hipLaunchKernelGGL(moe_align_kernel, dim3(1), dim3(1024), 0, glm_s(stream), topk_idx,
                       (const bf16_t*)topk_w, expert_offsets, sorted_token, (bf16_t*)sorted_scale,
                       pair_pos, T, k, E, with_shared);
    GLM_CHECK_LAUNCH("glm_moe_align");
    return GLM_OK;
}

extern "C" size_t glm_moe_align_workspace_bytes(int64_t T, int k, int E) {
    const int64_t P = (T > 0 ? T : 0) * (int64_t)(k > 0 ? k : 0);
    const int64_t chunks = (P + ALIGN_CHUNK - 1) / ALIGN_CHUNK;
    return (size_t)(E > 0 ? E : 0) * sizeof(int32_t) * (size_t)(1 + chunks);
}

// glm_moe_align over many workgroups (prefill sizes).  workspace: glm_moe_align_workspace_bytes(T, k, E), its first E
// words (the per-expert totals) ZERO on entry — the caller clears them (hipMemsetAsync / a zeroed tensor) per call.
extern "C" int glm_moe_align_parallel(const int64_t* topk_idx, const void* topk_w, int32_t* expert_offsets,
                                      int32_t* sorted_token, void* sorted_scale, int32_t* pair_pos, int64_t T, int k,
                                      int E, int with_shared, int32_t* workspace, size_t workspace_bytes,
                                      glm_stream_t stream) {
    GLM_CHECK_ARG(topk_idx && topk_w && expert_offsets && sorted_token && sorted_scale && pair_pos && workspace,
                  GLM_ERR_INVALID, "glm_moe_align_parallel: null pointer");
    GLM_CHECK_ARG(T >= 0 && k > 0 && E > 0, GLM_ERR_INVALID, "glm_moe_align_parallel: bad shape");
    GLM_CHECK_ARG(E < ALIGN_MAX_G && T * (int64_t)(k + 1) < (1ll << 31), GLM_ERR_UNSUPPORTED,
                  "glm_moe_align_parallel: needs E<%d and T*(k+1) < 2^31", ALIGN_MAX_G);
    GLM_CHECK_ARG(workspace_bytes >= glm_moe_align_workspace_bytes(T, k, E), GLM_ERR_WORKSPACE,
                  "glm_moe_align_parallel: workspace %zu < %zu bytes", workspace_bytes, glm_moe_align_workspace_bytes(T, k, E));
    const int64_t P = T * k;
    const unsigned chunks = (unsigned)((P + ALIGN_CHUNK - 1) / ALIGN_CHUNK);
    hipStream_t st = glm_s(stream);
    int32_t* total = workspace;
    int32_t* base = workspace + E;
    if (chunks > 0)
        hipLaunchKernelGGL(moe_align_count_kernel, dim3(chunks), dim3(1024), 0, st, topk_idx, P, E, total, base);
    hipLaunchKernelGGL(moe_align_offsets_kernel, dim3(1), dim3(64), 0, st, total, E, T, with_shared, expert_offsets);
    if (chunks > 0)
        hipLaunchKernelGGL(moe_align_scatter_kernel, dim3(chunks), dim3(1024), 0, st, topk_idx, (const bf16_t*)topk_w,
                           expert_offsets, base, sorted_token, (bf16_t*)sorted_scale, pair_pos, T, k, E, with_shared);
    GLM_CHECK_LAUNCH("glm_moe_align_parallel");
    return GLM_OK;
}

extern "C" int glm_moe_combine(const void* y, int64_t ldy, const int32_t* pair_pos, void* out,
                               int64_t T, int k, int H, int64_t shared_base, glm_stream_t stream) {
    GLM_CHECK_ARG(y && pair_pos && out, GLM_ERR_INVALID, "glm_moe_combine: null pointer");
    GLM_CHECK_ARG(T >= 0 && k > 0 && H > 0 && ldy >= H, GLM_ERR_INVALID, "glm_moe_combine: bad shape");
    GLM_CHECK_ARG(k <= COMBINE_MAX_K, GLM_ERR_UNSUPPORTED, "glm_moe_combine: k > %d", COMBINE_MAX_K);
    if (T == 0) return GLM_OK;
    const int vec_ok = H % 8 == 0 && ldy % 8 == 0 && glm_aligned16(y) && glm_aligned16(out);
    const int work = vec_ok ? H / 8 : H;                       // items per token row
    int slabs = (work + 255) / 256;                            // at most one item per thread
    while (slabs > 1 && (int64_t)T * slabs > 2048) slabs = (slabs + 1) / 2;  // enough WGs already
    GLM_CHECK_ARG(T <= 0x7fffffff, GLM_ERR_UNSUPPORTED, "glm_moe_combine: T too large");
    hipLaunchKernelGGL(moe_combine_kernel, dim3((unsigned)T, (unsigned)slabs), dim3(256), 0, glm_s(stream),
                       (const bf16_t*)y, ldy, pair_pos, (bf16_t*)out, k, H, shared_base, vec_ok, (const bf16_t*)nullptr,
                       (const bf16_t*)nullptr, (int64_t)0);
    GLM_CHECK_LAUNCH("glm_moe_combine");
    return GLM_OK;
}

extern "C" int glm_moe_combine_scaled(const void* y, int64_t ldy, const int32_t* pair_pos, const void* row_scale,
                                      const void* shared_rows, int64_t ld_shared, void* out, int64_t T, int k, int H,
                                      glm_stream_t stream) {
    GLM_CHECK_ARG(y && pair_pos && row_scale && out, GLM_ERR_INVALID, "glm_moe_combine_scaled: null pointer");
    GLM_CHECK_ARG(T >= 0 && k > 0 && H > 0 && ldy >= H && (shared_rows == nullptr || ld_shared >= H), GLM_ERR_INVALID,
                  "glm_moe_combine_scaled: bad shape");
    GLM_CHECK_ARG(k <= COMBINE_MAX_K && T <= 0x7fffffff, GLM_ERR_UNSUPPORTED, "glm_moe_combine_scaled: k > %d", COMBINE_MAX_K);
    if (T == 0) return GLM_OK;
    const int vec_ok = H % 8 == 0 && ldy % 8 == 0 && glm_aligned16(y) && glm_aligned16(out) &&
                       (shared_rows == nullptr || (ld_shared % 8 == 0 && glm_aligned16(shared_rows)));
    const int work = vec_ok ? H / 8 : H;
    int slabs = (work + 255) / 256;
    while (slabs > 1 && (int64_t)T * slabs > 2048) slabs = (slabs + 1) / 2;
    hipLaunchKernelGGL(moe_combine_kernel, dim3((unsigned)T, (unsigned)slabs), dim3(256), 0, glm_s(stream),
                       (const bf16_t*)y, ldy, pair_pos, (bf16_t*)out, k, H, (int64_t)-1, vec_ok, (const bf16_t*)row_scale,
                       (const bf16_t*)shared_rows, ld_shared);
    GLM_CHECK_LAUNCH("glm_moe_combine_scaled");
    return GLM_OK;
}

// ------------------------------------------------------------------------------------------
// Row pack / unpack of the expert-parallel dispatch: dst[i] = src[idx[i]] (gather) or dst[idx[i]] = src[i] (scatter),
// rows of `cols` bf16 (16-byte vectors when everything is aligned); idx < 0 skips the row.
namespace {
template <bool SCATTER>
__global__ __launch_bounds__(256) void permute_rows_kernel(const bf16_t* __restrict__ src, int64_t ld_src,
                                                           bf16_t* __restrict__ dst, int64_t ld_dst,
                                                           const int32_t* __restrict__ idx, int cols, int vec_ok) {
    const int64_t i = blockIdx.x;
    const int64_t j = idx[i];
    if (j < 0) return;
    const bf16_t* s = src + (SCATTER ? i : j) * ld_src;
    bf16_t* d = dst + (SCATTER ? j : i) * ld_dst;
    if (vec_ok) {
        for (int c = threadIdx.x; c < (cols >> 3); c += 256)
            reinterpret_cast<u32x4_t*>(d)[c] = reinterpret_cast<const u32x4_t*>(s)[c];
    } else {
        for (int c = threadIdx.x; c < cols; c += 256) d[c] = s[c];
    }
}
}  // namespace

static int permute_rows(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, const int32_t* idx, int64_t n,
                        int cols, bool scatter, glm_stream_t stream, const char* name) {
    GLM_CHECK_ARG(src && dst && idx, GLM_ERR_INVALID, "%s: null pointer", name);
    GLM_CHECK_ARG(n >= 0 && cols > 0 && ld_src >= cols && ld_dst >= cols && n <= 0x7fffffff, GLM_ERR_INVALID,
                  "%s: bad shape", name);
    if (n == 0) return GLM_OK;
    const int vec_ok = cols % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0 && glm_aligned16(src) && glm_aligned16(dst);
    if (scatter)
        hipLaunchKernelGGL(permute_rows_kernel<true>, dim3((unsigned)n), dim3(256), 0, glm_s(stream), (const bf16_t*)src,
                           ld_src, (bf16_t*)dst, ld_dst, idx, cols, vec_ok);
    else
        hipLaunchKernelGGL(permute_rows_kernel<false>, dim3((unsigned)n), dim3(256), 0, glm_s(stream), (const bf16_t*)src,
                           ld_src, (bf16_t*)dst, ld_dst, idx, cols, vec_ok);
    GLM_CHECK_LAUNCH(name);
    return GLM_OK;
}

extern "C" int glm_gather_rows(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, const int32_t* idx, int64_t n,
                               int cols, glm_stream_t stream) {
    return permute_rows(src, ld_src, dst, ld_dst, idx, n, cols, false, stream, "glm_gather_rows");
}
extern "C" int glm_scatter_rows(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, const int32_t* idx, int64_t n,
                                int cols, glm_stream_t stream) {
    return permute_rows(src, ld_src, dst, ld_dst, idx, n, cols, true, stream, "glm_scatter_rows");
}

// ------------------------------------------------------------------------------------------
// Expert-parallel dispatch plan with FIXED-capacity slots (N3 / BASELINE config 4, graph-capturable): everything the
// all-to-all needs, computed on the device from the replicated routing — no count exchange, no host read.
//   ranks: rank r owns experts [r El, (r+1) El), El = E / ep, and the token slice [r per, min(T, (r+1) per)), per = ceil(T/ep).
//   send buffer of a rank: [ep destinations][cap rows]; the rows for destination d sit at the head of slot d in
//   (expert ascending, then pair index ascending) order.  The all-to-all moves whole slots (equal splits), so the
//   receive buffer is [ep sources][cap rows] with source s's rows for MY experts at the head of slot s, in that order.
// One 1024-thread workgroup (the routing of a decode step is a few thousand pairs):
//   send_gather[ep*cap]  row of the FULL hidden-state matrix to copy into each send slot row, -1 = unused
//   slot_scale [ep*cap]  routing weight of the pair in that slot row (bf16; unused rows untouched)
//   pair_slot  [Tl*k]    slot row of pair (t - lo, j): where its expert output comes back to after the return trip
//   recv_offs  [El+1]    rows per local expert over all sources (prefix sums): expert_offsets of the grouped GEMMs
//   recv_gidx  [ep*cap]  receive-buffer row of expert-ordered row i (sorted_token of gate|up, scatter index on the way
//                        back), -1 beyond recv_offs[El]
//   overflow   [1]       max over (source, destination) of the row count when it exceeds cap, else 0 (cap = Tl *
//                        min(k, El) can never overflow; a smaller cap is the caller's bet)
namespace {
constexpr int EP_MAX_PAIRS = 32768;
constexpr int EP_MAX_BINS = 2048;  // ep * E
__global__ __launch_bounds__(1024) void ep_plan_kernel(const int64_t* __restrict__ topk_idx, const bf16_t* __restrict__ topk_w,
                                                       int T, int k, int E, int ep, int rank, int cap,
                                                       int32_t* __restrict__ send_gather, bf16_t* __restrict__ slot_scale,
                                                       int32_t* __restrict__ pair_slot, int32_t* __restrict__ recv_offs,
                                                       int32_t* __restrict__ recv_gidx, int32_t* __restrict__ overflow) {
    __shared__ int cnt[EP_MAX_BINS];        // [source][expert] pair counts
    __shared__ unsigned short eidx[EP_MAX_PAIRS / 2];  // my slice's expert ids (<= 16384 pairs per slice)
    __shared__ int ovf;
    const int tid = threadIdx.x;
    const int El = E / ep, per = (T + ep - 1) / ep;
    const int lo = min(T, rank * per), hi = min(T, lo + per), Tl = hi - lo;
    for (int i = tid; i < ep * E; i += 1024) cnt[i] = 0;
    for (int i = tid; i < ep * cap; i += 1024) {
        send_gather[i] = -1;
        recv_gidx[i] = -1;
    }
    if (tid == 0) ovf = 0;
    __syncthreads();
    const int P = T * k;
    for (int p = tid; p < P; p += 1024) {
        const int e = (int)topk_idx[p];
        const int t = p / k;
        if (e >= 0 && e < E) atomicAdd(&cnt[(t / per) * E + e], 1);
        if (t >= lo && t < hi) eidx[p - lo * k] = (unsigned short)(e >= 0 && e < E ? e : 0xffff);
    }
    __syncthreads();
    // ---- send side: thread e walks my slice's pairs in order and hands out the rows of expert e ----
    if (tid < E) {
        const int e = tid, d = e / El;
        int base = 0;
        for (int e2 = d * El; e2 < e; ++e2) base += cnt[rank * E + e2];
        int n = 0;
        for (int p = 0; p < Tl * k; ++p) {
            if (eidx[p] != e) continue;
            const int within = base + n++;
            if (within < cap) {
                const int slot = d * cap + within;
                send_gather[slot] = lo + p / k;
                slot_scale[slot] = topk_w[(int64_t)lo * k + p];
                pair_slot[p] = slot;
            } else {
                pair_slot[p] = -1;  // dropped (only with a capacity below the worst case); overflow reports it
            }
        }
    }
    // pairs with an invalid expert id never get a slot
    for (int p = tid; p < Tl * k; p += 1024)
        if (eidx[p] == 0xffff) pair_slot[p] = -1;
    // ---- capacity check: rows of every (source, destination) pair ----
    for (int i = tid; i < ep * ep; i += 1024) {
        const int s = i / ep, d = i % ep;
        int n = 0;
        for (int e2 = d * El; e2 < (d + 1) * El; ++e2) n += cnt[s * E + e2];
        if (n > cap) atomicMax(&ovf, n);
    }
    // ---- receive side: rows of (source s, local expert el) that fit = what the sender really put into its slot ----
    __shared__ int within_t[1024], fit_t[1024];  // [s * El + el]
    if (tid < ep * El) {
        const int s = tid / El, el = tid % El;
        int within = 0;
        for (int e3 = 0; e3 < el; ++e3) within += cnt[s * E + rank * El + e3];
        within_t[tid] = within;
        fit_t[tid] = max(0, min(cnt[s * E + rank * El + el], cap - within));
    }
    __syncthreads();
    // thread el lists, source by source, where its rows land in the receive buffer (expert-major order)
    if (tid < El) {
        int start = 0;  // rows of the local experts before this one, over all sources
        for (int e2 = 0; e2 < tid; ++e2)
            for (int s = 0; s < ep; ++s) start += fit_t[s * El + e2];
        recv_offs[tid] = start;
        int row = start;
        for (int s = 0; s < ep; ++s) {
            const int n = fit_t[s * El + tid], w0 = within_t[s * El + tid];
            for (int i = 0; i < n; ++i) recv_gidx[row++] = s * cap + w0 + i;
        }
        if (tid == El - 1) recv_offs[El] = row;
    }
    __syncthreads();
    if (tid == 0) overflow[0] = ovf;
}
}  // namespace

extern "C" int glm_ep_plan(const int64_t* topk_idx, const void* topk_w, int64_t T, int k, int E, int ep, int rank, int cap,
                           int32_t* send_gather, void* slot_scale, int32_t* pair_slot, int32_t* recv_offs,
                           int32_t* recv_gidx, int32_t* overflow, glm_stream_t stream) {
    GLM_CHECK_ARG(topk_idx && topk_w && send_gather && slot_scale && pair_slot && recv_offs && recv_gidx && overflow,
                  GLM_ERR_INVALID, "glm_ep_plan: null pointer");
    GLM_CHECK_ARG(T >= 1 && k >= 1 && E >= 1 && ep >= 1 && rank >= 0 && rank < ep && E % ep == 0 && cap >= 1,
                  GLM_ERR_INVALID, "glm_ep_plan: bad arguments (T=%ld k=%d E=%d ep=%d rank=%d cap=%d)", (long)T, k, E, ep, rank, cap);
    const int64_t per = (T + ep - 1) / ep;
    GLM_CHECK_ARG(E <= 1024 && (int64_t)ep * E <= EP_MAX_BINS && T * k <= 0x7fffffff && per * k <= EP_MAX_PAIRS / 2 &&
                      (int64_t)ep * cap <= 0x7fffffff,
                  GLM_ERR_UNSUPPORTED, "glm_ep_plan: needs ep*E <= %d and ceil(T/ep)*k <= %d (decode-sized steps)", EP_MAX_BINS,
                  EP_MAX_PAIRS / 2);
    hipLaunchKernelGGL(ep_plan_kernel, dim3(1), dim3(1024), 0, glm_s(stream), topk_idx, (const bf16_t*)topk_w, (int)T, k, E, ep,
                       rank, cap, send_gather, (bf16_t*)slot_scale, pair_slot, recv_offs, recv_gidx, overflow);
    GLM_CHECK_LAUNCH("glm_ep_plan");
    return GLM_OK;
}

extern "C" int glm_moe_combine_add_rmsnorm(const void* y, int64_t ldy, const int32_t* pair_pos, int64_t T, int k, int H,
                                           int64_t shared_base, const void* residual_in, const void* weight,
                                           void* out, void* residual_out, float eps, glm_stream_t stream) {
    GLM_CHECK_ARG(y && pair_pos && residual_in && weight && out && residual_out, GLM_ERR_INVALID,
                  "glm_moe_combine_add_rmsnorm: null pointer");
    GLM_CHECK_ARG(T >= 0 && k > 0 && H > 0 && ldy >= H, GLM_ERR_INVALID, "glm_moe_combine_add_rmsnorm: bad shape");
    GLM_CHECK_ARG(k <= COMBINE_MAX_K && H % 8 == 0 && H <= 8 * CN_NT * 4 && ldy % 8 == 0 && glm_aligned16(y) &&
                      glm_aligned16(residual_in) && glm_aligned16(weight) && glm_aligned16(out) &&
                      glm_aligned16(residual_out) && T <= 0x7fffffff,
                  GLM_ERR_UNSUPPORTED, "glm_moe_combine_add_rmsnorm: needs k<=%d, H%%8==0, H<=%d, 16B-aligned rows (H=%d)",
                  COMBINE_MAX_K, 8 * CN_NT * 4, H);
    if (T == 0) return GLM_OK;
    const int vpt = (H / 8 + CN_NT - 1) / CN_NT;
    hipStream_t st = glm_s(stream);
    const dim3 grid((unsigned)T), block(CN_NT);
#define GLM_CN_LAUNCH(V)                                                                                              \
    hipLaunchKernelGGL((moe_combine_add_rmsnorm_kernel<V>), grid, block, 0, st, (const bf16_t*)y, ldy, pair_pos, k,   \
                       shared_base, (const bf16_t*)residual_in, (const bf16_t*)weight, (bf16_t*)out,                  \
                       (bf16_t*)residual_out, H, eps)
    if (vpt <= 1) GLM_CN_LAUNCH(1);
    else if (vpt <= 2) GLM_CN_LAUNCH(2);
    else GLM_CN_LAUNCH(4);
#undef GLM_CN_LAUNCH
    GLM_CHECK_LAUNCH("glm_moe_combine_add_rmsnorm");
    return GLM_OK;
}
