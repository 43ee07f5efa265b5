// fused_rows.h — row-level device functions shared by a stand-alone launch and the kernel that absorbs it as a prologue.
// Sharing the code (and the 512-thread mapping it is written for) is what makes the fused launch bit-identical to the
// pair of launches it replaces.
#pragma once
#include "common.h"

#define GLM_COMBINE_MAX_K 64
constexpr int GLM_CN_NT = 512;  // threads of a workgroup that runs glm_combine_add_rmsnorm_row

// MoE combine of token t (moe.py:133-171: bf16 adds in ascending sorted-row position, then the shared row, one rounding)
// followed by the NEXT layer's input add-RMSNorm (layernorm.py:29-40):
//     x = bf16(combine);  s = f32(x) + f32(residual_in[t]);  residual_out[t] = bf16(s);  y = bf16(bf16(s * rstd) * w)
// Executed by ALL GLM_CN_NT threads of the workgroup.  Outputs, each optional (nullptr = skip): out_g / res_out_g rows in
// global memory, out_lds = this token's normalised row in LDS (a later __syncthreads() of the caller publishes it).
// `red` = GLM_CN_NT / 64 floats of LDS scratch.  The caller synchronises before the next call (red and the position list
// are reused).
template <int VPT>
__device__ __forceinline__ void glm_combine_add_rmsnorm_row(const bf16_t* __restrict__ y, int64_t ldy,
                                                            const int32_t* __restrict__ pair_pos, int k,
                                                            int64_t shared_base, const bf16_t* res_in,
                                                            const bf16_t* __restrict__ w, bf16_t* out_g, bf16_t* res_out_g,
                                                            bf16_t* out_lds, int H, float eps, int64_t t, float* red) {
    __shared__ int pos[GLM_COMBINE_MAX_K];
    __shared__ int npos;
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
    const int nvec = H >> 3;
    float s[VPT][8];
    float ss = 0.f;
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int c = threadIdx.x + v * GLM_CN_NT;
        if (c < nvec) {
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = 0.f;
            // the rounding chain is serial, the row loads are not: 8 rows are in flight at a time (clamped re-reads of
            // the last row fill the batch), together with the shared row and the residual
            u32x4_t sv = {0, 0, 0, 0};
            if (shared_base >= 0) sv = reinterpret_cast<const u32x4_t*>(y + (shared_base + t) * ldy)[c];
            const u32x4_t rv = reinterpret_cast<const u32x4_t*>(res_in + t * H)[c];
            for (int j0 = 0; j0 < n; j0 += 8) {
                u32x4_t vv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    vv[j] = reinterpret_cast<const u32x4_t*>(y + (int64_t)pos[min(j0 + j, n - 1)] * ldy)[c];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j0 + j < n) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc[2 * q] = rbf(acc[2 * q] + bflo(vv[j][q]));
                            acc[2 * q + 1] = rbf(acc[2 * q + 1] + bfhi(vv[j][q]));
                        }
                    }
            }
            if (shared_base >= 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[2 * q] = acc[2 * q] + bflo(sv[q]);
                    acc[2 * q + 1] = acc[2 * q + 1] + bfhi(sv[q]);
                }
            }
            u32x4_t ov;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s[v][2 * q] = rbf(acc[2 * q]) + bflo(rv[q]);          // rbf: the combine's own output rounding
                s[v][2 * q + 1] = rbf(acc[2 * q + 1]) + bfhi(rv[q]);
                ov[q] = pack2bf(s[v][2 * q], s[v][2 * q + 1]);
            }
            if (res_out_g) reinterpret_cast<u32x4_t*>(res_out_g + t * H)[c] = ov;
#pragma unroll
            for (int i = 0; i < 8; ++i) ss += s[v][i] * s[v][i];
        }
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < GLM_CN_NT / 64; ++i) tot += red[i];
    const float rstd = 1.0f / sqrtf(tot / (float)H + eps);
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int c = threadIdx.x + v * GLM_CN_NT;
        if (c < nvec) {
            const u32x4_t wv = reinterpret_cast<const u32x4_t*>(w)[c];
            u32x4_t o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = rbf(s[v][2 * i] * rstd) * bflo(wv[i]);
                const float b = rbf(s[v][2 * i + 1] * rstd) * bfhi(wv[i]);
                o[i] = pack2bf(a, b);
            }
            if (out_g) reinterpret_cast<u32x4_t*>(out_g + t * H)[c] = o;
            if (out_lds) reinterpret_cast<u32x4_t*>(out_lds)[c] = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// write-through (sc1) accesses for data handed between workgroups of ONE launch (Guideline 16 R1): stored this way and
// drained before the ticket, read this way after it, the hand-off needs neither the producer's agent-scope release fence
// nor the consumer's acquire (≈1.7 us each on this chip)
__device__ __forceinline__ unsigned ld_sc1_u32(const void* p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_sc1_u64(const void* p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1_u32(void* p, unsigned v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1_u64(void* p, unsigned long long v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


constexpr int ROUTER_MAX_E = 256;
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// R2..R7 for ONE token by ONE wave (moe.py:28-50): lane owns experts lane + 64 i.  lg: the token's E logits (fp32 values
// of bf16 precision), bias_l: the E correction biases, both in LDS; keys_scratch: 64 words of LDS (E <= 64 path),
// wl_scratch: k bf16 of LDS (write-through output form only).  Shared by the decode router kernel (its wave 0) and by
// router_select_kernel (prefill: logits from a GEMM) so that both select bit-identically.
template <int EPL>
__device__ __forceinline__ void router_select_wave(const float* lg, const float* bias_l, unsigned* keys_scratch,
                                                   bf16_t* wl_scratch, int lane, int E, int k, float rsf, int64_t t,
                                                   int64_t* __restrict__ topk_idx, bf16_t* __restrict__ topk_w,
                                                   bf16_t* __restrict__ logits_out, bool write_through,
                                                   int32_t* a1_offs = nullptr, int32_t* a1_tok = nullptr,
                                                   bf16_t* a1_scale = nullptr, int32_t* a1_pos = nullptr,
                                                   int a1_shared = 0) {
    // a1_*: single-token launch (T = 1) that also wants glm_moe_align's structure: with one token the bucketing IS the
    // selection — sorted row j = the j-th selected expert (ascending id), so this wave writes expert_offsets /
    // sorted_token / sorted_scale / pair_pos itself and the launch needs neither the global ticket nor the bucketing pass.
    // wave 0: R2..R7.  lane owns experts lane + 64*i.
    float sc[EPL], ch[EPL];
    bool sel[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        const int e = lane + 64 * i;
        sel[i] = false;
        if (e < E) {
            const float l = lg[e];
            if (logits_out) logits_out[t * E + e] = f2bf(l);
            sc[i] = rbf(sigmoid_f(l));  // R2: bf16 sigmoid
            ch[i] = sc[i] + bias_l[e];  // R3: fp32 (bias buffer is fp32)
        } else {
            sc[i] = 0.f;
            ch[i] = -INFINITY;
        }
    }
    // R4: k rounds of wave arg-max (ties → lowest expert id).  NaN scores are demoted to -inf so the
    // comparisons stay a total order and every lane agrees on the winner.
#pragma unroll
    for (int i = 0; i < EPL; ++i)
        if (!(ch[i] == ch[i])) ch[i] = -INFINITY;
    // Each candidate becomes one 64-bit key (order-preserving image of the score, then ~id so that the lowest id wins
    // a tie); "no candidate" is key 0.  The wave maximum runs on DPP row permutations and the two lane-swap
    // instructions instead of 12 LDS round trips (ds_bpermute) per round.
    unsigned kh[EPL], kl[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        const int e = lane + 64 * i;
        const unsigned u = __float_as_uint(ch[i] + 0.0f);  // -0 -> +0: equal scores must give equal keys
        kh[i] = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        kl[i] = ~(unsigned)e;
        if (e >= E) kh[i] = 0u, kl[i] = 0u;
    }
    auto take_max = [](unsigned& hi, unsigned& lo, unsigned ohi, unsigned olo) {
        const bool take = ohi > hi || (ohi == hi && olo > lo);
        hi = take ? ohi : hi;
        lo = take ? olo : lo;
    };
    if (E <= 64) {
        // one expert per lane: selected <=> fewer than k experts precede it in the total order (key descending, then id
        // ascending) — the same set the k arg-max rounds below pick, in ~1/4 of the instructions.  Keys go through LDS
        // (the x row is dead) and come back as broadcast 16-byte reads.
        unsigned* keys = keys_scratch;
        keys[lane] = kh[0];  // lanes >= E hold key 0, below every real key
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned mine = kh[0];
        int rank = 0;
#pragma unroll
        for (int j4 = 0; j4 < 16; ++j4) {
            const u32x4_t kk = reinterpret_cast<const u32x4_t*>(keys)[j4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = 4 * j4 + q;
                rank += (kk[q] > mine || (kk[q] == mine && j < lane)) ? 1 : 0;
            }
        }
        sel[0] = lane < E && rank < k;
    } else
    for (int r = 0; r < k; ++r) {
        unsigned hi = 0u, lo = 0u;
#pragma unroll
        for (int i = 0; i < EPL; ++i)
            if (!sel[i]) take_max(hi, lo, kh[i], kl[i]);
#define GLM_DPP_STEP(CTRL)                                                                                      \
    take_max(hi, lo, (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, CTRL, 0xF, 0xF, true),                   \
             (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, CTRL, 0xF, 0xF, true))
        GLM_DPP_STEP(0xB1);   // quad_perm [1,0,3,2]: lane ^ 1
        GLM_DPP_STEP(0x4E);   // quad_perm [2,3,0,1]: lane ^ 2
        GLM_DPP_STEP(0x141);  // row_half_mirror: the other quad of the 8
        GLM_DPP_STEP(0x140);  // row_mirror: the other 8 of the row
#undef GLM_DPP_STEP
        {
            const u32x2_t a = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
            const u32x2_t b = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
            hi = a[0], lo = b[0];
            take_max(hi, lo, a[1], b[1]);
            const u32x2_t c2 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
            const u32x2_t d = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
            hi = c2[0], lo = d[0];
            take_max(hi, lo, c2[1], d[1]);
        }
        const int bi = (int)~lo;  // wave-uniform; k <= E guarantees a candidate
#pragma unroll
        for (int i = 0; i < EPL; ++i)
            if (lane + 64 * i == bi) sel[i] = true;
    }
    // R5/R6: weights = scores (without bias) of the selected experts, renormalised in bf16
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < EPL; ++i)
        if (sel[i]) part += sc[i];
    const float den = rbf(rbf(wave_sum(part)) + 1e-20f);
    // output in ascending expert id: position = #selected experts with a smaller id.  When another workgroup of this
    // launch buckets the pairs (fuse_align, k even) the routing is stored write-through: ids as 8-byte sc1 stores, the
    // weights staged in LDS and stored as aligned 4-byte words
    const bool wt = write_through && (k & 1) == 0;
    bf16_t* wl = wl_scratch;  // the logits live in registers by now; 2 k <= 4 E bytes
    int base = 0;
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        const unsigned long long m = __ballot(sel[i]);
        const int before = base + __popcll(m & ((1ull << lane) - 1ull));  // selected experts with a smaller id
        if (a1_offs && lane + 64 * i < E) a1_offs[lane + 64 * i] = before;
        if (sel[i]) {
            const int pos = before;
            if (pos < k) {
                const bf16_t wv = f2bf(rbf(sc[i] / den) * rsf);  // R6 then R7
                if (wt) {
                    st_sc1_u64(topk_idx + t * k + pos, (unsigned long long)(lane + 64 * i));
                    wl[pos] = wv;
                } else {
                    topk_idx[t * k + pos] = lane + 64 * i;
                    topk_w[t * k + pos] = wv;
                }
                if (a1_offs) {
                    a1_tok[pos] = 0;
                    a1_scale[pos] = wv;
                    a1_pos[pos] = pos;
                }
            }
        }
        base += __popcll(m);
    }
    if (wt && lane < (k >> 1)) st_sc1_u32(topk_w + t * k + 2 * lane, reinterpret_cast<const unsigned*>(wl)[lane]);
    if (a1_offs && lane == 0) {
        a1_offs[E] = base;  // == k: start of the shared group
        if (a1_shared) {
            a1_offs[E + 1] = base + 1;
            a1_tok[base] = 0;
            a1_scale[base] = 0x3F80;  // bf16 1.0
        }
    }
}

