// prefill_attn.hip — varlen causal prefill attention (a9, reference:
// nanovllm/layers/attention.py:85-100 → flash_attn_varlen_func(causal=True[, block_table])).
// gfx950 only, head_dim 128.
//
// Flash-style: the score matrix is never materialised.  A workgroup owns 64*FA_QT query rows of
// one (sequence, query head); its 4 waves take 16*FA_QT rows each and walk the key tiles (64 keys) that the bottom-right-aligned causal mask leaves visible.  K/V
// tiles are staged once per workgroup by LDS-DMA with the same source-side swizzles as the decode kernel (K: chunk ^ (key & 15) for
// ds_read_b128 row reads, V: chunk ^ ((key & 7) << 1) for ds_read_b64_tr_b16 transposed reads),
// double-buffered.  Math per wave is the decode kernel's: S^T = K·Q^T with keys on the MFMA row
// axis and the wave's 16 query rows on the lane axis, fp32 online softmax per lane column, then
// O^T += V^T·P^T straight from registers.
//
// K/V come either from contiguous [Tk, nkv, hd] tensors or from the paged cache through
// block_tables (prefix-cache hit, attention.py:86-87).
#include "common.h"

namespace {

constexpr int FA_HD = 128, FA_ROWB = 256;
constexpr int FA_QT = 1;                   // 16-row query tiles per wave (2 pushes the kernel past 256 registers:
                                           // one wave per SIMD, 232 vs 366 TFLOP/s)
constexpr int FA_BQ = 64 * FA_QT, FA_BK = 64;
constexpr int FA_TILE = FA_BK * FA_ROWB;   // 16 KiB per K or V tile
constexpr int FA_STAGE = 2 * FA_TILE;
constexpr int FA_MAX_BT = 4096;
constexpr int FA_LDS = 2 * FA_STAGE + FA_MAX_BT * 4;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void_t*)gsrc, (lds_void_t*)lds_wave_base, 16, 0, 0);
}
template <int OFF>
__device__ __forceinline__ void lds_read_b128(u32x4_t& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_read_tr16_b64(u32x2_t& dst, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
// v_max3_f32 without the quieting v_max x,x that fmaxf() gets for each MFMA output
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(const lds_void_t*)p;
}

template <bool PAGED>
// launch bound of 2 waves per SIMD caps the kernel at 256 registers, which makes hipcc pick the VGPR form of the
// MFMAs: with the default (512) the accumulators live in AGPRs and every slab pays ~40 v_accvgpr copies
__global__ __launch_bounds__(256, 2) void prefill_attn_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const int32_t* __restrict__ cu_q, const int32_t* __restrict__ cu_k,
    const int32_t* __restrict__ block_tables, int64_t bt_stride, bf16_t* __restrict__ out, int nh,
    int nkv, int block_size, float scale_log2e) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* bt_lds = reinterpret_cast<int*>(smem + 2 * FA_STAGE);
    const int qblk = blockIdx.x, head = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int q0 = cu_q[b], lq = cu_q[b + 1] - q0;
    const int k0 = cu_k[b], lk = cu_k[b + 1] - k0;
    const int qbase = qblk * FA_BQ;
    if (qbase >= lq) return;  // uniform
    const int kvh = head / (nh / nkv);
    const int off = lk - lq;  // causal: query i sees keys j <= i + off
    // keys visible to this workgroup's last row
    const int kend = min(lk, min(lq, qbase + FA_BQ) + off);  // exclusive
    const int ntiles = (kend + FA_BK - 1) / FA_BK;

    if (PAGED) {
        const int nblk = (kend + block_size - 1) / block_size;
        for (int i = tid; i < nblk; i += 256) bt_lds[i] = block_tables[b * bt_stride + i];
    }

    // Q^T fragments: lane (c,g), tile qt = query row qbase + 32*wid + 16*qt + c, k-step ks → hd [32ks+8g, +8)
    int qi[FA_QT];
    bf16x8_t qf[FA_QT][4];
#pragma unroll
    for (int qt = 0; qt < FA_QT; ++qt) {
        qi[qt] = qbase + (16 * FA_QT) * wid + 16 * qt + c;
        const bool ok = qi[qt] < lq;
        const bf16_t* qrow = q + ((int64_t)(q0 + (ok ? qi[qt] : 0)) * nh + head) * FA_HD;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            u32x4_t t = {0, 0, 0, 0};
            if (ok) t = *reinterpret_cast<const u32x4_t*>(qrow + 32 * ks + 8 * g);
            qf[qt][ks] = __builtin_bit_cast(bf16x8_t, t);
        }
    }
    __syncthreads();
#pragma unroll
    for (int qt = 0; qt < FA_QT; ++qt)
        asm volatile("" : "+v"(qf[qt][0]), "+v"(qf[qt][1]), "+v"(qf[qt][2]), "+v"(qf[qt][3]));

    // staging: wave w issues instructions i=0..3 for K and V; instruction covers tile rows
    // (4w+i)*4 .. +3 (4 keys x 256 B); lane → row +(lane>>4), position lane&15
    auto stage = [&](int buf, int t) {
        char* kd = smem + buf * FA_STAGE + wid * 4 * 1024;
        char* vd = kd + FA_TILE;
        const int r4 = lane >> 4, pch = lane & 15;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wid * 4 + i) * 4 + r4;
            int key = t * FA_BK + row;
            key = key < lk ? key : lk - 1;  // masked keys re-read a valid row
            int64_t tok;
            if (PAGED) {
                const int pb = bt_lds[key / block_size];
                tok = (int64_t)pb * block_size + key % block_size;
            } else {
                tok = (int64_t)k0 + key;
            }
            const int64_t ro = (tok * nkv + kvh) * FA_HD;
            glds16(k + ro + ((pch ^ (row & 15)) << 3), kd + i * 1024);
            glds16(v + ro + ((pch ^ ((row & 7) << 1)) << 3), vd + i * 1024);
        }
    };

    unsigned k_rel[4], v_rel[8];
    {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) k_rel[ks] = c * FA_ROWB + (((4 * ks + g) ^ c) << 4);
        const int qq = (lane >> 2) & 3, pp = lane & 3;
        const int row = 4 * g + qq;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt)
            v_rel[dt] = FA_TILE + row * FA_ROWB + (((2 * dt + (pp >> 1)) ^ ((row & 7) << 1)) << 4) + ((pp & 1) << 3);
    }

    f32x4_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const int jmax = qi[0] + off;                     // last visible key of this lane's query row
    const int row_first = qbase + 16 * wid;           // first query row of the wave

    // The loop is software-pipelined inside the wave over 32-key slabs: the S MFMAs of slab i+1 are
    // issued before the softmax VALU work of slab i, so the matrix pipe runs under the exp/max/convert
    // instructions of the same wave; V fragments of slab i are fetched from LDS while that softmax
    // runs.  One workgroup barrier per 64-key tile.
    u32x4_t kf[2][4];
    u32x2_t vf[8][2];
    auto read_k = [&](unsigned hb) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            lds_read_b128<0>(kf[0][ks], hb + k_rel[ks]);
            lds_read_b128<4096>(kf[1][ks], hb + k_rel[ks]);
        }
    };
    auto read_v = [&](unsigned hb, int dlo, int dhi) {
#pragma unroll
        for (int dt = dlo; dt < dhi; ++dt) {
            lds_read_tr16_b64<0>(vf[dt][0], hb + v_rel[dt]);
            lds_read_tr16_b64<4096>(vf[dt][1], hb + v_rel[dt]);
        }
    };
#define FA_K_REGS                                                                                          \
    "+v"(kf[0][0]), "+v"(kf[0][1]), "+v"(kf[0][2]), "+v"(kf[0][3]), "+v"(kf[1][0]), "+v"(kf[1][1]),       \
        "+v"(kf[1][2]), "+v"(kf[1][3])
#define FA_V_REGS                                                                                          \
    "+v"(vf[0][0]), "+v"(vf[0][1]), "+v"(vf[1][0]), "+v"(vf[1][1]), "+v"(vf[2][0]), "+v"(vf[2][1]),       \
        "+v"(vf[3][0]), "+v"(vf[3][1]), "+v"(vf[4][0]), "+v"(vf[4][1]), "+v"(vf[5][0]), "+v"(vf[5][1]),   \
        "+v"(vf[6][0]), "+v"(vf[6][1]), "+v"(vf[7][0]), "+v"(vf[7][1])
    auto s_mfma = [&](f32x4_t (&s)[2]) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            s[tt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                s[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[tt][ks]),
                                                               qf[0][ks], s[tt], 0, 0, 0);
        }
    };
    // online softmax of one slab (keys [kbase, kbase+32)); returns the bf16 P^T fragment and leaves acc
    // rescaled.  The running maximum lives in the scaled (log2) domain: p = exp2(s*c - m).
    auto softmax = [&](f32x4_t (&s)[2], int kbase) -> bf16x8_t {
        // masking is only needed on slabs that touch the causal diagonal or the sequence end for this
        // wave's rows: the test is wave-uniform and false for almost every slab
        const bool need_mask = kbase + 31 > row_first + off || kbase + 31 >= lk;
        if (need_mask) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = kbase + 4 * g + 16 * tt + r;
                    if (j > jmax || j >= lk) s[tt][r] = -INFINITY;
                }
        }
        float mloc = vmax3(vmax3(vmax3(s[0][0], s[0][1], s[0][2]), s[0][3], s[1][0]), s[1][1], s[1][2]);
        mloc = vmax3(mloc, s[1][3], s[1][3]);
        {   // max over the four 16-lane groups that share a query column: VALU lane swaps, no LDS trip
            const unsigned u = __float_as_uint(mloc);
            const u32x2_t a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
            mloc = vmax3(__uint_as_float(a[0]), __uint_as_float(a[1]), __uint_as_float(a[1]));
            const unsigned w = __float_as_uint(mloc);
            const u32x2_t b2 = __builtin_amdgcn_permlane32_swap(w, w, false, false);
            mloc = vmax3(__uint_as_float(b2[0]), __uint_as_float(b2[1]), __uint_as_float(b2[1]));
        }
        const float ms = mloc * scale_log2e;
        const float m_new = vmax3(m_run, ms, ms);
        // a row with nothing visible yet keeps m_new = -inf: use 0 as the reference point so that
        // exp2(-inf - 0) = 0 and no NaN appears
        const float m_use = m_new == -INFINITY ? 0.f : m_new;
        float psum = 0.f;
        bf16x8_t pf;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(s[tt][r], scale_log2e, -m_use));
                psum += pv;
                pf[4 * tt + r] = (__bf16)pv;
            }
        // rescale only when some row's maximum actually grew (wave-uniform; after the first few
        // tiles it rarely does)
        float alpha = 1.0f;
        if (__any(m_new > m_run)) {
            alpha = __builtin_amdgcn_exp2f(m_run - m_use);
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) acc[dt] *= alpha;
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        return pf;
    };
    auto pv_mfma = [&](bf16x8_t pf) {
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            const u32x4_t vv = {vf[dt][0][0], vf[dt][0][1], vf[dt][1][0], vf[dt][1][1]};
            acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vv), pf, acc[dt], 0, 0, 0);
        }
    };

    if (ntiles > 0) {
        f32x4_t s_cur[2], s_nxt[2];
        stage(0, 0);
        if (ntiles > 1) {
            stage(1, 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        read_k(lds_addr(smem));
        asm volatile("s_waitcnt lgkmcnt(0)" : FA_K_REGS : : "memory");
        s_mfma(s_cur);
        for (int t = 0; t < ntiles; ++t) {
            const unsigned sb = lds_addr(smem + (t & 1) * FA_STAGE);
            // ---- slab 0 of tile t: S of slab 1 goes to the matrix pipe first
            read_k(sb + 32 * FA_ROWB);
            read_v(sb, 0, 4);  // the LDS counter holds 15: K (8 reads) + half of V (8 reads) in flight
            asm volatile("s_waitcnt lgkmcnt(8)" : FA_K_REGS : : "memory");
            __builtin_amdgcn_sched_barrier(0);
            s_mfma(s_nxt);
            read_v(sb, 4, 8);
            __builtin_amdgcn_sched_barrier(0);
            bf16x8_t pf = softmax(s_cur, t * FA_BK);
            asm volatile("s_waitcnt lgkmcnt(0)" : FA_V_REGS : : "memory");
            pv_mfma(pf);
            __builtin_amdgcn_sched_barrier(0);
            // ---- slab 1: its V fragments are the last LDS reads of this buffer
            read_v(sb + 32 * FA_ROWB, 0, 8);
            asm volatile("s_waitcnt lgkmcnt(0)" : FA_V_REGS : : "memory");
            if (t + 1 < ntiles) {
                // tile t+1 was issued a whole tile ago; after the barrier every wave is done with
                // buffer t&1, which tile t+2 may then overwrite
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (t + 2 < ntiles) stage(t & 1, t + 2);
                read_k(lds_addr(smem + ((t + 1) & 1) * FA_STAGE));
                asm volatile("s_waitcnt lgkmcnt(0)" : FA_K_REGS : : "memory");
                __builtin_amdgcn_sched_barrier(0);
                s_mfma(s_cur);
                __builtin_amdgcn_sched_barrier(0);
            }
            pf = softmax(s_nxt, t * FA_BK + 32);
            pv_mfma(pf);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef FA_K_REGS
#undef FA_V_REGS

    {
        float l = l_run;
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        if (qi[0] < lq) {
            const float inv = l > 0.f ? 1.0f / l : 0.f;
            bf16_t* orow = out + ((int64_t)(q0 + qi[0]) * nh + head) * FA_HD;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                u32x2_t o = {pack2bf(acc[dt][0] * inv, acc[dt][1] * inv),
                             pack2bf(acc[dt][2] * inv, acc[dt][3] * inv)};
                *reinterpret_cast<u32x2_t*>(orow + 16 * dt + 4 * g) = o;
            }
        }
    }
}

}  // namespace

extern "C" int glm_varlen_attn_prefill(const void* q, const void* k, const void* v,
                                       const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                                       const int32_t* block_tables, int64_t bt_stride, void* out,
                                       int64_t B, int max_seqlen_q, int nh, int nkv, int hd,
                                       int block_size, float scale, glm_stream_t stream) {
    GLM_CHECK_ARG(q && k && v && cu_seqlens_q && cu_seqlens_k && out, GLM_ERR_INVALID,
                  "glm_varlen_attn_prefill: null pointer");
    GLM_CHECK_ARG(B >= 0 && nh > 0 && nkv > 0 && nh % nkv == 0 && max_seqlen_q >= 0, GLM_ERR_INVALID,
                  "glm_varlen_attn_prefill: bad shape");
    GLM_CHECK_ARG(hd == FA_HD && nh <= 65535 && B <= 65535 &&
                      (block_tables == nullptr || (block_size > 0 && bt_stride > 0 && bt_stride <= FA_MAX_BT)),
                  GLM_ERR_UNSUPPORTED,
                  "glm_varlen_attn_prefill: needs head_dim=128 and <=%d blocks per sequence (hd=%d)", FA_MAX_BT, hd);
    GLM_CHECK_ARG(glm_aligned16(q) && glm_aligned16(k) && glm_aligned16(v) && glm_aligned16(out),
                  GLM_ERR_INVALID, "glm_varlen_attn_prefill: pointers must be 16-byte aligned");
    if (B == 0 || max_seqlen_q == 0) return GLM_OK;
    static bool attr_set[2] = {false, false};
    const int paged = block_tables != nullptr;
    if (!attr_set[paged]) {
        const void* fn = paged ? (const void*)prefill_attn_kernel<true> : (const void*)prefill_attn_kernel<false>;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, FA_LDS);
        if (e != hipSuccess) {
            glm_set_error("glm_varlen_attn_prefill: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return GLM_ERR_LAUNCH;
        }
        attr_set[paged] = true;
    }
    const dim3 grid((unsigned)((max_seqlen_q + FA_BQ - 1) / FA_BQ), (unsigned)nh, (unsigned)B);
    // K/V ring + the block-table slice (paged only): two workgroups per CU need <= 80 KiB each
    const size_t lds = 2 * FA_STAGE + (paged ? (size_t)bt_stride * 4 : 0);
    const float sl2 = scale * 1.4426950408889634f;
    if (paged)
        hipLaunchKernelGGL((prefill_attn_kernel<true>), grid, dim3(256), lds, glm_s(stream), (const bf16_t*)q,
                           (const bf16_t*)k, (const bf16_t*)v, cu_seqlens_q, cu_seqlens_k, block_tables,
                           bt_stride, (bf16_t*)out, nh, nkv, block_size, sl2);
    else
        hipLaunchKernelGGL((prefill_attn_kernel<false>), grid, dim3(256), lds, glm_s(stream), (const bf16_t*)q,
                           (const bf16_t*)k, (const bf16_t*)v, cu_seqlens_q, cu_seqlens_k, block_tables,
                           bt_stride, (bf16_t*)out, nh, nkv, block_size, sl2);
    GLM_CHECK_LAUNCH("glm_varlen_attn_prefill");
    return GLM_OK;
}
