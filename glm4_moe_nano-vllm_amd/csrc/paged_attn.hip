// paged_attn.hip — single-token decode attention over the paged KV cache (a8,
// reference: nanovllm/layers/attention.py:101-107 → flash_attn_with_kvcache).  gfx950 only.
//
// Roofline: HBM.  Algorithmic bytes = B * ctx * nkv * hd * 2 (K+V) * 2 B; every KV byte is read
// exactly once, MFMA/VALU time is <10 % of the stream time, so the design is about keeping
// 1-KiB-coalesced loads in flight on every CU:
//
//   * grid = (kv-split, kv-head, sequence); a workgroup owns ALL nh/nkv query heads of one
//     kv-head, so a KV row is fetched once for the whole GQA group (12 heads on GLM-4.5-Air).
//   * 4 waves per workgroup, each wave streams its own interleaved 32-key groups through a
//     private 2-stage LDS ring filled by LDS-DMA (global_load_lds_dwordx4: one instruction =
//     4 keys x 256 B, contiguous in HBM).  Waves never synchronise inside the loop — only the
//     wave's own counted s_waitcnt vmcnt orders DMA → ds_read.
//   * bank conflicts are removed by permuting the per-lane SOURCE chunk of each DMA (the LDS
//     image stays lane-linear): K image chunk' = chunk ^ (key & 15) for ds_read_b128 fragment
//     reads, V image chunk' = chunk ^ ((key & 7) << 1) for ds_read_b64_tr_b16 transposed reads.
//   * math: S^T = K·Q^T with mfma_f32_16x16x32_bf16 (keys on the MFMA row axis, query heads on
//     the lane axis), so each lane already holds 8 keys of ONE query head: the softmax needs two
//     cross-lane maxima per step and P^T feeds the second product O^T = V^T·P^T straight from
//     registers (k-slot j of lane group g ↔ key 16*(j>>2) + 4*g + (j&3)).
//   * per-wave online softmax (fp32), merged across the 4 waves through LDS at the end, and
//     across kv-splits by a small second kernel when num_splits > 1.
#include "common.h"
#include <string.h>

namespace {

constexpr int PA_HD = 128;               // head_dim supported by this kernel
constexpr int PA_ROWB = PA_HD * 2;       // 256 B per (key, kv-head) row
constexpr int PA_GK = 32;                // keys per wave step
constexpr int PA_STAGE = 2 * PA_GK * PA_ROWB;  // K image + V image = 16 KiB
constexpr int PA_NSTAGE = 2;
constexpr int PA_WAVES = 4;
constexpr int PA_RING = PA_NSTAGE * PA_STAGE;  // per wave
constexpr int PA_MAX_BT = 4096;                // block-table entries cached in LDS per workgroup
constexpr int PA_NEWROW = 2 * PA_ROWB;             // FUSED: the new token's rotated K row | V row, natural chunk order
constexpr int PA_LDS = PA_WAVES * PA_RING + PA_MAX_BT * 4 + PA_NEWROW;

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

// KV rows are read once per step: aux = 2 (nt) — 92.5 -> 84.5 us at C3 (6.37 TB/s); the same hint on the GEMMs'
// plain global_load weight stream made those 15 % SLOWER, so it is used here only
#ifndef GLM_PA_NT
#define GLM_PA_NT 1
#endif
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void*)gsrc, (lds_void*)lds_wave_base, 16, 0, GLM_PA_NT ? 2 : 0);
}

// LDS fragment reads as inline asm: hipcc cannot tell a ds_read (in particular the tr16 builtin)
// from the in-flight LDS-DMA of the OTHER ring stage and would drain vmcnt(0) before it.  The
// wave's own counted vmcnt orders DMA → read; lds_wait_all() below names every destination so
// no consumer is scheduled above the lgkmcnt(0) (cdna_hip_programming.md §5.7 form (ii)).
template <int OFF>
__device__ __forceinline__ void lds_read_b128(u32x4_t& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
__device__ __forceinline__ void lds_write_b128(unsigned addr, const u32x4_t& v) {
    asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read_tr16_b64(u32x2_t& dst, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(const lds_void*)p;
}

// write-through (sc1) stores for data another workgroup reads in the same launch (the kv-split partials): with every
// byte stored this way and drained before the ticket, the producer needs no agent-scope release fence, which costs
// 2-6 us behind freshly written lines (MI355X_MICROARCH.md, visibility table; Guideline 16 R1)
__device__ __forceinline__ void store16_sc1(float* base, int64_t elem_off, f32x4_t v) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, (int)(elem_off * 4), 0, 16);
}

__device__ __forceinline__ f32x4_t load16_sc1(const float* base, int64_t elem_off) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
    return __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(elem_off * 4), 0, 16));
}

// issue the 16 LDS-DMA instructions of one 32-key group into `stage` (wave-private)
__device__ __forceinline__ void issue_group(const bf16_t* __restrict__ kc, const bf16_t* __restrict__ vc,
                                            char* stage, int64_t phys_block, int in_block_off,
                                            int valid_keys /* >=1 */, int block_size, int nkv,
                                            int kvh, int lane) {
    const int r4 = lane >> 4, p = lane & 15;
#pragma unroll
    for (int i = 0; i < PA_GK / 4; ++i) {
        const int row = 4 * i + r4;
        const int key = row < valid_keys ? row : valid_keys - 1;  // clamp: masked keys re-read a valid row
        const int64_t tok = phys_block * block_size + in_block_off + key;
        const int64_t rowoff = (tok * nkv + kvh) * PA_HD;  // elements
        const int chk = p ^ (row & 15);
        const int chv = p ^ ((row & 7) << 1);
        glds16(kc + rowoff + chk * 8, stage + i * 1024);
        glds16(vc + rowoff + chv * 8, stage + PA_GK * PA_ROWB + i * 1024);
    }
}

// FUSED = 1: the kernel starts from the packed QKV projection output instead of a rotated Q and an already updated
// cache (attention_new.py:329-357 + layers/attention.py:10-42 folded in, head RMSNorm off, rotary_dim 64):
//   * Q: lane (head c, g) holds dims [32ks + 8g, +8) of its head; NeoX pairs (d, d+32) for d < 32 are this lane's
//     ks=0 / ks=1 fragments, so the rotation is lane-local (same arithmetic and rounding as rope8 in elementwise.hip).
//   * K/V of the new token: the workgroup whose key range holds position len-1 rotates the K row, writes K and V
//     to the cache slot (not waited for) and keeps the row in LDS; the wave that reaches the last key group patches it
//     into its landed stage (see the prologue comment in the kernel).
struct PaFuse {
    const bf16_t* qkv;     // [B, (nh + 2 nkv) * 128]
    int64_t qkv_stride;    // elements per row
    const int64_t* pos;    // [B]
    const float* cos_sin;  // [n_pos, rotary_dim] fp32: cos | sin halves
    int64_t n_pos;         // rows of cos_sin: positions are clamped into it (a padding row's position is arbitrary)
    const int32_t* slots;  // [B] cache slot of the new token
    int64_t n_slots;
    bf16_t* kc_w;          // writable views of the caches
    bf16_t* vc_w;
    // optional: the QKV projection handed over as fp32 K-slice partial sums (glm_gemm_bf16_partial) instead of bf16 rows;
    // the rows are then formed on load as bf16(sum of the slices in slice order + bias), the linear layer's own rounding
    const float* part;     // [nslices][B][qkv_stride] or nullptr
    int nslices;
    int64_t slice_stride;
    const bf16_t* bias;    // [qkv_stride] or nullptr
};

// 8 consecutive elements of row b of the packed QKV matrix, starting at column col (a multiple of 8)
__device__ __forceinline__ u32x4_t pa_qkv8(const PaFuse& fu, int64_t b, int64_t col) {
    if (fu.part == nullptr) return *reinterpret_cast<const u32x4_t*>(fu.qkv + b * fu.qkv_stride + col);
    const float* pp = fu.part + b * fu.qkv_stride + col;
    f32x4_t lo = *reinterpret_cast<const f32x4_t*>(pp), hi = *reinterpret_cast<const f32x4_t*>(pp + 4);
    for (int sl = 1; sl < fu.nslices; ++sl) {
        lo += *reinterpret_cast<const f32x4_t*>(pp + sl * fu.slice_stride);
        hi += *reinterpret_cast<const f32x4_t*>(pp + sl * fu.slice_stride + 4);
    }
    if (fu.bias != nullptr) {
        const u32x4_t bv = *reinterpret_cast<const u32x4_t*>(fu.bias + col);
        lo[0] += bflo(bv[0]); lo[1] += bfhi(bv[0]); lo[2] += bflo(bv[1]); lo[3] += bfhi(bv[1]);
        hi[0] += bflo(bv[2]); hi[1] += bfhi(bv[2]); hi[2] += bflo(bv[3]); hi[3] += bfhi(bv[3]);
    }
    return u32x4_t{pack2bf(lo[0], lo[1]), pack2bf(lo[2], lo[3]), pack2bf(hi[0], hi[1]), pack2bf(hi[2], hi[3])};
}

// NeoX rotation of 8 packed bf16 pairs: y1 = a*cos - b*sin, y2 = b*cos + a*sin (rope8, elementwise.hip)
__device__ __forceinline__ void pa_rope8(const u32x4_t a, const u32x4_t b, const float* cosp, const float* sinp,
                                         u32x4_t& y1, u32x4_t& y2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float c0 = cosp[2 * j], c1 = cosp[2 * j + 1];
        const float s0 = sinp[2 * j], s1 = sinp[2 * j + 1];
        const float a0 = bflo(a[j]), a1 = bfhi(a[j]);
        const float b0 = bflo(b[j]), b1 = bfhi(b[j]);
        y1[j] = pack2bf(a0 * c0 - b0 * s0, a1 * c1 - b1 * s1);
        y2[j] = pack2bf(b0 * c0 + a0 * s0, b1 * c1 + a1 * s1);
    }
}

template <int FUSED>
__global__ __launch_bounds__(256, 2) void paged_attn_decode_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ kc, const bf16_t* __restrict__ vc,
    const int32_t* __restrict__ context_lens, const int32_t* __restrict__ block_tables,
    int64_t bt_stride, bf16_t* __restrict__ out, float* part_o,
    float* part_ml, unsigned* tickets, int nh, int nkv, int block_size, float scale_log2e,
    int num_splits, const PaFuse fu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int split = blockIdx.x, kvh = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int gq = nh / nkv;  // query heads per kv head (<= 16)
    const int len = context_lens[b];
    const int ngroups = (len + PA_GK - 1) / PA_GK;
    const int gps = (ngroups + num_splits - 1) / num_splits;
    const int G0 = split * gps;
    const int G1 = min(ngroups, G0 + gps);

    char* ring = smem + wid * PA_RING;
    int* bt_lds = reinterpret_cast<int*>(smem + PA_WAVES * PA_RING);

    // block-table slice of this workgroup → LDS (so the loop has no VMEM loads besides the DMA)
    const int gpb = block_size / PA_GK;  // groups per block
    const int blk0 = G0 < G1 ? G0 / gpb : 0;
    const int blk1 = G0 < G1 ? (G1 - 1) / gpb + 1 : 0;
    for (int i = tid; i < blk1 - blk0; i += 256) bt_lds[i] = block_tables[b * bt_stride + blk0 + i];

    // ---- prologue, ordered so that the K/V stream starts as early as possible ----
    // One memory round trip per line: (1) block table, raw Q chunks, position, slot, the new token's raw K/V row; (2)
    // cos/sin of the position.  Only the block table gates the K/V DMAs.  The new token's row never has to come back from
    // the cache: wave 0 rotates it, leaves it in LDS (PA_NEWROW) and fires the cache store without waiting for it; the
    // wave that reaches the last key group overwrites that row (and the clamped copies of it behind it) in its landed
    // stage before it reads fragments.  Before r02 the row was stored, acknowledged and barriered in front of the first
    // DMA: ~4 round trips of prologue (in-graph 88.6 us at C3 against 85.8 us for the plain kernel).
    const int nmine = (G1 - G0 - wid + PA_WAVES - 1) / PA_WAVES;  // groups G0+wid, +4, ...
    auto issue = [&](int it) {
        const int gi = G0 + wid + it * PA_WAVES;
        const int pb = __builtin_amdgcn_readfirstlane(bt_lds[gi / gpb - blk0]);
        const int valid = min(PA_GK, len - gi * PA_GK);
        issue_group(kc, vc, ring + (it & 1) * PA_STAGE, pb, (gi % gpb) * PA_GK, valid, block_size,
                    nkv, kvh, lane);
    };
    const int glast = len > 0 ? (len - 1) / PA_GK : -1;
    const bool owner = FUSED && len > 0 && glast >= G0 && glast < G1;  // this workgroup reads the new token's key
    char* newrow = smem + PA_WAVES * PA_RING + PA_MAX_BT * 4;

    // (1) raw loads: Q^T chunks (B operand: lane (c,g), k-step ks holds Q[head c][32ks + 8g .. +8)), position, new row
    const bool hv = c < gq;
    u32x4_t v[4];
    {
        const int64_t qcol = ((int64_t)kvh * gq + (hv ? c : 0)) * PA_HD;  // FUSED: column of the head in the packed row
        const bf16_t* qrow = q + (b * nh + (int64_t)kvh * gq + (hv ? c : 0)) * PA_HD;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            v[ks] = u32x4_t{0, 0, 0, 0};
            if (hv) v[ks] = FUSED ? pa_qkv8(fu, b, qcol + 32 * ks + 8 * g) : *reinterpret_cast<const u32x4_t*>(qrow + 32 * ks + 8 * g);
        }
    }
    int64_t ps = 0;
    if (FUSED) ps = min(max(fu.pos[b], (int64_t)0), fu.n_pos - 1);
    const bool row_lane = owner && wid == 0 && lane < 32;  // lanes 0-15 rotate and store the K row, 16-31 copy the V row
    const int sub = lane & 15;
    const bool is_k = lane < 16;
    u32x4_t xv = {0, 0, 0, 0}, pv = {0, 0, 0, 0};
    int64_t slot = -1;
    if (FUSED && row_lane) {
        const int64_t rcol = ((int64_t)nh + (is_k ? 0 : nkv) + kvh) * PA_HD;
        xv = pa_qkv8(fu, b, rcol + sub * 8);
        if (is_k && sub < 8) pv = pa_qkv8(fu, b, rcol + (sub < 4 ? sub + 4 : sub - 4) * 8);
        slot = fu.slots[b];
    }
    __syncthreads();  // block table visible
    if (nmine > 0) issue(0);
    if (nmine > 1) issue(1);
    // (2) rotary (rotary_dim 64: chunk g of the first half pairs with chunk g of the second half)
    bf16x8_t qf[4];
    {
        if (FUSED) {
            const float* cosp = fu.cos_sin + ps * 64 + 8 * g;
            u32x4_t y1, y2;
            pa_rope8(v[0], v[1], cosp, cosp + 32, y1, y2);
            v[0] = y1;
            v[1] = y2;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = __builtin_bit_cast(bf16x8_t, v[ks]);
    }
    // (3) the new token's K/V row into the cache
    if (FUSED && row_lane) {
        if (is_k && sub < 8) {
            const float* cosp = fu.cos_sin + ps * 64 + 8 * (sub & 3);
            u32x4_t y1, y2;
            if (sub < 4) {
                pa_rope8(xv, pv, cosp, cosp + 32, y1, y2);
                xv = y1;
            } else {
                pa_rope8(pv, xv, cosp, cosp + 32, y1, y2);
                xv = y2;
            }
        }
        lds_write_b128(lds_addr(newrow) + (is_k ? 0 : PA_ROWB) + sub * 16, xv);
        if (slot >= 0 && slot < fu.n_slots)  // for later steps; nobody in this launch reads it back
            *reinterpret_cast<u32x4_t*>((is_k ? fu.kc_w : fu.vc_w) + (slot * nkv + kvh) * PA_HD + sub * 8) = xv;
    }
    if (owner) {  // workgroup-uniform: the row in LDS is visible to the wave that will reach the last group
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    f32x4_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    // loop-invariant LDS offsets of this lane's fragment reads, relative to the stage base
    unsigned k_rel[4], v_rel[8];
    {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) k_rel[ks] = c * PA_ROWB + (((4 * ks + g) ^ c) << 4);
        const int qq = (lane >> 2) & 3, pp = lane & 3;
        const int row = 4 * g + qq;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt)
            v_rel[dt] = row * PA_ROWB + (((2 * dt + (pp >> 1)) ^ ((row & 7) << 1)) << 4) + ((pp & 1) << 3);
    }
    // the Q fragments are complete here (hipcc waits for their loads; with the early ring that wait also covers the two
    // stages in flight, which iteration 0 needs anyway) — never inside the loop, where it would drain the ring
    asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]));

    for (int it = 0; it < nmine; ++it) {
        char* st = ring + (it & 1) * PA_STAGE;
        if (it + 1 < nmine) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (FUSED && owner && G0 + wid + it * PA_WAVES == glast) {
            // the stage holds the cache's (stale) image of position len-1 and, behind it, the clamped re-reads of that
            // row (issue_group): all of them become the new token's row, in the stage's swizzled chunk order
            const int r0 = (len - 1) % PA_GK, p16 = lane & 15;
            u32x4_t nk, nv;
            lds_read_b128<0>(nk, lds_addr(newrow) + p16 * 16);
            lds_read_b128<PA_ROWB>(nv, lds_addr(newrow) + p16 * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nk), "+v"(nv)::"memory");
            for (int row = r0 + (lane >> 4); row < PA_GK; row += 4) {
                lds_write_b128(lds_addr(st) + row * PA_ROWB + ((p16 ^ (row & 15)) << 4), nk);
                lds_write_b128(lds_addr(st) + PA_GK * PA_ROWB + row * PA_ROWB + ((p16 ^ ((row & 7) << 1)) << 4), nv);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }

        // ---- fragment reads: K (row reads, 16 B) and V (transposed reads, 8 B) ----
        // K: lane (c,g) reads row 16t+c, chunk (4ks+g)^c.  V: lane 4qq+pp of group g reads row
        // 16t+4g+qq, chunk (2dt+(pp>>1)) ^ ((row&7)<<1), half pp&1.  t=1 is +4096 B in both.
        const unsigned sbase = lds_addr(st);
        u32x4_t kf[2][4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned a = sbase + k_rel[ks];
            lds_read_b128<0>(kf[0][ks], a);
            lds_read_b128<4096>(kf[1][ks], a);
        }
        u32x2_t vf[8][2];
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            const unsigned a = sbase + v_rel[dt];
            lds_read_tr16_b64<PA_GK * PA_ROWB>(vf[dt][0], a);
            lds_read_tr16_b64<PA_GK * PA_ROWB + 4096>(vf[dt][1], a);
        }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(kf[0][0]), "+v"(kf[0][1]), "+v"(kf[0][2]), "+v"(kf[0][3]), "+v"(kf[1][0]),
                       "+v"(kf[1][1]), "+v"(kf[1][2]), "+v"(kf[1][3]), "+v"(vf[0][0]), "+v"(vf[0][1]),
                       "+v"(vf[1][0]), "+v"(vf[1][1]), "+v"(vf[2][0]), "+v"(vf[2][1]), "+v"(vf[3][0]),
                       "+v"(vf[3][1]), "+v"(vf[4][0]), "+v"(vf[4][1]), "+v"(vf[5][0]), "+v"(vf[5][1]),
                       "+v"(vf[6][0]), "+v"(vf[6][1]), "+v"(vf[7][0]), "+v"(vf[7][1])
                     :
                     : "memory");
        __builtin_amdgcn_sched_barrier(0);
        // stage is free again: prefetch the group two steps ahead into it
        if (it + 2 < nmine) issue(it + 2);

        // ---- S^T = K · Q^T (keys on rows, heads on lanes) ----
        f32x4_t s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            s[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[t][ks]),
                                                              qf[ks], s[t], 0, 0, 0);
        }
        // ---- online softmax for head c over this lane's 8 keys (+ 3 sibling lane groups) ----
        const int gi = G0 + wid + it * PA_WAVES;
        const int kbase = gi * PA_GK + 4 * g;
        float mloc = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = s[t][r] * scale_log2e;
                if (kbase + 16 * t + r >= len) v = -INFINITY;
                s[t][r] = v;
                mloc = fmaxf(mloc, v);
            }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);  // finite: every group holds >= 1 valid key
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float psum = 0.f;
        bf16x8_t pf;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(s[t][r] - m_new);
                psum += p;
                pf[4 * t + r] = (__bf16)p;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        // ---- O^T = alpha * O^T + V^T · P^T ----
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            acc[dt] *= alpha;
            const u32x4_t vv = {vf[dt][0][0], vf[dt][0][1], vf[dt][1][0], vf[dt][1][1]};
            acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vv), pf,
                                                             acc[dt], 0, 0, 0);
        }
    }

    // ---- merge the 4 waves through LDS (each wave first parks its partial in its own ring) ----
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    float* myo = reinterpret_cast<float*>(ring);             // [16 heads][128 d] fp32 = 8 KiB
    float* myml = reinterpret_cast<float*>(ring + 8192);     // [16][2]
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
        *reinterpret_cast<f32x4_t*>(myo + c * PA_HD + 16 * dt + 4 * g) = acc[dt];
    if (g == 0) {
        myml[2 * c] = m_run;
        myml[2 * c + 1] = l_run;
    }
    __syncthreads();
    {
        const int h = tid >> 4, d0 = (tid & 15) * 8;
        float mw[PA_WAVES], M = -INFINITY;
#pragma unroll
        for (int w = 0; w < PA_WAVES; ++w) {
            mw[w] = reinterpret_cast<const float*>(smem + w * PA_RING + 8192)[2 * h];
            M = fmaxf(M, mw[w]);
        }
        float L = 0.f, o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = 0.f;
#pragma unroll
        for (int w = 0; w < PA_WAVES; ++w) {
            const float f = (mw[w] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mw[w] - M);
            L += f * reinterpret_cast<const float*>(smem + w * PA_RING + 8192)[2 * h + 1];
            const float* src = reinterpret_cast<const float*>(smem + w * PA_RING) + h * PA_HD + d0;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] += f * src[i];
        }
        if (h < gq) {
            const int64_t head = (int64_t)kvh * gq + h;
            if (num_splits == 1) {
                const float inv = L > 0.f ? 1.0f / L : 0.f;
                u32x4_t ov;
#pragma unroll
                for (int i = 0; i < 4; ++i) ov[i] = pack2bf(o[2 * i] * inv, o[2 * i + 1] * inv);
                *reinterpret_cast<u32x4_t*>(out + (b * nh + head) * PA_HD + d0) = ov;
            } else {
                const int64_t prow = (b * nh + head) * num_splits + split;
                if (tickets != nullptr) {  // read by the last-arriving split of this launch: write-through stores
                    store16_sc1(part_o, prow * PA_HD + d0, f32x4_t{o[0], o[1], o[2], o[3]});
                    store16_sc1(part_o, prow * PA_HD + d0 + 4, f32x4_t{o[4], o[5], o[6], o[7]});
                    if ((tid & 15) == 0) {
                        const unsigned long long ml = (unsigned long long)__float_as_uint(M) |
                                                      ((unsigned long long)__float_as_uint(L) << 32);
                        __hip_atomic_store(reinterpret_cast<unsigned long long*>(part_ml + prow * 2), ml,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                } else {
                    float* po = part_o + prow * PA_HD + d0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) po[i] = o[i];
                    if ((tid & 15) == 0) {
                        part_ml[prow * 2] = M;
                        part_ml[prow * 2 + 1] = L;
                    }
                }
            }
        }
    }
    if (num_splits == 1 || tickets == nullptr) return;

    // ---- merge of the kv-splits by the LAST workgroup of this (sequence, kv-head) to finish (no second launch) ----
    // Hand-off (cdna_hip_programming.md Guideline 16 R1, counter form): the partials were stored write-through (sc1);
    // every wave drains its stores, the workgroup meets, one lane draws a ticket; the drawer of num_splits-1 puts the
    // counter back to zero, and behind the next workgroup barrier its waves read the other splits' partials with sc1
    // loads (no acquire: valid-forms table, row 1).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem + PA_WAVES * PA_RING);  // the block-table slice is dead
    if (tid == 0) {
        unsigned* tk = tickets + b * nkv + kvh;
        const unsigned got = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = got == (unsigned)num_splits - 1u;
        if (last) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag[0] = last;
    }
    __syncthreads();
    if (!flag[0]) return;
    {
        // same arithmetic and summation order as paged_attn_reduce_kernel: thread (head h, 8 dims at d0)
        const int h = tid >> 4, d0 = (tid & 15) * 8;
        if (h < gq) {
            const int64_t bh = b * nh + (int64_t)kvh * gq + h;
            // every partial was stored write-through and drained before its split's ticket: sc1 loads, no acquire
            float M = -INFINITY;
            for (int sp = 0; sp < num_splits; ++sp) {
                const unsigned long long ml = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(part_ml + (bh * num_splits + sp) * 2),
                                                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                M = fmaxf(M, __uint_as_float((unsigned)ml));
            }
            float L = 0.f, o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = 0.f;
            for (int sp = 0; sp < num_splits; ++sp) {
                const unsigned long long ml = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(part_ml + (bh * num_splits + sp) * 2),
                                                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float m = __uint_as_float((unsigned)ml);
                const float f = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - M);
                L += f * __uint_as_float((unsigned)(ml >> 32));
                const int64_t po = (bh * num_splits + sp) * PA_HD + d0;
                const f32x4_t a = load16_sc1(part_o, po), c4 = load16_sc1(part_o, po + 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    o[i] += f * a[i];
                    o[4 + i] += f * c4[i];
                }
            }
            u32x4_t ov;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                ov[i] = pack2bf(L > 0.f ? o[2 * i] / L : 0.f, L > 0.f ? o[2 * i + 1] / L : 0.f);
            *reinterpret_cast<u32x4_t*>(out + bh * PA_HD + d0) = ov;
        }
    }
}

// merge kv-splits: one 128-thread workgroup per (sequence, head)
__global__ __launch_bounds__(128) void paged_attn_reduce_kernel(const float* __restrict__ part_o,
                                                                const float* __restrict__ part_ml,
                                                                bf16_t* __restrict__ out,
                                                                int num_splits) {
    const int64_t bh = blockIdx.x;
    const int d = threadIdx.x;
    float M = -INFINITY;
    for (int s = 0; s < num_splits; ++s) M = fmaxf(M, part_ml[(bh * num_splits + s) * 2]);
    float L = 0.f, o = 0.f;
    for (int s = 0; s < num_splits; ++s) {
        const float m = part_ml[(bh * num_splits + s) * 2];
        const float f = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - M);
        L += f * part_ml[(bh * num_splits + s) * 2 + 1];
        o += f * part_o[(bh * num_splits + s) * PA_HD + d];
    }
    out[bh * PA_HD + d] = f2bf(L > 0.f ? o / L : 0.f);
}

// the first PA_TICKET_BYTES of the workspace are arrival counters, one per (sequence, kv-head): zero before the first
// launch on a workspace, put back to zero by the kernel
constexpr size_t PA_TICKET_BYTES = 65536;

int choose_splits(int64_t B, int nkv, int max_ctx) {
    // aim for >= 256 workgroups (one per CU) without making a split shorter than one wave-step per wave (128 keys):
    // at single-sequence decode the kernel is latency-bound, and ctx = 1024 over 2 splits kept 240 CUs idle (20.5 us)
    const int64_t base = B * nkv;
    int s = (int)((256 + base - 1) / base);
    const int max_by_len = max_ctx > 0 ? (max_ctx + PA_GK * PA_WAVES - 1) / (PA_GK * PA_WAVES) : 1;
    if (s > max_by_len) s = max_by_len;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return s;
}

}  // namespace

// the split count pa_launch uses for these arguments (kept in one place so that callers can size the workspace exactly)
static int pa_effective_splits(int64_t B, int nkv, int block_size, int64_t bt_stride, int num_splits, int max_context_len) {
    if (max_context_len <= 0) max_context_len = (int)(bt_stride * block_size);
    int splits = num_splits > 0 ? num_splits : choose_splits(B, nkv, max_context_len);
    return splits > 64 ? 64 : splits;
}

extern "C" int glm_paged_attn_decode_splits(int64_t B, int nkv, int block_size, int64_t bt_stride, int num_splits,
                                            int max_context_len) {
    if (B <= 0 || nkv <= 0) return 1;
    return pa_effective_splits(B, nkv, block_size, bt_stride, num_splits, max_context_len);
}

extern "C" size_t glm_paged_attn_decode_workspace_bytes(int64_t B, int nh, int hd, int num_splits) {
    const int s = num_splits <= 0 ? 64 : num_splits;
    if (s == 1) return 0;
    return PA_TICKET_BYTES + (size_t)B * nh * s * (hd + 2) * sizeof(float);
}

static int pa_launch(const void* q, const void* k_cache, const void* v_cache, const int32_t* context_lens,
                     const int32_t* block_tables, int64_t bt_stride, void* out, int64_t B, int nh, int nkv, int hd,
                     int block_size, float scale, int num_splits, int max_context_len, void* workspace,
                     size_t workspace_bytes, glm_stream_t stream, const PaFuse* fuse, const char* name) {
    GLM_CHECK_ARG(B >= 0 && nh > 0 && nkv > 0 && nh % nkv == 0 && block_size > 0 && bt_stride > 0,
                  GLM_ERR_INVALID, "%s: bad shape", name);
    GLM_CHECK_ARG(hd == PA_HD && nh / nkv <= 16 && block_size % PA_GK == 0 && bt_stride <= PA_MAX_BT &&
                      nkv <= 65535 && B <= 65535,
                  GLM_ERR_UNSUPPORTED,
                  "%s: needs head_dim=128, nh/nkv<=16, block_size%%32==0, "
                  "<=%d blocks per sequence (hd=%d group=%d block_size=%d bt_stride=%ld)",
                  name, PA_MAX_BT, hd, nh / nkv, block_size, (long)bt_stride);
    if (B == 0) return GLM_OK;
    const int splits = pa_effective_splits(B, nkv, block_size, bt_stride, num_splits, max_context_len);
    float* part_o = nullptr;
    float* part_ml = nullptr;
    unsigned* tickets = nullptr;
    if (splits > 1) {
        const size_t need = PA_TICKET_BYTES + (size_t)B * nh * splits * (PA_HD + 2) * sizeof(float);
        GLM_CHECK_ARG(workspace != nullptr && workspace_bytes >= need && glm_aligned16(workspace), GLM_ERR_WORKSPACE,
                      "%s: workspace %zu < %zu bytes (or not 16-byte aligned)", name, workspace_bytes, need);
        part_o = (float*)((char*)workspace + PA_TICKET_BYTES);
        part_ml = part_o + (size_t)B * nh * splits * PA_HD;
        // in-kernel merge by the last-arriving split when the counters fit; else the separate reduce launch
        if ((size_t)B * nkv * sizeof(unsigned) <= PA_TICKET_BYTES) tickets = (unsigned*)workspace;
    }
    const int fz = fuse != nullptr;
    if (fz) GLM_SET_MAX_LDS(paged_attn_decode_kernel<1>, PA_LDS, name);
    else GLM_SET_MAX_LDS(paged_attn_decode_kernel<0>, PA_LDS, name);
    hipStream_t st = glm_s(stream);
    const float scale_log2e = scale * 1.4426950408889634f;
    const dim3 grid((unsigned)splits, (unsigned)nkv, (unsigned)B);
    PaFuse fu;
    memset(&fu, 0, sizeof(fu));
    if (fz) {
        fu = *fuse;
        hipLaunchKernelGGL((paged_attn_decode_kernel<1>), grid, dim3(256), PA_LDS, st, (const bf16_t*)nullptr,
                           (const bf16_t*)k_cache, (const bf16_t*)v_cache, context_lens, block_tables, bt_stride,
                           (bf16_t*)out, part_o, part_ml, tickets, nh, nkv, block_size, scale_log2e, splits, fu);
    } else {
        hipLaunchKernelGGL((paged_attn_decode_kernel<0>), grid, dim3(256), PA_LDS, st, (const bf16_t*)q,
                           (const bf16_t*)k_cache, (const bf16_t*)v_cache, context_lens, block_tables, bt_stride,
                           (bf16_t*)out, part_o, part_ml, tickets, nh, nkv, block_size, scale_log2e, splits, fu);
    }
    GLM_CHECK_LAUNCH(name);
    if (splits > 1 && tickets == nullptr) {
        hipLaunchKernelGGL(paged_attn_reduce_kernel, dim3((unsigned)(B * nh)), dim3(128), 0, st, part_o,
                           part_ml, (bf16_t*)out, splits);
        GLM_CHECK_LAUNCH(name);
    }
    return GLM_OK;
}

extern "C" int glm_paged_attn_decode(const void* q, const void* k_cache, const void* v_cache,
                                     const int32_t* context_lens, const int32_t* block_tables,
                                     int64_t bt_stride, void* out, int64_t B, int nh, int nkv, int hd,
                                     int block_size, int64_t n_blocks, float scale, int num_splits,
                                     int max_context_len, void* workspace, size_t workspace_bytes,
                                     glm_stream_t stream) {
    (void)n_blocks;
    GLM_CHECK_ARG(q && k_cache && v_cache && context_lens && block_tables && out, GLM_ERR_INVALID,
                  "glm_paged_attn_decode: null pointer");
    GLM_CHECK_ARG(glm_aligned16(q) && glm_aligned16(k_cache) && glm_aligned16(v_cache) &&
                      glm_aligned16(out),
                  GLM_ERR_INVALID, "glm_paged_attn_decode: pointers must be 16-byte aligned");
    return pa_launch(q, k_cache, v_cache, context_lens, block_tables, bt_stride, out, B, nh, nkv, hd, block_size, scale,
                     num_splits, max_context_len, workspace, workspace_bytes, stream, nullptr, "glm_paged_attn_decode");
}

extern "C" int glm_paged_attn_decode_qkv(const void* qkv, int64_t qkv_stride, const int64_t* positions,
                                         const float* cos_sin_cache, int64_t n_pos, int rotary_dim,
                                         const int32_t* slot_mapping,
                                         void* k_cache, void* v_cache, const int32_t* context_lens,
                                         const int32_t* block_tables, int64_t bt_stride, void* out, int64_t B,
                                         int nh, int nkv, int hd, int block_size, int64_t n_blocks, float scale,
                                         int num_splits, int max_context_len, void* workspace,
                                         size_t workspace_bytes, glm_stream_t stream) {
    GLM_CHECK_ARG(qkv && positions && cos_sin_cache && slot_mapping && k_cache && v_cache && context_lens &&
                      block_tables && out, GLM_ERR_INVALID, "glm_paged_attn_decode_qkv: null pointer");
    GLM_CHECK_ARG(n_pos > 0, GLM_ERR_INVALID, "glm_paged_attn_decode_qkv: empty cos_sin_cache");
    GLM_CHECK_ARG(rotary_dim == 64 && qkv_stride >= (int64_t)(nh + 2 * nkv) * hd && qkv_stride % 8 == 0 && n_blocks > 0,
                  GLM_ERR_UNSUPPORTED, "glm_paged_attn_decode_qkv: needs rotary_dim=64 (got %d) and a packed qkv row",
                  rotary_dim);
    GLM_CHECK_ARG(glm_aligned16(qkv) && glm_aligned16(k_cache) && glm_aligned16(v_cache) && glm_aligned16(out),
                  GLM_ERR_INVALID, "glm_paged_attn_decode_qkv: pointers must be 16-byte aligned");
    PaFuse fu;
    fu.qkv = (const bf16_t*)qkv; fu.qkv_stride = qkv_stride; fu.pos = positions; fu.cos_sin = cos_sin_cache; fu.n_pos = n_pos;
    fu.slots = slot_mapping; fu.n_slots = n_blocks * block_size; fu.kc_w = (bf16_t*)k_cache; fu.vc_w = (bf16_t*)v_cache;
    fu.part = nullptr; fu.nslices = 0; fu.slice_stride = 0; fu.bias = nullptr;
    return pa_launch(nullptr, k_cache, v_cache, context_lens, block_tables, bt_stride, out, B, nh, nkv, hd, block_size,
                     scale, num_splits, max_context_len, workspace, workspace_bytes, stream, &fu,
                     "glm_paged_attn_decode_qkv");
}

extern "C" int glm_paged_attn_decode_qkv_partial(const float* qkv_part, int nslices, int64_t slice_stride,
                                                 const void* qkv_bias, const int64_t* positions,
                                                 const float* cos_sin_cache, int64_t n_pos, int rotary_dim,
                                                 const int32_t* slot_mapping, void* k_cache, void* v_cache,
                                                 const int32_t* context_lens, const int32_t* block_tables,
                                                 int64_t bt_stride, void* out, int64_t B, int nh, int nkv, int hd,
                                                 int block_size, int64_t n_blocks, float scale, int num_splits,
                                                 int max_context_len, void* workspace, size_t workspace_bytes,
                                                 glm_stream_t stream) {
    GLM_CHECK_ARG(qkv_part && positions && cos_sin_cache && slot_mapping && k_cache && v_cache && context_lens &&
                      block_tables && out, GLM_ERR_INVALID, "glm_paged_attn_decode_qkv_partial: null pointer");
    const int64_t width = (int64_t)(nh + 2 * nkv) * hd;
    GLM_CHECK_ARG(n_pos > 0 && nslices >= 1 && slice_stride >= B * width, GLM_ERR_INVALID,
                  "glm_paged_attn_decode_qkv_partial: bad shape (nslices=%d)", nslices);
    GLM_CHECK_ARG(rotary_dim == 64 && slice_stride % 4 == 0 && n_blocks > 0, GLM_ERR_UNSUPPORTED,
                  "glm_paged_attn_decode_qkv_partial: needs rotary_dim=64 (got %d)", rotary_dim);
    GLM_CHECK_ARG(glm_aligned16(qkv_part) && (qkv_bias == nullptr || glm_aligned16(qkv_bias)) && glm_aligned16(k_cache) &&
                      glm_aligned16(v_cache) && glm_aligned16(out),
                  GLM_ERR_INVALID, "glm_paged_attn_decode_qkv_partial: pointers must be 16-byte aligned");
    PaFuse fu;
    fu.qkv = nullptr; fu.qkv_stride = width; fu.pos = positions; fu.cos_sin = cos_sin_cache; fu.n_pos = n_pos;
    fu.slots = slot_mapping; fu.n_slots = n_blocks * block_size; fu.kc_w = (bf16_t*)k_cache; fu.vc_w = (bf16_t*)v_cache;
    fu.part = qkv_part; fu.nslices = nslices; fu.slice_stride = slice_stride; fu.bias = (const bf16_t*)qkv_bias;
    return pa_launch(nullptr, k_cache, v_cache, context_lens, block_tables, bt_stride, out, B, nh, nkv, hd, block_size,
                     scale, num_splits, max_context_len, workspace, workspace_bytes, stream, &fu,
                     "glm_paged_attn_decode_qkv_partial");
}
