// prefill_attn64.hip — varlen causal prefill attention, 64 query rows per wave at ONE wave per SIMD (r04; a9, reference:
// nanovllm/layers/attention.py:85-100 -> flash_attn_varlen_func(causal=True[, block_table])).  gfx950, head_dim 128.
//
// Why: the 32-rows-per-wave kernel (prefill_attn.hip) is bound by LDS bytes per FLOP — every wave reads the whole 32 KiB K/V tile
// for 32 query rows; its ablation puts the fragment reads at 28 % and the staging at 14 % of the time with the P.V MFMAs hidden
// completely.  64 rows per wave halve both per FLOP, but need the whole 512-register file of a SIMD for one wave, and hipcc
// will not hold MFMA operands in the accumulator half on its own (r03: three attempts, ~1400 v_accvgpr copies each).  Here the
// accumulator half is ASM-OWNED (cdna_hip_programming.md §5.7 item 4): the O accumulators, the Q^T fragments and the K / V
// fragments live in hand-numbered AGPRs that only inline-asm statements touch — MFMA A/B operands may be AGPRs on gfx950, and
// DS loads can target them — while hipcc allocates the architectural VGPRs for the scores, the softmax and the addresses.
//
//   a[  0:127]  O^T accumulators   O(qt, dt)  = a[32 qt + 4 dt, +4)      qt = 16-row query tile 0..3, dt = 16-wide hd block 0..7
//   a[128:191]  Q^T fragments      Q(qt, ks)  = a[128 + 16 qt + 4 ks, +4)  ks = 32-wide k-step over head_dim
//   a[192:223]  K fragments        K(tt, ks)  = a[192 + 16 tt + 4 ks, +4)  tt = 16-key half of a 32-key slab
//   a[224:255]  V^T fragments      V(dt)      = a[224 + 4 dt, +4)          two ds_read_b64_tr_b16 halves
//
// A workgroup = 4 waves = 256 query rows of one (sequence, head); K/V tiles of 64 keys, double-buffered by LDS-DMA with the
// swizzles of the other kernels.  A tile = 8 units (32-key slab x query tile).  Step u of a tile issues, in this order,
//   the S^T = K.Q^T MFMAs of unit u+1 interleaved with the softmax VALU of unit u (one fma + one exp2 per MFMA gap),
//   the bf16 conversion of P(u) interleaved with the first P.V MFMAs of unit u-1 (one unit late), the rest of them bare,
// so that the matrix pipe always has the next MFMA queued while the vector ALU works: the S results of unit u+1 are read one
// step later behind nine P.V MFMAs (no MFMA -> VALU hazard), P(u) is consumed one step later (no VALU -> MFMA hazard), and
// every LDS read is issued one step before its first use.  Scheduling fences pin the order (inline asm is opaque to hipcc).
// Arithmetic, rounding points, lazy maximum, masking rules, work order are the 32-row kernel's: same results up to the
// order of the fp32 sums inside the MFMAs (identical) — the parity tests are shared.
//
// STATUS (r04): parity green (test_varlen_attn_prefill_rows64), OPT-IN (option fa_rows64) and SLOWER than the 32-row kernel —
// 0.32 against 0.40 of the bf16 MFMA peak at 4 x 4096 tokens, 0.41 against 0.45 at 1 x 16384.  One wave per SIMD can issue one
// ordinary VALU instruction in an MFMA's shadow for free, but a softmax needs an fma AND an exp2 per S MFMA (8.5 + 4.5 + 8.6
// cycles against the pipe's 16.5; tools/mfma_valu_probe.hip), and one workgroup per CU has nothing to overlap its start and
// end with.  Measurements, ablations and what was tried: profiles/r04/prefill_attn64_ab.txt, DESIGN.md 4.3.
#include "common.h"
#include "gemm_internal.h"
#include <type_traits>

namespace {

constexpr int F6_HD = 128, F6_ROWB = 256, F6_BK = 64, F6_NW = 4, F6_NT = 256, F6_QT = 4;
constexpr int F6_BQ = 16 * F6_QT * F6_NW;      // 256 query rows per workgroup
constexpr int F6_PPW = 16 / F6_NW;             // 1 KiB staging pieces per wave, for K and for V
constexpr int F6_TILE = F6_BK * F6_ROWB;       // 16 KiB per K or V tile
constexpr int F6_STAGE = 2 * F6_TILE;
constexpr int F6_MAX_BT = 4096;
constexpr int A_O = 0, A_Q = 128, A_K = 192, A_V = 224;

typedef __attribute__((address_space(3))) void lds_void6_t;
typedef __attribute__((address_space(1))) const void glb_void6_t;

__device__ __forceinline__ void glds16_6(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void6_t*)gsrc, (lds_void6_t*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ unsigned lds_addr6(const void* p) { return (unsigned)(uintptr_t)(const lds_void6_t*)p; }

template <int N, class F>
__device__ __attribute__((always_inline)) inline void static_for6(F&& f) {
    if constexpr (N > 0) {
        static_for6<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// ---- asm-owned accumulator-file operands (every statement clobbers nothing the compiler owns; "a255" in ONE of them makes
// the kernel descriptor allocate all 256) ----
template <int R>
__device__ __forceinline__ void awrite(unsigned v) { asm volatile("v_accvgpr_write_b32 a[%c0], %1" ::"i"(R), "v"(v)); }
template <int R>
__device__ __forceinline__ float aread() {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(v) : "i"(R));
    return v;
}
template <int LO, int OFF>
__device__ __forceinline__ void lds_k(unsigned addr) {
    asm volatile("ds_read_b128 a[%c0:%c1], %2 offset:%3" ::"i"(LO), "i"(LO + 3), "v"(addr), "i"(OFF) : "memory");
}
template <int LO, int OFF>
__device__ __forceinline__ void lds_v(unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 a[%c0:%c1], %2 offset:%3" ::"i"(LO), "i"(LO + 1), "v"(addr), "i"(OFF) : "memory");
}
// SAFE variants (generic tiles, where runtime conditions select between statements that write the same variable): the compiler
// does not know these statements are MFMAs, so the register copies it places around one where such paths meet get no wait
// states — a v_mov into the accumulator right in front of the MFMA that reads it, or a v_mov out of it right behind (both
// seen in the generated code, and both wrong on the chip).  These variants carry the wait states themselves: bit 0 = two in
// front (VALU write -> MFMA SrcC read), bit 1 = 24 behind (MFMA write -> VALU read).  The interior loop is straight-line
// code with one writer per variable and nine MFMAs between a result and its first reader, and uses the bare forms.
#define F6_HEAD "s_nop 1\n\t"
#define F6_TAIL "\n\ts_nop 15\n\ts_nop 7"
template <int KLO, int QLO>
__device__ __forceinline__ void mfma_s_first(f32x4_t& s) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, a[%c1:%c2], a[%c3:%c4], 0" : "=v"(s) : "i"(KLO), "i"(KLO + 3), "i"(QLO), "i"(QLO + 3));
}
template <int KLO, int QLO, int SAFE = 0>
__device__ __forceinline__ void mfma_s(f32x4_t& s) {
    if constexpr (SAFE == 3)
        asm volatile(F6_HEAD "v_mfma_f32_16x16x32_bf16 %0, a[%c1:%c2], a[%c3:%c4], %0" F6_TAIL : "+v"(s) : "i"(KLO), "i"(KLO + 3), "i"(QLO), "i"(QLO + 3));
    else if constexpr (SAFE == 1)
        asm volatile(F6_HEAD "v_mfma_f32_16x16x32_bf16 %0, a[%c1:%c2], a[%c3:%c4], %0" : "+v"(s) : "i"(KLO), "i"(KLO + 3), "i"(QLO), "i"(QLO + 3));
    else
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, a[%c1:%c2], a[%c3:%c4], %0" : "+v"(s) : "i"(KLO), "i"(KLO + 3), "i"(QLO), "i"(QLO + 3));
}

// ---- fused gap statements: one MFMA and the VALU work that rides in its shadow, in ONE asm statement (the compiler puts a
// wait state after a statement whose VGPR result the next instruction reads — it must assume a transcendental — so every
// pairing that matters is spelled out here; measured cost model of one wave per SIMD in tools/mfma_valu_probe.hip) ----
// S MFMA + fma of score xf (xf = xf * scale - m) [+ exp2 of the previous score xe]
template <int KLO, int QLO, bool FIRST>
__device__ __forceinline__ void gap_s_fma(f32x4_t& s, float& xf, float scale, float m) {
    if constexpr (FIRST)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, a[%c4:%c5], a[%c6:%c7], 0\n\tv_fma_f32 %1, %1, %2, -%3"
                     : "=&v"(s), "+v"(xf) : "s"(scale), "v"(m), "i"(KLO), "i"(KLO + 3), "i"(QLO), "i"(QLO + 3));
    else
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, a[%c4:%c5], a[%c6:%c7], %0\n\tv_fma_f32 %1, %1, %2, -%3"
                     : "+v"(s), "+v"(xf) : "s"(scale), "v"(m), "i"(KLO), "i"(KLO + 3), "i"(QLO), "i"(QLO + 3));
}
template <int KLO, int QLO, bool FIRST, int SAFE = 0>
__device__ __forceinline__ void gap_s_fma_exp(f32x4_t& s, float& xf, float& xe, float scale, float m) {
    if constexpr (FIRST)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, a[%c5:%c6], a[%c7:%c8], 0\n\tv_fma_f32 %1, %1, %3, -%4\n\tv_exp_f32 %2, %2"
                     : "=&v"(s), "+v"(xf), "+v"(xe) : "s"(scale), "v"(m), "i"(KLO), "i"(KLO + 3), "i"(QLO), "i"(QLO + 3));
    else if constexpr (SAFE == 3)
        asm volatile(F6_HEAD "v_mfma_f32_16x16x32_bf16 %0, a[%c5:%c6], a[%c7:%c8], %0\n\tv_fma_f32 %1, %1, %3, -%4\n\tv_exp_f32 %2, %2" F6_TAIL
                     : "+v"(s), "+v"(xf), "+v"(xe) : "s"(scale), "v"(m), "i"(KLO), "i"(KLO + 3), "i"(QLO), "i"(QLO + 3));
    else if constexpr (SAFE == 1)
        asm volatile(F6_HEAD "v_mfma_f32_16x16x32_bf16 %0, a[%c5:%c6], a[%c7:%c8], %0\n\tv_fma_f32 %1, %1, %3, -%4\n\tv_exp_f32 %2, %2"
                     : "+v"(s), "+v"(xf), "+v"(xe) : "s"(scale), "v"(m), "i"(KLO), "i"(KLO + 3), "i"(QLO), "i"(QLO + 3));
    else
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, a[%c5:%c6], a[%c7:%c8], %0\n\tv_fma_f32 %1, %1, %3, -%4\n\tv_exp_f32 %2, %2"
                     : "+v"(s), "+v"(xf), "+v"(xe) : "s"(scale), "v"(m), "i"(KLO), "i"(KLO + 3), "i"(QLO), "i"(QLO + 3));
}
// row-sum MFMA + exp2 of the last score
template <bool SAFE = false>
__device__ __forceinline__ void gap_l_exp(f32x4_t& l, const bf16x8_t& ones, const u32x4_t& pf, float& xe) {
    if constexpr (SAFE) asm volatile(F6_HEAD "v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_exp_f32 %1, %1" F6_TAIL : "+v"(l), "+v"(xe) : "v"(ones), "v"(pf));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_exp_f32 %1, %1" : "+v"(l), "+v"(xe) : "v"(ones), "v"(pf));
}
template <bool SAFE = false>
__device__ __forceinline__ void gap_l(f32x4_t& l, const bf16x8_t& ones, const u32x4_t& pf) {
    if constexpr (SAFE) asm volatile(F6_HEAD "v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" F6_TAIL : "+v"(l) : "v"(ones), "v"(pf));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(l) : "v"(ones), "v"(pf));
}
// P.V MFMA alone / + one packed convert / + a piece of the next unit's maximum chain
template <int OLO, int VLO>
__device__ __forceinline__ void gap_pv(const u32x4_t& pf) {
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%c0:%c1], a[%c2:%c3], %4, a[%c0:%c1]" ::"i"(OLO), "i"(OLO + 3), "i"(VLO), "i"(VLO + 3), "v"(pf));
}
template <int OLO, int VLO>
__device__ __forceinline__ void gap_pv_cvt(const u32x4_t& pf, unsigned& out, float a, float b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%c4:%c5], a[%c6:%c7], %1, a[%c4:%c5]\n\tv_cvt_pk_bf16_f32 %0, %2, %3"
                 : "=&v"(out) : "v"(pf), "v"(a), "v"(b), "i"(OLO), "i"(OLO + 3), "i"(VLO), "i"(VLO + 3));
}
template <int OLO, int VLO>
__device__ __forceinline__ void gap_pv_max_a(const u32x4_t& pf, float& ms, float a0, float a1, float a2, float a3, float a4) {
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%c7:%c8], a[%c9:%c10], %1, a[%c7:%c8]\n\t"
                 "v_max3_f32 %0, %2, %3, %4\n\t"
                 "v_max3_f32 %0, %0, %5, %6"
                 : "=&v"(ms) : "v"(pf), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "i"(OLO), "i"(OLO + 3), "i"(VLO), "i"(VLO + 3));
}
template <int OLO, int VLO>
__device__ __forceinline__ void gap_pv_max_b(const u32x4_t& pf, float& ms, float a5, float a6, float a7) {
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%c5:%c6], a[%c7:%c8], %1, a[%c5:%c6]\n\t"
                 "v_max3_f32 %0, %0, %2, %3\n\t"
                 "v_max_f32 %0, %0, %4"
                 : "+v"(ms) : "v"(pf), "v"(a5), "v"(a6), "v"(a7), "i"(OLO), "i"(OLO + 3), "i"(VLO), "i"(VLO + 3));
}
template <int OLO, int VLO>
__device__ __forceinline__ void gap_pv_max_c(const u32x4_t& pf, float& ms, float scale) {
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%c3:%c4], a[%c5:%c6], %1, a[%c3:%c4]\n\t"
                 "v_mul_f32 %0, %2, %0"
                 : "+v"(ms) : "v"(pf), "s"(scale), "i"(OLO), "i"(OLO + 3), "i"(VLO), "i"(VLO + 3));
}

template <bool PAGED>
__global__ __launch_bounds__(F6_NT, 1) void prefill_attn64_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const int32_t* __restrict__ cu_q, const int32_t* __restrict__ cu_k, const int32_t* __restrict__ block_tables,
    int64_t bt_stride, bf16_t* __restrict__ out, int nh, int nkv, int block_size, float scale_log2e, int nqblk, int nseq,
    int xcd_groups) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* bt_lds = reinterpret_cast<int*>(smem + 2 * F6_STAGE);
    // work order: as prefill_attn_kernel (all workgroups in flight share one query block, longest first; an XCD keeps the same
    // (sequence, kv head) groups)
    int qblk, head;
    int64_t b;
    {
        const int L = blockIdx.x;
        const int hpg = nh / nkv;
        if (xcd_groups > 0) {
            const int xcd = L & 7, j = L >> 3;
            const int per_q = xcd_groups * hpg;
            const int sidx = j / per_q, r = j - sidx * per_q;
            const int group = (r / hpg) * 8 + xcd;
            qblk = nqblk - 1 - sidx;
            b = group / nkv;
            head = (group % nkv) * hpg + r % hpg;
        } else {
            const int per_q = nh * nseq;
            const int sidx = L / per_q, r = L - sidx * per_q;
            qblk = nqblk - 1 - sidx;
            b = r / nh;
            head = r % nh;
        }
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int q0 = cu_q[b], lq = cu_q[b + 1] - q0;
    const int k0 = cu_k[b], lk = cu_k[b + 1] - k0;
    const int qbase = qblk * F6_BQ;
    if (qbase >= lq) return;  // uniform
    const int kvh = head / (nh / nkv);
    const int off = lk - lq;  // causal: query i sees keys j <= i + off
    const int kend = min(lk, min(lq, qbase + F6_BQ) + off);
    const int ntiles = (kend + F6_BK - 1) / F6_BK;
    if (PAGED) {
        const int nblk = (kend + block_size - 1) / block_size;
        for (int i = tid; i < nblk; i += F6_NT) bt_lds[i] = block_tables[b * bt_stride + i];
    }
    const int r0w = qbase + 16 * F6_QT * wid;  // first query row of this wave

    // ---- Q^T fragments into a[128:191]: lane (c, g), tile qt: row r0w + 16 qt + c, k-step ks: hd [32 ks + 8 g, +8) ----
    int qi[F6_QT], jmax[F6_QT];
    static_for6<F6_QT>([&](auto qc) __attribute__((always_inline)) {
        constexpr int qt = decltype(qc)::value;
        qi[qt] = r0w + 16 * qt + c;
        jmax[qt] = qi[qt] + off;
        const bool ok = qi[qt] < lq;
        const bf16_t* qrow = q + ((int64_t)(q0 + (ok ? qi[qt] : 0)) * nh + head) * F6_HD;
        static_for6<4>([&](auto kc) __attribute__((always_inline)) {
            constexpr int ks = decltype(kc)::value;
            u32x4_t t = {0, 0, 0, 0};
            if (ok) t = *reinterpret_cast<const u32x4_t*>(qrow + 32 * ks + 8 * g);
            awrite<A_Q + 16 * qt + 4 * ks + 0>(t[0]);
            awrite<A_Q + 16 * qt + 4 * ks + 1>(t[1]);
            awrite<A_Q + 16 * qt + 4 * ks + 2>(t[2]);
            awrite<A_Q + 16 * qt + 4 * ks + 3>(t[3]);
        });
    });
    // O accumulators = 0 (the clobber makes the kernel descriptor allocate the whole accumulator file; ONE statement carries it —
    // the compiler puts a wait state after every statement with a physical-register clobber)
    asm volatile("v_accvgpr_write_b32 a255, %0" ::"v"(0u) : "a255");
    static_for6<128>([&](auto rc) __attribute__((always_inline)) { awrite<A_O + decltype(rc)::value>(0u); });
    __syncthreads();

    // ---- staging (as prefill_attn_kernel): wave w issues F6_PPW pieces for K and for V; piece i covers tile rows
    // (F6_PPW w + i) 4 .. +3; lane -> row + (lane >> 4), 16-byte position lane & 15, source-side swizzles ----
    const int64_t row_bytes = (int64_t)nkv * F6_ROWB;
    unsigned soff_k[F6_PPW], soff_v[F6_PPW];
    auto stage_offsets = [&](int t) __attribute__((always_inline)) {
        const int r4 = lane >> 4, pch = lane & 15;
#pragma unroll
        for (int i = 0; i < F6_PPW; ++i) {
            const int row = (wid * F6_PPW + i) * 4 + r4;
            const int rc = min(t * F6_BK + row, lk - 1) - t * F6_BK;  // masked keys re-read a valid row
            soff_k[i] = (unsigned)(rc * (int)row_bytes) + ((pch ^ (row & 15)) << 4);
            soff_v[i] = (unsigned)(rc * (int)row_bytes) + ((pch ^ ((row & 7) << 1)) << 4);
        }
    };
    if (!PAGED) stage_offsets(0);
    const int64_t tstride = (int64_t)F6_BK * row_bytes;
    const char* kbase_p = reinterpret_cast<const char*>(k) + ((int64_t)k0 * nkv + kvh) * F6_ROWB;
    const char* vbase_p = reinterpret_cast<const char*>(v) + ((int64_t)k0 * nkv + kvh) * F6_ROWB;
    auto stage_piece = [&](int buf, int t, int i) __attribute__((always_inline)) {
        char* kd = smem + buf * F6_STAGE + (wid * F6_PPW + i) * 1024;
        char* vd = kd + F6_TILE;
        if (!PAGED) {
            if (i == 0 && (t + 1) * F6_BK > lk) stage_offsets(t);  // uniform; last tile only
            const int64_t tb = (int64_t)t * tstride;
            glds16_6(kbase_p + tb + soff_k[i], kd);
            glds16_6(vbase_p + tb + soff_v[i], vd);
        } else {
            const int r4 = lane >> 4, pch = lane & 15;
            const int row = (wid * F6_PPW + i) * 4 + r4;
            int key = t * F6_BK + row;
            key = key < kend ? key : kend - 1;  // masked keys re-read a row of a block the table slice holds
            const int pb = bt_lds[key / block_size];
            const int64_t tok = (int64_t)pb * block_size + key % block_size;
            const int64_t ro = (tok * nkv + kvh) * F6_HD;
            glds16_6(k + ro + ((pch ^ (row & 15)) << 3), kd);
            glds16_6(v + ro + ((pch ^ ((row & 7) << 1)) << 3), vd);
        }
    };
    auto stage = [&](int buf, int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < F6_PPW; ++i) stage_piece(buf, t, i);
    };

    // LDS fragment addresses of the CURRENT buffer (flip once per tile)
    unsigned k_abs[4], v_abs[8];
    {
        const unsigned base = lds_addr6(smem);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) k_abs[ks] = base + c * F6_ROWB + (((4 * ks + g) ^ c) << 4);
        const int qq = (lane >> 2) & 3, pp = lane & 3;
        const int row = 4 * g + qq;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt)
            v_abs[dt] = base + F6_TILE + row * F6_ROWB + (((2 * dt + (pp >> 1)) ^ ((row & 7) << 1)) << 4) + ((pp & 1) << 3);
    }
    auto read_k = [&](auto slabc) __attribute__((always_inline)) {  // K of slab SLAB of the current buffer -> a[192:223]
        constexpr int SLAB = decltype(slabc)::value;
        static_for6<4>([&](auto kc) __attribute__((always_inline)) {
            constexpr int ks = decltype(kc)::value;
            lds_k<A_K + 4 * ks, SLAB * 8192>(k_abs[ks]);
            lds_k<A_K + 16 + 4 * ks, SLAB * 8192 + 4096>(k_abs[ks]);
        });
    };
    auto read_v = [&](auto slabc) __attribute__((always_inline)) {  // V^T of slab SLAB -> a[224:255]
        constexpr int SLAB = decltype(slabc)::value;
        static_for6<8>([&](auto dc) __attribute__((always_inline)) {
            constexpr int dt = decltype(dc)::value;
            lds_v<A_V + 4 * dt, SLAB * 8192>(v_abs[dt]);
            lds_v<A_V + 4 * dt + 2, SLAB * 8192 + 4096>(v_abs[dt]);
        });
    };

    // ---- architectural-register state ----
    f32x4_t acc_l[F6_QT];
    float m_run[F6_QT], m_thr[F6_QT];
#pragma unroll
    for (int qt = 0; qt < F6_QT; ++qt) {
        acc_l[qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        m_run[qt] = -INFINITY;
        m_thr[qt] = -INFINITY;
    }
    u32x4_t ones_u = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    asm volatile("" : "+v"(ones_u));
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_u);
    f32x4_t s0[2], s1[2];  // S of even / odd units
    u32x4_t p0 = {0u, 0u, 0u, 0u}, p1 = {0u, 0u, 0u, 0u};  // P of even / odd units (eight bf16 each)
    asm volatile("" ::"s"(scale_log2e), "s"(lk), "s"(off));
    if (ntiles <= 0) return;  // (cannot happen: qbase < lq)


    // exact part of a unit's softmax (mask, row maximum over the four lane groups, new reference point, rescale of O and l):
    // taken on masked units, on a query tile's first unit and when the lazy test fails.  Returns the reference point to use.
    auto slow_path = [&](f32x4_t (&s)[2], int kbase, auto qc, bool need_mask, float ms) __attribute__((always_inline)) -> float {
        constexpr int qt = decltype(qc)::value;
        float mtmp;
        asm("v_mov_b32 %1, %0\n\t"
            "s_nop 1\n\t"
            "v_permlane16_swap_b32 %0, %1\n\t"
            "v_max_f32 %0, %0, %1\n\t"
            "v_mov_b32 %1, %0\n\t"
            "s_nop 1\n\t"
            "v_permlane32_swap_b32 %0, %1\n\t"
            "v_max_f32 %0, %0, %1"
            : "+v"(ms), "=&v"(mtmp));
        const float m_new = ms > m_run[qt] ? ms : m_run[qt];
        float m_use = m_new;
        if (need_mask) m_use = m_new == -INFINITY ? 0.f : m_new;
        if (__any(m_new > m_run[qt])) {
            const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_use);
            static_for6<32>([&](auto rc) __attribute__((always_inline)) {
                constexpr int R = A_O + 32 * qt + decltype(rc)::value;
                const float o = aread<R>() * alpha;
                awrite<R>(__float_as_uint(o));
            });
            acc_l[qt] *= alpha;
        }
        m_run[qt] = m_new;
        m_thr[qt] = m_new + 8.0f;
        return m_use;
    };
    auto maxchain = [&](f32x4_t (&s)[2]) __attribute__((always_inline)) -> float {
        float mloc;
        asm("v_max3_f32 %0, %1, %2, %3\n\t"
            "v_max3_f32 %0, %0, %4, %5\n\t"
            "v_max3_f32 %0, %0, %6, %7\n\t"
            "v_max_f32 %0, %0, %8\n\t"
            "v_mul_f32 %0, %9, %0"
            : "=&v"(mloc)
            : "v"(s[0][0]), "v"(s[0][1]), "v"(s[0][2]), "v"(s[0][3]), "v"(s[1][0]), "v"(s[1][1]), "v"(s[1][2]), "v"(s[1][3]),
              "s"(scale_log2e));
        return mloc;
    };

    // decision part of a unit's softmax (unit = query tile qt x the 32 keys from kbase; its S in s): mask, lazy test, and on
    // the exact path the new reference point + rescale.  Returns the reference point the unit's exponentials use.
    auto decide = [&](auto qc, auto intc, f32x4_t (&s)[2], int kbase, bool have_ms, float ms_in) __attribute__((always_inline)) -> float {
        constexpr int qt = decltype(qc)::value;
        constexpr bool INT = decltype(intc)::value;
        const int row_first = r0w + 16 * qt;
        const bool need_mask = INT ? false : (kbase + 31 > row_first + off || kbase + 31 >= lk);
        if (need_mask) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = kbase + 4 * g + 16 * tt + r;
                    if (j > jmax[qt] || j >= lk) s[tt][r] = -INFINITY;
                }
        }
        const float ms = (have_ms && !need_mask) ? ms_in : maxchain(s);
        const bool first_unit = INT ? false : kbase == 0;
        if (need_mask || first_unit || !__all(ms <= m_thr[qt])) return slow_path(s, kbase, qc, need_mask, ms);
        return m_run[qt];
    };

    // ---- tiles of a wave, in key order: [0, n_int) INTERIOR (every key visible to every row of the wave: no masks),
    // [n_int, n_live) DIAGONAL (masked; the last tile with a visible key is always of this kind), [n_live, ntiles) DEAD (the
    // wave only keeps staging and meeting the barriers for the waves whose rows reach further).  Every step of a live tile has
    // the same straight-line shape — S(next unit) MFMAs with this unit's fma / exp2 in their shadow, then P.V(previous unit)
    // MFMAs with this unit's converts and the next unit's maximum chain in theirs, then the next unit's decision — so there
    // are no "is there a predecessor / successor" conditions: before the first unit P = 0 and V = 0 (the MFMAs add nothing),
    // a masked-out unit has P = exp2(-inf) = 0, and only the very last unit's successor is conditional.
    const int n_live = max(1, min(ntiles, (r0w + 63 + off) / F6_BK + 1));
    const int n_int = max(0, min(min((r0w + off + 1) >> 6, lk >> 6), n_live - 1));
    float m_cur = 0.f;
    auto step = [&](auto uc, auto kindc, int t) __attribute__((always_inline)) {
        constexpr int U = decltype(uc)::value;
        constexpr bool INT = decltype(kindc)::value;
        constexpr int V = (U + 1) % 8, vslab = V / 4, vqt = V % 4;     // the unit whose S is issued now
        constexpr int PQT = (U + 3) % 4;                               // query tile of the previous unit
        const int kb0 = t * F6_BK;
        const bool more = t + 1 < ntiles, more2 = t + 2 < ntiles;
        f32x4_t (&sc)[2] = (U & 1) ? s1 : s0;   // S of this unit
        f32x4_t (&sn)[2] = (U & 1) ? s0 : s1;   // S of the next unit
        u32x4_t& pc = (U & 1) ? p1 : p0;        // P of this unit (written here)
        const u32x4_t& pp = (U & 1) ? p0 : p1;  // P of the previous unit (consumed here)
        const int kbase_n = (U == 7) ? kb0 + F6_BK : kb0 + 32 * vslab;
        const bool has_next = (INT || U != 7) ? true : t + 1 < n_live;  // (a diagonal tile's last unit may be the wave's last)
        const float m_use = m_cur;
        // ---- (1) K of the unit whose S is issued now must have landed when it opens a slab
        if constexpr (V == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (V == 0) {
            if (more) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- (2) S(V) MFMAs: MFMA j carries the fma of this unit's score j and the exp2 of score j - 1 (the exponential of
        // the last score rides behind the row-sum MFMA below)
        float xs[8] = {sc[0][0], sc[0][1], sc[0][2], sc[0][3], sc[1][0], sc[1][1], sc[1][2], sc[1][3]};
        if (has_next) {
            static_for6<8>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value, tt = j & 1, ks = j >> 1;  // the two 16-key chains alternate
                constexpr int KLO = A_K + 16 * tt + 4 * ks, QLO = A_Q + 16 * vqt + 4 * ks;
                constexpr int SAFE = (INT || U != 7) ? 0 : (ks == 3 ? 3 : 1);  // (the one place where two paths write xs / sn)
                if constexpr (j == 0) gap_s_fma<KLO, QLO, true>(sn[tt], xs[0], scale_log2e, m_use);
                else gap_s_fma_exp<KLO, QLO, ks == 0, SAFE>(sn[tt], xs[j], xs[j - 1], scale_log2e, m_use);
            });
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float arg = fmaf(xs[j], scale_log2e, -m_use);
                if (j < 7) xs[j] = __builtin_amdgcn_exp2f(arg);
                else xs[j] = arg;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // K regs are free once the last S MFMA of a slab is issued: request the next K one step ahead of its use
        if constexpr (U == 2) read_k(std::integral_constant<int, 1>{});
        if constexpr (U == 6) {
            if (more) {
                // tile t+1 was staged a whole tile ago; behind the barrier every wave is done with this buffer (its last LDS
                // read, V of slab 1, was issued in step 4 and waited for in step 5), which tile t+2 may then overwrite
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (more2) stage(t & 1, t + 2);
                const int flip = (t & 1) ? -F6_STAGE : F6_STAGE;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) k_abs[ks] += flip;
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) v_abs[dt] += flip;
                read_k(std::integral_constant<int, 0>{});
            }
        }
        // ---- (3) V of the previous unit's slab must have landed when that unit opened a slab (units 0 and 4)
        if constexpr (U == 1 || U == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ---- (4) P.V of the previous unit; in its gaps the last exponential, P(U) -> bf16 (four packed converts) and, on
        // interior tiles, the maximum chain of the next unit's scores (their MFMAs are at least five MFMAs back by then)
        float ms_n = 0.f;
        unsigned pcu[4];
        if constexpr (INT || U != 7) gap_l_exp(acc_l[PQT], ones, pp, xs[7]);
        else gap_l_exp<true>(acc_l[PQT], ones, pp, xs[7]);
        gap_pv_cvt<A_O + 32 * PQT + 0, A_V + 0>(pp, pcu[0], xs[0], xs[1]);
        gap_pv_cvt<A_O + 32 * PQT + 4, A_V + 4>(pp, pcu[1], xs[2], xs[3]);
        gap_pv_cvt<A_O + 32 * PQT + 8, A_V + 8>(pp, pcu[2], xs[4], xs[5]);
        gap_pv_cvt<A_O + 32 * PQT + 12, A_V + 12>(pp, pcu[3], xs[6], xs[7]);
        if constexpr (INT) {
            gap_pv_max_a<A_O + 32 * PQT + 16, A_V + 16>(pp, ms_n, sn[0][0], sn[0][1], sn[0][2], sn[0][3], sn[1][0]);
            gap_pv_max_b<A_O + 32 * PQT + 20, A_V + 20>(pp, ms_n, sn[1][1], sn[1][2], sn[1][3]);
            gap_pv_max_c<A_O + 32 * PQT + 24, A_V + 24>(pp, ms_n, scale_log2e);
        } else {
            gap_pv<A_O + 32 * PQT + 16, A_V + 16>(pp);
            gap_pv<A_O + 32 * PQT + 20, A_V + 20>(pp);
            gap_pv<A_O + 32 * PQT + 24, A_V + 24>(pp);
        }
        gap_pv<A_O + 32 * PQT + 28, A_V + 28>(pp);
        pc = u32x4_t{pcu[0], pcu[1], pcu[2], pcu[3]};
        // V regs are free once the last P.V MFMA of a slab is issued (previous unit = unit 3 / 7): request the next V
        if constexpr (U == 4) read_v(std::integral_constant<int, 1>{});
        if constexpr (U == 0) read_v(std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        // ---- (5) decide the next unit (nine MFMAs lie between its scores and this read).  Unit 0 of the NEXT tile need not be
        // interior even when this tile is: its first keys may pass this wave's first rows
        if (has_next) {
            if constexpr (U == 7) m_cur = decide(std::integral_constant<int, 0>{}, std::false_type{}, sn, kbase_n, INT, ms_n);
            else m_cur = decide(std::integral_constant<int, vqt>{}, kindc, sn, kbase_n, INT, ms_n);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // a tile this wave has no visible key in: keep the ring turning for the others
    auto dead_tile = [&](int t) __attribute__((always_inline)) {
        if (t + 1 < ntiles) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (t + 2 < ntiles) stage(t & 1, t + 2);
        }
    };

    // ---- prologue ----
    static_for6<32>([&](auto rc) __attribute__((always_inline)) { awrite<A_V + decltype(rc)::value>(0u); });  // V = 0 for the first (empty) P.V
    stage(0, 0);
    if (ntiles > 1) {
        stage(1, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    read_k(std::integral_constant<int, 0>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    static_for6<8>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value, tt = j & 1, ks = j >> 1;
        if constexpr (ks == 0) mfma_s_first<A_K + 16 * tt + 4 * ks, A_Q + 4 * ks>(s0[tt]);
        else mfma_s<A_K + 16 * tt + 4 * ks, A_Q + 4 * ks>(s0[tt]);
    });
    asm volatile("s_nop 15" ::: "memory");  // S(0) is read right below: its MFMAs must have drained (no MFMAs in between here)
    asm volatile("s_nop 15" ::: "memory");
    asm volatile("s_nop 15" ::: "memory");
    asm volatile("s_nop 15" ::: "memory");
    m_cur = decide(std::integral_constant<int, 0>{}, std::false_type{}, s0, 0, false, 0.f);

    auto tile = [&](auto kindc, int t) __attribute__((always_inline)) {
        static_for6<8>([&](auto uc) __attribute__((always_inline)) { step(uc, kindc, t); });
    };
    {
        int t = 0;
        for (; t < n_int; ++t) tile(std::true_type{}, t);
        for (; t < n_live; ++t) tile(std::false_type{}, t);
        // the last unit's P.V
        gap_l<true>(acc_l[3], ones, p1);
        static_for6<8>([&](auto dc) __attribute__((always_inline)) {
            constexpr int dt = decltype(dc)::value;
            gap_pv<A_O + 96 + 4 * dt, A_V + 4 * dt>(p1);
        });
        for (; t < ntiles; ++t) dead_tile(t);
    }
    asm volatile("s_nop 15" ::: "memory");
    asm volatile("s_nop 15" ::: "memory");
    asm volatile("s_nop 15" ::: "memory");
    asm volatile("s_nop 15" ::: "memory");

    // ---- epilogue: O / l, bf16, 8-byte stores (row = query, 4 consecutive hd columns per lane and block) ----
    static_for6<F6_QT>([&](auto qc) __attribute__((always_inline)) {
        constexpr int qt = decltype(qc)::value;
        const float l = acc_l[qt][0];
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        const bool ok = qi[qt] < lq;
        bf16_t* orow = out + ((int64_t)(q0 + (ok ? qi[qt] : 0)) * nh + head) * F6_HD;
        static_for6<8>([&](auto dc) __attribute__((always_inline)) {
            constexpr int dt = decltype(dc)::value;
            constexpr int R = A_O + 32 * qt + 4 * dt;
            const u32x2_t o = {pack2bf(aread<R>() * inv, aread<R + 1>() * inv), pack2bf(aread<R + 2>() * inv, aread<R + 3>() * inv)};
            if (ok) *reinterpret_cast<u32x2_t*>(orow + 16 * dt + 4 * g) = o;
        });
    });
}

}  // namespace

// host side: called by glm_varlen_attn_prefill (prefill_attn.hip) when option fa_rows64 selects this kernel; arguments are
// validated there
int glm_launch_prefill_attn64(const void* q, const void* k, const void* v, const int32_t* cu_seqlens_q,
                              const int32_t* cu_seqlens_k, const int32_t* block_tables, int64_t bt_stride, void* out, int64_t B,
                              int max_seqlen_q, int nh, int nkv, int block_size, float scale, hipStream_t st) {
    const int paged = block_tables != nullptr;
    if (paged) GLM_SET_MAX_LDS(prefill_attn64_kernel<true>, 2 * F6_STAGE + F6_MAX_BT * 4, "glm_varlen_attn_prefill(64)");
    else GLM_SET_MAX_LDS(prefill_attn64_kernel<false>, 2 * F6_STAGE + F6_MAX_BT * 4, "glm_varlen_attn_prefill(64)");
    const int nqblk = (max_seqlen_q + F6_BQ - 1) / F6_BQ;
    const int64_t total = (int64_t)nqblk * nh * B;
    GLM_CHECK_ARG(total < (int64_t)1 << 31, GLM_ERR_UNSUPPORTED, "glm_varlen_attn_prefill: grid too large");
    const int xcd_groups = (B * nkv) % 8 == 0 ? (int)(B * nkv / 8) : 0;
    const size_t lds = 2 * F6_STAGE + (paged ? (size_t)bt_stride * 4 : 0);
    const float sl2 = scale * 1.4426950408889634f;
    if (paged)
        hipLaunchKernelGGL((prefill_attn64_kernel<true>), dim3((unsigned)total), dim3(F6_NT), lds, st, (const bf16_t*)q,
                           (const bf16_t*)k, (const bf16_t*)v, cu_seqlens_q, cu_seqlens_k, block_tables, bt_stride,
                           (bf16_t*)out, nh, nkv, block_size, sl2, nqblk, (int)B, xcd_groups);
    else
        hipLaunchKernelGGL((prefill_attn64_kernel<false>), dim3((unsigned)total), dim3(F6_NT), lds, st, (const bf16_t*)q,
                           (const bf16_t*)k, (const bf16_t*)v, cu_seqlens_q, cu_seqlens_k, block_tables, bt_stride,
                           (bf16_t*)out, nh, nkv, block_size, sl2, nqblk, (int)B, xcd_groups);
    GLM_CHECK_LAUNCH("glm_varlen_attn_prefill(64)");
    return GLM_OK;
}
