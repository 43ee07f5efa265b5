// prefill_attn.hip — varlen causal prefill attention (a9, reference:
// nanovllm/layers/attention.py:85-100 → flash_attn_varlen_func(causal=True[, block_table])).
// gfx950 only, head_dim 128.
//
// Flash-style: the score matrix is never materialised.  A workgroup owns 128 query rows of one
// (sequence, query head); each of its 4 waves takes two 16-row query tiles and walks the 64-key tiles
// that the bottom-right-aligned causal mask leaves visible.  K/V tiles are staged once per workgroup
// by LDS-DMA with the same source-side swizzles as the decode kernel (K: chunk ^ (key & 15) for
// ds_read_b128 row reads, V: chunk ^ ((key & 7) << 1) for ds_read_b64_tr_b16 transposed reads),
// double-buffered, one barrier per tile.  Math per wave is the decode kernel's: S^T = K·Q^T with keys
// on the MFMA row axis and 16 query rows on the lane axis, fp32 online softmax per lane column, then
// O^T += V^T·P^T straight from registers.
//
// What the loop is built around (ablation on MI355X, tools/ablate_prefill_attn.sh): the MFMAs and the
// exp2 are hidden; LDS fragment reads, the staging traffic and plain VALU instruction count are what
// cost time.  Hence: every K/V fragment read from LDS feeds two query tiles; LDS addresses are one
// register per fragment plus immediates; the non-paged staging path is a scalar base plus constant
// per-lane offsets; and the work is software-pipelined in (32-key slab, query tile) units — the S
// MFMAs of unit u+1 are issued before the softmax of unit u.
//
// K/V come either from contiguous [Tk, nkv, hd] tensors or from the paged cache through
// block_tables (prefix-cache hit, attention.py:86-87).
#include "common.h"
#include "gemm_internal.h"
#include <type_traits>
#include <stdlib.h>

// timing-only ablations (results are wrong): 1 no exp, 2 no barrier, 4 no S MFMA, 8 no PV MFMA, 16 no LDS reads,
// 32 no K/V staging after the first two tiles, 64 every unit live and unmasked (no liveness / mask predicates)
#ifndef GLM_FA_ABLATE
#define GLM_FA_ABLATE 0
#endif
#ifndef GLM_FA_PRIO
#define GLM_FA_PRIO 0
#endif
#ifndef GLM_FA_NW
#define GLM_FA_NW 4
#endif
#ifndef GLM_FA_QT
#define GLM_FA_QT 2   // 16-row query tiles per wave; 4 = 64 rows per wave at ONE wave per SIMD (generic unit loop)
#endif
#ifndef GLM_FA_GENERIC
#define GLM_FA_GENERIC (GLM_FA_QT != 2)  // 1: the unit loop written over FA_QT instead of the hand-unrolled two-tile one
#endif

// GLM_FA_STAMP=1 (timing study only): wave 0 of workgroup 0 accumulates s_memtime deltas per loop segment
#ifndef GLM_FA_LAZYMAX
#define GLM_FA_LAZYMAX 1
#endif
#ifndef GLM_FA_STAMP
#define GLM_FA_STAMP 0
#endif
// r04: 1 = a separate loop for the INTERIOR tiles of a wave — tiles every one of its rows sees in full — that carries no
// liveness or mask predicate, followed by the general loop for the diagonal / last tiles; the staging sources advance as
// running pointers.  (A form unrolled over the two LDS buffers as well, buffer select as an immediate of every LDS access,
// made hipcc spill 520 registers.)  The r02 loop evaluated ~10 scalar predicates per
// tile and rebuilt the 64-bit source address per staging slot: ~290 scalar instructions per tile and wave next to 68 MFMAs,
// and a wave issues one instruction per ~4 cycles whatever its kind.  0 = the r02 loop (kept for A/B).
#ifndef GLM_FA_EDGE
#define GLM_FA_EDGE 0  // 1: unmasked tiles behind the strict interior range take a body without liveness / mask tests (2: the
#endif                 // first tile as well).  Measured r04, one box, 3 rounds: 1655-1658 us against 1636 us without — not kept

#ifndef GLM_FA_LOOP
#define GLM_FA_LOOP 1
#endif
// r04 loop only: 1 = V of slab 1 is requested BEFORE the S MFMAs of unit (slab 1, q1) and waited for behind the softmax of
// unit (slab 1, q0) (the tile's barrier moves behind that softmax too); 0 = requested behind those MFMAs and waited for at once
#ifndef GLM_FA_VEARLY
#define GLM_FA_VEARLY 1
#endif
// r04: 1 = Q pre-scaled by scale*log2(e) and the S accumulators INITIALISED with minus the softmax reference point, so
// that the MFMA chain delivers S' = c*s - m and the fast path is p = exp2(S') with no per-element FMA (8 of a unit's ~30
// vector instructions) and a constant threshold; 0 = the r02/r03 arithmetic (fp32 scale after the dot product)
#ifndef GLM_FA_CINIT
#define GLM_FA_CINIT 0   // measured r04 (profiles/r04/prefill_attn_ab.txt): 913-918 vs 905-908 TFLOP/s (+0.7 %), and one element
                         // in 10^4-10^6 leaves the parity tolerance (the second rounding of q) => off
#endif
// TIMING STUDY ONLY (results unchecked against overflow): with GLM_FA_CINIT, skip the lazy-maximum test once a query tile
// has a reference point — what an optimistic loop with an overflow check at the end of the workgroup would execute
#ifndef GLM_FA_NOCHECK
#define GLM_FA_NOCHECK 0
#endif
#ifndef GLM_FA_DEFAULT_MFMA32
#define GLM_FA_DEFAULT_MFMA32 0  // kernel when GLM_FA_MFMA32 is unset: 0 = 16x16x32 (r02), 1 = 32x32x16
#endif
constexpr int FA_DEFAULT_MFMA32 = GLM_FA_DEFAULT_MFMA32;
// GLM_FA_WGSTAMP=1 (timing study only): every workgroup's wave 0 records s_memrealtime (100 MHz) at entry, behind the start-up
// barrier, behind the tile loop and behind the output stores
#ifndef GLM_FA_WGSTAMP
#define GLM_FA_WGSTAMP 0
#endif
#if GLM_FA_WGSTAMP
__device__ unsigned long long g_fa_wgstamps[16384][4];
#define FA_WG(i) do { if (tid == 0 && blockIdx.x < 16384) g_fa_wgstamps[blockIdx.x][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FA_WG(i) do {} while (0)
#endif
#if GLM_FA_STAMP
__device__ unsigned long long g_fa_stamps[16];
#define FA_T(i) do { unsigned long long t_ = __builtin_readcyclecounter(); if (i) tacc[(i) - 1] += t_ - tprev; tprev = t_; } while (0)
#else
#define FA_T(i) do {} while (0)
#endif

// r05 (r04 loop only): 1 = (a) a wave's two 16-row query tiles lie 64 rows apart (rows 16 w .. and 64 + 16 w ..) instead of
// back to back, so that the two half-dead tiles on the causal diagonal cost every wave the same (before: waves 0, 1 sat out
// the last tile); (b) the predicate-free loop starts at the block's FIRST tile (its rows have no reference point yet, so
// the lazy-maximum vote fails by itself and the first unit takes the exact path; the start-up stages pieces 0, 1 of tile 1
// only and tile 0 stages pieces 2, 3 like any other tile) and runs up to the diagonal whenever the sequence's last tile is
// a full one: two general tiles per block instead of four.  Same arithmetic per (row, key): bit-identical to 0.
#ifndef GLM_FA_R05
#define GLM_FA_R05 (GLM_FA_LOOP == 1 && !GLM_FA_STAMP && !GLM_FA_GENERIC && GLM_FA_NW == 4 && !GLM_FA_EDGE)
#endif
#ifndef GLM_FA_PKFMA
#define GLM_FA_PKFMA 0   // 1 (measured r05, not kept): the eight s * c - m of a unit as four v_pk_fma_f32 — 14 fewer vector instructions per
#endif                   // tile and 1.3 % (4 x 4096) / 2.1 % (1 x 16384) SLOWER, three interleaved rounds (profiles/r05/prefill_attn_pkfma_ab.txt)
#ifndef GLM_FA_STORE16
#define GLM_FA_STORE16 1   // r05: 16-byte output stores behind a lane-row exchange (0 = the 8-byte stores of r02-r04)
#endif
#ifndef GLM_FA_NOSCHED
#define GLM_FA_NOSCHED 0   // 1 (study): drop the scheduling fences of the unit loops and let hipcc interleave MFMA and VALU itself
#endif
#if GLM_FA_NOSCHED
#define FA_SCHED_FENCE() do {} while (0)
#else
#define FA_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

namespace {

constexpr int FA_HD = 128, FA_ROWB = 256;
constexpr int FA_QT = GLM_FA_QT;           // 16-row query tiles per wave
constexpr int FA_NW = GLM_FA_NW;           // waves per workgroup
constexpr int FA_NT = 64 * FA_NW;
constexpr int FA_BQ = 16 * FA_QT * FA_NW, FA_BK = 64;
constexpr int FA_PPW = 16 / FA_NW;         // 1 KiB staging pieces per wave, for K and for V
constexpr int FA_TILE = FA_BK * FA_ROWB;   // 16 KiB per K or V tile
constexpr int FA_STAGE = 2 * FA_TILE;
constexpr int FA_MAX_BT = 4096;
// first row (inside the workgroup's 128-row block) of query tile qt of wave w
__device__ __forceinline__ constexpr int fa_row0(int w, int qt) {
#if GLM_FA_R05
    return 16 * w + 64 * qt;
#else
    return (16 * FA_QT) * w + 16 * qt;
#endif
}

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
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(const lds_void_t*)p;
}
template <int N, class F>
__device__ __forceinline__ void fa_static_for(F&& f) {
    if constexpr (N > 0) {
        fa_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// launch bound of 2 waves per SIMD caps the kernel at 256 registers, which makes hipcc pick the VGPR form of the
// MFMAs: with the default (512) the accumulators live in AGPRs and every slab pays ~40 v_accvgpr copies
template <bool PAGED>
__global__ __launch_bounds__(FA_NT, (FA_QT >= 4 ? 1 : 2)) void prefill_attn_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const int32_t* __restrict__ cu_q, const int32_t* __restrict__ cu_k,
    const int32_t* __restrict__ block_tables, int64_t bt_stride, bf16_t* __restrict__ out, int nh,
    int nkv, int block_size, float scale_log2e, int nqblk, int nseq, int xcd_groups) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* bt_lds = reinterpret_cast<int*>(smem + 2 * FA_STAGE);
    // Work order.  Workgroups are handed to the 8 XCDs round-robin in linear-id order and, inside an XCD, to its
    // shader engines the same way, so a causal grid with the query block as the fast index gives some engines only
    // long workgroups and others only short ones (measured: CUs idle a third of the time).  Instead the linear id
    // is decoded so that (a) all workgroups in flight have the same query block, longest first, (b) an XCD always
    // serves the same (sequence, kv head) groups, whose K/V tiles then stay in its own L2.
    int qblk, head;
    int64_t b;
    {
        const int L = blockIdx.x;
        const int hpg = nh / nkv;
        if (xcd_groups > 0) {
            const int xcd = L & 7, j = L >> 3;
            const int per_q = xcd_groups * hpg;  // workgroups per query block on one XCD
            const int sidx = j / per_q, r = j - sidx * per_q;
            const int group = (r / hpg) * 8 + xcd;  // (sequence, kv head) pair
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
    FA_WG(0);
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
        for (int i = tid; i < nblk; i += FA_NT) bt_lds[i] = block_tables[b * bt_stride + i];
    }

    // Q^T fragments: lane (c,g), tile qt = query row qbase + 32*wid + 16*qt + c, k-step ks → hd [32ks+8g, +8)
    int qi[FA_QT];
    bf16x8_t qf[FA_QT][4];
#pragma unroll
    for (int qt = 0; qt < FA_QT; ++qt) {
        qi[qt] = qbase + fa_row0(wid, qt) + c;
        const bool ok = qi[qt] < lq;
        const bf16_t* qrow = q + ((int64_t)(q0 + (ok ? qi[qt] : 0)) * nh + head) * FA_HD;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            u32x4_t t = {0, 0, 0, 0};
            if (ok) t = *reinterpret_cast<const u32x4_t*>(qrow + 32 * ks + 8 * g);
#if GLM_FA_CINIT
            // q' = bf16(q * scale * log2 e): one more bf16 rounding of the query (relative 2^-9 per element, ~1e-3 of a
            // score's own standard deviation after the 128-term dot product), once per workgroup
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = pack2bf(bflo(t[i]) * scale_log2e, bfhi(t[i]) * scale_log2e);
#endif
            qf[qt][ks] = __builtin_bit_cast(bf16x8_t, t);
        }
    }
    __syncthreads();
#pragma unroll
    for (int qt = 0; qt < FA_QT; ++qt)
        asm volatile("" : "+v"(qf[qt][0]), "+v"(qf[qt][1]), "+v"(qf[qt][2]), "+v"(qf[qt][3]));

    // staging: wave w issues FA_PPW instructions for K and for V; instruction i covers tile rows
    // (FA_PPW*w+i)*4 .. +3 (4 keys x 256 B); lane → row +(lane>>4), 16-byte position lane&15.
    // Contiguous K/V: the source is a wave-uniform tile base plus a per-lane byte offset that is the same for
    // every full tile (only the last, partial tile clamps rows), so a tile costs no vector address math.
    const int64_t row_bytes = (int64_t)nkv * FA_ROWB;
    unsigned soff_k[FA_PPW], soff_v[FA_PPW];
    auto stage_offsets = [&](int t) {
        const int r4 = lane >> 4, pch = lane & 15;
#pragma unroll
        for (int i = 0; i < FA_PPW; ++i) {
            const int row = (wid * FA_PPW + i) * 4 + r4;
            const int rc = min(t * FA_BK + row, lk - 1) - t * FA_BK;  // masked keys re-read a valid row
            soff_k[i] = (unsigned)(rc * (int)row_bytes) + ((pch ^ (row & 15)) << 4);
            soff_v[i] = (unsigned)(rc * (int)row_bytes) + ((pch ^ ((row & 7) << 1)) << 4);
        }
    };
    if (!PAGED) stage_offsets(0);
    // one K piece and one V piece (1 KiB each) of tile t; pieces are spread over the loop's softmax phases, where
    // an LDS-DMA issue is cheapest (back to back after the barrier they cost ~115 cycles apiece)
    auto stage_piece = [&](int buf, int t, int i) {
        char* kd = smem + buf * FA_STAGE + (wid * FA_PPW + i) * 1024;
        char* vd = kd + FA_TILE;
        if (GLM_FA_ABLATE & 32) return;
        if (!PAGED) {
            if (i == 0 && (t + 1) * FA_BK > lk) stage_offsets(t);  // uniform; last tile only
            const int64_t tb = ((int64_t)(k0 + t * FA_BK) * nkv + kvh) * FA_ROWB;
            glds16(reinterpret_cast<const char*>(k) + tb + soff_k[i], kd);
            glds16(reinterpret_cast<const char*>(v) + tb + soff_v[i], vd);
        } else {
            const int r4 = lane >> 4, pch = lane & 15;
            const int row = (wid * FA_PPW + i) * 4 + r4;
            int key = t * FA_BK + row;
            key = key < lk ? key : lk - 1;
            const int pb = bt_lds[key / block_size];
            const int64_t tok = (int64_t)pb * block_size + key % block_size;
            const int64_t ro = (tok * nkv + kvh) * FA_HD;
            glds16(k + ro + ((pch ^ (row & 15)) << 3), kd);
            glds16(v + ro + ((pch ^ ((row & 7) << 1)) << 3), vd);
        }
    };
    auto stage = [&](int buf, int t) {
#pragma unroll
        for (int i = 0; i < FA_PPW; ++i) stage_piece(buf, t, i);
    };
    // piece i of a tile is issued in slot i: slots 0,1 = units 3,4 of the tile two before it (right after the
    // barrier that frees its buffer), slots 2,3 = units 1,2 of the tile before it
    constexpr int FA_SLOTS = FA_PPW < 4 ? FA_PPW : 4;
    constexpr int FA_PPS = FA_PPW / FA_SLOTS;  // pieces per slot
    auto stage_slot = [&](int buf, int t, int slot) {
        if (slot >= FA_SLOTS) return;
#pragma unroll
        for (int i = 0; i < FA_PPS; ++i) stage_piece(buf, t, slot * FA_PPS + i);
    };

    // LDS fragment addresses of the CURRENT buffer: one register per fragment, slab / half-slab as immediates.
    // They flip between the two buffers once per tile.
    unsigned k_abs[4], v_abs[8];
    {
        const unsigned base = lds_addr(smem);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) k_abs[ks] = base + c * FA_ROWB + (((4 * ks + g) ^ c) << 4);
        const int qq = (lane >> 2) & 3, pp = lane & 3;
        const int row = 4 * g + qq;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt)
            v_abs[dt] = base + FA_TILE + row * FA_ROWB + (((2 * dt + (pp >> 1)) ^ ((row & 7) << 1)) << 4) +
                        ((pp & 1) << 3);
    }

    f32x4_t acc[FA_QT][8], acc_l[FA_QT];
    float m_run[FA_QT], m_thr[FA_QT];  // softmax reference point (log2 domain) and m_run + 8
    // `exact` = false would skip the lazy-maximum test of unmasked units (an "optimistic" pass with ONE overflow check at
    // the end of the workgroup and a second, exact pass behind a detected overflow).  Built and measured in r04: the hot
    // loop loses 10 of its ~78 instructions per unit (268 per tile), the timing-only form (GLM_FA_NOCHECK) ran +3.9 %, but
    // the safe two-pass form ran 980-990 against 990 TFLOP/s — two instantiations of the tile loops (one copy with a back
    // edge spilled 25 registers, two copies 10-21 plus scratch set-up) — and was removed again.
    const bool exact = true;
#if GLM_FA_CINIT
    // minus the reference point the S accumulators of tile qt start from (0 while the row has seen nothing); m_thr is then
    // the threshold on S' itself: 8 once a reference point exists, -inf before (the first unit takes the exact path)
    f32x4_t cinit[FA_QT];
#endif
    int jmax[FA_QT];  // last visible key of this lane's query row, per tile
#pragma unroll
    for (int qt = 0; qt < FA_QT; ++qt) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[qt][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        acc_l[qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        m_run[qt] = -INFINITY;
        m_thr[qt] = -INFINITY;
#if GLM_FA_CINIT
        cinit[qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#endif
        jmax[qt] = qi[qt] + off;
    }

#if GLM_FA_ABLATE & 16
    u32x4_t kf[2][4] = {};
    u32x2_t vf[8][2] = {};
#else
    u32x4_t kf[2][4];
    u32x2_t vf[8][2];
#endif
    // SLAB = 0/1: 32-key half of the tile (8 KiB apart); the second 16 keys of a slab are 4 KiB further
#define FA_READ_K(SLAB)                                                     \
    if (!(GLM_FA_ABLATE & 16)) {                                            \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                  \
            lds_read_b128<(SLAB) * 8192>(kf[0][ks], k_abs[ks]);             \
            lds_read_b128<(SLAB) * 8192 + 4096>(kf[1][ks], k_abs[ks]);      \
        }                                                                   \
    }
#define FA_READ_V(SLAB)                                                     \
    if (!(GLM_FA_ABLATE & 16)) {                                            \
        _Pragma("unroll") for (int dt = 0; dt < 8; ++dt) {                  \
            lds_read_tr16_b64<(SLAB) * 8192>(vf[dt][0], v_abs[dt]);         \
            lds_read_tr16_b64<(SLAB) * 8192 + 4096>(vf[dt][1], v_abs[dt]);  \
        }                                                                   \
    }
#define FA_K_REGS                                                                                          \
    "+v"(kf[0][0]), "+v"(kf[0][1]), "+v"(kf[0][2]), "+v"(kf[0][3]), "+v"(kf[1][0]), "+v"(kf[1][1]),       \
        "+v"(kf[1][2]), "+v"(kf[1][3])
#define FA_V_REGS                                                                                          \
    "+v"(vf[0][0]), "+v"(vf[0][1]), "+v"(vf[1][0]), "+v"(vf[1][1]), "+v"(vf[2][0]), "+v"(vf[2][1]),       \
        "+v"(vf[3][0]), "+v"(vf[3][1]), "+v"(vf[4][0]), "+v"(vf[4][1]), "+v"(vf[5][0]), "+v"(vf[5][1]),   \
        "+v"(vf[6][0]), "+v"(vf[6][1]), "+v"(vf[7][0]), "+v"(vf[7][1])
    auto s_mfma = [&](f32x4_t (&s)[2], int qt) {
#if GLM_FA_PRIO
        __builtin_amdgcn_s_setprio(GLM_FA_PRIO);
#endif
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
#if GLM_FA_CINIT
            s[tt] = cinit[qt];
#else
            s[tt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#endif
            if (GLM_FA_ABLATE & 4) {
                s[tt] = __builtin_bit_cast(f32x4_t, kf[tt][0]) + __builtin_bit_cast(f32x4_t, kf[tt][1]) +
                        __builtin_bit_cast(f32x4_t, kf[tt][2]) + __builtin_bit_cast(f32x4_t, kf[tt][3]);
                continue;
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                s[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[tt][ks]),
                                                               qf[qt][ks], s[tt], 0, 0, 0);
        }
#if GLM_FA_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };
    // online softmax of one unit (keys [kbase, kbase+32) x query tile qt); returns the bf16 P^T fragment and
    // leaves acc[qt] rescaled.  The running maximum lives in the scaled (log2) domain: p = exp2(s*c - m).
    auto softmax = [&](f32x4_t (&s)[2], int kbase, int qt, auto interior_c) -> bf16x8_t {
        // masking is only needed on slabs that touch the causal diagonal or the sequence end for this
        // query tile: the test is wave-uniform and false for almost every slab (and known false on an interior tile)
        const int row_first = qbase + fa_row0(wid, qt);
        // interior_c: 0 general tile, 1 unmasked tile (every key visible to every row of the wave), 2 strict interior tile
        // (unmasked and not the wave's first tile)
        constexpr int TK = (int)decltype(interior_c)::value;
        const bool need_mask = ((GLM_FA_ABLATE & 64) || TK >= 1)
                                   ? false : (kbase + 31 > row_first + off || kbase + 31 >= lk);
        if (need_mask) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = kbase + 4 * g + 16 * tt + r;
                    if (j > jmax[qt] || j >= lk) s[tt][r] = -INFINITY;
                }
        }
        // Lazy maximum: the reference point m_run only has to keep exp2 in range, not to be the exact running maximum.
        // Fast path (almost every unit after the first): every score of the slab is within 8 (log2 units) of m_run, so
        // p = exp2(s*c - m_run) <= 256 and nothing is rescaled — the lane-local maximum and one wave vote decide it.
        // Slow path: exact row maximum over the four 16-lane groups that share a query column (VALU lane swaps, one
        // asm block: hipcc pads every separate asm VALU op with an s_nop), new reference point, rescale.
        auto maxchain = [&]() -> float {
            float mloc;
#if GLM_FA_CINIT
            asm("v_max3_f32 %0, %1, %2, %3\n\t"
                "v_max3_f32 %0, %0, %4, %5\n\t"
                "v_max3_f32 %0, %0, %6, %7\n\t"
                "v_max_f32 %0, %0, %8"
                : "=&v"(mloc)
                : "v"(s[0][0]), "v"(s[0][1]), "v"(s[0][2]), "v"(s[0][3]), "v"(s[1][0]), "v"(s[1][1]), "v"(s[1][2]),
                  "v"(s[1][3]));
#else
            // (the scale multiply sits inside the block: hipcc pads the boundary of an asm VALU block with an s_nop)
            asm("v_max3_f32 %0, %1, %2, %3\n\t"
                "v_max3_f32 %0, %0, %4, %5\n\t"
                "v_max3_f32 %0, %0, %6, %7\n\t"
                "v_max_f32 %0, %0, %8\n\t"
                "v_mul_f32 %0, %9, %0"
                : "=&v"(mloc)
                : "v"(s[0][0]), "v"(s[0][1]), "v"(s[0][2]), "v"(s[0][3]), "v"(s[1][0]), "v"(s[1][1]), "v"(s[1][2]),
                  "v"(s[1][3]), "s"(scale_log2e));
#endif
            return mloc;
        };
#if GLM_FA_CINIT
        // S' = c*s - m_cur, m_cur = the reference point the accumulators started from (m_run, or 0 before the first unit)
        float ms = maxchain();
        float shift = 0.f;
#if GLM_FA_NOCHECK
        // a query tile's first unit (keys 0..31) is either unmasked — it then sets the reference point of all 16 rows — or
        // on the causal diagonal, and then every later unit of the tile is masked or dead as well
        const bool slow = need_mask || kbase == 0;
#else
        const bool slow = need_mask || !GLM_FA_LAZYMAX || !__all(ms <= m_thr[qt]);
#endif
        if (slow) {
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
            const float m_cur = m_run[qt] == -INFINITY ? 0.f : m_run[qt];
            const float cand = m_cur + ms;                                  // this unit's row maximum, absolute (log2) domain
            const float m_new = cand > m_run[qt] ? cand : m_run[qt];
            // a row with nothing visible yet keeps m_new = -inf (only possible on a masked slab): 0 is its reference point,
            // exp2(-inf - 0) = 0 and no NaN appears
            const float m_use = m_new == -INFINITY ? 0.f : m_new;
            shift = m_use - m_cur;
            if (__any(m_new > m_run[qt])) {
                const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_use);  // 0 for a row that had seen nothing (acc = 0)
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) acc[qt][dt] *= alpha;
                acc_l[qt] *= alpha;
            }
            m_run[qt] = m_new;
            m_thr[qt] = m_new == -INFINITY ? -INFINITY : 8.0f;
            const float cneg = -m_use;
            cinit[qt] = f32x4_t{cneg, cneg, cneg, cneg};
            bf16x8_t pfs;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float arg = s[tt][r] - shift;
                    const float pv = (GLM_FA_ABLATE & 1) ? arg : __builtin_amdgcn_exp2f(arg);
                    pfs[4 * tt + r] = (__bf16)pv;
                }
            return pfs;
        }
        bf16x8_t pf;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = (GLM_FA_ABLATE & 1) ? s[tt][r] : __builtin_amdgcn_exp2f(s[tt][r]);
                pf[4 * tt + r] = (__bf16)pv;
            }
#else
        float ms = 0.f;  // slab maximum, scaled (inside the asm block)
        float m_use = m_run[qt];
        // a query tile's first unit (keys 0..31) is either unmasked — it then sets the reference point of all 16 rows — or on
        // the causal diagonal, and then every later unit of the tile is masked or dead as well
        const bool first_unit = TK == 2 ? false : kbase == 0;
        bool slow0;
#if GLM_FA_NOCHECK
        slow0 = need_mask || first_unit;  // TIMING STUDY: no overflow check at all
        if (slow0) ms = maxchain();
#else
        if (need_mask || first_unit || !GLM_FA_LAZYMAX) {
            ms = maxchain();
            slow0 = true;
        } else if (exact) {
            ms = maxchain();
            slow0 = !__all(ms <= m_thr[qt]);
        } else {
            slow0 = false;  // optimistic pass: checked once, at the end of the workgroup
        }
#endif
        if (slow0) {
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
            const float m_new = ms > m_run[qt] ? ms : m_run[qt];  // neither side is NaN-quieted: v_cmp + v_cndmask
            // a row with nothing visible yet keeps m_new = -inf (only possible on a masked slab): use 0 as the
            // reference point there so that exp2(-inf - 0) = 0 and no NaN appears
            m_use = m_new;
            if (need_mask) m_use = m_new == -INFINITY ? 0.f : m_new;
            if (__any(m_new > m_run[qt])) {
                const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_use);
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) acc[qt][dt] *= alpha;
                acc_l[qt] *= alpha;
            }
            m_run[qt] = m_new;
            m_thr[qt] = m_new + 8.0f;
        }
        bf16x8_t pf;
#if GLM_FA_PKFMA
        // r05: the eight s * c - m as four packed fp32 FMAs (v_pk_fma_f32: the same IEEE fma per element, half the issue slots)
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const f32x2_t c2 = {scale_log2e, scale_log2e}, nm2 = {-m_use, -m_use};
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const f32x2_t sv = {s[tt][r], s[tt][r + 1]};
                const f32x2_t arg = __builtin_elementwise_fma(sv, c2, nm2);
                pf[4 * tt + r] = (__bf16)((GLM_FA_ABLATE & 1) ? arg[0] : __builtin_amdgcn_exp2f(arg[0]));
                pf[4 * tt + r + 1] = (__bf16)((GLM_FA_ABLATE & 1) ? arg[1] : __builtin_amdgcn_exp2f(arg[1]));
            }
#else
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float arg = fmaf(s[tt][r], scale_log2e, -m_use);
                const float pv = (GLM_FA_ABLATE & 1) ? arg : __builtin_amdgcn_exp2f(arg);
                pf[4 * tt + r] = (__bf16)pv;
            }
#endif
#endif
        return pf;
    };
    // the softmax denominator rides on the matrix pipe: a ones fragment as the A operand gives, in every row of the
    // 16x16 result, the column sums of P^T over all 32 keys of the slab (bf16-rounded p, as used in the numerator)
    u32x4_t ones_u = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    // opaque: hipcc otherwise keeps ONE register and rebuilds the quad with three v_mov in front of every unit's MFMA
    asm volatile("" : "+v"(ones_u));
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_u);
    auto pv_mfma = [&](bf16x8_t pf, int qt) {
        acc_l[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf, acc_l[qt], 0, 0, 0);
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            const u32x4_t vv = {vf[dt][0][0], vf[dt][0][1], vf[dt][1][0], vf[dt][1][1]};
            if (GLM_FA_ABLATE & 8) {
                acc[qt][dt] += __builtin_bit_cast(f32x4_t, vv) * __builtin_bit_cast(f32x4_t, pf);
                continue;
            }
            acc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vv), pf,
                                                                 acc[qt][dt], 0, 0, 0);
        }
    };

    // consume the kernel-argument scalars here: otherwise the wait for their s_load lands at the first use inside
    // the loop as an lgkmcnt(0), which also drains every LDS read that was meant to overlap the softmax
    asm volatile("" ::"s"(scale_log2e), "s"(lk), "s"(off));
#if GLM_FA_GENERIC
    // ---- the unit loop over FA_QT query tiles per wave.  A tile has 2 * FA_QT units u = (slab u / FA_QT, query tile
    // u % FA_QT); step u: S MFMAs of unit u+1 first, then LDS reads / staging, then softmax + PV of unit u.  Register
    // reuse decides where the reads go: K of slab 1 once S(slab 0, last tile) is issued (step FA_QT-2), V of slab 1 once
    // PV(slab 0, last tile) is issued (step FA_QT), K of the next tile's slab 0 once S(slab 1, last tile) is issued (step
    // 2 FA_QT - 2) — behind the tile's barrier, which therefore sits in that step.  With FA_QT = 2 this is the
    // hand-unrolled schedule below, instruction for instruction.
    if (ntiles > 0) {
        f32x4_t sbuf[2][2];  // S of unit u in sbuf[u & 1]
        stage(0, 0);
        if (ntiles > 1) {
            stage(1, 1);
            if (FA_PPW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        FA_READ_K(0);
        asm volatile("s_waitcnt lgkmcnt(0)" : FA_K_REGS : : "memory");
        s_mfma(sbuf[0], 0);
        auto live = [&](int kbase, int qt) -> bool {
            return kbase <= qbase + (16 * FA_QT) * wid + 16 * qt + 15 + off && kbase < lk;
        };
        constexpr int NU = 2 * FA_QT;
#if GLM_FA_QT >= 4 && !defined(GLM_FA_NO_PVPIPE)
        // One wave per SIMD: nothing else fills the matrix pipe while this wave runs its softmax, so PV runs ONE UNIT
        // LATE — step u issues S(u+1), the softmax of unit u and the PV MFMAs of unit u-1 with no scheduling fence
        // between them (the compiler interleaves the independent VALU and MFMA streams).  V of a slab is then read at
        // the end of the step that issued the last PV of the previous slab and waited for one step later.
        bf16x8_t pf_prev;
        bool lv_prev = false;
        for (int t = 0; t < ntiles; ++t) {
            const int kb0 = t * FA_BK;
            const bool more = t + 1 < ntiles;
            fa_static_for<NU>([&](auto uc) {
                constexpr int u = decltype(uc)::value, slab = u / FA_QT, qt = u % FA_QT, v = u + 1;
                constexpr int pq = (u + NU - 1) % NU % FA_QT;  // query tile of the previous unit
                bf16x8_t pf;
                if constexpr (v < NU) {
                    if constexpr (v == FA_QT) asm volatile("s_waitcnt lgkmcnt(0)" : FA_K_REGS : : "memory");
                    if (live(kb0 + 32 * (v / FA_QT), v % FA_QT)) s_mfma(sbuf[v & 1], v % FA_QT);
                } else if (more) {
                    asm volatile("s_waitcnt lgkmcnt(0)" : FA_K_REGS : : "memory");
                    if (live(kb0 + FA_BK, 0)) s_mfma(sbuf[0], 0);
                }
                if constexpr (u == FA_QT - 2) { FA_READ_K(1); }
                if constexpr (u == 0) { if (t >= 1 && more) stage_slot((t + 1) & 1, t + 1, 2); }
                if constexpr (u == 1) { if (t >= 1 && more) stage_slot((t + 1) & 1, t + 1, 3); }
                if constexpr (u == NU - 2) {
                    if (more) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        if (t + 2 < ntiles) stage_slot(t & 1, t + 2, 0);
                    }
                }
                if constexpr (u == NU - 1) { if (more && t + 2 < ntiles) stage_slot(t & 1, t + 2, 1); }
                const bool lv = live(kb0 + 32 * slab, qt);
                if (lv) pf = softmax(sbuf[u & 1], kb0 + 32 * slab, qt, std::false_type{});
                // V of this unit's predecessor: slab 0 was read at the end of step 0, slab 1 at the end of step FA_QT
                if constexpr (u == 1 || u == FA_QT + 1) asm volatile("s_waitcnt lgkmcnt(0)" : FA_V_REGS : : "memory");
                if (lv_prev) pv_mfma(pf_prev, pq);
                pf_prev = pf;
                lv_prev = lv;
                FA_SCHED_FENCE();
                if constexpr (u == 0) { FA_READ_V(0); }       // PV of the previous tile's last unit is issued
                if constexpr (u == FA_QT) { FA_READ_V(1); }   // PV(slab 0, last query tile) is issued
                if constexpr (u == NU - 2) {
                    if (more) {  // K of the next tile (kf is free: S(slab 1, last query tile) was issued in this step)
                        const int flip = (t & 1) ? -FA_STAGE : FA_STAGE;
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) k_abs[ks] += flip;
                        FA_READ_K(0);
                    }
                }
                if constexpr (u == NU - 1) {
                    if (more) {  // V addresses flip after the last V read of this tile (step FA_QT) and its use
                        const int flip = (t & 1) ? -FA_STAGE : FA_STAGE;
#pragma unroll
                        for (int dt = 0; dt < 8; ++dt) v_abs[dt] += flip;
                    }
                }
            });
        }
        if (lv_prev) pv_mfma(pf_prev, FA_QT - 1);
    }
#else
        for (int t = 0; t < ntiles; ++t) {
            const int kb0 = t * FA_BK;
            const bool more = t + 1 < ntiles;
            fa_static_for<NU>([&](auto uc) {
                constexpr int u = decltype(uc)::value, slab = u / FA_QT, qt = u % FA_QT, v = u + 1;
                bf16x8_t pf;
                // ---- S of the next unit
                if constexpr (v < NU) {
                    if constexpr (v == FA_QT) asm volatile("s_waitcnt lgkmcnt(0)" : FA_K_REGS : : "memory");  // K of slab 1
                    if (live(kb0 + 32 * (v / FA_QT), v % FA_QT)) s_mfma(sbuf[v & 1], v % FA_QT);
                } else if (more) {
                    asm volatile("s_waitcnt lgkmcnt(0)" : FA_K_REGS : : "memory");  // K of the next tile's slab 0
                    if (live(kb0 + FA_BK, 0)) s_mfma(sbuf[0], 0);
                }
                FA_SCHED_FENCE();
                // ---- LDS reads, staging, tile switch
                if constexpr (u == 0) { FA_READ_V(0); }
                if constexpr (u == FA_QT - 2) { FA_READ_K(1); }
                if constexpr (u == FA_QT) { FA_READ_V(1); }
                if constexpr (u == 0) { if (t >= 1 && more) stage_slot((t + 1) & 1, t + 1, 2); }
                if constexpr (u == 1) { if (t >= 1 && more) stage_slot((t + 1) & 1, t + 1, 3); }
                if constexpr (u == NU - 2) {
                    if constexpr (u == FA_QT) asm volatile("s_waitcnt lgkmcnt(0)" : FA_V_REGS : : "memory");  // the last read
                    if (more) {
                        // tile t+1 was issued a whole tile ago; after the barrier every wave is done with this buffer,
                        // which tile t+2 may then overwrite
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (!(GLM_FA_ABLATE & 2)) __builtin_amdgcn_s_barrier();
                        if (t + 2 < ntiles) stage_slot(t & 1, t + 2, 0);
                        const int flip = (t & 1) ? -FA_STAGE : FA_STAGE;
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) k_abs[ks] += flip;
#pragma unroll
                        for (int dt = 0; dt < 8; ++dt) v_abs[dt] += flip;
                        FA_READ_K(0);
                    }
                }
                if constexpr (u == NU - 1) { if (more && t + 2 < ntiles) stage_slot(t & 1, t + 2, 1); }
                // ---- softmax and PV of this unit
                const bool lv = live(kb0 + 32 * slab, qt);
                if (lv) pf = softmax(sbuf[u & 1], kb0 + 32 * slab, qt, std::false_type{});
                if constexpr (u == 0) {
                    if constexpr (FA_QT == 2) asm volatile("s_waitcnt lgkmcnt(8)" : FA_V_REGS : : "memory");  // K of slab 1 may be in flight
                    else asm volatile("s_waitcnt lgkmcnt(0)" : FA_V_REGS : : "memory");
                }
                if constexpr (u == FA_QT && u != NU - 2) asm volatile("s_waitcnt lgkmcnt(0)" : FA_V_REGS : : "memory");
                if (lv) pv_mfma(pf, qt);
                FA_SCHED_FENCE();
            });
        }
    }
#endif  // PV one unit late
#else
    if (ntiles > 0) {
        static_assert(FA_QT == 2, "the unit schedule below is written for two query tiles per wave");
        f32x4_t sa[2], sb[2];  // S of the unit being softmaxed / of the next unit
        stage(0, 0);
        if (ntiles > 1) {
#if GLM_FA_R05
            // pieces 0, 1 of tile 1; tile 0 issues pieces 2, 3 in its first two units like every later tile
            stage_slot(1, 1, 0);
            stage_slot(1, 1, 1);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#else
            stage(1, 1);
            if (FA_PPW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#endif
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        FA_WG(1);
        FA_READ_K(0);
        asm volatile("s_waitcnt lgkmcnt(0)" : FA_K_REGS : : "memory");
        s_mfma(sa, 0);
        // a unit whose 32 keys are all past the causal diagonal of its 16 query rows (or past the sequence end)
        // is skipped; wave-uniform.  Only the last tiles of a workgroup have such units.
        auto live = [&](int kbase, int qt) -> bool {
            if (GLM_FA_ABLATE & 64) return true;
            return kbase <= qbase + fa_row0(wid, qt) + 15 + off && kbase < lk;
        };
#if GLM_FA_LOOP == 1 && !GLM_FA_STAMP
        // ---- r04 loop ----
        // interior tiles of THIS wave: every key of the tile is visible to every one of the wave's 32 rows and inside the
        // sequence (t*64 + 63 <= first row + off, t*64 + 63 < lk).  They come first; the diagonal / last tiles follow.
        const int r0w = qbase + fa_row0(wid, 0);  // the wave's first row
        int n_int = min((r0w + off + 1) >> 6, lk >> 6);
        n_int = max(0, min(n_int, ntiles));
        // running staging sources (contiguous K/V): byte offset of the tile that slots 2,3 of the current tile stage (t + 1);
        // slots 0,1 stage t + 2 = one tile stride further
        const int64_t tstride = (int64_t)FA_BK * row_bytes;
        auto uni = [](const char* ptr) -> const char* {  // provably wave-uniform for the "s" constraints below
            const unsigned long long u = (unsigned long long)ptr;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
            return (const char*)(((unsigned long long)hi << 32) | lo);
        };
        const char* ksrc = uni(reinterpret_cast<const char*>(k) + ((int64_t)k0 * nkv + kvh) * FA_ROWB + tstride);
        const char* vsrc = uni(reinterpret_cast<const char*>(v) + ((int64_t)k0 * nkv + kvh) * FA_ROWB + tstride);
        const char* ksrc2 = uni(ksrc + tstride);  // tile t + 2
        const char* vsrc2 = uni(vsrc + tstride);
        // NOCHK: the caller vouches that tile t + ahead is a full tile (no row clamping needed)
        auto stage_slot2 = [&](int BUF, int t, int ahead, int slot, bool nochk) {  // pieces of tile t + ahead into buffer BUF
            if (slot >= FA_SLOTS) return;
#pragma unroll
            for (int j = 0; j < FA_PPS; ++j) {
                const int i = slot * FA_PPS + j;
                if (PAGED) {
                    stage_piece(BUF, t + ahead, i);
                } else {
                    if (GLM_FA_ABLATE & 32) continue;
                    if (!nochk && i == 0 && (t + ahead + 1) * FA_BK > lk) stage_offsets(t + ahead);  // uniform; last tile only
                    char* kd = smem + BUF * FA_STAGE + (wid * FA_PPW + i) * 1024;
                    glds16((ahead == 2 ? ksrc2 : ksrc) + soff_k[i], kd);
                    glds16((ahead == 2 ? vsrc2 : vsrc) + soff_v[i], kd + FA_TILE);
                }
            }
        };
#define FA_RK(SLAB, BUF) FA_READ_K(SLAB)
#define FA_RV(SLAB, BUF) FA_READ_V(SLAB)
        // one tile out of buffer t & 1.  INT: STRICT interior tile — all four units live, nothing masked, and not among the
        // first or the last three tiles, so that every staging / barrier / next-tile condition of the general body is true
        // and tiles t + 1, t + 2 are full tiles: the body has no scalar predicate at all.  Same schedule as the r02 loop:
        // per unit the S MFMAs of the next unit first, then LDS reads / staging, then softmax + PV of this unit.
        auto tile = [&](auto intc, int t) {
            constexpr int TK = decltype(intc)::value;  // 0 general, 1 unmasked (all units live, staging conditions at run time), 2 strict interior
            constexpr bool INT = TK == 2;
            const int BUF = t & 1;
            const int kb0 = t * FA_BK;
            const bool more = INT ? true : t + 1 < ntiles;
            const bool more2 = INT ? true : t + 2 < ntiles;
            const bool first = (INT || GLM_FA_R05) ? false : t < 1;
            auto lv = [&](int kbase, int qt) -> bool { return TK >= 1 ? true : live(kbase, qt); };
            bf16x8_t pf;
            // ---- unit (slab 0, q0)
            if (lv(kb0, 1)) s_mfma(sb, 1);
            FA_SCHED_FENCE();
            FA_RV(0, BUF);
            FA_RK(1, BUF);
            if (!first && more) stage_slot2(BUF ^ 1, t, 1, 2, INT);
            if (lv(kb0, 0)) pf = softmax(sa, kb0, 0, intc);
            asm volatile("s_waitcnt lgkmcnt(8)" : FA_V_REGS : : "memory");  // V landed, K may be in flight
            if (lv(kb0, 0)) pv_mfma(pf, 0);
            FA_SCHED_FENCE();
            // ---- unit (slab 0, q1)
            asm volatile("s_waitcnt lgkmcnt(0)" : FA_K_REGS : : "memory");
            if (lv(kb0 + 32, 0)) s_mfma(sa, 0);  // S(slab 1, q0)
            FA_SCHED_FENCE();
            if (!first && more) stage_slot2(BUF ^ 1, t, 1, 3, INT);
            if (lv(kb0, 1)) {
                pf = softmax(sb, kb0, 1, intc);
                pv_mfma(pf, 1);
            }
            FA_SCHED_FENCE();
            // ---- unit (slab 1, q0): V of slab 1 is the last LDS read of this buffer
#if GLM_FA_VEARLY
            FA_RV(1, BUF);
            if (lv(kb0 + 32, 1)) s_mfma(sb, 1);  // S(slab 1, q1)
            FA_SCHED_FENCE();
            bf16x8_t pf_e;
            const bool lv_e = lv(kb0 + 32, 0);
            if (lv_e) pf_e = softmax(sa, kb0 + 32, 0, intc);
            asm volatile("s_waitcnt lgkmcnt(0)" : FA_V_REGS : : "memory");
#else
            if (lv(kb0 + 32, 1)) s_mfma(sb, 1);  // S(slab 1, q1)
            FA_SCHED_FENCE();
            FA_RV(1, BUF);
            asm volatile("s_waitcnt lgkmcnt(0)" : FA_V_REGS : : "memory");
#endif
            if (more) {
                // tile t+1 was issued a whole tile ago; after the barrier every wave is done with this buffer, which tile
                // t+2 may then overwrite
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!(GLM_FA_ABLATE & 2)) __builtin_amdgcn_s_barrier();
                if (more2) stage_slot2(BUF, t, 2, 0, INT);
                const int flip = BUF ? -FA_STAGE : FA_STAGE;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) k_abs[ks] += flip;
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) v_abs[dt] += flip;
                FA_RK(0, BUF ^ 1);
            }
#if GLM_FA_VEARLY
            if (lv_e) pv_mfma(pf_e, 0);
#else
            if (lv(kb0 + 32, 0)) {
                pf = softmax(sa, kb0 + 32, 0, intc);
                pv_mfma(pf, 0);
            }
#endif
            FA_SCHED_FENCE();
            // ---- unit (slab 1, q1)
            if (more) {
                asm volatile("s_waitcnt lgkmcnt(0)" : FA_K_REGS : : "memory");
                if (INT || live(kb0 + FA_BK, 0)) s_mfma(sa, 0);  // S(tile t+1 slab 0, q0): live behind an interior tile
                FA_SCHED_FENCE();
                if (more2) stage_slot2(BUF, t, 2, 1, INT);
            }
            if (lv(kb0 + 32, 1)) {
                pf = softmax(sb, kb0 + 32, 1, intc);
                pv_mfma(pf, 1);
            }
            FA_SCHED_FENCE();
            ksrc += tstride;
            vsrc += tstride;
            ksrc2 += tstride;
            vsrc2 += tstride;
            // keep the four running pointers as such (hipcc otherwise re-forms base + offset in front of every DMA)
            asm volatile("" : "+s"(ksrc), "+s"(vsrc), "+s"(ksrc2), "+s"(vsrc2));
        };
        {
            // strict interior range [1, n1): see `tile`.  GLM_FA_EDGE (study, off): the unmasked tiles on either side of it keep
            // the run-time staging conditions but none of the per-unit liveness and mask tests.  (A two-shape fit of the r04
            // timings — 1 x 16384 and 16 x 1024 — puts a workgroup's four non-interior tiles at 2.8 interior ones each, 28 % of
            // the 4 x 4096 time; the predicates are not where that goes.)
            using K0 = std::integral_constant<int, 0>;
            using K1 = std::integral_constant<int, 1>;
            using K2 = std::integral_constant<int, 2>;
#if GLM_FA_R05
            // predicate-free range [0, n1): tiles t + 1 and t + 2 exist and are full tiles (the sequence's last tile may be partial)
            const int n1 = min(n_int, ntiles - 2 - (ntiles * FA_BK > lk ? 1 : 0));
            int t = 0;
#else
            const int n1 = min(n_int, ntiles - 3);
            int t = 0;
            if (ntiles > 0) {
                if (GLM_FA_EDGE == 2 && n_int >= 1) tile(K1{}, t++);  // (a third copy of the body in front of the loop: spills)
                else tile(K0{}, t++);
            }
#endif
            for (; t < n1; ++t) tile(K2{}, t);
            if (GLM_FA_EDGE)
                for (; t < n_int; ++t) tile(K1{}, t);
            for (; t < ntiles; ++t) tile(K0{}, t);
        }
#undef FA_RK
#undef FA_RV
#else
#if GLM_FA_STAMP
        unsigned long long tacc[16] = {}, tprev = 0;
#endif
        for (int t = 0; t < ntiles; ++t) {
            const int kb0 = t * FA_BK;
            bf16x8_t pf;
            FA_T(0);
            // ---- unit (slab 0, q0): S(slab 0, q1) goes to the matrix pipe first, then K of slab 1 and V of
            // slab 0 are fetched while the softmax runs
            if (live(kb0, 1)) s_mfma(sb, 1);
            FA_SCHED_FENCE();
            FA_T(1);  // 0: S issue
            FA_READ_V(0);
            FA_READ_K(1);
            if (t >= 1 && t + 1 < ntiles) stage_slot((t + 1) & 1, t + 1, 2);
            FA_T(2);  // 1: LDS read + DMA issue
            if (live(kb0, 0)) pf = softmax(sa, kb0, 0, std::false_type{});
            FA_T(3);  // 2: softmax
            asm volatile("s_waitcnt lgkmcnt(8)" : FA_V_REGS : : "memory");  // V landed, K may be in flight
            FA_T(4);  // 3: V wait
            if (live(kb0, 0)) pv_mfma(pf, 0);
            FA_SCHED_FENCE();
            FA_T(5);  // 4: PV issue
            // ---- unit (slab 0, q1)
            asm volatile("s_waitcnt lgkmcnt(0)" : FA_K_REGS : : "memory");
            FA_T(6);  // 5: K wait
            if (live(kb0 + 32, 0)) s_mfma(sa, 0);  // S(slab 1, q0)
            FA_SCHED_FENCE();
            if (t >= 1 && t + 1 < ntiles) stage_slot((t + 1) & 1, t + 1, 3);
            FA_T(7);  // 6: S issue + DMA
            if (live(kb0, 1)) {
                pf = softmax(sb, kb0, 1, std::false_type{});
                pv_mfma(pf, 1);
            }
            FA_SCHED_FENCE();
            FA_T(8);  // 7: softmax + PV
            // ---- unit (slab 1, q0): V of slab 1 is the last LDS read of this buffer
            if (live(kb0 + 32, 1)) s_mfma(sb, 1);  // S(slab 1, q1)
            FA_SCHED_FENCE();
            FA_READ_V(1);
            FA_T(9);  // 8: S issue + V read issue
            asm volatile("s_waitcnt lgkmcnt(0)" : FA_V_REGS : : "memory");
            FA_T(10);  // 9: V wait
            const bool more = t + 1 < ntiles;
            if (more) {
                // tile t+1 was issued a whole tile ago; after the barrier every wave is done with this
                // buffer, which tile t+2 may then overwrite
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                FA_T(11);  // 10: DMA wait
                if (!(GLM_FA_ABLATE & 2)) __builtin_amdgcn_s_barrier();
                FA_T(12);  // 11: barrier
                if (t + 2 < ntiles) stage_slot(t & 1, t + 2, 0);
                const int flip = (t & 1) ? -FA_STAGE : FA_STAGE;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) k_abs[ks] += flip;
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) v_abs[dt] += flip;
                FA_READ_K(0);
                FA_T(13);  // 12: DMA issue + K read issue
            }
            if (live(kb0 + 32, 0)) {
                pf = softmax(sa, kb0 + 32, 0, std::false_type{});
                pv_mfma(pf, 0);
            }
            FA_SCHED_FENCE();
            FA_T(14);  // 13: softmax + PV
            // ---- unit (slab 1, q1)
            if (more) {
                asm volatile("s_waitcnt lgkmcnt(0)" : FA_K_REGS : : "memory");
                if (live(kb0 + FA_BK, 0)) s_mfma(sa, 0);  // S(tile t+1 slab 0, q0)
                FA_SCHED_FENCE();
                if (t + 2 < ntiles) stage_slot(t & 1, t + 2, 1);
            }
            FA_T(15);  // 14: K wait + S issue + DMA
            if (live(kb0 + 32, 1)) {
                pf = softmax(sb, kb0 + 32, 1, std::false_type{});
                pv_mfma(pf, 1);
            }
            FA_SCHED_FENCE();
            FA_T(16);  // 15: softmax + PV
        }
#if GLM_FA_STAMP
        if (blockIdx.x == 0 && tid == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) g_fa_stamps[i] = tacc[i];
        }
#endif
#endif  // GLM_FA_LOOP
    }
#endif
#undef FA_READ_K
#undef FA_READ_V
#undef FA_K_REGS
#undef FA_V_REGS
    FA_WG(2);

#pragma unroll
    for (int qt = 0; qt < FA_QT; ++qt) {
        const float l = acc_l[qt][0];  // already summed over the four lane groups by the MFMA
        if (qi[qt] < lq) {
            const float inv = l > 0.f ? 1.0f / l : 0.f;
            bf16_t* orow = out + ((int64_t)(q0 + qi[qt]) * nh + head) * FA_HD;
#if GLM_FA_STORE16
            // r05: a lane holds hd [16 dt + 4 g, +4) of its query row; lane rows g and g ^ 1 exchange one block each
            // (v_permlane16_swap: odd rows of the first operand <-> even rows of the second), after which an even row holds
            // hd [16 dt + 4 g, +8) and an odd row hd [16 (dt + 1) + 4 (g - 1), +8): 4 x 16-byte stores instead of 8 x 8
            // (the store tail of a workgroup is bound by store INSTRUCTIONS, cdna_hip_programming.md T21)
#pragma unroll
            for (int dt = 0; dt < 8; dt += 2) {
                const unsigned a0 = pack2bf(acc[qt][dt][0] * inv, acc[qt][dt][1] * inv);
                const unsigned a1 = pack2bf(acc[qt][dt][2] * inv, acc[qt][dt][3] * inv);
                const unsigned b0 = pack2bf(acc[qt][dt + 1][0] * inv, acc[qt][dt + 1][1] * inv);
                const unsigned b1 = pack2bf(acc[qt][dt + 1][2] * inv, acc[qt][dt + 1][3] * inv);
                const u32x2_t x = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                const u32x2_t y = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                const int col = 16 * (dt + (g & 1)) + 4 * (g & 2);
                *reinterpret_cast<u32x4_t*>(orow + col) = u32x4_t{x[0], y[0], x[1], y[1]};
            }
#else
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                u32x2_t o = {pack2bf(acc[qt][dt][0] * inv, acc[qt][dt][1] * inv),
                             pack2bf(acc[qt][dt][2] * inv, acc[qt][dt][3] * inv)};
                *reinterpret_cast<u32x2_t*>(orow + 16 * dt + 4 * g) = o;
            }
#endif
        }
    }
#if GLM_FA_WGSTAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FA_WG(3);
#endif
}


#if GLM_STUDY_KERNELS
// ==========================================================================================================================
// r03 (OPT-IN, GLM_FA_MFMA32=1): the same kernel on v_mfma_f32_32x32x16_bf16 (prefill_attn32_kernel).  Why it was built: with 16x16x32 MFMAs a (32 keys x 32 query
// rows) step costs 34 MFMA issues (8 of a SIMD's 16 cycles each) next to ~260 cycles of softmax VALU issue — the SIMD's
// ONE vector issue port is as busy as its matrix pipe (r02: 36 % of peak).  The 32x32x16 shape does the same FLOPs in half
// the instructions (8 issue cycles of 32), which leaves the port to the softmax: issue ~410 against 512 pipe cycles per step.
//   * a wave owns 32 query rows (column n = lane & 31 of every tile), a workgroup 4 x 32 = 128 rows as before;
//   * S^T = K·Q^T: A = K rows (ds_read_b128, lane (r, h) reads hd [16 ks + 8 h, +8) of key r: the K image and its
//     swizzle are unchanged), B = Q^T fragments kept in registers; the 32x32 result has a query per lane and 16 of the
//     slab's 32 keys in its registers (key = (reg & 3) + 8 (reg >> 2) + 4 h), the other 16 in lane ^ 32: the softmax is
//     lane-local up to ONE lane swap for the exact row maximum (slow path only);
//   * P^T is the accumulator itself: registers 8 s .. 8 s + 7, rounded to bf16 pairwise, are the B fragment of k-step s
//     of O^T += V^T·P^T with the k order permuted (element j of half h = key 16 s + 8 (j >> 2) + 4 h + (j & 3)); the V^T
//     A fragment is fetched in that same order by two ds_read_b64_tr_b16 per (k-step, 32-wide hd block) from a V image
//     swizzled chunk ^ (((key & 3) << 2) | ((key >> 2) & 3)) (conflict-free for this read: cdna_hip_programming.md T10 (b));
//   * the denominator is a lane-local fp32 sum of the slab's (unrounded) p, the two halves of a row added once at the
//     end, instead of an MFMA against a ones fragment.
// Staging, buffer handling, work order, lazy maximum, masking rules and skip of dead slabs are the r02 kernel's.
// QB = 32-row query blocks per wave.  Only QB = 1 is built (two waves per SIMD, 128 rows per workgroup).  QB = 2 — 64 rows
// per wave at ONE wave per SIMD, every K/V fragment read feeding two blocks, with its own schedule (S(0), S(1) | softmax(0)
// beside S(1) in the pipe | PV(0) | softmax(1) beside PV(0) | PV(1)) — was built and measured in r03: hipcc spreads the 512
// registers as 256 VGPR + 251 AGPR with ~1400 v_accvgpr copies in the kernel, 526 TFLOP/s and one failing parity case; that
// structure needs asm-pinned accumulators (cdna_hip_programming.md §5.7) and is not in the tree.
template <bool PAGED, int QB>
__global__ __launch_bounds__(256, (QB == 1 ? 2 : 1)) void prefill_attn32_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const int32_t* __restrict__ cu_q, const int32_t* __restrict__ cu_k,
    const int32_t* __restrict__ block_tables, int64_t bt_stride, bf16_t* __restrict__ out, int nh,
    int nkv, int block_size, float scale_log2e, int nqblk, int nseq, int xcd_groups) {
    constexpr int BQ = 128 * QB;  // query rows per workgroup
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* bt_lds = reinterpret_cast<int*>(smem + 2 * FA_STAGE);
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
    const int n32 = lane & 31, h = lane >> 5;
    const int q0 = cu_q[b], lq = cu_q[b + 1] - q0;
    const int k0 = cu_k[b], lk = cu_k[b + 1] - k0;
    const int qbase = qblk * BQ;
    if (qbase >= lq) return;  // uniform
    const int kvh = head / (nh / nkv);
    const int off = lk - lq;
    const int kend = min(lk, min(lq, qbase + BQ) + off);
    const int ntiles = (kend + FA_BK - 1) / FA_BK;
    if (PAGED) {
        const int nblk = (kend + block_size - 1) / block_size;
        for (int i = tid; i < nblk; i += 256) bt_lds[i] = block_tables[b * bt_stride + i];
    }
    // Q^T fragments: lane (n32, h), block qb, k-step ks -> Q[row][16 ks + 8 h, +8)
    int qi[QB], row_first[QB], jmax[QB];
    bf16x8_t qf[QB][8];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        row_first[qb] = qbase + 32 * QB * wid + 32 * qb;
        qi[qb] = row_first[qb] + n32;
        jmax[qb] = qi[qb] + off;
        const bool ok = qi[qb] < lq;
        const bf16_t* qrow = q + ((int64_t)(q0 + (ok ? qi[qb] : 0)) * nh + head) * FA_HD;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            u32x4_t t = {0, 0, 0, 0};
            if (ok) t = *reinterpret_cast<const u32x4_t*>(qrow + 16 * ks + 8 * h);
            qf[qb][ks] = __builtin_bit_cast(bf16x8_t, t);
        }
    }
    __syncthreads();
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
        asm volatile("" : "+v"(qf[qb][0]), "+v"(qf[qb][1]), "+v"(qf[qb][2]), "+v"(qf[qb][3]), "+v"(qf[qb][4]), "+v"(qf[qb][5]),
                     "+v"(qf[qb][6]), "+v"(qf[qb][7]));

    // ---- staging (as the r02 kernel; only the V swizzle differs) ----
    constexpr int PPW = 4;  // 1 KiB pieces per wave, for K and for V
    const int64_t row_bytes = (int64_t)nkv * FA_ROWB;
    unsigned soff_k[PPW], soff_v[PPW];
    auto vswz = [](int row) { return ((row & 3) << 2) | ((row >> 2) & 3); };
    auto stage_offsets = [&](int t) {
        const int r4 = lane >> 4, pch = lane & 15;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int row = (wid * PPW + i) * 4 + r4;
            const int rc = min(t * FA_BK + row, lk - 1) - t * FA_BK;
            soff_k[i] = (unsigned)(rc * (int)row_bytes) + ((pch ^ (row & 15)) << 4);
            soff_v[i] = (unsigned)(rc * (int)row_bytes) + ((pch ^ vswz(row)) << 4);
        }
    };
    if (!PAGED) stage_offsets(0);
    auto stage_piece = [&](int buf, int t, int i) {
        char* kd = smem + buf * FA_STAGE + (wid * PPW + i) * 1024;
        char* vd = kd + FA_TILE;
        if (!PAGED) {
            if (i == 0 && (t + 1) * FA_BK > lk) stage_offsets(t);
            const int64_t tb = ((int64_t)(k0 + t * FA_BK) * nkv + kvh) * FA_ROWB;
            glds16(reinterpret_cast<const char*>(k) + tb + soff_k[i], kd);
            glds16(reinterpret_cast<const char*>(v) + tb + soff_v[i], vd);
        } else {
            const int r4 = lane >> 4, pch = lane & 15;
            const int row = (wid * PPW + i) * 4 + r4;
            int key = t * FA_BK + row;
            key = key < lk ? key : lk - 1;
            const int pb = bt_lds[key / block_size];
            const int64_t tok = (int64_t)pb * block_size + key % block_size;
            const int64_t ro = (tok * nkv + kvh) * FA_HD;
            glds16(k + ro + ((pch ^ (row & 15)) << 3), kd);
            glds16(v + ro + ((pch ^ vswz(row)) << 3), vd);
        }
    };
    auto stage = [&](int buf, int t) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) stage_piece(buf, t, i);
    };

    // ---- LDS fragment addresses of the CURRENT buffer ----
    // K: row n32 (+32 per slab: immediate), chunk (2 ks + h) ^ (row & 15): one register per k-step
    // V: lane = 16 g16 + 4 qq + pp supplies row r0 + qq, hd columns c0 + 4 pp .. +3 with r0 = 16 s + 8 jh + 4 (g16 >> 1),
    //    c0 = 32 blk + 16 (g16 & 1): one register per (blk, jh); k-step s (+4096) and slab (+8192) are immediates
    unsigned k_abs[8], v_abs[4][2];
    {
        const unsigned base = lds_addr(smem);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) k_abs[ks] = base + n32 * FA_ROWB + (((2 * ks + h) ^ (n32 & 15)) << 4);
        const int g16 = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {
                const int row = 8 * jh + 4 * (g16 >> 1) + qq;
                const int ch = 4 * blk + 2 * (g16 & 1) + (pp >> 1);
                v_abs[blk][jh] = base + FA_TILE + row * FA_ROWB + ((ch ^ vswz(row)) << 4) + ((pp & 1) << 3);
            }
    }

    f32x16_t acc[QB][4];  // O^T: hd block blk, lane = query column, reg -> hd 32 blk + (reg & 3) + 8 (reg >> 2) + 4 h
    float l_part[QB], m_run[QB], m_thr[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[qb][i][r] = 0.f;
        l_part[qb] = 0.f;
        m_run[qb] = -INFINITY;
        m_thr[qb] = -INFINITY;
    }

    u32x4_t kf[8];
    u32x2_t vf[2][4][2];  // [k-step s][hd block][jh]
#define FB_READ_K(SLAB)                                                                      \
    _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) lds_read_b128<(SLAB) * 8192>(kf[ks], k_abs[ks]);
#define FB_READ_V(SLAB)                                                                      \
    _Pragma("unroll") for (int blk = 0; blk < 4; ++blk) {                                    \
        lds_read_tr16_b64<(SLAB) * 8192>(vf[0][blk][0], v_abs[blk][0]);                      \
        lds_read_tr16_b64<(SLAB) * 8192>(vf[0][blk][1], v_abs[blk][1]);                      \
        lds_read_tr16_b64<(SLAB) * 8192 + 4096>(vf[1][blk][0], v_abs[blk][0]);               \
        lds_read_tr16_b64<(SLAB) * 8192 + 4096>(vf[1][blk][1], v_abs[blk][1]);               \
    }
#define FB_K_REGS "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(kf[4]), "+v"(kf[5]), "+v"(kf[6]), "+v"(kf[7])
#define FB_V_REGS                                                                                                  \
    "+v"(vf[0][0][0]), "+v"(vf[0][0][1]), "+v"(vf[0][1][0]), "+v"(vf[0][1][1]), "+v"(vf[0][2][0]), "+v"(vf[0][2][1]), \
        "+v"(vf[0][3][0]), "+v"(vf[0][3][1]), "+v"(vf[1][0][0]), "+v"(vf[1][0][1]), "+v"(vf[1][1][0]),               \
        "+v"(vf[1][1][1]), "+v"(vf[1][2][0]), "+v"(vf[1][2][1]), "+v"(vf[1][3][0]), "+v"(vf[1][3][1])

    auto s_mfma = [&](f32x16_t& s, int qb) {
        const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, kf[0]), qf[qb][0], zero, 0, 0, 0);  // C = 0 inline
#pragma unroll
        for (int ks = 1; ks < 8; ++ks)
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, kf[ks]), qf[qb][ks], s, 0, 0, 0);
    };
    // online softmax of one slab (keys [kbase, kbase + 32) x query block qb): returns the two bf16 P^T fragments (k-steps 0
    // and 1) and leaves acc[qb] rescaled.  Reference point and lazy maximum exactly as the r02 kernel.
    auto softmax = [&](f32x16_t& s, int kbase, int qb, bf16x8_t (&pf)[2]) {
        const bool need_mask = kbase + 31 > row_first[qb] + off || kbase + 31 >= lk;
        if (need_mask) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = kbase + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (j > jmax[qb] || j >= lk) s[r] = -INFINITY;
            }
        }
        float mloc = fmaxf(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])), fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7])));
        mloc = fmaxf(mloc, fmaxf(fmaxf(fmaxf(s[8], s[9]), fmaxf(s[10], s[11])), fmaxf(fmaxf(s[12], s[13]), fmaxf(s[14], s[15]))));
        float ms = mloc * scale_log2e;
        float m_use = m_run[qb];
        if (need_mask || !__all(ms <= m_thr[qb])) {
            // exact row maximum: the other 16 keys of the slab live in lane ^ 32
            const u32x2_t sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ms), __float_as_uint(ms), false, false);
            ms = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            const float m_new = ms > m_run[qb] ? ms : m_run[qb];
            m_use = m_new;
            if (need_mask) m_use = m_new == -INFINITY ? 0.f : m_new;
            if (__any(m_new > m_run[qb])) {
                const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_use);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[qb][i][r] *= alpha;
                l_part[qb] *= alpha;
            }
            m_run[qb] = m_new;
            m_thr[qb] = m_new + 8.0f;
        }
        float lsum = 0.f;
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
            u32x4_t pk;
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) {
                const float p0 = __builtin_amdgcn_exp2f(fmaf(s[8 * sidx + 2 * j2], scale_log2e, -m_use));
                const float p1 = __builtin_amdgcn_exp2f(fmaf(s[8 * sidx + 2 * j2 + 1], scale_log2e, -m_use));
                lsum += p0;  // fp32 denominator (flash-attention's convention: normalise in fp32, round P for the second product)
                lsum += p1;
                unsigned w;
                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(p0), "v"(p1));  // one instruction per pair (T12)
                pk[j2] = w;
            }
            pf[sidx] = __builtin_bit_cast(bf16x8_t, pk);
        }
        l_part[qb] += lsum;
    };
    auto pv_mfma = [&](const bf16x8_t (&pf)[2], int qb) {
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                const u32x4_t vv = {vf[sidx][blk][0][0], vf[sidx][blk][0][1], vf[sidx][blk][1][0], vf[sidx][blk][1][1]};
                acc[qb][blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vv), pf[sidx], acc[qb][blk], 0, 0, 0);
            }
    };
    asm volatile("" ::"s"(scale_log2e), "s"(lk), "s"(off));

    if (ntiles > 0) {
        f32x16_t sbuf[2][QB];
        bf16x8_t pf[QB][2];
        stage(0, 0);
        if (ntiles > 1) {
            stage(1, 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        // a slab whose 32 keys are all past the causal diagonal of a 32-row query block (or past the sequence end)
        auto live = [&](int kbase, int qb) -> bool { return kbase <= row_first[qb] + 31 + off && kbase < lk; };
        FB_READ_K(0);
        asm volatile("s_waitcnt lgkmcnt(0)" : FB_K_REGS : : "memory");
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) s_mfma(sbuf[0][qb], qb);
        FA_SCHED_FENCE();
        FB_READ_K(1);
        for (int t = 0; t < ntiles; ++t) {
            const int kb0 = t * FA_BK;
            const bool more = t + 1 < ntiles;
            // ---- slab 0: S(slab 1) goes to the matrix pipe first; V of slab 0 arrives while the softmax runs
            asm volatile("s_waitcnt lgkmcnt(0)" : FB_K_REGS : : "memory");  // K of slab 1 (requested a whole step ago)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
                if (live(kb0 + 32, qb)) s_mfma(sbuf[1][qb], qb);
            FA_SCHED_FENCE();
            FB_READ_V(0);
            if (t >= 1 && more) { stage_piece((t + 1) & 1, t + 1, 2); }
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
                if (live(kb0, qb)) softmax(sbuf[0][qb], kb0, qb, pf[qb]);
            asm volatile("s_waitcnt lgkmcnt(0)" : FB_V_REGS : : "memory");
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
                if (live(kb0, qb)) pv_mfma(pf[qb], qb);
            FA_SCHED_FENCE();
            // ---- slab 1: its V is the last LDS read of this buffer
            FB_READ_V(1);
            if (t >= 1 && more) { stage_piece((t + 1) & 1, t + 1, 3); }
            if (more) {
                asm volatile("s_waitcnt lgkmcnt(0)" : FB_V_REGS : : "memory");
                // tile t+1 was issued a whole tile ago; after the barrier every wave is done with this buffer, which tile
                // t+2 may then overwrite
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (t + 2 < ntiles) { stage_piece(t & 1, t + 2, 0); }
                const int flip = (t & 1) ? -FA_STAGE : FA_STAGE;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) k_abs[ks] += flip;
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) {
                    v_abs[blk][0] += flip;
                    v_abs[blk][1] += flip;
                }
                FB_READ_K(0);
                asm volatile("s_waitcnt lgkmcnt(0)" : FB_K_REGS : : "memory");
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
                    if (live(kb0 + FA_BK, qb)) s_mfma(sbuf[0][qb], qb);
                FA_SCHED_FENCE();
                FB_READ_K(1);
                if (t + 2 < ntiles) { stage_piece(t & 1, t + 2, 1); }
            }
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
                if (live(kb0 + 32, qb)) softmax(sbuf[1][qb], kb0 + 32, qb, pf[qb]);
            if (!more) asm volatile("s_waitcnt lgkmcnt(0)" : FB_V_REGS : : "memory");
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
                if (live(kb0 + 32, qb)) pv_mfma(pf[qb], qb);
            FA_SCHED_FENCE();
        }
    }
#undef FB_READ_K
#undef FB_READ_V
#undef FB_K_REGS
#undef FB_V_REGS
    // the two halves of a query's keys: one lane swap
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const u32x2_t sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_part[qb]), __float_as_uint(l_part[qb]), false, false);
        const float l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        if (qi[qb] < lq) {
            const float inv = l > 0.f ? 1.0f / l : 0.f;
            bf16_t* orow = out + ((int64_t)(q0 + qi[qb]) * nh + head) * FA_HD;
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const u32x2_t o = {pack2bf(acc[qb][blk][4 * rg] * inv, acc[qb][blk][4 * rg + 1] * inv),
                                       pack2bf(acc[qb][blk][4 * rg + 2] * inv, acc[qb][blk][4 * rg + 3] * inv)};
                    *reinterpret_cast<u32x2_t*>(orow + 32 * blk + 8 * rg + 4 * h) = o;
                }
        }
    }
}

#endif  // GLM_STUDY_KERNELS

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
    const int paged = block_tables != nullptr;
    if (paged) GLM_SET_MAX_LDS(prefill_attn_kernel<true>, 2 * FA_STAGE + FA_MAX_BT * 4, "glm_varlen_attn_prefill");
    else GLM_SET_MAX_LDS(prefill_attn_kernel<false>, 2 * FA_STAGE + FA_MAX_BT * 4, "glm_varlen_attn_prefill");
    const int nqblk = (max_seqlen_q + FA_BQ - 1) / FA_BQ;
    const int64_t total = (int64_t)nqblk * nh * B;
    GLM_CHECK_ARG(total < (int64_t)1 << 31, GLM_ERR_UNSUPPORTED, "glm_varlen_attn_prefill: grid too large");
    // (sequence, kv head) groups per XCD when they divide evenly over the 8 XCDs, else 0 (plain longest-first order)
    const int xcd_groups = (B * nkv) % 8 == 0 ? (int)(B * nkv / 8) : 0;
    const dim3 grid((unsigned)total);
    // K/V ring + the block-table slice (paged only): two workgroups per CU need <= 80 KiB each
    const size_t lds = 2 * FA_STAGE + (paged ? (size_t)bt_stride * 4 : 0);
    const float sl2 = scale * 1.4426950408889634f;
    // option fa_mfma32 = 1 (GLM_FA_MFMA32 at start-up, glm_set_option afterwards): the 32x32x16 kernel (same grid and workgroup shape as the default 16x16x32
    // kernel).  Parity green (test_varlen_attn_prefill_mfma32); measured r03, one box, interleaved processes: 746-754 against
    // 789-805 TFLOP/s at 4 x 4096 tokens, so it stays opt-in.
#if GLM_STUDY_KERNELS
    if (glm_option(GLM_OPT_FA_ROWS64) == 1)
        return glm_launch_prefill_attn64(q, k, v, cu_seqlens_q, cu_seqlens_k, block_tables, bt_stride, out, B, max_seqlen_q, nh, nkv,
                                         block_size, scale, glm_s(stream));
    const int o32 = glm_option(GLM_OPT_FA_MFMA32);
    const bool use32 = o32 == 1 || (o32 < 0 && FA_DEFAULT_MFMA32 == 1);
    if (use32 && FA_BQ == 128) {
        if (paged) {
            GLM_SET_MAX_LDS((prefill_attn32_kernel<true, 1>), 2 * FA_STAGE + FA_MAX_BT * 4, "glm_varlen_attn_prefill");
            hipLaunchKernelGGL((prefill_attn32_kernel<true, 1>), grid, dim3(256), lds, glm_s(stream), (const bf16_t*)q,
                               (const bf16_t*)k, (const bf16_t*)v, cu_seqlens_q, cu_seqlens_k, block_tables, bt_stride,
                               (bf16_t*)out, nh, nkv, block_size, sl2, nqblk, (int)B, xcd_groups);
        } else {
            GLM_SET_MAX_LDS((prefill_attn32_kernel<false, 1>), 2 * FA_STAGE + FA_MAX_BT * 4, "glm_varlen_attn_prefill");
            hipLaunchKernelGGL((prefill_attn32_kernel<false, 1>), grid, dim3(256), lds, glm_s(stream), (const bf16_t*)q,
                               (const bf16_t*)k, (const bf16_t*)v, cu_seqlens_q, cu_seqlens_k, block_tables, bt_stride,
                               (bf16_t*)out, nh, nkv, block_size, sl2, nqblk, (int)B, xcd_groups);
        }
        GLM_CHECK_LAUNCH("glm_varlen_attn_prefill");
        return GLM_OK;
    }
    // option fa_persist = 1 (r05): persistent workgroups with a run-time block queue (prefill_attn_persist.hip; contiguous K/V).
    // Bit-identical; measured at parity (4 x 4096) to -12 % (64 x 256) against the kernel below: DESIGN.md 4.3
    if (!paged && glm_option(GLM_OPT_FA_PERSIST) == 1) {
        const int rc = glm_launch_prefill_attn_persist(q, k, v, cu_seqlens_q, cu_seqlens_k, out, B, max_seqlen_q, nh, nkv, scale,
                                                       glm_s(stream));
        if (rc != GLM_ERR_UNSUPPORTED) return rc;  // unsupported = too many sequences for its LDS tables: the kernel below takes it
    }
#else
    if (glm_option(GLM_OPT_FA_ROWS64) == 1 || glm_option(GLM_OPT_FA_MFMA32) == 1 || glm_option(GLM_OPT_FA_PERSIST) == 1) {
        glm_set_error("glm_varlen_attn_prefill: options fa_rows64 / fa_mfma32 / fa_persist select study kernels (measured slower or no faster, DESIGN.md 4.3); "
                      "this is the product library — build and load lib/libglm4moe_hip_study.so (build.py --study)");
        return GLM_ERR_UNSUPPORTED;
    }
#endif
    if (paged)
        hipLaunchKernelGGL((prefill_attn_kernel<true>), grid, dim3(FA_NT), lds, glm_s(stream), (const bf16_t*)q,
                           (const bf16_t*)k, (const bf16_t*)v, cu_seqlens_q, cu_seqlens_k, block_tables,
                           bt_stride, (bf16_t*)out, nh, nkv, block_size, sl2, nqblk, (int)B, xcd_groups);
    else
        hipLaunchKernelGGL((prefill_attn_kernel<false>), grid, dim3(FA_NT), lds, glm_s(stream), (const bf16_t*)q,
                           (const bf16_t*)k, (const bf16_t*)v, cu_seqlens_q, cu_seqlens_k, block_tables,
                           bt_stride, (bf16_t*)out, nh, nkv, block_size, sl2, nqblk, (int)B, xcd_groups);
    GLM_CHECK_LAUNCH("glm_varlen_attn_prefill");
    return GLM_OK;
}

// debug aid (not part of the C-ABI in include/glm4moe_hip.h): resident workgroups per CU the runtime
// predicts for the contiguous-K/V kernel with `lds_bytes` of dynamic LDS
extern "C" int glm_debug_prefill_attn_occupancy(int lds_bytes) {
    int nb = -1;
    (void)hipFuncSetAttribute((const void*)prefill_attn_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              2 * FA_STAGE + FA_MAX_BT * 4);
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, prefill_attn_kernel<false>, FA_NT,
                                                                (size_t)lds_bytes);
    return e == hipSuccess ? nb : -(int)e;
}

#if GLM_FA_WGSTAMP
extern "C" int glm_debug_prefill_attn_wgstamps(unsigned long long* out /* [16384][4] */) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fa_wgstamps), sizeof(unsigned long long) * 16384 * 4) == hipSuccess ? 0 : -1;
}
#endif
#if GLM_FA_STAMP
// timing study only: per-segment cycle sums of wave 0 of workgroup 0 (the longest one), [16]
extern "C" int glm_debug_prefill_attn_stamps(unsigned long long* out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_fa_stamps), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif
