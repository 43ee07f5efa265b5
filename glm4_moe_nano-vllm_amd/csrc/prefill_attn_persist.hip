// prefill_attn_persist.hip — r05: the varlen causal prefill attention of prefill_attn.hip (a9, reference:
// nanovllm/layers/attention.py:97-100 → flash_attn_varlen_func(causal=True)) as PERSISTENT workgroups.  gfx950 only,
// head_dim 128, contiguous K/V (the paged-prefix form stays on prefill_attn_kernel<true>).
//
// Why: r04 measured that everything of a workgroup that is not an interior tile — dispatch, the cu_seqlens → query rows →
// first K/V tiles chain of dependent loads, the output rows — costs ~9 us of wall time per 128-row block with two
// workgroups sharing a CU (13-28 % of the 4 x 4096 launch by two different fits), and that running two blocks back to back
// in one workgroup does not help as long as the second block's start is as uncovered as the first's.  Here a launch is
// 2 workgroups per CU that stay, and the K/V tiles of ALL of a workgroup's blocks form ONE stream through the two LDS
// buffers: tile τ lives in buffer τ & 1 whatever block it belongs to, the staging look-ahead (pieces of tile τ + 1 and
// τ + 2 are issued during tile τ, exactly the r04 schedule) simply crosses the block boundary, so the next block's first
// two tiles are in LDS before the current block's last tile is done; the next block's descriptor (two cu_seqlens pairs)
// is fetched a whole block ahead; its query rows are requested into the registers of the current block's as soon as
// the last S MFMA that reads them has been issued, and wait behind the last two softmax + P·V units and the output stores.
//
// Per (query row, key) arithmetic, tile walk and rounding points are prefill_attn_kernel<false>'s: results are bit-identical
// (tests/test_hip_kernels.py::test_varlen_attn_prefill_persist_bit_identical).
//
// Work order: the r04 order (longest block first, an XCD keeps its (sequence, kv head) groups), handed out at RUN TIME: a
// workgroup's first block is its slot's, every further one comes from a counter per XCD group (one atomic per block,
// issued two blocks ahead by wave 0 and published through LDS behind two of the tile barriers).  A first form with static
// per-slot lists (perfectly balanced in tiles) lost 4-9 %: the two workgroups of a CU do NOT share it evenly — the one
// that arrived first finished 20-25 % earlier in every CU (profiles/r05/prefill_attn_persist_static_stamps.txt) and its
// partner then ran the rest alone at one wave per SIMD.  With one-block workgroups the hardware dispatcher hides that; a
// persistent form has to do the dispatcher's job itself.
// Rows of a 128-row block are dealt to the waves as two 16-row tiles 64 rows apart (wave w: rows 16 w .. and 64 + 16 w ..),
// so that the two half-dead tiles on the causal diagonal cost every wave the same (r04: waves 0, 1 idle through the last tile).
#include "common.h"
#include "gemm_internal.h"
#include <type_traits>

#ifndef GLM_FAP_NOSNAKE
#define GLM_FAP_NOSNAKE 0   // 1 (study): plain round-robin rounds
#endif
#ifndef GLM_FAP_STAMP
#define GLM_FAP_STAMP 0
#endif

#if GLM_FAP_STAMP
// timing study only: per workgroup (wave 0) [0] first / [1] last s_memrealtime (100 MHz), [2] blocks, and s_memtime cycle sums:
// [3] wait for the query rows at a block's start, [4] general tiles, [5] interior tiles, [6] output rows + block switch,
// [7] start-up (table, first query rows, first two tiles)
__device__ unsigned long long g_fap_stamps[1024][8];
#define FAP_T(slot) do { const unsigned long long t_ = __builtin_readcyclecounter(); if ((slot) >= 0) tacc[(slot)] += t_ - tprev; tprev = t_; } while (0)
#else
#define FAP_T(slot) do {} while (0)
#endif

namespace {

constexpr int P_HD = 128, P_ROWB = 256, P_NT = 256, P_BQ = 128, P_BK = 64;
constexpr int P_TILE = P_BK * P_ROWB;   // 16 KiB per K or V tile
constexpr int P_STAGE = 2 * P_TILE;

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
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(const lds_void_t*)p; }

#define FAP_FENCE() __builtin_amdgcn_sched_barrier(0)

struct FapItem {  // one 128-row query block of one (sequence, head); wave-uniform
    int valid, q0, lq, k0, lk, head, kvh, qbase;
};
struct FapCur {   // a stream tile to stage: source pointers of its first row; lim = (sequence's last key) - (tile's first key):
    const char* kp;  // < 0: no such tile (nothing is staged), < 63: the sequence's last, partial tile (rows are clamped)
    const char* vp;
    int lim;
};

__global__ __launch_bounds__(P_NT, 2) void prefill_attn_persist_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const int32_t* __restrict__ cu_q, const int32_t* __restrict__ cu_k, bf16_t* __restrict__ out, int nh, int nkv,
    float scale_log2e, int nqblk, int nseq, int xcd_groups, int total, unsigned* __restrict__ ctr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int hpg = nh / nkv;
#if GLM_FAP_STAMP
    unsigned long long tacc[8] = {}, tprev = 0;
    const unsigned long long t_first = __builtin_amdgcn_s_memrealtime();
    FAP_T(-1);
#endif

    // ---- blocks: descriptors live in a 4-entry ring in LDS (8 ints: q0, lq, k0, lk, head, kvh, qbase, valid); block n of this
    // workgroup is entry n & 3.  Wave 0 produces them — index from the XCD group's counter, decoded with vector
    // arithmetic only (the constants come from LDS, cu_seqlens too): nothing of this occupies a scalar register across
    // the tile loops (a first form that decoded on the scalar unit spilled 66 of them).
    int* tab = reinterpret_cast<int*>(smem + 2 * P_STAGE);  // [4][8]
    int* cst = tab + 32;                                     // [16]
    int* cuq = cst + 16;                                     // [nseq + 1]
    int* cuk = cuq + nseq + 1;                               // [nseq + 1]
    const int ng = xcd_groups > 0 ? 8 : 1;
    {
        for (int i = tid; i <= nseq; i += P_NT) {
            cuq[i] = cu_q[i];
            cuk[i] = cu_k[i];
        }
        if (tid == 0) {
            cst[0] = nqblk; cst[1] = nh * nseq; cst[2] = xcd_groups; cst[3] = hpg; cst[4] = nkv; cst[5] = nh;
            cst[6] = total / ng;                  // blocks of one XCD group
            cst[7] = (int)gridDim.x / ng;         // slots of one XCD group = the statically assigned first blocks
            cst[8] = (int)blockIdx.x % ng;
        }
        __syncthreads();
    }
    unsigned* myctr = ctr + (int)blockIdx.x % ng;
    // list position jl (a per-lane value as far as the compiler knows: vector arithmetic) → ring entry `slot`
    auto decode_to = [&](int jl, int slot) {
        const int c_nqblk = cst[0], c_perq1 = cst[1], c_xg = cst[2], c_hpg = cst[3], c_nkv = cst[4], c_nh = cst[5], c_nper = cst[6],
                  c_xs = cst[8];
        int e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (jl < c_nper) {
            int qblk, head, b;
            if (c_xg > 0) {
                const int per_q = c_xg * c_hpg;  // blocks per query-block index on one XCD
                const int sidx = jl / per_q, r = jl - sidx * per_q;
                const int group = (r / c_hpg) * 8 + c_xs;  // (sequence, kv head) pair
                qblk = c_nqblk - 1 - sidx;
                b = group / c_nkv;
                head = (group % c_nkv) * c_hpg + r % c_hpg;
            } else {
                const int sidx = jl / c_perq1, r = jl - sidx * c_perq1;
                qblk = c_nqblk - 1 - sidx;
                b = r / c_nh;
                head = r % c_nh;
            }
            const int q0 = cuq[b], lq = cuq[b + 1] - q0;
            const int k0 = cuk[b], lk = cuk[b + 1] - k0;
            e[0] = q0; e[1] = lq; e[2] = k0; e[3] = lk; e[4] = head; e[5] = head / c_hpg; e[6] = qblk * P_BQ;
            e[7] = qblk * P_BQ < lq ? 1 : 2;  // 2: a block past this sequence's last query row (skipped)
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) tab[slot * 8 + i] = e[i];
        }
    };
    // draw list positions until one holds a block with rows (or the list ends) and publish it as entry `slot`; wave 0 only.
    // `first` = a position already drawn (the caller's atomic, in flight across a tile)
    auto fetch_to = [&](unsigned first, int slot) {
        unsigned f = first;
        for (;;) {
            const int jl = cst[7] + (int)f;
            decode_to(jl, slot);
            const int st = __builtin_amdgcn_readfirstlane(tab[slot * 8 + 7]);
            if (st != 2) break;  // a block with rows, or the end of the list (0)
            f = __hip_atomic_fetch_add(myctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ragged batches only
        }
    };
    auto draw = [&]() -> unsigned {  // lane 0's value counts; one atomic per wave
        unsigned f = 0;
        if (lane == 0) f = __hip_atomic_fetch_add(myctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return __shfl(f, 0, 64);
    };
    auto read_entry = [&](int slot) -> FapItem {
        FapItem it;
        const int w = tab[slot * 8 + (lane & 7)];
        it.q0 = __builtin_amdgcn_readlane(w, 0);
        it.lq = __builtin_amdgcn_readlane(w, 1);
        it.k0 = __builtin_amdgcn_readlane(w, 2);
        it.lk = __builtin_amdgcn_readlane(w, 3);
        it.head = __builtin_amdgcn_readlane(w, 4);
        it.kvh = __builtin_amdgcn_readlane(w, 5);
        it.qbase = __builtin_amdgcn_readlane(w, 6);
        it.valid = __builtin_amdgcn_readlane(w, 7) == 1;
        return it;
    };
    auto item_ntiles = [&](const FapItem& it) -> int {
        const int off = it.lk - it.lq;
        const int kend = min(it.lk, min(it.lq, it.qbase + P_BQ) + off);
        return kend > 0 ? (kend + P_BK - 1) / P_BK : 0;
    };
    // this workgroup's first block: its slot's list position, or — a block without rows — the first drawn one that has rows
    if (wid == 0) {
        decode_to((int)blockIdx.x / ng + (lane & 0), 0);
        if (__builtin_amdgcn_readfirstlane(tab[7]) == 2) fetch_to(draw(), 0);
        fetch_to(draw(), 1);
    }
    __syncthreads();
    // the last workgroup to leave puts the counters back to zero for the next launch on this counter set
    auto leave = [&]() {
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(ctr + 8, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (old == gridDim.x - 1) {
#pragma unroll
                for (int i = 0; i < 9; ++i) __hip_atomic_store(ctr + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    int nblk = 0;  // index of the block being computed (ring entry nblk & 3)
    FapItem it = read_entry(0);
    FapItem nx = read_entry(1);
    unsigned pend = 0;  // wave 0: the list position drawn for block nblk + 2
    if (!it.valid) {  // uniform: the list was exhausted before this workgroup got a block
        leave();
        return;
    }

    // ---- per-lane constants -----------------------------------------------------------------------------------------
    const int64_t row_bytes = (int64_t)nkv * P_ROWB;
    const int64_t tstride = (int64_t)P_BK * row_bytes;
    // staging: wave w issues 4 pieces for K and for V per tile; piece i covers tile rows (4 w + i) * 4 .. + 3 (4 keys x 256 B);
    // lane → row + (lane >> 4), 16-byte position lane & 15 (source-side swizzles as prefill_attn.hip)
    unsigned soff_k[4], soff_v[4];
    {
        const int r4 = lane >> 4, pch = lane & 15;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wid * 4 + i) * 4 + r4;
            soff_k[i] = (unsigned)(row * (int)row_bytes) + ((pch ^ (row & 15)) << 4);
            soff_v[i] = (unsigned)(row * (int)row_bytes) + ((pch ^ ((row & 7) << 1)) << 4);
        }
    }
    auto stage_piece = [&](int buf, const FapCur& cu, int i) {
        char* kd = smem + buf * P_STAGE + (wid * 4 + i) * 1024;
        unsigned ok = soff_k[i], ov = soff_v[i];
        if (cu.lim < P_BK - 1) {  // uniform; a sequence's last, partial tile: masked keys re-read a valid row
            const int r4 = lane >> 4, pch = lane & 15;
            const int row = (wid * 4 + i) * 4 + r4;
            const int rc = min(row, cu.lim);
            ok = (unsigned)(rc * (int)row_bytes) + ((pch ^ (row & 15)) << 4);
            ov = (unsigned)(rc * (int)row_bytes) + ((pch ^ ((row & 7) << 1)) << 4);
        }
        glds16(cu.kp + ok, kd);
        glds16(cu.vp + ov, kd + P_TILE);
    };
    // LDS fragment addresses of the CURRENT buffer: one register per fragment; they flip once per stream tile
    unsigned k_abs[4], v_abs[8];
    {
        const unsigned base = lds_addr(smem);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) k_abs[ks] = base + c * P_ROWB + (((4 * ks + g) ^ c) << 4);
        const int qq = (lane >> 2) & 3, pp = lane & 3;
        const int row = 4 * g + qq;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt)
            v_abs[dt] = base + P_TILE + row * P_ROWB + (((2 * dt + (pp >> 1)) ^ ((row & 7) << 1)) << 4) + ((pp & 1) << 3);
    }
    u32x4_t ones_u = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    asm volatile("" : "+v"(ones_u));
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_u);

    // ---- state of the block being computed --------------------------------------------------------------------------
    int q0, lq, k0, lk, head, qbase, off, kvh, kend, ntiles, nst;
    auto derive = [&]() {
        q0 = it.q0; lq = it.lq; k0 = it.k0; lk = it.lk; head = it.head; qbase = it.qbase;
        off = lk - lq;
        kvh = it.kvh;
        kend = min(lk, min(lq, qbase + P_BQ) + off);
        ntiles = kend > 0 ? (kend + P_BK - 1) / P_BK : 0;
        nst = max(ntiles, 2);  // stream tiles: at least two, so that the staging look-ahead (two tiles) never leaves the NEXT block
    };
    derive();

    bf16x8_t qf[2][4];
    int qi[2], jmax[2];
    f32x4_t acc[2][8], acc_l[2];
    float m_run[2], m_thr[2];
    u32x4_t kf[2][4];
    u32x2_t vf[8][2];
    f32x4_t sa[2], sb[2];

    // Q^T fragments of block `b`: lane (c, g), tile qt = query row qbase + 16 wid + 64 qt + c, k-step ks → hd [32 ks + 8 g, +8)
    auto load_q = [&](const FapItem& b) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int row = b.qbase + 16 * wid + 64 * qt + c;
            const bool ok = row < b.lq;
            const bf16_t* qrow = q + ((int64_t)(b.q0 + (ok ? row : 0)) * nh + b.head) * P_HD;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                u32x4_t t = {0, 0, 0, 0};
                if (ok) t = *reinterpret_cast<const u32x4_t*>(qrow + 32 * ks + 8 * g);
                qf[qt][ks] = __builtin_bit_cast(bf16x8_t, t);
            }
        }
    };

#define FAP_READ_K(SLAB)                                                    \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                      \
        lds_read_b128<(SLAB) * 8192>(kf[0][ks], k_abs[ks]);                 \
        lds_read_b128<(SLAB) * 8192 + 4096>(kf[1][ks], k_abs[ks]);          \
    }
#define FAP_READ_V(SLAB)                                                    \
    _Pragma("unroll") for (int dt = 0; dt < 8; ++dt) {                      \
        lds_read_tr16_b64<(SLAB) * 8192>(vf[dt][0], v_abs[dt]);             \
        lds_read_tr16_b64<(SLAB) * 8192 + 4096>(vf[dt][1], v_abs[dt]);      \
    }
#define FAP_K_REGS                                                                                         \
    "+v"(kf[0][0]), "+v"(kf[0][1]), "+v"(kf[0][2]), "+v"(kf[0][3]), "+v"(kf[1][0]), "+v"(kf[1][1]),       \
        "+v"(kf[1][2]), "+v"(kf[1][3])
#define FAP_V_REGS                                                                                         \
    "+v"(vf[0][0]), "+v"(vf[0][1]), "+v"(vf[1][0]), "+v"(vf[1][1]), "+v"(vf[2][0]), "+v"(vf[2][1]),       \
        "+v"(vf[3][0]), "+v"(vf[3][1]), "+v"(vf[4][0]), "+v"(vf[4][1]), "+v"(vf[5][0]), "+v"(vf[5][1]),   \
        "+v"(vf[6][0]), "+v"(vf[6][1]), "+v"(vf[7][0]), "+v"(vf[7][1])

    auto s_mfma = [&](f32x4_t (&s)[2], int qt) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            s[tt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                s[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[tt][ks]), qf[qt][ks], s[tt], 0, 0, 0);
        }
    };
    // online softmax of one unit (keys [kbase, kbase + 32) x query tile qt): prefill_attn.hip's, statement for statement
    // (lazy maximum: the reference point only has to keep exp2 in range; exact row maximum on the slow path)
    auto softmax = [&](f32x4_t (&s)[2], int kbase, int qt, auto tkc) -> bf16x8_t {
        constexpr int TK = (int)decltype(tkc)::value;  // 0 general tile, 2 strict interior tile (unmasked, not the block's first)
        const int row_first = qbase + 16 * wid + 64 * qt;
        const bool need_mask = TK >= 1 ? false : (kbase + 31 > row_first + off || kbase + 31 >= lk);
        if (need_mask) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = kbase + 4 * g + 16 * tt + r;
                    if (j > jmax[qt] || j >= lk) s[tt][r] = -INFINITY;
                }
        }
        auto maxchain = [&]() -> float {
            float mloc;
            asm("v_max3_f32 %0, %1, %2, %3\n\t"
                "v_max3_f32 %0, %0, %4, %5\n\t"
                "v_max3_f32 %0, %0, %6, %7\n\t"
                "v_max_f32 %0, %0, %8\n\t"
                "v_mul_f32 %0, %9, %0"
                : "=&v"(mloc)
                : "v"(s[0][0]), "v"(s[0][1]), "v"(s[0][2]), "v"(s[0][3]), "v"(s[1][0]), "v"(s[1][1]), "v"(s[1][2]),
                  "v"(s[1][3]), "s"(scale_log2e));
            return mloc;
        };
        float ms = 0.f;
        float m_use = m_run[qt];
        const bool first_unit = TK == 2 ? false : kbase == 0;
        bool slow0;
        if (need_mask || first_unit) {
            ms = maxchain();
            slow0 = true;
        } else {
            ms = maxchain();
            slow0 = !__all(ms <= m_thr[qt]);
        }
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
            const float m_new = ms > m_run[qt] ? ms : m_run[qt];
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
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float arg = fmaf(s[tt][r], scale_log2e, -m_use);
                pf[4 * tt + r] = (__bf16)__builtin_amdgcn_exp2f(arg);
            }
        return pf;
    };
    auto pv_mfma = [&](bf16x8_t pf, int qt) {
        acc_l[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf, acc_l[qt], 0, 0, 0);
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            const u32x4_t vv = {vf[dt][0][0], vf[dt][0][1], vf[dt][1][0], vf[dt][1][1]};
            acc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vv), pf, acc[qt][dt], 0, 0, 0);
        }
    };
    // a unit whose 32 keys are all past the block's last visible key or past the causal diagonal of its 16 query rows is
    // skipped (wave-uniform).  kbase < kend also covers the padding tiles of a block with fewer than two real tiles.
    auto live = [&](int kbase, int qt) -> bool { return kbase < kend && kbase <= qbase + 16 * wid + 64 * qt + 15 + off; };

    // stream tile `ts` counted from the current block's first tile: ts < nst → this block, else the next one
    auto cursor = [&](int ts) -> FapCur {
        FapCur cu;
        const bool here = ts < nst;
        const int tt = here ? ts : ts - nst;
        const int ck0 = here ? k0 : nx.k0, clk = here ? lk : nx.lk;
        const int ckvh = here ? kvh : nx.kvh;
        const int cnt = here ? ntiles : item_ntiles(nx);
        cu.lim = ((here || nx.valid) && tt < cnt) ? clk - 1 - tt * P_BK : -1;
        const int64_t ob = ((int64_t)ck0 * nkv + ckvh) * P_ROWB + (int64_t)tt * tstride;
        cu.kp = reinterpret_cast<const char*>(k) + ob;
        cu.vp = reinterpret_cast<const char*>(v) + ob;
        return cu;
    };

    asm volatile("" ::"s"(scale_log2e));

    // ---- one stream tile out of buffer BUF.  Unit schedule of the r04 loop: per unit the S MFMAs of the NEXT unit first, then
    // LDS reads / staging, then softmax + P·V of this unit.  TK = 2: strict interior tile — every unit live, nothing masked,
    // tiles t + 1 and t + 2 are full tiles of the same block: no predicate at all.  TK = 0: everything decided at run time:
    // A / B = the stream tiles t + 1 / t + 2 (pieces 2, 3 of A and 0, 1 of B are issued here), `more` = the stream goes on
    // behind this tile, `boundary` = this is the block's last stream tile and another block follows: the S look-ahead into
    // the next tile is left to the next block's start and the next block's query rows are requested instead.
    auto tile = [&](auto tkc, int t, int BUF, const FapCur& A, const FapCur& B, bool more, bool skip23, bool boundary) {
        constexpr int TK = decltype(tkc)::value;
        constexpr bool INT = TK == 2;
        const int kb0 = t * P_BK;
        auto lv = [&](int kbase, int qt) -> bool { return INT ? true : live(kbase, qt); };
        const bool stA = INT ? true : (A.lim >= 0 && !skip23);
        const bool stB = INT ? true : B.lim >= 0;
        bf16x8_t pf;
        // ---- unit (slab 0, q0)
        if (lv(kb0, 1)) s_mfma(sb, 1);
        FAP_FENCE();
        FAP_READ_V(0);
        FAP_READ_K(1);
        if (stA) stage_piece(BUF ^ 1, A, 2);
        if (lv(kb0, 0)) pf = softmax(sa, kb0, 0, tkc);
        asm volatile("s_waitcnt lgkmcnt(8)" : FAP_V_REGS : : "memory");  // V landed, K may be in flight
        if (lv(kb0, 0)) pv_mfma(pf, 0);
        FAP_FENCE();
        // ---- unit (slab 0, q1)
        asm volatile("s_waitcnt lgkmcnt(0)" : FAP_K_REGS : : "memory");
        if (lv(kb0 + 32, 0)) s_mfma(sa, 0);  // S(slab 1, q0)
        FAP_FENCE();
        if (stA) stage_piece(BUF ^ 1, A, 3);
        if (lv(kb0, 1)) {
            pf = softmax(sb, kb0, 1, tkc);
            pv_mfma(pf, 1);
        }
        FAP_FENCE();
        // ---- unit (slab 1, q0): V of slab 1 is the last LDS read of this buffer
        FAP_READ_V(1);
        if (lv(kb0 + 32, 1)) s_mfma(sb, 1);  // S(slab 1, q1): the block's last use of its query rows on a boundary tile
        FAP_FENCE();
        bf16x8_t pf_e;
        const bool lv_e = lv(kb0 + 32, 0);
        if (lv_e) pf_e = softmax(sa, kb0 + 32, 0, tkc);
        asm volatile("s_waitcnt lgkmcnt(0)" : FAP_V_REGS : : "memory");
        if (INT || more) {
            // stream tile t + 1 was issued a whole tile ago; behind the barrier every wave is done with this buffer, which
            // stream tile t + 2 may then overwrite
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (stB) stage_piece(BUF, B, 0);
            const int flip = BUF ? -P_STAGE : P_STAGE;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) k_abs[ks] += flip;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) v_abs[dt] += flip;
            FAP_READ_K(0);
        }
        if (lv_e) pv_mfma(pf_e, 0);
        FAP_FENCE();
        // ---- unit (slab 1, q1)
        if (INT || more) {
            if (INT || !boundary) {
                asm volatile("s_waitcnt lgkmcnt(0)" : FAP_K_REGS : : "memory");
                if (INT || live(kb0 + P_BK, 0)) s_mfma(sa, 0);  // S(tile t + 1 slab 0, q0)
                FAP_FENCE();
            }
            if (stB) stage_piece(BUF, B, 1);
        }
        if (!INT && boundary) load_q(nx);  // in flight behind the last unit, the output stores and the next block's set-up
        if (lv(kb0 + 32, 1)) {
            pf = softmax(sb, kb0 + 32, 1, tkc);
            pv_mfma(pf, 1);
        }
        FAP_FENCE();
    };
    using K0 = std::integral_constant<int, 0>;
    using K2 = std::integral_constant<int, 2>;

    // ---- the first block's start: query rows, its first two tiles -----------------------------------------------------------
    load_q(it);
    {
        const FapCur c0 = cursor(0), c1 = cursor(1);
        if (c0.lim >= 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) stage_piece(0, c0, i);
        }
        if (c1.lim >= 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) stage_piece(1, c1, i);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // the query rows and tile 0
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        FAP_READ_K(0);
    }
    int par = 0;          // buffer of the current block's tile 0
    bool first_item = true;
    FAP_T(7);
    for (;;) {
        // ---- block start: K of tile 0 / slab 0 has been requested (start-up above, or the previous block's last tile)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[qt][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            acc_l[qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            m_run[qt] = -INFINITY;
            m_thr[qt] = -INFINITY;
            qi[qt] = qbase + 16 * wid + 64 * qt + c;
            jmax[qt] = qi[qt] + off;
        }
        asm volatile("" : "+v"(qf[0][0]), "+v"(qf[0][1]), "+v"(qf[0][2]), "+v"(qf[0][3]), "+v"(qf[1][0]), "+v"(qf[1][1]),
                     "+v"(qf[1][2]), "+v"(qf[1][3]));
#if GLM_FAP_STAMP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FAP_T(3);
        tacc[2] += 1;
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" : FAP_K_REGS : : "memory");
        if (live(0, 0)) s_mfma(sa, 0);
        FAP_FENCE();

        // wave 0 draws the list position of the block after next now; it is decoded behind at least one tile (the draw has then
        // returned: every tile waits for vmcnt(0) in front of its barrier) and read by all waves at the end of this block,
        // behind at least one more barrier
        if (wid == 0 && nx.valid) pend = draw();

        // interior tiles of THIS wave: every key of the tile is visible to every one of the wave's 32 rows and inside the sequence
        const int r0w = qbase + 16 * wid;  // the wave's first row (its second tile starts 64 rows further)
        int n_int = min((r0w + off + 1) >> 6, lk >> 6);
        n_int = max(0, min(n_int, ntiles));
        // predicate-free range [first block: 1, else 0 .. n1): tiles t + 1 and t + 2 belong to this block and are full tiles.  (A
        // block's first tile needs no special case: its rows have no reference point yet, so the lazy-maximum vote fails by
        // itself and the first unit takes the exact path.)
        const int n1 = min(n_int, ntiles - 2 - (ntiles * P_BK > lk ? 1 : 0));
        int t = 0;
        auto general = [&](int tt) {
            const FapCur A = cursor(tt + 1), B = cursor(tt + 2);
            const bool last = tt + 1 >= nst;
            tile(K0{}, tt, (par + tt) & 1, A, B, !last || nx.valid, first_item && tt == 0, last && nx.valid);
        };
        if (first_item) general(t++);
        FAP_T(4);
        if (t < n1) {
            FapCur A = cursor(t + 1), B = cursor(t + 2);
            A.lim = B.lim = P_BK - 1;  // full tiles of this block (known here, not to the compiler)
            for (; t < n1; ++t) {
                tile(K2{}, t, (par + t) & 1, A, B, true, false, false);
                A.kp += tstride;
                A.vp += tstride;
                B.kp += tstride;
                B.vp += tstride;
                asm volatile("" : "+s"(A.kp), "+s"(A.vp), "+s"(B.kp), "+s"(B.vp));
            }
        }
        FAP_T(5);
        if (t == 0) general(t++);  // a short block: still one tile between the draw and its use
        if (wid == 0 && nx.valid) fetch_to(pend, (nblk + 2) & 3);  // at least one tile (one barrier) follows: t <= nst - 1
        for (; t < nst; ++t) general(t);
        FAP_T(4);

        // ---- output rows: O / l, bf16; 16-byte stores behind a lane-row exchange (see prefill_attn.hip)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const float l = acc_l[qt][0];
            if (qi[qt] < lq) {
                const float inv = l > 0.f ? 1.0f / l : 0.f;
                bf16_t* orow = out + ((int64_t)(q0 + qi[qt]) * nh + head) * P_HD;
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
            }
        }
        if (!nx.valid) break;
        par = (par + nst) & 1;
        it = nx;
        nx = read_entry((nblk + 2) & 3);
        ++nblk;
        derive();
        first_item = false;
        FAP_T(6);
    }
#if GLM_FAP_STAMP
    FAP_T(6);
    if (tid == 0 && blockIdx.x < 1024) {
        unsigned hw_id, xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        tacc[7] = ((unsigned long long)xcc_id << 32) | hw_id;  // (start-up cycles dropped for the placement word)
        g_fap_stamps[blockIdx.x][0] = t_first;
        g_fap_stamps[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int i = 2; i < 8; ++i) g_fap_stamps[blockIdx.x][i] = tacc[i];
    }
#endif
    leave();
#undef FAP_READ_K
#undef FAP_READ_V
#undef FAP_K_REGS
#undef FAP_V_REGS
}

// counters of the run-time block lists: [set][0..7] next list position per XCD group, [8] workgroups that have left.  A
// launch takes the next set (round-robin), so launches on different streams do not share one; the last workgroup of a
// launch zeroes its set.  (Two launches 64 apart in issue order that are STILL both in flight would share a set — not a
// situation a prefill step can produce.)
constexpr int P_CTR_SETS = 64;
__device__ unsigned g_fap_ctr[P_CTR_SETS][16];

}  // namespace

// arguments validated by glm_varlen_attn_prefill (prefill_attn.hip); contiguous K/V only.  GLM_ERR_UNSUPPORTED (nothing
// enqueued, no error text) for more sequences than the LDS copy of cu_seqlens holds: the caller then takes the
// one-block-per-workgroup kernel.
constexpr int P_MAX_SEQS = 1024;
int glm_launch_prefill_attn_persist(const void* q, const void* k, const void* v, const int32_t* cu_seqlens_q,
                                    const int32_t* cu_seqlens_k, void* out, int64_t B, int max_seqlen_q, int nh, int nkv,
                                    float scale, hipStream_t st) {
    static int n_cu[GLM_MAX_DEVICES] = {};
    static unsigned* ctr_base[GLM_MAX_DEVICES] = {};
    static unsigned launch_seq = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= GLM_MAX_DEVICES) dev = 0;
    if (n_cu[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        void* sym = nullptr;
        if (hipGetSymbolAddress(&sym, HIP_SYMBOL(g_fap_ctr)) != hipSuccess || sym == nullptr) {
            glm_set_error("glm_varlen_attn_prefill(persistent): no address for the block counters");
            return GLM_ERR_LAUNCH;
        }
        ctr_base[dev] = (unsigned*)sym;
        n_cu[dev] = n;
    }
    if (B > P_MAX_SEQS) return GLM_ERR_UNSUPPORTED;
    const int nqblk = (max_seqlen_q + P_BQ - 1) / P_BQ;
    const int64_t total = (int64_t)nqblk * nh * B;
    if (total >= (int64_t)1 << 31) return GLM_ERR_UNSUPPORTED;
    const int xcd_groups = (B * nkv) % 8 == 0 ? (int)(B * nkv / 8) : 0;
    const int ng = xcd_groups > 0 ? 8 : 1;
    int64_t slots = 2 * (int64_t)n_cu[dev];  // two workgroups (64 KiB of LDS + descriptors, 256 registers) per CU
    slots = slots / ng * ng;
    const int64_t G = total < slots ? total : slots;  // total is a multiple of 8 whenever xcd_groups > 0
    const size_t lds = 2 * P_STAGE + (32 + 16 + 2 * ((size_t)B + 1)) * sizeof(int);
    GLM_SET_MAX_LDS(prefill_attn_persist_kernel, 2 * P_STAGE + (32 + 16 + 2 * (P_MAX_SEQS + 1)) * sizeof(int),
                    "glm_varlen_attn_prefill(persistent)");
    unsigned* ctr = ctr_base[dev] + 16 * (__atomic_fetch_add(&launch_seq, 1u, __ATOMIC_RELAXED) % P_CTR_SETS);
    const float sl2 = scale * 1.4426950408889634f;
    hipLaunchKernelGGL(prefill_attn_persist_kernel, dim3((unsigned)G), dim3(P_NT), lds, st, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, cu_seqlens_q, cu_seqlens_k, (bf16_t*)out, nh, nkv, sl2, nqblk, (int)B, xcd_groups,
                       (int)total, ctr);
    GLM_CHECK_LAUNCH("glm_varlen_attn_prefill(persistent)");
    return GLM_OK;
}

#if GLM_FAP_STAMP
extern "C" int glm_debug_fap_stamps(unsigned long long* out /* [1024][8] */) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fap_stamps), sizeof(unsigned long long) * 1024 * 8) == hipSuccess ? 0 : -1;
}
#endif
