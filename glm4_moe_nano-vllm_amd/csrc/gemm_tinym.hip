// gemm_tinym.hip — weight-streaming GEMV for ONE token row per weight matrix: the greedy
// single-sequence decode of BASELINE config 2 (bs = 1).  Same operators as gemm_smallm.hip (a5/a10 dense projections,
// a12/a13 grouped expert gate|up + SiLU·mul and down), same rounding points, gfx950 only.
//
// Why a second body: with one token row the MFMA tile of gemm_smallm is 1/16 used and, worse, its workgroups are short
// (a wave streams 16-32 weight rows of one 128-wide K chunk at a time through a 2-stage ring): at bs = 1 its launches
// ran at 4.1-5.5 TB/s (profiles/r02/decode_c2_L46_kernel_stats.csv).  Here the arithmetic is plain fp32 FMA on the vector
// ALU (8 FMAs per 16 weight bytes and token: 1/4 of the VALU rate the stream needs) and everything is arranged for
// bytes in flight:
//   * ONE persistent 512-thread workgroup per CU (256 of them); the weight rows are cut into tiles of 4 rows x all K,
//     a workgroup owns a contiguous run of tiles, its 8 waves pull tiles from an LDS counter;
//   * a wave streams a tile in units of 4 rows x SEGS x 1 KiB (lane l of a wave reads 16 B at k = 8 (64 s + l): every
//     wave-instruction is one whole KiB of one row) with TWO units in registers: the loads of unit i+1 are issued before
//     the FMAs of unit i, across tile boundaries, so 16-32 KiB per wave stay in flight from the first to the last byte;
//   * the token rows live in LDS for the whole kernel (fragment read = one conflict-free ds_read_b128 per segment).
// Optional prologue (PRO_COMBINE): the activation row is not loaded but COMPUTED by every workgroup — MoE combine of the
// previous layer + the input add-RMSNorm (decode_layer.py:47-49 of the next layer) — behind the workgroup's first weight
// loads, which removes the combine + norm launch of a one-token decode layer.  Bit-identical to that launch: same code
// (fused_rows.h), same 512-thread mapping.
#include "gemm_internal.h"
#include "fused_rows.h"
#include <string.h>
#include <type_traits>

namespace {

constexpr int TM_NT = 512;
constexpr int TM_NW = TM_NT / 64;
constexpr int TM_RPW = 4;      // weight rows per tile (dense launches; grouped launches: template parameter RPW)
constexpr int TM_MAXG = 320;   // groups a grouped launch may have (list of non-empty groups in LDS)

#ifndef GLM_TM_NT_LOADS
#define GLM_TM_NT_LOADS 1      // weight loads with the nt hint (read once, by one CU); 0 for A/B runs
#endif

__device__ __forceinline__ u32x4_t tm_wload(const void* p) {
    typedef __attribute__((address_space(1))) const u32x4_t gvec_t;
#if GLM_TM_NT_LOADS
    return __builtin_nontemporal_load(reinterpret_cast<gvec_t*>((const __attribute__((address_space(1))) void*)p));
#else
    return *reinterpret_cast<gvec_t*>((const __attribute__((address_space(1))) void*)p);
#endif
}

__device__ const u32x4_t g_tm_zero_page[4] = {};

// GLM_TM_STAMP=1 (timing study only): per workgroup the 100 MHz real-time counter at kernel entry, when its first weight unit has
// landed (first FMA), and at its end — where do the ~4.5 us of a launch that move no bytes go (dispatch skew, first byte, tail)?
#ifndef GLM_TM_STAMP
#define GLM_TM_STAMP 0
#endif
#ifndef GLM_TM_RPW_PLAIN
#define GLM_TM_RPW_PLAIN 2   // weight rows per tile of the dense one-token projections (r03 sweep, QKV / o_proj at Air dims:
                             // 4 rows x 4 KiB units 23.3 / 20.4 us, 2 x 4 KiB 22.4 / 19.5, 2 x 8 KiB 23.2 / 19.4, 1 x 8 KiB 24.5 / 20.4;
                             // with 4 rows o_proj is 4 tiles per workgroup, i.e. 4 of its 8 waves stream)
#endif
#ifndef GLM_TM_SEGS_PLAIN
#define GLM_TM_SEGS_PLAIN 4  // KiB of a weight row per unit
#endif
#if GLM_TM_STAMP
__device__ unsigned long long g_tm_stamps[512][3];
#endif

struct TinyParams {
    GemmParams g;
    int n_tokens;             // grouped gate|up: rows of x (= tokens) kept in LDS
    // PRO_COMBINE: the activation rows are bf16(add_rmsnorm(moe_combine(y_prev), residual)) computed in the kernel
    const bf16_t* pc_y;       // rows of the previous layer's grouped down projection
    int64_t pc_ldy;
    const int32_t* pc_pos;    // pair_pos [T, k]
    int pc_k;
    int64_t pc_shared_base;
    const bf16_t* pc_res_in;  // [T, H]
    const bf16_t* pc_norm_w;  // [H]
    bf16_t* pc_res_out;       // [T, H]  (written by workgroup 0)
    bf16_t* pc_x_out;         // [T, H] or nullptr: the normalised rows, for callers that want them (workgroup 0)
    float pc_eps;
    // TAIL_ROUTER (one token): the projection's output row is the attention's o_proj; the workgroup that finishes LAST
    // continues with the post-attention add-RMSNorm, the router (R1..R7) and glm_moe_align's structure for that token
    unsigned* tr_ticket;       // TWO zeroed uint32 (projection workgroups done, designated workgroups done); left zero
    float* tr_logits;          // [E] scratch (logits in bf16 precision, handed between the designated workgroups)
    const bf16_t* tr_res_in;   // [H]
    const bf16_t* tr_norm_w;   // [H]
    bf16_t* tr_y;              // [H] normalised hidden state (the experts' input)
    bf16_t* tr_res_out;        // [H]
    const bf16_t* tr_gate_w;   // [E, H]
    const float* tr_bias;      // [E]
    int64_t* tr_idx;           // [k]
    bf16_t* tr_w;              // [k]
    int32_t* tr_offs;          // moe_align outputs
    int32_t* tr_tok;
    bf16_t* tr_scale;
    int32_t* tr_pos;
    int tr_E, tr_k, tr_shared;
    float tr_eps, tr_rsf;
};

template <int M>
__device__ __forceinline__ void tm_fma8(float (&acc)[M], const u32x4_t w, const u32x4_t (&x)[M]) {
    float wf[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        wf[2 * q] = bflo(w[q]);
        wf[2 * q + 1] = bfhi(w[q]);
    }
#pragma unroll
    for (int m = 0; m < M; ++m) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[m] = __builtin_fmaf(wf[2 * q], bflo(x[m][q]), acc[m]);
            acc[m] = __builtin_fmaf(wf[2 * q + 1], bfhi(x[m][q]), acc[m]);
        }
    }
}

// Post-attention add-RMSNorm + router + bucketing of ONE token, continued inside the o_proj GEMV launch by ND DESIGNATED
// workgroups (blockIdx.x < ND, ND = ceil(E / TR_ROWS)).  A single workgroup cannot do it: the E gate rows are 512 KiB that
// nobody has touched yet, and one CU takes in 25-50 GB/s (measured r03: the single-workgroup tail cost +13 us per layer).
//   * at kernel START workgroup d requests ITS gate rows [d TR_ROWS, +TR_ROWS) into LDS by LDS-DMA (no registers held;
//     the rows do not depend on the activations) and then streams its share of the projection like everybody else;
//   * every workgroup publishes its output rows write-through, drains them and adds to the ticket; the designated ones
//     then poll the ticket (one lane, sc1 loads, s_sleep; they wait for workgroups that never wait for anything, so the
//     wait ends whatever the residency; the spin is bounded all the same);
//   * each designated workgroup forms the normalised row (all of them, redundantly: 8 KiB), its TR_ROWS logits out of LDS,
//     stores them write-through and adds to a second counter; the one whose add comes last gathers the E logits, selects
//     (router_select_wave) and writes glm_moe_align's structure for the token, then puts both counters back to zero.
// Arithmetic, rounding points and summation orders are router_topk_kernel<PRE = 1>'s (moe_misc.hip): the squares chain
// of threads 0..255 over vectors j, j + 256, the per-expert FMA chain over a lane's chunks lane, lane + 64, ... ending
// in wave_sum, router_select_wave itself — bit-identical to that launch fed with the projection's bf16 row.
constexpr int TR_ROWS = 16;            // gate rows per designated workgroup (16 x 8 KiB = 128 KiB of LDS at H = 4096)
constexpr unsigned TR_SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ size_t tr_gate_lds_off(int Kp) { return ((size_t)Kp * 2 + 15) & ~(size_t)15; }

__device__ __forceinline__ void tinym_router_prefetch(const TinyParams& tp, char* smem, int Kp) {
    // rows [d TR_ROWS, +TR_ROWS) of the gate matrix -> LDS behind the activation row; wave w takes rows w, w + 8, ...
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int H = tp.g.N, nvec = H >> 3, e0 = blockIdx.x * TR_ROWS;
    char* dst = smem + tr_gate_lds_off(Kp);
    for (int r = wid; r < TR_ROWS && e0 + r < tp.tr_E; r += TM_NW) {
        const bf16_t* grow = tp.tr_gate_w + (int64_t)(e0 + r) * H;
        for (int u = 0; 64 * u < nvec; ++u)  // lanes past the row re-read its last chunk into padding nobody reads
            __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)(grow + (int64_t)min(64 * u + lane, nvec - 1) * 8),
                                             (__attribute__((address_space(3))) void*)(dst + (size_t)r * ((nvec + 63) / 64) * 1024 + u * 1024),
                                             16, 0, 0);
    }
}

__device__ __forceinline__ void tinym_router_tail(const TinyParams& tp, char* smem, float* red, int Kp) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = tp.g.N, E = tp.tr_E, nvec = H >> 3;
    const int d = blockIdx.x, nd = (E + TR_ROWS - 1) / TR_ROWS;
    unsigned* t_all = tp.tr_ticket;       // workgroups whose projection rows are published
    unsigned* t_des = tp.tr_ticket + 1;   // designated workgroups whose logits are published
    __shared__ int tr_ok, tr_last;
    // ---- wait for the whole projection (one lane polls; everybody else parks at the barrier) ----
    if (tid == 0) {
        unsigned spins = 0;
        int ok = 1;
        while (__hip_atomic_load(t_all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > TR_SPIN_LIMIT) {
                ok = 0;
                break;
            }
        }
        tr_ok = ok;
    }
    __syncthreads();
    if (!tr_ok) {
        // Cannot happen while the GPU makes progress.  Never hang, never let stale data pass for a result, and say so:
        // the status word (ticket[2], sticky) is read by the host where it synchronises anyway (hip_ops.check_router_tail
        // raises and re-zeroes the workspace: the two counters are left non-zero here, and workgroups that were late may
        // still add to them — only the host, after a device synchronisation, can put them back).  Every output of the
        // launch is poisoned: the normalised row and the routing weights with NaNs, the bucketing with "no rows".
        if (tid == 0) __hip_atomic_store(tp.tr_ticket + 2, 1u + (unsigned)d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d == 0) {
            for (int i = tid; i < (H >> 1); i += TM_NT) reinterpret_cast<unsigned*>(tp.tr_y)[i] = 0x7fc07fc0u;  // NaN
            const int G = E + (tp.tr_shared ? 1 : 0);
            for (int i = tid; i <= G; i += TM_NT) tp.tr_offs[i] = 0;
            for (int i = tid; i < tp.tr_k; i += TM_NT) {
                tp.tr_idx[i] = 0;
                tp.tr_w[i] = (bf16_t)0x7fc0;
                tp.tr_pos[i] = -1;
            }
        }
        return;
    }
    // the x rows are dead (every wave is past its last tile): xs / hand-over area / logits / bias live there
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);                         // [H] normalised row
    float* xch = reinterpret_cast<float*>(smem + (size_t)H * 2);           // [(nvec - 256) * 8]
    float* lg = xch + (nvec > 256 ? (nvec - 256) * 8 : 0);                 // [E]
    float* bias_l = lg + ROUTER_MAX_E;                                     // [E]
    unsigned* keys = reinterpret_cast<unsigned*>(bias_l + ROUTER_MAX_E);   // [64]
    bf16_t* wl = reinterpret_cast<bf16_t*>(keys + 64);                     // [k]
    const bool act = tid < nvec;
    float s8[8], nwf[8];
    float ss = 0.f;
    if (act) {
        const unsigned long long a = ld_sc1_u64(tp.g.y + (int64_t)tid * 8), b = ld_sc1_u64(tp.g.y + (int64_t)tid * 8 + 4);
        const u32x4_t xv = {(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
        const u32x4_t rv = reinterpret_cast<const u32x4_t*>(tp.tr_res_in)[tid];
        const u32x4_t wv = reinterpret_cast<const u32x4_t*>(tp.tr_norm_w)[tid];
        u32x4_t ov;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s8[2 * j] = bflo(xv[j]) + bflo(rv[j]);
            s8[2 * j + 1] = bfhi(xv[j]) + bfhi(rv[j]);
            ov[j] = pack2bf(s8[2 * j], s8[2 * j + 1]);
            nwf[2 * j] = bflo(wv[j]);
            nwf[2 * j + 1] = bfhi(wv[j]);
        }
        if (d == 0) reinterpret_cast<u32x4_t*>(tp.tr_res_out)[tid] = ov;
    }
    if (act && tid >= 256) {
#pragma unroll
        for (int j = 0; j < 8; ++j) xch[(tid - 256) * 8 + j] = s8[j];
    }
    __syncthreads();
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
    if (lane == 0) red[wid] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < TM_NW; ++w) tot += red[w];  // (the 1024-thread kernel adds eight more waves' zeros)
    const float rstd = 1.0f / sqrtf(tot / (float)H + tp.tr_eps);
    if (act) {
        u32x4_t o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a = rbf(s8[2 * i] * rstd) * nwf[2 * i];
            const float b = rbf(s8[2 * i + 1] * rstd) * nwf[2 * i + 1];
            o[i] = pack2bf(a, b);
        }
        if (d == 0) reinterpret_cast<u32x4_t*>(tp.tr_y)[tid] = o;
        reinterpret_cast<u32x4_t*>(xs)[tid] = o;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's gate rows (requested at kernel start) have landed
    __syncthreads();
    // R1 for this workgroup's rows, out of LDS: logits = bf16(fp32 dot), the FMA chain of router_topk_kernel
    const char* gl = smem + tr_gate_lds_off(Kp);
    const size_t rowb = (size_t)((nvec + 63) / 64) * 1024;
    for (int r = wid; r < TR_ROWS && d * TR_ROWS + r < E; r += TM_NW) {
        const u32x4_t* grow = reinterpret_cast<const u32x4_t*>(gl + (size_t)r * rowb);
        float acc = 0.f;
        for (int c = lane; c < nvec; c += 64) {
            const u32x4_t wv = grow[c];
            const u32x4_t xv = reinterpret_cast<const u32x4_t*>(xs)[c];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc = fmaf(bflo(xv[q]), bflo(wv[q]), acc);
                acc = fmaf(bfhi(xv[q]), bfhi(wv[q]), acc);
            }
        }
        const float sdot = wave_sum(acc);
        if (lane == 0) st_sc1_u32(tp.tr_logits + d * TR_ROWS + r, __float_as_uint(rbf(sdot)));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned got = __hip_atomic_fetch_add(t_des, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tr_last = got == (unsigned)nd - 1u;
    }
    __syncthreads();
    if (!tr_last) return;
    // ---- the last designated workgroup: every logit was stored write-through and drained before its ticket ----
    for (int e = tid; e < E; e += TM_NT) {
        lg[e] = __uint_as_float(ld_sc1_u32(tp.tr_logits + e));
        bias_l[e] = tp.tr_bias[e];
    }
    if (tid == 0) {  // both counters back to zero for the next launch (every designated workgroup is past its polls)
        __hip_atomic_store(t_all, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(t_des, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (wid == 0)  // R2..R7 + the token's moe_align structure (one token: the selecting wave buckets)
        router_select_wave<ROUTER_MAX_E / 64>(lg, bias_l, keys, wl, lane, E, tp.tr_k, tp.tr_rsf, 0, tp.tr_idx, tp.tr_w, nullptr,
                                              false, tp.tr_offs, tp.tr_tok, tp.tr_scale, tp.tr_pos, tp.tr_shared);
}

// MODE as gemm_smallm; M = token rows per group block (1 or 2); SEGS = 1-KiB segments of a weight row per unit
template <int MODE, int M, int SEGS, bool PRO, bool TAIL = false, int RPW = TM_RPW>
__global__ __launch_bounds__(TM_NT) void gemm_tinym_kernel(const TinyParams tp) {
    static_assert(RPW == 1 || RPW == 2 || RPW == 4, "1, 2 or 4 weight rows per tile");
    static_assert(MODE != GEMM_GATEUP || RPW >= 2, "gate|up: a tile is gate rows + the matching up rows");
    constexpr int HALF = RPW >= 2 ? RPW / 2 : 1;  // gate|up: HALF gate rows + the matching HALF up rows
    const GemmParams& p = tp.g;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int sh_grp[TM_MAXG];   // non-empty groups
    __shared__ int sh_nact;
    __shared__ int sh_next;           // next tile of this workgroup
    __shared__ float sh_red[TM_NW];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = p.K;
    constexpr int KU = SEGS * 512;                 // k elements per unit
    const int nunits = (K + KU - 1) / KU;
    const int Kp = nunits * KU;                    // padded row length in LDS (pad = zeros)
    const int tpg = MODE == GEMM_GATEUP ? p.N / HALF : p.N / RPW;  // tiles per group

    // ---- non-empty groups (grouped modes) ----
    int nact = 1;
    if (p.offs != nullptr) {
        if (wid == 0) {
            int base = 0;
            for (int b = 0; b < p.G; b += 64) {
                const int gi = b + lane;
                const bool on = gi < p.G && p.offs[gi + 1] > p.offs[gi];
                const unsigned long long mask = __ballot(on);
                if (on) sh_grp[base + __popcll(mask & ((1ull << lane) - 1))] = gi;
                base += __popcll(mask);
            }
            if (lane == 0) sh_nact = base;
        }
        __syncthreads();
        nact = sh_nact;
    }
    const int64_t total = (int64_t)nact * tpg;
    // tiles of this workgroup: a contiguous run, positions q = 0 .. cnt-1 (every gridDim-th tile instead measured the same: r03)
    const int tb = (int)(total * blockIdx.x / gridDim.x), te = (int)(total * (blockIdx.x + 1) / gridDim.x);
    const int cnt = te - tb;
    auto tile_of = [&](int q) -> int { return tb + q; };
    if (tid == 0) sh_next = TM_NW;  // waves start with positions wid
#if GLM_TM_STAMP
    if (tid == 0 && blockIdx.x < 512) g_tm_stamps[blockIdx.x][0] = __builtin_amdgcn_s_memrealtime();
    bool stamped_first = false;
#endif

    // ---- per-tile set-up ----
    struct Tile {
        const char* wrow[RPW];
        int grp, ncol, m0, mcnt;
    };
    auto set_tile = [&](Tile& t, int tile) {
        const int slot = tile / tpg, ti = tile - slot * tpg;
        const int grp = p.offs ? sh_grp[slot] : 0;
        t.grp = grp;
        t.m0 = p.offs ? p.offs[grp] : 0;
        t.mcnt = p.offs ? min(p.offs[grp + 1] - t.m0, M) : (int)min((int64_t)M, p.R);
        const char* wg = reinterpret_cast<const char*>(p.w + (int64_t)grp * p.w_gstride);
        if (MODE == GEMM_GATEUP) {  // rows: gate n .. n+HALF-1, then up n .. n+HALF-1
            t.ncol = HALF * ti;
#pragma unroll
            for (int r = 0; r < RPW; ++r) t.wrow[r] = wg + ((int64_t)((r / HALF) * p.N + t.ncol + (r % HALF)) * p.ldw) * 2;
        } else {
            t.ncol = RPW * ti;
#pragma unroll
            for (int r = 0; r < RPW; ++r) t.wrow[r] = wg + ((int64_t)(t.ncol + r) * p.ldw) * 2;
        }
    };

    u32x4_t wbuf[2][SEGS][RPW];
    const char* zero = reinterpret_cast<const char*>(g_tm_zero_page);
    auto issue = [&](auto buf_tag, const Tile& t, int u) {
        constexpr int B = decltype(buf_tag)::value;
#pragma unroll
        for (int s = 0; s < SEGS; ++s) {
            const int k = u * KU + (s * 64 + lane) * 8;
#pragma unroll
            for (int r = 0; r < RPW; ++r) wbuf[B][s][r] = tm_wload(k < K ? t.wrow[r] + (int64_t)k * 2 : zero);
        }
    };

    // issue state runs one unit ahead of the compute state
    Tile it, ct;
    int i_tile = wid, i_u = 0;
    bool i_live = i_tile < cnt;
    if (i_live) {
        set_tile(it, tile_of(i_tile));
        issue(std::integral_constant<int, 0>{}, it, 0);
    }
    if constexpr (TAIL) {  // a designated workgroup's gate rows -> LDS, behind its first weight loads
        if ((int)blockIdx.x < (tp.tr_E + TR_ROWS - 1) / TR_ROWS) tinym_router_prefetch(tp, smem, Kp);
    }
    ct = it;
    int c_tile = i_tile, c_u = 0;
    bool c_live = i_live;
    auto advance_issue = [&]() {  // next (tile, unit) of this wave's stream
        if (++i_u == nunits) {
            i_u = 0;
            int nt = 0;
            if (lane == 0) nt = __hip_atomic_fetch_add(&sh_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            i_tile = __builtin_amdgcn_readfirstlane(nt);
            i_live = i_tile < cnt;
            if (i_live) set_tile(it, tile_of(i_tile));
        }
    };

    // ---- activation rows into LDS: [rows][Kp] bf16, pad zeroed ----
    //   PLAIN: rows 0..M-1 = x rows;  GATEUP: row t = token t (T = p.R / ... tokens, all of x);  DOWN: row r = act row r
    const int xrows = MODE == GEMM_PLAIN ? M : (MODE == GEMM_GATEUP ? tp.n_tokens : (int)p.R);
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
    if (PRO) {
        // combine of the previous MoE block + input add-RMSNorm, token by token, exactly the stand-alone launch's code;
        // the first unit's weight loads are already in flight
        for (int t = 0; t < M; ++t) {
            if (t >= (int)p.R) break;
            glm_combine_add_rmsnorm_row<1>(tp.pc_y, tp.pc_ldy, tp.pc_pos, tp.pc_k, tp.pc_shared_base, tp.pc_res_in,
                                           tp.pc_norm_w, blockIdx.x == 0 ? tp.pc_x_out : nullptr,
                                           blockIdx.x == 0 ? tp.pc_res_out : nullptr, xs + (int64_t)t * Kp, K, tp.pc_eps, t,
                                           sh_red);
            __syncthreads();  // sh_red and the position list are reused by the next token
        }
        for (int i = tid; i < M * (Kp - K) / 8; i += TM_NT) {
            const int r = i / ((Kp - K) / 8), c = i % ((Kp - K) / 8);
            reinterpret_cast<u32x4_t*>(xs + (int64_t)r * Kp + K)[c] = u32x4_t{0, 0, 0, 0};
        }
        for (int t = (int)p.R; t < M; ++t)  // a missing second row: zeros
            for (int i = tid; i < Kp / 8; i += TM_NT) reinterpret_cast<u32x4_t*>(xs + (int64_t)t * Kp)[i] = u32x4_t{0, 0, 0, 0};
    } else {
        const int vpr = Kp / 8;
        for (int i = tid; i < xrows * vpr; i += TM_NT) {
            const int r = i / vpr, c = i - r * vpr;
            u32x4_t v = {0, 0, 0, 0};
            const int64_t srow = MODE == GEMM_PLAIN ? min((int64_t)r, p.R - 1) : r;
            if (c * 8 < K) v = *reinterpret_cast<const u32x4_t*>(p.x + srow * p.ldx + c * 8);
            reinterpret_cast<u32x4_t*>(xs)[i] = v;
        }
    }
    __syncthreads();

    float acc[RPW][M];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int m = 0; m < M; ++m) acc[r][m] = 0.f;

    auto xrow_of = [&](const Tile& t, int m) -> const bf16_t* {
        const int j = m < t.mcnt ? m : 0;  // a group with fewer rows than M: the extra column is computed and dropped
        int row;
        if (MODE == GEMM_PLAIN) row = j;
        else if (MODE == GEMM_GATEUP) row = p.src ? p.src[t.m0 + j] : t.m0 + j;
        else row = t.m0 + j;
        return xs + (int64_t)row * Kp;
    };

    auto compute = [&](auto buf_tag, const Tile& t, int u, const bf16_t* const (&xr)[M]) {
        constexpr int B = decltype(buf_tag)::value;
#pragma unroll
        for (int s = 0; s < SEGS; ++s) {
            u32x4_t xv[M];
#pragma unroll
            for (int m = 0; m < M; ++m) xv[m] = *reinterpret_cast<const u32x4_t*>(xr[m] + u * KU + (s * 64 + lane) * 8);
#pragma unroll
            for (int r = 0; r < RPW; ++r) tm_fma8<M>(acc[r], wbuf[B][s][r], xv);
        }
    };

    auto finish = [&](const Tile& t) {
        // cross-lane sums (every lane ends with the total), then the mode's epilogue from lane 0
        float o[RPW][M];
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int m = 0; m < M; ++m) {
                o[r][m] = wave_sum(acc[r][m]);
                acc[r][m] = 0.f;
            }
        if (lane != 0) return;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            if (m >= t.mcnt) continue;
            const int64_t row = t.m0 + m;
            if (MODE == GEMM_GATEUP) {
                float a[HALF];
#pragma unroll
                for (int j = 0; j < HALF; ++j) a[j] = rbf(glm_silu(rbf(o[j][m]))) * rbf(o[HALF + j][m]);
                if constexpr (HALF == 2) *reinterpret_cast<unsigned*>(p.y + row * p.ldy + t.ncol) = pack2bf(a[0], a[1]);
                else p.y[row * p.ldy + t.ncol] = f2bf(a[0]);
            } else {
                float v[RPW];
                float sc = 1.0f;
                const bool scaled = MODE == GEMM_DOWN && t.grp < p.n_scaled;
                if (scaled) sc = bf2f(p.scale[row]);
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    float a = o[r][m];
                    if (MODE == GEMM_PLAIN && p.bias) a += bf2f(p.bias[t.ncol + r]);
                    if (scaled) a = rbf(a) * sc;
                    v[r] = a;
                }
                if constexpr (RPW == 4) {
                    const u32x2_t o2 = u32x2_t{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
                    if (TAIL)  // read by another workgroup of this launch: write-through (Guideline 16 R1)
                        st_sc1_u64(p.y + row * p.ldy + t.ncol, (unsigned long long)o2[0] | ((unsigned long long)o2[1] << 32));
                    else
                        *reinterpret_cast<u32x2_t*>(p.y + row * p.ldy + t.ncol) = o2;
                } else if constexpr (RPW == 2) {
                    if (TAIL) st_sc1_u32(p.y + row * p.ldy + t.ncol, pack2bf(v[0], v[1]));
                    else *reinterpret_cast<unsigned*>(p.y + row * p.ldy + t.ncol) = pack2bf(v[0], v[1]);
                } else {
                    if (TAIL)
                        __hip_atomic_store(reinterpret_cast<unsigned short*>(p.y + row * p.ldy + t.ncol), f2bf(v[0]), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    else
                        p.y[row * p.ldy + t.ncol] = f2bf(v[0]);
                }
            }
        }
    };

    // ---- the stream: units alternate between the two register buffers ----
    const bf16_t* xr[M];
    if (c_live) {
#pragma unroll
        for (int m = 0; m < M; ++m) xr[m] = xrow_of(ct, m);
    }
    while (c_live) {
        // buffer 0 holds (c_tile, c_u)
        advance_issue();
        if (i_live) issue(std::integral_constant<int, 1>{}, it, i_u);
        compute(std::integral_constant<int, 0>{}, ct, c_u, xr);
#if GLM_TM_STAMP
        if (!stamped_first) {
            stamped_first = true;
            asm volatile("" ::"v"(acc[0][0]));
            if (tid == 0 && blockIdx.x < 512) g_tm_stamps[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
        }
#endif
        if (++c_u == nunits) {
            finish(ct);
            c_u = 0;
            c_tile = i_tile;
            c_live = i_live;
            ct = it;
            if (c_live) {
#pragma unroll
                for (int m = 0; m < M; ++m) xr[m] = xrow_of(ct, m);
            }
        }
        if (!c_live) break;
        // buffer 1 holds (c_tile, c_u)
        advance_issue();
        if (i_live) issue(std::integral_constant<int, 0>{}, it, i_u);
        compute(std::integral_constant<int, 1>{}, ct, c_u, xr);
        if (++c_u == nunits) {
            finish(ct);
            c_u = 0;
            c_tile = i_tile;
            c_live = i_live;
            ct = it;
            if (c_live) {
#pragma unroll
                for (int m = 0; m < M; ++m) xr[m] = xrow_of(ct, m);
            }
        }
    }
#if GLM_TM_STAMP
    __syncthreads();
    if (tid == 0 && blockIdx.x < 512) g_tm_stamps[blockIdx.x][2] = __builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (TAIL) {
        // ---- hand-off (valid-forms table, row 1): every wave drains its write-through stores, the workgroup meets, one
        // lane adds to the ticket; the designated workgroups poll it (sc1 loads) and load behind their barrier ----
        static_assert(MODE == GEMM_PLAIN && M == 1, "the router tail continues a one-row dense projection");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(tp.tr_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)blockIdx.x >= (tp.tr_E + TR_ROWS - 1) / TR_ROWS) return;
        tinym_router_tail(tp, smem, sh_red, Kp);
    }
}

int n_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return n;
}

template <int MODE, int M, int SEGS, bool PRO, bool TAIL = false, int RPW = TM_RPW>
int launch_tiny(const TinyParams& tp, int xrows, hipStream_t st) {
    const GemmParams& p = tp.g;
    constexpr int KU = SEGS * 512;
    const int Kp = (p.K + KU - 1) / KU * KU;
    // >= 84 KiB of LDS per workgroup: exactly one workgroup per CU, so the 256 of them cover the chip evenly
    size_t lds = (size_t)xrows * Kp * 2;
    if (lds < 84 * 1024) lds = 84 * 1024;
    GLM_CHECK_ARG(lds <= 156 * 1024, GLM_ERR_UNSUPPORTED, "gemm_tinym: %d activation rows of %d do not fit the LDS", xrows, Kp);
    auto kern = gemm_tinym_kernel<MODE, M, SEGS, PRO, TAIL, RPW>;
    GLM_SET_MAX_LDS(kern, 156 * 1024, "gemm_tinym");
    const int64_t tiles = (int64_t)(p.offs ? p.G : 1) * (MODE == GEMM_GATEUP ? p.N / (RPW / 2) : p.N / RPW);
    int nwg = n_cus();
    if (tiles < nwg) nwg = (int)tiles;
    if (TAIL) {  // the tail's scratch lives in the (dead) activation area; behind it TR_ROWS gate rows
        const size_t scratch = (size_t)p.N * 2 + (size_t)(p.N / 8 > 256 ? (p.N / 8 - 256) * 32 : 0) + 2 * ROUTER_MAX_E * 4 + 64 * 4 + 256;
        const size_t x_area = (((size_t)Kp * 2 + 15) & ~(size_t)15);
        GLM_CHECK_ARG(scratch <= x_area, GLM_ERR_UNSUPPORTED, "gemm_tinym(router tail): K too short for the tail's scratch");
        lds = x_area + (size_t)TR_ROWS * ((p.N / 8 + 63) / 64) * 1024;
        GLM_CHECK_ARG(lds <= 156 * 1024, GLM_ERR_UNSUPPORTED, "gemm_tinym(router tail): %zu B of LDS", lds);
        if (lds < 84 * 1024) lds = 84 * 1024;
    }
    if (nwg <= 0) return GLM_OK;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(TM_NT), lds, st, tp);
    GLM_CHECK_LAUNCH("gemm_tinym");
    return GLM_OK;
}

bool tiny_shape_ok(int mode, const GemmParams& p) {
    if (p.K % 8 != 0 || p.ldx % 8 != 0 || p.ldw % 8 != 0 || !glm_aligned16(p.x) || !glm_aligned16(p.w)) return false;
    if (mode == GEMM_GATEUP) return p.N % 2 == 0 && p.ldy % 2 == 0 && (((uintptr_t)p.y) & 3) == 0;
    return p.N % TM_RPW == 0 && p.ldy % 4 == 0 && (((uintptr_t)p.y) & 7) == 0;
}

}  // namespace

// token rows this body takes: ONE.  (Two rows were built and measured in r03 — tools/tinym_probe.py: QKV 26.0 vs 21.8 us,
// grouped down 41.4 vs 34.7 us against the small-M MFMA kernels — and dropped.)
int glm_tinym_max_rows() { return 1; }

// One token row per weight matrix / per expert group (T = 1).  n_tokens: rows of x (grouped gate|up), must be 1.
int glm_launch_gemm_tinym(int mode, const GemmParams& p, int n_tokens, hipStream_t st) {
    if (!tiny_shape_ok(mode, p) || (p.offs != nullptr && p.G > TM_MAXG)) {
        glm_set_error("gemm_tinym: shape not covered (mode %d K=%d N=%d G=%d)", mode, p.K, p.N, p.G);
        return GLM_ERR_UNSUPPORTED;
    }
    TinyParams tp;
    memset(&tp, 0, sizeof(tp));
    tp.g = p;
    tp.n_tokens = n_tokens;
    const int M = mode == GEMM_PLAIN ? (int)p.R : n_tokens;
    GLM_CHECK_ARG(M == 1, GLM_ERR_UNSUPPORTED, "gemm_tinym: %d rows per block", M);
    // units: 4 rows x 4 KiB for long rows; a short K (expert down projection, 1408) as ONE unit of 3-KiB rows
    if (mode == GEMM_PLAIN) return launch_tiny<GEMM_PLAIN, 1, GLM_TM_SEGS_PLAIN, false, false, GLM_TM_RPW_PLAIN>(tp, 1, st);
    // grouped launches: rows per tile chosen for the TAIL of the launch — a workgroup's waves pull tiles from a counter, and
    // the launch ends when the slowest wave has finished its last tile; at Air dims one token gives 24.75 tiles of 4 rows per
    // workgroup for gate|up (7 waves take 3, one takes 4: +29 % on the critical wave), 49.5 tiles of 2 rows even that out
#ifndef GLM_TM_RPW_GATEUP
#define GLM_TM_RPW_GATEUP 2
#endif
#ifndef GLM_TM_RPW_DOWN
#define GLM_TM_RPW_DOWN 4
#endif
    if (mode == GEMM_GATEUP) return launch_tiny<GEMM_GATEUP, 1, (GLM_TM_RPW_GATEUP == 2 ? 8 : 4), false, false, GLM_TM_RPW_GATEUP>(tp, n_tokens, st);
    if (p.K <= 1536) return launch_tiny<GEMM_DOWN, 1, 3, false, false, GLM_TM_RPW_DOWN>(tp, (int)p.R, st);
    return launch_tiny<GEMM_DOWN, 1, 4, false, false, GLM_TM_RPW_DOWN>(tp, (int)p.R, st);
}

// QKV projection (+bias) of ONE token row whose input is computed in the kernel: moe_combine of the previous layer's expert
// rows + add-RMSNorm with `residual_in` (see the file header).  p.x is ignored; p.K = H.
int glm_launch_gemm_tinym_combine_norm(const GemmParams& p, const void* y_prev, int64_t ld_yprev, const int32_t* pair_pos,
                                       int k, int64_t shared_base, const void* residual_in, const void* norm_w,
                                       void* residual_out, void* x_out, float eps, hipStream_t st) {
    GemmParams q = p;
    q.x = (const bf16_t*)residual_in;  // any aligned pointer: the shape check wants one
    q.ldx = p.K;
    if (!tiny_shape_ok(GEMM_PLAIN, q) || p.K % 8 != 0 || p.K / 8 > TM_NT || p.R != 1 || k > GLM_COMBINE_MAX_K ||
        ld_yprev % 8 != 0 || !glm_aligned16(y_prev) || !glm_aligned16(residual_in) || !glm_aligned16(norm_w) ||
        !glm_aligned16(residual_out) || (x_out && !glm_aligned16(x_out))) {
        glm_set_error("gemm_tinym(combine+norm): shape not covered (K=%d N=%d T=%ld k=%d)", p.K, p.N, (long)p.R, k);
        return GLM_ERR_UNSUPPORTED;
    }
    TinyParams tp;
    memset(&tp, 0, sizeof(tp));
    tp.g = q;
    tp.pc_y = (const bf16_t*)y_prev; tp.pc_ldy = ld_yprev; tp.pc_pos = pair_pos; tp.pc_k = k; tp.pc_shared_base = shared_base;
    tp.pc_res_in = (const bf16_t*)residual_in; tp.pc_norm_w = (const bf16_t*)norm_w; tp.pc_res_out = (bf16_t*)residual_out;
    tp.pc_x_out = (bf16_t*)x_out; tp.pc_eps = eps;
    return launch_tiny<GEMM_PLAIN, 1, GLM_TM_SEGS_PLAIN, true, false, GLM_TM_RPW_PLAIN>(tp, 1, st);
}

// o_proj of ONE token row continued, in the last workgroup to finish, by the post-attention add-RMSNorm + router +
// bucketing (glm_add_rmsnorm_router_topk_align at T = 1).  p: the projection (y = its bf16 output row, written
// write-through and read back by the tail); ticket: two zeroed uint32 that the launch leaves zero + a third, sticky status
// word (0, or 1 + the designated workgroup whose bounded wait expired); logits_scratch: E floats.
int glm_launch_gemm_tinym_router_tail(const GemmParams& p, unsigned* ticket, float* logits_scratch, const void* residual_in, const void* norm_w,
                                      float eps, void* y_norm, void* residual_out, const void* gate_w, const float* bias, int E,
                                      int k, float rsf, int64_t* topk_idx, void* topk_w, int32_t* offs, int32_t* tok,
                                      void* scale, int32_t* pos, int with_shared, hipStream_t st) {
    const int H = p.N;
    if (!tiny_shape_ok(GEMM_PLAIN, p) || p.R != 1 || p.bias != nullptr || H % 8 != 0 || H / 8 > TM_NT || H < 2048 || E > ROUTER_MAX_E ||
        k > 64 || k > E || !glm_aligned16(residual_in) || !glm_aligned16(norm_w) || !glm_aligned16(y_norm) ||
        !glm_aligned16(residual_out) || !glm_aligned16(gate_w) || !glm_aligned16(p.y) || p.N / TM_RPW < 2 * ((E + TR_ROWS - 1) / TR_ROWS) ||
        logits_scratch == nullptr) {
        glm_set_error("gemm_tinym(router tail): shape not covered (K=%d H=%d E=%d k=%d)", p.K, H, E, k);
        return GLM_ERR_UNSUPPORTED;
    }
    TinyParams tp;
    memset(&tp, 0, sizeof(tp));
    tp.g = p;
    tp.tr_ticket = ticket; tp.tr_logits = logits_scratch; tp.tr_res_in = (const bf16_t*)residual_in; tp.tr_norm_w = (const bf16_t*)norm_w; tp.tr_eps = eps;
    tp.tr_y = (bf16_t*)y_norm; tp.tr_res_out = (bf16_t*)residual_out; tp.tr_gate_w = (const bf16_t*)gate_w; tp.tr_bias = bias;
    tp.tr_E = E; tp.tr_k = k; tp.tr_rsf = rsf; tp.tr_idx = topk_idx; tp.tr_w = (bf16_t*)topk_w;
    tp.tr_offs = offs; tp.tr_tok = tok; tp.tr_scale = (bf16_t*)scale; tp.tr_pos = pos; tp.tr_shared = with_shared;
    return launch_tiny<GEMM_PLAIN, 1, GLM_TM_SEGS_PLAIN, false, true, GLM_TM_RPW_PLAIN>(tp, 1, st);
}

#if GLM_TM_STAMP
extern "C" int glm_debug_tinym_stamps(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tm_stamps), sizeof(unsigned long long) * 3 * (n < 512 ? n : 512)) == hipSuccess ? 0 : -1;
}
#endif
