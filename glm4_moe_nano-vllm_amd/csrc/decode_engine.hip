// decode_engine.hip — BASELINE config 2 (greedy decode of ONE sequence): the MLP half of a decoder layer as ONE persistent
// launch (r05).  It replaces three launches of the one-token path — o_proj + post-attention add-RMSNorm + router + bucketing
// (glm_gemm_bf16_router_tail), the grouped gate|up GEMV + SiLU·mul and the grouped down GEMV x routing weight
// (gemm_tinym.hip) — i.e. attention_new.py o_proj, decode_layer.py:86-93, moe.py:28-50 and the expert loop moe.py:135-151
// with mlp.py:49-54 of the reference, whose CUDA-graph replay (engine/model_runner.py:223-258) is what the launch
// structure stands in for.  Same arithmetic, rounding points and fp32 summation orders as those launches: every output is
// BIT-IDENTICAL to theirs (tests/test_hip_kernels.py::test_decode_mlp_engine_*).
//
// Structure (cdna_hip_programming.md 5.6 / Guideline 16; MI355X_MICROARCH.md rows allgather, prefetch-credit, gather-pass):
//   * one 512-thread workgroup per CU, resident for the whole launch; a wave streams weight rows HBM -> registers in units
//     of 8-16 KiB with TWO units in flight (the GEMV body of gemm_tinym.hip) and ISSUES THE NEXT UNIT BEFORE IT WAITS for
//     whatever its current unit depends on — the weights of o_proj, of the shared expert and of every down projection do not
//     depend on activations, so the stream runs across the seams; only the identity of the k routed experts is late;
//   * hand-offs are 8-byte {tag = epoch, value} granules written by ONE write-through store each (R2 form: the data is the
//     flag; no fence on either side) and gathered into LDS by waves that have no weight loads in flight (vmcnt is in
//     order: a gathering wave would otherwise wait for its own prefetch first):
//       seam A  o_proj row (H/2 granules of two bf16)  -> every workgroup: add-RMSNorm redundantly, the normalised row stays
//               in LDS; workgroup e < E then forms logit e from the gate row it requested into LDS at kernel start
//               (LDS-DMA) and publishes it; wave 7 of every workgroup gathers the E logits and selects (router_select_wave:
//               the same code, hence the same set, order and weights on every workgroup); workgroup 0 also writes the
//               routing and glm_moe_align's structure for the next launch.  Meanwhile waves 0..2 compute the SHARED
//               expert's gate|up tiles (their weights were requested before the seam).
//       seam B  act rows ((k+1) I/2 granules)          -> wave 7 gathers them group by group WHILE the other waves stream
//               gate|up, in the order the groups are produced (shared, then ascending expert id); a down tile waits for
//               the LDS flag of its own group only.
//   * tiles are dealt round-robin (tile t -> workgroup t mod grid), pulled from LDS counters by the waves, groups in
//     production order, so that a wave that runs out of gate|up tiles starts on down tiles of groups that are complete.
//   * the epoch lives in device memory (state[0]) and is advanced by workgroup 0 once it is past seam A (every workgroup has
//     read it by then): graph-replay safe, no memset node.  Every spin is bounded; a wait that expires stores a sticky code
//     in state[1], poisons y with NaNs and lets the wave run on (never hang, never pass stale data silently).
#include "gemm_internal.h"
#include "fused_rows.h"
#include <string.h>
#include <type_traits>

namespace {

constexpr int EN_NT = 512;
constexpr int EN_NW = EN_NT / 64;
constexpr int EN_KU = 2048;          // k elements per unit of a 4-KiB-per-row unit (o_proj, gate|up)
constexpr int EN_IP = 1536;          // padded act row (down: ONE unit of 3 KiB per row)
constexpr unsigned EN_SPIN_GLOBAL = 1u << 20;
constexpr unsigned EN_SPIN_LDS = 1u << 24;

#ifndef GLM_EN_STAMP
#define GLM_EN_STAMP 0               // timing study: per-workgroup real-time stamps at the phase edges (profiles/r05)
#endif
#if GLM_EN_STAMP
__device__ unsigned long long g_en_stamps[256][8];
#endif

struct EngineParams {
    const bf16_t* x_o;        // [Ko] attention output row
    const bf16_t* w_o;        // [H, Ko]
    int64_t ldw_o;
    int Ko, H;
    const bf16_t* res_in;     // [H]
    const bf16_t* norm_w;     // [H]
    float eps;
    bf16_t* y_norm;           // [H]
    bf16_t* res_out;          // [H]
    const bf16_t* gate_w;     // [E, H]
    const float* bias;        // [E]
    int E, k;
    float rsf;
    int64_t* topk_idx;        // [k]
    bf16_t* topk_w;           // [k]
    int32_t* offs;            // [E + 2]
    int32_t* tok;             // [k + 1]
    bf16_t* scale;            // [k + 1]
    int32_t* pos;             // [k]
    const bf16_t* w13;        // [E + 1, 2 I, H]
    int64_t w13_gs, w13_ld;
    const bf16_t* w2;         // [E + 1, H, I]
    int64_t w2_gs, w2_ld;
    int I;
    bf16_t* act;              // [k + 1, I] or nullptr
    bf16_t* y;                // [k + 1, H]
    int64_t ldy;
    unsigned long long* gran_h;   // [H / 2]
    unsigned long long* gran_l;   // [E]
    unsigned long long* gran_a;   // [(k + 1) * I / 2]
    unsigned* state;              // [0] epoch  [1] sticky status
};

__device__ const u32x4_t g_en_zero_page[4] = {};

__device__ __forceinline__ u32x4_t en_wload(const void* p) {
    typedef __attribute__((address_space(1))) const u32x4_t gvec_t;
    return __builtin_nontemporal_load(reinterpret_cast<gvec_t*>((const __attribute__((address_space(1))) void*)p));
}

__device__ __forceinline__ void en_fma8(float& acc, const u32x4_t w, const u32x4_t x) {
    // gemm_tinym's tm_fma8<1>: the weight is the first factor, element order lo, hi of words 0..3
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        acc = __builtin_fmaf(bflo(w[q]), bflo(x[q]), acc);
        acc = __builtin_fmaf(bfhi(w[q]), bfhi(x[q]), acc);
    }
}

__host__ __device__ inline int en_roundup(int v, int m) { return (v + m - 1) / m * m; }

// LDS carve, shared by the kernel and the launcher
struct EnLds {
    int xo, hbuf, xs, act, gate, xch, lg, bias, keys, wl, ridx, rw, roffs, rtok, rscale, rpos, flags, red, total;
    __host__ __device__ EnLds(int Ko, int H, int k) {
        const int nvec = H >> 3;
        int o = 0;
        auto take = [&](int bytes) { const int at = o; o += (bytes + 15) & ~15; return at; };
        xo = take(en_roundup(Ko, EN_KU) * 2);
        hbuf = take(H * 2);
        xs = take(en_roundup(H, EN_KU) * 2);
        act = take((k + 1) * EN_IP * 2);
        gate = take(((nvec + 63) / 64) * 1024);
        xch = take(nvec > 256 ? (nvec - 256) * 32 : 16);
        lg = take(ROUTER_MAX_E * 4);
        bias = take(ROUTER_MAX_E * 4);
        keys = take(64 * 4);
        wl = take(64 * 2);
        ridx = take(64 * 8);
        rw = take(64 * 2);
        roffs = take((ROUTER_MAX_E + 2) * 4);
        rtok = take(65 * 4);
        rscale = take(66 * 2);
        rpos = take(64 * 4);
        flags = take((8 + 65) * 4);   // [0] route ready [1] abort [2] epoch [3] next_o [4] next_g [5] next_d [8 + row] act ready
        red = take(EN_NW * 4);
        total = o;
    }
};

__device__ __forceinline__ int en_lds_read(const volatile int* p) {
    const int v = *p;
    asm volatile("" ::: "memory");
    return __builtin_amdgcn_readfirstlane(v);   // wave-uniform by construction; tell the compiler (scalar control flow)
}

__device__ __forceinline__ void en_fail(const EngineParams& p, volatile int* flags, unsigned code) {
    __hip_atomic_store(p.state + 1, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    flags[1] = 1;
}

__global__ __launch_bounds__(EN_NT) void decode_mlp_engine_kernel(const EngineParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x, nb = gridDim.x;
    const int H = p.H, Ko = p.Ko, I = p.I, E = p.E, k = p.k;
    const int nvec = H >> 3;
    const int Hp = en_roundup(H, EN_KU);
    const int nuo = en_roundup(Ko, EN_KU) / EN_KU, nug = Hp / EN_KU;
    const int R = k + 1;
    const int tpgG = I / 2, tpgD = H / 4, ntile_h = H / 2;
    const EnLds L(Ko, H, k);
    bf16_t* xo = reinterpret_cast<bf16_t*>(smem + L.xo);
    unsigned* hbuf = reinterpret_cast<unsigned*>(smem + L.hbuf);
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem + L.xs);
    bf16_t* actl = reinterpret_cast<bf16_t*>(smem + L.act);
    float* xch = reinterpret_cast<float*>(smem + L.xch);
    float* lg = reinterpret_cast<float*>(smem + L.lg);
    float* bias_l = reinterpret_cast<float*>(smem + L.bias);
    unsigned* keys = reinterpret_cast<unsigned*>(smem + L.keys);
    bf16_t* wl = reinterpret_cast<bf16_t*>(smem + L.wl);
    int64_t* r_idx = reinterpret_cast<int64_t*>(smem + L.ridx);
    bf16_t* r_w = reinterpret_cast<bf16_t*>(smem + L.rw);
    int32_t* r_offs = reinterpret_cast<int32_t*>(smem + L.roffs);
    int32_t* r_tok = reinterpret_cast<int32_t*>(smem + L.rtok);
    bf16_t* r_scale = reinterpret_cast<bf16_t*>(smem + L.rscale);
    int32_t* r_pos = reinterpret_cast<int32_t*>(smem + L.rpos);
    volatile int* flags = reinterpret_cast<volatile int*>(smem + L.flags);
    float* red = reinterpret_cast<float*>(smem + L.red);
    const char* zero = reinterpret_cast<const char*>(g_en_zero_page);

#if GLM_EN_STAMP
    if (tid == 0 && b < 256) g_en_stamps[b][0] = __builtin_amdgcn_s_memrealtime();
#endif
    if (tid == 0) {
        unsigned e = __hip_atomic_load(p.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (e == 0u) e = 1u;
        flags[0] = 0;
        flags[1] = 0;
        flags[2] = (int)e;
        flags[3] = EN_NW;   // o_proj: the waves start with positions wid
        flags[4] = 0;
        flags[5] = 0;
    }
    for (int i = tid; i < R; i += EN_NT) flags[8 + i] = 0;

    // ================================================================ phase O: o_proj, 2 weight rows per tile ==========
    const int tbo = (int)((int64_t)ntile_h * b / nb), teo = (int)((int64_t)ntile_h * (b + 1) / nb);
    const int cnto = teo - tbo;
    u32x4_t wb[2][16];
    struct OT {
        const char* w0;
        const char* w1;
        int tile;
    };
    auto set_o = [&](OT& t, int tile) {
        t.tile = tile;
        t.w0 = reinterpret_cast<const char*>(p.w_o) + (int64_t)(2 * tile) * p.ldw_o * 2;
        t.w1 = t.w0 + p.ldw_o * 2;
    };
    auto issue_o = [&](auto bt, const OT& t, int u) {
        constexpr int B = decltype(bt)::value;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int kk = u * EN_KU + (s * 64 + lane) * 8;   // Ko is a multiple of EN_KU (launcher): no bound check
            wb[B][2 * s] = en_wload(t.w0 + (int64_t)kk * 2);
            wb[B][2 * s + 1] = en_wload(t.w1 + (int64_t)kk * 2);
        }
    };
    OT it, ct;
    int i_tile = wid, i_u = 0;
    bool i_live = i_tile < cnto;
    if (i_live) {
        set_o(it, tbo + i_tile);
        issue_o(std::integral_constant<int, 0>{}, it, 0);
    }
    // workgroup e < E: gate row e -> LDS by LDS-DMA, requested by wave 6 (the wave that will form the logit)
    if (wid == 6 && b < E) {
        const bf16_t* grow = p.gate_w + (int64_t)b * H;
        char* dst = smem + L.gate;
        for (int u = 0; 64 * u < nvec; ++u)
            __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)(grow + (int64_t)min(64 * u + lane, nvec - 1) * 8),
                                             (__attribute__((address_space(3))) void*)(dst + u * 1024), 16, 0, 0);
    }
    ct = it;
    int c_u = 0;
    bool c_live = i_live;
    // activation row + zero pads
    {
        const int vpr = en_roundup(Ko, EN_KU) / 8;
        for (int i = tid; i < vpr; i += EN_NT) {
            u32x4_t v = {0, 0, 0, 0};
            if (i * 8 < Ko) v = *reinterpret_cast<const u32x4_t*>(p.x_o + i * 8);
            reinterpret_cast<u32x4_t*>(xo)[i] = v;
        }
        for (int i = tid; i < (Hp - H) / 8; i += EN_NT) reinterpret_cast<u32x4_t*>(xs + H)[i] = u32x4_t{0, 0, 0, 0};
        for (int i = tid; i < R * EN_IP / 8; i += EN_NT) reinterpret_cast<u32x4_t*>(actl)[i] = u32x4_t{0, 0, 0, 0};
        if (wid == 7)
            for (int e = lane; e < E; e += 64) bias_l[e] = p.bias[e];
    }
    __syncthreads();
    const unsigned epoch = (unsigned)__builtin_amdgcn_readfirstlane(flags[2]);
    const unsigned long long etag = (unsigned long long)epoch << 32;

    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    auto advance_o = [&]() {
        if (++i_u == nuo) {
            i_u = 0;
            int nt = 0;
            if (lane == 0) nt = __hip_atomic_fetch_add((int*)&flags[3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            i_tile = __builtin_amdgcn_readfirstlane(nt);
            i_live = i_tile < cnto;
            if (i_live) set_o(it, tbo + i_tile);
        }
    };
    auto compute_o = [&](auto bt, int u) {
        constexpr int B = decltype(bt)::value;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const u32x4_t xv = *reinterpret_cast<const u32x4_t*>(xo + u * EN_KU + (s * 64 + lane) * 8);
            en_fma8(acc[0], wb[B][2 * s], xv);
            en_fma8(acc[1], wb[B][2 * s + 1], xv);
        }
    };
    auto finish_o = [&](const OT& t) {
        const float o0 = wave_sum(acc[0]), o1 = wave_sum(acc[1]);
        acc[0] = acc[1] = 0.f;
        if (lane == 0) st_sc1_u64(p.gran_h + t.tile, etag | pack2bf(o0, o1));
    };

    // ---- the shared expert's gate|up tile of this wave (weights known: requested before seam A) ----
    struct Item {
        const char* w[4];
        int phase, row, ncol, u, scaled;
    };
    auto setup_g = [&](Item& t, int grp, int row, int ti) {
        t.phase = 0;
        t.row = row;
        t.ncol = 2 * ti;
        t.u = 0;
        t.scaled = 0;
        const char* wg = reinterpret_cast<const char*>(p.w13 + (int64_t)grp * p.w13_gs);
#pragma unroll
        for (int r = 0; r < 4; ++r) t.w[r] = wg + ((int64_t)((r >> 1) * I + t.ncol + (r & 1)) * p.w13_ld) * 2;
    };
    auto setup_d = [&](Item& t, int grp, int row, int ti) {
        t.phase = 1;
        t.row = row;
        t.ncol = 4 * ti;
        t.u = 0;
        t.scaled = grp < E ? 1 : 0;
        const char* wg = reinterpret_cast<const char*>(p.w2 + (int64_t)grp * p.w2_gs);
#pragma unroll
        for (int r = 0; r < 4; ++r) t.w[r] = wg + ((int64_t)(t.ncol + r) * p.w2_ld) * 2;
    };
    auto issue_gd = [&](auto bt, const Item& t) {
        constexpr int B = decltype(bt)::value;
        if (t.phase == 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int kk = t.u * EN_KU + (s * 64 + lane) * 8;   // H is a multiple of EN_KU (launcher)
#pragma unroll
                for (int r = 0; r < 4; ++r) wb[B][4 * s + r] = en_wload(t.w[r] + (int64_t)kk * 2);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int kk = (s * 64 + lane) * 8;   // 1024 <= I <= 1536 (launcher): only the third KiB of a row may be short
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    wb[B][4 * s + r] = en_wload(s < 2 || kk < I ? t.w[r] + (int64_t)kk * 2 : zero);
            }
        }
    };
    Item cur, nxt;
    bool have = false;
    const int my_shared = wid * nb + b;           // shared gate|up tile of this wave (tiles 0 .. tpgG-1 of group E)
    const int n_sh = (tpgG + nb - 1) / nb;        // waves 0 .. n_sh-1 may hold one; the rest gather at seam A
    bool shared_issued = false;
    auto prefetch_shared = [&]() {
        if (!shared_issued && my_shared < tpgG) {
            setup_g(cur, E, k, my_shared);
            issue_gd(std::integral_constant<int, 0>{}, cur);  // buffer 0: see the order of the two streams below
            have = true;
        }
        shared_issued = true;
    };

    // o_proj stream: units alternate between the two register buffers.  The shared tile's first unit goes into buffer 0 as soon
    // as nothing of o_proj will be issued into it any more: behind the compute of the last unit that sits there, or — when
    // the last unit sits in buffer 1 — in front of that unit's compute.
    while (c_live) {
        advance_o();
        if (i_live) issue_o(std::integral_constant<int, 1>{}, it, i_u);
        compute_o(std::integral_constant<int, 0>{}, c_u);
        if (!i_live) prefetch_shared();   // buffer 0 has just been consumed, nothing more goes into it
        if (++c_u == nuo) {
            finish_o(ct);
            c_u = 0;
            c_live = i_live;
            ct = it;
        }
        if (!c_live) break;
        advance_o();
        if (i_live) issue_o(std::integral_constant<int, 0>{}, it, i_u);
        else prefetch_shared();           // buffer 0 was consumed one step ago and nothing of o_proj goes into it any more
        compute_o(std::integral_constant<int, 1>{}, c_u);
        if (++c_u == nuo) {
            finish_o(ct);
            c_u = 0;
            c_live = i_live;
            ct = it;
        }
    }
    prefetch_shared();
#if GLM_EN_STAMP
    if (tid == 0 && b < 256) g_en_stamps[b][1] = __builtin_amdgcn_s_memrealtime();
#endif

    // ================================================================ seam A: gather the o_proj row ====================
    if (wid >= n_sh) {
        const int nsw = EN_NW - n_sh, per = nsw * 64;
        const int g0 = (wid - n_sh) * 64 + lane;
        unsigned done = 0u;
        const int cntg = g0 < ntile_h ? (ntile_h - g0 + per - 1) / per : 0;  // <= 32 (checked by the launcher)
        const unsigned want = cntg >= 32 ? 0xffffffffu : ((1u << cntg) - 1u);
        for (unsigned spins = 0;;) {
            for (int i = 0; i < cntg; ++i) {
                if (done >> i & 1u) continue;
                const unsigned long long x = ld_sc1_u64(p.gran_h + g0 + i * per);
                if ((unsigned)(x >> 32) == epoch) {
                    hbuf[g0 + i * per] = (unsigned)x;
                    done |= 1u << i;
                }
            }
            if (__all(done == want)) break;
            __builtin_amdgcn_s_sleep(4);
            if (++spins > EN_SPIN_GLOBAL) {
                if (lane == 0) en_fail(p, flags, 0x100u + (unsigned)b);
                break;
            }
        }
    }
    __syncthreads();
#if GLM_EN_STAMP
    if (tid == 0 && b < 256) g_en_stamps[b][2] = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- post-attention add-RMSNorm, every workgroup redundantly: tinym_router_tail's arithmetic and summation order ----
    {
        const bool act_t = tid < nvec;
        float s8[8], nwf[8];
        float ss = 0.f;
        if (act_t) {
            const u32x4_t xv = reinterpret_cast<const u32x4_t*>(hbuf)[tid];
            const u32x4_t rv = reinterpret_cast<const u32x4_t*>(p.res_in)[tid];
            const u32x4_t wv = reinterpret_cast<const u32x4_t*>(p.norm_w)[tid];
            u32x4_t ov;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s8[2 * j] = bflo(xv[j]) + bflo(rv[j]);
                s8[2 * j + 1] = bfhi(xv[j]) + bfhi(rv[j]);
                ov[j] = pack2bf(s8[2 * j], s8[2 * j + 1]);
                nwf[2 * j] = bflo(wv[j]);
                nwf[2 * j + 1] = bfhi(wv[j]);
            }
            if (b == 0) reinterpret_cast<u32x4_t*>(p.res_out)[tid] = ov;
        }
        if (act_t && tid >= 256) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xch[(tid - 256) * 8 + j] = s8[j];
        }
        __syncthreads();
        if (act_t && tid < 256) {
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
        for (int w = 0; w < EN_NW; ++w) tot += red[w];
        const float rstd = 1.0f / sqrtf(tot / (float)H + p.eps);
        if (act_t) {
            u32x4_t o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = rbf(s8[2 * i] * rstd) * nwf[2 * i];
                const float c = rbf(s8[2 * i + 1] * rstd) * nwf[2 * i + 1];
                o[i] = pack2bf(a, c);
            }
            if (b == 0) reinterpret_cast<u32x4_t*>(p.y_norm)[tid] = o;
            reinterpret_cast<u32x4_t*>(xs)[tid] = o;
        }
    }
    __syncthreads();
#if GLM_EN_STAMP
    if (tid == 0 && b < 256) g_en_stamps[b][3] = __builtin_amdgcn_s_memrealtime();
#endif

    // ================================================================ roles behind the seam ============================
    auto wait_lds = [&](int idx, unsigned code) {
        if (en_lds_read(flags + idx)) return;
        for (unsigned spins = 0;;) {
            __builtin_amdgcn_s_sleep(1);
            if (en_lds_read(flags + idx)) return;
            if (++spins > EN_SPIN_LDS) {
                if (lane == 0) en_fail(p, flags, code + (unsigned)b);
                return;
            }
        }
    };
    if (wid == 6 && b < E) {
        // R1 for expert b: logits = bf16(fp32 dot), router_topk_kernel's FMA chain over a lane's chunks lane, lane + 64, ...
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the gate row (LDS-DMA, requested by this wave) has landed
        const u32x4_t* grow = reinterpret_cast<const u32x4_t*>(smem + L.gate);
        float a = 0.f;
        for (int c = lane; c < nvec; c += 64) {
            const u32x4_t wv = grow[c];
            const u32x4_t xv = reinterpret_cast<const u32x4_t*>(xs)[c];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a = fmaf(bflo(xv[q]), bflo(wv[q]), a);
                a = fmaf(bfhi(xv[q]), bfhi(wv[q]), a);
            }
        }
        const float sdot = wave_sum(a);
        if (lane == 0) st_sc1_u64(p.gran_l + b, etag | __float_as_uint(rbf(sdot)));
    }
    if (wid == 7) {
        // ---- gather the E logits, select (R2..R7) — identically on every workgroup — and publish the routing ----
        {
            unsigned done = 0u;
            const int cntg = (E - lane + 63) / 64 > 0 ? (E - lane + 63) / 64 : 0;
            const unsigned want = (1u << cntg) - 1u;
            for (unsigned spins = 0;;) {
                for (int i = 0; i < cntg; ++i) {
                    if (done >> i & 1u) continue;
                    const unsigned long long x = ld_sc1_u64(p.gran_l + lane + 64 * i);
                    if ((unsigned)(x >> 32) == epoch) {
                        lg[lane + 64 * i] = __uint_as_float((unsigned)x);
                        done |= 1u << i;
                    }
                }
                if (__all(done == want)) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > EN_SPIN_GLOBAL) {
                    if (lane == 0) en_fail(p, flags, 0x200u + (unsigned)b);
                    break;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        router_select_wave<ROUTER_MAX_E / 64>(lg, bias_l, keys, wl, lane, E, k, p.rsf, 0, r_idx, r_w, nullptr, false, r_offs,
                                              r_tok, r_scale, r_pos, 1);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // flat stores into LDS
        if (lane == 0) flags[0] = 1;
        if (b == 0) {  // the next launch (combine + norm + QKV of the next layer) and the caller read these
            for (int i = lane; i < k; i += 64) {
                p.topk_idx[i] = r_idx[i];
                p.topk_w[i] = r_w[i];
                p.pos[i] = r_pos[i];
            }
            for (int i = lane; i < E + 2; i += 64) p.offs[i] = r_offs[i];
            for (int i = lane; i < R; i += 64) {
                p.tok[i] = r_tok[i];
                p.scale[i] = r_scale[i];
            }
            // every workgroup has read the epoch (it published o_proj granules with it): advance it for the next launch
            if (lane == 0) __hip_atomic_store(p.state, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#if GLM_EN_STAMP
        if (lane == 0 && b < 256) g_en_stamps[b][4] = __builtin_amdgcn_s_memrealtime();
#endif
        // ---- seam B: gather the act rows group by group, in production order (shared, then slots 0..k-1) ----
        const int cntg = (tpgG - lane + 63) / 64 > 0 ? (tpgG - lane + 63) / 64 : 0;  // <= 32 (launcher)
        const unsigned want = cntg >= 32 ? 0xffffffffu : ((1u << cntg) - 1u);
        for (int oi = 0; oi < R; ++oi) {
            const int row = oi == 0 ? k : oi - 1;
            const unsigned long long* gsrc = p.gran_a + (int64_t)row * tpgG;
            unsigned* dst = reinterpret_cast<unsigned*>(actl + (int64_t)row * EN_IP);
            unsigned done = 0u;
            for (unsigned spins = 0;;) {
                for (int i = 0; i < cntg; ++i) {
                    if (done >> i & 1u) continue;
                    const unsigned long long x = ld_sc1_u64(gsrc + lane + 64 * i);
                    if ((unsigned)(x >> 32) == epoch) {
                        dst[lane + 64 * i] = (unsigned)x;
                        done |= 1u << i;
                    }
                }
                if (__all(done == want)) break;
                __builtin_amdgcn_s_sleep(8);
                if (++spins > EN_SPIN_GLOBAL) {
                    if (lane == 0) en_fail(p, flags, 0x300u + (unsigned)b);
                    break;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) flags[8 + row] = 1;
        }
#if GLM_EN_STAMP
        if (lane == 0 && b < 256) g_en_stamps[b][5] = __builtin_amdgcn_s_memrealtime();
#endif
    }

    // ================================================================ phases G and D: one stream per wave =================
    int stage = 1;  // 1: routed gate|up tiles, 2: down tiles, 3: none left
    auto pull = [&](Item& t) -> int {  // 1 = t filled, 0 = nothing left, 2 = the routing is not there yet
        if (!en_lds_read(flags + 0)) return 2;
        if (stage == 1) {
            int j = 0;
            if (lane == 0) j = __hip_atomic_fetch_add((int*)&flags[4], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            j = __builtin_amdgcn_readfirstlane(j);
            const int64_t t64 = (int64_t)b + (int64_t)nb * j;
            if (t64 < (int64_t)k * tpgG) {
                const int slot = (int)(t64 / tpgG), ti = (int)(t64 - (int64_t)slot * tpgG);
                setup_g(t, __builtin_amdgcn_readfirstlane((int)r_idx[slot]), slot, ti);
                return 1;
            }
            stage = 2;
        }
        if (stage == 2) {
            int j = 0;
            if (lane == 0) j = __hip_atomic_fetch_add((int*)&flags[5], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            j = __builtin_amdgcn_readfirstlane(j);
            const int64_t t64 = (int64_t)b + (int64_t)nb * j;
            if (t64 < (int64_t)R * tpgD) {
                const int o = (int)(t64 / tpgD), ti = (int)(t64 - (int64_t)o * tpgD);
                const int row = o == 0 ? k : o - 1;
                setup_d(t, o == 0 ? E : __builtin_amdgcn_readfirstlane((int)r_idx[row]), row, ti);
                return 1;
            }
            stage = 3;
        }
        return 0;
    };
    auto compute_gd = [&](auto bt, const Item& t) {
        constexpr int B = decltype(bt)::value;
        if (t.phase == 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const u32x4_t xv = *reinterpret_cast<const u32x4_t*>(xs + t.u * EN_KU + (s * 64 + lane) * 8);
#pragma unroll
                for (int r = 0; r < 4; ++r) en_fma8(acc[r], wb[B][4 * s + r], xv);
            }
        } else {
            const bf16_t* ar = actl + (int64_t)t.row * EN_IP;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const u32x4_t xv = *reinterpret_cast<const u32x4_t*>(ar + (s * 64 + lane) * 8);
#pragma unroll
                for (int r = 0; r < 4; ++r) en_fma8(acc[r], wb[B][4 * s + r], xv);
            }
        }
    };
    auto finish_gd = [&](const Item& t) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o[r] = wave_sum(acc[r]);
            acc[r] = 0.f;
        }
        if (lane != 0) return;
        if (t.phase == 0) {  // act = bf16(bf16(silu(bf16(g))) * bf16(u)), two columns
            const float a0 = rbf(glm_silu(rbf(o[0]))) * rbf(o[2]);
            const float a1 = rbf(glm_silu(rbf(o[1]))) * rbf(o[3]);
            const unsigned v = pack2bf(a0, a1);
            st_sc1_u64(p.gran_a + (int64_t)t.row * tpgG + (t.ncol >> 1), etag | v);
            if (p.act) *reinterpret_cast<unsigned*>(p.act + (int64_t)t.row * I + t.ncol) = v;
        } else {             // y = bf16(bf16(acc) * scale[row]) for routed groups, bf16(acc) for the shared one
            float v[4];
            const float sc = t.scaled ? bf2f(r_scale[t.row]) : 1.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = t.scaled ? rbf(o[r]) * sc : o[r];
            *reinterpret_cast<u32x2_t*>(p.y + (int64_t)t.row * p.ldy + t.ncol) = u32x2_t{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        }
    };
    auto step = [&](auto bt) {
        constexpr int B = decltype(bt)::value;
        int st;
        if (cur.phase == 0 && cur.u + 1 < nug) {
            nxt = cur;
            nxt.u = cur.u + 1;
            st = 1;
        } else {
            st = pull(nxt);
        }
        if (st == 1) issue_gd(std::integral_constant<int, 1 - B>{}, nxt);
        if (cur.phase == 1) wait_lds(8 + cur.row, 0x400u);
        compute_gd(bt, cur);
        if (cur.phase == 1 || cur.u == nug - 1) finish_gd(cur);
        if (st == 2) {
            wait_lds(0, 0x500u);
            st = pull(nxt);
            if (st == 2) st = 0;  // the bounded wait expired: give up this wave's stream (status is set)
            if (st == 1) issue_gd(std::integral_constant<int, 1 - B>{}, nxt);
        }
        have = st == 1;
        cur = nxt;
    };
    if (!have) {
        wait_lds(0, 0x500u);
        const int st = pull(cur);
        have = st == 1;
        if (have) issue_gd(std::integral_constant<int, 0>{}, cur);
    }
    while (have) {
        step(std::integral_constant<int, 0>{});
        if (!have) break;
        step(std::integral_constant<int, 1>{});
    }
#if GLM_EN_STAMP
    __syncthreads();
    if (tid == 0 && b < 256) g_en_stamps[b][6] = __builtin_amdgcn_s_memrealtime();
#endif
    if (en_lds_read(flags + 1)) {  // a bounded wait of this workgroup expired: never let its rows pass for a result
        for (int64_t i = lane; i < (int64_t)R * (H / 2); i += 64) {
            const int64_t r = i / (H / 2), c = i - r * (H / 2);
            reinterpret_cast<unsigned*>(p.y + r * p.ldy)[c] = 0x7fc07fc0u;
        }
    }
}

int en_n_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return n;
}

}  // namespace

// bytes of the engine's workspace: [16 B state][H/2 + E + (k+1) I/2 granules of 8 B]; zero before the first launch
extern "C" size_t glm_decode_mlp_engine_workspace_bytes(int H, int E, int k, int I) {
    if (H <= 0 || E <= 0 || k <= 0 || I <= 0) return 0;
    return 16 + 8 * ((size_t)H / 2 + (size_t)E + (size_t)(k + 1) * ((size_t)I / 2));
}

extern "C" int glm_decode_mlp_engine(const void* x, const void* w_o, int64_t ldw_o, int K, int H, const void* residual_in,
                                     const void* norm_weight, float eps, void* y_norm, void* residual_out, const void* gate_w,
                                     const float* e_score_correction_bias, int E, int k, float routed_scaling_factor,
                                     int64_t* topk_idx, void* topk_w, int32_t* expert_offsets, int32_t* sorted_token,
                                     void* sorted_scale, int32_t* pair_pos, const void* w13, int64_t w13_group_stride,
                                     int64_t w13_row_stride, const void* w2, int64_t w2_group_stride, int64_t w2_row_stride,
                                     int I, void* act, void* y, int64_t ldy, void* workspace, glm_stream_t stream) {
    GLM_CHECK_ARG(x && w_o && residual_in && norm_weight && y_norm && residual_out && gate_w && e_score_correction_bias &&
                      topk_idx && topk_w && expert_offsets && sorted_token && sorted_scale && pair_pos && w13 && w2 && y && workspace,
                  GLM_ERR_INVALID, "glm_decode_mlp_engine: null pointer");
    GLM_CHECK_ARG(K > 0 && H > 0 && ldw_o >= K && E > 0 && k > 0 && k <= E && I > 0 && w13_row_stride >= H && w2_row_stride >= I &&
                      ldy >= H, GLM_ERR_INVALID, "glm_decode_mlp_engine: bad shape K=%d H=%d E=%d k=%d I=%d", K, H, E, k, I);
    const int nb = en_n_cus();
    const EnLds L(K, H, k);
    const int tpgG = I / 2, nsh = (tpgG + nb - 1) / nb;
    const bool ok = H % EN_KU == 0 && H >= 2048 && H <= 4096 && K % EN_KU == 0 && I % 8 == 0 && I >= 1024 && I <= EN_IP && E <= ROUTER_MAX_E &&
                    k <= 64 && E <= nb && nsh <= 5 && (H / 2 + (EN_NW - nsh) * 64 - 1) / ((EN_NW - nsh) * 64) <= 32 &&
                    (tpgG + 63) / 64 <= 32 && L.total <= 156 * 1024 && ldw_o % 8 == 0 && w13_row_stride % 8 == 0 && w2_row_stride % 8 == 0 &&
                    w13_group_stride % 8 == 0 && w2_group_stride % 8 == 0 && ldy % 4 == 0 && glm_aligned16(x) && glm_aligned16(w_o) &&
                    glm_aligned16(residual_in) && glm_aligned16(norm_weight) && glm_aligned16(y_norm) && glm_aligned16(residual_out) &&
                    glm_aligned16(gate_w) && glm_aligned16(w13) && glm_aligned16(w2) && (((uintptr_t)y) & 7) == 0 &&
                    (act == nullptr || (((uintptr_t)act) & 3) == 0) && glm_aligned16(workspace);
    if (!ok) {
        glm_set_error("glm_decode_mlp_engine: shape not covered (K=%d H=%d E=%d k=%d I=%d, %d CUs, %d B LDS)", K, H, E, k, I, nb, L.total);
        return GLM_ERR_UNSUPPORTED;
    }
    EngineParams p;
    memset(&p, 0, sizeof(p));
    p.x_o = (const bf16_t*)x; p.w_o = (const bf16_t*)w_o; p.ldw_o = ldw_o; p.Ko = K; p.H = H;
    p.res_in = (const bf16_t*)residual_in; p.norm_w = (const bf16_t*)norm_weight; p.eps = eps;
    p.y_norm = (bf16_t*)y_norm; p.res_out = (bf16_t*)residual_out;
    p.gate_w = (const bf16_t*)gate_w; p.bias = e_score_correction_bias; p.E = E; p.k = k; p.rsf = routed_scaling_factor;
    p.topk_idx = topk_idx; p.topk_w = (bf16_t*)topk_w; p.offs = expert_offsets; p.tok = sorted_token;
    p.scale = (bf16_t*)sorted_scale; p.pos = pair_pos;
    p.w13 = (const bf16_t*)w13; p.w13_gs = w13_group_stride; p.w13_ld = w13_row_stride;
    p.w2 = (const bf16_t*)w2; p.w2_gs = w2_group_stride; p.w2_ld = w2_row_stride; p.I = I;
    p.act = (bf16_t*)act; p.y = (bf16_t*)y; p.ldy = ldy;
    char* ws = (char*)workspace;
    p.state = (unsigned*)ws;
    p.gran_h = (unsigned long long*)(ws + 16);
    p.gran_l = p.gran_h + H / 2;
    p.gran_a = p.gran_l + E;
    size_t lds = (size_t)L.total;
    if (lds < 84 * 1024) lds = 84 * 1024;  // exactly one workgroup per CU: every workgroup of the grid is resident
    GLM_SET_MAX_LDS(decode_mlp_engine_kernel, 156 * 1024, "glm_decode_mlp_engine");
    hipLaunchKernelGGL(decode_mlp_engine_kernel, dim3((unsigned)nb), dim3(EN_NT), lds, glm_s(stream), p);
    GLM_CHECK_LAUNCH("glm_decode_mlp_engine");
    return GLM_OK;
}

#if GLM_EN_STAMP
extern "C" int glm_debug_engine_stamps(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_en_stamps), sizeof(unsigned long long) * 8 * (n < 256 ? n : 256)) == hipSuccess ? 0 : -1;
}
#endif
