// gemm_largem.hip — LDS-tiled bf16 MFMA GEMM for the prefill regime (many rows per weight
// matrix): dense QKV/O/MLP projections and the grouped expert projections at T >> E.  gfx950.
//
// Roofline: MFMA (bf16 dense peak 2.5 PFLOP/s).  At 2048 rows per expert the grouped gate|up
// GEMM has ~700 FLOP per weight byte.
//   * out^T = W · x^T: weight rows on the MFMA row axis (A), tokens on the column axis (B), so a
//     lane's 4 accumulator registers are 4 consecutive output columns of one token (8-B stores)
//     and gate/up sit in the same lanes for the fused SiLU·mul epilogue.
//   * 256(n) x 128(m) x 64(k) workgroup tile, 8 waves as 4(n) x 2(m), each wave 64x64 = 4x4 MFMA
//     tiles of mfma_f32_16x16x32_bf16 (64 accumulator registers); two waves per SIMD alternate
//     between their LDS-read and MFMA phases.
//   * both operand tiles are staged by LDS-DMA (global_load_lds_dwordx4: 8 rows x 128 B per
//     instruction, 6 instructions per wave per K tile); the LDS image is lane-linear, bank
//     conflicts of the 16-row fragment reads are removed by permuting the SOURCE chunk:
//     chunk' = chunk ^ ((row >> 1) & 7) (rows are 128 B = half a bank row).
//   * token rows of the B operand are gathered through sorted_token (per-lane source address of
//     the DMA), so the grouped GEMM needs no pre-gathered activation copy.
//   * 3-stage LDS ring (3 x 48 KiB), prefetch distance 2, raw s_barriers and a COUNTED
//     s_waitcnt vmcnt(6): the DMA of the newest tile stays in flight across the barriers.
//     Fragment reads are inline asm (hipcc would drain the DMA before them).
//   * two phases per K tile (loads | MFMAs) and waves 4-7 staggered one barrier behind waves 0-3,
//     so the two waves of each SIMD alternate between feeding the matrix pipe and issuing loads
//     (all 8 waves in lockstep left the pipe idle ~60 % of the cycles: 830 TFLOP/s).
//   * 1-D grid remapped so that one XCD gets a contiguous run of tiles, ordered in bands of 8 row
//     blocks x all n-tiles (n-tile major): the workgroups co-resident on an XCD form a supertile
//     whose operand panels are shared through that XCD's L2.
#include "gemm_internal.h"

namespace {

constexpr int LM_BM = 128;                               // token rows per tile (MFMA column axis)
constexpr int LM_BN = 256;                               // weight rows per tile (MFMA row axis)
constexpr int LM_BK = 64;
constexpr int LM_BAND = 8;                               // row blocks per L2 supertile band
constexpr int LM_WT = LM_BN * LM_BK * 2;                 // 32 KiB weight tile
constexpr int LM_XT = LM_BM * LM_BK * 2;                 // 16 KiB activation tile
constexpr int LM_STAGE = LM_WT + LM_XT;                  // 48 KiB
constexpr int LM_NSTAGE = 3;
constexpr int LM_LDS = LM_NSTAGE * LM_STAGE + 16;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

__device__ const u32x4_t g_zero_page_lm[4] = {};

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void_t*)gsrc, (lds_void_t*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void lds_read_b128(u32x4_t& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(const lds_void_t*)p;
}

#define LM_WAIT8(a)                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                               \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), \
                   "+v"(a[7])                                                                         \
                 :                                                                                    \
                 : "memory")

template <int MODE>
__global__ __launch_bounds__(512) void gemm_largem_kernel(const GemmParams p, int n_tiles, int total_wg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* sh = reinterpret_cast<int*>(smem + LM_NSTAGE * LM_STAGE);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid & 3, wm = wid >> 2;
    const int c = lane & 15, g4 = lane >> 4;

    // XCD-aware remap (bijective for any grid size): blocks with equal id%8 share an XCD
    int bid = blockIdx.x;
    {
        const int q = total_wg >> 3, r = total_wg & 7, x = bid & 7, i = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int band = bid / (LM_BAND * n_tiles);
    const int rem = bid % (LM_BAND * n_tiles);
    const int ntile = rem / LM_BAND;
    const int combo = band * LM_BAND + rem % LM_BAND;
    glm_locate_block(p.offs, p.G, p.R, LM_BM, combo, sh);
    __syncthreads();
    const int grp = sh[0];
    if (grp < 0) return;
    const int m0 = sh[1], mcnt = sh[2];
    const int K = p.K;
    const int nk = (K + LM_BK - 1) / LM_BK;

    // ---- DMA sources.  One instruction = 8 tile rows x 8 chunks; lane → row +(lane>>3), LDS
    // position lane&7 holds global chunk (lane&7) ^ ((row>>1)&7).  Wave w issues weight
    // instructions 4w..4w+3 (rows 32w..32w+31) and activation instructions 2w, 2w+1. ----
    const bf16_t* wsrc[4];
    const bf16_t* xsrc[2];
    int wch[4], xch[2];
    const bf16_t* wg = p.w + (int64_t)grp * p.w_gstride;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wid * 4 + i) * 8 + (lane >> 3);  // 0..255
        const int ch = (lane & 7) ^ ((row >> 1) & 7);
        wch[i] = ch;
        int wr;
        if (MODE == GEMM_GATEUP) {
            const int half = row >> 7;  // 0: gate rows, 1: up rows
            wr = half * p.N + min(ntile * 128 + (row & 127), p.N - 1);
        } else {
            wr = min(ntile * LM_BN + row, p.N - 1);
        }
        wsrc[i] = wg + (int64_t)wr * p.ldw + ch * 8;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wid * 2 + i) * 8 + (lane >> 3);  // 0..127
        const int ch = (lane & 7) ^ ((row >> 1) & 7);
        xch[i] = ch;
        const int mr = row < mcnt ? row : 0;
        const int64_t srow = p.src ? (int64_t)p.src[m0 + mr] : (int64_t)(m0 + mr);
        xsrc[i] = p.x + srow * p.ldx + ch * 8;
    }
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_lm);
    auto stage = [&](int slot, int kt) {
        char* wdst = smem + slot * LM_STAGE + wid * 4 * 1024;
        char* xdst = smem + slot * LM_STAGE + LM_WT + wid * 2 * 1024;
        const int k0 = kt * LM_BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(k0 + wch[i] * 8 < K ? wsrc[i] + k0 : zero, wdst + i * 1024);
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(k0 + xch[i] * 8 < K ? xsrc[i] + k0 : zero, xdst + i * 1024);
    };

    // ---- fragment read offsets relative to the stage base: lane (c,g4) reads row +c, chunk 4ks+g4
    unsigned wrel[2][4], xrel[2][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int wrow;
        if (MODE == GEMM_GATEUP) wrow = (t < 2 ? wn * 32 + 16 * t : 128 + wn * 32 + 16 * (t - 2)) + c;
        else wrow = wn * 64 + 16 * t + c;
        const int xrow = wm * 64 + 16 * t + c;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            wrel[ks][t] = wrow * 128 + (((4 * ks + g4) ^ ((wrow >> 1) & 7)) << 4);
            xrel[ks][t] = LM_WT + xrow * 128 + (((4 * ks + g4) ^ ((xrow >> 1) & 7)) << 4);
        }
    }

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- main loop: two phases per K tile, waves 4-7 run ONE BARRIER BEHIND waves 0-3 ----
    //   phase 1: LDS fragment reads of tile kt (16 x b128) + DMA issue of tile kt+2 + counted
    //            wait for this wave's share of tile kt+1, then barrier
    //   phase 2: 32 MFMAs, then barrier
    // The two waves that share a SIMD (w, w+4) are therefore always in opposite phases: one feeds
    // the matrix pipe while the other issues loads (in lockstep the pipe idles ~60 % of the time).
    // Ordering under the stagger (X' = the other group's program point at the same hardware barrier):
    //   RAW  tile kt+1 is read in phase 1 of kt+1, i.e. after this group's B2(kt) = other group's
    //        B1(kt) (or later); every wave waited for its share of kt+1 BEFORE its own B1(kt).
    //   WAR  tile kt+2 overwrites the slot of kt-1 in phase 1 of kt, i.e. after this group's B2(kt-1)
    //        >= other group's B1(kt-1); every wave finished reading kt-1 before its own B1(kt-1).
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // every share of tile 0 landed
    if (wid >= 4) __builtin_amdgcn_s_barrier();   // the stagger
    int slot = 0;  // ring slot of tile kt
    const unsigned base = lds_addr(smem);
    for (int kt = 0; kt < nk; ++kt) {
        // ---------------- phase 1 ----------------
        const unsigned sb = base + slot * LM_STAGE;
        u32x4_t f0[8], f1[8];  // [0..3] weight fragments of n-tiles, [4..7] activation fragments
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            lds_read_b128(f0[t], sb + wrel[0][t]);
            lds_read_b128(f0[4 + t], sb + xrel[0][t]);
        }
        if (kt + 2 < nk) stage(slot == 0 ? 2 : slot - 1, kt + 2);  // into the slot tile kt-1 used
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            lds_read_b128(f1[t], sb + wrel[1][t]);
            lds_read_b128(f1[4 + t], sb + xrel[1][t]);
        }
        // this wave's share of tile kt+1 (everything older than the 6 instructions just issued)
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LM_WAIT8(f0);
        LM_WAIT8(f1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();  // B1
        // ---------------- phase 2 ----------------
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8_t, f0[nt]), __builtin_bit_cast(bf16x8_t, f0[4 + mt]), acc[nt][mt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8_t, f1[nt]), __builtin_bit_cast(bf16x8_t, f1[4 + mt]), acc[nt][mt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();  // B2
        slot = slot == 2 ? 0 : slot + 1;
    }
    if (wid < 4) __builtin_amdgcn_s_barrier();  // match the other group's extra barrier

    // ---- epilogue ----
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int row = wm * 64 + 16 * mt + c;
        if (row >= mcnt) continue;
        bf16_t* yrow = p.y + (int64_t)(m0 + row) * p.ldy;
        if (MODE == GEMM_GATEUP) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int n = ntile * 128 + wn * 32 + 16 * t + 4 * g4;
                if (n >= p.N) continue;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    o[r] = rbf(glm_silu(rbf(acc[t][mt][r]))) * rbf(acc[t + 2][mt][r]);
                u32x2_t v = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
                *reinterpret_cast<u32x2_t*>(yrow + n) = v;
            }
        } else {
            float sc = 1.0f;
            bool scaled = false;
            if (MODE == GEMM_DOWN && grp < p.n_scaled) {
                sc = bf2f(p.scale[m0 + row]);
                scaled = true;
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int n = ntile * LM_BN + wn * 64 + 16 * nt + 4 * g4;
                if (n >= p.N) continue;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float a = acc[nt][mt][r];
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

template <int MODE>
int launch_largem(const GemmParams& p, hipStream_t st) {
    const int cols_per_wg = MODE == GEMM_GATEUP ? 128 : LM_BN;
    const int n_tiles = (p.N + cols_per_wg - 1) / cols_per_wg;
    const int64_t combos = p.offs ? (p.R / LM_BM + p.G) : (p.R + LM_BM - 1) / LM_BM;
    const int64_t total = (combos + LM_BAND - 1) / LM_BAND * LM_BAND * n_tiles;
    if (total <= 0) return GLM_OK;
    GLM_CHECK_ARG(total < (1ll << 31), GLM_ERR_UNSUPPORTED, "gemm_largem: grid too large");
    auto kern = gemm_largem_kernel<MODE>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LM_LDS);
        if (e != hipSuccess) {
            glm_set_error("gemm_largem: hipFuncSetAttribute(%d): %s", LM_LDS, hipGetErrorString(e));
            return GLM_ERR_LAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(512), LM_LDS, st, p, n_tiles, (int)total);
    GLM_CHECK_LAUNCH("gemm_largem");
    return GLM_OK;
}

}  // namespace

int glm_launch_gemm_largem(int mode, const GemmParams& p, hipStream_t st) {
    GLM_CHECK_ARG(p.K % 8 == 0 && p.N % 4 == 0 && p.ldx % 8 == 0 && p.ldy % 4 == 0 && p.ldw % 8 == 0 &&
                      glm_aligned16(p.x) && glm_aligned16(p.w) && (((uintptr_t)p.y) & 7) == 0,
                  GLM_ERR_UNSUPPORTED,
                  "gemm_largem: needs K%%8==0, N%%4==0, ldx%%8==0, ldy%%4==0, 16B-aligned x/W (K=%d N=%d)",
                  p.K, p.N);
    if (mode == GEMM_GATEUP) return launch_largem<GEMM_GATEUP>(p, st);
    if (mode == GEMM_DOWN) return launch_largem<GEMM_DOWN>(p, st);
    return launch_largem<GEMM_PLAIN>(p, st);
}
