// gemm_largem.hip — LDS-tiled bf16 MFMA GEMM for the prefill regime (many rows per weight
// matrix): dense QKV/O/MLP projections and the grouped expert projections at T >> E.  gfx950.
//
// Roofline: MFMA (bf16 dense peak 2.5 PFLOP/s).  At 2048 rows per expert the grouped gate|up
// GEMM has ~700 FLOP per weight byte.
//   * out^T = W · x^T: weight rows on the MFMA row axis (A), tokens on the column axis (B), so a
//     lane's 4 accumulator registers are 4 consecutive output columns of one token (8-B stores)
//     and gate/up sit in the same lanes for the fused SiLU·mul epilogue.
//   * 128(n) x 128(m) x 64(k) workgroup tile, 4 waves as 2x2, each wave 64x64 = 4x4 MFMA tiles
//     of mfma_f32_16x16x32_bf16 (64 accumulator registers).
//   * both operand tiles are staged by LDS-DMA (global_load_lds_dwordx4: 8 rows x 128 B per
//     instruction); the LDS image is lane-linear, bank conflicts of the 16-row fragment reads are
//     removed by permuting the SOURCE chunk: chunk' = chunk ^ ((row >> 1) & 7) (rows are 128 B =
//     half a bank row, so (row & 1, chunk') enumerates all 16 slots of a bank row).
//   * token rows of the B operand are gathered through sorted_token (per-lane source address
//     of the DMA), so the grouped GEMM needs no pre-gathered activation copy.
//   * double-buffered LDS, next tile's DMA issued before the current tile's MFMAs, one
//     vmcnt(0)+barrier per K tile; fragment reads are inline asm so hipcc does not drain the
//     DMA before them (same reasoning as paged_attn.hip).
//   * 1-D grid, n-tile fastest inside a (group, 128-row block), remapped so that one XCD gets a
//     contiguous run of tiles (neighbours share the activation panel in that XCD's L2).
#include "gemm_internal.h"

namespace {

constexpr int LM_BM = 128, LM_BN = 128, LM_BK = 64;
constexpr int LM_TILE_BYTES = 128 * LM_BK * 2;          // 16 KiB per operand tile
constexpr int LM_STAGE = 2 * LM_TILE_BYTES;             // W tile + X tile
constexpr int LM_LDS = 2 * LM_STAGE + 16;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

__device__ const u32x4_t g_zero_page_lm[4] = {};

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void_t*)gsrc, (lds_void_t*)lds_wave_base, 16, 0, 0);
}
template <int OFF>
__device__ __forceinline__ void lds_read_b128(u32x4_t& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(const lds_void_t*)p;
}

template <int MODE>
__global__ __launch_bounds__(256) void gemm_largem_kernel(const GemmParams p, int n_tiles, int total_wg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* sh = reinterpret_cast<int*>(smem + 2 * LM_STAGE);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid & 1, wm = wid >> 1;
    const int c = lane & 15, g4 = lane >> 4;

    // XCD-aware remap (bijective for any grid size): blocks with equal id%8 share an XCD
    int bid = blockIdx.x;
    {
        const int q = total_wg >> 3, r = total_wg & 7, x = bid & 7, i = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int ntile = bid % n_tiles;
    const int combo = bid / n_tiles;
    glm_locate_block(p.offs, p.G, p.R, LM_BM, combo, sh);
    __syncthreads();
    const int grp = sh[0];
    if (grp < 0) return;
    const int m0 = sh[1], mcnt = sh[2];
    const int K = p.K;
    const int nk = (K + LM_BK - 1) / LM_BK;

    // ---- DMA source rows: instruction j of a wave covers tile rows 8*(4*j' ...) ----
    // each operand tile = 128 rows x 8 chunks = 16 instructions; wave w issues 4 for W and 4 for X:
    // instruction i (0..3) of wave w covers rows (w*4 + i)*8 .. +7; lane → row +(lane>>3), pos lane&7
    const bf16_t* wsrc[4];
    const bf16_t* xsrc[4];
    int chunk_of[4];  // global chunk (0..7) this lane fetches, per instruction
    const bf16_t* wg = p.w + (int64_t)grp * p.w_gstride;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wid * 4 + i) * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ ((row >> 1) & 7);
        chunk_of[i] = ch;
        int wr;
        if (MODE == GEMM_GATEUP) {
            const int half = row >> 6;                          // 0: gate rows, 1: up rows
            const int col = min(ntile * 64 + (row & 63), p.N - 1);
            wr = half * p.N + col;
        } else {
            wr = min(ntile * LM_BN + row, p.N - 1);
        }
        wsrc[i] = wg + (int64_t)wr * p.ldw + ch * 8;
        const int mr = row < mcnt ? row : 0;
        const int64_t srow = p.src ? (int64_t)p.src[m0 + mr] : (int64_t)(m0 + mr);
        xsrc[i] = p.x + srow * p.ldx + ch * 8;
    }
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_lm);
    auto stage = [&](int buf, int kt) {
        char* wdst = smem + buf * LM_STAGE + wid * 4 * 1024;
        char* xdst = wdst + LM_TILE_BYTES;
        const int k0 = kt * LM_BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = k0 + chunk_of[i] * 8 < K;
            glds16(ok ? wsrc[i] + k0 : zero, wdst + i * 1024);
            glds16(ok ? xsrc[i] + k0 : zero, xdst + i * 1024);
        }
    };

    // ---- fragment read offsets (relative to the stage base) ----
    // W rows of this wave's n-tile nt; X rows of m-tile mt; lane (c,g4) reads row +c, chunk 4ks+g4
    unsigned wrel[4][2], xrel[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int wrow;
        if (MODE == GEMM_GATEUP) wrow = (t < 2 ? wn * 32 + 16 * t : 64 + wn * 32 + 16 * (t - 2)) + c;
        else wrow = wn * 64 + 16 * t + c;
        const int xrow = wm * 64 + 16 * t + c;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            wrel[t][ks] = wrow * 128 + (((4 * ks + g4) ^ ((wrow >> 1) & 7)) << 4);
            xrel[t][ks] = LM_TILE_BYTES + xrow * 128 + (((4 * ks + g4) ^ ((xrow >> 1) & 7)) << 4);
        }
    }

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const unsigned sb = lds_addr(smem + cur * LM_STAGE);
        u32x4_t wf[4][2], xf[4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                lds_read_b128<0>(wf[t][ks], sb + wrel[t][ks]);
                lds_read_b128<0>(xf[t][ks], sb + xrel[t][ks]);
            }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(wf[0][0]), "+v"(wf[0][1]), "+v"(wf[1][0]), "+v"(wf[1][1]), "+v"(wf[2][0]),
                       "+v"(wf[2][1]), "+v"(wf[3][0]), "+v"(wf[3][1]), "+v"(xf[0][0]), "+v"(xf[0][1]),
                       "+v"(xf[1][0]), "+v"(xf[1][1]), "+v"(xf[2][0]), "+v"(xf[2][1]), "+v"(xf[3][0]),
                       "+v"(xf[3][1])
                     :
                     : "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8_t, wf[nt][ks]), __builtin_bit_cast(bf16x8_t, xf[mt][ks]),
                        acc[nt][mt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue ----
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int row = wm * 64 + 16 * mt + c;
        if (row >= mcnt) continue;
        bf16_t* yrow = p.y + (int64_t)(m0 + row) * p.ldy;
        if (MODE == GEMM_GATEUP) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int n = ntile * 64 + wn * 32 + 16 * t + 4 * g4;
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
    const int cols_per_wg = MODE == GEMM_GATEUP ? 64 : LM_BN;
    const int n_tiles = (p.N + cols_per_wg - 1) / cols_per_wg;
    const int64_t combos = p.offs ? (p.R / LM_BM + p.G) : (p.R + LM_BM - 1) / LM_BM;
    const int64_t total = combos * n_tiles;
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
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), LM_LDS, st, p, n_tiles, (int)total);
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
