// gemm_largem.hip — LDS-tiled bf16 MFMA GEMM for the prefill regime (many rows per weight
// matrix): dense QKV/O/MLP projections and the grouped expert projections at T >> E.  gfx950.
//
// Roofline: MFMA (bf16 dense peak 2.5 PFLOP/s).  At 2048 rows per expert the grouped gate|up
// GEMM has ~700 FLOP per weight byte.
//   * out^T = W · x^T: weight rows on the MFMA row axis (A), tokens on the column axis (B), so a
//     lane's 4 accumulator registers are 4 consecutive output columns of one token (8-B stores)
//     and gate/up sit in the same lanes for the fused SiLU·mul epilogue.
//   * 8 waves as 4(n) x 2(m), wave tile 64(n) x 128(m).  Default for large groups (CfgC, r02): 256(n) x 256(m) x 64(k) tiles
//     on TWO 64-KiB LDS buffers, a K tile = four phases of (fragment reads + 2 DMA | barrier | 16 MFMAs | barrier); an
//     operand's slot is refilled as soon as its own last fragment read has retired (argument above the loop).  Measured at
//     T=16384 in one process: 1255 / 1058 TFLOP/s (gate|up / down) against 1081 / 923 for the loop it replaced (CfgB:
//     256 x 256 x 32, 4-stage ring, two 32-MFMA phases per K tile — still variant 3), all bit-identical.  Smaller groups
//     (< 192 rows on average): CfgA, 256 x 128 x 64 with a 3-stage ring (wave tile 64x64).
//   * both operand tiles are staged by LDS-DMA (global_load_lds_dwordx4, 1 KiB per instruction);
//     the LDS image is lane-linear, bank conflicts of the 16-row ds_read_b128 fragment reads are
//     removed by permuting the SOURCE chunk: chunk' = chunk ^ ((row >> 1) & 7) for 128-B rows,
//     chunk ^ ((row >> 1) & 2) for 64-B rows (found by enumerating the b128 lane groups).
//   * token rows of the B operand are gathered through sorted_token (per-lane source address of
//     the DMA), so the grouped GEMM needs no pre-gathered activation copy.
//   * LDS ring with prefetch distance NSTAGE-1, raw s_barriers and COUNTED s_waitcnt vmcnt: the
//     DMA of the newest tiles stays in flight across the barriers.  Fragment reads are inline asm
//     (hipcc would drain the DMA before them).
//   * load and MFMA intervals alternate and waves 4-7 are staggered one barrier behind waves 0-3,
//     so the two waves of each SIMD alternate between feeding the matrix pipe and issuing loads
//     (all 8 waves in lockstep left the pipe idle ~60 % of the cycles: 830 TFLOP/s).
//   * 1-D grid remapped so that one XCD gets a contiguous run of tiles, ordered in bands of 4 row
//     blocks x all n-tiles (n-tile major): the workgroups co-resident on an XCD form a supertile
//     whose operand panels are shared through that XCD's L2.
#include "gemm_internal.h"
#include <stdlib.h>
#include <type_traits>


namespace {

constexpr int LM_BN = 256;    // weight rows per tile (MFMA row axis); 4 waves x 64 rows
#ifndef GLM_LM_BAND
#define GLM_LM_BAND 4
#endif
#ifndef GLM_LM_STAGE_POS
#define GLM_LM_STAGE_POS 0  // 0: DMA issue before the fragment reads (measured best, +0.8 %), 1: between, 2: after
#endif
#ifndef GLM_LM_DEFAULT_VAR
#define GLM_LM_DEFAULT_VAR -1  // < 0: per mode (launch_largem); a value: that variant for every mode
#endif
#ifndef GLM_LM_STUDY
#define GLM_LM_STUDY 0  // 1: also build the measured loop variants + glm_debug_largem_* (build.py --variant _study GLM_LM_STUDY=1)
#endif
#ifndef GLM_LM_SCHED
#define GLM_LM_SCHED 0  // 0: two barriers per K tile with waves 4-7 one barrier behind; 1: one barrier per K tile
#endif
int g_largem_band = GLM_LM_BAND;  // row blocks per L2 supertile band (glm_debug_largem_band for sweeps)

// Tile configurations (the kernel is written once over these):
//   CfgA 256(n) x 128(m) x 64(k), 3-stage ring, wave tile 64x64  (16 LDS reads + 6 DMA per 32 MFMA)
//   CfgB 256(n) x 256(m) x 32(k), 4-stage ring, wave tile 64x128 (12 LDS reads + 4 DMA per 32 MFMA)
template <int BM_, int BK_, int NSTAGE_, bool PH8_ = false, bool W8_ = false, bool A8_ = false>
struct LmCfg {
    static constexpr int BM = BM_, BK = BK_, NSTAGE = NSTAGE_;
    static constexpr bool PH8 = PH8_;              // quadrant-phase main loop (CfgC)
    // A8 (r05): BOTH operands are OCP fp8 e4m3 in memory and in LDS, BK = 128 elements = 128-byte tile rows — the LDS image,
    // the DMA pieces, the swizzle and the two ds_read_b128 per 16-row tile are byte for byte those of the bf16 CfgC (chunk g4
    // and chunk 4 + g4 of a row are exactly the two register halves of a v_mfma_scale_f32_16x16x128_f8f6f4 operand:
    // profiles/r05/mfma_f8f6f4_probe_v2.txt); a quadrant is 8 block-scaled MFMAs of K = 128 instead of 16 of K = 32.
    static constexpr bool A8 = A8_;
    static_assert(!A8 || (BK_ == 128 && !W8_), "fp8 x fp8: 128-byte rows of 128 elements");
    // W8: the weight operand is OCP fp8 e4m3 in memory AND in LDS (half the DMA instructions, half the LDS bytes and
    // fragment-read bytes of the weight tile); a fragment is converted to bf16 (exactly: every e4m3 value is a bf16
    // value) between its ds_read and the MFMA, and the per-row dequantisation scale multiplies the fp32 accumulator in the
    // epilogue (WSCALE) — the arithmetic of the expansion pass + bf16 kernel it replaces, bit for bit.
    static constexpr bool W8 = W8_;
    static constexpr int RB = A8 ? BK : BK * 2;    // bytes per ACTIVATION tile row (and per weight tile row unless W8)
    static constexpr int CPR = RB / 16;            // 16-B chunks per row
    static constexpr int RPI = 1024 / RB;          // rows per DMA instruction
    static constexpr int WRB = W8 ? BK : RB;       // bytes per WEIGHT tile row
    static constexpr int WCPR = WRB / 16;
    static constexpr int WRPI = 1024 / WRB;
    static_assert(!W8 || WRB == 64, "fp8 weight tiles: 64-byte rows (BK = 64)");
    static constexpr int W_DMA = LM_BN / WRPI / 8; // weight DMA instructions per wave per tile
    static constexpr int X_DMA = BM / RPI / 8;     // activation DMA instructions per wave per tile
    static constexpr int DPW = W_DMA + X_DMA;
    static constexpr int KS = A8 ? 2 : BK / 32;    // MFMA k-steps per tile (A8: the two 16-byte halves of ONE K = 128 operand)
    static constexpr int MT = BM / 2 / 16;         // 16-token MFMA tiles per wave (2 waves along m)
    static constexpr int WT = LM_BN * WRB;         // weight tile bytes
    static constexpr int XT = BM * RB;
    static constexpr int STAGE = WT + XT;
    // the 16 bytes of the tile search normally follow the ring; a ring that takes the CU's whole LDS (5 x 32 KiB) lends
    // them the tail of its LAST slot, which no DMA targets between the end of a tile's K loop and the next tile's start
    static constexpr bool ALIAS_SH = NSTAGE * STAGE + 16 > 160 * 1024;
    static constexpr int LDS = ALIAS_SH ? NSTAGE * STAGE : NSTAGE * STAGE + 16 + 64 * 16;  // + tile search + tile table
    static constexpr int DIST = NSTAGE - 1;        // prefetch distance in tiles
    static constexpr int NF = KS * (4 + MT);       // fragment registers (u32x4) per tile
    // source-chunk permutation that makes the 16-row ds_read_b128 fragment reads conflict free
    __device__ static constexpr int swz(int row) { return RB == 128 ? ((row >> 1) & 7) : ((row >> 1) & 2); }
    // fp8 weight rows are 64 B and a fragment read is a ds_read_b64 at row*64 + 16*slot + 8*(g4 & 1): the 32 lanes of a
    // half wave are rows c = 0..15 x two 8-byte halves; rows c, c+4, c+8, c+12 share their banks unless their slots
    // differ, so slot = chunk ^ ((row >> 2) & 3) — each half wave then covers all 64 banks exactly once
    __device__ static constexpr int wswz(int row) { return W8 ? ((row >> 2) & 3) : swz(row); }
};
using LmCfgA = LmCfg<128, 64, 3>;
using LmCfgB = LmCfg<256, 32, 4>;
using LmCfgB5 = LmCfg<256, 32, 5>;  // variant LM_DEEP: prefetch distance 4
//   CfgC 256(n) x 256(m) x 64(k), TWO 64 KiB buffers, wave tile 64x128, four phases of 16 MFMAs per K tile (variant
//        LM_PH8).  128-byte tile rows: one DMA instruction moves 8 whole cache lines (CfgB: 16 half lines).  Two
//        buffers are enough because an operand's slot is refilled as soon as ITS last fragment read has retired, not
//        when the whole K tile is done: the weight fragments of a K tile are read in phases 0-1, the activation
//        fragments in phases 0 and 2 (see the main loop).
using LmCfgC = LmCfg<256, 64, 2, true>;
using LmCfgC8 = LmCfg<256, 64, 2, true, true>;  // fp8 weight operand (two 48-KiB buffers)
using LmCfgA8 = LmCfg<128, 64, 3, false, true>;
using LmCfgC88 = LmCfg<256, 128, 2, true, false, true>;  // fp8 x fp8 on the block-scaled MFMA (two 64-KiB buffers)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

__device__ const u32x4_t g_zero_page_lm[4] = {};

// GLM_LM_STAMP=1 (timing study only): waves 0 and 4 of workgroup 0 accumulate s_memtime deltas per phase segment
#ifndef GLM_LM_STAMP
#define GLM_LM_STAMP 0
#endif
#if GLM_LM_STAMP
__device__ unsigned long long g_lm_stamps[2][8];
__device__ unsigned long long g_lm_tile_stamps[8];  // workgroup 8, wave 0: cycles per tile switch segment, [7] = tiles
// r05: workgroup 8, wave 0, product quadrant loop: cycles by K-tile POSITION summed over the workgroup's tiles — [0..5] the first
// six K tiles of an output tile, [6..11] the last six (11 = the last), [12] every other one; [13..25] the counts
__device__ unsigned long long g_lm_kt_stamps[26];
#define LM_TS(i) do { unsigned long long t_ = __builtin_readcyclecounter(); if (i) ts_acc[(i) - 1] += t_ - ts_prev; ts_prev = t_; } while (0)
#define LM_T(i) do { if (GLM_LM_STAMP) { unsigned long long t_ = __builtin_readcyclecounter(); if (i) tacc[i - 1] += t_ - tprev; tprev = t_; } } while (0)
#else
#define LM_T(i) do {} while (0)
#define LM_TS(i) do {} while (0)
#endif

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void_t*)gsrc, (lds_void_t*)lds_wave_base, 16, 0, 0);
}
// buffer_load_dwordx4 ... offen lds: per-lane byte offset in a VGPR, K position in the scalar offset
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rsrc, void* lds_wave_base, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)lds_wave_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void lds_read_b128(u32x4_t& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(const lds_void_t*)p;
}
template <int OFF>
__device__ __forceinline__ void lds_read_b128_o(u32x4_t& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_read_b64_o(u32x2_t& dst, unsigned addr) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
// 8 fp8 e4m3 (k ascending from byte 0) -> the bf16x8 MFMA fragment, exact
__device__ __forceinline__ u32x4_t lm_fp8x8_to_bf16x8(const u32x2_t d) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
    const bf2_t a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d[0], 1.0f, false);
    const bf2_t b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d[0], 1.0f, true);
    const bf2_t c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d[1], 1.0f, false);
    const bf2_t e = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d[1], 1.0f, true);
    return u32x4_t{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, c),
                   __builtin_bit_cast(unsigned, e)};
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}
// Kernel variants (template parameter VAR, bit set; glm_debug_largem_variant() selects one for A/B runs):
//   LM_LEAN   the load phase carries no vector ALU work: operand tiles come in by buffer_load ... lds with a per-lane
//             byte offset that is fixed for the whole tile and the K position in the SCALAR offset; fragment reads
//             use one base register per k-step plus immediate offsets.  Needs K % BK == 0 and x/W spans < 2 GiB.
//   LM_WIDEST 16-byte epilogue stores: the packed results of two neighbouring 16-column tiles are exchanged between
//             the lane rows (v_permlane16_swap), so a lane stores 8 consecutive columns.
//   LM_NOSTORE timing study only: the epilogue computes everything and stores nothing.
//   LM_DMA2 / LM_DMA4 (with LM_LEAN) 2 / all 4 of a K tile's DMA instructions are issued from the MFMA phase, spread
//             between the MFMAs, instead of from the (longer) load phase.
//   LM_LDSEPI (down / plain) the epilogue transposes each 16-row x 64-column block of a wave through a private LDS
//             patch in the free ring slot, so that a store instruction writes 8 whole 128-byte lines.
//   LM_DESYNC the workgroups of XCD x start x * GLM_LM_DESYNC_US later: every workgroup runs tiles of the same length, so
//             without it all 256 CUs reach their epilogues together and the output (1.2 GB for the down projection at
//             T=16384) leaves in chip-wide bursts that the HBM absorbs at its write rate while every matrix pipe waits.
//             An XCD stays in step with itself (its co-resident tiles share operand panels through its L2).
//   LM_NT     non-temporal epilogue stores (the output is not read again by this kernel).
constexpr int LM_LEAN = 1, LM_WIDEST = 2, LM_NOSTORE = 4, LM_DMA2 = 8, LM_DMA4 = 16, LM_LDSEPI = 32, LM_DESYNC = 64,
              LM_NT = 128, LM_EPIORD = 256, LM_NOPRIO = 512, LM_LOADPRIO = 1024,  // issue-priority studies
              LM_DEEP = 2048,  // 5-stage ring (all 160 KiB of LDS): one more K tile of prefetch distance
              LM_PH8 = 4096,   // CfgC: BK = 64, two buffers, four 16-MFMA phases per K tile (needs LM_LEAN)
              LM_P3X = 8192, LM_QW = 16384, LM_RPRE = 32768, LM_RFIRST = 65536, LM_MIR = 131072,  // schedules of that loop (described there)
              LM_PARK = 262144,  // part of the epilogue's stores deferred into the next tile's K loop through LDS
              LM_CONT = 524288;  // r05: the K tiles of consecutive output tiles form ONE stream through the two buffers (below)
//   LM_CONT   (grouped gate|up / down on the product quadrant loop, bf16 weights, even K-tile count) the operand pieces of
//             the NEXT output tile's K tiles 0 and 1 are issued from the last two K tiles of the current one, in the places
//             where the loop would issue W(kt+2) and X(kt+1) if the tile went on: the per-tile ring fill, its full drain at
//             the top of a tile (r05 stamps: 4.4k cycles per tile, a chip-wide burst: every workgroup switches tiles at the
//             same time) and the source set-up in front of the epilogue disappear; the epilogue runs between K tile nk - 1
//             and K tile 0' with W(1') in flight.  The next tile is looked up (tile table) and its gathered source rows are
//             loaded at the START of the current tile; the per-row scales of the finished tile are requested in front of
//             the last W pieces.  Same MFMAs in the same order: bit-identical.  MEASURED r05 (study builds only): gate|up 2685
//             against 2630 us, down 1565 against 1522 us at T = 16384.  The stamps by K-tile position say why: the 4.4k cycles
//             did not go away, they moved into K tile 0' (8.1-9.0k cycles against 4.0k for every other K tile) — what the
//             top-of-tile wait drains is not the ring fill but the finished tile's STORES (every workgroup reaches its
//             epilogue at the same time: a chip-wide write burst), and with one VM counter for loads and stores the first
//             counted wait behind the epilogue has to let them retire whatever it is waiting for.
//   LM_EPIORD the epilogue's global LOADS (per-row scales) are issued before the next tile's ring fill, so that waiting
//             for them does not drain the fill; and the wait that opens a tile leaves the previous tile's stores in
//             flight (vmcnt counts in issue order and the fill was issued before them) whenever their number is
//             known exactly — full tiles.
#ifndef GLM_LM_DESYNC_MODE
#define GLM_LM_DESYNC_MODE 0  // 0: r02, one step of GLM_LM_DESYNC_CYCLES per XCD; 1: r05, every workgroup its own phase of one tile time
#endif
#ifndef GLM_LM_DESYNC_CYCLES
#define GLM_LM_DESYNC_CYCLES 5120  // per XCD step, shader clocks (~2.5 us)
#endif
// wait for all LDS reads; every fragment register is named so that no consumer moves above it
__device__ __forceinline__ void lds_wait(u32x4_t (&f)[12]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]),
                   "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11])
                 :
                 : "memory");
}
__device__ __forceinline__ void lds_wait(u32x4_t (&f)[16]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]),
                   "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]), "+v"(f[12]), "+v"(f[13]), "+v"(f[14]), "+v"(f[15])
                 :
                 : "memory");
}
// counted wait: leave the newest `tiles` tiles' DMA (DPW instructions each) in flight
template <int DPW>
__device__ __forceinline__ void dma_wait_leave(int tiles) {
    if (tiles <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (tiles == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(DPW) : "memory");
    else if (tiles == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * DPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(3 * DPW) : "memory");  // 5-stage ring: three tiles stay in flight
}

// glm_locate_block for ANY wave of the workgroup, result in (uniform) registers: (group, first row, rows) of row-block
// combo `target`, group -1 if there is none.
__device__ __forceinline__ void lm_locate_wave(const int32_t* __restrict__ offs, int G, int64_t R, int MB, int target, int lane,
                                               int& g, int& m0, int& cnt) {
    g = -1;
    m0 = 0;
    cnt = 0;
    if (offs == nullptr) {
        const int64_t r0 = (int64_t)target * MB;
        if (r0 < R) {
            g = 0;
            m0 = (int)r0;
            cnt = (int)((R - r0) < MB ? (R - r0) : MB);
        }
        return;
    }
    int base = 0;
    for (int b = 0; b < G; b += 64) {
        const int gi = b + lane;
        int lo = 0, c = 0;
        if (gi < G) {
            lo = offs[gi];
            c = offs[gi + 1] - lo;
        }
        const int nt = (c + MB - 1) / MB;
        int incl = nt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        const int excl = incl - nt;
        const bool hit = target >= base + excl && target < base + incl;
        const unsigned long long m = __ballot(hit);
        if (m != 0) {
            const int src = __ffsll((long long)m) - 1;
            const int r0 = lo + (target - base - excl) * MB;
            g = __shfl(gi, src, 64);
            m0 = __shfl(r0, src, 64);
            cnt = __shfl((lo + c - r0) < MB ? (lo + c - r0) : MB, src, 64);
            return;
        }
        base += __shfl(incl, 63, 64);
        if (base > target) break;  // uniform
    }
}

// Persistent: one workgroup per CU walks its XCD's contiguous run of tiles with a stride of (workgroups per XCD).
// Between two tiles the next tile's first ring stages are put in flight BEFORE the finished tile's epilogue, so the
// ring-fill latency and the workgroup relaunch (~6 % of a 256x256x1408 tile) hide behind the epilogue's stores.
template <int MODE, class C, int VAR, bool WSCALE>
__global__ __launch_bounds__(512) void gemm_largem_kernel(const GemmParams p, int n_tiles, int total_tiles, int LM_BAND) {
    constexpr bool LEAN = (VAR & LM_LEAN) != 0;
    // DMA instructions of a K tile issued from the MFMA phase (the rest from the load phase)
    constexpr int S2 = !LEAN ? 0 : (VAR & LM_DMA4) ? C::DPW : (VAR & LM_DMA2) ? C::DPW / 2 : 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LM_PARK (quadrant-phase loop with 16-byte stores): the last PARK_N store instructions of a tile's epilogue go to the
    // 32 KiB of LDS the two buffers leave free and leave for HBM one per K tile from the NEXT tile's K loop (right behind
    // its phase-3 wait), while the store path is otherwise idle.  The tile search then borrows the tail of buffer 1.
    constexpr bool PARK = C::PH8 && (VAR & LM_PARK) != 0 && (VAR & LM_WIDEST) != 0 && (VAR & LM_NOSTORE) == 0;
    constexpr int PARK_N = 4;
    constexpr bool CONT = (VAR & LM_CONT) != 0 && C::PH8 && (VAR & LM_QW) != 0 && (VAR & LM_RFIRST) != 0 && (VAR & LM_P3X) == 0 &&
                          !PARK && !C::A8 && !C::W8 && !WSCALE && MODE != GEMM_PLAIN && (VAR & LM_EPIORD) != 0;
    int* sh = reinterpret_cast<int*>(smem + C::NSTAGE * C::STAGE - ((C::ALIAS_SH || PARK) ? 16 : 0));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid & 3, wm = wid >> 2;
    const int c = lane & 15, g4 = lane >> 4;
    const int K = p.K;
    const int nk = (K + C::BK - 1) / C::BK;

    // XCD-aware tile list (bijective for any tile count): workgroups with equal id%8 share an XCD and get one
    // contiguous run of the band-ordered tile list
    int chunk_base, chunk_len;
    {
        const int q = total_tiles >> 3, r = total_tiles & 7, x = blockIdx.x & 7;
        chunk_base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        chunk_len = q + (x < r ? 1 : 0);
    }
    const int pstep = gridDim.x >> 3;  // host launches a multiple of 8 workgroups
    int pos = blockIdx.x >> 3;

    // tile state: *_c = tile being computed / stored, plain = tile whose operands are being staged
    int grp = -1, m0 = 0, mcnt = 0, ntile = 0;
    // Tile table: the (group, first row, rows, n-tile) of this workgroup's first TAB_N tile positions, found ONCE at
    // kernel start by the 8 waves in parallel (a search per tile switch cost ~1.9k cycles of a 80k-cycle down tile: two
    // workgroup barriers and a dependent chain of L2 round trips) and read back with one LDS load per switch.
    constexpr bool TAB = !C::ALIAS_SH && !PARK;
    constexpr int TAB_N = 64;
    const unsigned tab_addr = lds_addr(smem) + C::NSTAGE * C::STAGE + 16;
    const int pos0 = pos;
    if constexpr (TAB) {
        for (int i = wid; i < TAB_N && pos0 + i * pstep < chunk_len; i += 8) {
            const int bid = chunk_base + pos0 + i * pstep;
            const int band = bid / (LM_BAND * n_tiles);
            const int rem = bid % (LM_BAND * n_tiles);
            int g_, a_, b_;
            lm_locate_wave(p.offs, p.G, p.R, C::BM, band * LM_BAND + rem % LM_BAND, lane, g_, a_, b_);
            if (lane == 0) {
                const u32x4_t e = {(unsigned)g_, (unsigned)a_, (unsigned)b_, (unsigned)(rem / LM_BAND)};
                asm volatile("ds_write_b128 %0, %1" ::"v"(tab_addr + i * 16), "v"(e) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }
    int ti = 0;  // index of `pos` in this workgroup's position list
    // next valid tile at or after `pos` (uniform); leaves pos on it
    auto find_tile = [&]() -> bool {
        while (pos < chunk_len) {
            int g_, a_, b_, nt_;
            if (TAB && ti < TAB_N) {
                u32x4_t e;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(tab_addr + ti * 16) : "memory");
                g_ = __builtin_amdgcn_readfirstlane((int)e[0]);
                a_ = __builtin_amdgcn_readfirstlane((int)e[1]);
                b_ = __builtin_amdgcn_readfirstlane((int)e[2]);
                nt_ = __builtin_amdgcn_readfirstlane((int)e[3]);
            } else {
                const int bid = chunk_base + pos;
                const int band = bid / (LM_BAND * n_tiles);
                const int rem = bid % (LM_BAND * n_tiles);
                const int combo = band * LM_BAND + rem % LM_BAND;
                glm_locate_block(p.offs, p.G, p.R, C::BM, combo, sh);
                __syncthreads();
                g_ = sh[0], a_ = sh[1], b_ = sh[2];
                nt_ = rem / LM_BAND;
                __syncthreads();  // sh is rewritten by the next search step
            }
            if (g_ >= 0) {
                grp = g_;
                m0 = a_;
                mcnt = b_;
                ntile = nt_;
                return true;
            }
            pos += pstep;
            ++ti;
        }
        return false;
    };

    // ---- DMA sources.  One instruction = RPI tile rows x CPR chunks (1 KiB); lane → row
    // +(lane / CPR), LDS position lane % CPR holds global chunk (lane % CPR) ^ swz(row). ----
    const bf16_t* wsrc[C::W_DMA];
    const bf16_t* xsrc[C::X_DMA];
    int wch[C::W_DMA], xch[C::X_DMA];
    unsigned wvo[C::W_DMA], xvo[C::X_DMA];  // LEAN: byte offsets of this lane's chunk at k = 0
    __amdgpu_buffer_rsrc_t wrs, xrs;
    if (LEAN) xrs = make_rsrc(p.x);
    // (CONT) the next tile, found at the start of the current one: tile-table entry, list position, gathered source rows
    bool nxt = false;
    int n_grp = -1, n_m0 = 0, n_mcnt = 0, n_ntile = 0, n_pos = 0, n_ti = 0;
    int nsrow[C::X_DMA];
    auto setup_sources = [&](auto prec) {
        constexpr bool PRE = decltype(prec)::value;  // the source rows come from nsrow[] (loaded a tile ago)
        if (LEAN) {
            // descriptor of the group's weights (wave-uniform; grp came through LDS, so make that provable)
            const int64_t goff = (int64_t)__builtin_amdgcn_readfirstlane(grp) * p.w_gstride;
            if constexpr (C::W8 || C::A8) wrs = make_rsrc(reinterpret_cast<const char*>(p.w) + goff);  // fp8: strides in bytes
            else wrs = make_rsrc(p.w + goff);
#pragma unroll
            for (int i = 0; i < C::W_DMA; ++i) {
                const int row = (wid * C::W_DMA + i) * C::WRPI + lane / C::WCPR;
                const int ch = (lane % C::WCPR) ^ C::wswz(row);
                int wr;
                if (MODE == GEMM_GATEUP) wr = (row >> 7) * p.N + min(ntile * 128 + (row & 127), p.N - 1);
                else wr = min(ntile * LM_BN + row, p.N - 1);
                if constexpr (C::W8 || C::A8) wvo[i] = (unsigned)(wr * (int)p.ldw + ch * 16);
                else wvo[i] = (unsigned)(wr * (int)p.ldw + ch * 8) * 2u;
            }
#pragma unroll
            for (int i = 0; i < C::X_DMA; ++i) {
                const int row = (wid * C::X_DMA + i) * C::RPI + lane / C::CPR;
                const int ch = (lane % C::CPR) ^ C::swz(row);
                const int mr = row < mcnt ? row : 0;
                const int srow = PRE ? nsrow[i] : (p.src ? p.src[m0 + mr] : m0 + mr);
                if constexpr (C::A8) xvo[i] = (unsigned)(srow * (int)p.ldx + ch * 16);  // fp8 activations: ldx in bytes
                else xvo[i] = (unsigned)(srow * (int)p.ldx + ch * 8) * 2u;
            }
            return;
        }
        const bf16_t* wg = p.w + (int64_t)grp * p.w_gstride;
#pragma unroll
        for (int i = 0; i < C::W_DMA; ++i) {
            const int row = (wid * C::W_DMA + i) * C::RPI + lane / C::CPR;  // 0..255
            const int ch = (lane % C::CPR) ^ C::swz(row);
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
        for (int i = 0; i < C::X_DMA; ++i) {
            const int row = (wid * C::X_DMA + i) * C::RPI + lane / C::CPR;  // 0..BM-1
            const int ch = (lane % C::CPR) ^ C::swz(row);
            xch[i] = ch;
            const int mr = row < mcnt ? row : 0;
            const int64_t srow = p.src ? (int64_t)p.src[m0 + mr] : (int64_t)(m0 + mr);
            xsrc[i] = p.x + srow * p.ldx + ch * 8;
        }
    };
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_lm);
    // LEAN only: DMA instruction j (0 .. DPW-1; weights first) of K tile kt into ring slot `slot`
    auto stage_one = [&](int slot, int kt, auto jc) {
        constexpr int j = decltype(jc)::value;
        const int k0 = kt * C::BK;
        if constexpr (j < C::W_DMA)
            blds16(wrs, smem + slot * C::STAGE + wid * C::W_DMA * 1024 + j * 1024, wvo[j], k0 * ((C::W8 || C::A8) ? 1 : 2));
        else
            blds16(xrs, smem + slot * C::STAGE + C::WT + wid * C::X_DMA * 1024 + (j - C::W_DMA) * 1024,
                   xvo[j - C::W_DMA], k0 * (C::A8 ? 1 : 2));
    };
    auto stage = [&](int slot, int kt) {
        char* wdst = smem + slot * C::STAGE + wid * C::W_DMA * 1024;
        char* xdst = smem + slot * C::STAGE + C::WT + wid * C::X_DMA * 1024;
        const int k0 = kt * C::BK;
        if (LEAN) {
#pragma unroll
            for (int i = 0; i < C::W_DMA; ++i)
                blds16(wrs, wdst + i * 1024, wvo[i], k0 * ((C::W8 || C::A8) ? 1 : 2));
#pragma unroll
            for (int i = 0; i < C::X_DMA; ++i)
                blds16(xrs, xdst + i * 1024, xvo[i], k0 * (C::A8 ? 1 : 2));
            return;
        }
#pragma unroll
        for (int i = 0; i < C::W_DMA; ++i) glds16(k0 + wch[i] * 8 < K ? wsrc[i] + k0 : zero, wdst + i * 1024);
#pragma unroll
        for (int i = 0; i < C::X_DMA; ++i) glds16(k0 + xch[i] * 8 < K ? xsrc[i] + k0 : zero, xdst + i * 1024);
    };
    // A8 DOWN: the E8M0 block scales of the tile's activation rows behind the tile table (16 KiB), two 1-KiB LDS-DMA pieces
    // per wave.  LDS image: 16-byte slot (g * 256 + row) = the scales of (row, natural block 4 kt + g) for kt = 0 .. 15 — a
    // lane (row c of X tile mt, lane row g4) reads ONE dword of its slot every four K tiles (bytes kt & ~3 .. + 3) and shifts
    // the byte of the K tile into place (r05 first form: one ds_read_u8 per X tile and K tile out of [row][64 B] records —
    // 8-way bank conflicts, the down projection gained nothing over bf16: profiles/r05/fp8a8_probe.txt).
    constexpr bool XBS = C::A8 && MODE == GEMM_DOWN;
    constexpr int XBS_OFF = C::NSTAGE * C::STAGE + 16 + 64 * 16;
    auto stage_scales = [&]() {
        if constexpr (XBS) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int slot = (wid * 2 + i) * 64 + lane;       // lane-linear destination: slot = g * 256 + row
                const int row = slot & 255, g = slot >> 8;
                const int mr = row < mcnt ? row : 0;
                glds16(p.xbscale + (int64_t)(m0 + mr) * 64 + g * 16, smem + XBS_OFF + (wid * 2 + i) * 1024);
            }
        }
    };
    auto fill_ring = [&]() {
        stage_scales();
        if constexpr (C::PH8) {
            // both operands of K tile 0 and the weights of K tile 1 (their slot is the first one the loop refills)
            static_for<C::DPW>([&](auto jc) { stage_one(0, 0, jc); });
            if (nk > 1) {
                static_for<C::W_DMA>([&](auto jc) { stage_one(1, 1, jc); });
                if constexpr ((VAR & LM_P3X) != 0) {  // what phase 3 of a K tile "-1" would have issued
                    stage_one(1, 1, std::integral_constant<int, C::W_DMA + 0>{});
                    stage_one(1, 1, std::integral_constant<int, C::W_DMA + 1>{});
                }
            }
            return;
        }
#pragma unroll
        for (int d = 0; d < C::DIST; ++d)
            if (d < nk) stage(d, d);
    };

    // ---- fragment read offsets relative to the stage base: lane (c,g4) reads row +c, chunk 4ks+g4
    unsigned rel[C::NF];  // per k-step: 4 weight fragments then MT activation fragments
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int wrow;
            if (MODE == GEMM_GATEUP) wrow = (t < 2 ? wn * 32 + 16 * t : 128 + wn * 32 + 16 * (t - 2)) + c;
            else wrow = wn * 64 + 16 * t + c;
            if constexpr (C::W8)  // 8 bytes: k = 32 ks + 8 g4 .. +8 = chunk 2 ks + (g4 >> 1), half g4 & 1
                rel[ks * (4 + C::MT) + t] = wrow * C::WRB + (((2 * ks + (g4 >> 1)) ^ C::wswz(wrow)) << 4) + ((g4 & 1) << 3);
            else
                rel[ks * (4 + C::MT) + t] = wrow * C::RB + (((4 * ks + g4) ^ C::swz(wrow)) << 4);
        }
#pragma unroll
        for (int t = 0; t < C::MT; ++t) {
            const int xrow = wm * (C::BM / 2) + 16 * t + c;
            rel[ks * (4 + C::MT) + 4 + t] = C::WT + xrow * C::RB + (((4 * ks + g4) ^ C::swz(xrow)) << 4);
        }
    }
    const unsigned base = lds_addr(smem);
    // LEAN: one base per k-step and operand; the tile index inside the operand is an immediate offset
    // (the swizzle depends on the low bits of the row only, i.e. on c)
    unsigned fbw[C::KS], fbx[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
        fbw[ks] = base + rel[ks * (4 + C::MT)];
        fbx[ks] = base + rel[ks * (4 + C::MT) + 4];
    }
    // all fragment reads of a K tile (slot byte offset so, wave-uniform)
    // W8: behind the read wait, the weight fragments fp8 -> bf16 (the asm reads are invisible to the compiler: nothing may
    // touch fr[] between a read and the lgkmcnt wait, which is why they have registers of their own)
    auto frags_cvt = [&](u32x4_t (&f)[C::NF], u32x2_t (&fr)[C::W8 ? C::KS * 4 : 1]) {
        if constexpr (C::W8) {
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    asm volatile("" : "+v"(fr[ks * 4 + t]));
                    f[ks * (4 + C::MT) + t] = lm_fp8x8_to_bf16x8(fr[ks * 4 + t]);
                }
        }
    };
    auto read_frags = [&](u32x4_t (&f)[C::NF], u32x2_t (&fr)[C::W8 ? C::KS * 4 : 1], unsigned so, auto half_tag) {
        constexpr int HALF = decltype(half_tag)::value;  // 0: all, 1: first half, 2: second half (index order as rel[])
        static_for<C::KS>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            const unsigned aw = fbw[ks] + so, ax = fbx[ks] + so;
            static_for<4>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int idx = ks * (4 + C::MT) + t;
                constexpr int off = MODE == GEMM_GATEUP ? (t < 2 ? 16 * t : 128 + 16 * (t - 2)) * C::WRB : 16 * t * C::WRB;
                if (HALF == 0 || (HALF == 1) == (idx < C::NF / 2)) {
                    if constexpr (C::W8) lds_read_b64_o<off>(fr[ks * 4 + t], aw);  // fp8 as read; frags_cvt() fills f[idx]
                    else lds_read_b128_o<off>(f[idx], aw);
                }
            });
            static_for<C::MT>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int idx = ks * (4 + C::MT) + 4 + t;
                if (HALF == 0 || (HALF == 1) == (idx < C::NF / 2)) lds_read_b128_o<16 * t * C::RB>(f[idx], ax);
            });
        });
    };

    // store instructions per wave of a full tile (16-byte stores: one per pair of output tiles)
    constexpr int LM_NOUT = MODE == GEMM_GATEUP ? 2 : 4;
    constexpr int LM_STORES_FULL = C::MT * ((VAR & LM_WIDEST) ? LM_NOUT / 2 : LM_NOUT);
    int pending_stores = 0;
    bool park_pending = false;
    int park_m0 = 0, park_mcnt = 0, park_ncolw = 0;
    const unsigned park_addr = lds_addr(smem) + C::NSTAGE * C::STAGE + wid * (PARK_N * 1024) + lane * 16;
    auto park_flush = [&](int j) {  // parked store j of the previous tile (same lane that parked it)
        u32x4_t val;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(val) : "v"(park_addr + j * 1024) : "memory");
        const int sidx = LM_STORES_FULL - PARK_N + j;
        const int mt = sidx / (LM_NOUT / 2), t = (sidx % (LM_NOUT / 2)) * 2;
        const int row = wm * (C::BM / 2) + 16 * mt + c;
        const int n = park_ncolw + 16 * (t + (g4 & 1)) + 8 * (g4 >> 1);
        if (row < park_mcnt && n < p.N)
            *reinterpret_cast<u32x4_t*>(p.y + (int64_t)(park_m0 + row) * p.ldy + n) = val;
    };
    if (!find_tile()) return;
    if constexpr ((VAR & LM_DESYNC) != 0) {
#if GLM_LM_DESYNC_MODE == 1
        // r05 (study): the 32 workgroups of an XCD spread evenly over ONE output tile's time (K loop ~2600 clocks per K tile +
        // ~11k of tile switch), XCDs interleaved by a 1/256 step: no two workgroups of the chip reach their epilogue together.
        // Measured (profiles/r05/largem_desync_per_workgroup_ab.txt): gate|up 2810 against 2650 us, down 1540 against 1530 us —
        // the start delay (up to one tile of 28-41) and the lost co-scheduling of tiles that share operand panels through an
        // XCD's L2 cost more than the smoothed write burst returns.
        const int tile_clocks = nk * 2600 + 11000;
        const int slot = (((int)blockIdx.x >> 3) & 31) * 8 + ((int)blockIdx.x & 7);
        for (int i = 0; i < (int)((long long)slot * tile_clocks / 256 / 1024); ++i) __builtin_amdgcn_s_sleep(16);
#else
        for (int i = 0; i < (int)(blockIdx.x & 7) * (GLM_LM_DESYNC_CYCLES / 1024); ++i) __builtin_amdgcn_s_sleep(16);  // 16 x 64 clocks
#endif
    }
    setup_sources(std::false_type{});
    fill_ring();
    bool cont_in = false;  // (CONT) this tile's first K tiles were put in flight by the previous tile's K loop
#if GLM_LM_STAMP
    unsigned long long ts_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ts_prev = 0;
#endif
    for (;;) {
        // the tile whose ring is filling is now the tile being computed
        const int grp_c = grp, m0_c = m0, mcnt_c = mcnt, ntile_c = ntile;
        LM_TS(0);
        f32x4_t acc[4][C::MT];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < C::MT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        float sc_pre[C::MT];  // per-row scales of this tile (down projection), loaded before its epilogue (LM_EPIORD)

        // ---- main loop: two phases per K tile, waves 4-7 run ONE BARRIER BEHIND waves 0-3 ----
        //   phase 1: LDS fragment reads of tile kt + DMA issue of tile kt+DIST + counted wait for
        //            this wave's share of tile kt+1, then barrier
        //   phase 2: the tile's MFMAs, then barrier
        // The two waves that share a SIMD (w, w+4) are therefore always in opposite phases: one feeds
        // the matrix pipe while the other issues loads (in lockstep the pipe idles ~60 % of the time).
        // Ordering under the stagger (the other group sits one barrier earlier/later):
        //   RAW  tile kt+1 is read in phase 1 of kt+1, i.e. after this group's B2(kt) = other group's
        //        B1(kt) (or later); every wave waited for its share of kt+1 BEFORE its own B1(kt).
        //   WAR  tile kt+DIST overwrites the slot of kt-1 in phase 1 of kt, i.e. after this group's
        //        B2(kt-1) >= other group's B1(kt-1); every wave finished reading kt-1 before its own
        //        B1(kt-1).
        // The ring fill was issued before the previous tile's epilogue stores, and stores retire out of order with
        // loads, so the first wait of a tile is a full drain (the stores are long acknowledged by then).
        if (!(CONT && cont_in)) {
        if constexpr ((VAR & LM_EPIORD) != 0) {
            // pending_stores = store instructions this wave issued after the ring fill (0 when not known exactly)
            if (pending_stores == LM_STORES_FULL) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LM_STORES_FULL) : "memory");
            else if (PARK && pending_stores == LM_STORES_FULL - PARK_N)
                asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LM_STORES_FULL - PARK_N) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                 // every share of the first DIST tiles landed
        }
        // (CONT && cont_in: every wave waited for its own pieces of K tile 0 in phase 3 of the previous tile's last K tile and
        // has passed workgroup barriers since; W of K tile 1 is in flight, as behind fill_ring)
        if constexpr (CONT) {
            // the next tile of this workgroup's list, if the table holds it, and its gathered source rows: consumed two K
            // tiles before the end of this tile's K loop
            nxt = false;
            if ((nk & 1) == 0 && nk >= 4) {
                int pp = pos + pstep, tt = ti + 1;
                while (pp < chunk_len && tt < TAB_N) {
                    u32x4_t e;
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(tab_addr + tt * 16) : "memory");
                    const int g_ = __builtin_amdgcn_readfirstlane((int)e[0]);
                    if (g_ >= 0) {
                        nxt = true;
                        n_grp = g_;
                        n_m0 = __builtin_amdgcn_readfirstlane((int)e[1]);
                        n_mcnt = __builtin_amdgcn_readfirstlane((int)e[2]);
                        n_ntile = __builtin_amdgcn_readfirstlane((int)e[3]);
                        n_pos = pp;
                        n_ti = tt;
                        break;
                    }
                    pp += pstep;
                    ++tt;
                }
            }
            if (nxt) {
#pragma unroll
                for (int i = 0; i < C::X_DMA; ++i) {
                    const int row = (wid * C::X_DMA + i) * C::RPI + lane / C::CPR;
                    const int mr = row < n_mcnt ? row : 0;
                    nsrow[i] = p.src ? p.src[n_m0 + mr] : n_m0 + mr;
                }
            }
        }
        LM_TS(1);  // 0: top-of-tile wait + barrier
        if constexpr (C::PH8) {
        // ---- CfgC: four phases per K tile, each = load interval | barrier | 16 MFMAs | barrier; waves 4-7 run one
        // barrier behind waves 0-3, so the two waves of a SIMD alternate between the two kinds of interval.
        //   phase   fragment reads (this K tile, buffer b)          DMA issued                         MFMAs
        //     0     W tiles 0,1 (4) + X tiles 0-3 (8)               X(kt+1) pieces 0,1 -> buffer !b     W01 x X0-3
        //     1     W tiles 2,3 (4)                                 X(kt+1) pieces 2,3 -> buffer !b     W23 x X0-3
        //     2     X tiles 4-7 (8)                                 W(kt+2) pieces 0,1 -> buffer b      W23 x X4-7
        //     3     -                                               W(kt+2) pieces 2,3 -> buffer b      W01 x X4-7
        //           then the counted wait for this wave's share of K tile kt+1 (W(kt+2) stays in flight)
        // Every load interval ends with lgkmcnt(0) BEFORE its barrier, so a fragment read has retired when its wave
        // passes that barrier.  Barrier epochs: waves 0-3 run load interval P (= 4 kt + phase) in epoch 2P, waves 4-7
        // in epoch 2P+1.
        //   WAR  W(kt) is last read in load interval 4kt+1 (retired by the end of epoch 8kt+3 at the latest) and its
        //        slot refilled from load interval 4kt+2 (epoch >= 8kt+4).  X(kt-1) is last read in load interval
        //        4kt-2 (epoch <= 8kt-3) and its slot refilled from load interval 4kt (epoch >= 8kt).
        //   RAW  K tile kt+1 is first read in load interval 4kt+4 (epoch >= 8kt+8); every wave waited for its own
        //        pieces of it at the end of load interval 4kt+3 (epoch <= 8kt+7), i.e. before barrier 8kt+8.
        static_assert(!C::PH8 || LEAN, "the quadrant-phase loop needs the lean load phase");
        // Variants of the same loop (bit-identical results):
        //   LM_P3X  X(kt+2) pieces 0,1 are issued in phase 3 of K tile kt (its slot is free from there on) and pieces
        //           2,3 in phase 1 of kt+1: phase 0, which carries 12 fragment reads, issues no DMA, phase 3, which
        //           carries no reads, issues 4; the counted wait of phase 3 then leaves 6 in flight.
        //   LM_QW   a quadrant is all 4 W tiles x 2 X tiles: reads 12 / 4 / 4 / 4 instead of 12 / 4 / 8 / 0 and 16
        //           fragment registers less; W(kt) is dead after phase 0, X(kt) after phase 3 (excludes LM_P3X).
        constexpr bool P3X = (VAR & LM_P3X) != 0, QW = (VAR & LM_QW) != 0;
        static_assert(!(P3X && QW), "LM_P3X refills the X slot while LM_QW still reads it");
        if (wid >= 4) __builtin_amdgcn_s_barrier();   // the stagger
        u32x4_t fw[8], fx[8];  // [k-step][W tile 0-3], [k-step][X tile of the current part]
        int sxw[8] = {127, 127, 127, 127, 127, 127, 127, 127};  // A8: per X tile, the E8M0 scale bytes of four K tiles (127 = 2^0:
                                                                 // gate|up, whose activation scale is per token)
        int ksh = 0;                                             // 8 * (K tile & 3): the byte of the current K tile
        const unsigned sx_base = lds_addr(smem) + XBS_OFF + (g4 * 256 + wm * (C::BM / 2) + c) * 16;
        auto rd_s_ = [&](int (&w)[8], int kt_) {  // every fourth K tile, in its phase 0 (behind the same lgkmcnt wait)
            if constexpr (XBS) {
                if ((kt_ & 3) == 0) {
                    const unsigned a_ = sx_base + kt_;
#define GLM_RD_S(J) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(w[J]) : "v"(a_), "i"((J) * 256))
                    GLM_RD_S(0); GLM_RD_S(1); GLM_RD_S(2); GLM_RD_S(3); GLM_RD_S(4); GLM_RD_S(5); GLM_RD_S(6); GLM_RD_S(7);
#undef GLM_RD_S
                }
            }
        };
#define rd_s(KT) rd_s_(sxw, KT)
        auto pin_s_ = [&](int (&w)[8]) {
            if constexpr (XBS) {
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(w[j]));
            }
        };
#define pin_s() pin_s_(sxw)
        u32x2_t fwr[C::W8 ? 8 : 1];  // W8: the fragments as read (fp8), converted into fw behind the read wait
        auto cvt_w = [&]() {
            if constexpr (C::W8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    asm volatile("" : "+v"(fwr[i]));
                    fw[i] = lm_fp8x8_to_bf16x8(fwr[i]);
                }
            }
        };
        auto pin = [&](u32x4_t& r) { asm volatile("" : "+v"(r)); };
        // W tiles t0 .. t0+NT-1, both k-steps
        auto rd_w = [&](unsigned so, auto t0c, auto ntc) {
            constexpr int t0 = decltype(t0c)::value, NT = decltype(ntc)::value;
            static_for<C::KS>([&](auto ksc) {
                constexpr int ks = decltype(ksc)::value;
                const unsigned aw = fbw[ks] + so;
                static_for<NT>([&](auto tc) {
                    constexpr int t = t0 + decltype(tc)::value;
                    constexpr int off = MODE == GEMM_GATEUP ? (t < 2 ? 16 * t : 128 + 16 * (t - 2)) * C::WRB : 16 * t * C::WRB;
                    if constexpr (C::W8) lds_read_b64_o<off>(fwr[ks * 4 + t], aw);
                    else lds_read_b128_o<off>(fw[ks * 4 + t], aw);
                });
            });
        };
        // X tiles m0 .. m0+NX-1 into fx[ks * NX + j], both k-steps
        auto rd_x = [&](unsigned so, auto m0c, auto nxc, auto xbc) {
            constexpr int mt0 = decltype(m0c)::value, NX = decltype(nxc)::value, XB = decltype(xbc)::value;
            static_for<C::KS>([&](auto ksc) {
                constexpr int ks = decltype(ksc)::value;
                const unsigned ax = fbx[ks] + so;
                static_for<NX>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    lds_read_b128_o<16 * (mt0 + j) * C::RB>(fx[XB + ks * NX + j], ax);
                });
            });
        };
        // W tiles t0 .. t0+NT-1 x X tiles m0 .. m0+NX-1 (16 MFMAs)
        auto quad = [&](auto t0c, auto ntc, auto m0c, auto nxc, auto xbc) {
            constexpr int t0 = decltype(t0c)::value, NT = decltype(ntc)::value;
            constexpr int mt0 = decltype(m0c)::value, NX = decltype(nxc)::value, XB = decltype(xbc)::value;
            if constexpr ((VAR & LM_LOADPRIO) != 0) __builtin_amdgcn_s_setprio(0);
            else if constexpr ((VAR & LM_NOPRIO) == 0) __builtin_amdgcn_s_setprio(1);
            if constexpr (C::A8) {
                // ONE K = 128 MFMA per (W tile, X tile): registers 0-3 = chunk g4, registers 4-7 = chunk 4 + g4 of the row
                typedef int v8i_t __attribute__((ext_vector_type(8)));
                typedef unsigned u32x8_t __attribute__((ext_vector_type(8)));
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int j = 0; j < NX; ++j) {
                        const u32x8_t a8 = __builtin_shufflevector(fw[t0 + tt], fw[4 + t0 + tt], 0, 1, 2, 3, 4, 5, 6, 7);
                        const u32x8_t b8 = __builtin_shufflevector(fx[XB + j], fx[XB + NX + j], 0, 1, 2, 3, 4, 5, 6, 7);
                        acc[t0 + tt][mt0 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                            __builtin_bit_cast(v8i_t, a8), __builtin_bit_cast(v8i_t, b8), acc[t0 + tt][mt0 + j], 0, 0, 0, 127, 0,
                            XBS ? (int)((unsigned)sxw[mt0 + j] >> ksh) : 127);
                    }
            } else
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks)
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int j = 0; j < NX; ++j)
                        acc[t0 + tt][mt0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8_t, fw[ks * 4 + t0 + tt]),
                            __builtin_bit_cast(bf16x8_t, fx[XB + ks * NX + j]), acc[t0 + tt][mt0 + j], 0, 0, 0);
            if constexpr ((VAR & LM_LOADPRIO) != 0) __builtin_amdgcn_s_setprio(1);  // the load interval runs at priority 1
            else if constexpr ((VAR & LM_NOPRIO) == 0) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto reads_done = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
        auto close_load = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        };
        auto dma2 = [&](int slot, int kt_, auto j0c) {
            constexpr int j0 = decltype(j0c)::value;
            const int kk = (CONT && kt_ >= nk) ? kt_ - nk : kt_;  // (CONT) past the end: K tile 0 / 1 of the NEXT output tile
            stage_one(slot, kk, std::integral_constant<int, j0>{});
            stage_one(slot, kk, std::integral_constant<int, j0 + 1>{});
        };
        using I0 = std::integral_constant<int, 0>;
        using I2 = std::integral_constant<int, 2>;
        using I4 = std::integral_constant<int, 4>;
        using I6 = std::integral_constant<int, 6>;
        using XP0 = std::integral_constant<int, C::W_DMA + 0>;  // X pieces 0,1
        using XP2 = std::integral_constant<int, C::W_DMA + 2>;  // X pieces 2,3
        if constexpr ((VAR & LM_MIR) != 0) {
        //   LM_MIR  2 x 4-tile quadrants in MIRRORED order on alternate K tiles, reads before DMA: even K tiles run
        //           (W01,X0-3) (W23,X0-3) (W23,X4-7) (W01,X4-7), odd ones (W23,..) (W01,..) (W01,..) (W23,..).  The W half
        //           that opens K tile kt+1 is the one whose registers the last quadrant of K tile kt does not use, so
        //           its 4 fragment reads move into load interval 3 of kt: reads per interval 8 / 4 / 8 / 4 instead of
        //           12 / 4 / 8 / 0.  W(kt+1) (issued in phases 2-3 of kt-1) is waited for at the end of load interval 2
        //           (vmcnt leaves the 6 younger pieces X(kt+1) 0-3, W(kt+2) 0,1 in flight), i.e. before barrier 8kt+6
        //           at the latest, and first read in load interval 4kt+3 (epoch >= 8kt+6).
        static_assert(!P3X && !QW, "LM_MIR is its own schedule");
        static_assert(!C::W8 && !C::A8, "fp8 operands: product loop only");
        static_assert(!PARK, "LM_PARK is wired into the plain quadrant loop only");
        auto ktile = [&](int kt, auto parc) {
            constexpr int PAR = decltype(parc)::value;
            using HA = std::integral_constant<int, 2 * PAR>;        // first W tile of the half that opens this K tile
            using HB = std::integral_constant<int, 2 * (1 - PAR)>;  // ... of the other half
            const int b = kt & 1;
            const unsigned so = b * C::STAGE, so_n = (b ^ 1) * C::STAGE;
            const bool has1 = kt + 1 < nk, has2 = kt + 2 < nk;
            auto pin_w = [&](auto hc) {
                constexpr int t0 = decltype(hc)::value;
                pin(fw[t0]); pin(fw[t0 + 1]); pin(fw[4 + t0]); pin(fw[4 + t0 + 1]);
            };
            // ---- phase 0
            if (kt == 0) rd_w(so, HA{}, I2{});  // no previous K tile fetched them
            rd_x(so, I0{}, I4{}, I0{});
            __builtin_amdgcn_sched_barrier(0);
            if (has1) dma2(b ^ 1, kt + 1, XP0{});
            reads_done();
            pin_w(HA{});
#pragma unroll
            for (int i = 0; i < 8; ++i) pin(fx[i]);
            close_load();
            quad(HA{}, I2{}, I0{}, I4{}, I0{});
            __builtin_amdgcn_s_barrier();
            // ---- phase 1
            rd_w(so, HB{}, I2{});
            __builtin_amdgcn_sched_barrier(0);
            if (has1) dma2(b ^ 1, kt + 1, XP2{});
            reads_done();
            pin_w(HB{});
            close_load();
            quad(HB{}, I2{}, I0{}, I4{}, I0{});
            __builtin_amdgcn_s_barrier();
            // ---- phase 2
            rd_x(so, I4{}, I4{}, I0{});
            __builtin_amdgcn_sched_barrier(0);
            if (has2) dma2(b, kt + 2, I0{});
            reads_done();
#pragma unroll
            for (int i = 0; i < 8; ++i) pin(fx[i]);
            // W(kt+1) has landed: younger than it are X(kt+1) 0-3 and, if issued, W(kt+2) 0,1
            if (has2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(C::X_DMA + 2) : "memory");
            else if (has1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(C::X_DMA) : "memory");
            close_load();
            quad(HB{}, I2{}, I4{}, I4{}, I0{});
            __builtin_amdgcn_s_barrier();
            // ---- phase 3
            if (has1) rd_w(so_n, HB{}, I2{});  // opens K tile kt+1
            __builtin_amdgcn_sched_barrier(0);
            if (has2) {
                dma2(b, kt + 2, I2{});
                reads_done();
                asm volatile("s_waitcnt vmcnt(%0)" ::"i"(C::W_DMA) : "memory");
            } else {
                reads_done();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (has1) pin_w(HB{});
            close_load();
            quad(HA{}, I2{}, I4{}, I4{}, I0{});
            __builtin_amdgcn_s_barrier();
        };
        for (int kt = 0; kt < nk; kt += 2) {
            ktile(kt, I0{});
            if (kt + 1 < nk) ktile(kt + 1, std::integral_constant<int, 1>{});
        }
        } else
        if constexpr ((VAR & LM_RPRE) != 0) {
        //   LM_RPRE (with LM_QW) the fragment reads of phases 1-3 (two X tiles each) are issued at the START of the
        //           previous phase's MFMA interval into the other half of fx and retire behind its MFMAs: the load
        //           intervals of phases 1-3 carry DMA issue only.  X(kt) is then last read in MFMA interval 4kt+2
        //           (retired by the lgkmcnt(0) before its closing barrier, epoch <= 8kt+6) and refilled from load
        //           interval 4kt+4 (epoch >= 8kt+8); W(kt) is last read in load interval 4kt as under LM_QW.
        static_assert(QW, "LM_RPRE is a schedule of the LM_QW quadrants");
        static_assert(!C::W8 && !C::A8, "fp8 operands: product loop only");
        static_assert(!PARK, "LM_PARK is wired into the plain quadrant loop only");
        auto pre_x = [&](unsigned so, auto m0c, auto xbc) {  // issue the reads, keep the MFMAs behind them
            rd_x(so, m0c, I2{}, xbc);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto post_x = [&](auto xbc) {  // after the MFMAs: the prefetched fragments are in registers
            constexpr int XB = decltype(xbc)::value;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            pin(fx[XB + 0]); pin(fx[XB + 1]); pin(fx[XB + 2]); pin(fx[XB + 3]);
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int kt = 0; kt < nk; ++kt) {
            const int b = kt & 1;
            const unsigned so = b * C::STAGE;
            const bool has1 = kt + 1 < nk, has2 = kt + 2 < nk;
            // ---- phase 0
            if (has1) dma2(b ^ 1, kt + 1, XP0{});
            rd_w(so, I0{}, I4{});
            rd_x(so, I0{}, I2{}, I0{});
            reads_done();
#pragma unroll
            for (int i = 0; i < 8; ++i) pin(fw[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) pin(fx[i]);
            close_load();
            pre_x(so, I2{}, I4{});
            quad(I0{}, I4{}, I0{}, I2{}, I0{});
            post_x(I4{});
            __builtin_amdgcn_s_barrier();
            // ---- phase 1
            if (has1) dma2(b ^ 1, kt + 1, XP2{});
            close_load();
            pre_x(so, I4{}, I0{});
            quad(I0{}, I4{}, I2{}, I2{}, I4{});
            post_x(I0{});
            __builtin_amdgcn_s_barrier();
            // ---- phase 2
            if (has2) dma2(b, kt + 2, I0{});
            close_load();
            pre_x(so, I6{}, I4{});
            quad(I0{}, I4{}, I4{}, I2{}, I0{});
            post_x(I4{});
            __builtin_amdgcn_s_barrier();
            // ---- phase 3
            if (has2) {
                dma2(b, kt + 2, I2{});
                asm volatile("s_waitcnt vmcnt(%0)" ::"i"(C::W_DMA) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            close_load();
            quad(I0{}, I4{}, I6{}, I2{}, I4{});
            __builtin_amdgcn_s_barrier();
        }
        } else
#ifndef GLM_LM_A8_PHASES
#define GLM_LM_A8_PHASES 2   // 2: the two-phase fp8 x fp8 loop below (r05); 4: the bf16 product loop's four quadrant phases
#endif
        if constexpr (C::A8 && GLM_LM_A8_PHASES == 2) {
        // ---- fp8 x fp8, TWO phases per K tile (r05).  A block-scaled MFMA does the work of four bf16 ones, so a 4 x 2-tile
        // quadrant is 8 instructions (256 pipe cycles) behind the same two barriers: PMC of the four-phase form says the
        // matrix pipe is 42 % busy at 2.15 GHz (profiles/r05/pmc_prefill_gemm_a8_four_phase.json) — not the clock, the
        // barriers.  Here a phase is 4 W tiles x 4 X tiles = 16 MFMAs:
        //   phase   fragment reads (K tile kt, buffer b)        DMA issued                          MFMAs
        //     A     W tiles 0-3 (8) + X tiles 0-3 (8) [+ scales]  X(kt+1) pieces 0-3 -> buffer !b    W0-3 x X0-3
        //     B     X tiles 4-7 (8)                               W(kt+2) pieces 0-3 -> buffer b     W0-3 x X4-7
        //           then the counted wait for this wave's share of K tile kt+1 (W(kt+2) stays in flight)
        // Barrier epochs: waves 0-3 run load interval P (= 2 kt + phase) in epoch 2P, waves 4-7 in epoch 2P+1.
        //   WAR  W(kt) is last read in load interval 2kt (retired by the end of epoch 4kt+1) and its slot refilled from load
        //        interval 2kt+1 (epoch >= 4kt+2).  X(kt-1) is last read in load interval 2kt-1 (epoch <= 4kt-1) and its slot
        //        refilled from load interval 2kt (epoch >= 4kt).
        //   RAW  K tile kt+1 is first read in load interval 2kt+2 (epoch >= 4kt+4); every wave waited for its own pieces of
        //        it at the end of load interval 2kt+1 (epoch <= 4kt+3), i.e. before barrier 4kt+4.
        static_assert(QW && !P3X && !PARK && (VAR & LM_RFIRST) != 0, "fp8 x fp8: the product variant's switches");
        for (int kt = 0; kt < nk; ++kt) {
            const int b = kt & 1;
            const unsigned so = b * C::STAGE;
            const bool has1 = kt + 1 < nk, has2 = kt + 2 < nk;
            ksh = (kt & 3) * 8;
            // ---- phase A
            rd_w(so, I0{}, I4{});
            rd_x(so, I0{}, I4{}, I0{});
            rd_s(kt);
            __builtin_amdgcn_sched_barrier(0);
            if (has1) {
                dma2(b ^ 1, kt + 1, XP0{});
                dma2(b ^ 1, kt + 1, XP2{});
            }
            reads_done();
#pragma unroll
            for (int i = 0; i < 8; ++i) pin(fw[i]);
#pragma unroll
            for (int i = 0; i < 8; ++i) pin(fx[i]);
            pin_s();
            close_load();
            quad(I0{}, I4{}, I0{}, I4{}, I0{});
            __builtin_amdgcn_s_barrier();
            // ---- phase B
            rd_x(so, I4{}, I4{}, I0{});
            __builtin_amdgcn_sched_barrier(0);
            if (has2) {
                dma2(b, kt + 2, I0{});
                dma2(b, kt + 2, I2{});
            }
            reads_done();
#pragma unroll
            for (int i = 0; i < 8; ++i) pin(fx[i]);
            if (has2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(C::W_DMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            close_load();
            quad(I0{}, I4{}, I4{}, I4{}, I0{});
            __builtin_amdgcn_s_barrier();
        }
        } else {
        //   LM_RFIRST  the fragment reads of a load interval are issued BEFORE its DMA instructions: the ~70-cycle DMA
        //           issues then run under the LDS round trip of the reads instead of in front of it.
        constexpr bool RF = (VAR & LM_RFIRST) != 0;
        static_assert(!C::W8 || (QW && !P3X && !PARK), "fp8 weight operand: wired into the product quadrant loop");
        static_assert(!C::A8 || (QW && !P3X && !PARK), "fp8 x fp8: wired into the product quadrant loop");
#if GLM_LM_STAMP
        unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
        unsigned long long kt_t0 = __builtin_readcyclecounter();
#endif
        for (int kt = 0; kt < nk; ++kt) {
            const int b = kt & 1;
            const unsigned so = b * C::STAGE;
            // (CONT) with a next tile in hand the stream goes on past this tile's last K tile
            const bool has1 = kt + 1 < nk || (CONT && nxt), has2 = kt + 2 < nk || (CONT && nxt);
            auto mid = [&]() { __builtin_amdgcn_sched_barrier(0); };
            ksh = (kt & 3) * 8;
            // ---- phase 0
            auto dma0 = [&]() { if (!P3X && has1) dma2(b ^ 1, kt + 1, XP0{}); };
            LM_T(0);
            if (!RF) dma0();
            if constexpr (QW) {
                rd_w(so, I0{}, I4{});
                rd_x(so, I0{}, I2{}, I0{});
                rd_s(kt);
            } else {
                rd_w(so, I0{}, I2{});
                rd_x(so, I0{}, I4{}, I0{});
            }
            if (RF) { mid(); dma0(); }
            LM_T(1);  // 0: phase 0 read + DMA issue
            reads_done();
            LM_T(2);  // 1: phase 0 LDS wait
            if constexpr (QW) {
                cvt_w();
#pragma unroll
                for (int i = 0; i < 8; ++i) pin(fw[i]);
#pragma unroll
                for (int i = 0; i < 4; ++i) pin(fx[i]);
                pin_s();
            } else {
                pin(fw[0]); pin(fw[1]); pin(fw[4]); pin(fw[5]);
#pragma unroll
                for (int i = 0; i < 8; ++i) pin(fx[i]);
            }
            close_load();
            LM_T(3);  // 2: phase 0 barrier a
            if constexpr (QW) quad(I0{}, I4{}, I0{}, I2{}, I0{});
            else quad(I0{}, I2{}, I0{}, I4{}, I0{});
            LM_T(4);  // 3: phase 0 MFMA issue
            __builtin_amdgcn_s_barrier();
            LM_T(5);  // 4: phase 0 barrier b
            // ---- phase 1
            auto dma1 = [&]() { if (has1) dma2(b ^ 1, kt + 1, XP2{}); };
            if (!RF) dma1();
            if constexpr (QW) rd_x(so, I2{}, I2{}, I0{});
            else rd_w(so, I2{}, I2{});
            if (RF) { mid(); dma1(); }
            reads_done();
            if constexpr (QW) {
#pragma unroll
                for (int i = 0; i < 4; ++i) pin(fx[i]);
            } else {
                pin(fw[2]); pin(fw[3]); pin(fw[6]); pin(fw[7]);
            }
            close_load();
            if constexpr (QW) quad(I0{}, I4{}, I2{}, I2{}, I0{});
            else quad(I2{}, I2{}, I0{}, I4{}, I0{});
            __builtin_amdgcn_s_barrier();
            // ---- phase 2
            if constexpr (CONT) {
                if (nxt && kt == nk - 2) {
                    // X(nk - 1) has been issued (phases 0, 1): from here on every piece belongs to the next output tile —
                    // W(0') now and in phase 3, X(0') and W(1') from the last K tile
                    grp = n_grp;
                    m0 = n_m0;
                    mcnt = n_mcnt;
                    ntile = n_ntile;
                    pos = n_pos;
                    ti = n_ti;
                    setup_sources(std::true_type{});
                }
                if constexpr (MODE == GEMM_DOWN) {
                    if (nxt && kt == nk - 1) {
                        // the finished tile's per-row scales, in front of W(1'): they return before the counted wait of phase 3
                        const bool sc_on = grp_c < p.n_scaled;
#pragma unroll
                        for (int mt = 0; mt < C::MT; ++mt) {
                            const int row = wm * (C::BM / 2) + 16 * mt + c;
                            sc_pre[mt] = (sc_on && row < mcnt_c) ? bf2f(p.scale[m0_c + row]) : 1.0f;
                        }
                    }
                }
            }
            auto dma2_ = [&]() { if (has2) dma2(b, kt + 2, I0{}); };
            if (!RF) dma2_();
            if constexpr (QW) rd_x(so, I4{}, I2{}, I0{});
            else rd_x(so, I4{}, I4{}, I0{});
            if (RF) { mid(); dma2_(); }
            reads_done();
#pragma unroll
            for (int i = 0; i < (QW ? 4 : 8); ++i) pin(fx[i]);
            close_load();
            if constexpr (QW) quad(I0{}, I4{}, I4{}, I2{}, I0{});
            else quad(I2{}, I2{}, I4{}, I4{}, I0{});
            __builtin_amdgcn_s_barrier();
            // ---- phase 3: the last pieces, then this wave's share of K tile kt+1 must have landed
            auto dma3 = [&]() {
                if (has2) {
                    if constexpr (C::W_DMA > 2) dma2(b, kt + 2, I2{});  // (an fp8 weight tile is two pieces: both went in phase 2)
                    if constexpr (P3X) dma2(b, kt + 2, XP0{});
                }
            };
            if (!RF || !QW) dma3();
            if constexpr (QW) {
                rd_x(so, I6{}, I2{}, I0{});
                if (RF) { mid(); dma3(); }
                reads_done();
#pragma unroll
                for (int i = 0; i < 4; ++i) pin(fx[i]);
            }
            if (has2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(C::W_DMA + (P3X ? 2 : 0)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (PARK) {
                // behind the wait: the store is then older than every DMA piece the next counted waits leave in flight
                if (park_pending && kt < PARK_N) park_flush(kt);
            }
            close_load();
            if constexpr (QW) quad(I0{}, I4{}, I6{}, I2{}, I0{});
            else quad(I0{}, I2{}, I4{}, I4{}, I0{});
            __builtin_amdgcn_s_barrier();
            LM_T(6);  // 5: phases 1-3
#if GLM_LM_STAMP
            if (blockIdx.x == 8 && wid == 0) {
                const unsigned long long t1_ = __builtin_readcyclecounter();
                const int idx_ = kt < 6 ? kt : (nk - 1 - kt < 6 ? 11 - (nk - 1 - kt) : 12);
                if (lane == 0) {
                    g_lm_kt_stamps[idx_] += t1_ - kt_t0;
                    g_lm_kt_stamps[13 + idx_] += 1;
                }
                kt_t0 = t1_;
            }
#endif
        }
#if GLM_LM_STAMP
        if (blockIdx.x == 8 && lane == 0 && (wid == 0 || wid == 4)) {
#pragma unroll
            for (int i = 0; i < 7; ++i) g_lm_stamps[wid >> 2][i] = tacc[i];
            g_lm_stamps[wid >> 2][7] = (unsigned long long)nk;
        }
#endif
        }  // !LM_RPRE
        if (wid < 4) __builtin_amdgcn_s_barrier();  // match the other group's extra barrier
        } else {
#if GLM_LM_SCHED == 0
        if (wid >= 4) __builtin_amdgcn_s_barrier();   // the stagger
        int slot = 0;  // ring slot of tile kt
#if GLM_LM_STAMP
        unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
#endif
        for (int kt = 0; kt < nk; ++kt) {
            // ---------------- phase 1 ----------------
            const unsigned sb = base + slot * C::STAGE;
            u32x4_t f[C::NF];
            u32x2_t fr[C::W8 ? C::KS * 4 : 1];
            const int nslot = slot == 0 ? C::NSTAGE - 1 : slot - 1;  // slot of tile kt-1
            LM_T(0);
            if constexpr ((VAR & LM_LOADPRIO) != 0) __builtin_amdgcn_s_setprio(1);
            if (S2 > 0) {
                if (kt + C::DIST < nk)
                    static_for<C::DPW - S2>([&](auto jc) { stage_one(nslot, kt + C::DIST, jc); });
            } else if (GLM_LM_STAGE_POS == 0 && kt + C::DIST < nk) {
                stage(nslot, kt + C::DIST);
            }
            LM_T(1);  // 0: DMA issue
            if (LEAN) {
                read_frags(f, fr, slot * C::STAGE, std::integral_constant<int, 0>{});
            } else {
#pragma unroll
                for (int i = 0; i < C::NF / 2; ++i) lds_read_b128(f[i], sb + rel[i]);
                if (GLM_LM_STAGE_POS == 1 && kt + C::DIST < nk) stage(nslot, kt + C::DIST);
#pragma unroll
                for (int i = C::NF / 2; i < C::NF; ++i) lds_read_b128(f[i], sb + rel[i]);
            }
            if (GLM_LM_STAGE_POS == 2 && kt + C::DIST < nk) stage(nslot, kt + C::DIST);
            LM_T(2);  // 1: LDS read issue
            // this wave's share of tile kt+1: leave only the tiles newer than kt+1 in flight
            if (S2 > 0) {
                // issued so far beyond tile kt+1: whole tiles up to kt+DIST-1, and the load-phase part of tile kt+DIST
                const int full = max(0, min(kt + C::DIST - 1, nk - 1) - (kt + 1));
                const int leave = full * C::DPW + (kt + C::DIST < nk ? C::DPW - S2 : 0);
                if (leave == 2 * C::DPW - S2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * C::DPW - S2) : "memory");
                else if (leave == C::DPW) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(C::DPW) : "memory");
                else if (leave == C::DPW - S2 && S2 < C::DPW) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(C::DPW - S2) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                dma_wait_leave<C::DPW>(min(kt + C::DIST, nk - 1) - (kt + 1));
            }
            LM_T(3);  // 2: counted DMA wait
            lds_wait(f);
            frags_cvt(f, fr);
            LM_T(4);  // 3: LDS wait
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((VAR & LM_LOADPRIO) != 0) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();  // B1
            LM_T(5);  // 4: barrier 1
            // ---------------- phase 2 ----------------
            if constexpr ((VAR & (LM_NOPRIO | LM_LOADPRIO)) == 0) __builtin_amdgcn_s_setprio(1);
            if constexpr (S2 > 0) {
                // the same MFMAs in the same order, with the remaining S2 DMA instructions of tile kt+DIST spread
                // between them (pinned: nothing moves across a sched_barrier)
                constexpr int NM = C::KS * 4 * C::MT;
                const bool more_dma = kt + C::DIST < nk;
                static_for<NM>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    constexpr int ks = i / (4 * C::MT), nt = (i / C::MT) % 4, mt = i % C::MT;
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8_t, f[ks * (4 + C::MT) + nt]),
                        __builtin_bit_cast(bf16x8_t, f[ks * (4 + C::MT) + 4 + mt]), acc[nt][mt], 0, 0, 0);
                    // after MFMA number (j+1) * NM / (S2+1), j = 0 .. S2-1
                    static_for<S2>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        if constexpr (i + 1 == (j + 1) * NM / (S2 + 1)) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (more_dma) stage_one(nslot, kt + C::DIST, std::integral_constant<int, C::DPW - S2 + j>{});
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    });
                });
            } else {
#pragma unroll
                for (int ks = 0; ks < C::KS; ++ks)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int mt = 0; mt < C::MT; ++mt)
                            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8_t, f[ks * (4 + C::MT) + nt]),
                                __builtin_bit_cast(bf16x8_t, f[ks * (4 + C::MT) + 4 + mt]), acc[nt][mt], 0, 0, 0);
            }
            if constexpr ((VAR & (LM_NOPRIO | LM_LOADPRIO)) == 0) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            LM_T(6);  // 5: MFMA issue
            __builtin_amdgcn_s_barrier();  // B2
            LM_T(7);  // 6: barrier 2
            slot = slot == C::NSTAGE - 1 ? 0 : slot + 1;
        }
#if GLM_LM_STAMP
        if (blockIdx.x == 8 && lane == 0 && (wid == 0 || wid == 4)) {
#pragma unroll
            for (int i = 0; i < 7; ++i) g_lm_stamps[wid >> 2][i] = tacc[i];
            g_lm_stamps[wid >> 2][7] = (unsigned long long)nk;
        }
#endif
        if (wid < 4) __builtin_amdgcn_s_barrier();  // match the other group's extra barrier

#else
        // ---- one barrier per K tile.  Every wave runs the same stream — fragment reads of tile kt (+ DMA issue of tile
        // kt+DIST, counted wait for its share of tile kt+1), then the tile's MFMAs — but waves 0-3 take the barrier
        // BETWEEN the two phases and waves 4-7 AFTER the MFMAs.  Between two barriers waves 0-3 therefore run
        // MFMA(kt-1) then the reads of kt while waves 4-7 run the reads of kt then MFMA(kt): the two waves of a
        // SIMD (w, w+4) start every interval in opposite roles and swap half way without a barrier to wait at.
        //   RAW  tile kt+1 is read in the interval after barrier kt; every wave waited for its share before taking it.
        //   WAR  tile kt+DIST overwrites the slot of tile kt-1, read by all waves before barrier kt-1; it is issued
        //        after that barrier.
        const bool lead = wid < 4;
        int slot = 0;  // ring slot of tile kt
        for (int kt = 0; kt < nk; ++kt) {
            const unsigned sb = base + slot * C::STAGE;
            u32x4_t f[C::NF];
            u32x2_t fr[C::W8 ? C::KS * 4 : 1];
            const int nslot = slot == 0 ? C::NSTAGE - 1 : slot - 1;  // slot of tile kt-1
            if (kt + C::DIST < nk) stage(nslot, kt + C::DIST);
            if (LEAN) {
                read_frags(f, fr, slot * C::STAGE, std::integral_constant<int, 0>{});
            } else {
#pragma unroll
                for (int i = 0; i < C::NF; ++i) lds_read_b128(f[i], sb + rel[i]);
            }
            dma_wait_leave<C::DPW>(min(kt + C::DIST, nk - 1) - (kt + 1));
            lds_wait(f);
            frags_cvt(f, fr);
            __builtin_amdgcn_sched_barrier(0);
            if (lead) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < C::MT; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8_t, f[ks * (4 + C::MT) + nt]),
                            __builtin_bit_cast(bf16x8_t, f[ks * (4 + C::MT) + 4 + mt]), acc[nt][mt], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (!lead) __builtin_amdgcn_s_barrier();
            slot = slot == C::NSTAGE - 1 ? 0 : slot + 1;
        }
        // waves 4-7 are past every LDS read of this tile here, waves 0-3 finished theirs before the last barrier
#endif
        }  // !C::PH8
#undef rd_s
#undef pin_s

        // per-row scales of the finished tile: requested before the ring fill (waiting for a load issued after the fill
        // would drain the fill first)
        const bool cont_out = CONT && nxt;  // the next tile's K tiles 0 and 1 are in flight already
        if constexpr ((VAR & LM_EPIORD) != 0 && MODE == GEMM_DOWN) {
            if (!cont_out) {
            const bool sc_on = grp_c < p.n_scaled;
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt) {
                const int row = wm * (C::BM / 2) + 16 * mt + c;
                sc_pre[mt] = (sc_on && row < mcnt_c) ? bf2f(p.scale[m0_c + row]) : 1.0f;
            }
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt) asm volatile("" : "+v"(sc_pre[mt]));  // the loads stay above the fill
            }
        }
        // ---- next tile: every wave is past its last LDS read, so the ring can be refilled right away ----
        LM_TS(2);  // 1: K loop
        bool more = true;
        if (!cont_out) {
            pos += pstep;
            ++ti;
            more = find_tile();
        }
        LM_TS(3);  // 2: tile search
        // A8 (r05): the epilogue of the fp8 x fp8 kernels LOADS (weight-row scales, token scales) and vector-memory results
        // return in issue order — behind the ring fill they cost a whole fill latency per tile (stamps of the first form:
        // epilogue 23.5k cycles against 6.0k for bf16, and the late stores then drain into the next tile's short K loop:
        // profiles/r05/largem_stamps_fp8a8_first_form.txt).  There the next tile's fill is issued once the first 16-row
        // block of the epilogue has its loads back, i.e. behind them and in front of the bulk of the stores.
        // (r05, also measured: the scale vectors as LDS-DMA pieces of the ring fill, no global load in the epilogue, the fill in
        // front of it as in the bf16 kernel — pair 1817 against 1819 TFLOP/s, profiles/r05/fp8a8_probe_lds_scales_form.txt: the
        // dependent loads are not what a tile switch waits for either; not kept.)
        constexpr bool LATE_FILL = C::A8;
        if (more && !cont_out) {
            setup_sources(std::false_type{});   // (its own loads — the gathered source rows — go out here, in front of the epilogue's)
            LM_TS(4);  // 3: source set-up
        }
        auto start_next = [&]() {
            if (more && !cont_out) {
                fill_ring();
                LM_TS(5);  // 4: ring fill issue
            }
        };
        if constexpr (!LATE_FILL) start_next();

        // ---- epilogue of the finished tile ----
        constexpr bool WIDE = (VAR & LM_WIDEST) != 0, NOSTORE = (VAR & LM_NOSTORE) != 0;
        constexpr int NOUT = MODE == GEMM_GATEUP ? 2 : 4;  // 16-column output tiles per wave
        const int ncol_w = MODE == GEMM_GATEUP ? ntile_c * 128 + wn * 32 : ntile_c * LM_BN + wn * 64;
        const bool scaled = MODE == GEMM_DOWN && grp_c < p.n_scaled;
        bool qkv_epi = false;
        if constexpr (MODE == GEMM_PLAIN && WIDE && !WSCALE) qkv_epi = p.qkv_q != nullptr;
        const bool do_park = PARK && more && nk >= PARK_N && !qkv_epi;  // the next tile's K loop has a K tile for every parked store
        if (qkv_epi) {
            // ---- QKV epilogue (r04): bias, bf16 rounding (the packed qkv value of the reference, linear.py:75), rotary on the
            // first 64 dims of every q / k head, split, cache scatter.  A wave's 64 columns are one half of one head (ncol_w
            // is a multiple of 64): the rotary half pairs column i with i + 32 = accumulator tiles t and t + 2 in the SAME
            // lane and register, so the rotation is lane-local.  Arithmetic and rounding points are qkv_post_kernel's
            // (rope8: products and the difference / sum rounded separately, -ffp-contract=off): bit-identical outputs.
            const int head = ncol_w >> 7, half = (ncol_w >> 6) & 1;
            const int nh = p.qkv_nh, nkv = p.qkv_nkv;
            const bool is_q = head < nh, is_k = !is_q && head < nh + nkv;
            const bool rot = half == 0 && (is_q || is_k);
            const int kh = is_q ? head : (is_k ? head - nh : head - nh - nkv);
            bf16_t* dense = is_q ? p.qkv_q : (is_k ? p.qkv_k : p.qkv_v);
            bf16_t* cache = is_q ? nullptr : (is_k ? p.qkv_kc : p.qkv_vc);
            const int64_t ldd = (int64_t)(is_q ? nh : nkv) * 128;
            const bool col_ok = ncol_w < p.N;
            u32x2_t bias_p[4];  // four consecutive bias values per tile, packed (N is a whole number of 64-column blocks)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                bias_p[t] = (p.bias && col_ok) ? *reinterpret_cast<const u32x2_t*>(p.bias + ncol_w + 16 * t + 4 * g4) : u32x2_t{0u, 0u};
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt) {
                const int row = wm * (C::BM / 2) + 16 * mt + c;
                const bool row_ok = row < mcnt_c && col_ok;
                const int64_t m = m0_c + (row_ok ? row : 0);
                float o[4][4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    o[t][0] = rbf(acc[t][mt][0] + bflo(bias_p[t][0]));
                    o[t][1] = rbf(acc[t][mt][1] + bfhi(bias_p[t][0]));
                    o[t][2] = rbf(acc[t][mt][2] + bflo(bias_p[t][1]));
                    o[t][3] = rbf(acc[t][mt][3] + bfhi(bias_p[t][1]));
                }
                if (rot) {  // wave-uniform
                    const float* cs = p.qkv_cs + p.qkv_pos[m] * 64;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const f32x4_t cv = *reinterpret_cast<const f32x4_t*>(cs + 16 * t + 4 * g4);
                        const f32x4_t sv = *reinterpret_cast<const f32x4_t*>(cs + 32 + 16 * t + 4 * g4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float x1 = o[t][r], x2 = o[t + 2][r];
                            o[t][r] = x1 * cv[r] - x2 * sv[r];
                            o[t + 2][r] = x2 * cv[r] + x1 * sv[r];
                        }
                    }
                }
                u32x2_t v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = u32x2_t{pack2bf(o[t][0], o[t][1]), pack2bf(o[t][2], o[t][3])};
                int64_t slot = -1;
                if (cache != nullptr && p.qkv_slots != nullptr && row_ok) slot = p.qkv_slots[m];
                const bool to_cache = slot >= 0 && slot < p.qkv_nslots;
#pragma unroll
                for (int t = 0; t < 4; t += 2) {
                    const u32x2_t a = __builtin_amdgcn_permlane16_swap(v[t][0], v[t + 1][0], false, false);
                    const u32x2_t b = __builtin_amdgcn_permlane16_swap(v[t][1], v[t + 1][1], false, false);
                    const int cw = 64 * half + 16 * (t + (g4 & 1)) + 8 * (g4 >> 1);  // column inside the head
                    const u32x4_t o4 = u32x4_t{a[0], b[0], a[1], b[1]};
                    if (row_ok) {
                        if (dense != nullptr) *reinterpret_cast<u32x4_t*>(dense + m * ldd + kh * 128 + cw) = o4;
                        if (to_cache) *reinterpret_cast<u32x4_t*>(cache + (slot * nkv + kh) * 128 + cw) = o4;
                    }
                }
            }
        } else if constexpr (C::A8 && MODE == GEMM_GATEUP) {
            // ---- fp8 x fp8 gate|up (r05): the accumulator x weight-row scale x token scale is the bf16 GEMM's value up to the
            // fp8 roundings of its inputs; same rounding points behind it (mlp.py:49-53: bf16 gate_up, bf16 silu, bf16
            // product).  The product is then quantised for the down projection's fp8 operand in MX form, decided locally:
            // a wave's 32 act columns of a token ARE one 32-element block (ncol_w is a multiple of 32) — amax over the lane's 8
            // values and the 3 other lane rows of the token, E8M0 scale 2^(floor(log2 amax) - 8) (OCP MX: e4m3 emax = 8) or,
            // where that would push amax past 448, the next power of two (nothing saturates), e4m3 = RNE(v / scale).  act8 [R, I] bytes + one scale byte per (row, block) at
            // (block & 3) * 16 + (block >> 2) of the row's 64-byte record — the order the down kernel's lanes read it in.
            const float* ws = p.wscale + (int64_t)grp_c * p.wscale_gstride;
            float wsg[2][4], wsu[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n1 = min(ncol_w + 16 * t + 4 * g4 + r, p.N - 1);
                    wsg[t][r] = ws[n1];
                    wsu[t][r] = ws[p.N + n1];
                }
            unsigned char* y8 = reinterpret_cast<unsigned char*>(p.y);
            const bool col_ok = ncol_w < p.N;
            float xs_all[C::MT];   // token scales of the wave's 8 row blocks: all requested in front of the ring fill
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt) {
                const int row = wm * (C::BM / 2) + 16 * mt + c;
                const int64_t m = m0_c + ((row < mcnt_c && col_ok) ? row : 0);
                xs_all[mt] = p.xscale[p.src ? p.src[m] : m];
            }
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt) {
                const int row = wm * (C::BM / 2) + 16 * mt + c;
                const bool row_ok = row < mcnt_c && col_ok;
                const int64_t m = m0_c + (row_ok ? row : 0);
                const float xs = xs_all[mt];
                float v[2][4];
                float amax = 0.f;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float ga = acc[t][mt][r] * wsg[t][r] * xs, ua = acc[t + 2][mt][r] * wsu[t][r] * xs;
                        v[t][r] = rbf(rbf(glm_silu(rbf(ga))) * rbf(ua));
                        amax = fmaxf(amax, fabsf(v[t][r]));
                    }
                amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
                amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
                const unsigned ab = __float_as_uint(amax);
                const int eb = (int)((ab >> 23) & 0xffu);   // biased exponent (0: zero or subnormal block)
                // 2^(floor(log2 amax) - 8) maps amax into [256, 512); above 448 = 1.75 x 2^8 it would saturate: one step up then
                const int sb = eb == 0 ? 127 : min(max(eb - 8 + ((ab & 0x7fffffu) > 0x600000u ? 1 : 0), 1), 253);
                const float inv = __uint_as_float((unsigned)(254 - sb) << 23);    // 2^(127 - sb), exact
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float q[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) q[r] = fminf(fmaxf(v[t][r] * inv, -448.0f), 448.0f);
                    int d = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], 0, false);
                    d = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], d, true);
                    const int n = ncol_w + 16 * t + 4 * g4;
                    if (row_ok && !NOSTORE && n < p.N) *reinterpret_cast<int*>(y8 + m * p.ldy + n) = d;
                }
                if (row_ok && !NOSTORE && g4 == 0) p.ybscale[m * 64 + wn * 16 + ntile_c] = (unsigned char)sb;
                if (mt == 0) start_next();
            }
        } else {
        // weight-row scales of this wave's output columns: loaded ONCE, in front of the row blocks (and, for the fp8 x fp8
        // kernels, in front of the next tile's ring fill)
        float wsv[WSCALE && MODE != GEMM_GATEUP ? 4 : 1][4];
        if constexpr (WSCALE && MODE != GEMM_GATEUP) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    wsv[t][r] = p.wscale[(int64_t)grp_c * p.wscale_gstride + min(ncol_w + 16 * t + 4 * g4 + r, p.N - 1)];
        }
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) {
            if constexpr (LATE_FILL) {
                if (mt == 1) start_next();
            }
            const int row = wm * (C::BM / 2) + 16 * mt + c;
            const bool row_ok = row < mcnt_c;
            bf16_t* yrow = p.y + (int64_t)(m0_c + (row_ok ? row : 0)) * p.ldy;
            float sc;
            if constexpr ((VAR & LM_EPIORD) != 0 && MODE == GEMM_DOWN) sc = sc_pre[mt];
            else sc = (scaled && row_ok) ? bf2f(p.scale[m0_c + row]) : 1.0f;
            u32x2_t v[NOUT];  // packed columns 4*g4 .. 4*g4+3 of output tile t
#pragma unroll
            for (int t = 0; t < NOUT; ++t) {
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (MODE == GEMM_GATEUP) {
                        float ga = acc[t][mt][r], ua = acc[t + 2][mt][r];
                        if constexpr (WSCALE) {  // weights expanded from fp8: per-row scale on the fp32 accumulators (uniform branch)
                            const float* ws = p.wscale + (int64_t)grp_c * p.wscale_gstride;
                            const int n1 = min(ncol_w + 16 * t + 4 * g4 + r, p.N - 1);
                            ga *= ws[n1];
                            ua *= ws[p.N + n1];
                        }
                        o[r] = rbf(glm_silu(rbf(ga))) * rbf(ua);
                    } else {
                        float a = acc[t][mt][r];
                        if constexpr (WSCALE) a *= wsv[t][r];
                        if (MODE == GEMM_PLAIN && p.bias) a += bf2f(p.bias[min(ncol_w + 16 * t + 4 * g4 + r, p.N - 1)]);
                        if (scaled) a = rbf(a) * sc;
                        o[r] = a;
                    }
                }
                v[t] = u32x2_t{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
            }
            if (NOSTORE) {
#pragma unroll
                for (int t = 0; t < NOUT; ++t) asm volatile("" ::"v"(v[t][0]), "v"(v[t][1]));
                continue;
            }
            if constexpr ((VAR & LM_LDSEPI) != 0 && NOUT == 4) {
                // 16 rows x 64 columns of this wave through its private 4 KiB patch in the ring slot that the next
                // tile's fill does not touch (slot NSTAGE-1; fill_ring uses 0 .. DIST-1): lane (c, g4) writes its four
                // 8-byte pieces of row c, then 8 lanes read one whole 128-byte row.  Row pitch 144 B (2-way write
                // conflicts at most).  LDS instructions of one wave execute in order, so neither the read after the
                // writes nor the next block's writes after these reads need a wait; the stores wait for the reads.
                const unsigned patch = base + (C::NSTAGE - 1) * C::STAGE + wid * 4096;
                const unsigned wa = patch + c * 144 + g4 * 8;
                asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:32\n\t"
                             "ds_write_b64 %0, %3 offset:64\n\tds_write_b64 %0, %4 offset:96"
                             :: "v"(wa), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "memory");
                const unsigned ra = patch + (lane >> 3) * 144 + (lane & 7) * 16;
                u32x4_t r0, r1;
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1152\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(r0), "=&v"(r1) : "v"(ra) : "memory");
                const int rrow = wm * (C::BM / 2) + 16 * mt + (lane >> 3);
                const int n = ncol_w + 8 * (lane & 7);
                if (n < p.N) {
                    if (rrow < mcnt_c) *reinterpret_cast<u32x4_t*>(p.y + (int64_t)(m0_c + rrow) * p.ldy + n) = r0;
                    if (rrow + 8 < mcnt_c) *reinterpret_cast<u32x4_t*>(p.y + (int64_t)(m0_c + rrow + 8) * p.ldy + n) = r1;
                }
                continue;
            }
            if (WIDE) {
                // tiles (t, t+1): after the swaps lane row g4 holds 8 consecutive columns of tile t + (g4 & 1),
                // starting at column 8 * (g4 >> 1)
#pragma unroll
                for (int t = 0; t < NOUT; t += 2) {
                    const u32x2_t a = __builtin_amdgcn_permlane16_swap(v[t][0], v[t + 1][0], false, false);
                    const u32x2_t b = __builtin_amdgcn_permlane16_swap(v[t][1], v[t + 1][1], false, false);
                    const int n = ncol_w + 16 * (t + (g4 & 1)) + 8 * (g4 >> 1);
                    if constexpr (PARK) {
                        const int sidx = mt * (NOUT / 2) + t / 2;
                        if (do_park && sidx >= LM_STORES_FULL - PARK_N) {
                            const u32x4_t o4 = u32x4_t{a[0], b[0], a[1], b[1]};
                            asm volatile("ds_write_b128 %0, %1" ::"v"(park_addr + (sidx - (LM_STORES_FULL - PARK_N)) * 1024), "v"(o4) : "memory");
                            continue;
                        }
                    }
                    if (row_ok && n < p.N) {
                        const u32x4_t o4 = u32x4_t{a[0], b[0], a[1], b[1]};
                        if constexpr ((VAR & LM_NT) != 0) __builtin_nontemporal_store(o4, reinterpret_cast<u32x4_t*>(yrow + n));
                        else *reinterpret_cast<u32x4_t*>(yrow + n) = o4;
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < NOUT; ++t) {
                    const int n = ncol_w + 16 * t + 4 * g4;
                    if (row_ok && n < p.N) *reinterpret_cast<u32x2_t*>(yrow + n) = v[t];
                }
            }
        }
        }
        if constexpr ((VAR & LM_EPIORD) != 0) {
            // exact only when every store instruction of the wave ran: all rows and columns of the tile exist
            const bool full = !LATE_FILL && mcnt_c == C::BM && ncol_w + (MODE == GEMM_GATEUP ? 32 : 64) <= p.N && !NOSTORE &&
                              !((VAR & LM_LDSEPI) != 0 && NOUT == 4) && !(MODE == GEMM_PLAIN && p.qkv_q != nullptr);
            pending_stores = full ? LM_STORES_FULL - (do_park ? PARK_N : 0) : 0;
        }
        if constexpr (PARK) {
            park_pending = do_park;
            park_m0 = m0_c;
            park_mcnt = mcnt_c;
            park_ncolw = ncol_w;
        }
        LM_TS(6);  // 5: epilogue
#if GLM_LM_STAMP
        ts_acc[7] += 1;
        if (!more && blockIdx.x == 8 && tid == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) g_lm_tile_stamps[i] = ts_acc[i];
        }
#endif
        if (!more) break;
        cont_in = cont_out;
    }
}

int g_largem_variant = -1;  // < 0: the default for the shape

// option largem_variant (environment GLM_LARGEM_VARIANT at start-up, glm_set_option afterwards) makes a variant the
// process default; consulted at every launch, so a test can switch between variant 0 and the product loop inside one process
int env_variant() { return glm_option(GLM_OPT_LARGEM_VARIANT); }

template <int MODE, class C, int VAR>
int launch_largem_cfg(const GemmParams& p, hipStream_t st) {
    const int cols_per_wg = MODE == GEMM_GATEUP ? 128 : LM_BN;
    const int n_tiles = (p.N + cols_per_wg - 1) / cols_per_wg;
    const int64_t combos = p.offs ? (p.R / C::BM + p.G) : (p.R + C::BM - 1) / C::BM;
    const int LM_BAND = g_largem_band;
    const int64_t total = (combos + LM_BAND - 1) / LM_BAND * LM_BAND * n_tiles;
    if (total <= 0) return GLM_OK;
    GLM_CHECK_ARG(total < (1ll << 31), GLM_ERR_UNSUPPORTED, "gemm_largem: grid too large");
    // the per-row scale of fp8-expanded weights is its own instantiation: as a run-time test it sat inside the epilogue's
    // element loop (a branch per accumulator register)
    auto kern = (C::W8 || C::A8 || p.wscale) ? gemm_largem_kernel<MODE, C, VAR, true> : gemm_largem_kernel<MODE, C, VAR, !(C::W8 || C::A8) ? false : true>;
    constexpr int LDS_BYTES = (C::PH8 && (VAR & LM_PARK) != 0) ? 160 * 1024 : C::LDS + ((C::A8 && MODE == GEMM_DOWN) ? 16 * 1024 : 0);
    GLM_SET_MAX_LDS(kern, LDS_BYTES, "gemm_largem");
    // one workgroup per CU (the ring takes most of the LDS), a multiple of 8 so that every XCD gets the same count
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            glm_set_error("gemm_largem: hipGetDeviceProperties failed");
            return GLM_ERR_LAUNCH;
        }
        n_cu = prop.multiProcessorCount;
    }
    int64_t nwg = (n_cu / 8) * 8;
    if (nwg < 8) nwg = 8;
    const int64_t need = (total + 7) / 8 * 8;
    if (nwg > need) nwg = need;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(512), LDS_BYTES, st, p, n_tiles, (int)total, LM_BAND);
    GLM_CHECK_LAUNCH("gemm_largem");
    return GLM_OK;
}

// option largem_cfg (GLM_LARGEM_CFG=A|B at start-up) overrides the tile configuration (A/B measurements)
int cfg_choice() { return glm_option(GLM_OPT_LARGEM_CFG); }

template <int MODE>
int launch_largem(const GemmParams& p, hipStream_t st) {
    // 256-row blocks pay off once a group fills most of one; smaller groups keep the 128-row tile
    const int ch = cfg_choice();
    const int64_t avg_rows = (p.R_hint > 0 ? p.R_hint : p.R) / (p.G > 0 ? p.G : 1);
    // ... and a projection onto <= 128 columns (the router logits at prefill: N = E = 64) is one column tile however the rows
    // are cut, so its parallelism is its row-block count: 128-row blocks (75.8 -> 55.1 us at T = 16384)
    const bool use_b = ch == 2 || (ch == 0 && avg_rows >= 192 && !(MODE == GEMM_PLAIN && p.offs == nullptr && p.N <= 128));
    // the lean load phase addresses x and the group's weights through 32-bit buffer offsets and has no K tail path
    // default (tools/largem_ab.py, one process, T = 16384): the quadrant-phase loop with 4 x 2-tile quadrants and the
    // fragment reads of a load interval issued before its DMA, no priority flips (86531: +1 % over 86019), epilogue
    // ordering (86787: down +1 %, gate|up unchanged)
    int var = g_largem_variant >= 0 ? g_largem_variant : env_variant() >= 0 ? env_variant() : GLM_LM_DEFAULT_VAR;
    if (var < 0) var = LM_LEAN | LM_WIDEST | LM_PH8 | LM_QW | LM_RFIRST | LM_NOPRIO | LM_EPIORD;
    const int bk = use_b ? ((var & LM_PH8) ? LmCfgC::BK : LmCfgB::BK) : LmCfgA::BK;
    const int64_t wrows = MODE == GEMM_GATEUP ? 2 * (int64_t)p.N : (int64_t)p.N;
    const bool lean_ok = p.K % bk == 0 && p.R * p.ldx * 2 < (1ll << 31) && wrows * p.ldw * 2 < (1ll << 31) &&
                         p.N % 8 == 0 && p.ldy % 8 == 0 && glm_aligned16(p.y);
    if (p.w_fp8) {
        // fp8 weight operand: the product loops only (quadrant loop on 256-row blocks, its plain form on 128-row blocks);
        // what they do not cover (a K tail, offsets past 2^31) goes back to the caller, which expands the stack instead
        constexpr int V8 = LM_LEAN | LM_WIDEST | LM_PH8 | LM_QW | LM_RFIRST | LM_NOPRIO | LM_EPIORD;
        if (!lean_ok || p.K % 64 != 0 || p.wscale == nullptr || p.ldw % 16 != 0 || p.w_gstride % 16 != 0) {
            glm_set_error("gemm_largem(fp8 weights): shape not covered (K=%d N=%d ldw=%ld)", p.K, p.N, (long)p.ldw);
            return GLM_ERR_UNSUPPORTED;
        }
        if constexpr (MODE == GEMM_PLAIN) {
            glm_set_error("gemm_largem(fp8 weights): only the grouped expert projections are quantised");
            return GLM_ERR_UNSUPPORTED;
        } else {
            if (use_b) return launch_largem_cfg<MODE, LmCfgC8, V8>(p, st);
            return launch_largem_cfg<MODE, LmCfgA8, V8 & ~LM_PH8>(p, st);
        }
    }
    if (!lean_ok) var = 0;
    // the QKV epilogue does all its stores itself: a variant that parks part of a tile's stores for the next tile's K loop
    // would flush never-written LDS through p.y (= qkv_q, another row pitch) — refused, not silently wrong
    if (p.qkv_q != nullptr && ((var & LM_WIDEST) == 0 || (var & LM_PARK) != 0 || p.wscale != nullptr)) {  // 16-byte-store form, bf16 weights only
        glm_set_error("gemm_largem(qkv epilogue): loop variant %d has no such epilogue", var);
        return GLM_ERR_UNSUPPORTED;
    }
#define GLM_LM_CASE(V)                                                                                         \
    case V:                                                                                                    \
        if (use_b && ((V) & LM_DEEP)) return launch_largem_cfg<MODE, LmCfgB5, V>(p, st);                       \
        if constexpr (((V) & LM_PH8) != 0) {                                                                   \
            if (use_b) return launch_largem_cfg<MODE, LmCfgC, V>(p, st);                                       \
            return launch_largem_cfg<MODE, LmCfgA, (V) & ~LM_PH8>(p, st);                                      \
        } else                                                                                                 \
            return use_b ? launch_largem_cfg<MODE, LmCfgB, V>(p, st) : launch_largem_cfg<MODE, LmCfgA, V>(p, st)
    switch (var) {
        GLM_LM_CASE(86787);  // THE product loop: quadrant phases (LM_QW | LM_RFIRST | LM_NOPRIO) with LM_EPIORD
#if GLM_LM_STUDY  // the steps that led there (profiles/r02/largem_ab_*.json); study builds only
        GLM_LM_CASE(611075); // 86787 with LM_CONT (r05): bit-identical, 2-3 % slower (profiles/r05/largem_cont_*.txt)
        GLM_LM_CASE(86851);  // 86787 with LM_DESYNC (r05: -DGLM_LM_DESYNC_MODE=1 = every workgroup its own phase; slower)
        GLM_LM_CASE(3);
        GLM_LM_CASE(4099);  // lean + wide + quadrant phases on two 64-KiB buffers (CfgC)
        GLM_LM_CASE(12291); // ... with the DMA of phase 0 moved to phase 3
        GLM_LM_CASE(20483); // ... with 4 x 2-tile quadrants
        GLM_LM_CASE(53251); // ... and the reads of phases 1-3 issued from the previous MFMA interval
        GLM_LM_CASE(69635); // quadrant phases, reads before DMA in every load interval
        GLM_LM_CASE(77827); // ... with LM_P3X
        GLM_LM_CASE(86019); // ... with LM_QW
        GLM_LM_CASE(86531);  // LM_QW | LM_RFIRST without the MFMA-interval priority
        GLM_LM_CASE(87043);  // ... with the LOAD interval at priority 1
#endif
#if GLM_LM_STUDY >= 2
        GLM_LM_CASE(135171); // quadrant phases, mirrored quadrant order, reads 8/4/8/4 (no gain: profiles/r02)
        GLM_LM_CASE(348931); // 86787 with LM_PARK (bit-identical, 1.5-2.5 % slower: profiles/r02/largem_ab_ph8_park.json)
#endif
#if GLM_LM_STUDY >= 2  // measured and not kept (profiles/r02/largem_ab_*.json)
        GLM_LM_CASE(1);
        GLM_LM_CASE(2);
        GLM_LM_CASE(11);   // lean + wide + 2 DMA in the MFMA phase
        GLM_LM_CASE(19);   // lean + wide + 4 DMA in the MFMA phase
        GLM_LM_CASE(2051); // lean + wide + 5-stage ring
        GLM_LM_CASE(259);  // lean + wide + epilogue ordering
        GLM_LM_CASE(771);  // ... without the MFMA-phase priority
        GLM_LM_CASE(1283); // ... with the LOAD phase at priority 1
        GLM_LM_CASE(67);   // lean + wide + XCD desync
        GLM_LM_CASE(131);  // lean + wide + nt stores
        GLM_LM_CASE(195);  // lean + wide + desync + nt
        GLM_LM_CASE(35);   // lean + wide + LDS-transposed epilogue
        GLM_LM_CASE(43);   // + 2 DMA in the MFMA phase
        GLM_LM_CASE(51);   // + 4 DMA in the MFMA phase
#endif
#if GLM_LM_STUDY
        GLM_LM_CASE(4);
        GLM_LM_CASE(7);
        GLM_LM_CASE(4103);  // quadrant phases, no stores
#endif
        default:
            GLM_LM_CASE(0);
    }
#undef GLM_LM_CASE
}

}  // namespace

// fp8 x fp8 grouped expert projections on the block-scaled MFMA (r05; BASELINE configs[4]): the product quadrant loop on
// 256-row blocks.  Strides of x / w in BYTES.
int glm_launch_gemm_largem_a8(int mode, const GemmParams& p, hipStream_t st) {
    constexpr int V88 = LM_LEAN | LM_WIDEST | LM_PH8 | LM_QW | LM_RFIRST | LM_NOPRIO | LM_EPIORD;
    const int64_t avg_rows = (p.R_hint > 0 ? p.R_hint : p.R) / (p.G > 0 ? p.G : 1);
    const int64_t wrows = mode == GEMM_GATEUP ? 2 * (int64_t)p.N : (int64_t)p.N;
    const bool ok = (mode == GEMM_GATEUP || mode == GEMM_DOWN) && p.a8 && p.wscale != nullptr && p.offs != nullptr && avg_rows >= 192 &&
                    p.K % 128 == 0 && p.N % 8 == 0 && p.ldx % 16 == 0 && p.ldw % 16 == 0 && p.w_gstride % 16 == 0 &&
                    p.R * p.ldx < (1ll << 31) && wrows * p.ldw < (1ll << 31) && glm_aligned16(p.x) && glm_aligned16(p.w) &&
                    (mode == GEMM_GATEUP ? (p.xscale != nullptr && p.ybscale != nullptr && p.ldy % 4 == 0 && (((uintptr_t)p.y) & 3) == 0 &&
                                            p.N % 32 == 0 && p.N <= 2048)
                                         : (p.xbscale != nullptr && glm_aligned16(p.xbscale) && p.K <= 2048 && p.ldy % 8 == 0 && glm_aligned16(p.y)));
    if (!ok) {
        glm_set_error("gemm_largem(fp8 x fp8): shape not covered (mode %d R=%ld G=%d N=%d K=%d)", mode, (long)p.R, p.G, p.N, p.K);
        return GLM_ERR_UNSUPPORTED;
    }
    if (mode == GEMM_GATEUP) return launch_largem_cfg<GEMM_GATEUP, LmCfgC88, V88>(p, st);
    return launch_largem_cfg<GEMM_DOWN, LmCfgC88, V88>(p, st);
}

// QKV projection with the split / rotary / cache-store epilogue: the product loop on the tiled configurations only
int glm_launch_gemm_largem_qkv(const GemmParams& p, hipStream_t st) {
    const int Nexp = (p.qkv_nh + 2 * p.qkv_nkv) * 128;
    if (p.qkv_q == nullptr || p.N != Nexp || p.offs != nullptr || p.K % 64 != 0 || p.R < 1 || p.ldx % 8 != 0 || p.ldw % 8 != 0 ||
        !glm_aligned16(p.x) || !glm_aligned16(p.w) || !glm_aligned16(p.qkv_q) || !glm_aligned16(p.qkv_cs) ||
        (p.qkv_k && !glm_aligned16(p.qkv_k)) || (p.qkv_v && !glm_aligned16(p.qkv_v)) || (p.qkv_kc && !glm_aligned16(p.qkv_kc)) ||
        (p.qkv_vc && !glm_aligned16(p.qkv_vc)) || (p.bias && (((uintptr_t)p.bias) & 7) != 0) || p.R * p.ldx * 2 >= (1ll << 31) || (int64_t)p.N * p.ldw * 2 >= (1ll << 31)) {
        glm_set_error("gemm_largem(qkv epilogue): shape not covered (R=%ld N=%d K=%d)", (long)p.R, p.N, p.K);
        return GLM_ERR_UNSUPPORTED;
    }
    GemmParams q = p;
    q.y = p.qkv_q;  // never stored through; keeps the generic alignment checks meaningful
    q.ldy = p.N;
    return launch_largem<GEMM_PLAIN>(q, st);
}

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

#if GLM_LM_STUDY  // study builds only (build.py --variant _study GLM_LM_STUDY=1): not part of include/glm4moe_hip.h
// sweeps: row blocks per L2 band for the next launches (>= 1); returns the previous value
extern "C" int glm_debug_largem_band(int b) {
    const int old = g_largem_band;
    if (b >= 1) g_largem_band = b;
    return old;
}

// A/B studies: kernel variant for the next launches (bit set of LM_LEAN | LM_WIDEST | LM_NOSTORE; < 0 = default)
extern "C" int glm_debug_largem_variant(int v) {
    const int old = g_largem_variant;
    g_largem_variant = v;
    return old;
}
#endif

#if GLM_LM_STAMP
// timing study only: per-segment cycle sums of the last tile workgroup 8 ran (waves 0 and 4), [2][8]; [7] = K tiles
extern "C" int glm_debug_largem_stamps(unsigned long long* out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_lm_stamps), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
// same: [0..5] = top-of-tile wait, K loop, tile search, source set-up, ring fill issue, epilogue; [7] = tiles
extern "C" int glm_debug_largem_kt_stamps(unsigned long long* out26, int reset) {
    if (reset) {
        unsigned long long z[26] = {};
        return hipMemcpyToSymbol(HIP_SYMBOL(g_lm_kt_stamps), z, sizeof(z)) == hipSuccess ? 0 : -1;
    }
    return hipMemcpyFromSymbol(out26, HIP_SYMBOL(g_lm_kt_stamps), sizeof(unsigned long long) * 26) == hipSuccess ? 0 : -1;
}
extern "C" int glm_debug_largem_tile_stamps(unsigned long long* out8) {
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_lm_tile_stamps), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -1;
}
#endif
