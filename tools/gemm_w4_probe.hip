// gemm_w4_probe.hip — PROTOTYPE (tools/, not part of the library): dense bf16 GEMM y = x · Wᵀ (+bias) with a 128 x 128 WAVE tile.
//
// Why: the product loop (gemm_largem.hip, 8 waves, 64 x 128 wave tiles) is paced by its load intervals, not by its MFMAs
// (profiles/r03/largem_residency_probe.txt): a 64 x 128 wave tile reads 24 KB of fragments per 64-wide K tile for 64 MFMAs.
// This probe measures the other corner: 4 waves (one per SIMD), each 128(n) x 128(m) = 64 accumulator tiles (256 registers,
// AGPRs), 16 fragment reads per 32-wide K step for 64 MFMAs (0.25 KB per MFMA instead of 0.375), ONE barrier per K step, a
// 4-stage ring of 32-KiB stages (prefetch distance 3 K steps), fragment reads of step k+1 issued between the MFMAs of step k.
// Same MFMA shape and k order per accumulator as the product loop, so the result must be bit-identical to it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

namespace {
constexpr int BN = 256, BM = 256, BK = 32, NST = 4;
constexpr int RB = BK * 2;                  // 64 bytes per tile row
constexpr int WT = BN * RB, XT = BM * RB;   // 16 KiB each
constexpr int STAGE = WT + XT;              // 32 KiB
constexpr int LDS_BYTES = NST * STAGE;      // 128 KiB

__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    const __bf16 a = (__bf16)lo, b = (__bf16)hi;
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((glb_void_t*)g, (lds_void_t*)l, 16, 0, 0);
}
template <int OFF>
__device__ __forceinline__ void lds_rd(u32x4_t& d, unsigned a) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "i"(OFF));
}
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 2; }  // 64-byte rows, b128 reads (as CfgB of gemm_largem)

__device__ __forceinline__ void mfma_a(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void wait_frags16(u32x4_t (&a)[8], u32x4_t (&b)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(b[0]),
                   "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7])
                 :
                 : "memory");
}
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (N > 0) {
        sfor<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

__global__ __launch_bounds__(256) void gemm_w4_kernel(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ w,
                                                      int64_t ldw, const bf16_t* __restrict__ bias, bf16_t* __restrict__ y,
                                                      int64_t ldy, int M, int N, int K, int n_tiles, int total) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid & 1, wm = wid >> 1;
    const int c = lane & 15, g4 = lane >> 4;
    // XCD-aware order: workgroups with equal id % 8 share an XCD and get a contiguous run of the band-ordered tile list
    // (bands of 4 row blocks x all n-tiles, n-tile major inside a band)
    const int q = total >> 3;
    const int logical = (blockIdx.x & 7) * q + (blockIdx.x >> 3);
    const int band = logical / (4 * n_tiles), rem = logical % (4 * n_tiles);
    const int m0 = (band * 4 + (rem & 3)) * BM, n0 = (rem >> 2) * BN;
    const int nk = K / BK;

    // ---- DMA sources: per K step this wave moves 4 KiB of W and 4 KiB of X (4 + 4 instructions of 16 rows x 64 B) ----
    const bf16_t* wsrc[4];
    const bf16_t* xsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wid * 4 + i) * 16 + (lane >> 2);
        const int ch = (lane & 3) ^ swz(row);
        wsrc[i] = w + (int64_t)min(n0 + row, N - 1) * ldw + ch * 8;
        xsrc[i] = x + (int64_t)min(m0 + row, M - 1) * ldx + ch * 8;
    }
    auto stage_piece = [&](int kt, auto jc) {  // piece j of this wave's 8 DMA instructions of K step kt
        constexpr int j = decltype(jc)::value;
        char* base = smem + (kt & (NST - 1)) * STAGE + wid * 4096;
        const int k0 = kt * BK;
        if constexpr (j < 4) glds16(wsrc[j] + k0, base + j * 1024);
        else glds16(xsrc[j - 4] + k0, base + WT + (j - 4) * 1024);
    };
    auto stage = [&](int kt) { sfor<8>([&](auto jc) { stage_piece(kt, jc); }); };
    // ---- fragment addresses: lane (c, g4) reads row +c, chunk g4 (k = 8 g4 .. +8 of the 32) ----
    const unsigned lbase = (unsigned)(uintptr_t)(const lds_void_t*)smem;
    const int arow = wn * 128 + c, brow = wm * 128 + c;
    const unsigned a_rel = arow * RB + ((g4 ^ swz(arow)) << 4);
    const unsigned b_rel = WT + brow * RB + ((g4 ^ swz(brow)) << 4);

    f32x4_t acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    u32x4_t fa[2][8], fb[2][8];

    // read all 16 fragments of K step kt into buffer P
    auto read_all = [&](auto pc, int kt) {
        constexpr int P = decltype(pc)::value;
        const unsigned so = (kt & (NST - 1)) * STAGE;
        const unsigned aa = lbase + a_rel + so, ab = lbase + b_rel + so;
        sfor<8>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            lds_rd<t * 16 * RB>(fa[P][t], aa);
        });
        sfor<8>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            lds_rd<t * 16 * RB>(fb[P][t], ab);
        });
    };
    // ---- prologue: four stages in flight, step 0 in registers ----
    stage(0);
    stage(1);
    stage(2);
    stage(3);
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_all(std::integral_constant<int, 0>{}, 0);
    wait_frags16(fa[0], fb[0]);

    // one K step.  At its barrier every wave has the fragments of step kt in registers (read during step kt-1), so stage kt & 3
    // is free again: the 64 MFMAs of step kt run with the 16 fragment reads of step kt+1 (one per 4 MFMAs) and the 8 DMA
    // instructions of step kt+4 (one per 8 MFMAs) issued BETWEEN them — a DMA issue costs ~70 cycles of the wave's issue
    // slot; in front of the MFMAs (first build of this probe) the matrix pipe sat idle for 8 of them per step.
    // The accumulators are pinned to AGPRs ("+a"): left to hipcc, a 256-register accumulator set is shuffled between the two
    // files (1000 v_accvgpr moves and 256 spilled registers).  No accumulator is used twice within 64 MFMAs.
    auto kstep = [&](auto pc, auto stagec, auto vmc, auto readc, int kt) {
        constexpr int P = decltype(pc)::value, VM = decltype(vmc)::value;
        constexpr bool DO_STAGE = decltype(stagec)::value, DO_READS = decltype(readc)::value;
        if constexpr (VM == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if constexpr (VM == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr (VM == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (DO_READS) __builtin_amdgcn_s_barrier();
        const unsigned so = ((kt + 1) & (NST - 1)) * STAGE;
        const unsigned aa = lbase + a_rel + so, ab = lbase + b_rel + so;
        sfor<8>([&](auto tnc) {
            constexpr int tn = decltype(tnc)::value;
            sfor<8>([&](auto tmc) {
                constexpr int tm = decltype(tmc)::value;
                constexpr int i = tn * 8 + tm;
                if constexpr (DO_READS && i % 4 == 0) {
                    constexpr int r = i / 4;  // 0..15: A fragments first, then B
                    if constexpr (r < 8) lds_rd<r * 16 * RB>(fa[1 - P][r], aa);
                    else lds_rd<(r - 8) * 16 * RB>(fb[1 - P][r - 8], ab);
                }
                if constexpr (DO_STAGE && i % 8 == 2) stage_piece(kt + 4, std::integral_constant<int, i / 8>{});
                mfma_a(acc[tn][tm], fa[P][tn], fb[P][tm]);
            });
        });
        if constexpr (DO_READS) wait_frags16(fa[1 - P], fb[1 - P]);
    };
    using T = std::true_type;
    using F = std::false_type;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using V16 = std::integral_constant<int, 16>;
    using V8 = std::integral_constant<int, 8>;
    using V0 = std::integral_constant<int, 0>;
    using VN = std::integral_constant<int, -1>;
    // nk is even and >= 6 (host check): full steps in pairs, then the four steps that drain the ring.
    // Wait of step kt = this wave's share of step kt+1; younger than it: steps kt+2, kt+3 (step kt+4 is issued behind the wait).
    int kt = 0;
    for (; kt < nk - 4; kt += 2) {
        kstep(P0{}, T{}, V16{}, T{}, kt);
        kstep(P1{}, T{}, V16{}, T{}, kt + 1);
    }
    kstep(P0{}, F{}, V16{}, T{}, kt);       // nk-4: steps nk-2, nk-1 stay in flight
    kstep(P1{}, F{}, V8{}, T{}, kt + 1);    // nk-3: step nk-1 stays in flight
    kstep(P0{}, F{}, V0{}, T{}, kt + 2);    // nk-2
    kstep(P1{}, F{}, VN{}, F{}, kt + 3);    // nk-1: no next step
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs retire before the accumulators are read

    // ---- epilogue: lane (c, g4) of tile (tn, tm) holds token m = .. + 16 tm + c, columns n = .. + 16 tn + 4 g4 + r ----
#pragma unroll
    for (int tm = 0; tm < 8; ++tm) {
        const int m = m0 + wm * 128 + 16 * tm + c;
        if (m >= M) continue;
#pragma unroll
        for (int tn = 0; tn < 8; ++tn) {
            const int n = n0 + wn * 128 + 16 * tn + 4 * g4;
            if (n >= N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[tn][tm][r] + (bias ? bf2f(bias[n + r]) : 0.f);
            *reinterpret_cast<u32x2_t*>(y + (int64_t)m * ldy + n) = u32x2_t{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        }
    }
}
}  // namespace

extern "C" int gemm_w4(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, void* y, int64_t ldy, int M,
                       int N, int K, void* stream) {
    if (K % (2 * BK) != 0 || K < 6 * BK || N % 4 != 0 || M <= 0 || N <= 0) return 1;
    const int n_tiles = (N + BN - 1) / BN, m_tiles = (M + BM - 1) / BM;
    const int m_bands = (m_tiles + 3) / 4;
    const int total = m_bands * 4 * n_tiles;
    if (total % 8 != 0) return 2;  // prototype: shapes whose tile count is a multiple of 8
    static bool set = false;
    if (!set) {
        if (hipFuncSetAttribute((const void*)gemm_w4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return 3;
        set = true;
    }
    hipLaunchKernelGGL(gemm_w4_kernel, dim3((unsigned)total), dim3(256), LDS_BYTES, (hipStream_t)stream, (const bf16_t*)x, ldx,
                       (const bf16_t*)w, ldw, (const bf16_t*)bias, (bf16_t*)y, ldy, M, N, K, n_tiles, total);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}
