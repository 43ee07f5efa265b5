// stream_probe.hip — micro-benchmark: HBM read rate of the access shapes used by the decode kernels.
//   mode 0: contiguous — a wave instruction reads 1 KiB contiguous (fragment-major / packed layout)
//   mode 1: row fragments — a wave instruction reads 16 rows x 64 B, rows `row_bytes` apart (gemm_smallm today)
// Each wave keeps 8 x 16 B loads in flight (unrolled), 4 waves per workgroup, WGs stream `bytes_per_wg`.
// build: hipcc --offload-arch=gfx950 -O3 -o stream_probe tools/stream_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int MODE, int BAR>
__global__ __launch_bounds__(256) void probe(const u32x4* __restrict__ w, unsigned* sink, long row_bytes, int rows_per_wave,
                                             long wave_span, long row_len) {
    extern __shared__ char dyn_lds[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const long gw = (long)blockIdx.x * 4 + wid;  // global wave id
    unsigned acc = 0;
    if (MODE == 0) {
        // wave owns a contiguous span of wave_span bytes
        const u32x4* p = w + gw * (wave_span / 16) + lane;
        const long n = wave_span / 1024;  // instructions
        for (long i = 0; i < n; i += 8) {
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = p[(i + j) * 64];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j][0] ^ v[j][1] ^ v[j][2] ^ v[j][3];
        }
    } else {
        // wave owns rows_per_wave(=16 or 32) rows of row_bytes; lane (c=lane&15, g=lane>>4) reads row c (+16), 16 B at 64*ks+16*g
        const int c = lane & 15, g = lane >> 4;
        const char* base = (const char*)w + gw * rows_per_wave * row_bytes;
        const long nk = row_len / 64;  // k-steps
        const int rt = rows_per_wave / 16;
        for (long ks = 0; ks < nk; ks += 8 / rt) {
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int t = j % rt, kk = j / rt;
                v[j] = *(const u32x4*)(base + (long)(t * 16 + c) * row_bytes + (ks + kk) * 64 + g * 16);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j][0] ^ v[j][1] ^ v[j][2] ^ v[j][3];
            if (BAR == 1) __syncthreads();
            if (BAR == 2 && ((ks / (8 / rt)) & 7) == 7) __syncthreads();
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const long total = 1400l << 20;
    u32x4* w; unsigned* sink;
    hipMalloc(&w, total + (256 << 20)); hipMalloc(&sink, 64);
    hipMemset(w, 1, total + (256 << 20));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void*)probe<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)probe<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)probe<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)probe<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // GATEUP-like: 2x16 rows of 8192 B, pitch 8192+128; vary resident workgroups per CU through LDS
    int lds_kb[] = {0, 20, 26, 40, 53, 80, 160};
    for (int kb : lds_kb) {
        long row_bytes = 8192 + 128, row_len = 8192; int rows = 32;
        long wave_span = (long)rows * row_bytes; long waves = total / wave_span; long wgs = waves / 4;
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL((probe<1, 0>), dim3(wgs), dim3(256), kb * 1024, 0, w, sink, row_bytes, rows, wave_span, row_len);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (rep && ms < best) best = ms;
        }
        int wg_per_cu = kb ? 160 / kb : 8; if (wg_per_cu > 8) wg_per_cu = 8;
        printf("2x16 rows pitch 8320, LDS %3d KB/WG (~%d WG = %2d waves per CU): %7.1f us %7.1f GB/s\n", kb, wg_per_cu, wg_per_cu * 4,
               best * 1e3, wgs * 4.0 * rows * row_len / best / 1e6);
    }
    // same but the grid sized like the real kernel: 1342 workgroups of 1 MiB (64 gate + 64 up rows: 4 waves x 32 rows)
    for (int kb : {0, 40}) {
        long row_bytes = 8192 + 128, row_len = 8192; int rows = 32; long wgs = 1342;
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL((probe<1, 0>), dim3(wgs), dim3(256), kb * 1024, 0, w, sink, row_bytes, rows, (long)rows * row_bytes, row_len);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (rep && ms < best) best = ms;
        }
        printf("1342 WGs x 1 MiB, LDS %d KB: %7.1f us %7.1f GB/s\n", kb, best * 1e3, wgs * 4.0 * rows * row_len / best / 1e6);
    }
    // barrier coupling: same stream, 1342 WGs, __syncthreads() after every 8 loads (one "chunk") / every 8 chunks
    {
        long row_bytes = 8192 + 128, row_len = 8192; int rows = 32; long wgs = 1342;
        for (int bar = 0; bar < 3; ++bar) {
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(a);
                if (bar == 0) hipLaunchKernelGGL((probe<1, 0>), dim3(wgs), dim3(256), 40 * 1024, 0, w, sink, row_bytes, rows, (long)rows * row_bytes, row_len);
                if (bar == 1) hipLaunchKernelGGL((probe<1, 1>), dim3(wgs), dim3(256), 40 * 1024, 0, w, sink, row_bytes, rows, (long)rows * row_bytes, row_len);
                if (bar == 2) hipLaunchKernelGGL((probe<1, 2>), dim3(wgs), dim3(256), 40 * 1024, 0, w, sink, row_bytes, rows, (long)rows * row_bytes, row_len);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (rep && ms < best) best = ms;
            }
            printf("1342 WGs, barrier mode %d (0 none, 1 per chunk, 2 per 8 chunks): %7.1f us %7.1f GB/s\n", bar, best * 1e3,
                   wgs * 4.0 * rows * row_len / best / 1e6);
        }
    }
    // TLB / cold-region effect: same launch, but every rep streams a different 1.4 GB region of a 42 GB buffer
    {
        char* big; const long region = 1500l << 20; const int nreg = 28;
        if (hipMalloc(&big, region * nreg) == hipSuccess) {
            hipMemset(big, 1, region * nreg);
            hipDeviceSynchronize();
            long row_bytes = 8192 + 128, row_len = 8192; int rows = 32; long wgs = 1342;
            float tot = 0; int cnt = 0;
            for (int rep = 0; rep < nreg; ++rep) {
                hipEventRecord(a);
                hipLaunchKernelGGL((probe<1, 0>), dim3(wgs), dim3(256), 40 * 1024, 0, (const u32x4*)(big + rep * region), sink, row_bytes, rows,
                                   (long)rows * row_bytes, row_len);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); tot += ms; ++cnt;
            }
            printf("1342 WGs, rotating over %d cold regions: mean %7.1f us %7.1f GB/s\n", nreg, tot / cnt * 1e3,
                   wgs * 4.0 * rows * row_len / (tot / cnt) / 1e6);
            tot = 0; cnt = 0;
            for (int rep = 0; rep < 8; ++rep) {
                hipEventRecord(a);
                hipLaunchKernelGGL((probe<1, 0>), dim3(wgs), dim3(256), 40 * 1024, 0, (const u32x4*)(big), sink, row_bytes, rows,
                                   (long)rows * row_bytes, row_len);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (rep) { tot += ms; ++cnt; }
            }
            printf("1342 WGs, same region repeatedly:        mean %7.1f us %7.1f GB/s\n", tot / cnt * 1e3,
                   wgs * 4.0 * rows * row_len / (tot / cnt) / 1e6);
        }
    }
    return 0;
}
