// r05 probe: v_mfma_scale_f32_16x16x128_f8f6f4 / _32x32x64_ with OCP e4m3 operands on gfx950 — operand lane map, scale
// semantics, issue rate.  hipcc --offload-arch=gfx950 -O3 tools/mfma_f8f6f4_probe.hip -o /tmp/mfma_f8_probe && /tmp/mfma_f8_probe
//   (1) exact-integer check of the assumed map: lane l holds, for row (A) / column (B) l & 15, the 32 consecutive k
//       32 (l >> 4) .. + 31 in its 8 operand registers (byte j of the 32-byte fragment = k 32 (l >> 4) + j); D as every
//       16x16 MFMA (col = l & 15, row = 4 (l >> 4) + reg).  Asymmetric A and B.
//   (2) the E8M0 block scale: a VGPR per operand, byte `opsel` of lane l scales lane l's 32-element block by 2^(s - 127).
//   (3) cycles per instruction, one wave per SIMD, 4 independent accumulators; and the bf16 16x16x32 form beside it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));

static float e4m3_to_f(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f;
    if (e == 0) f = ldexpf((float)m, -9);
    else if (e == 15 && m == 7) f = NAN;
    else f = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -f : f;
}

__global__ void k16(const unsigned char* A, const unsigned char* B, float* D, int sa, int sb) {
    const int l = threadIdx.x;
    v8i a, b;
    const int* ap = reinterpret_cast<const int*>(A + (l & 15) * 128 + (l >> 4) * 32);
    const int* bp = reinterpret_cast<const int*>(B + (l & 15) * 128 + (l >> 4) * 32);
    for (int i = 0; i < 8; ++i) { a[i] = ap[i]; b[i] = bp[i]; }
    v4f c = {0, 0, 0, 0};
    // scale operand: byte 0 of the VGPR (opsel 0); lane-group g of A uses exponent sa + g, B uses sb
    const int scale_a = (sa + (l >> 4)) & 0xff, scale_b = sb & 0xff;
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}

__global__ void k32(const unsigned char* A, const unsigned char* B, float* D) {
    // 32x32x64: assumed lane l: row/col l & 31, k = 32 (l >> 5) + j
    const int l = threadIdx.x;
    v8i a, b;
    const int* ap = reinterpret_cast<const int*>(A + (l & 31) * 64 + (l >> 5) * 32);
    const int* bp = reinterpret_cast<const int*>(B + (l & 31) * 64 + (l >> 5) * 32);
    for (int i = 0; i < 8; ++i) { a[i] = ap[i]; b[i] = bp[i]; }
    v16f c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

template <int KIND>
__global__ void rate(float* out, int iters) {
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x; b[i] = 0x30303030 + i; }
    v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    bf8v xa, xb;
    for (int i = 0; i < 8; ++i) { xa[i] = (__bf16)1.0f; xb[i] = (__bf16)0.5f; }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
            c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c0, 0, 0, 0, 127, 0, 127);
            c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c1, 0, 0, 0, 127, 0, 127);
            c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c2, 0, 0, 0, 127, 0, 127);
            c3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c3, 0, 0, 0, 127, 0, 127);
        } else {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, xb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, xb, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, xb, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, xb, c3, 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (4.0f * iters);
    out[1 + threadIdx.x + 256 * blockIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
    srand(3);
    std::vector<unsigned char> A(16 * 128), B(16 * 128);
    // exact data: small integers / halves representable in e4m3, asymmetric
    const unsigned char vals[8] = {0x00, 0x38 /*1*/, 0x40 /*2*/, 0x44 /*3*/, 0x30 /*.5*/, 0xb8 /*-1*/, 0x48 /*4*/, 0xc0 /*-2*/};
    for (auto& v : A) v = vals[rand() % 8];
    for (auto& v : B) v = vals[rand() % 8];
    unsigned char *dA, *dB; float* dD;
    hipMalloc(&dA, 4096); hipMalloc(&dB, 4096); hipMalloc(&dD, 4096 * 4);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    float D[1024];
    for (int variant = 0; variant < 2; ++variant) {
        const int sa = variant ? 125 : 127, sb = variant ? 130 : 127;
        hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dD, sa, sb);
        hipMemcpy(D, dD, 256 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
            double ref = 0;
            for (int kk = 0; kk < 128; ++kk)
                ref += (double)e4m3_to_f(A[r * 128 + kk]) * e4m3_to_f(B[c * 128 + kk]) * ldexp(1.0, (sa + kk / 32) - 127) * ldexp(1.0, sb - 127);
            if (fabs(ref - D[r * 16 + c]) > 1e-6 * fabs(ref) + 1e-9) { if (bad < 4) printf("  [%d][%d] got %g want %g\n", r, c, D[r * 16 + c], ref); ++bad; }
        }
        printf("16x16x128 e4m3 x e4m3, map (row l&15, k 32(l>>4)+j), scales A 2^(%d+g-127) B 2^(%d-127): %s (%d bad of 256)\n", sa, sb, bad ? "MISMATCH" : "exact", bad);
    }
    {
        std::vector<unsigned char> A2(32 * 64), B2(32 * 64);
        for (auto& v : A2) v = vals[rand() % 8];
        for (auto& v : B2) v = vals[rand() % 8];
        hipMemcpy(dA, A2.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B2.data(), 2048, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        hipMemcpy(D, dD, 1024 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
            double ref = 0;
            for (int kk = 0; kk < 64; ++kk) ref += (double)e4m3_to_f(A2[r * 64 + kk]) * e4m3_to_f(B2[c * 64 + kk]);
            if (fabs(ref - D[r * 32 + c]) > 1e-6 * fabs(ref) + 1e-9) ++bad;
        }
        printf("32x32x64 e4m3 x e4m3, map (row l&31, k 32(l>>5)+j): %s (%d bad of 1024)\n", bad ? "MISMATCH" : "exact", bad);
    }
    float* dO; hipMalloc(&dO, (1 + 256 * 1024) * 4);
    for (int kind = 0; kind < 2; ++kind) {
        for (int chip = 0; chip < 2; ++chip) {
            const int blocks = chip ? 1024 : 1;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const int iters = 20000;
            if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(256), 0, 0, dO, 100); else hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(256), 0, 0, dO, 100);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(256), 0, 0, dO, iters); else hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(256), 0, 0, dO, iters);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms = 0, cyc = 0; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&cyc, dO, 4, hipMemcpyDeviceToHost);
            const double flop = 2.0 * 16 * 16 * (kind == 0 ? 128 : 32) * 4.0 * iters * 4 /*waves*/ * blocks;
            printf("%s: %.1f cycles per MFMA (wave 0's clock64), %d blocks x 4 waves: %.1f TFLOP/s\n", kind == 0 ? "mfma_scale 16x16x128 f8f6f4 (e4m3)" : "mfma 16x16x32 bf16", cyc, blocks, flop / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
