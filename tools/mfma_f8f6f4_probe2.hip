// r05 probe 2: which lane's E8M0 scale byte applies to which (row, k-block) of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3), and
// the k map under a uniform scale.  A and B are written through the ASSUMED map (lane l: row/col l & 15, bytes j = 0..31 of its
// fragment = k 32 (l >> 4) + j).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
static float e4m3_to_f(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -f : f;
}
// sa[l], sb[l]: the scale VGPR of lane l (all four bytes given)
template <int OPA, int OPB>
__global__ void k16(const unsigned char* A, const unsigned char* B, const int* sa, const int* sb, float* D) {
    const int l = threadIdx.x;
    v8i a, b;
    const int* ap = reinterpret_cast<const int*>(A + (l & 15) * 128 + (l >> 4) * 32);
    const int* bp = reinterpret_cast<const int*>(B + (l & 15) * 128 + (l >> 4) * 32);
    for (int i = 0; i < 8; ++i) { a[i] = ap[i]; b[i] = bp[i]; }
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, OPA, sa[l], OPB, sb[l]);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
template <int KIND>
__global__ void rate(float* out, int iters) {
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x; b[i] = 0x30303030 + i; }
    v16f c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 0, 0, 0, 127, 0, 127);
        c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 0, 0, 0, 127, 0, 127);
        c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 0, 0, 0, 127, 0, 127);
        c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 0, 0, 0, 127, 0, 127);
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (4.0f * iters);
    out[1 + threadIdx.x + 256 * blockIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
    srand(5);
    const unsigned char vals[8] = {0x00, 0x38, 0x40, 0x44, 0x30, 0xb8, 0x48, 0xc0};
    std::vector<unsigned char> A(2048), B(2048);
    unsigned char *dA, *dB; float* dD; int *dsa, *dsb;
    hipMalloc(&dA, 4096); hipMalloc(&dB, 4096); hipMalloc(&dD, 4096); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
    float D[256];
    int sa[64], sb[64];
    auto run = [&](int opa = 0, int opb = 0) {
        hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
        hipMemcpy(dsa, sa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb, 256, hipMemcpyHostToDevice);
        if (opa == 0 && opb == 0) hipLaunchKernelGGL((k16<0, 0>), dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
        else if (opa == 1) hipLaunchKernelGGL((k16<1, 0>), dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
        else hipLaunchKernelGGL((k16<2, 3>), dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
        hipMemcpy(D, dD, 1024, hipMemcpyDeviceToHost);
    };
    // (a) uniform scale: the k map
    for (auto& v : A) v = vals[rand() % 8];
    for (auto& v : B) v = vals[rand() % 8];
    for (int l = 0; l < 64; ++l) sa[l] = sb[l] = 0x7f7f7f7f;
    run();
    int bad = 0;
    for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
        double ref = 0;
        for (int kk = 0; kk < 128; ++kk) ref += (double)e4m3_to_f(A[r * 128 + kk]) * e4m3_to_f(B[c * 128 + kk]);
        if (fabs(ref - D[r * 16 + c]) > 1e-6 * fabs(ref) + 1e-9) ++bad;
    }
    printf("(a) 16x16x128, uniform scale 2^0, assumed k map: %s (%d bad of 256)\n", bad ? "MISMATCH" : "exact", bad);
    // (b) A = one-hot k-block kb (ones), B = ones; scale_a of lane group g = 2^1, everything else 2^0: D = 32 * (2 if that scale applies)
    for (auto& v : B) v = 0x38;
    printf("(b) rows: k-block of A that is non-zero; columns: lane group g whose scale_a byte 0 is 2^1; entry = D[0][0] / 32\n");
    for (int kb = 0; kb < 4; ++kb) {
        for (int r = 0; r < 16; ++r) for (int kk = 0; kk < 128; ++kk) A[r * 128 + kk] = kk / 32 == kb ? 0x38 : 0;
        printf("    kb %d:", kb);
        for (int g = 0; g < 4; ++g) {
            for (int l = 0; l < 64; ++l) { sa[l] = (l >> 4) == g ? 0x7f7f7f80 : 0x7f7f7f7f; sb[l] = 0x7f7f7f7f; }
            run();
            bool uni = true;
            for (int i = 0; i < 256; ++i) uni &= D[i] == D[0];
            printf("  %g%s", D[0] / 32, uni ? "" : "*");
        }
        printf("\n");
    }
    // (c) per-row: scale_a = 2^1 in ONE lane L only; A = ones: which rows of D change (D = 128 normally)
    for (auto& v : A) v = 0x38;
    for (int L : {0, 5, 21, 37, 63}) {
        for (int l = 0; l < 64; ++l) { sa[l] = l == L ? 0x7f7f7f80 : 0x7f7f7f7f; sb[l] = 0x7f7f7f7f; }
        run();
        printf("(c) scale_a 2^1 in lane %2d only: rows with D != 128:", L);
        for (int r = 0; r < 16; ++r) { bool ch = false; for (int c = 0; c < 16; ++c) ch |= D[r * 16 + c] != 128.f; if (ch) printf(" %d(D=%g)", r, D[r * 16]); }
        printf("\n");
    }
    // (d) the same for scale_b: columns
    for (int L : {0, 5, 21, 37}) {
        for (int l = 0; l < 64; ++l) { sb[l] = l == L ? 0x7f7f7f80 : 0x7f7f7f7f; sa[l] = 0x7f7f7f7f; }
        run();
        printf("(d) scale_b 2^1 in lane %2d only: cols with D != 128:", L);
        for (int c = 0; c < 16; ++c) { bool ch = false; for (int r = 0; r < 16; ++r) ch |= D[r * 16 + c] != 128.f; if (ch) printf(" %d(D=%g)", c, D[c]); }
        printf("\n");
    }
    // (e) opsel: byte 1 of scale_a (OPA = 1): scale word 0x7f7f807f -> 2^1 expected everywhere
    for (int l = 0; l < 64; ++l) { sa[l] = 0x7f7f807f; sb[l] = 0x7f7f7f7f; }
    run(1, 0);
    printf("(e) opsel_a = 1 with byte 1 = 2^1: D[0][0] = %g (256 if byte 1 is taken)\n", D[0]);
    for (int l = 0; l < 64; ++l) { sa[l] = 0x7f807f7f; sb[l] = 0x807f7f7f; }
    run(2, 3);
    printf("(e) opsel_a = 2 (byte 2 = 2^1), opsel_b = 3 (byte 3 = 2^1): D[0][0] = %g (512 if taken)\n", D[0]);
    float* dO; hipMalloc(&dO, (1 + 256 * 1024) * 4);
    for (int chip = 0; chip < 2; ++chip) {
        const int blocks = chip ? 1024 : 1, iters = 20000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(256), 0, 0, dO, 100); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(256), 0, 0, dO, iters); hipEventRecord(e1); hipDeviceSynchronize();
        float ms = 0, cyc = 0; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&cyc, dO, 4, hipMemcpyDeviceToHost);
        printf("mfma_scale 32x32x64 f8f6f4 (e4m3): %.1f cycles per MFMA, %d blocks x 4 waves: %.1f TFLOP/s\n", cyc, blocks,
               2.0 * 32 * 32 * 64 * 4.0 * iters * 4 * blocks / (ms * 1e-3) / 1e12);
    }
    return 0;
}
