// mfma_valu_probe.hip — what ONE wave per SIMD pays for VALU / transcendental / LDS instructions placed between its own MFMAs
// on gfx950 (r04; design input for csrc/prefill_attn64.hip).  Each variant runs a loop whose body holds 16 independent
// v_mfma_f32_16x16x32_bf16 (16 accumulators) with a fixed filler after every MFMA; one workgroup of 256 threads per CU
// (LDS request blocks a second one), time from s_memtime of wave 0 and from the wall clock.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_probe.hip -o /tmp/mvp && /tmp/mvp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define MFMA(i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i & 7]) : "v"(sc), "v"(mm));
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i & 7]));
#define CVT(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(y[i & 3]) : "v"(x[i & 7]), "v"(x[(i + 1) & 7]));
#define FMAJ(i, j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(i + j) & 7]) : "v"(sc), "v"(mm));
#define EXPJ(i, j) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(i + j) & 7]));
#define NOP0 asm volatile("s_nop 0");
#define LDSR(i) asm volatile("ds_read_b128 %0, %1" : "=v"(l[i & 3]) : "v"(laddr) : "memory");

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ char smem[];
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (threadIdx.x & 7)); b[i] = (__bf16)(0.02f * (threadIdx.x & 3)); }
    float x[8], sc = 1.0001f, mm = -0.5f;
    unsigned y[4] = {0, 0, 0, 0};
    f32x4 l[4];
    for (int i = 0; i < 4; ++i) l[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned laddr = (unsigned)(threadIdx.x * 16);
    for (int i = 0; i < 8; ++i) x[i] = 0.001f * threadIdx.x + i;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define STEP(i) MFMA(i) \
        if (MODE == 1) { FMA(i) } \
        if (MODE == 2) { EXP(i) } \
        if (MODE == 3) { FMA(i) EXP(i) } \
        if (MODE == 4) { FMA(i) EXP(i) NOP0 } \
        if (MODE == 5) { FMA(i) FMA(i) FMA(i) } \
        if (MODE == 6) { FMA(i) FMA(i) FMA(i) FMA(i) FMA(i) FMA(i) } \
        if (MODE == 7) { LDSR(i) } \
        if (MODE == 8) { EXP(i) EXP(i) } \
        if (MODE == 9) { CVT(i) } \
        if (MODE == 10) { NOP0 NOP0 NOP0 } \
        if (MODE == 11) { FMAJ(i, 0) EXPJ(i, 4) } \
        if (MODE == 12) { FMAJ(i, 0) FMAJ(i, 2) FMAJ(i, 4) } \
        if (MODE == 13) { FMAJ(i, 0) FMAJ(i, 1) FMAJ(i, 2) FMAJ(i, 3) FMAJ(i, 4) FMAJ(i, 5) } \
        if (MODE == 14) { EXPJ(i, 0) EXPJ(i, 4) } \
        if (MODE == 15) { FMAJ(i, 0) EXPJ(i, 4) FMAJ(i, 2) } \
        if (MODE == 16) { FMAJ(i, 0) EXPJ(i, 4) CVT(i) LDSR(i) } \
        if (MODE == 17) { FMAJ(i, 0) FMAJ(i, 1) FMAJ(i, 2) FMAJ(i, 3) }
        STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7) STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
        if (MODE == 7 || MODE == 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += x[i];
    for (int i = 0; i < 4; ++i) s += __uint_as_float(y[i]) + l[i][0];
    if (s == 123.456f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// dependent chains: 2 accumulators alternate (as the S = K.Q chains of the attention kernel)
template <int NCHAIN>
__global__ __launch_bounds__(256, 1) void probe_dep(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (threadIdx.x & 7)); b[i] = (__bf16)(0.02f * (threadIdx.x & 3)); }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define DSTEP(i) MFMA((i) % NCHAIN)
        DSTEP(0) DSTEP(1) DSTEP(2) DSTEP(3) DSTEP(4) DSTEP(5) DSTEP(6) DSTEP(7) DSTEP(8) DSTEP(9) DSTEP(10) DSTEP(11) DSTEP(12) DSTEP(13) DSTEP(14) DSTEP(15)
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0];
    if (s == 123.456f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <class K>
static void run(const char* name, K kern, int lds) {
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 64));
    const int iters = 20000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), lds, 0, out, cyc, 100);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), lds, 0, out, cyc, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double n = (double)iters * 16;
    const double tf = 256.0 * 4 * n * 16384 / (ms * 1e-3) / 1e12;
    printf("%-44s %7.2f ns per MFMA (wall), %7.2f counter ticks per MFMA, %7.1f TFLOP/s chip\n", name, ms * 1e6 / n, (double)c / n, tf);
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
    const int lds = 100 * 1024;  // one workgroup per CU
    run("16 independent MFMAs", probe<0>, lds);
    run("+ 1 v_fma after each", probe<1>, lds);
    run("+ 1 v_exp after each", probe<2>, lds);
    run("+ v_fma + v_exp after each", probe<3>, lds);
    run("+ v_fma + v_exp + s_nop 0 after each", probe<4>, lds);
    run("+ 3 v_fma after each", probe<5>, lds);
    run("+ 6 v_fma after each", probe<6>, lds);
    run("+ 1 ds_read_b128 after each", probe<7>, lds);
    run("+ 2 v_exp after each", probe<8>, lds);
    run("+ 1 v_cvt_pk_bf16 after each", probe<9>, lds);
    run("+ 3 s_nop 0 after each", probe<10>, lds);
    run("+ v_fma, v_exp (independent regs)", probe<11>, lds);
    run("+ 3 v_fma (independent regs)", probe<12>, lds);
    run("+ 4 v_fma (independent regs)", probe<17>, lds);
    run("+ 6 v_fma (independent regs)", probe<13>, lds);
    run("+ 2 v_exp (independent regs)", probe<14>, lds);
    run("+ v_fma, v_exp, v_fma (independent regs)", probe<15>, lds);
    run("+ v_fma, v_exp, v_cvt, ds_read_b128", probe<16>, lds);
    run("dependent: 1 chain", probe_dep<1>, 0);
    run("dependent: 2 chains alternating", probe_dep<2>, 0);
    run("dependent: 4 chains", probe_dep<4>, 0);
    return 0;
}
