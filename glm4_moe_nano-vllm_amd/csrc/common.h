// common.h — shared device/host helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/glm4moe_hip.h"

#define GLM_WAVE 64

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;       // MFMA A/B fragment (16x16x32)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;          // MFMA C/D fragment (16x16)
typedef __attribute__((ext_vector_type(16))) float f32x16_t;        // MFMA C/D fragment (32x32)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;   // 16-byte vector
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;   // 8-byte vector
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

// ---- bf16 <-> f32 -------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
// round-to-nearest-even, NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
// round an f32 value to bf16 precision and return it as f32 (a "rounding point")
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}
__device__ __forceinline__ float bflo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bfhi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// ---- wave reductions ----------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- host-side error plumbing -----------------------------------------------------------------
void glm_set_error(const char* fmt, ...);
#define GLM_CHECK_ARG(cond, code, ...)            \
    do {                                          \
        if (!(cond)) {                            \
            glm_set_error(__VA_ARGS__);           \
            return (code);                        \
        }                                         \
    } while (0)
#define GLM_CHECK_LAUNCH(name)                                                       \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            glm_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));    \
            return GLM_ERR_LAUNCH;                                                   \
        }                                                                            \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel instantiation, device): the attribute belongs to the
// device the call is made on, so a process that drives several GPUs must set it on each of them
#define GLM_MAX_DEVICES 16
#define GLM_SET_MAX_LDS(kern, bytes, name)                                                                   \
    do {                                                                                                     \
        static bool lds_set__[GLM_MAX_DEVICES] = {};                                                         \
        int dev__ = -1;                                                                                      \
        if (hipGetDevice(&dev__) != hipSuccess) dev__ = -1;                                                  \
        if (dev__ < 0 || dev__ >= GLM_MAX_DEVICES || !lds_set__[dev__]) {                                    \
            hipError_t e__ = hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes)); \
            if (e__ != hipSuccess) {                                                                         \
                glm_set_error("%s: hipFuncSetAttribute(%d B LDS): %s", name, (int)(bytes), hipGetErrorString(e__)); \
                return GLM_ERR_LAUNCH;                                                                       \
            }                                                                                                \
            if (dev__ >= 0 && dev__ < GLM_MAX_DEVICES) lds_set__[dev__] = true;                              \
        }                                                                                                    \
    } while (0)

static inline bool glm_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
static inline hipStream_t glm_s(glm_stream_t s) { return (hipStream_t)s; }
