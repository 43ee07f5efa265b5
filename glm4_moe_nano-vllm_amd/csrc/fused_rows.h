// fused_rows.h — row-level device functions shared by a stand-alone launch and the kernel that absorbs it as a prologue.
// Sharing the code (and the 512-thread mapping it is written for) is what makes the fused launch bit-identical to the
// pair of launches it replaces.
#pragma once
#include "common.h"

#define GLM_COMBINE_MAX_K 64
constexpr int GLM_CN_NT = 512;  // threads of a workgroup that runs glm_combine_add_rmsnorm_row

// MoE combine of token t (moe.py:133-171: bf16 adds in ascending sorted-row position, then the shared row, one rounding)
// followed by the NEXT layer's input add-RMSNorm (layernorm.py:29-40):
//     x = bf16(combine);  s = f32(x) + f32(residual_in[t]);  residual_out[t] = bf16(s);  y = bf16(bf16(s * rstd) * w)
// Executed by ALL GLM_CN_NT threads of the workgroup.  Outputs, each optional (nullptr = skip): out_g / res_out_g rows in
// global memory, out_lds = this token's normalised row in LDS (a later __syncthreads() of the caller publishes it).
// `red` = GLM_CN_NT / 64 floats of LDS scratch.  The caller synchronises before the next call (red and the position list
// are reused).
template <int VPT>
__device__ __forceinline__ void glm_combine_add_rmsnorm_row(const bf16_t* __restrict__ y, int64_t ldy,
                                                            const int32_t* __restrict__ pair_pos, int k,
                                                            int64_t shared_base, const bf16_t* res_in,
                                                            const bf16_t* __restrict__ w, bf16_t* out_g, bf16_t* res_out_g,
                                                            bf16_t* out_lds, int H, float eps, int64_t t, float* red) {
    __shared__ int pos[GLM_COMBINE_MAX_K];
    __shared__ int npos;
    if (threadIdx.x == 0) {
        int n = 0;
        for (int j = 0; j < k; ++j) {
            const int p = pair_pos[t * k + j];
            if (p < 0) continue;
            int i = n++;
            while (i > 0 && pos[i - 1] > p) {
                pos[i] = pos[i - 1];
                --i;
            }
            pos[i] = p;
        }
        npos = n;
    }
    __syncthreads();
    const int n = npos;
    const int nvec = H >> 3;
    float s[VPT][8];
    float ss = 0.f;
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int c = threadIdx.x + v * GLM_CN_NT;
        if (c < nvec) {
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = 0.f;
            // the rounding chain is serial, the row loads are not: 8 rows are in flight at a time (clamped re-reads of
            // the last row fill the batch), together with the shared row and the residual
            u32x4_t sv = {0, 0, 0, 0};
            if (shared_base >= 0) sv = reinterpret_cast<const u32x4_t*>(y + (shared_base + t) * ldy)[c];
            const u32x4_t rv = reinterpret_cast<const u32x4_t*>(res_in + t * H)[c];
            for (int j0 = 0; j0 < n; j0 += 8) {
                u32x4_t vv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    vv[j] = reinterpret_cast<const u32x4_t*>(y + (int64_t)pos[min(j0 + j, n - 1)] * ldy)[c];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j0 + j < n) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc[2 * q] = rbf(acc[2 * q] + bflo(vv[j][q]));
                            acc[2 * q + 1] = rbf(acc[2 * q + 1] + bfhi(vv[j][q]));
                        }
                    }
            }
            if (shared_base >= 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[2 * q] = acc[2 * q] + bflo(sv[q]);
                    acc[2 * q + 1] = acc[2 * q + 1] + bfhi(sv[q]);
                }
            }
            u32x4_t ov;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s[v][2 * q] = rbf(acc[2 * q]) + bflo(rv[q]);          // rbf: the combine's own output rounding
                s[v][2 * q + 1] = rbf(acc[2 * q + 1]) + bfhi(rv[q]);
                ov[q] = pack2bf(s[v][2 * q], s[v][2 * q + 1]);
            }
            if (res_out_g) reinterpret_cast<u32x4_t*>(res_out_g + t * H)[c] = ov;
#pragma unroll
            for (int i = 0; i < 8; ++i) ss += s[v][i] * s[v][i];
        }
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < GLM_CN_NT / 64; ++i) tot += red[i];
    const float rstd = 1.0f / sqrtf(tot / (float)H + eps);
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int c = threadIdx.x + v * GLM_CN_NT;
        if (c < nvec) {
            const u32x4_t wv = reinterpret_cast<const u32x4_t*>(w)[c];
            u32x4_t o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = rbf(s[v][2 * i] * rstd) * bflo(wv[i]);
                const float b = rbf(s[v][2 * i + 1] * rstd) * bfhi(wv[i]);
                o[i] = pack2bf(a, b);
            }
            if (out_g) reinterpret_cast<u32x4_t*>(out_g + t * H)[c] = o;
            if (out_lds) reinterpret_cast<u32x4_t*>(out_lds)[c] = o;
        }
    }
}
