// elementwise.hip — HBM/latency-bound row kernels of the GLM-4-MoE layer for gfx950:
//   RMSNorm (a2), residual-add RMSNorm (a3), per-head RMSNorm (a4), NeoX rotary (a6),
//   paged-KV scatter (a7), the fused split+norm+rotary+scatter used by the attention wrapper,
//   and SiluAndMul.  All arithmetic is fp32 with the reference's bf16 rounding points
//   (SURVEY.md §8a); this file is compiled with -ffp-contract=off so that a*b-c*d rounds the
//   way torch's separate mul/sub ops do.
//
// Layout choices (MI355X): every thread moves 16-byte (8 x bf16) vectors; a row of H=4096 is
// one 256-thread workgroup x 2 vectors per thread, kept in registers between the reduction and
// the scale pass so each row is read from HBM exactly once.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------
// block-wide sum for 256-thread workgroups (4 waves)
__device__ __forceinline__ float block_sum_256(float v, float* red /* [4] LDS */) {
    v = wave_sum(v);
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wid] = v;
    __syncthreads();
    float t = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return t;
}

// y = bf16( bf16(s * rstd) * w ) for 8 packed elements
__device__ __forceinline__ u32x4_t norm_scale8(const float (&s)[8], float rstd, u32x4_t wv) {
    u32x4_t o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float a = rbf(s[2 * i] * rstd) * bflo(wv[i]);
        float b = rbf(s[2 * i + 1] * rstd) * bfhi(wv[i]);
        o[i] = pack2bf(a, b);
    }
    return o;
}

// RMSNorm / add-RMSNorm, vector path: H % 8 == 0, H <= 256 * 8 * VPT.
template <int VPT, bool ADD>
__global__ __launch_bounds__(256) void rmsnorm_vec_kernel(const bf16_t* x, const bf16_t* res_in,
                                                          const bf16_t* __restrict__ w, bf16_t* y,
                                                          bf16_t* res_out, int H, float eps) {
    // x / res_in / y / res_out may alias each other (no __restrict__): every element is read
    // and then written by the same thread only.
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const int nvec = H >> 3;
    const u32x4_t* xr = reinterpret_cast<const u32x4_t*>(x + row * H);
    const u32x4_t* rr = ADD ? reinterpret_cast<const u32x4_t*>(res_in + row * H) : nullptr;
    const u32x4_t* wr = reinterpret_cast<const u32x4_t*>(w);
    u32x4_t* yr = reinterpret_cast<u32x4_t*>(y + row * H);
    u32x4_t* ro = ADD ? reinterpret_cast<u32x4_t*>(res_out + row * H) : nullptr;

    float s[VPT][8];
    float ss = 0.f;
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int i = threadIdx.x + v * 256;
        if (i < nvec) {
            u32x4_t xv = xr[i];
            if (ADD) {
                u32x4_t rv = rr[i];
                u32x4_t ov;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s[v][2 * j] = bflo(xv[j]) + bflo(rv[j]);
                    s[v][2 * j + 1] = bfhi(xv[j]) + bfhi(rv[j]);
                    ov[j] = pack2bf(s[v][2 * j], s[v][2 * j + 1]);
                }
                ro[i] = ov;  // residual_out = bf16(sum); variance uses the unrounded sum
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s[v][2 * j] = bflo(xv[j]);
                    s[v][2 * j + 1] = bfhi(xv[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += s[v][j] * s[v][j];
        }
    }
    const float tot = block_sum_256(ss, red);
    const float var = tot / (float)H;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int i = threadIdx.x + v * 256;
        if (i < nvec) yr[i] = norm_scale8(s[v], rstd, wr[i]);
    }
}

// generic scalar fallback (any H): re-reads the row for the scale pass
template <bool ADD>
__global__ __launch_bounds__(256) void rmsnorm_scalar_kernel(const bf16_t* __restrict__ x,
                                                             const bf16_t* res_in,
                                                             const bf16_t* __restrict__ w,
                                                             bf16_t* y, bf16_t* res_out, int H,
                                                             float eps) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    float ss = 0.f;
    for (int i = threadIdx.x; i < H; i += 256) {
        float v = bf2f(x[row * H + i]);
        if (ADD) v += bf2f(res_in[row * H + i]);
        ss += v * v;
    }
    const float tot = block_sum_256(ss, red);
    const float rstd = 1.0f / sqrtf(tot / (float)H + eps);
    for (int i = threadIdx.x; i < H; i += 256) {
        float v = bf2f(x[row * H + i]);
        if (ADD) {
            v += bf2f(res_in[row * H + i]);
            // x/res_in may alias y/res_out: each element is read and written by this thread only
            res_out[row * H + i] = f2bf(v);
        }
        y[row * H + i] = f2bf(rbf(v * rstd) * bf2f(w[i]));
    }
}

// ------------------------------------------------------------------------------------------
// per-head RMSNorm (a4): LPR lanes per head-row (hd = 8*LPR), weight multiply in fp32.
template <int LPR>
__global__ __launch_bounds__(256) void head_rmsnorm_kernel(const bf16_t* __restrict__ x,
                                                           int64_t xs, const bf16_t* __restrict__ w,
                                                           bf16_t* __restrict__ y, int64_t ys,
                                                           int64_t rows, int n_heads, float eps) {
    const int64_t r = (int64_t)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    const int sub = threadIdx.x % LPR;
    const bool valid = r < rows;
    const int64_t t = valid ? r / n_heads : 0;
    const int h = valid ? (int)(r % n_heads) : 0;
    const int hd = 8 * LPR;
    u32x4_t xv = {0, 0, 0, 0};
    if (valid) xv = *reinterpret_cast<const u32x4_t*>(x + t * xs + (int64_t)h * hd + sub * 8);
    float s[8];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s[2 * j] = bflo(xv[j]);
        s[2 * j + 1] = bfhi(xv[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += s[j] * s[j];
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float rstd = 1.0f / sqrtf(ss / (float)hd + eps);
    if (valid) {
        u32x4_t wv = *reinterpret_cast<const u32x4_t*>(w + sub * 8);
        u32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = pack2bf(bflo(wv[j]) * (s[2 * j] * rstd), bfhi(wv[j]) * (s[2 * j + 1] * rstd));
        *reinterpret_cast<u32x4_t*>(y + t * ys + (int64_t)h * hd + sub * 8) = o;
    }
}

__global__ __launch_bounds__(256) void head_rmsnorm_scalar_kernel(const bf16_t* x, int64_t xs,
                                                                  const bf16_t* w, bf16_t* y,
                                                                  int64_t ys, int64_t rows,
                                                                  int n_heads, int hd, float eps) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const int64_t t = r / n_heads;
    const int h = (int)(r % n_heads);
    const bf16_t* xr = x + t * xs + (int64_t)h * hd;
    bf16_t* yr = y + t * ys + (int64_t)h * hd;
    float ss = 0.f;
    for (int i = 0; i < hd; ++i) {
        float v = bf2f(xr[i]);
        ss += v * v;
    }
    const float rstd = 1.0f / sqrtf(ss / (float)hd + eps);
    for (int i = 0; i < hd; ++i) yr[i] = f2bf(bf2f(w[i]) * (bf2f(xr[i]) * rstd));
}

// ------------------------------------------------------------------------------------------
// NeoX rotary (a6), scalar-pair formulation: one thread per (token, head, 8-element chunk).
// chunk c of a head: c < nc → first half (x1), c < 2nc → second half (x2), else pass-through,
// with nc = rotary_dim/16.  Generic fallback below handles rotary_dim % 16 != 0.
__device__ __forceinline__ void rope8(const u32x4_t a /*x1*/, const u32x4_t b /*x2*/,
                                      const float* cosp, const float* sinp, u32x4_t& y1,
                                      u32x4_t& y2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float c0 = cosp[2 * j], c1 = cosp[2 * j + 1];
        const float s0 = sinp[2 * j], s1 = sinp[2 * j + 1];
        const float a0 = bflo(a[j]), a1 = bfhi(a[j]);
        const float b0 = bflo(b[j]), b1 = bfhi(b[j]);
        y1[j] = pack2bf(a0 * c0 - b0 * s0, a1 * c1 - b1 * s1);
        y2[j] = pack2bf(b0 * c0 + a0 * s0, b1 * c1 + a1 * s1);
    }
}

__global__ __launch_bounds__(256) void rope_vec_kernel(const bf16_t* q, int64_t qs, const bf16_t* k,
                                                       int64_t ks, bf16_t* qo, int64_t qos,
                                                       bf16_t* ko, int64_t kos,
                                                       const int64_t* __restrict__ pos,
                                                       const float* __restrict__ cs, int64_t T,
                                                       int nh, int nkv, int hd, int rd) {
    // work item = (token, head in [0,nh+nkv), unit u): units 0..nc-1 rotate a chunk pair,
    // units nc.. copy one pass-through chunk each.
    const int nc = rd >> 4;
    const int npass = (hd - rd) >> 3;
    const int upr = nc + npass;
    const int64_t total = T * (int64_t)(nh + nkv) * upr;
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int u = (int)(id % upr);
    const int64_t th = id / upr;
    const int hh = (int)(th % (nh + nkv));
    const int64_t t = th / (nh + nkv);
    const bf16_t* src;
    bf16_t* dst;
    if (hh < nh) {
        src = q + t * qs + (int64_t)hh * hd;
        dst = qo + t * qos + (int64_t)hh * hd;
    } else {
        src = k + t * ks + (int64_t)(hh - nh) * hd;
        dst = ko + t * kos + (int64_t)(hh - nh) * hd;
    }
    if (u < nc) {
        const int64_t p = pos[t];
        const float* cosp = cs + p * rd + u * 8;
        const float* sinp = cosp + (rd >> 1);
        u32x4_t a = *reinterpret_cast<const u32x4_t*>(src + u * 8);
        u32x4_t b = *reinterpret_cast<const u32x4_t*>(src + (rd >> 1) + u * 8);
        u32x4_t y1, y2;
        rope8(a, b, cosp, sinp, y1, y2);
        *reinterpret_cast<u32x4_t*>(dst + u * 8) = y1;
        *reinterpret_cast<u32x4_t*>(dst + (rd >> 1) + u * 8) = y2;
    } else if (dst != src) {
        const int c = rd + (u - nc) * 8;
        *reinterpret_cast<u32x4_t*>(dst + c) = *reinterpret_cast<const u32x4_t*>(src + c);
    }
}

__global__ __launch_bounds__(256) void rope_scalar_kernel(const bf16_t* q, int64_t qs,
                                                          const bf16_t* k, int64_t ks, bf16_t* qo,
                                                          int64_t qos, bf16_t* ko, int64_t kos,
                                                          const int64_t* pos, const float* cs,
                                                          int64_t T, int nh, int nkv, int hd,
                                                          int rd) {
    // one thread per (token, head, i) with i in [0, rd/2) ∪ pass-through elements
    const int half = rd >> 1;
    const int upr = half + (hd - rd);
    const int64_t total = T * (int64_t)(nh + nkv) * upr;
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int u = (int)(id % upr);
    const int64_t th = id / upr;
    const int hh = (int)(th % (nh + nkv));
    const int64_t t = th / (nh + nkv);
    const bf16_t* src;
    bf16_t* dst;
    if (hh < nh) {
        src = q + t * qs + (int64_t)hh * hd;
        dst = qo + t * qos + (int64_t)hh * hd;
    } else {
        src = k + t * ks + (int64_t)(hh - nh) * hd;
        dst = ko + t * kos + (int64_t)(hh - nh) * hd;
    }
    if (u < half) {
        const int64_t p = pos[t];
        const float c = cs[p * rd + u], s = cs[p * rd + half + u];
        const float x1 = bf2f(src[u]), x2 = bf2f(src[half + u]);
        dst[u] = f2bf(x1 * c - x2 * s);
        dst[half + u] = f2bf(x2 * c + x1 * s);
    } else {
        const int c = rd + (u - half);
        dst[c] = src[c];
    }
}

// ------------------------------------------------------------------------------------------
// paged-KV scatter (a7): one workgroup per token; slot -1 skipped.
__global__ __launch_bounds__(256) void kv_store_kernel(const bf16_t* __restrict__ k, int64_t ks,
                                                       const bf16_t* __restrict__ v, int64_t vs,
                                                       bf16_t* __restrict__ kc,
                                                       bf16_t* __restrict__ vc,
                                                       const int32_t* __restrict__ slots, int D,
                                                       int64_t n_slots, int vec_ok) {
    const int64_t i = blockIdx.x;
    const int64_t slot = slots[i];
    if (slot < 0 || slot >= n_slots) return;  // -1 = padding (layers/attention.py:23)
    if (vec_ok) {
        const int nvec = D >> 3;
        for (int j = threadIdx.x; j < 2 * nvec; j += 256) {
            const bool isv = j >= nvec;
            const int c = isv ? j - nvec : j;
            const bf16_t* s = isv ? v + i * vs : k + i * ks;
            bf16_t* d = (isv ? vc : kc) + slot * D;
            reinterpret_cast<u32x4_t*>(d)[c] = reinterpret_cast<const u32x4_t*>(s)[c];
        }
    } else {
        for (int j = threadIdx.x; j < 2 * D; j += 256) {
            const bool isv = j >= D;
            const int c = isv ? j - D : j;
            if (isv) vc[slot * D + c] = v[i * vs + c];
            else kc[slot * D + c] = k[i * ks + c];
        }
    }
}

// ------------------------------------------------------------------------------------------
// fused split + optional per-head RMSNorm + rotary + cache scatter on the packed qkv row.
// LPH = lanes per head (hd = 8*LPH).  One LPH-lane group handles one head of one token:
// heads 0..nh-1 = q, nh..nh+nkv-1 = k, nh+nkv..nh+2nkv-1 = v.
template <int LPH>
__global__ __launch_bounds__(256) void qkv_post_kernel(
    const bf16_t* __restrict__ qkv, int64_t qkvs, bf16_t* __restrict__ qo, bf16_t* __restrict__ ko,
    bf16_t* __restrict__ vo, const bf16_t* __restrict__ qnw, const bf16_t* __restrict__ knw,
    float eps, const int64_t* __restrict__ pos, const float* __restrict__ cs, int rd,
    bf16_t* __restrict__ kc, bf16_t* __restrict__ vc, const int32_t* __restrict__ slots,
    int64_t n_slots, int64_t T, int nh, int nkv) {
    constexpr int hd = 8 * LPH;
    constexpr int TPB = 4;  // tokens per workgroup, their loads in flight together
    const int nheads = nh + 2 * nkv;
    // grid = (block of TPB tokens, block of 256 / LPH heads): no per-thread 64-bit division, and a thread moves 4 x 16 B
    // instead of 16 B — at T = 16384 the one-chunk-per-thread form launched 114688 workgroups of 8 KB each and ran at
    // 2.2 TB/s (459 us for 1 GB)
    const int hraw = (int)blockIdx.y * (256 / LPH) + (int)threadIdx.x / LPH;
    const int sub = threadIdx.x % LPH;
    const bool hvalid = hraw < nheads;
    const int hh = hvalid ? hraw : 0;
    const bool is_q = hh < nh, is_k = !is_q && hh < nh + nkv;
    const bf16_t* nw = is_q ? qnw : (is_k ? knw : nullptr);
    const int nc = rd >> 4;
    const bool rot = (is_q || is_k) && sub < 2 * nc;
    const bool to_cache = !is_q && kc != nullptr && slots != nullptr;

    u32x4_t xv[TPB];
    int64_t pp[TPB], sl[TPB];
    bool tv[TPB];
#pragma unroll
    for (int i = 0; i < TPB; ++i) {
        const int64_t t = (int64_t)blockIdx.x * TPB + i;
        tv[i] = hvalid && t < T;
        xv[i] = u32x4_t{0, 0, 0, 0};
        pp[i] = 0;
        sl[i] = -1;
        if (tv[i]) {
            xv[i] = *reinterpret_cast<const u32x4_t*>(qkv + t * qkvs + (int64_t)hh * hd + sub * 8);
            if (rot) pp[i] = pos[t];
            if (to_cache) sl[i] = slots[t];
        }
    }
#pragma unroll
    for (int i = 0; i < TPB; ++i) {
        const int64_t t = (int64_t)blockIdx.x * TPB + i;
        // optional per-head RMSNorm on q / k (a4): y = bf16(w32 * (x32 * rstd))
        if (nw != nullptr) {  // uniform per LPH-group; shuffles below run for all lanes
            float s[8];
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[2 * j] = bflo(xv[i][j]);
                s[2 * j + 1] = bfhi(xv[i][j]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += s[j] * s[j];
#pragma unroll
            for (int o = LPH / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
            const float rstd = 1.0f / sqrtf(ss / (float)hd + eps);
            u32x4_t wv = *reinterpret_cast<const u32x4_t*>(nw + sub * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                xv[i][j] = pack2bf(bflo(wv[j]) * (s[2 * j] * rstd), bfhi(wv[j]) * (s[2 * j + 1] * rstd));
        }
        // rotary on q and k: chunk `sub` pairs with chunk sub±nc
        {
            const int partner = sub < nc ? sub + nc : sub - nc;  // only meaningful for sub < 2nc
            const int src_lane = (threadIdx.x & 63) - sub + (partner & (LPH - 1));
            u32x4_t pv;
#pragma unroll
            for (int j = 0; j < 4; ++j) pv[j] = __shfl(xv[i][j], src_lane, 64);
            if (tv[i] && rot) {
                const int u = sub < nc ? sub : sub - nc;
                const float* cosp = cs + pp[i] * rd + u * 8;
                const float* sinp = cosp + (rd >> 1);
                u32x4_t y1, y2;
                if (sub < nc) {
                    rope8(xv[i], pv, cosp, sinp, y1, y2);
                    xv[i] = y1;
                } else {
                    rope8(pv, xv[i], cosp, sinp, y1, y2);
                    xv[i] = y2;
                }
            }
        }
        if (!tv[i]) continue;
        if (is_q) {
            *reinterpret_cast<u32x4_t*>(qo + (t * nh + hh) * hd + sub * 8) = xv[i];
        } else {
            const int kh = is_k ? hh - nh : hh - nh - nkv;
            bf16_t* o = is_k ? ko : vo;
            if (o != nullptr) *reinterpret_cast<u32x4_t*>(o + (t * nkv + kh) * hd + sub * 8) = xv[i];
            if (to_cache && sl[i] >= 0 && sl[i] < n_slots) {
                bf16_t* c = (is_k ? kc : vc) + (sl[i] * nkv + kh) * hd + sub * 8;
                *reinterpret_cast<u32x4_t*>(c) = xv[i];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// SiluAndMul: y = bf16( bf16(silu(g)) * u ), silu(g) = g / (1 + exp(-g)) in fp32
__device__ __forceinline__ float silu_f(float g) { return g / (1.0f + expf(-g)); }

__global__ __launch_bounds__(256) void silu_mul_kernel(const bf16_t* __restrict__ x,
                                                       bf16_t* __restrict__ y, int64_t T, int I,
                                                       int vec_ok) {
    if (vec_ok) {
        const int nvec = I >> 3;
        const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (id >= T * nvec) return;
        const int64_t t = id / nvec;
        const int c = (int)(id % nvec);
        u32x4_t g = reinterpret_cast<const u32x4_t*>(x + t * 2 * I)[c];
        u32x4_t u = reinterpret_cast<const u32x4_t*>(x + t * 2 * I + I)[c];
        u32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = pack2bf(rbf(silu_f(bflo(g[j]))) * bflo(u[j]), rbf(silu_f(bfhi(g[j]))) * bfhi(u[j]));
        reinterpret_cast<u32x4_t*>(y + t * I)[c] = o;
    } else {
        const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (id >= T * I) return;
        const int64_t t = id / I;
        const int i = (int)(id % I);
        y[id] = f2bf(rbf(silu_f(bf2f(x[t * 2 * I + i]))) * bf2f(x[t * 2 * I + I + i]));
    }
}

inline unsigned nblocks(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

template <bool ADD>
int launch_rmsnorm(const void* x, const void* res_in, const void* w, void* y, void* res_out,
                   int64_t T, int H, float eps, hipStream_t st) {
    const bf16_t* xp = (const bf16_t*)x;
    const bf16_t* rp = (const bf16_t*)res_in;
    const bf16_t* wp = (const bf16_t*)w;
    bf16_t* yp = (bf16_t*)y;
    bf16_t* op = (bf16_t*)res_out;
    const bool vec = (H % 8 == 0) && H <= 256 * 8 * 8 && glm_aligned16(x) && glm_aligned16(w) &&
                     glm_aligned16(y) && (!ADD || (glm_aligned16(res_in) && glm_aligned16(res_out)));
    dim3 grid((unsigned)T), block(256);
    if (vec) {
        const int vpt = (H / 8 + 255) / 256;
        if (vpt <= 1)
            hipLaunchKernelGGL((rmsnorm_vec_kernel<1, ADD>), grid, block, 0, st, xp, rp, wp, yp, op, H, eps);
        else if (vpt <= 2)
            hipLaunchKernelGGL((rmsnorm_vec_kernel<2, ADD>), grid, block, 0, st, xp, rp, wp, yp, op, H, eps);
        else if (vpt <= 4)
            hipLaunchKernelGGL((rmsnorm_vec_kernel<4, ADD>), grid, block, 0, st, xp, rp, wp, yp, op, H, eps);
        else
            hipLaunchKernelGGL((rmsnorm_vec_kernel<8, ADD>), grid, block, 0, st, xp, rp, wp, yp, op, H, eps);
    } else {
        hipLaunchKernelGGL((rmsnorm_scalar_kernel<ADD>), grid, block, 0, st, xp, rp, wp, yp, op, H, eps);
    }
    return GLM_OK;
}

// add-RMSNorm whose x is given as fp32 K-slice partial sums of a linear layer (glm_gemm_bf16_partial):
// x = bf16(sum_s part[s]) — the linear's own output rounding — then exactly rmsnorm_vec_kernel<ADD>.
template <int VPT>
__global__ __launch_bounds__(256) void add_rmsnorm_partial_kernel(const float* __restrict__ part, int nslices,
                                                                  int64_t slice_stride, const bf16_t* res_in,
                                                                  const bf16_t* __restrict__ w, bf16_t* y,
                                                                  bf16_t* res_out, int H, float eps) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const int nvec = H >> 3;
    const u32x4_t* rr = reinterpret_cast<const u32x4_t*>(res_in + row * H);
    const u32x4_t* wr = reinterpret_cast<const u32x4_t*>(w);
    u32x4_t* yr = reinterpret_cast<u32x4_t*>(y + row * H);
    u32x4_t* ro = reinterpret_cast<u32x4_t*>(res_out + row * H);
    float s[VPT][8];
    float ss = 0.f;
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int i = threadIdx.x + v * 256;
        if (i < nvec) {
            const float* pp = part + row * H + (int64_t)i * 8;
            f32x4_t a = *reinterpret_cast<const f32x4_t*>(pp), b = *reinterpret_cast<const f32x4_t*>(pp + 4);
            // slices are added in index order (deterministic); loads of up to 8 slices are in flight together
            for (int s0 = 1; s0 < nslices; s0 += 8) {
                f32x4_t ta[8], tb[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int sl = min(s0 + j, nslices - 1);  // clamped re-read, not added
                    ta[j] = *reinterpret_cast<const f32x4_t*>(pp + sl * slice_stride);
                    tb[j] = *reinterpret_cast<const f32x4_t*>(pp + sl * slice_stride + 4);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (s0 + j < nslices) {
                        a += ta[j];
                        b += tb[j];
                    }
            }
            const u32x4_t rv = rr[i];
            u32x4_t ov;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x0 = rbf(j < 2 ? a[2 * j] : b[2 * j - 4]);
                const float x1 = rbf(j < 2 ? a[2 * j + 1] : b[2 * j - 3]);
                s[v][2 * j] = x0 + bflo(rv[j]);
                s[v][2 * j + 1] = x1 + bfhi(rv[j]);
                ov[j] = pack2bf(s[v][2 * j], s[v][2 * j + 1]);
            }
            ro[i] = ov;
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += s[v][j] * s[v][j];
        }
    }
    const float tot = block_sum_256(ss, red);
    const float var = tot / (float)H;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int i = threadIdx.x + v * 256;
        if (i < nvec) yr[i] = norm_scale8(s[v], rstd, wr[i]);
    }
}

}  // namespace

extern "C" int glm_add_rmsnorm_partial(const float* part, int nslices, int64_t slice_stride,
                                       const void* residual_in, const void* weight, void* y, void* residual_out,
                                       int64_t T, int H, float eps, glm_stream_t stream) {
    GLM_CHECK_ARG(part && residual_in && weight && y && residual_out, GLM_ERR_INVALID,
                  "glm_add_rmsnorm_partial: null pointer");
    GLM_CHECK_ARG(T >= 0 && H > 0 && nslices >= 1 && slice_stride >= T * (int64_t)H, GLM_ERR_INVALID,
                  "glm_add_rmsnorm_partial: bad shape");
    GLM_CHECK_ARG(H % 8 == 0 && H <= 256 * 8 * 8 && glm_aligned16(part) && slice_stride % 4 == 0 &&
                      glm_aligned16(residual_in) && glm_aligned16(weight) && glm_aligned16(y) &&
                      glm_aligned16(residual_out),
                  GLM_ERR_UNSUPPORTED, "glm_add_rmsnorm_partial: needs H%%8==0, H<=16384, 16-byte aligned buffers (H=%d)", H);
    if (T == 0) return GLM_OK;
    const bf16_t* rp = (const bf16_t*)residual_in;
    const bf16_t* wp = (const bf16_t*)weight;
    bf16_t* yp = (bf16_t*)y;
    bf16_t* op = (bf16_t*)residual_out;
    hipStream_t st = glm_s(stream);
    dim3 grid((unsigned)T), block(256);
    const int vpt = (H / 8 + 255) / 256;
    if (vpt <= 1)
        hipLaunchKernelGGL((add_rmsnorm_partial_kernel<1>), grid, block, 0, st, part, nslices, slice_stride, rp, wp, yp, op, H, eps);
    else if (vpt <= 2)
        hipLaunchKernelGGL((add_rmsnorm_partial_kernel<2>), grid, block, 0, st, part, nslices, slice_stride, rp, wp, yp, op, H, eps);
    else if (vpt <= 4)
        hipLaunchKernelGGL((add_rmsnorm_partial_kernel<4>), grid, block, 0, st, part, nslices, slice_stride, rp, wp, yp, op, H, eps);
    else
        hipLaunchKernelGGL((add_rmsnorm_partial_kernel<8>), grid, block, 0, st, part, nslices, slice_stride, rp, wp, yp, op, H, eps);
    GLM_CHECK_LAUNCH("glm_add_rmsnorm_partial");
    return GLM_OK;
}

extern "C" int glm_rmsnorm(const void* x, const void* weight, void* y, int64_t T, int H, float eps,
                           glm_stream_t stream) {
    GLM_CHECK_ARG(x && weight && y, GLM_ERR_INVALID, "glm_rmsnorm: null pointer");
    GLM_CHECK_ARG(T >= 0 && H > 0, GLM_ERR_INVALID, "glm_rmsnorm: bad shape T=%ld H=%d", (long)T, H);
    if (T == 0) return GLM_OK;
    launch_rmsnorm<false>(x, nullptr, weight, y, nullptr, T, H, eps, glm_s(stream));
    GLM_CHECK_LAUNCH("glm_rmsnorm");
    return GLM_OK;
}

extern "C" int glm_add_rmsnorm(const void* x, const void* residual_in, const void* weight, void* y,
                               void* residual_out, int64_t T, int H, float eps,
                               glm_stream_t stream) {
    GLM_CHECK_ARG(x && residual_in && weight && y && residual_out, GLM_ERR_INVALID,
                  "glm_add_rmsnorm: null pointer");
    GLM_CHECK_ARG(T >= 0 && H > 0, GLM_ERR_INVALID, "glm_add_rmsnorm: bad shape");
    if (T == 0) return GLM_OK;
    launch_rmsnorm<true>(x, residual_in, weight, y, residual_out, T, H, eps, glm_s(stream));
    GLM_CHECK_LAUNCH("glm_add_rmsnorm");
    return GLM_OK;
}

extern "C" int glm_head_rmsnorm(const void* x, int64_t x_token_stride, const void* weight, void* y,
                                int64_t y_token_stride, int64_t T, int n_heads, int hd, float eps,
                                glm_stream_t stream) {
    GLM_CHECK_ARG(x && weight && y, GLM_ERR_INVALID, "glm_head_rmsnorm: null pointer");
    GLM_CHECK_ARG(T >= 0 && n_heads > 0 && hd > 0, GLM_ERR_INVALID, "glm_head_rmsnorm: bad shape");
    const int64_t rows = T * n_heads;
    if (rows == 0) return GLM_OK;
    const bf16_t* xp = (const bf16_t*)x;
    const bf16_t* wp = (const bf16_t*)weight;
    bf16_t* yp = (bf16_t*)y;
    hipStream_t st = glm_s(stream);
    const int lpr = hd / 8;
    const bool vec = hd % 8 == 0 && (lpr & (lpr - 1)) == 0 && lpr <= 64 && glm_aligned16(x) &&
                     glm_aligned16(y) && glm_aligned16(weight) && x_token_stride % 8 == 0 &&
                     y_token_stride % 8 == 0;
#define HRN(L)                                                                                   \
    hipLaunchKernelGGL((head_rmsnorm_kernel<L>), dim3(nblocks(rows, 256 / L)), dim3(256), 0, st, \
                       xp, x_token_stride, wp, yp, y_token_stride, rows, n_heads, eps)
    if (vec) {
        switch (lpr) {
            case 1: HRN(1); break;
            case 2: HRN(2); break;
            case 4: HRN(4); break;
            case 8: HRN(8); break;
            case 16: HRN(16); break;
            case 32: HRN(32); break;
            default: HRN(64); break;
        }
    } else {
        hipLaunchKernelGGL(head_rmsnorm_scalar_kernel, dim3(nblocks(rows, 256)), dim3(256), 0, st, xp,
                           x_token_stride, wp, yp, y_token_stride, rows, n_heads, hd, eps);
    }
#undef HRN
    GLM_CHECK_LAUNCH("glm_head_rmsnorm");
    return GLM_OK;
}

extern "C" int glm_rope_neox(const void* q, int64_t q_stride, const void* k, int64_t k_stride,
                             void* q_out, int64_t q_out_stride, void* k_out, int64_t k_out_stride,
                             const int64_t* positions, const float* cos_sin_cache, int64_t max_pos,
                             int64_t T, int nh, int nkv, int hd, int rotary_dim,
                             glm_stream_t stream) {
    GLM_CHECK_ARG(q && k && q_out && k_out && positions && cos_sin_cache, GLM_ERR_INVALID,
                  "glm_rope_neox: null pointer");
    GLM_CHECK_ARG(rotary_dim > 0 && rotary_dim <= hd && rotary_dim % 2 == 0 && nh >= 0 && nkv >= 0 &&
                      max_pos > 0,
                  GLM_ERR_INVALID, "glm_rope_neox: bad rotary_dim=%d hd=%d", rotary_dim, hd);
    if (T == 0 || nh + nkv == 0) return GLM_OK;
    const bf16_t* qp = (const bf16_t*)q;
    const bf16_t* kp = (const bf16_t*)k;
    bf16_t* qo = (bf16_t*)q_out;
    bf16_t* ko = (bf16_t*)k_out;
    hipStream_t st = glm_s(stream);
    const bool vec = rotary_dim % 16 == 0 && hd % 8 == 0 && glm_aligned16(q) && glm_aligned16(k) &&
                     glm_aligned16(q_out) && glm_aligned16(k_out) && q_stride % 8 == 0 &&
                     k_stride % 8 == 0 && q_out_stride % 8 == 0 && k_out_stride % 8 == 0;
    if (vec) {
        const int upr = rotary_dim / 16 + (hd - rotary_dim) / 8;
        const int64_t total = T * (int64_t)(nh + nkv) * upr;
        hipLaunchKernelGGL(rope_vec_kernel, dim3(nblocks(total, 256)), dim3(256), 0, st, qp, q_stride,
                           kp, k_stride, qo, q_out_stride, ko, k_out_stride, positions, cos_sin_cache,
                           T, nh, nkv, hd, rotary_dim);
    } else {
        const int upr = rotary_dim / 2 + (hd - rotary_dim);
        const int64_t total = T * (int64_t)(nh + nkv) * upr;
        hipLaunchKernelGGL(rope_scalar_kernel, dim3(nblocks(total, 256)), dim3(256), 0, st, qp,
                           q_stride, kp, k_stride, qo, q_out_stride, ko, k_out_stride, positions,
                           cos_sin_cache, T, nh, nkv, hd, rotary_dim);
    }
    GLM_CHECK_LAUNCH("glm_rope_neox");
    return GLM_OK;
}

extern "C" int glm_kv_store(const void* k, int64_t k_stride, const void* v, int64_t v_stride,
                            void* k_cache, void* v_cache, const int32_t* slot_mapping, int64_t N,
                            int D, int64_t n_slots, glm_stream_t stream) {
    GLM_CHECK_ARG(k && v && k_cache && v_cache && slot_mapping, GLM_ERR_INVALID,
                  "glm_kv_store: null pointer");
    GLM_CHECK_ARG(N >= 0 && D > 0 && n_slots >= 0, GLM_ERR_INVALID, "glm_kv_store: bad shape");
    if (N == 0) return GLM_OK;
    const int vec_ok = D % 8 == 0 && k_stride % 8 == 0 && v_stride % 8 == 0 && glm_aligned16(k) &&
                       glm_aligned16(v) && glm_aligned16(k_cache) && glm_aligned16(v_cache);
    hipLaunchKernelGGL(kv_store_kernel, dim3((unsigned)N), dim3(256), 0, glm_s(stream),
                       (const bf16_t*)k, k_stride, (const bf16_t*)v, v_stride, (bf16_t*)k_cache,
                       (bf16_t*)v_cache, slot_mapping, D, n_slots, vec_ok);
    GLM_CHECK_LAUNCH("glm_kv_store");
    return GLM_OK;
}

extern "C" int glm_qkv_split_norm_rope_store(const void* qkv, int64_t qkv_stride, void* q_out,
                                             void* k_out, void* v_out, const void* q_norm_w,
                                             const void* k_norm_w, float eps,
                                             const int64_t* positions, const float* cos_sin_cache,
                                             int64_t max_pos, int rotary_dim, void* k_cache,
                                             void* v_cache, const int32_t* slot_mapping,
                                             int64_t n_slots, int64_t T, int nh, int nkv, int hd,
                                             glm_stream_t stream) {
    GLM_CHECK_ARG(qkv && q_out && positions && cos_sin_cache, GLM_ERR_INVALID,
                  "glm_qkv_split_norm_rope_store: null pointer");
    GLM_CHECK_ARG((k_cache == nullptr) == (v_cache == nullptr), GLM_ERR_INVALID,
                  "glm_qkv_split_norm_rope_store: k_cache/v_cache must both be set or both NULL");
    GLM_CHECK_ARG((q_norm_w == nullptr) == (k_norm_w == nullptr), GLM_ERR_INVALID,
                  "glm_qkv_split_norm_rope_store: q_norm_w/k_norm_w must both be set or both NULL");
    const int lph = hd / 8;
    GLM_CHECK_ARG(hd % 8 == 0 && (lph & (lph - 1)) == 0 && lph >= 2 && lph <= 64 &&
                      rotary_dim % 16 == 0 && rotary_dim > 0 && rotary_dim <= hd &&
                      qkv_stride % 8 == 0 && glm_aligned16(qkv) && glm_aligned16(q_out),
                  GLM_ERR_UNSUPPORTED,
                  "glm_qkv_split_norm_rope_store: needs hd=8*2^n (<=512), rotary_dim%%16==0, 16B alignment "
                  "(hd=%d rotary_dim=%d)", hd, rotary_dim);
    if (T == 0) return GLM_OK;
    hipStream_t st = glm_s(stream);
    GLM_CHECK_ARG(T <= 0x7fffffffll, GLM_ERR_UNSUPPORTED, "glm_qkv_split_norm_rope_store: too many rows (%ld)", (long)T);
    const int nheads_all = nh + 2 * nkv;
#define QKVP(L)                                                                                      \
    hipLaunchKernelGGL((qkv_post_kernel<L>), dim3((unsigned)((T + 3) / 4), (unsigned)((nheads_all + 256 / L - 1) / (256 / L))), \
                       dim3(256), 0, st,                                                             \
                       (const bf16_t*)qkv, qkv_stride, (bf16_t*)q_out, (bf16_t*)k_out,               \
                       (bf16_t*)v_out, (const bf16_t*)q_norm_w, (const bf16_t*)k_norm_w, eps,        \
                       positions, cos_sin_cache, rotary_dim, (bf16_t*)k_cache, (bf16_t*)v_cache,     \
                       slot_mapping, n_slots, T, nh, nkv)
    switch (lph) {
        case 2: QKVP(2); break;
        case 4: QKVP(4); break;
        case 8: QKVP(8); break;
        case 16: QKVP(16); break;
        case 32: QKVP(32); break;
        default: QKVP(64); break;
    }
#undef QKVP
    GLM_CHECK_LAUNCH("glm_qkv_split_norm_rope_store");
    return GLM_OK;
}

extern "C" int glm_silu_and_mul(const void* x, void* y, int64_t T, int I, glm_stream_t stream) {
    GLM_CHECK_ARG(x && y, GLM_ERR_INVALID, "glm_silu_and_mul: null pointer");
    GLM_CHECK_ARG(T >= 0 && I > 0, GLM_ERR_INVALID, "glm_silu_and_mul: bad shape");
    if (T == 0) return GLM_OK;
    const int vec_ok = I % 8 == 0 && glm_aligned16(x) && glm_aligned16(y);
    const int64_t n = vec_ok ? T * (I / 8) : T * (int64_t)I;
    hipLaunchKernelGGL(silu_mul_kernel, dim3(nblocks(n, 256)), dim3(256), 0, glm_s(stream),
                       (const bf16_t*)x, (bf16_t*)y, T, I, vec_ok);
    GLM_CHECK_LAUNCH("glm_silu_and_mul");
    return GLM_OK;
}

// ------------------------------------------------------------------------------------------
// N4: exact expansion of OCP fp8 e4m3 weights to bf16 (every e4m3 value is representable in bf16), one 16-element
// vector per thread.  Prefill runs the MFMA-bound tiled GEMM on the expanded copy (weight bytes do not matter there) and
// applies the per-row dequantisation scale to the fp32 accumulator, exactly as the streaming decode kernel does.
namespace {
__global__ __launch_bounds__(256) void fp8_expand_kernel(const unsigned char* __restrict__ q, int64_t ld_q,
                                                         int64_t gs_q, bf16_t* __restrict__ out, int64_t ld_out,
                                                         int64_t gs_out, int rows, int cols) {
    const int vpr = cols >> 4;  // 16-element vectors per row
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= (int64_t)rows * vpr) return;
    const int64_t g = blockIdx.y;
    const int64_t r = v / vpr;
    const int c = (int)(v % vpr) * 16;
    const u32x4_t w = *reinterpret_cast<const u32x4_t*>(q + g * gs_q + r * ld_q + c);
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    u32x4_t o[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2_t a = __builtin_amdgcn_cvt_pk_f32_fp8(w[i], false), b = __builtin_amdgcn_cvt_pk_f32_fp8(w[i], true);
        o[i >> 1][2 * (i & 1)] = pack2bf(a[0], a[1]);
        o[i >> 1][2 * (i & 1) + 1] = pack2bf(b[0], b[1]);
    }
    u32x4_t* dst = reinterpret_cast<u32x4_t*>(out + g * gs_out + r * ld_out + c);
    dst[0] = o[0];
    dst[1] = o[1];
}
}  // namespace

int glm_launch_fp8_expand(const void* q, int64_t ld_q, int64_t gs_q, void* out, int64_t ld_out, int64_t gs_out, int G,
                          int rows, int cols, hipStream_t st) {
    GLM_CHECK_ARG(cols % 16 == 0 && ld_q % 16 == 0 && gs_q % 16 == 0 && ld_out % 8 == 0 && gs_out % 8 == 0 &&
                      glm_aligned16(q) && glm_aligned16(out),
                  GLM_ERR_UNSUPPORTED, "fp8 expand: needs cols%%16==0 and 16-byte aligned rows");
    const int64_t n = (int64_t)rows * (cols / 16);
    hipLaunchKernelGGL(fp8_expand_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)G), dim3(256), 0, st,
                       (const unsigned char*)q, ld_q, gs_q, (bf16_t*)out, ld_out, gs_out, rows, cols);
    GLM_CHECK_LAUNCH("fp8 expand");
    return GLM_OK;
}



// ------------------------------------------------------------------------------------------------------------------
// r05, N4 / BASELINE configs[4]: dynamic per-row quantisation of activation rows to OCP fp8 e4m3 for the fp8 x fp8 expert
// GEMM (csrc/gemm_largem.hip, LmCfgC88):  s[t] = max(amax(|x[t]|) / 448, 2^-126),  q[t, :] = RNE(clamp(x[t, :] / s[t], +-448)).
// One wave per row (H / 8 sixteen-byte pieces over its lanes, H <= 16384), the row is read once and kept in registers.
// No reference counterpart (its quantised path is int4 AWQ dequant-then-GEMM, layers/linear_awq_new.py:254-255).
constexpr int QF8_MAX_VEC = 32;   // 16-byte pieces per lane: H <= 64 * 8 * 32 = 16384
__global__ __launch_bounds__(256) void quantize_fp8_rows_kernel(const bf16_t* __restrict__ x, int64_t ldx,
                                                                unsigned char* __restrict__ q, int64_t ldq,
                                                                float* __restrict__ scale, int64_t T, int H) {
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const int nvec = H >> 3;
    const u32x4_t* xr = reinterpret_cast<const u32x4_t*>(x + t * ldx);
    u32x4_t v[QF8_MAX_VEC];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < QF8_MAX_VEC; ++i) {
        const int c = lane + 64 * i;
        v[i] = u32x4_t{0, 0, 0, 0};
        if (c < nvec) {
            v[i] = xr[c];
#pragma unroll
            for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(bflo(v[i][j])), fabsf(bfhi(v[i][j]))));
        }
    }
    amax = wave_max(amax);
    const float s = fmaxf(amax / 448.0f, 1.17549435e-38f);
    if (lane == 0) scale[t] = s;
    unsigned char* qr = q + t * ldq;
#pragma unroll
    for (int i = 0; i < QF8_MAX_VEC; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) {
            float f[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f[2 * j] = fminf(fmaxf(bflo(v[i][j]) / s, -448.0f), 448.0f);
                f[2 * j + 1] = fminf(fmaxf(bfhi(v[i][j]) / s, -448.0f), 448.0f);
            }
            int d0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
            d0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], d0, true);
            int d1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
            d1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], d1, true);
            *reinterpret_cast<u32x2_t*>(qr + (int64_t)c * 8) = u32x2_t{(unsigned)d0, (unsigned)d1};
        }
    }
}

extern "C" int glm_quantize_fp8_rows(const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int64_t T, int H,
                                     glm_stream_t stream) {
    GLM_CHECK_ARG(x && q && scale, GLM_ERR_INVALID, "glm_quantize_fp8_rows: null pointer");
    GLM_CHECK_ARG(T >= 0 && H > 0 && ldx >= H && ldq >= H, GLM_ERR_INVALID, "glm_quantize_fp8_rows: bad shape");
    GLM_CHECK_ARG(H % 8 == 0 && H <= 64 * 8 * QF8_MAX_VEC && ldx % 8 == 0 && ldq % 8 == 0 && glm_aligned16(x) && (((uintptr_t)q) & 7) == 0,
                  GLM_ERR_UNSUPPORTED, "glm_quantize_fp8_rows: needs H %% 8 == 0, H <= 16384, 16-byte aligned rows (H=%d)", H);
    if (T == 0) return GLM_OK;
    hipLaunchKernelGGL(quantize_fp8_rows_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, glm_s(stream), (const bf16_t*)x, ldx,
                       (unsigned char*)q, ldq, scale, T, H);
    GLM_CHECK_LAUNCH("glm_quantize_fp8_rows");
    return GLM_OK;
}
