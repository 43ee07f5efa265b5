// oneshot_allreduce.hip — one-shot all-reduce(sum) of a small bf16 tensor over directly mapped peer buffers (N3).
//
// Why: the two RowParallel all-reduces of a decoder layer (layers/linear.py:153-154) move [T, H] bf16 = 256 KiB at T=32,
// 2 MiB at T=256.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring is latency- and per-link bound at these
// sizes, while a one-shot exchange uses all 7 links at once in one hop — every rank publishes its tensor in a buffer the
// peers have mapped, signals them, waits for their signals and sums all `world` buffers itself, in ascending rank order
// (so every rank computes bit-identical sums, which the replicated router needs).
//
// Protocol per call (epoch e = 1, 2, ... chosen by the caller, the same on every rank):
//   1. every workgroup copies its slice of `in` into THIS rank's comm buffer with write-through (sc0 sc1) stores and
//      drains them; the workgroups meet at a local counter; the last one stores flag[peer][rank] = e on every peer
//      (system-scope release) — "rank's data for epoch e is complete";
//   2. every workgroup waits until flag[rank][p] == e for all p (bounded spin, relaxed system-scope loads, one acquire);
//   3. it sums its slice over the peers' buffers in rank order in fp32 and stores bf16.
// The comm buffers alternate between two halves by epoch parity, so epoch e+1's writes never race epoch e's reads; the
// local counter is self-resetting.  Every spin is bounded: on expiry the kernel sets *status and returns (the caller sees
// GLM_ERR_LAUNCH at its next sync-free check, never a hung GPU).
//
// Status: compiled, unit-tested with world = 1 and with two "ranks" on two streams of ONE device (peer pointers = the
// other stream's buffers).  It has NOT run across GPUs (the builder's lease is one GPU); RCCL stays the default and the
// only measured path (nanovllm/distributed/oneshot_allreduce.py, GLM_ONESHOT_ALLREDUCE=1 to opt in).
#include "common.h"

namespace {
constexpr int OS_MAX_WORLD = 8;
constexpr unsigned OS_SPIN_LIMIT = 1u << 20;  // ~1 s at one relaxed system-scope load + s_sleep per iteration

struct OsPeers {
    bf16_t* buf[OS_MAX_WORLD];      // peer comm buffers (2 halves of n_max elements each), as mapped in THIS process
    unsigned* flag[OS_MAX_WORLD];   // peer flag arrays [OS_MAX_WORLD]
};

__global__ __launch_bounds__(256) void oneshot_allreduce_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                                const OsPeers peers, int rank, int world, int64_t n,
                                                                int64_t n_max, unsigned epoch, unsigned* local_cnt,
                                                                unsigned* status) {
    const int64_t nvec = n >> 3;
    const int64_t half = (int64_t)(epoch & 1u) * n_max;
    bf16_t* mine = peers.buf[rank] + half;
    // 1. publish
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
        const u32x4_t x = reinterpret_cast<const u32x4_t*>(in)[v];
        __builtin_nontemporal_store(x, reinterpret_cast<u32x4_t*>(mine) + v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int go;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: the peers read over xGMI
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned got = __hip_atomic_fetch_add(local_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (got == gridDim.x - 1u) {  // every workgroup of this rank has published
            __hip_atomic_store(local_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int p = 0; p < world; ++p)
                __hip_atomic_store(peers.flag[p] + rank, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // 2. wait for every rank's signal
        int ok = 1;
        for (int p = 0; p < world && ok; ++p) {
            unsigned spins = 0;
            while (__hip_atomic_load(peers.flag[rank] + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != epoch) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > OS_SPIN_LIMIT) {
                    ok = 0;
                    __hip_atomic_store(status, 1u + (unsigned)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        go = ok;
    }
    __syncthreads();
    if (!go) return;
    // 3. reduce in ascending rank order
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        for (int p = 0; p < world; ++p) {
            const u32x4_t x = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(peers.buf[p] + half) + v);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[2 * i] += bflo(x[i]);
                acc[2 * i + 1] += bfhi(x[i]);
            }
        }
        u32x4_t o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = pack2bf(acc[2 * i], acc[2 * i + 1]);
        reinterpret_cast<u32x4_t*>(out)[v] = o;
    }
}
}  // namespace

extern "C" int glm_allreduce_oneshot(const void* in, void* out, void* const* peer_bufs, void* const* peer_flags, int rank,
                                     int world, int64_t n, int64_t n_max, unsigned epoch, void* local_counter,
                                     void* status, glm_stream_t stream) {
    GLM_CHECK_ARG(in && out && peer_bufs && peer_flags && local_counter && status, GLM_ERR_INVALID,
                  "glm_allreduce_oneshot: null pointer");
    GLM_CHECK_ARG(world >= 1 && world <= OS_MAX_WORLD && rank >= 0 && rank < world && n >= 0 && n <= n_max && epoch != 0,
                  GLM_ERR_INVALID, "glm_allreduce_oneshot: bad arguments (world=%d rank=%d n=%ld epoch=%u)", world, rank,
                  (long)n, epoch);
    GLM_CHECK_ARG(n % 8 == 0 && n_max % 8 == 0 && glm_aligned16(in) && glm_aligned16(out), GLM_ERR_UNSUPPORTED,
                  "glm_allreduce_oneshot: needs n %% 8 == 0 and 16-byte aligned tensors");
    if (n == 0) return GLM_OK;
    OsPeers peers;
    for (int p = 0; p < OS_MAX_WORLD; ++p) {
        peers.buf[p] = (bf16_t*)(p < world ? peer_bufs[p] : nullptr);
        peers.flag[p] = (unsigned*)(p < world ? peer_flags[p] : nullptr);
        GLM_CHECK_ARG(p >= world || (peers.buf[p] && peers.flag[p] && glm_aligned16(peers.buf[p])), GLM_ERR_INVALID,
                      "glm_allreduce_oneshot: peer %d not mapped", p);
    }
    // few workgroups: the tensor is small and every workgroup must be resident while it waits for the peers
    int grid = (int)((n / 8 + 255) / 256);
    if (grid > 64) grid = 64;
    hipLaunchKernelGGL(oneshot_allreduce_kernel, dim3((unsigned)grid), dim3(256), 0, glm_s(stream), (const bf16_t*)in,
                       (bf16_t*)out, peers, rank, world, n, n_max, epoch, (unsigned*)local_counter, (unsigned*)status);
    GLM_CHECK_LAUNCH("glm_allreduce_oneshot");
    return GLM_OK;
}
