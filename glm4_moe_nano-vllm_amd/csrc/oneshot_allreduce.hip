// oneshot_allreduce.hip — one-shot all-reduce(sum) of a small bf16 tensor over directly mapped peer buffers (N3).
//
// Why: the two RowParallel all-reduces of a decoder layer (layers/linear.py:153-154) move [T, H] bf16 = 256 KiB at T=32,
// 2 MiB at T=256.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring is latency- and per-link bound at these
// sizes, while a one-shot exchange uses all 7 links at once in one hop.  PUSH form (r03): every rank WRITES its tensor into
// a receive slot of every peer (posted writes, no read round trip over the fabric), signals the peers, waits for their
// signals and sums the `world` contributions out of ITS OWN memory, in ascending rank order — so every rank computes
// bit-identical sums, which the replicated router needs.
//
// State, all in device memory of the owning rank (`state`, 4 words, zero before the first call):
//   [0] epoch   — number of completed calls.  A call works on e = epoch + 1; the LAST workgroup to finish stores e back.
//                 The epoch is never a host value: a HIP graph that captured k calls advances it by k on EVERY replay
//                 (the r02 form baked a host integer into the graph and replays repeated old epochs).
//   [1] arrive  — workgroups of this call whose pushes are complete (self-resetting)
//   [2] done    — workgroups of this call that have finished (self-resetting)
//   [3] status  — 0, or q + 1 after a call gave up waiting for rank q
// Receive buffer of rank r (mapped into every peer): [2 slots][world senders][n_max] bf16; a call uses slot e & 1.
// Flags of rank r (mapped into every peer): flag[q] = last epoch whose data from rank q is complete in r's buffer.
//
// Protocol of call e on rank r:
//   1. every workgroup stores its slice of `in` into rbuf[p][e & 1][r] of every peer p (nontemporal stores), drains them;
//      thread 0 releases at system scope and adds to `arrive`; the workgroup that arrives last stores flag[p][r] = e on
//      every peer ("rank r's data of epoch e is complete in p's memory");
//   2. thread 0 of every workgroup waits until (int)(flag[r][q] - e) >= 0 for every q (bounded spin, relaxed system-scope
//      loads, then ONE system-scope acquire).  The comparison is monotonic: a peer that has already moved on to e + 1
//      still satisfies it (the r02 form compared for equality and a late workgroup could spin to the bound);
//   3. it sums its slice over q = 0 .. world-1 in ascending order in fp32 (own contribution straight from `in`), one bf16
//      rounding, into `out` (may alias `in`);
//   4. `done` count; the last workgroup stores epoch = e.
// Why two slots are enough: a peer can only be at epoch e + 1 (writing slot (e+1) & 1) while a late workgroup of this
// rank still reads slot e & 1; it cannot reach e + 2, because finishing e + 1 needs THIS rank's flag of e + 1, and this
// rank's call e + 1 starts only after every workgroup of call e has finished (stream order; graph replays keep it).
// A wait that expires sets status, poisons this workgroup's output slice with NaNs (a failed all-reduce must never look
// like data) and still counts as done, so the epochs of all ranks stay aligned; TPGroup.check() raises at the engine's
// next synchronisation point.
//
// Evidence: world = 1; two "ranks" on two streams of ONE device, one of them delayed; two PROCESSES on one device over
// IPC under HIP-graph replay with an odd number of calls per graph (tools/oneshot_allreduce_2proc.py).  It has NOT run
// across GPUs (one-GPU lease); RCCL stays the default (GLM_ONESHOT_ALLREDUCE=1 opts in).
#include "common.h"
#include <stdio.h>
#include <string.h>
#include <mutex>
#include <vector>

namespace {
constexpr int OS_MAX_WORLD = 8;
constexpr unsigned OS_SPIN_LIMIT = 1u << 20;  // ~1 s at one relaxed system-scope load + s_sleep per iteration

struct OsPeers {
    bf16_t* rbuf[OS_MAX_WORLD];     // receive buffers [2][world][n_max], as mapped in THIS process (entry `rank` = own)
    unsigned* flag[OS_MAX_WORLD];   // flag arrays [OS_MAX_WORLD]
};

__global__ __launch_bounds__(256) void oneshot_allreduce_kernel(const bf16_t* in, bf16_t* out, const OsPeers peers, int rank,
                                                                int world, int64_t n, int64_t n_max, unsigned* state) {
    const int64_t nvec = n >> 3;
    const unsigned e = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    const int64_t slot_off = (int64_t)(e & 1u) * world * n_max;
    // 1. push my rows into every peer's receive slot
    for (int p = 0; p < world; ++p) {
        if (p == rank) continue;
        bf16_t* dst = peers.rbuf[p] + slot_off + (int64_t)rank * n_max;
        for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256)
            __builtin_nontemporal_store(reinterpret_cast<const u32x4_t*>(in)[v], reinterpret_cast<u32x4_t*>(dst) + v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int go;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: the data sits in the peers' memory
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned got = __hip_atomic_fetch_add(state + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (got == gridDim.x - 1u) {  // every workgroup of this rank has pushed
            __hip_atomic_store(state + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int p = 0; p < world; ++p)
                __hip_atomic_store(peers.flag[p] + rank, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // 2. wait for every rank's signal (monotonic comparison: a peer already at e + 1 passes)
        int ok = 1;
        for (int q = 0; q < world && ok; ++q) {
            if (q == rank) continue;  // own rows are read from `in`
            unsigned spins = 0;
            while ((int)(__hip_atomic_load(peers.flag[rank] + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > OS_SPIN_LIMIT) {
                    ok = 0;
                    __hip_atomic_store(state + 3, 1u + (unsigned)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        go = ok;
    }
    __syncthreads();
    const bool ok = go != 0;
    // 3. reduce in ascending rank order out of this rank's own memory (poison on a failed wait)
    const bf16_t* mine = peers.rbuf[rank] + slot_off;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
        u32x4_t o = {0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u};  // bf16 NaN pairs
        if (ok) {
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = 0.f;
            for (int q = 0; q < world; ++q) {
                const u32x4_t x = q == rank ? reinterpret_cast<const u32x4_t*>(in)[v]
                                            : __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(mine + (int64_t)q * n_max) + v);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[2 * i] += bflo(x[i]);
                    acc[2 * i + 1] += bfhi(x[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = pack2bf(acc[2 * i], acc[2 * i + 1]);
        }
        reinterpret_cast<u32x4_t*>(out)[v] = o;
    }
    // 4. the last workgroup to finish advances the epoch (in device memory: graph replays advance it too)
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned got = __hip_atomic_fetch_add(state + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (got == gridDim.x - 1u) {
            __hip_atomic_store(state + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(state, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
}  // namespace

// ---- communication memory (host-only helpers; r04) ------------------------------------------------------------------
// The receive buffer and the flag words are written by PEER GPUs over xGMI and polled / read by the owner INSIDE a running
// kernel.  Ordinary hipMalloc memory is coarse-grained: its lines may sit in the owner's L2 across the whole kernel, and a
// same-device test (two streams or two processes on one GPU share that L2) can never show a stale line.  So this memory is
// allocated UNCACHED (hipExtMallocWithFlags(hipDeviceMallocUncached): MTYPE UC, every access goes to memory) or, on
// request, fine-grained (coherent at system scope).  Which accesses of oneshot_allreduce_kernel rely on it:
//   * the owner's relaxed system-scope polls of flag[rank][q] (written by rank q's release store over the fabric);
//   * the owner's loads of rbuf[rank][slot][q] behind the acquire fence (written by rank q's pushes);
// the peers' stores themselves travel over the fabric either way.  The memory is exported / opened with hipIpc so that it
// can be mapped by the other ranks' processes exactly like the torch storage it replaces.
// Blocks handed back by glm_comm_free are kept for the life of the process and reused by glm_comm_alloc (same device, kind and
// size) instead of going back to the runtime.  r04: with hipFree here, the GPU test suite saw STALE READS of ordinary cached
// tensors later in the same process — 4-KiB pieces, by host copies and by kernels alike, gone once the caches had been evicted
// (profiles/r04/c1_stale_readback.txt) — in every run that had allocated and freed MiB-sized uncached blocks before, and in none
// that had not: pages that were mapped uncached come back to the allocator PyTorch draws its next segments from.
// Communication buffers are long-lived objects anyway; a block never changes its mapping again this way.
namespace {
struct CommBlock { void* p; size_t bytes; int kind; int dev; };
std::mutex g_comm_mu;
std::vector<CommBlock> g_comm_free;   // parked blocks
std::vector<CommBlock> g_comm_live;   // blocks handed out (to know size / kind / device at free time)
}  // namespace

extern "C" int glm_comm_alloc(void** ptr, size_t bytes, int kind) {
    GLM_CHECK_ARG(ptr && bytes > 0 && (kind == GLM_COMM_UNCACHED || kind == GLM_COMM_FINEGRAINED), GLM_ERR_INVALID,
                  "glm_comm_alloc: bad arguments");
    int dev = 0;
    (void)hipGetDevice(&dev);
    void* p = nullptr;
    hipError_t e = hipSuccess;
    {
        std::lock_guard<std::mutex> lk(g_comm_mu);
        for (size_t i = 0; i < g_comm_free.size(); ++i)
            if (g_comm_free[i].bytes == bytes && g_comm_free[i].kind == kind && g_comm_free[i].dev == dev) {
                p = g_comm_free[i].p;
                g_comm_free.erase(g_comm_free.begin() + (long)i);
                break;
            }
    }
    const bool fresh = p == nullptr;
    if (fresh) e = hipExtMallocWithFlags(&p, bytes, kind == GLM_COMM_UNCACHED ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
    if (e == hipSuccess) e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        // a fresh block that could not be cleared goes back to the runtime (it was never handed out, nothing maps it); a
        // REUSED block is parked again — freeing blocks that were mapped uncached is what this pool exists to avoid
        if (p && fresh) (void)hipFree(p);
        else if (p) {
            std::lock_guard<std::mutex> lk(g_comm_mu);
            g_comm_free.push_back(CommBlock{p, bytes, kind, dev});
        }
        (void)hipGetLastError();
        glm_set_error("glm_comm_alloc(%zu B, kind %d): %s", bytes, kind, hipGetErrorString(e));
        return GLM_ERR_LAUNCH;
    }
    {
        std::lock_guard<std::mutex> lk(g_comm_mu);
        g_comm_live.push_back(CommBlock{p, bytes, kind, dev});
    }
    *ptr = p;
    return GLM_OK;
}

extern "C" int glm_comm_free(void* ptr) {
    if (!ptr) return GLM_OK;
    std::lock_guard<std::mutex> lk(g_comm_mu);
    for (size_t i = 0; i < g_comm_live.size(); ++i)
        if (g_comm_live[i].p == ptr) {
            g_comm_free.push_back(g_comm_live[i]);  // parked, not returned to the runtime (see above)
            g_comm_live.erase(g_comm_live.begin() + (long)i);
            // the pool only grows with DISTINCT (device, kind, size) triples — a serving process has two (buffer, flags);
            // say so once if something allocates and frees ever-changing sizes
            size_t parked = 0;
            for (const CommBlock& b : g_comm_free) parked += b.bytes;
            static bool warned = false;
            if (!warned && (g_comm_free.size() > 64 || parked > ((size_t)1 << 30))) {
                warned = true;
                fprintf(stderr, "[glm4moe_hip] glm_comm_free: %zu parked communication blocks, %zu MiB — blocks are never "
                                "returned to the runtime; allocate them once at start-up\n", g_comm_free.size(), parked >> 20);
            }
            return GLM_OK;
        }
    glm_set_error("glm_comm_free: %p was not allocated by glm_comm_alloc (or freed twice)", ptr);
    return GLM_ERR_INVALID;
}

extern "C" int glm_comm_ipc_export(const void* ptr, void* handle) {
    static_assert(sizeof(hipIpcMemHandle_t) == GLM_COMM_IPC_HANDLE_BYTES, "handle size is part of the ABI");
    GLM_CHECK_ARG(ptr && handle, GLM_ERR_INVALID, "glm_comm_ipc_export: null pointer");
    hipError_t e = hipIpcGetMemHandle((hipIpcMemHandle_t*)handle, const_cast<void*>(ptr));
    GLM_CHECK_ARG(e == hipSuccess, GLM_ERR_LAUNCH, "glm_comm_ipc_export: %s", hipGetErrorString(e));
    return GLM_OK;
}

extern "C" int glm_comm_ipc_open(const void* handle, void** ptr) {
    GLM_CHECK_ARG(ptr && handle, GLM_ERR_INVALID, "glm_comm_ipc_open: null pointer");
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    GLM_CHECK_ARG(e == hipSuccess, GLM_ERR_LAUNCH, "glm_comm_ipc_open: %s", hipGetErrorString(e));
    *ptr = p;
    return GLM_OK;
}

extern "C" int glm_comm_ipc_close(void* ptr) {
    if (!ptr) return GLM_OK;
    hipError_t e = hipIpcCloseMemHandle(ptr);
    GLM_CHECK_ARG(e == hipSuccess, GLM_ERR_LAUNCH, "glm_comm_ipc_close: %s", hipGetErrorString(e));
    return GLM_OK;
}

extern "C" size_t glm_allreduce_oneshot_buffer_elems(int world, int64_t n_max) {
    return (size_t)2 * (size_t)(world > 0 ? world : 0) * (size_t)(n_max > 0 ? n_max : 0);
}

extern "C" int glm_allreduce_oneshot(const void* in, void* out, void* const* peer_bufs, void* const* peer_flags, int rank,
                                     int world, int64_t n, int64_t n_max, void* state, glm_stream_t stream) {
    GLM_CHECK_ARG(in && out && peer_bufs && peer_flags && state, GLM_ERR_INVALID, "glm_allreduce_oneshot: null pointer");
    GLM_CHECK_ARG(world >= 1 && world <= OS_MAX_WORLD && rank >= 0 && rank < world && n >= 0 && n <= n_max,
                  GLM_ERR_INVALID, "glm_allreduce_oneshot: bad arguments (world=%d rank=%d n=%ld n_max=%ld)", world, rank,
                  (long)n, (long)n_max);
    GLM_CHECK_ARG(n % 8 == 0 && n_max % 8 == 0 && glm_aligned16(in) && glm_aligned16(out) && glm_aligned16(state),
                  GLM_ERR_UNSUPPORTED, "glm_allreduce_oneshot: needs n %% 8 == 0 and 16-byte aligned tensors");
    if (n == 0) return GLM_OK;  // NOTE: every rank must skip the same calls (n is the same on all ranks)
    OsPeers peers;
    for (int p = 0; p < OS_MAX_WORLD; ++p) {
        peers.rbuf[p] = (bf16_t*)(p < world ? peer_bufs[p] : nullptr);
        peers.flag[p] = (unsigned*)(p < world ? peer_flags[p] : nullptr);
        GLM_CHECK_ARG(p >= world || (peers.rbuf[p] && peers.flag[p] && glm_aligned16(peers.rbuf[p])), GLM_ERR_INVALID,
                      "glm_allreduce_oneshot: peer %d not mapped", p);
    }
    // few workgroups: the tensor is small; a workgroup waits for PEERS only (never for a workgroup of its own launch that
    // might not be resident yet), so the grid size is a bandwidth choice, not a correctness one
    int grid = (int)((n / 8 + 255) / 256);
    if (grid > 64) grid = 64;
    hipLaunchKernelGGL(oneshot_allreduce_kernel, dim3((unsigned)grid), dim3(256), 0, glm_s(stream), (const bf16_t*)in,
                       (bf16_t*)out, peers, rank, world, n, n_max, (unsigned*)state);
    GLM_CHECK_LAUNCH("glm_allreduce_oneshot");
    return GLM_OK;
}
