// persistent_layer_probe.hip — what a ONE-LAUNCH decode layer at batch 1 would pay for its seams on this chip (r04, VERDICT r03
// item 3).  Geometry of the single-sequence GEMV body (csrc/gemm_tinym.hip): 256 workgroups x 512 threads, 84 KiB of LDS each
// (one workgroup per CU).  Measured:
//   (a) a grid barrier inside such a launch — flat counter (relaxed sc1 poll, release fence before the arrive, acquire fence
//       after) and the XCD-hierarchical form (per-XCD counter, the XCD's last arriver goes to the top counter, leaders publish
//       a generation word) — per barrier, from a launch that executes N of them back to back and from one with ~20 us of
//       streaming (a 4 KiB-per-lane read loop) between two barriers;
//   (b) the host/graph cost of launching that grid cooperatively (hipLaunchCooperativeKernel) against a plain launch, eager
//       and replayed from a captured graph.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/persistent_layer_probe.hip -o /tmp/plp && /tmp/plp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <functional>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int NWG = 256, NT = 512;
constexpr unsigned SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// flat barrier: one monotonic counter; generation g completes when counter == g * NWG
__device__ __forceinline__ bool barrier_flat(unsigned* ctr, unsigned gen) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (ld_relaxed(ctr) < gen * NWG) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { ok = false; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    return ok;
}

// XCD-hierarchical barrier: xcc[8] per-XCD arrival counters, top counter, gen word per XCD (each on its own 128-B line)
__device__ __forceinline__ bool barrier_xcd(unsigned* base, unsigned gen, int xcc, int per_xcc) {
    unsigned* xc = base + 32 * xcc;          // arrivals on this XCD
    unsigned* top = base + 32 * 8;           // XCDs complete
    unsigned* gw = base + 32 * (9 + xcc);    // generation published to this XCD's workgroups
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned got = __hip_atomic_fetch_add(xc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (got == gen * per_xcc - 1) {  // this XCD's last arriver
            const unsigned t = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == gen * 8 - 1) {      // last XCD: release everybody
                for (int x = 0; x < 8; ++x) __hip_atomic_store(base + 32 * (9 + x), gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        unsigned spins = 0;
        while (ld_relaxed(gw) < gen) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { ok = false; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    return ok;
}

// mode 0: no barrier (work only), 1: flat, 2: xcd.  work_kb: KiB streamed per workgroup between two barriers.
__global__ __launch_bounds__(NT) void probe_kernel(unsigned* sync, const uint4* stream, size_t stream_vecs, int n_bar, int mode,
                                                   int work_vecs_per_thread, unsigned* status, float* sink) {
    extern __shared__ char smem[];
    (void)smem;
    unsigned xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    const int xcc = (int)(xcc_id & 7);
    const int per_xcc = NWG / 8;
    float acc = 0.f;
    size_t pos = ((size_t)blockIdx.x * NT + threadIdx.x);
    for (int b = 1; b <= n_bar; ++b) {
        for (int i = 0; i < work_vecs_per_thread; ++i) {
            typedef __attribute__((ext_vector_type(4))) unsigned u4_t;
            const u4_t v = __builtin_nontemporal_load(reinterpret_cast<const u4_t*>(stream) + (pos % stream_vecs));
            acc += __uint_as_float(v[0]) + __uint_as_float(v[3]);
            pos += (size_t)NWG * NT;
        }
        bool ok = true;
        if (mode == 1) ok = barrier_flat(sync, (unsigned)b);
        else if (mode == 2) ok = barrier_xcd(sync, (unsigned)b, xcc, per_xcc);
        if (!ok) {
            if (threadIdx.x == 0) status[0] = 1;
            break;
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

__global__ void trivial_kernel(unsigned* p) { if (threadIdx.x == 0 && blockIdx.x == 0 && p[0] == 0xffffffffu) p[1] = 1; }

static float time_us(hipStream_t st, int reps, const std::function<void()>& fn) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    fn(); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < reps; ++i) fn();
    CK(hipEventRecord(b, st));
    CK(hipStreamSynchronize(st));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3f / reps;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    unsigned *sync, *status; float* sink; uint4* stream;
    const size_t stream_bytes = 1ull << 30;  // 1 GiB: far beyond the Infinity Cache
    CK(hipMalloc(&sync, 4096 * 4)); CK(hipMalloc(&status, 64)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&stream, stream_bytes));
    CK(hipMemset(stream, 1, stream_bytes)); CK(hipMemset(status, 0, 64));
    const size_t vecs = stream_bytes / 16;
    const int lds = 84 * 1024;
    CK(hipFuncSetAttribute((const void*)probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe_kernel, NT, lds));
    printf("geometry: %d workgroups x %d threads, %d KiB LDS, occupancy API %d per CU\n", NWG, NT, lds / 1024, occ);
    auto run = [&](int n_bar, int mode, int work) {
        CK(hipMemsetAsync(sync, 0, 4096 * 4, st));
        hipLaunchKernelGGL(probe_kernel, dim3(NWG), dim3(NT), lds, st, sync, stream, vecs, n_bar, mode, work, status, sink);
    };
    // (a) barriers back to back: per-barrier cost = (t(N) - t(launch with 0 barriers)) / N
    for (int mode = 1; mode <= 2; ++mode) {
        const float t0 = time_us(st, 20, [&] { run(0, mode, 0); });
        const float t1 = time_us(st, 20, [&] { run(200, mode, 0); });
        printf("barrier %-4s back to back: %.2f us per barrier (launch of 200: %.1f us, empty launch + memset: %.1f us)\n",
               mode == 1 ? "flat" : "xcd", (t1 - t0) / 200.f, t1, t0);
    }
    // ... with streaming between barriers: 16 vectors of 16 B per thread = 32 MiB per phase over the chip (~5-6 us of HBM time);
    // 64 vectors = 128 MiB (~20 us): per-barrier cost = (t(barriers) - t(no barriers)) / N at equal bytes
    for (int work : {16, 64}) {
        const float tw = time_us(st, 10, [&] { run(40, 0, work); });
        for (int mode = 1; mode <= 2; ++mode) {
            const float tb = time_us(st, 10, [&] { run(40, mode, work); });
            printf("phase of %3d MiB + barrier %-4s: %.2f us per phase, %.2f us without the barrier => %.2f us per barrier\n",
                   (int)((size_t)work * 16 * NWG * NT >> 20), mode == 1 ? "flat" : "xcd", tb / 40.f, tw / 40.f, (tb - tw) / 40.f);
        }
    }
    unsigned h = 0; CK(hipMemcpy(&h, status, 4, hipMemcpyDeviceToHost));
    printf("spin status: %u (0 = every barrier completed)\n", h);
    // (b) cooperative vs plain launch of a trivial 256 x 512 grid, eager and from a graph of 5 such launches
    void* args[] = {&sync};
    const float pe = time_us(st, 200, [&] { hipLaunchKernelGGL(trivial_kernel, dim3(NWG), dim3(NT), 0, st, sync); });
    const float ce = time_us(st, 200, [&] { CK(hipLaunchCooperativeKernel((const void*)trivial_kernel, dim3(NWG), dim3(NT), args, 0, st)); });
    printf("eager: plain launch %.2f us, cooperative launch %.2f us (per launch, queue kept full)\n", pe, ce);
    for (int coop = 0; coop <= 1; ++coop) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 5; ++i) {
            if (coop) CK(hipLaunchCooperativeKernel((const void*)trivial_kernel, dim3(NWG), dim3(NT), args, 0, st));
            else hipLaunchKernelGGL(trivial_kernel, dim3(NWG), dim3(NT), 0, st, sync);
        }
        hipError_t e = hipStreamEndCapture(st, &g);
        if (e != hipSuccess) { printf("graph capture of %s launches failed: %s\n", coop ? "cooperative" : "plain", hipGetErrorString(e)); continue; }
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        const float t = time_us(st, 100, [&] { CK(hipGraphLaunch(ge, st)); });
        printf("graph of 5 %s launches: %.2f us per replay = %.2f us per launch\n", coop ? "cooperative" : "plain", t, t / 5.f);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
