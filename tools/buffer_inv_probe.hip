// buffer_inv_probe.hip — does an agent-scope ACQUIRE fence (buffer_inv sc1) executed by one wave discard dirty L2 lines that
// other waves of the same XCD stored and nobody has written back yet?  (r04: the one fence of that kind in the library sat in
// the post-attention norm + router launch for odd top-k, and the only configuration with an odd top-k — C1, k = 1 — lost
// 4-KiB pieces of its K/V cache and residual rows when run behind the rest of the GPU suite.)
// Every workgroup writes its own 4-KiB piece of a buffer (value = piece index + 1), drains its stores, and then, by `mode`:
//   0 nothing   1 lane 0 of wave 0: fence(acquire, agent)   2 lane 0: fence(release, agent) then fence(acquire, agent)
//   3 as 1 but the stores are NOT drained first
// The host then reads the buffer back and counts pieces that do not hold their value.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/buffer_inv_probe.hip -o /tmp/bip && /tmp/bip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u4;

__global__ __launch_bounds__(256) void probe(u4* buf, int mode, int spin) {
    const unsigned v = blockIdx.x + 1;
    buf[(size_t)blockIdx.x * 256 + threadIdx.x] = u4{v, v, v, v};
    if (mode != 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0) {
        if (mode == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (mode >= 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
}
// a later kernel that touches none of the data: lane 0 of every workgroup executes the fence(s)
__global__ __launch_bounds__(256) void fence_only(unsigned* sink, int with_release) {
    if (threadIdx.x == 0) {
        if (with_release) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (sink[0] == 0xdeadbeefu) sink[1] = 1;
    }
}
__global__ void zero(u4* buf, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = u4{0, 0, 0, 0};
}

int main() {
    const int pieces = 8192;  // 32 MiB
    u4* buf; CK(hipMalloc(&buf, (size_t)pieces * 4096));
    std::vector<unsigned> h((size_t)pieces * 1024);
    for (int spin : {0, 8}) {
        for (int mode = 0; mode <= 3; ++mode) {
            long bad_pieces = 0, bad_launches = 0;
            for (int rep = 0; rep < 20; ++rep) {
                hipLaunchKernelGGL(zero, dim3(1024), dim3(256), 0, 0, buf, (size_t)pieces * 256);
                hipLaunchKernelGGL(probe, dim3(pieces), dim3(256), 0, 0, buf, mode, spin);
                CK(hipMemcpy(h.data(), buf, (size_t)pieces * 4096, hipMemcpyDeviceToHost));
                long b = 0;
                for (int p = 0; p < pieces; ++p) {
                    bool ok = true;
                    for (int i = 0; i < 1024; i += 37) ok &= h[(size_t)p * 1024 + i] == (unsigned)p + 1;
                    b += !ok;
                }
                bad_pieces += b; bad_launches += b != 0;
            }
            printf("spin %d mode %d: %ld of 20 launches lost data, %ld pieces of %d in total\n", spin, mode, bad_launches, bad_pieces, pieces * 20);
        }
    }
    // three launches back to back: zero fill, the writer (no fence), then a kernel that only executes fences
    unsigned* sink; CK(hipMalloc(&sink, 64)); CK(hipMemset(sink, 0, 64));
    for (int with_release = 0; with_release <= 1; ++with_release) {
        long bad_pieces = 0, bad_launches = 0;
        for (int rep = 0; rep < 40; ++rep) {
            hipLaunchKernelGGL(zero, dim3(1024), dim3(256), 0, 0, buf, (size_t)pieces * 256);
            hipLaunchKernelGGL(probe, dim3(pieces), dim3(256), 0, 0, buf, 0, 0);
            hipLaunchKernelGGL(fence_only, dim3(2048), dim3(256), 0, 0, sink, with_release);
            CK(hipMemcpy(h.data(), buf, (size_t)pieces * 4096, hipMemcpyDeviceToHost));
            long b = 0;
            for (int p = 0; p < pieces; ++p) {
                bool ok = true;
                for (int i = 0; i < 1024; i += 37) ok &= h[(size_t)p * 1024 + i] == (unsigned)p + 1;
                b += !ok;
            }
            bad_pieces += b; bad_launches += b != 0;
        }
        printf("zero -> write -> fence-only kernel (%s): %ld of 40 sequences lost data, %ld pieces in total\n",
               with_release ? "release + acquire" : "acquire only", bad_launches, bad_pieces);
    }
    return 0;
}
