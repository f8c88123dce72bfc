// sync_latency.hip -- what one cross-workgroup exchange costs on MI355X, against one in-workgroup exchange (measurement helper,
// not part of the library).  FPS iterations are dependent: a multi-workgroup FPS would pay one cross-workgroup exchange per iteration.
//   mode 0: two workgroups (different CUs) ping-pong through L2: WG a does an agent-scope atomic store of `i`, WG b spins on an
//           agent-scope atomic load until it sees it and answers the same way; `iters` round trips.
//   mode 1: the same ping-pong between two WAVES of one workgroup through LDS (ds atomics + spin).
//   mode 2: one wave, `iters` dependent global atomic adds with return (L2 atomic round trip).
//   mode 3: one workgroup of 1024 threads, `iters` x (LDS atomic max + __syncthreads + LDS read): the exchange FPS does today.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void sync_latency_kernel(int mode, int iters, unsigned *flags, unsigned long long *cycles) {
    __shared__ unsigned lflag[2];
    __shared__ unsigned long long slot[3];
    const int tid = threadIdx.x;
    if (tid < 2) lflag[tid] = 0;
    if (tid < 3) slot[tid] = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (mode == 0) {
        if (tid == 0) {
            const int me = blockIdx.x, other = 1 - me;
            for (int i = 1; i <= iters; ++i) {
                if (me == 0) {
                    __hip_atomic_store(&flags[0], (unsigned)i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    while (__hip_atomic_load(&flags[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)i) {}
                } else {
                    while (__hip_atomic_load(&flags[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)i) {}
                    __hip_atomic_store(&flags[1], (unsigned)i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
                (void)other;
            }
        }
    } else if (mode == 1) {
        const int wave = tid >> 6;
        if ((tid & 63) == 0 && wave < 2) {
            for (int i = 1; i <= iters; ++i) {
                if (wave == 0) {
                    __hip_atomic_store(&lflag[0], (unsigned)i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    while (__hip_atomic_load(&lflag[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)i) {}
                } else {
                    while (__hip_atomic_load(&lflag[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)i) {}
                    __hip_atomic_store(&lflag[1], (unsigned)i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    } else if (mode == 2) {
        if (tid == 0 && blockIdx.x == 0) {
            unsigned v = 0;
            for (int i = 0; i < iters; ++i) v = atomicAdd(&flags[2], v + 1);
            flags[3] = v;
        }
    } else {
        unsigned long long key = (unsigned long long)tid;
        int cur = 0, nxt = 1;
        for (int i = 0; i < iters; ++i) {
            if ((tid & 63) == 0) __hip_atomic_fetch_max(&slot[cur], key + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (tid == 0) slot[nxt] = 0;
            __syncthreads();
            key += slot[cur] & 1;
            const int nn = 3 - cur - nxt;
            cur = nxt;
            nxt = nn;
        }
        if (key == 12345) flags[3] = 1;
    }
    __syncthreads();
    if (tid == 0) cycles[blockIdx.x] = __builtin_readcyclecounter() - t0;
}

extern "C" int sync_latency(int mode, int iters, unsigned *flags, unsigned long long *cycles, void *stream) {
    const int blocks = mode == 0 ? 2 : 1;
    const int threads = mode == 3 ? 1024 : (mode == 1 ? 128 : 64);
    hipLaunchKernelGGL(sync_latency_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, mode, iters, flags, cycles);
    return (int)hipGetLastError();
}
