// overlap2_probe.hip -- do the matrix pipe and the VALU overlap ACROSS waves of one SIMD?  (measurement helper, not product code)
// One workgroup of 512 threads per CU (8 waves: wave w sits on SIMD w & 3, so every SIMD holds exactly one wave of each role).
// Waves 0-3 ("matrix") run a dependent chain of v_mfma_f32_32x32x16_bf16; waves 4-7 ("vector") run independent v_fma_f32 chains.
// Timed with hipEvents over the whole launch: matrix waves alone (vector waves exit at once), vector waves alone, both roles.
// If both ~ max(alone, alone) the pipes overlap across waves (role-specialised waves would pay); if both ~ sum they do not.
// Also: both roles in EVERY wave, alternating phases of PH instructions (the shape of the fused layers: build phase, MFMA phase).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/peak/overlap2_probe tools/peak/overlap2_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// mode 0: roles by wave (mask bit 0: matrix waves run, bit 1: vector waves run); mode 1: every wave alternates MFMA / VALU phases
template <int NV>
__global__ __launch_bounds__(512) void probe(int mode, int mask, int iters, int nm, float *sink, float seed) {
    const int wave = threadIdx.x >> 6;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = seed * r;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + (threadIdx.x & 63) * 0.001f + i); b[i] = (__bf16)(seed - i * 0.5f); }
    float v[NV];
    for (int i = 0; i < NV; ++i) v[i] = seed + i + threadIdx.x;
    const float m = 1.0001f * seed, c = 0.5f;
    if (mode == 0) {
        const bool matrix = wave < 4;
        if (matrix && (mask & 1)) {
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        } else if (!matrix && (mask & 2)) {
            for (int it = 0; it < iters * nm; ++it)
#pragma unroll
                for (int k = 0; k < 8; ++k)
#pragma unroll
                    for (int f = 0; f < NV; ++f) v[f] = __builtin_fmaf(v[f], m, c);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            if (mask & 1) {
#pragma unroll
                for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (mask & 2) {
                for (int j = 0; j < nm; ++j)
#pragma unroll
                    for (int k = 0; k < 8; ++k)
#pragma unroll
                        for (int f = 0; f < NV; ++f) v[f] = __builtin_fmaf(v[f], m, c);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int i = 0; i < NV; ++i) s += v[i];
    sink[blockIdx.x * 512 + threadIdx.x] = s;
}

static float run(int mode, int mask, int iters, int nm, float *ds) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<8>), dim3(256), dim3(512), 0, 0, mode, mask, iters, nm, ds, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<8>), dim3(256), dim3(512), 0, 0, mode, mask, iters, nm, ds, 1.0f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}

int main() {
    float *ds;
    hipMalloc(&ds, 256 * 512 * sizeof(float));
    const int iters = 4000;
    // nm: vector work per 16 MFMAs = nm * 64 v_fma (16 MFMAs = 512 pipe cycles; 64 v_fma = 256 issue cycles)
    for (int nm : {1, 2, 3}) {
        const float tm = run(0, 1, iters, nm, ds), tv = run(0, 2, iters, nm, ds), tb = run(0, 3, iters, nm, ds);
        printf("roles by wave, %3d v_fma per 16 MFMAs: matrix waves alone %8.1f us, vector waves alone %8.1f us, both %8.1f us (sum %8.1f, max %8.1f)\n",
               nm * 64, tm, tv, tb, tm + tv, tm > tv ? tm : tv);
    }
    for (int nm : {1, 2, 3}) {
        const float tm = run(1, 1, iters, nm, ds), tv = run(1, 2, iters, nm, ds), tb = run(1, 3, iters, nm, ds);
        printf("phases in every wave (2 waves per SIMD), %3d v_fma per 16 MFMAs: MFMA phases only %8.1f us, VALU phases only %8.1f us, both %8.1f us (sum %8.1f, max %8.1f)\n",
               nm * 64, tm, tv, tb, tm + tv, tm > tv ? tm : tv);
    }
    return 0;
}
