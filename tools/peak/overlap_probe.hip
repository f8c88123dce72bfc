// overlap_probe.hip -- can VALU instructions of the SAME wave execute under a dependent chain of v_mfma_f32_32x32x16_bf16?
// One wave per SIMD (grid = 1024 waves); per iteration: 16 MFMAs on one accumulator and F independent v_fma_f32 per MFMA placed
// between them, the order pinned with sched_group_barrier pipelines.  Prints cycles per MFMA (s_memtime) for F = 0, 2, 4, 6, 8, 12 and for the
// VALU work alone.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/peak/overlap_probe tools/peak/overlap_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int F, bool MFMA>
__global__ __launch_bounds__(64) void probe(unsigned long long *cycles, float *sink, int iters, float seed) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = seed * r;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 0.001f + i); b[i] = (__bf16)(seed - i * 0.5f); }
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = seed + i + threadIdx.x;
    const float m = 1.0001f * seed, c = 0.5f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (MFMA) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int f = 0; f < F; ++f) v[f] = __builtin_fmaf(v[f], m, c);
        }
        // the order wanted: one MFMA, then its F vector instructions, sixteen times (a plain sched_barrier(0) per slot did not hold:
        // the compiler issued the MFMAs in bursts of five)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (MFMA) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (F > 0) __builtin_amdgcn_sched_group_barrier(0x002, F, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int i = 0; i < 12; ++i) s += v[i];
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int F, bool MFMA>
void run(const char *name, unsigned long long *dc, float *ds, int grid) {
    const int iters = 2000;
    hipLaunchKernelGGL((probe<F, MFMA>), dim3(grid), dim3(64), 0, 0, dc, ds, iters, 1.0f);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((probe<F, MFMA>), dim3(grid), dim3(64), 0, 0, dc, ds, iters, 1.0f);
    hipDeviceSynchronize();
    static unsigned long long h[4096];
    hipMemcpy(h, dc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < grid; ++i) sum += (double)h[i];
    printf("%-34s waves %4d  %.1f cycles per slot (16 slots per iteration)\n", name, grid, sum / grid / iters / 16.0);
}

int main() {
    unsigned long long *dc;
    float *ds;
    hipMalloc(&dc, 4096 * sizeof(unsigned long long));
    hipMalloc(&ds, 4096 * 64 * sizeof(float));
    for (int grid : {1024, 2048}) {   // one / two waves per SIMD
        run<0, true>("MFMA only", dc, ds, grid);
        run<2, true>("MFMA + 2 v_fma per slot", dc, ds, grid);
        run<4, true>("MFMA + 4 v_fma per slot", dc, ds, grid);
        run<6, true>("MFMA + 6 v_fma per slot", dc, ds, grid);
        run<8, true>("MFMA + 8 v_fma per slot", dc, ds, grid);
        run<12, true>("MFMA + 12 v_fma per slot", dc, ds, grid);
        run<4, false>("4 v_fma per slot alone", dc, ds, grid);
        run<8, false>("8 v_fma per slot alone", dc, ds, grid);
        run<12, false>("12 v_fma per slot alone", dc, ds, grid);
    }
    return 0;
}
