// mfma_peak.hip -- measurement helper (not part of the product library): what the bf16 matrix pipe sustains on this chip for the
// instruction stream shape the fused layers use (v_mfma_f32_32x32x16_bf16 chains on one accumulator), as a function of resident
// waves per SIMD, the number of independent accumulator chains per wave, and whether the A operand is re-read from LDS per MFMA
// group.  Built by tools/peak/build.sh into build/libmfma_peak.so; driven by tools/mfma_peak.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CHAINS, bool LDS_A, bool F32>
__global__ __launch_bounds__(256) void peak_kernel(int iters, float *out) {
    extern __shared__ uint4 lds[];
    const int lane = threadIdx.x & 63;
    for (int e = threadIdx.x; e < 64 * 24; e += blockDim.x) {
        if (iters < 0) {  // random mantissas and signs, exponents near 1: the operand bits toggle like real data (power), sums stay finite
            uint32_t hsh = (uint32_t)e * 2654435761u;
            uint32_t v[4];
            for (int i = 0; i < 4; ++i) {
                hsh = hsh * 1664525u + 1013904223u;
                v[i] = (hsh & 0x807F807Fu) | 0x3C003C00u | ((hsh >> 7) & 0x01800180u);
            }
            lds[e] = make_uint4(v[0], v[1], v[2], v[3]);
        } else {
            lds[e] = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);
        }
    }
    if (iters < 0) iters = -iters;
    __syncthreads();
    f32x16 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    uint4 a = lds[lane], b = lds[64 * 23 + lane];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 24; ++s) {  // 24 MFMAs per chain per iteration = one 64-channel output tile of the split layers
            if (LDS_A && (s % 6) == 0) { a = lds[(s / 6 * 3) * 64 + lane]; b = lds[(s / 6 * 3 + 1) * 64 + lane]; }
#pragma unroll
            for (int c = 0; c < CHAINS; ++c)
                if (F32)  // v_mfma_f32_32x32x2_f32: one dword of each operand per MFMA
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(s & 1 ? a.y : a.x), __uint_as_float(s & 2 ? b.y : b.x), acc[c], 0, 0, 0);
                else
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[c], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 12345.f) out[0] = s;
}

// waves_per_simd in {1,2,4}: workgroups of 256 threads, one / two / four per CU (LDS padding limits residency); grid = 256 CUs * that
extern "C" __attribute__((visibility("default"))) int mfma_peak(int chains, int lds_a, int waves_per_simd, int iters, float *out, void *stream, int f32) {
    const size_t lds = waves_per_simd == 1 ? 100 * 1024 : waves_per_simd == 2 ? 70 * 1024 : 36 * 1024;
    const dim3 grid(256 * waves_per_simd), block(256);
#define LAUNCH2(C, L, F)                                                                                                        \
    {                                                                                                                        \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(peak_kernel<C, L, F>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((peak_kernel<C, L, F>), grid, block, lds, (hipStream_t)stream, iters, out);                            \
    }
#define LAUNCH(C, L) { if (f32) LAUNCH2(C, L, true) else LAUNCH2(C, L, false) }
    if (chains == 1 && !lds_a) LAUNCH(1, false)
    else if (chains == 1) LAUNCH(1, true)
    else if (chains == 2 && !lds_a) LAUNCH(2, false)
    else if (chains == 2) LAUNCH(2, true)
    else if (chains == 4 && !lds_a) LAUNCH(4, false)
    else LAUNCH(4, true)
    return (int)hipGetLastError();
}
