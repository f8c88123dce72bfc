// fusion.hip -- fused 64-neighbour attentive fusion (MultiFrameEstimatier.knn_group + fusion,
// mocopci.py:798-819) for gfx950.
//
// Reference formulation at N=8192: gather (B,N,64,3), build (B,4,N,64) features, three
// Conv2d+BatchNorm+ReLU layers 4->64->64->128 materialising (B,128,N,64) fp32 (2 GiB at B=8),
// channel max, softmax over the 64 neighbours, weighted sum of neighbour coordinates.
// Here one wave owns one point at a time and nothing leaves registers:
//   * neighbours sit on the MFMA column (lane & 31), two column tiles of 32 per point;
//   * each layer is X_out[ch x nb] = W[ch x ch_in] . X_in[ch_in x nb] on v_mfma_f32_32x32x2_f32
//     (exact fp32, k-ordered fma chain); the accumulator tile of one layer IS the B operand of the
//     next: register r of lane-half h holds channel (r&3)+8(r>>2)+4h, so k-step r pairs that
//     register of the two halves, and the weight (A) operand is pre-permuted in LDS to match --
//     no LDS round trip and no cross-lane movement between layers;
//   * bias enters as the accumulator's initial value; BatchNorm (eval) is folded into W,b by the host;
//   * layer 3 is consumed one 32-channel row tile at a time into a running channel max, so the
//     (128 x 64) activation never exists;
//   * softmax over the 64 neighbours and the weighted coordinate sum are wave reductions.
// Per point: 392 MFMAs (1.6 MFLOP); HBM traffic is the compulsory idx + coordinates + output.
//
// fp32 on the bf16 matrix pipe (the f32-input MFMA form exists only in -DMCP_AB builds, for A/B runs): layers 2 and 3
// evaluate every fp32 product from the exact three-way bf16 split of mfma_split.h -- six bf16 MFMAs per 16 k-values instead
// of eight f32-input ones.  The weight pieces are split once while staging to LDS; an activation tile is split in registers
// right where the previous layer's ReLU leaves it, in the accumulator layout.
#include <stdlib.h>

#include "common.h"
#include "mfma_split.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int C1 = 64, C2 = 64, NB = 64;  // layer widths 4 -> 64 -> 64 -> 128  // mocopci.py:749-755, fusion k = 32 + 32
constexpr int WAVES = 4;


__device__ __forceinline__ int chan_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

#ifdef MCP_AB  // A/B builds only (make AB=1): the f32-input MFMA form of the kernel, 2.54 ms against 1.30 ms per 24x8192 launch
// LDS image (floats):
//   w1 [2 tiles][2 ksteps][64 lanes]                      256
//   w2 [2 tiles][8 kquads][64 lanes][4]                  4096
//   w3 [4 tiles][8 kquads][64 lanes][4]                  8192
//   b1 [2 tiles][2 halves][16], b2 same, b3 [4][2][16]    64 + 64 + 128
constexpr int OFF_W1 = 0, OFF_W2 = 256, OFF_W3 = OFF_W2 + 4096, OFF_B1 = OFF_W3 + 8192, OFF_B2 = OFF_B1 + 64,
              OFF_B3 = OFF_B2 + 64, LDS_FLOATS = OFF_B3 + 128;
__global__ __launch_bounds__(64 * WAVES, 2) void fusion_kernel(long long total, int n, const float *__restrict__ p1,
                                                            const float *__restrict__ p2, const int *__restrict__ idx, const int *__restrict__ idx2,
                                                            const float *__restrict__ w1, const float *__restrict__ b1,
                                                            const float *__restrict__ w2, const float *__restrict__ b2,
                                                            const float *__restrict__ w3, const float *__restrict__ b3,
                                                            float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    const int tid = threadIdx.x;
    // ---- stage the permuted weights once per workgroup ----
    for (int e = tid; e < 256; e += 64 * WAVES) {  // w1: [t][s][lane] = W1[32t + (lane&31)][2s + (lane>>5)]
        const int lane = e & 63, s = (e >> 6) & 1, t = e >> 7;
        lds[OFF_W1 + e] = w1[(32 * t + (lane & 31)) * 4 + 2 * s + (lane >> 5)];
    }
    for (int e = tid; e < 4096; e += 64 * WAVES) {  // w2: [t][q][lane][j]: k-step s=4q+j -> tin=s>>4, r=s&15
        const int j = e & 3, lane = (e >> 2) & 63, q = (e >> 8) & 7, t = e >> 11;
        const int s = 4 * q + j, tin = s >> 4, r = s & 15;
        lds[OFF_W2 + e] = w2[(32 * t + (lane & 31)) * C1 + 32 * tin + chan_of(r, lane >> 5)];
    }
    for (int e = tid; e < 8192; e += 64 * WAVES) {
        const int j = e & 3, lane = (e >> 2) & 63, q = (e >> 8) & 7, t = e >> 11;
        const int s = 4 * q + j, tin = s >> 4, r = s & 15;
        lds[OFF_W3 + e] = w3[(32 * t + (lane & 31)) * C2 + 32 * tin + chan_of(r, lane >> 5)];
    }
    for (int e = tid; e < 64; e += 64 * WAVES) {  // biases: [t][h][r]
        const int r = e & 15, h = (e >> 4) & 1, t = e >> 5;
        lds[OFF_B1 + e] = b1[32 * t + chan_of(r, h)];
        lds[OFF_B2 + e] = b2[32 * t + chan_of(r, h)];
    }
    for (int e = tid; e < 128; e += 64 * WAVES) {
        const int r = e & 15, h = (e >> 4) & 1, t = e >> 5;
        lds[OFF_B3 + e] = b3[32 * t + chan_of(r, h)];
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    const float4 *w2q = reinterpret_cast<const float4 *>(lds + OFF_W2);
    const float4 *w3q = reinterpret_cast<const float4 *>(lds + OFF_W3);

    const McpUnits units = mcp_units_by_xcd(total, WAVES);   // XCD x takes the x-th eighth of the points (common.h)
    for (long long p = units.first + wave; p < units.limit; p += units.stride) {
        const long long bb = mcp_div(p, n, mcp_fits32(total));
        const float cx = p1[p * 3 + 0], cy = p1[p * 3 + 1], cz = p1[p * 3 + 2];
        float score[2], nbx[2], nby[2], nbz[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            // idx2 != NULL: the two 32-neighbour halves come as separate (B,N,32) lists (no concatenation pass by the caller)
            const int id = idx2 ? (ct ? idx2 : idx)[p * 32 + col] : idx[p * NB + 32 * ct + col];
            const float *q = p2 + ((long long)bb * n + id) * 3;
            const float x = q[0], y = q[1], z = q[2];
            nbx[ct] = x; nby[ct] = y; nbz[ct] = z;
            const float rx = x - cx, ry = y - cy, rz = z - cz;
            const float dist = sqrtf((rx * rx + ry * ry) + rz * rz);
            const float in0 = h ? ry : rx, in1 = h ? dist : rz;  // k-step 0: (dx,dy); k-step 1: (dz,|d|)
            // ---- layer 1: 4 -> 64 ----
            f32x16 a1[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = lds[OFF_B1 + (t * 2 + h) * 16 + r];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[OFF_W1 + (t * 2 + 0) * 64 + lane], in0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[OFF_W1 + (t * 2 + 1) * 64 + lane], in1, acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f);
                a1[t] = acc;
            }
            // ---- layer 2: 64 -> 64 ----
            f32x16 a2[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = lds[OFF_B2 + (t * 2 + h) * 16 + r];
#pragma unroll
                for (int q4 = 0; q4 < 8; ++q4) {
                    const float4 w = w2q[(t * 8 + q4) * 64 + lane];
                    const int tin = q4 >> 2, r0 = (q4 & 3) * 4;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, a1[tin][r0 + 0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, a1[tin][r0 + 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, a1[tin][r0 + 2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, a1[tin][r0 + 3], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f);
                a2[t] = acc;
            }
            // ---- layer 3: 64 -> 128, consumed into the channel max (max over ReLU = ReLU of max) ----
            float m = 0.f;
#pragma unroll 1
            for (int t = 0; t < 4; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = lds[OFF_B3 + (t * 2 + h) * 16 + r];
#pragma unroll
                for (int q4 = 0; q4 < 8; ++q4) {
                    const float4 w = w3q[(t * 8 + q4) * 64 + lane];
                    const int tin = q4 >> 2, r0 = (q4 & 3) * 4;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, a2[tin][r0 + 0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, a2[tin][r0 + 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, a2[tin][r0 + 2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, a2[tin][r0 + 3], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[r]);
            }
            score[ct] = fmaxf(m, __shfl_xor(m, 32));  // the two lane halves hold the other 64 channels
        }
        // ---- softmax over the 64 neighbours + weighted coordinate sum (each neighbour appears in both halves) ----
        const float mx = wave_max(fmaxf(score[0], score[1]));
        const float e0 = expf(score[0] - mx), e1 = expf(score[1] - mx);
        const float den = wave_sum(e0 + e1);
        const float sx = wave_sum(e0 * nbx[0] + e1 * nbx[1]);
        const float sy = wave_sum(e0 * nby[0] + e1 * nby[1]);
        const float sz = wave_sum(e0 * nbz[0] + e1 * nbz[1]);
        if (lane == 0) {
            out[p * 3 + 0] = sx / den;
            out[p * 3 + 1] = sy / den;
            out[p * 3 + 2] = sz / den;
        }
    }
}
#endif  // MCP_AB


// ---- split-bf16 build ------------------------------------------------------------------------------------------------------
// LDS image of the split build: w1 / biases as above (fp32), then per layer [t_out][kstep][piece][lane] x 16 B
constexpr int SP_W1 = 0, SP_B1 = 256, SP_B2 = SP_B1 + 64, SP_B3 = SP_B2 + 64, SP_F32_FLOATS = SP_B3 + 128;  // fp32 part (floats)
constexpr int SP_W2_U4 = 2 * 4 * 3 * 64, SP_W3_U4 = 4 * 4 * 3 * 64;                                         // uint4 counts
constexpr size_t SP_LDS_BYTES = SP_F32_FLOATS * 4 + (size_t)(SP_W2_U4 + SP_W3_U4) * 16;

__global__ __launch_bounds__(64 * WAVES, 2) void fusion_split_kernel(long long total, int n, const float *__restrict__ p1,
                                                                  const float *__restrict__ p2, const int *__restrict__ idx, const int *__restrict__ idx2,
                                                                  const float *__restrict__ w1, const float *__restrict__ b1,
                                                                  const float *__restrict__ w2, const float *__restrict__ b2,
                                                                  const float *__restrict__ w3, const float *__restrict__ b3,
                                                                  float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    uint4 *w2s = reinterpret_cast<uint4 *>(lds + SP_F32_FLOATS);
    uint4 *w3s = w2s + SP_W2_U4;
    const int tid = threadIdx.x;
    for (int e = tid; e < 256; e += 64 * WAVES) {  // w1 (fp32, K = 4): [t][s][lane] = W1[32t + (lane&31)][2s + (lane>>5)]
        const int lane = e & 63, s = (e >> 6) & 1, t = e >> 7;
        lds[SP_W1 + e] = w1[(32 * t + (lane & 31)) * 4 + 2 * s + (lane >> 5)];
    }
    mcp_split_weights(w2s, w2, C1, 2, tid, 64 * WAVES);
    mcp_split_weights(w3s, w3, C2, 4, tid, 64 * WAVES);
    for (int e = tid; e < 64; e += 64 * WAVES) {  // biases: [t][h][r]
        const int r = e & 15, h = (e >> 4) & 1, t = e >> 5;
        lds[SP_B1 + e] = b1[32 * t + chan_of(r, h)];
        lds[SP_B2 + e] = b2[32 * t + chan_of(r, h)];
    }
    for (int e = tid; e < 128; e += 64 * WAVES) {
        const int r = e & 15, h = (e >> 4) & 1, t = e >> 5;
        lds[SP_B3 + e] = b3[32 * t + chan_of(r, h)];
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    const McpUnits units = mcp_units_by_xcd(total, WAVES);   // XCD x takes the x-th eighth of the points (common.h)
    for (long long p = units.first + wave; p < units.limit; p += units.stride) {
        const long long bb = mcp_div(p, n, mcp_fits32(total));
        const float cx = p1[p * 3 + 0], cy = p1[p * 3 + 1], cz = p1[p * 3 + 2];
        float score[2], nbx[2], nby[2], nbz[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            // idx2 != NULL: the two 32-neighbour halves come as separate (B,N,32) lists (no concatenation pass by the caller)
            const int id = idx2 ? (ct ? idx2 : idx)[p * 32 + col] : idx[p * NB + 32 * ct + col];
            const float *q = p2 + ((long long)bb * n + id) * 3;
            const float x = q[0], y = q[1], z = q[2];
            nbx[ct] = x; nby[ct] = y; nbz[ct] = z;
            const float rx = x - cx, ry = y - cy, rz = z - cz;
            const float dist = sqrtf((rx * rx + ry * ry) + rz * rz);
            const float in0 = h ? ry : rx, in1 = h ? dist : rz;
            // ---- layer 1: 4 -> 64 on the f32-input MFMA (K = 4: two k-steps), split for layer 2 as it is produced ----
            McpSplit3 x1[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = lds[SP_B1 + (t * 2 + h) * 16 + r];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[SP_W1 + (t * 2 + 0) * 64 + lane], in0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[SP_W1 + (t * 2 + 1) * 64 + lane], in1, acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f);
                x1[2 * t + 0] = mcp_split_kstep(acc, 0);
                x1[2 * t + 1] = mcp_split_kstep(acc, 1);
            }
            // ---- layer 2: 64 -> 64 ----
            McpSplit3 x2[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = lds[SP_B2 + (t * 2 + h) * 16 + r];
                acc = mcp_tile_split<4>(w2s + (size_t)t * 4 * 3 * 64 + lane, x1, acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f);
                x2[2 * t + 0] = mcp_split_kstep(acc, 0);
                x2[2 * t + 1] = mcp_split_kstep(acc, 1);
            }
            // ---- layer 3: 64 -> 128, consumed into the channel max ----
            float m = 0.f;
#pragma unroll 1
            for (int t = 0; t < 4; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = lds[SP_B3 + (t * 2 + h) * 16 + r];
                acc = mcp_tile_split<4>(w3s + (size_t)t * 4 * 3 * 64 + lane, x2, acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[r]);
            }
            score[ct] = fmaxf(m, __shfl_xor(m, 32));
        }
        const float mx = wave_max(fmaxf(score[0], score[1]));
        const float e0 = expf(score[0] - mx), e1 = expf(score[1] - mx);
        const float den = wave_sum(e0 + e1);
        const float sx = wave_sum(e0 * nbx[0] + e1 * nbx[1]);
        const float sy = wave_sum(e0 * nby[0] + e1 * nby[1]);
        const float sz = wave_sum(e0 * nbz[0] + e1 * nbz[1]);
        if (lane == 0) {
            out[p * 3 + 0] = sx / den;
            out[p * 3 + 1] = sy / den;
            out[p * 3 + 2] = sz / den;
        }
    }
}

}  // namespace

MCP_EXPORT int mcp_fusion(int b, int n, int nb, const float *p1, const float *p2, const int *idx, const int *idx2, const float *w1,
                          const float *b1, const float *w2, const float *b2, const float *w3, const float *b3, float *out,
                          mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && p1 && p2 && idx && w1 && b1 && w2 && b2 && w3 && b3 && out);
    if (nb != NB) return MCP_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const long long total = (long long)b * n;
    // 8 rounds of workgroups over the 512 resident slots rather than 2: the points are dealt out statically, so when another
    // stream's kernels hold some CUs (the next step's FPS chains take 16-24 of them, workgroups that leave no room for one of
    // these) a 2-round grid ends on the slowest slots -- measured 1.63 ms in the step against 1.34 ms alone; 8 rounds let the
    // dispatcher rebalance (1.38 ms in the step) and cost nothing alone (the weight staging is ~1 % of a 48-point workgroup).
    long long grid_cap = 4096;
#ifdef MCP_AB
    static const long long env_cap = [] { const char *v = getenv("MCP_FUSION_GRID"); return (long long)(v && *v ? atoi(v) : 0); }();
    if (env_cap >= 1) grid_cap = env_cap;
    static const bool f32_mfma = [] { const char *v = getenv("MCP_FUSION_F32_MFMA"); return v && *v == '1'; }();
#endif
    const unsigned grid = (unsigned)min((total + WAVES - 1) / WAVES, grid_cap);
    mcp_prof_begin(MCP_KERNEL_FUSION, s);
#ifdef MCP_AB
    if (f32_mfma) {
        hipLaunchKernelGGL(fusion_kernel, dim3(grid), dim3(64 * WAVES), 0, s, total, n, p1, p2, idx, idx2, w1, b1, w2, b2, w3, b3, out);
        mcp_prof_end(MCP_KERNEL_FUSION, s);
        return mcp_launch_status();
    }
#endif
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        { const hipError_t attr_e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(fusion_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (attr_e_ != hipSuccess) return (int)attr_e_; }
        attr_once.done();
    }
    hipLaunchKernelGGL(fusion_split_kernel, dim3(grid), dim3(64 * WAVES), SP_LDS_BYTES, s, total, n, p1, p2, idx, idx2, w1, b1, w2, b2, w3, b3, out);
    mcp_prof_end(MCP_KERNEL_FUSION, s);
    return mcp_launch_status();
}
