// fusion_bn.hip -- the attentive fusion layer (MultiFrameEstimatier.knn_group + fusion, mocopci.py:798-819) in net.train() mode: its
// three Conv2d + BatchNorm2d(eps 1e-3) + ReLU layers normalise with BATCH statistics (train.py:130 calls net.train()), so nothing can
// be folded into the weights before the statistics of a layer's input are known, and the backward has the BatchNorm terms
//     dz = (gamma / sigma) (dy' - mean(dy') - zhat mean(dy' zhat)),   dy' = dy [y > 0],
// whose means run over ALL rows (every neighbour of every point of the call).  The reference (and this repo's unfused path,
// model.py:fusion_batch_stats) materialises the (rows, 64 | 64 | 128) activations -- 1 / 1 / 2 GiB per tensor per call at B = 8,
// N = 8192.  Here every pass re-evaluates what it needs in the fused forward's MFMA layout (one wave per point, neighbours on the MFMA
// column), and a pass ends where the next quantity needs a reduction over all rows:
//   forward    S1, S2, S3: statistics of z1, z2, z3 (sums of W h and (W h)^2: the conv bias is the shift that keeps the variance's
//              cancellation small); F: the layer with all three (mean, 1/sigma) known -> out.
//   backward   B1: the forward with arg-max channel, softmax and ds per neighbour -> per-row (c*, dy3', a_j) and sum dy3', sum dy3' zhat3;
//              B2: dz3 (dense: the mean terms reach every channel), dW3, dh2 -> dy2' (stored), sum dy2', sum dy2' zhat2;
//              B3: dz2, dW2, dh1 -> dy1' (stored), its two sums;   B4: dz1, dW1, dx0 -> d_nb, d_p1.
// Every per-channel sum is per-lane over a wave's points, then a fixed butterfly, waves in wave order, workgroups in workgroup
// order: results repeat bit for bit.  All kernels of one call (one set of statistics) see only that call's clouds.
#include "common.h"
#include "mfma_split.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int C1 = 64, C2 = 64, C3 = 128, NB = 64;
constexpr int WAVES = 4;
constexpr int TS = 36;  // row stride (floats) of the transposition tile (backward passes)
// bn vector (floats, natural channel order): per layer mean | rstd | gamma | beta
constexpr int BN_L1 = 0, BN_L2 = 4 * C1, BN_L3 = 4 * C1 + 4 * C2, BN_FLOATS = BN_L3 + 4 * C3;  // 1024

// LDS, floats: W1 MFMA image | conv biases (accumulator order) | per layer rs | nm = -mean rs | gamma | beta (accumulator order)
constexpr int L_W1 = 0, L_B1 = 256, L_B2 = L_B1 + 64, L_B3 = L_B2 + 64, L_BN1 = L_B3 + 128, L_BN2 = L_BN1 + 4 * C1, L_BN3 = L_BN2 + 4 * C2,
              L_F32 = L_BN3 + 4 * C3;  // 1536
constexpr int W2_U4 = 2 * 4 * 3 * 64, W3_U4 = 4 * 4 * 3 * 64;

__device__ __forceinline__ int chan_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ int acc_to_channel(int e) { return 32 * (e >> 5) + chan_of(e & 15, (e >> 4) & 1); }  // [t][h][r] -> channel

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
__device__ __forceinline__ float half_sum(float v) {  // over the 32 lanes that share lane >> 5
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// stages the fp32 part of the LDS image: W1, conv biases and the BatchNorm vectors of layers 1 .. layers
__device__ __forceinline__ void stage_f32(float *lds, const float *__restrict__ w1, const float *__restrict__ b1, const float *__restrict__ b2,
                                          const float *__restrict__ b3, const float *__restrict__ bn, int layers, int tid, int threads) {
    for (int e = tid; e < 256; e += threads) {  // w1 (fp32, K = 4): [t][s][lane] = W1[32t + (lane&31)][2s + (lane>>5)]
        const int l = e & 63, s = (e >> 6) & 1, t = e >> 7;
        lds[L_W1 + e] = w1[(32 * t + (l & 31)) * 4 + 2 * s + (l >> 5)];
    }
    for (int e = tid; e < 128; e += threads) {
        const int c = acc_to_channel(e);
        if (e < 64) {
            lds[L_B1 + e] = b1[c];
            lds[L_B2 + e] = b2[c];
        }
        lds[L_B3 + e] = b3[c];
    }
    for (int e = tid; e < 128; e += threads) {
        const int c = acc_to_channel(e);
        if (e < 64 && layers >= 1) {
            const float rs = bn[BN_L1 + C1 + c];
            lds[L_BN1 + e] = rs; lds[L_BN1 + C1 + e] = -bn[BN_L1 + c] * rs; lds[L_BN1 + 2 * C1 + e] = bn[BN_L1 + 2 * C1 + c]; lds[L_BN1 + 3 * C1 + e] = bn[BN_L1 + 3 * C1 + c];
        }
        if (e < 64 && layers >= 2) {
            const float rs = bn[BN_L2 + C2 + c];
            lds[L_BN2 + e] = rs; lds[L_BN2 + C2 + e] = -bn[BN_L2 + c] * rs; lds[L_BN2 + 2 * C2 + e] = bn[BN_L2 + 2 * C2 + c]; lds[L_BN2 + 3 * C2 + e] = bn[BN_L2 + 3 * C2 + c];
        }
        if (layers >= 3) {
            const float rs = bn[BN_L3 + C3 + c];
            lds[L_BN3 + e] = rs; lds[L_BN3 + C3 + e] = -bn[BN_L3 + c] * rs; lds[L_BN3 + 2 * C3 + e] = bn[BN_L3 + 2 * C3 + c]; lds[L_BN3 + 3 * C3 + e] = bn[BN_L3 + 3 * C3 + c];
        }
    }
}

// zhat = (z - mean) rstd, v = gamma zhat + beta for one accumulator tile; `at` = the layer's LDS block + (t * 2 + h) * 16, C its width
template <int C>
__device__ __forceinline__ void bn_tile(const float *at, const f32x16 &z, f32x16 &zhat, f32x16 &v) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        zhat[r] = __builtin_fmaf(z[r], at[r], at[C + r]);
        v[r] = __builtin_fmaf(at[2 * C + r], zhat[r], at[3 * C + r]);
    }
}

// layer 1 pre-activation tile t: W1 [r, |r|] (+ conv bias unless RAW)
template <bool RAW>
__device__ __forceinline__ f32x16 layer1_tile(const float *lds, int t, int h, int lane, float in0, float in1) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = RAW ? 0.f : lds[L_B1 + (t * 2 + h) * 16 + r];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[L_W1 + (t * 2 + 0) * 64 + lane], in0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[L_W1 + (t * 2 + 1) * 64 + lane], in1, acc, 0, 0, 0);
    return acc;
}
template <bool RAW>
__device__ __forceinline__ f32x16 bias_tile(const float *lds, int off) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = RAW ? 0.f : lds[off + r];
    return acc;
}

// MODE 1 / 2 / 3: sums of W h and (W h)^2 of layer MODE over all rows (partial[blockIdx][2 C]); MODE 0: the layer -> out (B,N,3)
template <int MODE>
__global__ __launch_bounds__(64 * WAVES, 2) void fusion_bn_fwd_kernel(long long total, int n, const float *__restrict__ p1, const float *__restrict__ p2,
                                                                   const int *__restrict__ idx, const int *__restrict__ idx2,
                                                                   const float *__restrict__ w1, const float *__restrict__ b1,
                                                                   const float *__restrict__ w2, const float *__restrict__ b2,
                                                                   const float *__restrict__ w3, const float *__restrict__ b3,
                                                                   const float *__restrict__ bn, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    uint4 *w2s = reinterpret_cast<uint4 *>(lds + L_F32);
    uint4 *w3s = w2s + W2_U4;
    const int tid = threadIdx.x;
    stage_f32(lds, w1, b1, b2, b3, bn, MODE == 0 ? 3 : MODE - 1, tid, 64 * WAVES);
    if (MODE != 1) mcp_split_weights(w2s, w2, C1, 2, tid, 64 * WAVES);
    if (MODE == 0 || MODE == 3) mcp_split_weights(w3s, w3, C2, 4, tid, 64 * WAVES);
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    constexpr int ST = MODE == 3 ? 4 : (MODE == 0 ? 1 : 2);  // statistics tiles
    f32x16 s1[ST], s2[ST];
#pragma unroll
    for (int t = 0; t < ST; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { s1[t][r] = 0.f; s2[t][r] = 0.f; }

    for (long long p = (long long)blockIdx.x * WAVES + wave; p < total; p += (long long)gridDim.x * WAVES) {
        const long long bb = mcp_div(p, n, mcp_fits32(total));
        const float cx = p1[p * 3 + 0], cy = p1[p * 3 + 1], cz = p1[p * 3 + 2];
        float score[2], nbx[2], nby[2], nbz[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int id = idx2 ? (ct ? idx2 : idx)[p * 32 + col] : idx[p * NB + 32 * ct + col];
            const float *q = p2 + ((long long)bb * n + id) * 3;
            const float x = q[0], y = q[1], z = q[2];
            nbx[ct] = x; nby[ct] = y; nbz[ct] = z;
            const float rx = x - cx, ry = y - cy, rz = z - cz;
            const float dist = sqrtf((rx * rx + ry * ry) + rz * rz);
            const float in0 = h ? ry : rx, in1 = h ? dist : rz;
            McpSplit3 x1[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x16 z1 = layer1_tile<MODE == 1>(lds, t, h, lane, in0, in1);
                if (MODE == 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { s1[MODE == 1 ? t : 0][r] += z1[r]; s2[MODE == 1 ? t : 0][r] = __builtin_fmaf(z1[r], z1[r], s2[MODE == 1 ? t : 0][r]); }
                } else {
                    f32x16 zh, v;
                    bn_tile<C1>(lds + L_BN1 + (t * 2 + h) * 16, z1, zh, v);
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
                    x1[2 * t + 0] = mcp_split_kstep(v, 0);
                    x1[2 * t + 1] = mcp_split_kstep(v, 1);
                }
            }
            if (MODE == 1) continue;
            McpSplit3 x2[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16 z2 = bias_tile<MODE == 2>(lds, L_B2 + (t * 2 + h) * 16);
                z2 = mcp_tile_split<4>(w2s + (size_t)t * 4 * 3 * 64 + lane, x1, z2);
                if (MODE == 2) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { s1[MODE == 2 ? t : 0][r] += z2[r]; s2[MODE == 2 ? t : 0][r] = __builtin_fmaf(z2[r], z2[r], s2[MODE == 2 ? t : 0][r]); }
                } else {
                    f32x16 zh, v;
                    bn_tile<C2>(lds + L_BN2 + (t * 2 + h) * 16, z2, zh, v);
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
                    x2[2 * t + 0] = mcp_split_kstep(v, 0);
                    x2[2 * t + 1] = mcp_split_kstep(v, 1);
                }
            }
            if (MODE == 2) continue;
            float m = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x16 z3 = bias_tile<MODE == 3>(lds, L_B3 + (t * 2 + h) * 16);
                z3 = mcp_tile_split<4>(w3s + (size_t)t * 4 * 3 * 64 + lane, x2, z3);
                if (MODE == 3) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { s1[MODE == 3 ? t : 0][r] += z3[r]; s2[MODE == 3 ? t : 0][r] = __builtin_fmaf(z3[r], z3[r], s2[MODE == 3 ? t : 0][r]); }
                } else {
                    f32x16 zh, v;
                    bn_tile<C3>(lds + L_BN3 + (t * 2 + h) * 16, z3, zh, v);
#pragma unroll
                    for (int r = 0; r < 16; ++r) m = fmaxf(m, v[r]);
                }
            }
            score[ct] = fmaxf(m, __shfl_xor(m, 32));
        }
        if (MODE == 0) {
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
    if (MODE != 0) {
        // per channel: butterfly over the neighbours' lanes, waves in wave order, natural channel order out
        constexpr int C = MODE == 3 ? C3 : C1;
        __syncthreads();
        float *red = lds;  // [WAVES][2 C]
#pragma unroll
        for (int t = 0; t < ST; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float a = half_sum(s1[t][r]), b = half_sum(s2[t][r]);
                if (col == 0) {
                    red[wave * 2 * C + (t * 2 + h) * 16 + r] = a;
                    red[wave * 2 * C + C + (t * 2 + h) * 16 + r] = b;
                }
            }
        __syncthreads();
        if (tid < 2 * C) {
            float v = red[tid];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) v += red[w * 2 * C + tid];
            const int which = tid >= C, e = tid - which * C;
            out[(size_t)blockIdx.x * 2 * C + which * C + acc_to_channel(e)] = v;
        }
    }
}

// mean = b + S1 / R, var = S2 / R - (S1 / R)^2 (biased), rstd = 1 / sqrt(var + eps): written into the bn vector; var also to var_out
__global__ __launch_bounds__(128) void fusion_bn_stats_kernel(const float *__restrict__ partial, int parts, int c, double rows, const float *__restrict__ bias,
                                                             float eps, float *__restrict__ bn_layer, float *__restrict__ var_out) {
    const int e = threadIdx.x;
    if (e >= c) return;
    float a = 0.f, b = 0.f;
    for (int g = 0; g < parts; ++g) {
        a += partial[(size_t)g * 2 * c + e];
        b += partial[(size_t)g * 2 * c + c + e];
    }
    const double m = (double)a / rows;
    double var = (double)b / rows - m * m;
    if (var < 0.0) var = 0.0;
    bn_layer[e] = (float)((double)bias[e] + m);
    bn_layer[c + e] = (float)(1.0 / sqrt(var + (double)eps));
    var_out[e] = (float)var;
}

unsigned fwd_grid(long long total) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const long long want = (total + WAVES - 1) / WAVES, cap = 2LL * cus;  // two resident workgroups per CU
    return (unsigned)(want < cap ? want : cap);
}

template <int MODE>
int launch_fwd(long long total, int n, const float *p1, const float *p2, const int *idx, const int *idx2, const float *w1, const float *b1, const float *w2,
               const float *b2, const float *w3, const float *b3, const float *bn, float *out, unsigned grid, hipStream_t s) {
    auto kern = fusion_bn_fwd_kernel<MODE>;
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_once.done();
    }
    const size_t lds = (size_t)L_F32 * 4 + (size_t)(W2_U4 + W3_U4) * 16;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), lds, s, total, n, p1, p2, idx, idx2, w1, b1, w2, b2, w3, b3, bn, out);
    return mcp_launch_status();
}

}  // namespace

MCP_EXPORT int mcp_fusion_bn_floats(void) { return BN_FLOATS; }

MCP_EXPORT size_t mcp_fusion_bn_workspace_bytes(int b, int n) {
    if (b <= 0 || n <= 0) return 0;
    return (size_t)fwd_grid((long long)b * n) * 2 * C3 * sizeof(float);
}

// The layer on BATCH statistics, forward.  bn (1024 floats): per layer mean | rstd | gamma | beta, natural channel order, layers
// 4 -> 64 -> 64 -> 128; the caller fills gamma and beta, this call fills mean and rstd (from the b clouds it is given: one
// reference call = one set of statistics) and writes the biased variances to var (64 | 64 | 128 floats) for the running estimates.
MCP_EXPORT int mcp_fusion_bn_forward(int b, int n, int nb, const float *p1, const float *p2, const int *idx, const int *idx2, const float *w1,
                                     const float *b1, const float *w2, const float *b2, const float *w3, const float *b3, float eps, float *bn, float *var,
                                     float *out, void *workspace, size_t workspace_bytes, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && p1 && p2 && idx && w1 && b1 && w2 && b2 && w3 && b3 && bn && var && out && workspace);
    if (nb != NB) return MCP_ERR_UNSUPPORTED;
    const long long total = (long long)b * n;
    const unsigned grid = fwd_grid(total);
    if (workspace_bytes < (size_t)grid * 2 * C3 * sizeof(float)) return MCP_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    float *partial = static_cast<float *>(workspace);
    const double rows = (double)total * NB;
    mcp_prof_begin(MCP_KERNEL_FUSION, s);
    int rc = launch_fwd<1>(total, n, p1, p2, idx, idx2, w1, b1, w2, b2, w3, b3, bn, partial, grid, s);
    if (rc) return rc;
    hipLaunchKernelGGL(fusion_bn_stats_kernel, dim3(1), dim3(128), 0, s, partial, (int)grid, C1, rows, b1, eps, bn + BN_L1, var);
    rc = launch_fwd<2>(total, n, p1, p2, idx, idx2, w1, b1, w2, b2, w3, b3, bn, partial, grid, s);
    if (rc) return rc;
    hipLaunchKernelGGL(fusion_bn_stats_kernel, dim3(1), dim3(128), 0, s, partial, (int)grid, C2, rows, b2, eps, bn + BN_L2, var + C1);
    rc = launch_fwd<3>(total, n, p1, p2, idx, idx2, w1, b1, w2, b2, w3, b3, bn, partial, grid, s);
    if (rc) return rc;
    hipLaunchKernelGGL(fusion_bn_stats_kernel, dim3(1), dim3(128), 0, s, partial, (int)grid, C3, rows, b3, eps, bn + BN_L3, var + C1 + C2);
    rc = launch_fwd<0>(total, n, p1, p2, idx, idx2, w1, b1, w2, b2, w3, b3, bn, out, grid, s);
    mcp_prof_end(MCP_KERNEL_FUSION, s);
    return rc;
}
