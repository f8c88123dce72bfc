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
#include "mfma_grad.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int C1 = 64, C2 = 64, C3 = 128, NB = 64;
constexpr int WAVES = 4;
constexpr int TS = MCP_TS;
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

// MODE 1 / 2 / 3: sums of W h and (W h)^2 of layer MODE over all rows (partial[blockIdx][2 C]); MODE 0: the layer -> out (B,N,3).
// SAVE (MODE 0): also every neighbour's score (the maximum over the channels of the layer-3 output, floored at 0), the channel it sits at
// and zhat3 there -- what the backward's first pass (fusion_bn_b1_kernel) re-evaluates the whole layer for; with them it is a light
// kernel (fusion_bn_b1_saved_kernel).  The maximum is tracked exactly as that pass does (strict >, lowest channel among equals).
template <int MODE, bool SAVE = false>
__global__ __launch_bounds__(64 * WAVES, 2) void fusion_bn_fwd_kernel(long long total, int n, const float *__restrict__ p1, const float *__restrict__ p2,
                                                                   const int *__restrict__ idx, const int *__restrict__ idx2,
                                                                   const float *__restrict__ w1, const float *__restrict__ b1,
                                                                   const float *__restrict__ w2, const float *__restrict__ b2,
                                                                   const float *__restrict__ w3, const float *__restrict__ b3,
                                                                   const float *__restrict__ bn, float *__restrict__ out, int *__restrict__ save_c,
                                                                   float *__restrict__ save_z, float *__restrict__ save_s) {
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

    const McpUnits units = mcp_units_by_xcd(total, WAVES);   // XCD x takes the x-th eighth of the points (common.h)
    for (long long p = units.first + wave; p < units.limit; p += units.stride) {
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
            float m = 0.f, zb = 0.f;
            int mr = 0;
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
                    if (SAVE) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const bool up = v[r] > m;
                            m = up ? v[r] : m;
                            zb = up ? zh[r] : zb;
                            mr = up ? 16 * t + r : mr;
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) m = fmaxf(m, v[r]);
                    }
                }
            }
            if (SAVE) {
                const int mc = 32 * (mr >> 4) + chan_of(mr & 15, h);
                const float om = __shfl_xor(m, 32), oz = __shfl_xor(zb, 32);
                const int oc = __shfl_xor(mc, 32);
                const bool other = om > m || (om == m && oc < mc);
                score[ct] = other ? om : m;
                if (h == 0) {
                    const long long row = p * NB + 32 * ct + col;
                    save_c[row] = other ? oc : mc;
                    save_z[row] = other ? oz : zb;
                    save_s[row] = score[ct];
                }
            } else {
                score[ct] = fmaxf(m, __shfl_xor(m, 32));
            }
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

// mean = b + S1 / R, var = S2 / R - (S1 / R)^2 (biased), rstd = 1 / sqrt(var + eps): written into the bn vector; var also to var_out.
// S1, S2 are sums of (z - b) and (z - b)^2 (the conv bias is the shift).  The workgroup partials (fp32, each lane's own short chain)
// are added HERE in double, in fixed order, and the subtraction is done in double: the error of var is then the fp32 rounding of
// the partials, ~1e-7 (mean - b)^2 -- a channel whose conv output sits 300 standard deviations away from its bias still gets its
// variance to 1 %; beyond ~3000 the estimate degrades (torch's two-pass var does not).  (the partial sums used to be
// added in fp32 as well, which lost another factor of the number of workgroups.)
__global__ __launch_bounds__(256) void fusion_bn_stats_kernel(const float *__restrict__ partial, int parts, int c, double rows, const float *__restrict__ bias,
                                                             float eps, float *__restrict__ bn_layer, float *__restrict__ var_out) {
    // 16 channels per block, 16 chains per channel (workgroups ch, ch + 16, ... in order), chains added in chain order
    __shared__ double ca[16][17], cb[16][17];
    const int el = threadIdx.x & 15, ch = threadIdx.x >> 4, e = blockIdx.x * 16 + el;
    double sa = 0.0, sb = 0.0;
    if (e < c)
        for (int g = ch; g < parts; g += 16) {
            sa += (double)partial[(size_t)g * 2 * c + e];
            sb += (double)partial[(size_t)g * 2 * c + c + e];
        }
    ca[ch][el] = sa;
    cb[ch][el] = sb;
    __syncthreads();
    if (ch != 0 || e >= c) return;
    double a = ca[0][el], b = cb[0][el];
#pragma unroll
    for (int i = 1; i < 16; ++i) {
        a += ca[i][el];
        b += cb[i][el];
    }
    const double m = a / rows;
    double var = b / rows - m * m;
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

template <int MODE, bool SAVE = false>
int launch_fwd(long long total, int n, const float *p1, const float *p2, const int *idx, const int *idx2, const float *w1, const float *b1, const float *w2,
               const float *b2, const float *w3, const float *b3, const float *bn, float *out, unsigned grid, hipStream_t s, int *save_c = nullptr,
               float *save_z = nullptr, float *save_s = nullptr) {
    auto kern = fusion_bn_fwd_kernel<MODE, SAVE>;
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_once.done();
    }
    const size_t lds = (size_t)L_F32 * 4 + (size_t)(W2_U4 + W3_U4) * 16;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), lds, s, total, n, p1, p2, idx, idx2, w1, b1, w2, b2, w3, b3, bn, out, save_c, save_z, save_s);
    return mcp_launch_status();
}

__global__ __launch_bounds__(256) void transposed_image_kernel(uint4 *dst, const float *__restrict__ w, int m_total, int k_total) {
    mcp_split_weights_transposed(dst, w, m_total, k_total, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}
// per-channel constants of dz = a dy' - c1 - zhat c2 in accumulator order: a = gamma rstd, c1 = a sum(dy') / R, c2 = a sum(dy' zhat) / R
__device__ __forceinline__ void stage_dz_consts(float *dst, const float *__restrict__ bn_layer, const float *__restrict__ sums, int c, float inv_rows, int tid,
                                                int threads) {
    for (int e = tid; e < c; e += threads) {
        const int ch = acc_to_channel(e);
        const float a = bn_layer[2 * c + ch] * bn_layer[c + ch];
        dst[e] = a;
        dst[c + e] = a * (sums[ch] * inv_rows);
        dst[2 * c + e] = a * (sums[c + ch] * inv_rows);
    }
}
template <int C>
__device__ __forceinline__ void dz_tile(const float *at, const f32x16 &dy, const f32x16 &zhat, f32x16 &dz) {
#pragma unroll
    for (int r = 0; r < 16; ++r) dz[r] = __builtin_fmaf(-zhat[r], at[2 * C + r], __builtin_fmaf(at[r], dy[r], -at[C + r]));
}
// rows of a (rows, 64) fp32 array in accumulator layout: channels 32 t + 8 q + 4 h + (0..3) = registers 4 q .. 4 q + 3 of tile t
__device__ __forceinline__ void load_row64(const float *row, int h, f32x16 *v) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a = reinterpret_cast<const float4 *>(row)[8 * t + 2 * q + h];
            v[t][4 * q + 0] = a.x; v[t][4 * q + 1] = a.y; v[t][4 * q + 2] = a.z; v[t][4 * q + 3] = a.w;
        }
}
__device__ __forceinline__ void store_row64(float *row, int h, const f32x16 *v) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            reinterpret_cast<float4 *>(row)[8 * t + 2 * q + h] = make_float4(v[t][4 * q + 0], v[t][4 * q + 1], v[t][4 * q + 2], v[t][4 * q + 3]);
}
// sum over the neighbours of dy' and dy' zhat with lane = channel (both tensors through the transposition tile): 4 registers of
// accumulators instead of 64.  sums[mt] / sumz[mt]: channel 32 mt + col, the two lane halves hold the two 8-neighbour groups.
__device__ __forceinline__ void channel_sums(float *tb, const f32x16 *dy, const f32x16 *zhat, int col, int h, float *sums, float *sumz) {
    float dyv[2][2][8];
    __builtin_amdgcn_wave_barrier();
    mcp_write_tile(tb, dy[0], col, h);
    mcp_write_tile(tb + 32 * TS, dy[1], col, h);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            mcp_read8(tb + (32 * mt + col) * TS + 16 * ks + 8 * h, dyv[mt][ks]);
            sums[mt] += mcp_sum8(dyv[mt][ks]);
        }
    __builtin_amdgcn_wave_barrier();
    mcp_write_tile(tb, zhat[0], col, h);
    mcp_write_tile(tb + 32 * TS, zhat[1], col, h);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float zv[8];
            mcp_read8(tb + (32 * mt + col) * TS + 16 * ks + 8 * h, zv);
            float a = dyv[mt][ks][0] * zv[0];
#pragma unroll
            for (int i = 1; i < 8; ++i) a = __builtin_fmaf(dyv[mt][ks][i], zv[i], a);
            sumz[mt] += a;
        }
}

// ---- B1: the forward with arg-max channel, softmax, ds -> per-row (c*, dy3' at c*, softmax weight), sums of dy3' and dy3' zhat3 ----
constexpr int B1_SCR = 96;  // per wave: c* [32] | dy3' [32] | dy3' zhat3 [32]
__global__ __launch_bounds__(64 * WAVES, 2) void fusion_bn_b1_kernel(long long total, int n, const float *__restrict__ p1, const float *__restrict__ p2,
                                                                  const int *__restrict__ idx, const int *__restrict__ idx2,
                                                                  const float *__restrict__ w1, const float *__restrict__ b1,
                                                                  const float *__restrict__ w2, const float *__restrict__ b2,
                                                                  const float *__restrict__ w3, const float *__restrict__ b3,
                                                                  const float *__restrict__ bn, const float *__restrict__ gout, int *__restrict__ row_c,
                                                                  float *__restrict__ row_dy, float *__restrict__ row_a, float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    uint4 *w2s = reinterpret_cast<uint4 *>(lds + L_F32);
    uint4 *w3s = w2s + W2_U4;
    const int tid = threadIdx.x;
    stage_f32(lds, w1, b1, b2, b3, bn, 3, tid, 64 * WAVES);
    mcp_split_weights(w2s, w2, C1, 2, tid, 64 * WAVES);
    mcp_split_weights(w3s, w3, C2, 4, tid, 64 * WAVES);
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    float *scr = reinterpret_cast<float *>(w3s + W3_U4) + wave * B1_SCR;
    int *csb = reinterpret_cast<int *>(scr);
    float *dzb = scr + 32, *ezb = scr + 64;
    float sb[4] = {0.f, 0.f, 0.f, 0.f}, sg[4] = {0.f, 0.f, 0.f, 0.f};  // lane = channel 32 mt + col

    const McpUnits units = mcp_units_by_xcd(total, WAVES);   // XCD x takes the x-th eighth of the points (common.h)
    for (long long p = units.first + wave; p < units.limit; p += units.stride) {
        const long long bb = mcp_div(p, n, mcp_fits32(total));
        const float cx = p1[p * 3 + 0], cy = p1[p * 3 + 1], cz = p1[p * 3 + 2];
        const float gx = gout[p * 3 + 0], gy = gout[p * 3 + 1], gz = gout[p * 3 + 2];
        float score[2], zstar[2], nbx[2], nby[2], nbz[2];
        int cstar[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int id = idx2 ? (ct ? idx2 : idx)[p * 32 + col] : idx[p * NB + 32 * ct + col];
            const float *q = p2 + ((long long)bb * n + id) * 3;
            const float x = q[0], y = q[1], z = q[2];
            nbx[ct] = x; nby[ct] = y; nbz[ct] = z;
            const float rx = x - cx, ry = y - cy, rz = z - cz;
            const float dist = sqrtf((rx * rx + ry * ry) + rz * rz);
            const float in0 = h ? ry : rx, in1 = h ? dist : rz;
            McpSplit3 x1[4], x2[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x16 z1 = layer1_tile<false>(lds, t, h, lane, in0, in1);
                f32x16 zh, v;
                bn_tile<C1>(lds + L_BN1 + (t * 2 + h) * 16, z1, zh, v);
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
                x1[2 * t + 0] = mcp_split_kstep(v, 0);
                x1[2 * t + 1] = mcp_split_kstep(v, 1);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16 z2 = bias_tile<false>(lds, L_B2 + (t * 2 + h) * 16);
                z2 = mcp_tile_split<4>(w2s + (size_t)t * 4 * 3 * 64 + lane, x1, z2);
                f32x16 zh, v;
                bn_tile<C2>(lds + L_BN2 + (t * 2 + h) * 16, z2, zh, v);
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
                x2[2 * t + 0] = mcp_split_kstep(v, 0);
                x2[2 * t + 1] = mcp_split_kstep(v, 1);
            }
            float m = 0.f, zb = 0.f;
            int mr = 0;
#pragma unroll 1
            for (int t = 0; t < 4; ++t) {
                f32x16 z3 = bias_tile<false>(lds, L_B3 + (t * 2 + h) * 16);
                z3 = mcp_tile_split<4>(w3s + (size_t)t * 4 * 3 * 64 + lane, x2, z3);
                f32x16 zh, v;
                bn_tile<C3>(lds + L_BN3 + (t * 2 + h) * 16, z3, zh, v);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool up = v[r] > m;
                    m = up ? v[r] : m;
                    zb = up ? zh[r] : zb;
                    mr = up ? 16 * t + r : mr;
                }
            }
            const int mc = 32 * (mr >> 4) + chan_of(mr & 15, h);
            const float om = __shfl_xor(m, 32), oz = __shfl_xor(zb, 32);
            const int oc = __shfl_xor(mc, 32);
            const bool other = om > m || (om == m && oc < mc);
            score[ct] = other ? om : m;
            zstar[ct] = other ? oz : zb;
            cstar[ct] = other ? oc : mc;
        }
        const float mx = wave_max(fmaxf(score[0], score[1]));
        const float e0 = expf(score[0] - mx), e1 = expf(score[1] - mx);
        const float den = 0.5f * wave_sum(e0 + e1);  // every neighbour sits in both lane halves
        const float a0 = e0 / den, a1 = e1 / den;
        const float da0 = (gx * nbx[0] + gy * nby[0]) + gz * nbz[0], da1 = (gx * nbx[1] + gy * nby[1]) + gz * nbz[1];
        const float sdot = 0.5f * wave_sum(a0 * da0 + a1 * da1);
        const float dy[2] = {score[0] > 0.f ? a0 * (da0 - sdot) : 0.f, score[1] > 0.f ? a1 * (da1 - sdot) : 0.f};
        const float aw[2] = {a0, a1};
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            __builtin_amdgcn_wave_barrier();
            if (h == 0) {
                const long long row = p * NB + 32 * ct + col;
                row_c[row] = cstar[ct];
                row_dy[row] = dy[ct];
                row_a[row] = aw[ct];
                csb[col] = cstar[ct];
                dzb[col] = dy[ct];
                ezb[col] = dy[ct] * zstar[ct];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int j0 = 16 * ks + 8 * h;
                const int4 ca = reinterpret_cast<const int4 *>(csb + j0)[0], cb = reinterpret_cast<const int4 *>(csb + j0)[1];
                const int cs8[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
                float dz8[8], ez8[8];
                mcp_read8(dzb + j0, dz8);
                mcp_read8(ezb + j0, ez8);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int c = 32 * mt + col;
                    float s = 0.f, g = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        s += cs8[i] == c ? dz8[i] : 0.f;
                        g += cs8[i] == c ? ez8[i] : 0.f;
                    }
                    sb[mt] += s;
                    sg[mt] += g;
                }
            }
        }
    }
    __syncthreads();
    float *red = lds;  // [WAVES][256]: sum dy3' (128) | sum dy3' zhat3 (128), natural channel order
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const float a = sb[mt] + __shfl_xor(sb[mt], 32), b = sg[mt] + __shfl_xor(sg[mt], 32);
        if (h == 0) {
            red[wave * 256 + 32 * mt + col] = a;
            red[wave * 256 + 128 + 32 * mt + col] = b;
        }
    }
    __syncthreads();
    if (tid < 256) {
        float v = red[tid];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) v += red[w * 256 + tid];
        partial[(size_t)blockIdx.x * 256 + tid] = v;
    }
}

// ---- B1 with the forward's per-neighbour (score, c*, zhat3 at c*) saved (mcp_fusion_bn_forward_save): no layer to re-evaluate -- the
// softmax over the 64 scores, ds, the per-row outputs and the two per-channel sums, exactly as above from its `score / zstar / cstar` on ----
__global__ __launch_bounds__(64 * WAVES, 2) void fusion_bn_b1_saved_kernel(long long total, int n, const float *__restrict__ p2, const int *__restrict__ idx,
                                                                        const int *__restrict__ idx2, const float *__restrict__ gout,
                                                                        const int *__restrict__ save_c, const float *__restrict__ save_z,
                                                                        const float *__restrict__ save_s, int *__restrict__ row_c,
                                                                        float *__restrict__ row_dy, float *__restrict__ row_a, float *__restrict__ partial) {
    __shared__ float lds[WAVES * 256];   // the per-wave scratch first, the reduction buffer afterwards
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    float *scr = lds + wave * B1_SCR;
    int *csb = reinterpret_cast<int *>(scr);
    float *dzb = scr + 32, *ezb = scr + 64;
    float sb[4] = {0.f, 0.f, 0.f, 0.f}, sg[4] = {0.f, 0.f, 0.f, 0.f};  // lane = channel 32 mt + col

    const McpUnits units = mcp_units_by_xcd(total, WAVES);   // the same points per workgroup as fusion_bn_b1_kernel: the same summation order
    for (long long p = units.first + wave; p < units.limit; p += units.stride) {
        const long long bb = mcp_div(p, n, mcp_fits32(total));
        const float gx = gout[p * 3 + 0], gy = gout[p * 3 + 1], gz = gout[p * 3 + 2];
        float score[2], zstar[2], nbx[2], nby[2], nbz[2];
        int cstar[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int id = idx2 ? (ct ? idx2 : idx)[p * 32 + col] : idx[p * NB + 32 * ct + col];
            const float *q = p2 + ((long long)bb * n + id) * 3;
            nbx[ct] = q[0]; nby[ct] = q[1]; nbz[ct] = q[2];
            const long long row = p * NB + 32 * ct + col;
            score[ct] = save_s[row];
            zstar[ct] = save_z[row];
            cstar[ct] = save_c[row];
        }
        const float mx = wave_max(fmaxf(score[0], score[1]));
        const float e0 = expf(score[0] - mx), e1 = expf(score[1] - mx);
        const float den = 0.5f * wave_sum(e0 + e1);  // every neighbour sits in both lane halves
        const float a0 = e0 / den, a1 = e1 / den;
        const float da0 = (gx * nbx[0] + gy * nby[0]) + gz * nbz[0], da1 = (gx * nbx[1] + gy * nby[1]) + gz * nbz[1];
        const float sdot = 0.5f * wave_sum(a0 * da0 + a1 * da1);
        const float dy[2] = {score[0] > 0.f ? a0 * (da0 - sdot) : 0.f, score[1] > 0.f ? a1 * (da1 - sdot) : 0.f};
        const float aw[2] = {a0, a1};
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            __builtin_amdgcn_wave_barrier();
            if (h == 0) {
                const long long row = p * NB + 32 * ct + col;
                row_c[row] = cstar[ct];
                row_dy[row] = dy[ct];
                row_a[row] = aw[ct];
                csb[col] = cstar[ct];
                dzb[col] = dy[ct];
                ezb[col] = dy[ct] * zstar[ct];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int j0 = 16 * ks + 8 * h;
                const int4 ca = reinterpret_cast<const int4 *>(csb + j0)[0], cb = reinterpret_cast<const int4 *>(csb + j0)[1];
                const int cs8[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
                float dz8[8], ez8[8];
                mcp_read8(dzb + j0, dz8);
                mcp_read8(ezb + j0, ez8);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int c = 32 * mt + col;
                    float s = 0.f, g = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        s += cs8[i] == c ? dz8[i] : 0.f;
                        g += cs8[i] == c ? ez8[i] : 0.f;
                    }
                    sb[mt] += s;
                    sg[mt] += g;
                }
            }
        }
    }
    __syncthreads();
    float *red = lds;  // [WAVES][256]: sum dy3' (128) | sum dy3' zhat3 (128), natural channel order
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const float a = sb[mt] + __shfl_xor(sb[mt], 32), b = sg[mt] + __shfl_xor(sg[mt], 32);
        if (h == 0) {
            red[wave * 256 + 32 * mt + col] = a;
            red[wave * 256 + 128 + 32 * mt + col] = b;
        }
    }
    __syncthreads();
    if (tid < 256) {
        float v = red[tid];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) v += red[w * 256 + tid];
        partial[(size_t)blockIdx.x * 256 + tid] = v;
    }
}

// ---- B2: dz3 (dense), dW3 += dz3 . y2^T, dh2 = W3^T dz3 -> dy2' (stored), sums of dy2' and dy2' zhat2 ----
constexpr int B2_L_DZ = L_F32, B2_F32 = L_F32 + 3 * C3;                  // + a | c1 | c2 of layer 3 (accumulator order)
constexpr int B2_SCR = 64 * TS + 32 * TS;                                // per wave: tile [64][TS] (y2^T, then the sums) | tile [32][TS] (a dz3 tile)
constexpr int B2_G = C3 * C2 + 2 * C2;                                   // the workgroup's vector: dW3 (128,64) | sum dy2' | sum dy2' zhat2
constexpr size_t B2_LDS = (size_t)B2_F32 * 4 + (size_t)(W2_U4 + W3_U4) * 16 + (size_t)WAVES * B2_SCR * 4;
static_assert(B2_LDS <= 160 * 1024, "LDS budget");
__global__ __launch_bounds__(64 * WAVES, 1) void fusion_bn_b2_kernel(long long total, int n, const float *__restrict__ p1, const float *__restrict__ p2,
                                                                  const int *__restrict__ idx, const int *__restrict__ idx2,
                                                                  const float *__restrict__ w1, const float *__restrict__ b1,
                                                                  const float *__restrict__ w2, const float *__restrict__ b2,
                                                                  const float *__restrict__ w3, const float *__restrict__ b3,
                                                                  const float *__restrict__ bn, const float *__restrict__ sums3, float inv_rows,
                                                                  const uint4 *__restrict__ w3t, const int *__restrict__ row_c,
                                                                  const float *__restrict__ row_dy, float *__restrict__ dy2, float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    uint4 *w2s = reinterpret_cast<uint4 *>(lds + B2_F32);
    uint4 *w3s = w2s + W2_U4;
    const int tid = threadIdx.x;
    stage_f32(lds, w1, b1, b2, b3, bn, 3, tid, 64 * WAVES);
    stage_dz_consts(lds + B2_L_DZ, bn + BN_L3, sums3, C3, inv_rows, tid, 64 * WAVES);
    mcp_split_weights(w2s, w2, C1, 2, tid, 64 * WAVES);
    mcp_split_weights(w3s, w3, C2, 4, tid, 64 * WAVES);
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    float *tb = reinterpret_cast<float *>(w3s + W3_U4) + wave * B2_SCR;
    float *tz = tb + 64 * TS;

    f32x16 dW3a[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) dW3a[a][b][r] = 0.f;
    float s2[2] = {0.f, 0.f}, g2[2] = {0.f, 0.f};

    const McpUnits units = mcp_units_by_xcd(total, WAVES);   // XCD x takes the x-th eighth of the points (common.h)
    for (long long p = units.first + wave; p < units.limit; p += units.stride) {
        const long long bb = mcp_div(p, n, mcp_fits32(total));
        const float cx = p1[p * 3 + 0], cy = p1[p * 3 + 1], cz = p1[p * 3 + 2];
#pragma unroll 1
        for (int ct = 0; ct < 2; ++ct) {
            const int id = idx2 ? (ct ? idx2 : idx)[p * 32 + col] : idx[p * NB + 32 * ct + col];
            const long long row = p * NB + 32 * ct + col;
            const int cst = row_c[row];
            const float dys = row_dy[row];
            const float *q = p2 + ((long long)bb * n + id) * 3;
            const float rx = q[0] - cx, ry = q[1] - cy, rz = q[2] - cz;
            const float dist = sqrtf((rx * rx + ry * ry) + rz * rz);
            const float in0 = h ? ry : rx, in1 = h ? dist : rz;
            uint32_t live2 = 0u;  // bit 16 t + r: y2 > 0 in register r of tile t
            McpSplit3 x2[4], bs[2][2];
            {
                f32x16 zh2[2];
                McpSplit3 x1[4];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f32x16 z1 = layer1_tile<false>(lds, t, h, lane, in0, in1);
                    f32x16 zh, v;
                    bn_tile<C1>(lds + L_BN1 + (t * 2 + h) * 16, z1, zh, v);
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
                    x1[2 * t + 0] = mcp_split_kstep(v, 0);
                    x1[2 * t + 1] = mcp_split_kstep(v, 1);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x16 z2 = bias_tile<false>(lds, L_B2 + (t * 2 + h) * 16);
                    z2 = mcp_tile_split<4>(w2s + (size_t)t * 4 * 3 * 64 + lane, x1, z2);
                    f32x16 v;
                    bn_tile<C2>(lds + L_BN2 + (t * 2 + h) * 16, z2, zh2[t], v);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        live2 |= v[r] > 0.f ? 1u << (16 * t + r) : 0u;
                        v[r] = fmaxf(v[r], 0.f);
                    }
                    x2[2 * t + 0] = mcp_split_kstep(v, 0);
                    x2[2 * t + 1] = mcp_split_kstep(v, 1);
                    mcp_write_tile(tb + 32 * t * TS, v, col, h);  // y2^T: the B operand of dW3
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        float v[8];
                        mcp_read8(tb + (32 * nt + col) * TS + 16 * ks + 8 * h, v);
                        bs[nt][ks] = mcp_split8(v);
                    }
                __builtin_amdgcn_wave_barrier();
                mcp_write_tile(tb, zh2[0], col, h);  // zhat2^T stays in the tile until the sums at the end of the half
                mcp_write_tile(tb + 32 * TS, zh2[1], col, h);
            }
            f32x16 dh2[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) dh2[t][r] = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                __builtin_amdgcn_sched_barrier(0);  // a tile's loads are not hoisted above the previous tile (registers)
                f32x16 z3 = bias_tile<false>(lds, L_B3 + (t * 2 + h) * 16);
                z3 = mcp_tile_split<4>(w3s + (size_t)t * 4 * 3 * 64 + lane, x2, z3);
                f32x16 zh, v, dy, dz;
                bn_tile<C3>(lds + L_BN3 + (t * 2 + h) * 16, z3, zh, v);
#pragma unroll
                for (int r = 0; r < 16; ++r) dy[r] = (32 * t + chan_of(r, h)) == cst ? dys : 0.f;
                dz_tile<C3>(lds + B2_L_DZ + (t * 2 + h) * 16, dy, zh, dz);
                __builtin_amdgcn_wave_barrier();
                mcp_write_tile(tz, dz, col, h);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    float v8[8];
                    mcp_read8(tz + col * TS + 16 * ks + 8 * h, v8);
                    const McpSplit3 as = mcp_split8(v8);
                    dW3a[t][0] = mcp_mfma_split6(as, bs[0][ks], dW3a[t][0]);
                    dW3a[t][1] = mcp_mfma_split6(as, bs[1][ks], dW3a[t][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                const McpSplit3 z0 = mcp_split_kstep(dz, 0), z1s = mcp_split_kstep(dz, 1);
#pragma unroll
                for (int to = 0; to < 2; ++to) {
                    const uint4 *wk = w3t + ((size_t)to * 8 + 2 * t) * 3 * 64 + lane;
                    dh2[to] = mcp_mfma_split(wk, z0, dh2[to]);
                    dh2[to] = mcp_mfma_split(wk + 3 * 64, z1s, dh2[to]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // dy2' = dh2 [y2 > 0]; its sums over the neighbours with lane = channel (dy2' through the small tile, zhat2^T is in the large one)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) dh2[t][r] = (live2 >> (16 * t + r)) & 1u ? dh2[t][r] : 0.f;
            store_row64(dy2 + row * C2, h, dh2);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                __builtin_amdgcn_wave_barrier();
                mcp_write_tile(tz, dh2[mt], col, h);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    float dv[8], zv[8];
                    mcp_read8(tz + col * TS + 16 * ks + 8 * h, dv);
                    mcp_read8(tb + (32 * mt + col) * TS + 16 * ks + 8 * h, zv);
                    s2[mt] += mcp_sum8(dv);
                    float a = dv[0] * zv[0];
#pragma unroll
                    for (int i = 1; i < 8; ++i) a = __builtin_fmaf(dv[i], zv[i], a);
                    g2[mt] += a;
                }
            }
        }
    }
    // ---- the workgroup's vector: waves in wave order through LDS ----
    __syncthreads();
    float *red = lds;
    for (int e = tid; e < B2_G; e += 64 * WAVES) red[e] = 0.f;
    __syncthreads();
#pragma unroll 1
    for (int w = 0; w < WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[(32 * mt + chan_of(r, h)) * C2 + 32 * nt + col] += dW3a[mt][nt][r];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float a = s2[mt] + __shfl_xor(s2[mt], 32), b = g2[mt] + __shfl_xor(g2[mt], 32);
                if (h == 0) {
                    red[C3 * C2 + 32 * mt + col] += a;
                    red[C3 * C2 + C2 + 32 * mt + col] += b;
                }
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < B2_G; e += 64 * WAVES) partial[(size_t)blockIdx.x * B2_G + e] = red[e];
}

// ---- B3: dz2, dW2 += dz2 . y1^T, dh1 = W2^T dz2 -> dy1' (stored), sums of dy1' and dy1' zhat1 ----
constexpr int B3_L_DZ = L_F32, B3_F32 = L_F32 + 3 * C2;
constexpr int B3_SCR = 64 * TS;
constexpr int B3_G = C2 * C1 + 2 * C1;  // dW2 (64,64) | sum dy1' | sum dy1' zhat1
constexpr size_t B3_LDS = (size_t)B3_F32 * 4 + (size_t)(2 * W2_U4) * 16 + (size_t)WAVES * B3_SCR * 4;
__global__ __launch_bounds__(64 * WAVES, 1) void fusion_bn_b3_kernel(long long total, int n, const float *__restrict__ p1, const float *__restrict__ p2,
                                                                  const int *__restrict__ idx, const int *__restrict__ idx2,
                                                                  const float *__restrict__ w1, const float *__restrict__ b1,
                                                                  const float *__restrict__ w2, const float *__restrict__ b2,
                                                                  const float *__restrict__ b3, const float *__restrict__ bn,
                                                                  const float *__restrict__ sums2, float inv_rows, const float *__restrict__ dy2,
                                                                  float *__restrict__ dy1, float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    uint4 *w2s = reinterpret_cast<uint4 *>(lds + B3_F32);
    uint4 *w2ts = w2s + W2_U4;
    const int tid = threadIdx.x;
    stage_f32(lds, w1, b1, b2, b3, bn, 2, tid, 64 * WAVES);
    stage_dz_consts(lds + B3_L_DZ, bn + BN_L2, sums2, C2, inv_rows, tid, 64 * WAVES);
    mcp_split_weights(w2s, w2, C1, 2, tid, 64 * WAVES);
    mcp_split_weights_transposed(w2ts, w2, C1, C2, tid, 64 * WAVES);
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    float *tb = reinterpret_cast<float *>(w2ts + W2_U4) + wave * B3_SCR;

    f32x16 dW2a[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) dW2a[a][b][r] = 0.f;
    float s1[2] = {0.f, 0.f}, g1[2] = {0.f, 0.f};

    const McpUnits units = mcp_units_by_xcd(total, WAVES);   // XCD x takes the x-th eighth of the points (common.h)
    for (long long p = units.first + wave; p < units.limit; p += units.stride) {
        const long long bb = mcp_div(p, n, mcp_fits32(total));
        const float cx = p1[p * 3 + 0], cy = p1[p * 3 + 1], cz = p1[p * 3 + 2];
#pragma unroll 1
        for (int ct = 0; ct < 2; ++ct) {
            const int id = idx2 ? (ct ? idx2 : idx)[p * 32 + col] : idx[p * NB + 32 * ct + col];
            const long long row = p * NB + 32 * ct + col;
            const float *q = p2 + ((long long)bb * n + id) * 3;
            const float rx = q[0] - cx, ry = q[1] - cy, rz = q[2] - cz;
            const float dist = sqrtf((rx * rx + ry * ry) + rz * rz);
            const float in0 = h ? ry : rx, in1 = h ? dist : rz;
            f32x16 zh1[2], y1[2], dz2[2];
            {
                McpSplit3 x1[4];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f32x16 z1 = layer1_tile<false>(lds, t, h, lane, in0, in1);
                    f32x16 v;
                    bn_tile<C1>(lds + L_BN1 + (t * 2 + h) * 16, z1, zh1[t], v);
#pragma unroll
                    for (int r = 0; r < 16; ++r) y1[t][r] = fmaxf(v[r], 0.f);
                    x1[2 * t + 0] = mcp_split_kstep(y1[t], 0);
                    x1[2 * t + 1] = mcp_split_kstep(y1[t], 1);
                }
                f32x16 dy[2];
                load_row64(dy2 + row * C2, h, dy);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x16 z2 = bias_tile<false>(lds, L_B2 + (t * 2 + h) * 16);
                    z2 = mcp_tile_split<4>(w2s + (size_t)t * 4 * 3 * 64 + lane, x1, z2);
                    f32x16 zh, v;
                    bn_tile<C2>(lds + L_BN2 + (t * 2 + h) * 16, z2, zh, v);
                    dz_tile<C2>(lds + B3_L_DZ + (t * 2 + h) * 16, dy[t], zh, dz2[t]);
                }
            }
            // dW2 += dz2 . y1^T
            __builtin_amdgcn_wave_barrier();
            mcp_write_tile(tb, dz2[0], col, h);
            mcp_write_tile(tb + 32 * TS, dz2[1], col, h);
            __builtin_amdgcn_wave_barrier();
            McpSplit3 as[2][2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    float v[8];
                    mcp_read8(tb + (32 * mt + col) * TS + 16 * ks + 8 * h, v);
                    as[mt][ks] = mcp_split8(v);
                }
            __builtin_amdgcn_wave_barrier();
            mcp_write_tile(tb, y1[0], col, h);
            mcp_write_tile(tb + 32 * TS, y1[1], col, h);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    float v[8];
                    mcp_read8(tb + (32 * nt + col) * TS + 16 * ks + 8 * h, v);
                    const McpSplit3 bsp = mcp_split8(v);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) dW2a[mt][nt] = mcp_mfma_split6(as[mt][ks], bsp, dW2a[mt][nt]);
                }
            // dh1 = W2^T dz2; dy1' = dh1 [y1 > 0]
            f32x16 dh1[2];
            {
                McpSplit3 xs[4];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    xs[2 * t + 0] = mcp_split_kstep(dz2[t], 0);
                    xs[2 * t + 1] = mcp_split_kstep(dz2[t], 1);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                    acc = mcp_tile_split<4>(w2ts + (size_t)t * 4 * 3 * 64 + lane, xs, acc);
#pragma unroll
                    for (int r = 0; r < 16; ++r) dh1[t][r] = y1[t][r] > 0.f ? acc[r] : 0.f;
                }
            }
            store_row64(dy1 + row * C1, h, dh1);
            channel_sums(tb, dh1, zh1, col, h, s1, g1);
        }
    }
    __syncthreads();
    float *red = lds;
    for (int e = tid; e < B3_G; e += 64 * WAVES) red[e] = 0.f;
    __syncthreads();
#pragma unroll 1
    for (int w = 0; w < WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[(32 * mt + chan_of(r, h)) * C1 + 32 * nt + col] += dW2a[mt][nt][r];
                const float a = s1[mt] + __shfl_xor(s1[mt], 32), b = g1[mt] + __shfl_xor(g1[mt], 32);
                if (h == 0) {
                    red[C2 * C1 + 32 * mt + col] += a;
                    red[C2 * C1 + C1 + 32 * mt + col] += b;
                }
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < B3_G; e += 64 * WAVES) partial[(size_t)blockIdx.x * B3_G + e] = red[e];
}

// ---- B4: dz1, dW1 += dz1 . x0^T, dx0 = W1^T dz1 -> d_nb, d_p1 ----
constexpr int B4_L_DZ = L_F32, B4_L_W1R = L_F32 + 3 * C1, B4_F32 = B4_L_W1R + 256;
constexpr int B4_SCR = 64 * TS + 128;  // per wave: tile | x0 [32][4]
constexpr int B4_G = C1 * 4;           // dW1 (64,4)
constexpr size_t B4_LDS = (size_t)B4_F32 * 4 + (size_t)WAVES * B4_SCR * 4;
__global__ __launch_bounds__(64 * WAVES, 2) void fusion_bn_b4_kernel(long long total, int n, const float *__restrict__ p1, const float *__restrict__ p2,
                                                                  const int *__restrict__ idx, const int *__restrict__ idx2,
                                                                  const float *__restrict__ w1, const float *__restrict__ b1,
                                                                  const float *__restrict__ b2, const float *__restrict__ b3,
                                                                  const float *__restrict__ bn, const float *__restrict__ sums1, float inv_rows,
                                                                  const float *__restrict__ dy1, const float *__restrict__ row_a,
                                                                  const float *__restrict__ gout, float *__restrict__ d_p1, float *__restrict__ d_nb,
                                                                  float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    stage_f32(lds, w1, b1, b2, b3, bn, 1, tid, 64 * WAVES);
    stage_dz_consts(lds + B4_L_DZ, bn + BN_L1, sums1, C1, inv_rows, tid, 64 * WAVES);
    for (int e = tid; e < 256; e += 64 * WAVES) {  // W1 rows in accumulator order, for dx0 = W1^T dz1
        const int k = e & 3, r = (e >> 2) & 15, hh = (e >> 6) & 1, tt = e >> 7;
        lds[B4_L_W1R + e] = w1[(32 * tt + chan_of(r, hh)) * 4 + k];
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    float *tb = lds + B4_F32 + wave * B4_SCR;
    float4 *x0b = reinterpret_cast<float4 *>(tb + 64 * TS);
    float dW1a[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};

    const McpUnits units = mcp_units_by_xcd(total, WAVES);   // XCD x takes the x-th eighth of the points (common.h)
    for (long long p = units.first + wave; p < units.limit; p += units.stride) {
        const long long bb = mcp_div(p, n, mcp_fits32(total));
        const float cx = p1[p * 3 + 0], cy = p1[p * 3 + 1], cz = p1[p * 3 + 2];
        const float gx = gout[p * 3 + 0], gy = gout[p * 3 + 1], gz = gout[p * 3 + 2];
        float dcx = 0.f, dcy = 0.f, dcz = 0.f;
#pragma unroll 1
        for (int ct = 0; ct < 2; ++ct) {
            const int id = idx2 ? (ct ? idx2 : idx)[p * 32 + col] : idx[p * NB + 32 * ct + col];
            const long long row = p * NB + 32 * ct + col;
            const float aw = row_a[row];
            const float *q = p2 + ((long long)bb * n + id) * 3;
            const float rx = q[0] - cx, ry = q[1] - cy, rz = q[2] - cz;
            const float dist = sqrtf((rx * rx + ry * ry) + rz * rz);
            const float in0 = h ? ry : rx, in1 = h ? dist : rz;
            f32x16 dy[2], dz1[2];
            load_row64(dy1 + row * C1, h, dy);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x16 z1 = layer1_tile<false>(lds, t, h, lane, in0, in1);
                f32x16 zh, v;
                bn_tile<C1>(lds + L_BN1 + (t * 2 + h) * 16, z1, zh, v);
                dz_tile<C1>(lds + B4_L_DZ + (t * 2 + h) * 16, dy[t], zh, dz1[t]);
            }
            __builtin_amdgcn_wave_barrier();
            mcp_write_tile(tb, dz1[0], col, h);
            mcp_write_tile(tb + 32 * TS, dz1[1], col, h);
            if (h == 0) x0b[col] = make_float4(rx, ry, rz, dist);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    float v[8];
                    mcp_read8(tb + (32 * mt + col) * TS + 16 * ks + 8 * h, v);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float4 xj = x0b[16 * ks + 8 * h + i];
                        dW1a[mt][0] = __builtin_fmaf(v[i], xj.x, dW1a[mt][0]);
                        dW1a[mt][1] = __builtin_fmaf(v[i], xj.y, dW1a[mt][1]);
                        dW1a[mt][2] = __builtin_fmaf(v[i], xj.z, dW1a[mt][2]);
                        dW1a[mt][3] = __builtin_fmaf(v[i], xj.w, dW1a[mt][3]);
                    }
                }
            float dx0 = 0.f, dx1 = 0.f, dx2 = 0.f, dx3 = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float4 w = reinterpret_cast<const float4 *>(lds + B4_L_W1R)[(t * 2 + h) * 16 + r];
                    dx0 = __builtin_fmaf(w.x, dz1[t][r], dx0);
                    dx1 = __builtin_fmaf(w.y, dz1[t][r], dx1);
                    dx2 = __builtin_fmaf(w.z, dz1[t][r], dx2);
                    dx3 = __builtin_fmaf(w.w, dz1[t][r], dx3);
                }
            dx0 += __shfl_xor(dx0, 32);
            dx1 += __shfl_xor(dx1, 32);
            dx2 += __shfl_xor(dx2, 32);
            dx3 += __shfl_xor(dx3, 32);
            const float sc = dist > 0.f ? dx3 / dist : 0.f;
            const float drx = __builtin_fmaf(sc, rx, dx0), dry = __builtin_fmaf(sc, ry, dx1), drz = __builtin_fmaf(sc, rz, dx2);
            if (h == 0) {
                float *o = d_nb + row * 3;
                o[0] = __builtin_fmaf(aw, gx, drx);
                o[1] = __builtin_fmaf(aw, gy, dry);
                o[2] = __builtin_fmaf(aw, gz, drz);
            }
            dcx += drx; dcy += dry; dcz += drz;
            __builtin_amdgcn_wave_barrier();
        }
        const float sx = wave_sum(dcx), sy = wave_sum(dcy), sz = wave_sum(dcz);
        if (lane == 0) {
            d_p1[p * 3 + 0] = -0.5f * sx;
            d_p1[p * 3 + 1] = -0.5f * sy;
            d_p1[p * 3 + 2] = -0.5f * sz;
        }
    }
    __syncthreads();
    float *red = lds;  // [WAVES][256]
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float v = dW1a[mt][k] + __shfl_xor(dW1a[mt][k], 32);
            if (h == 0) red[wave * B4_G + (32 * mt + col) * 4 + k] = v;
        }
    __syncthreads();
    if (tid < B4_G) {
        float v = red[tid];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) v += red[w * B4_G + tid];
        partial[(size_t)blockIdx.x * B4_G + tid] = v;
    }
}

// out[e] = sum over the workgroups' partial vectors, in workgroup order
// 16 elements per block, 16 threads per element: thread c of an element adds the workgroups c, c + 16, ... in order, the 16 chains are
// then added in chain order -- a fixed order again, and 16 dependent loads deep instead of `parts`.
__global__ __launch_bounds__(256) void sum_partials_kernel(const float *__restrict__ partial, int parts, int stride, int count, float *__restrict__ out) {
    __shared__ float chains[16][17];
    const int el = threadIdx.x & 15, c = threadIdx.x >> 4, e = blockIdx.x * 16 + el;
    float s = 0.f;
    if (e < count)
        for (int g = c; g < parts; g += 16) s += partial[(size_t)g * stride + e];
    chains[c][el] = s;
    __syncthreads();
    if (c == 0 && e < count) {
        float t = chains[0][el];
#pragma unroll
        for (int i = 1; i < 16; ++i) t += chains[i][el];
        out[e] = t;
    }
}

unsigned bwd_grid(long long total, int per_cu) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const long long want = (total + WAVES - 1) / WAVES, cap = (long long)per_cu * cus;
    return (unsigned)(want < cap ? want : cap);
}
constexpr int W3T_U4 = 2 * 8 * 3 * 64;  // image of W3^T (M = 64, K = 128)
// workspace (floats): partial vectors (the largest pass: B2) | sums3 (256) | sums2 (128) | sums1 (128) | the W3^T image
size_t bwd_workspace_floats(long long total) {
    const size_t parts = (size_t)bwd_grid(total, 2);  // >= every pass's grid
    return parts * B2_G + 512 + (size_t)W3T_U4 * 4;
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
namespace {
int bn_forward(int b, int n, int nb, const float *p1, const float *p2, const int *idx, const int *idx2, const float *w1, const float *b1, const float *w2,
               const float *b2, const float *w3, const float *b3, float eps, float *bn, float *var, float *out, int *save_c, float *save_z, float *save_s,
               void *workspace, size_t workspace_bytes, mcp_stream_t stream);
}

MCP_EXPORT int mcp_fusion_bn_forward(int b, int n, int nb, const float *p1, const float *p2, const int *idx, const int *idx2, const float *w1,
                                     const float *b1, const float *w2, const float *b2, const float *w3, const float *b3, float eps, float *bn, float *var,
                                     float *out, void *workspace, size_t workspace_bytes, mcp_stream_t stream) {
    return bn_forward(b, n, nb, p1, p2, idx, idx2, w1, b1, w2, b2, w3, b3, eps, bn, var, out, nullptr, nullptr, nullptr, workspace, workspace_bytes, stream);
}

// The same forward that also keeps, per (point, neighbour) row (rows = b * n * 64), the neighbour's score, the layer-3 channel it comes
// from and zhat3 there: save_c (int32), save_z, save_s (floats), caller-owned -- mcp_fusion_bn_backward_saved then skips the
// re-evaluation of the whole layer in its first pass.
MCP_EXPORT int mcp_fusion_bn_forward_save(int b, int n, int nb, const float *p1, const float *p2, const int *idx, const int *idx2, const float *w1,
                                          const float *b1, const float *w2, const float *b2, const float *w3, const float *b3, float eps, float *bn,
                                          float *var, float *out, int *save_c, float *save_z, float *save_s, void *workspace, size_t workspace_bytes,
                                          mcp_stream_t stream) {
    MCP_CHECK_ARGS(save_c && save_z && save_s);
    return bn_forward(b, n, nb, p1, p2, idx, idx2, w1, b1, w2, b2, w3, b3, eps, bn, var, out, save_c, save_z, save_s, workspace, workspace_bytes, stream);
}

namespace {
int bn_forward(int b, int n, int nb, const float *p1, const float *p2, const int *idx, const int *idx2, const float *w1, const float *b1, const float *w2,
               const float *b2, const float *w3, const float *b3, float eps, float *bn, float *var, float *out, int *save_c, float *save_z, float *save_s,
               void *workspace, size_t workspace_bytes, mcp_stream_t stream) {
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
    hipLaunchKernelGGL(fusion_bn_stats_kernel, dim3(C1 / 16), dim3(256), 0, s, partial, (int)grid, C1, rows, b1, eps, bn + BN_L1, var);
    rc = launch_fwd<2>(total, n, p1, p2, idx, idx2, w1, b1, w2, b2, w3, b3, bn, partial, grid, s);
    if (rc) return rc;
    hipLaunchKernelGGL(fusion_bn_stats_kernel, dim3(C2 / 16), dim3(256), 0, s, partial, (int)grid, C2, rows, b2, eps, bn + BN_L2, var + C1);
    rc = launch_fwd<3>(total, n, p1, p2, idx, idx2, w1, b1, w2, b2, w3, b3, bn, partial, grid, s);
    if (rc) return rc;
    hipLaunchKernelGGL(fusion_bn_stats_kernel, dim3(C3 / 16), dim3(256), 0, s, partial, (int)grid, C3, rows, b3, eps, bn + BN_L3, var + C1 + C2);
    rc = save_c ? launch_fwd<0, true>(total, n, p1, p2, idx, idx2, w1, b1, w2, b2, w3, b3, bn, out, grid, s, save_c, save_z, save_s)
                : launch_fwd<0>(total, n, p1, p2, idx, idx2, w1, b1, w2, b2, w3, b3, bn, out, grid, s);
    mcp_prof_end(MCP_KERNEL_FUSION, s);
    return rc;
}
}  // namespace

MCP_EXPORT size_t mcp_fusion_bn_grad_workspace_bytes(int b, int n) {
    if (b <= 0 || n <= 0) return 0;
    return bwd_workspace_floats((long long)b * n) * sizeof(float);
}

// The layer on batch statistics, backward (one reference call; bn as mcp_fusion_bn_forward left it).  grad_out (B,N,3).  Scratch the
// caller provides: row_c (rows int32), row_dy, row_a (rows floats), dy2, dy1 (rows x 64 floats), rows = b * n * 64.  Writes grad_p1
// (B,N,3), grad_nb (B,N,64,3) (for the caller's scatter into dL/dp2), grad_weights (12800 floats, mcp_fusion_grad's layout; the
// conv-bias entries are 0: a bias in front of a batch-statistics BatchNorm has no gradient) and grad_affine (512 floats: per layer
// dgamma | dbeta -- 64 | 64, 64 | 64, 128 | 128).
namespace {
int bn_backward(int b, int n, int nb, const float *p1, const float *p2, const int *idx, const int *idx2, const float *w1, const float *b1, const float *w2,
                const float *b2, const float *w3, const float *b3, const float *bn, const float *grad_out, const int *save_c, const float *save_z,
                const float *save_s, int *row_c, float *row_dy, float *row_a, float *dy2, float *dy1, float *grad_p1, float *grad_nb, float *grad_weights,
                float *grad_affine, void *workspace, size_t workspace_bytes, mcp_stream_t stream);
}

MCP_EXPORT int mcp_fusion_bn_backward(int b, int n, int nb, const float *p1, const float *p2, const int *idx, const int *idx2, const float *w1,
                                      const float *b1, const float *w2, const float *b2, const float *w3, const float *b3, const float *bn,
                                      const float *grad_out, int *row_c, float *row_dy, float *row_a, float *dy2, float *dy1, float *grad_p1,
                                      float *grad_nb, float *grad_weights, float *grad_affine, void *workspace, size_t workspace_bytes,
                                      mcp_stream_t stream) {
    return bn_backward(b, n, nb, p1, p2, idx, idx2, w1, b1, w2, b2, w3, b3, bn, grad_out, nullptr, nullptr, nullptr, row_c, row_dy, row_a, dy2, dy1, grad_p1,
                       grad_nb, grad_weights, grad_affine, workspace, workspace_bytes, stream);
}

// mcp_fusion_bn_backward with what mcp_fusion_bn_forward_save kept: the first pass reads (score, c*, zhat3) instead of re-evaluating the layer.
MCP_EXPORT int mcp_fusion_bn_backward_saved(int b, int n, int nb, const float *p1, const float *p2, const int *idx, const int *idx2, const float *w1,
                                            const float *b1, const float *w2, const float *b2, const float *w3, const float *b3, const float *bn,
                                            const float *grad_out, const int *save_c, const float *save_z, const float *save_s, int *row_c, float *row_dy,
                                            float *row_a, float *dy2, float *dy1, float *grad_p1, float *grad_nb, float *grad_weights, float *grad_affine,
                                            void *workspace, size_t workspace_bytes, mcp_stream_t stream) {
    MCP_CHECK_ARGS(save_c && save_z && save_s);
    return bn_backward(b, n, nb, p1, p2, idx, idx2, w1, b1, w2, b2, w3, b3, bn, grad_out, save_c, save_z, save_s, row_c, row_dy, row_a, dy2, dy1, grad_p1,
                       grad_nb, grad_weights, grad_affine, workspace, workspace_bytes, stream);
}

namespace {
int bn_backward(int b, int n, int nb, const float *p1, const float *p2, const int *idx, const int *idx2, const float *w1, const float *b1, const float *w2,
                const float *b2, const float *w3, const float *b3, const float *bn, const float *grad_out, const int *save_c, const float *save_z,
                const float *save_s, int *row_c, float *row_dy, float *row_a, float *dy2, float *dy1, float *grad_p1, float *grad_nb, float *grad_weights,
                float *grad_affine, void *workspace, size_t workspace_bytes, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && p1 && p2 && idx && w1 && b1 && w2 && b2 && w3 && b3 && bn && grad_out && row_c && row_dy && row_a && dy2 && dy1 &&
                   grad_p1 && grad_nb && grad_weights && grad_affine && workspace);
    if (nb != NB) return MCP_ERR_UNSUPPORTED;
    if ((((uintptr_t)dy2) | ((uintptr_t)dy1) | ((uintptr_t)workspace)) & 15) return MCP_ERR_BAD_ARG;
    const long long total = (long long)b * n;
    if (workspace_bytes < bwd_workspace_floats(total) * sizeof(float)) return MCP_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    float *ws = static_cast<float *>(workspace);
    const size_t parts_cap = (size_t)bwd_grid(total, 2);
    float *partial = ws, *sums3 = ws + parts_cap * B2_G, *sums2 = sums3 + 256, *sums1 = sums2 + 128;
    uint4 *w3t = reinterpret_cast<uint4 *>(sums1 + 128);
    const float inv_rows = (float)(1.0 / ((double)total * NB));
    // grad_weights layout (mcp_fusion_grad): dW1 (64,4) | db1 | dW2 (64,64) | db2 | dW3 (128,64) | db3
    constexpr int G_W1 = 0, G_W2 = 320, G_B2 = G_W2 + 4096, G_W3 = G_B2 + 64, G_B3 = G_W3 + 8192, G_FLOATS = G_B3 + 128;
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        const void *kerns[4] = {reinterpret_cast<const void *>(fusion_bn_b1_kernel), reinterpret_cast<const void *>(fusion_bn_b2_kernel),
                                reinterpret_cast<const void *>(fusion_bn_b3_kernel), reinterpret_cast<const void *>(fusion_bn_b4_kernel)};
        for (const void *k : kerns) {
            const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
        }
        attr_once.done();
    }
    mcp_prof_begin(MCP_KERNEL_FUSION, s);
    (void)hipMemsetAsync(grad_weights, 0, G_FLOATS * sizeof(float), s);
    hipLaunchKernelGGL(transposed_image_kernel, dim3(16), dim3(256), 0, s, w3t, w3, C2, C3);
    {   // B1
        const unsigned grid = bwd_grid(total, 2);
        const size_t lds = (size_t)L_F32 * 4 + (size_t)(W2_U4 + W3_U4) * 16 + (size_t)WAVES * B1_SCR * 4;
        if (save_c)
            hipLaunchKernelGGL(fusion_bn_b1_saved_kernel, dim3(grid), dim3(64 * WAVES), 0, s, total, n, p2, idx, idx2, grad_out, save_c, save_z, save_s, row_c,
                               row_dy, row_a, partial);
        else
            hipLaunchKernelGGL(fusion_bn_b1_kernel, dim3(grid), dim3(64 * WAVES), lds, s, total, n, p1, p2, idx, idx2, w1, b1, w2, b2, w3, b3, bn, grad_out, row_c,
                               row_dy, row_a, partial);
        hipLaunchKernelGGL(sum_partials_kernel, dim3(16), dim3(256), 0, s, partial, (int)grid, 256, 256, sums3);
    }
    {   // B2
        const unsigned grid = bwd_grid(total, 1);
        hipLaunchKernelGGL(fusion_bn_b2_kernel, dim3(grid), dim3(64 * WAVES), B2_LDS, s, total, n, p1, p2, idx, idx2, w1, b1, w2, b2, w3, b3, bn, sums3, inv_rows,
                           w3t, row_c, row_dy, dy2, partial);
        hipLaunchKernelGGL(sum_partials_kernel, dim3(C3 * C2 / 16), dim3(256), 0, s, partial, (int)grid, B2_G, C3 * C2, grad_weights + G_W3);
        hipLaunchKernelGGL(sum_partials_kernel, dim3(2 * C2 / 16), dim3(256), 0, s, partial + C3 * C2, (int)grid, B2_G, 2 * C2, sums2);
    }
    {   // B3
        const unsigned grid = bwd_grid(total, 1);
        hipLaunchKernelGGL(fusion_bn_b3_kernel, dim3(grid), dim3(64 * WAVES), B3_LDS, s, total, n, p1, p2, idx, idx2, w1, b1, w2, b2, b3, bn, sums2, inv_rows, dy2,
                           dy1, partial);
        hipLaunchKernelGGL(sum_partials_kernel, dim3(C2 * C1 / 16), dim3(256), 0, s, partial, (int)grid, B3_G, C2 * C1, grad_weights + G_W2);
        hipLaunchKernelGGL(sum_partials_kernel, dim3(2 * C1 / 16), dim3(256), 0, s, partial + C2 * C1, (int)grid, B3_G, 2 * C1, sums1);
    }
    {   // B4
        const unsigned grid = bwd_grid(total, 2);
        hipLaunchKernelGGL(fusion_bn_b4_kernel, dim3(grid), dim3(64 * WAVES), B4_LDS, s, total, n, p1, p2, idx, idx2, w1, b1, b2, b3, bn, sums1, inv_rows, dy1,
                           row_a, grad_out, grad_p1, grad_nb, partial);
        hipLaunchKernelGGL(sum_partials_kernel, dim3(B4_G / 16), dim3(256), 0, s, partial, (int)grid, B4_G, B4_G, grad_weights + G_W1);
    }
    // grad_affine: per layer dgamma (= sum dy' zhat) | dbeta (= sum dy')
    (void)hipMemcpyAsync(grad_affine + 0, sums1 + C1, C1 * sizeof(float), hipMemcpyDeviceToDevice, s);
    (void)hipMemcpyAsync(grad_affine + C1, sums1, C1 * sizeof(float), hipMemcpyDeviceToDevice, s);
    (void)hipMemcpyAsync(grad_affine + 2 * C1, sums2 + C2, C2 * sizeof(float), hipMemcpyDeviceToDevice, s);
    (void)hipMemcpyAsync(grad_affine + 2 * C1 + C2, sums2, C2 * sizeof(float), hipMemcpyDeviceToDevice, s);
    (void)hipMemcpyAsync(grad_affine + 2 * C1 + 2 * C2, sums3 + C3, C3 * sizeof(float), hipMemcpyDeviceToDevice, s);
    (void)hipMemcpyAsync(grad_affine + 2 * C1 + 2 * C2 + C3, sums3, C3 * sizeof(float), hipMemcpyDeviceToDevice, s);
    mcp_prof_end(MCP_KERNEL_FUSION, s);
    return mcp_launch_status();
}
}  // namespace
