// ptblock.hip -- fused Point-Transformer vector attention (TransformerBlock.forward, models/pointT_layer2.py:58-77,
// d_model = 64, k = 16) for gfx950, after the neighbour search and the q/k/v projections:
//     delta_j = fc_delta(xyz_i - xyz_j)                      Linear(3,64) ReLU Linear(64,64)
//     attn_j  = fc_gamma(q_i - k_j + delta_j)                Linear(64,64) ReLU Linear(64,64)
//     res_i   = sum_j softmax_j(attn_j / sqrt(64)) * (v_j + delta_j)          (softmax per channel over the 16 neighbours)
// The reference materialises five (B,N,16,64) tensors and argsorts a full (N,N) distance matrix for the 16
// neighbours.  Here a wave owns TWO points: the 2x16 neighbours sit on the MFMA column (lane & 31), every Linear is
// a chain of v_mfma_f32_32x32x2_f32 whose accumulator tile is the next layer's B operand (as in fusion.hip), gathered
// k / v rows are loaded directly in accumulator layout, and -- because one point's 16 neighbours of one lane-half are
// exactly one 16-lane DPP row -- the per-channel softmax and the weighted sum are DPP row reductions.
// The three 64 -> 64 layers run on the bf16 matrix pipe through the exact three-way operand split of mfma_split.h (weights
// split once by mcp_ptblock_pack, activations in registers); the K = 4 first layer of fc_delta stays on the f32-input MFMA.
#include "common.h"
#include "mfma_split.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int C = 64, KNB = 16, WAVES = 4;
// packed image (floats): wd1 [2][2][64] | wd2, wg1, wg2 each as split pieces [2 tiles][4 k-steps][3 pieces][64 lanes] x uint4
// | bd2, bg1, bg2 each [2][2][16]
constexpr int WSPLIT = 2 * 4 * 3 * 64 * 4;  // floats per 64x64 layer
constexpr int OFF_D1 = 0, OFF_D2 = 256, OFF_G1 = OFF_D2 + WSPLIT, OFF_G2 = OFF_G1 + WSPLIT, OFF_BD2 = OFF_G2 + WSPLIT,
              OFF_BG1 = OFF_BD2 + 64, OFF_BG2 = OFF_BG1 + 64, PACK_FLOATS = OFF_BG2 + 64;

__device__ __forceinline__ int chan_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
    return __uint_as_float(mcp_dpp<CTRL>(__float_as_uint(v)));
}
__device__ __forceinline__ float row_max(float v) {
    v = fmaxf(v, dppf<0xB1>(v));
    v = fmaxf(v, dppf<0x4E>(v));
    v = fmaxf(v, dppf<0x141>(v));
    return fmaxf(v, dppf<0x140>(v));
}
__device__ __forceinline__ float row_sum(float v) {
    v += dppf<0xB1>(v);
    v += dppf<0x4E>(v);
    v += dppf<0x141>(v);
    return v + dppf<0x140>(v);
}

__global__ __launch_bounds__(256) void ptblock_pack_kernel(const float *__restrict__ wd1, const float *__restrict__ bd1,
                                                           const float *__restrict__ wd2, const float *__restrict__ bd2,
                                                           const float *__restrict__ wg1, const float *__restrict__ bg1,
                                                           const float *__restrict__ wg2, const float *__restrict__ bg2,
                                                           float *__restrict__ packed) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < PACK_FLOATS; e += gridDim.x * 256) {
        float v;
        if (e < OFF_D2) {  // [t][s][lane]: columns (dx,dy | dz,1) of [wd1 | bd1]
            const int lane = e & 63, s = (e >> 6) & 1, t = e >> 7;
            const int row = 32 * t + (lane & 31), c = 2 * s + (lane >> 5);
            v = c < 3 ? wd1[row * 3 + c] : bd1[row];
        } else if (e < OFF_BD2) {  // three 64x64 layers: written below as split pieces
            continue;
        } else {  // biases [t][h][r]
            const int f = (e - OFF_BD2) & 63, which = (e - OFF_BD2) >> 6;
            const float *bb = which == 0 ? bd2 : which == 1 ? bg1 : bg2;
            const int r = f & 15, h = (f >> 4) & 1, t = f >> 5;
            v = bb[32 * t + chan_of(r, h)];
        }
        packed[e] = v;
    }
    const int first = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    mcp_split_weights(reinterpret_cast<uint4 *>(packed + OFF_D2), wd2, C, 2, first, stride);
    mcp_split_weights(reinterpret_cast<uint4 *>(packed + OFF_G1), wg1, C, 2, first, stride);
    mcp_split_weights(reinterpret_cast<uint4 *>(packed + OFF_G2), wg2, C, 2, first, stride);
}

// one 64 -> 64 layer on accumulator-layout input x[2]; bias as the initial accumulator
__device__ __forceinline__ void layer64(const float *lds, int off_w, int off_b, int lane, int h, const f32x16 (&x)[2], f32x16 (&y)[2],
                                        bool relu) {
    const uint4 *ws = reinterpret_cast<const uint4 *>(lds + off_w) + lane;
    McpSplit3 xs[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) xs[s] = mcp_split_kstep(x[s >> 1], s & 1);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = lds[off_b + (t * 2 + h) * 16 + r];
        acc = mcp_tile_split<4>(ws + (size_t)t * 4 * 3 * 64, xs, acc);
        if (relu) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f);
        }
        y[t] = acc;
    }
}

__global__ __launch_bounds__(64 * WAVES, 2) void ptblock_kernel(long long total, int n, int rs, const float *__restrict__ xyz,
                                                                 const float *__restrict__ q, const float *__restrict__ kf,
                                                                 const float *__restrict__ vf, const int *__restrict__ idx,
                                                                 const float *__restrict__ packed, float scale_log2e,
                                                                 float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float lds[PACK_FLOATS];
    const int tid = threadIdx.x;
    for (int e = tid; e < PACK_FLOATS / 4; e += 64 * WAVES) reinterpret_cast<float4 *>(lds)[e] = reinterpret_cast<const float4 *>(packed)[e];
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31, sel = col >> 4, j = col & 15;
    const long long pairs = (total + 1) / 2;
    const McpUnits units = mcp_units_by_xcd(pairs, WAVES);   // XCD x takes the x-th eighth of the point pairs (common.h)
    for (long long pp = units.first + wave; pp < units.limit; pp += units.stride) {
        long long p = 2 * pp + sel;
        const bool live = p < total;
        if (!live) p = total - 1;  // odd tail: the second half of the wave recomputes the last point and does not store
        const long long bb = mcp_div(p, n, mcp_fits32(total));
        const int id = idx[p * KNB + j];
        const float *cj = xyz + ((long long)bb * n + id) * 3;
        const float dx = xyz[p * 3 + 0] - cj[0], dy = xyz[p * 3 + 1] - cj[1], dz = xyz[p * 3 + 2] - cj[2];  // xyz_i - xyz_j
        const float in0 = h ? dy : dx, in1 = h ? 1.0f : dz;
        // delta1 = relu(Wd1 d + bd1): K = 4 (dx,dy,dz,1)
        f32x16 d1[2], delta[2], g[2], a1[2], attn[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[OFF_D1 + (t * 2 + 0) * 64 + lane], in0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[OFF_D1 + (t * 2 + 1) * 64 + lane], in1, acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f);
            d1[t] = acc;
        }
        layer64(lds, OFF_D2, OFF_BD2, lane, h, d1, delta, false);
        // g = (q_i - k_j) + delta_j, rows loaded straight into accumulator layout
        // rs: row stride of q / k / v in floats (C when they are separate tensors, 3C when they are the packed projection)
        const float4 *qrow = reinterpret_cast<const float4 *>(q + p * rs);
        const float4 *krow = reinterpret_cast<const float4 *>(kf + ((long long)bb * n + id) * rs);
        const float4 *vrow = reinterpret_cast<const float4 *>(vf + ((long long)bb * n + id) * rs);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int c4 = (32 * t + 8 * gq + 4 * h) >> 2;
                const float4 qq = qrow[c4], kk = krow[c4];
                g[t][4 * gq + 0] = (qq.x - kk.x) + delta[t][4 * gq + 0];
                g[t][4 * gq + 1] = (qq.y - kk.y) + delta[t][4 * gq + 1];
                g[t][4 * gq + 2] = (qq.z - kk.z) + delta[t][4 * gq + 2];
                g[t][4 * gq + 3] = (qq.w - kk.w) + delta[t][4 * gq + 3];
            }
        }
        layer64(lds, OFF_G1, OFF_BG1, lane, h, g, a1, true);
        layer64(lds, OFF_G2, OFF_BG2, lane, h, a1, attn, false);
        // per-channel softmax over the 16 neighbours (one DPP row) and the weighted sum of (v + delta)
        float4 *orow = reinterpret_cast<float4 *>(out + p * C);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float res[16];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 vv = vrow[(32 * t + 8 * gq + 4 * h) >> 2];
                const float vals[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = 4 * gq + u;
                    const float a = attn[t][r] * scale_log2e;
                    const float e = __builtin_amdgcn_exp2f(a - row_max(a));
                    res[r] = row_sum(e * (vals[u] + delta[t][r])) / row_sum(e);
                }
            }
            if (live && j == 0) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    orow[(32 * t + 8 * gq + 4 * h) >> 2] = make_float4(res[4 * gq], res[4 * gq + 1], res[4 * gq + 2], res[4 * gq + 3]);
            }
        }
    }
}

}  // namespace

MCP_EXPORT int mcp_ptblock_packed_floats(void) { return PACK_FLOATS; }

MCP_EXPORT int mcp_ptblock_pack(const float *wd1, const float *bd1, const float *wd2, const float *bd2, const float *wg1,
                                const float *bg1, const float *wg2, const float *bg2, float *packed, mcp_stream_t stream) {
    MCP_CHECK_ARGS(wd1 && bd1 && wd2 && bd2 && wg1 && bg1 && wg2 && bg2 && packed);
    if (((uintptr_t)packed) & 15) return MCP_ERR_BAD_ARG;
    hipLaunchKernelGGL(ptblock_pack_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream, wd1, bd1, wd2, bd2, wg1, bg1, wg2, bg2, packed);
    return mcp_launch_status();
}

MCP_EXPORT int mcp_ptblock_attention(int b, int n, int c, int k, int qkv_stride, const float *xyz, const float *q, const float *kf,
                                     const float *vf, const int *idx, const float *packed, float *out, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && xyz && q && kf && vf && idx && packed && out);
    if (c != C || k != KNB) return MCP_ERR_UNSUPPORTED;
    if (qkv_stride < C || (qkv_stride & 3)) return MCP_ERR_BAD_ARG;
    if ((((uintptr_t)q) | ((uintptr_t)kf) | ((uintptr_t)vf) | ((uintptr_t)out) | ((uintptr_t)packed)) & 15) return MCP_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const long long total = (long long)b * n;
    const long long want = ((total + 1) / 2 + WAVES * 4 - 1) / (WAVES * 4);  // >= 4 point pairs per wave: amortises the weight staging
    const unsigned grid = (unsigned)max(1LL, min(want, 768LL));
    const float scale_log2e = 1.44269504088896340736f / 8.0f;  // softmax(attn / sqrt(64)), pointT_layer2.py:73
    mcp_prof_begin(MCP_KERNEL_PTBLOCK, s);
    hipLaunchKernelGGL(ptblock_kernel, dim3(grid), dim3(64 * WAVES), 0, s, total, n, qkv_stride, xyz, q, kf, vf, idx, packed, scale_log2e, out);
    mcp_prof_end(MCP_KERNEL_PTBLOCK, s);
    return mcp_launch_status();
}
