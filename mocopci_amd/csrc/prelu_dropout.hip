// prelu_dropout.hip -- PReLU followed by dropout on the hidden activation of Mlp_T, forward and backward, for gfx950 (net.train():
// mocopci.py:1558-1565 applies `self.act` then `self.drop` to the (rows, 4C) output of fc1 + dwconv; :1592-1595 likewise).
//
// The reference runs F.prelu and F.dropout as two elementwise passes and keeps the activation, the mask and the dropped activation for
// autograd, whose PReLU backward is a two-output elementwise kernel (dz and a per-element slope gradient) followed by a reduction of the
// per-element slope gradients: at the pipeline's shape (196608 rows x 256 channels = 201 MB per tensor) 1.05 ms + 0.11 ms per layer,
// five layers per step (round 5 trace).  Here
//     y_i  = m_i * (z_i > 0 ? z_i : a z_i)                                              one pass: read z, write y
//     dz_i = g_i m_i (z_i > 0 ? 1 : a),   da = sum_i g_i m_i min(z_i, 0)                 one pass: read z and g, write dz
// with m_i = 1 / (1 - p) or 0 from a counter-based hash of (seed, i): the backward regenerates the mask, nothing but z is kept.
// The slope gradient is summed per thread, per workgroup (LDS, fixed order) and over the workgroups by a second kernel in workgroup
// order; the grid is a function of the element count alone, so da repeats bit for bit on any device.
// HBM-bound: 8 bytes per element forward, 12 backward.
#include "common.h"

namespace {

constexpr int BLK = 256, VEC = 4, MAX_WGS = 2048;

// keep / drop of element i: the mixer of attention_grad.hip's drop_scale on the two halves of the 64-bit element index
__device__ __forceinline__ float drop_scale(uint32_t seed, unsigned long long i, uint32_t threshold, float inv_keep) {
    uint32_t x = seed ^ ((uint32_t)i * 0x9E3779B1u) ^ ((uint32_t)(i >> 32) * 0x85EBCA77u);
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x >= threshold ? inv_keep : 0.f;
}

__global__ __launch_bounds__(BLK) void prelu_dropout_kernel(long long total, const float *__restrict__ z, const float *__restrict__ slope,
                                                           uint32_t seed, uint32_t threshold, float inv_keep, float *__restrict__ out, int vec) {
    const float a = *slope;
    const long long stride = (long long)gridDim.x * BLK;
    if (vec) {
        const long long quads = total / VEC;
        for (long long e = (long long)blockIdx.x * BLK + threadIdx.x; e < quads; e += stride) {
            const float4 v = reinterpret_cast<const float4 *>(z)[e];
            float4 o;
            o.x = (v.x > 0.f ? v.x : a * v.x) * drop_scale(seed, 4 * e + 0, threshold, inv_keep);
            o.y = (v.y > 0.f ? v.y : a * v.y) * drop_scale(seed, 4 * e + 1, threshold, inv_keep);
            o.z = (v.z > 0.f ? v.z : a * v.z) * drop_scale(seed, 4 * e + 2, threshold, inv_keep);
            o.w = (v.w > 0.f ? v.w : a * v.w) * drop_scale(seed, 4 * e + 3, threshold, inv_keep);
            reinterpret_cast<float4 *>(out)[e] = o;
        }
    } else {
        for (long long e = (long long)blockIdx.x * BLK + threadIdx.x; e < total; e += stride) {
            const float v = z[e];
            out[e] = (v > 0.f ? v : a * v) * drop_scale(seed, e, threshold, inv_keep);
        }
    }
}

__global__ __launch_bounds__(BLK) void prelu_dropout_grad_kernel(long long total, const float *__restrict__ z, const float *__restrict__ slope,
                                                                const float *__restrict__ g, uint32_t seed, uint32_t threshold, float inv_keep,
                                                                float *__restrict__ dz, float *__restrict__ partial, int vec) {
    __shared__ float red[BLK];
    const float a = *slope;
    const long long stride = (long long)gridDim.x * BLK;
    float da = 0.f;
    auto one = [&](float v, float gv, unsigned long long i) {
        const float gm = gv * drop_scale(seed, i, threshold, inv_keep);
        da += v > 0.f ? 0.f : gm * v;
        return v > 0.f ? gm : a * gm;
    };
    if (vec) {
        const long long quads = total / VEC;
        for (long long e = (long long)blockIdx.x * BLK + threadIdx.x; e < quads; e += stride) {
            const float4 v = reinterpret_cast<const float4 *>(z)[e], gv = reinterpret_cast<const float4 *>(g)[e];
            float4 o;
            o.x = one(v.x, gv.x, 4 * e + 0);
            o.y = one(v.y, gv.y, 4 * e + 1);
            o.z = one(v.z, gv.z, 4 * e + 2);
            o.w = one(v.w, gv.w, 4 * e + 3);
            reinterpret_cast<float4 *>(dz)[e] = o;
        }
    } else {
        for (long long e = (long long)blockIdx.x * BLK + threadIdx.x; e < total; e += stride) dz[e] = one(z[e], g[e], e);
    }
    red[threadIdx.x] = da;
    __syncthreads();
#pragma unroll
    for (int o = BLK / 2; o > 0; o >>= 1) {   // a fixed tree: thread t adds thread t + o
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// the workgroups' partial slope gradients, added in workgroup order by one wave (lane l takes partials l, l + 64, ...; then a fixed tree)
__global__ __launch_bounds__(64) void prelu_dropout_reduce_kernel(const float *__restrict__ partial, int parts, float *__restrict__ dslope) {
    float v = 0.f;
    for (int i = threadIdx.x; i < parts; i += 64) v += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (threadIdx.x == 0) *dslope = v;
}

int wgs_of(long long total) {
    const long long want = (total + (long long)BLK * VEC * 4 - 1) / ((long long)BLK * VEC * 4);   // ~16 elements per thread at least
    return (int)(want < 1 ? 1 : want < MAX_WGS ? want : MAX_WGS);   // a function of the element count alone: the summation order is fixed
}

bool drop_params(float drop_p, uint32_t *threshold, float *inv_keep) {
    if (!(drop_p >= 0.f && drop_p < 1.f)) return false;
    *threshold = (uint32_t)((double)drop_p * 4294967296.0);
    *inv_keep = (float)(1.0 / (1.0 - (double)drop_p));
    return true;
}

}  // namespace

MCP_EXPORT int mcp_prelu_dropout(long long total, const float *z, const float *slope, float drop_p, unsigned seed, float *out, mcp_stream_t stream) {
    MCP_CHECK_ARGS(total > 0 && z && slope && out);
    uint32_t threshold;
    float inv_keep;
    if (!drop_params(drop_p, &threshold, &inv_keep)) return MCP_ERR_BAD_ARG;
    const int vec = !(total % VEC) && !((((uintptr_t)z) | ((uintptr_t)out)) & 15);
    hipLaunchKernelGGL(prelu_dropout_kernel, dim3(wgs_of(total)), dim3(BLK), 0, (hipStream_t)stream, total, z, slope, (uint32_t)seed, threshold, inv_keep,
                       out, vec);
    return mcp_launch_status();
}

MCP_EXPORT size_t mcp_prelu_dropout_grad_workspace_bytes(long long total) { return total > 0 ? (size_t)wgs_of(total) * sizeof(float) : 0; }

MCP_EXPORT int mcp_prelu_dropout_grad(long long total, const float *z, const float *slope, const float *grad_out, float drop_p, unsigned seed,
                                      float *grad_z, float *grad_slope, void *workspace, size_t workspace_bytes, mcp_stream_t stream) {
    MCP_CHECK_ARGS(total > 0 && z && slope && grad_out && grad_z && grad_slope && workspace);
    uint32_t threshold;
    float inv_keep;
    if (!drop_params(drop_p, &threshold, &inv_keep)) return MCP_ERR_BAD_ARG;
    const int wgs = wgs_of(total);
    if (workspace_bytes < (size_t)wgs * sizeof(float)) return MCP_ERR_BAD_ARG;
    const int vec = !(total % VEC) && !((((uintptr_t)z) | ((uintptr_t)grad_out) | ((uintptr_t)grad_z)) & 15);
    float *partial = reinterpret_cast<float *>(workspace);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(prelu_dropout_grad_kernel, dim3(wgs), dim3(BLK), 0, s, total, z, slope, grad_out, (uint32_t)seed, threshold, inv_keep, grad_z,
                       partial, vec);
    const int rc = mcp_launch_status();
    if (rc != MCP_OK) return rc;
    hipLaunchKernelGGL(prelu_dropout_reduce_kernel, dim3(1), dim3(64), 0, s, partial, wgs, grad_slope);
    return mcp_launch_status();
}
