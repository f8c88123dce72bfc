// norm.hip -- row normalisation with the additions in front of it, for the dense caller chains (gfx950).
//
// EI_Crossformer (mocopci.py:58-151) normalises its two inputs four times (query_norm / feat_norm of Injector and Extractor) and
// the Extractor's residual sum once more (ffn_norm): five torch layer-norm launches plus the additions.  The affine part of a
// LayerNorm in front of a Linear folds into that Linear, so what remains per site is the plain normalisation
//     out[r, :] = (z - mean(z)) * rsqrt(var(z) + eps),   z = x[r, :] (+ y[r, :]) (+ bias)
// -- one kernel, one wave per row, the row held in registers (C <= 1024), biased variance as nn.LayerNorm.
#include "common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int PER>  // floats per lane: C <= 64 * PER
__global__ __launch_bounds__(256) void add_layernorm_kernel(long long rows, int c, const float *__restrict__ x, long long xs,
                                                            const float *__restrict__ y, long long ys, const float *__restrict__ bias,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                            float *__restrict__ out, long long os) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float z[PER];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int ch = lane + 64 * j;
        float v = 0.f;
        if (ch < c) {
            v = x[r * xs + ch];
            if (y) v += y[r * ys + ch];
            if (bias) v += bias[ch];
        }
        z[j] = v;
        sum += v;
    }
    const float mean = wave_sum(sum) / (float)c;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const float d = (lane + 64 * j < c) ? z[j] - mean : 0.f;
        z[j] = d;
        sq = __builtin_fmaf(d, d, sq);
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)c + eps);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int ch = lane + 64 * j;
        if (ch < c) {
            float v = z[j] * rstd;
            if (gamma) v = __builtin_fmaf(v, gamma[ch], beta ? beta[ch] : 0.f);
            out[r * os + ch] = v;
        }
    }
}

}  // namespace

MCP_EXPORT int mcp_add_layernorm(long long rows, int c, const float *x, long long x_stride, const float *y, long long y_stride,
                                 const float *bias, const float *gamma, const float *beta, float eps, float *out, long long out_stride,
                                 mcp_stream_t stream) {
    MCP_CHECK_ARGS(rows > 0 && c > 0 && x && out && x_stride >= c && out_stride >= c && (!y || y_stride >= c));
    if (c > 1024) return MCP_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (c <= 64) hipLaunchKernelGGL(add_layernorm_kernel<1>, grid, dim3(256), 0, s, rows, c, x, x_stride, y, y_stride, bias, gamma, beta, eps, out, out_stride);
    else if (c <= 128) hipLaunchKernelGGL(add_layernorm_kernel<2>, grid, dim3(256), 0, s, rows, c, x, x_stride, y, y_stride, bias, gamma, beta, eps, out, out_stride);
    else if (c <= 256) hipLaunchKernelGGL(add_layernorm_kernel<4>, grid, dim3(256), 0, s, rows, c, x, x_stride, y, y_stride, bias, gamma, beta, eps, out, out_stride);
    else hipLaunchKernelGGL(add_layernorm_kernel<16>, grid, dim3(256), 0, s, rows, c, x, x_stride, y, y_stride, bias, gamma, beta, eps, out, out_stride);
    return mcp_launch_status();
}
