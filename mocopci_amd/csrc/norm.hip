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

// ---- inputs of Multi_Frame_Att in one pass (r3) ------------------------------------------------------------------------------
// Multiframe_Attention (mocopci.py:200-208) stacks the flow embeddings of the three iterations with time codes, the block's norm1
// (eval BatchNorm: a per-channel affine map) normalises them, and its attention pairs frame f with frame R-1-f of the flipped
// stack.  In torch ops that was: zero-fill + index_put of the computed members, a permuted add of the time codes, the affine map,
// a flip, and three row gathers when only some (sample, frame) rows are read -- eight launches over the same few MB.  Here, per
// output row r: x = fea[src_self[r]] + te_self[r];  xn = scale * x + shift;  xr = scale * (fea[src_partner[r]] + te_partner[r]) + shift.
namespace {
__global__ __launch_bounds__(256) void mfa_prepare_kernel(long long total4, int nc4, int c4, const float4 *__restrict__ fea,
                                                          const int *__restrict__ src_self, const int *__restrict__ src_partner,
                                                          const float4 *__restrict__ te_self, const float4 *__restrict__ te_partner,
                                                          const float4 *__restrict__ scale, const float4 *__restrict__ shift,
                                                          float4 *__restrict__ x, float4 *__restrict__ xn, float4 *__restrict__ xr) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long long)gridDim.x * 256) {
        const int r = (int)(e / nc4);
        const int in = (int)(e - (long long)r * nc4), ch = in % c4;
        const float4 sc = scale[ch], sh = shift[ch];
        const float4 a = fea[(long long)src_self[r] * nc4 + in], ta = te_self[r * c4 + ch];
        const float4 b = fea[(long long)src_partner[r] * nc4 + in], tb = te_partner[r * c4 + ch];
        const float4 xv = make_float4(a.x + ta.x, a.y + ta.y, a.z + ta.z, a.w + ta.w);
        const float4 pv = make_float4(b.x + tb.x, b.y + tb.y, b.z + tb.z, b.w + tb.w);
        x[e] = xv;
        xn[e] = make_float4(__builtin_fmaf(xv.x, sc.x, sh.x), __builtin_fmaf(xv.y, sc.y, sh.y), __builtin_fmaf(xv.z, sc.z, sh.z), __builtin_fmaf(xv.w, sc.w, sh.w));
        xr[e] = make_float4(__builtin_fmaf(pv.x, sc.x, sh.x), __builtin_fmaf(pv.y, sc.y, sh.y), __builtin_fmaf(pv.z, sc.z, sh.z), __builtin_fmaf(pv.w, sc.w, sh.w));
    }
}
}  // namespace

MCP_EXPORT int mcp_mfa_prepare(int rows, int n, int c, const float *fea, const int *src_self, const int *src_partner, const float *te_self,
                               const float *te_partner, const float *scale, const float *shift, float *x, float *xn, float *xr,
                               mcp_stream_t stream) {
    MCP_CHECK_ARGS(rows > 0 && n > 0 && c > 0 && fea && src_self && src_partner && te_self && te_partner && scale && shift && x && xn && xr);
    if (c & 3) return MCP_ERR_UNSUPPORTED;
    if ((((uintptr_t)fea) | ((uintptr_t)te_self) | ((uintptr_t)te_partner) | ((uintptr_t)scale) | ((uintptr_t)shift) | ((uintptr_t)x) | ((uintptr_t)xn) |
         ((uintptr_t)xr)) & 15)
        return MCP_ERR_BAD_ARG;
    const int c4 = c / 4, nc4 = n * c4;
    const long long total4 = (long long)rows * nc4;
    const unsigned grid = (unsigned)min((total4 + 255) / 256, 1LL << 16);
    hipLaunchKernelGGL(mfa_prepare_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, total4, nc4, c4, reinterpret_cast<const float4 *>(fea), src_self,
                       src_partner, reinterpret_cast<const float4 *>(te_self), reinterpret_cast<const float4 *>(te_partner),
                       reinterpret_cast<const float4 *>(scale), reinterpret_cast<const float4 *>(shift), reinterpret_cast<float4 *>(x),
                       reinterpret_cast<float4 *>(xn), reinterpret_cast<float4 *>(xr));
    return mcp_launch_status();
}
